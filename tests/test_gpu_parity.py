"""GPU parity: libgpd_hip.so (through the C-ABI) against the CPU oracle on the same inputs.

Bars (SURVEY.md §9 parity contract): bit-exact validity / finger placement / u8 images;
frames and closing boxes to 1e-12 relative; scores within 1e-4 absolute.
"""
import numpy as np
import pytest

from gpd_amd import api, synth

pytestmark = pytest.mark.gpu


def _hands_equal(g, o):
    assert g.shape == o.shape
    assert np.array_equal(g["valid"], o["valid"])
    v = o["valid"].astype(bool)
    assert np.array_equal(g["finger_placement_index"][v], o["finger_placement_index"][v])
    assert np.array_equal(g["half_antipodal"][v], o["half_antipodal"][v])
    assert np.array_equal(g["full_antipodal"][v], o["full_antipodal"][v])
    assert np.array_equal(g["set_index"], o["set_index"]) and np.array_equal(g["slot"], o["slot"])
    for f in ("sample", "frame", "position", "top", "bottom", "center", "grasp_width"):
        a, b = g[f][v], o[f][v]
        assert np.allclose(a, b, rtol=1e-12, atol=1e-15), f
    # invalid slots still carry the frame and the pre-deepening box
    assert np.allclose(g["frame"], o["frame"], rtol=1e-12, atol=1e-15)
    return {f: bool(np.array_equal(g[f][v], o[f][v])) for f in ("frame", "position", "grasp_width", "top")}


@pytest.fixture(scope="module")
def ctx15(lenet15_real):
    c = api.Context(api.default_params(15))
    c.set_lenet_weights(lenet15_real)
    yield c
    c.close()


def test_lenet_scores_match_oracle(ctx15, oracle_mod, lenet15_real):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(37, 60, 60, 15)).astype(np.uint8)
    img[3] = 0
    img[4] = 255
    want = oracle_mod.lenet(img, lenet15_real)
    import ref_cases
    ref_cases.assert_scores(ctx15, img, want)  # fmaf-chain mode: bit-identical; the default split mode: within 1e-4


def test_search_matches_oracle(ctx15, oracle_mod, cloud30k):
    si = synth.sample_indices(cloud30k, 150)
    ctx15.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
    got = ctx15.search(si)
    want = oracle_mod.search(oracle_mod.default_params(15), cloud30k["xyz"], cloud30k["normals"], si)
    exact = _hands_equal(got, want)
    assert want["valid"].sum() > 100
    assert all(exact.values()), exact


@pytest.mark.parametrize("channels", [15, 12, 3, 1])
def test_images_match_oracle(oracle_mod, cloud30k, channels):
    si = synth.sample_indices(cloud30k, 60)
    ctx = api.Context(api.default_params(channels))
    try:
        ctx.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
        p = oracle_mod.default_params(channels)
        hands = oracle_mod.filter_workspace(p, ctx.search(si))
        got, gidx = ctx.images(hands)
        want, widx = oracle_mod.images(p, cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"], hands)
        assert np.array_equal(gidx, widx)
        assert got.shape == want.shape and got.shape[0] > 50
        diff = (got != want)
        assert diff.sum() == 0, "differing pixels: %d of %d (per channel %s)" % (diff.sum(), diff.size, diff.reshape(-1, channels).sum(0))
    finally:
        ctx.close()


def test_detect_end_to_end(ctx15, oracle_mod, cloud30k, lenet15_real):
    si = synth.sample_indices(cloud30k, 80)
    ctx15.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
    hands, n_cand = ctx15.detect(si)
    p = oracle_mod.default_params(15)
    ohands, on_cand, _ = oracle_mod.detect(p, cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"], si, lenet15_real)
    assert n_cand == on_cand and n_cand > 50
    assert np.array_equal(hands["valid"], ohands["valid"])
    v = ohands["valid"].astype(bool)
    assert np.abs(hands["score"][v] - ohands["score"][v]).max() <= 1e-4
