"""The oracle (CPU) and the HIP path (GPU) against the committed golden case."""
import os

import numpy as np
import pytest

from gpd_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _case(C):
    d = np.load(os.path.join(GOLD, "case_small_c%d.npz" % C))
    cl = synth.make_cloud(4242, 8000)
    g = os.path.join(GOLD, "lenet%d_params.npz" % C)
    w = synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None)
    return d, cl, w


@pytest.mark.parametrize("C", [15, 12, 3, 1])
def test_oracle_reproduces_golden(oracle_mod, C):
    d, cl, w = _case(C)
    p = oracle_mod.default_params(C)
    hands = oracle_mod.search(p, cl["xyz"], cl["normals"], d["sample_indices"])
    assert hands.view(np.uint8).tobytes() == d["hands"].tobytes()
    hf = oracle_mod.filter_workspace(p, hands.copy())
    assert np.array_equal(hf["valid"], d["hands_filtered_valid"])
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hf)
    assert np.array_equal(cand, d["cand_index"]) and np.array_equal(img, d["images"])
    assert np.array_equal(oracle_mod.lenet(img, w), d["scores"])


@pytest.mark.gpu
@pytest.mark.parametrize("C", [15, 12, 3, 1])
def test_hip_reproduces_golden(C):
    from gpd_amd import api
    d, cl, w = _case(C)
    ctx = api.Context(api.default_params(C))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands = ctx.search(d["sample_indices"])
        want = d["hands"].view(api.HAND_DTYPE).reshape(hands.shape)
        assert np.array_equal(hands["valid"], want["valid"])
        v = want["valid"].astype(bool)
        assert np.array_equal(hands["finger_placement_index"][v], want["finger_placement_index"][v])
        for f in ("frame", "position", "top", "bottom", "center", "grasp_width"):
            assert np.allclose(hands[f][v], want[f][v], rtol=1e-12, atol=1e-15)
        hands["valid"] = d["hands_filtered_valid"]
        img, cand = ctx.images(hands)
        assert np.array_equal(cand, d["cand_index"])
        assert np.array_equal(img, d["images"])
        assert np.abs(ctx.score(img) - d["scores"]).max() <= 1e-4
    finally:
        ctx.close()
