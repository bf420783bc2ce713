"""Samples given by coordinates (Cloud::getSamples, hand_search.cpp:37-39) and
HandSearch::reevaluateHypotheses (hand_search.cpp:66-134, 190-228)."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth


def _offcloud_samples(cl, n, seed=5):
    """Doubles near object points that are neither cloud points nor float32 values."""
    rng = np.random.RandomState(seed)
    si = synth.sample_indices(cl, n)
    return cl["xyz"][si].astype(np.float64) + rng.uniform(-0.004, 0.004, (n, 3)) + 1e-9


# ---- oracle properties (CPU) -------------------------------------------------------------
def test_oracle_xyz_samples_on_cloud_points_equal_index_samples(oracle_mod):
    cl = synth.make_cloud(77, 6000)
    p = oracle_mod.default_params(15)
    si = synth.sample_indices(cl, 40)
    a = oracle_mod.search(p, cl["xyz"], cl["normals"], si)
    b = oracle_mod.search_xyz(p, cl["xyz"], cl["normals"], cl["xyz"][si].astype(np.float64))
    assert a.tobytes() == b.tobytes()


def test_oracle_xyz_samples_keep_the_double_and_drop_frameless(oracle_mod):
    cl = synth.make_cloud(77, 6000)
    p = oracle_mod.default_params(15)
    sm = _offcloud_samples(cl, 30)
    sm[7] = [5.0, 5.0, 5.0]  # nothing within 1 cm: no frame, the sample is dropped (frame_estimator.cpp:55-60)
    h = oracle_mod.search_xyz(p, cl["xyz"], cl["normals"], sm)
    kept = np.delete(sm, 7, axis=0)
    # samples whose 1 cm ball is empty are dropped too; the rest keep their double coordinates, in order
    got = h[:, 0]["sample"]
    assert len(got) <= len(kept) and len(got) > 15
    j = 0
    for row in got:
        while not np.array_equal(kept[j], row):
            j += 1
        j += 1
    assert h["valid"].sum() > 10


def test_oracle_reevaluate_on_the_same_cloud_reproduces_labels(oracle_mod):
    cl = synth.make_cloud(77, 6000)
    p = oracle_mod.default_params(15)
    hands = oracle_mod.search(p, cl["xyz"], cl["normals"], synth.sample_indices(cl, 60)).reshape(-1)
    v = hands[hands["valid"].astype(bool)]
    labels, out = oracle_mod.reevaluate(p, cl["xyz"], cl["normals"], v)
    # the deepen loop's last success is exactly evaluateFingers(points, top, idx): same closing region, same label
    assert np.array_equal(labels, v["full_antipodal"].astype(np.int32))
    # reevaluate sets `half` only for HALF_GRASP (hand_search.cpp:122-124); the search sets it for label >= 1
    assert np.array_equal(out["half_antipodal"].astype(bool), v["half_antipodal"].astype(bool) & ~v["full_antipodal"].astype(bool))
    assert labels.sum() > 0


# ---- HIP path (GPU) ------------------------------------------------------------------------
def _weights(C):
    g = os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


@pytest.mark.gpu
def test_search_and_detect_by_coordinates_match_oracle(oracle_mod, cloud30k):
    cl = cloud30k
    sm = _offcloud_samples(cl, 300)
    sm[11] = [5.0, 5.0, 5.0]
    w = _weights(15)
    p = oracle_mod.default_params(15)
    want = oracle_mod.search_xyz(p, cl["xyz"], cl["normals"], sm)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        got = ctx.search_samples(sm)
        assert got.shape == want.shape and got.tobytes() == want.tobytes()
        hands, n_cand = ctx.detect_samples(sm)
        fw = oracle_mod.filter_workspace(p, want)
        img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], fw)
        sc = oracle_mod.lenet(img, w)
        assert n_cand == len(cand) and n_cand > 300
        assert np.array_equal(hands["valid"], fw["valid"])
        assert np.abs(hands.reshape(-1)[cand]["score"] - sc).max() <= 1e-4
        gi, gc = ctx.images(hands)
        assert np.array_equal(gc, cand) and np.array_equal(gi, img)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_reevaluate_matches_oracle(oracle_mod, cloud30k):
    cl = cloud30k
    p = oracle_mod.default_params(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands = ctx.search(synth.sample_indices(cl, 400)).reshape(-1)
        v = hands[hands["valid"].astype(bool)]
        # same cloud: the labels are the antipodal flags of the search
        labels, out = ctx.reevaluate(v)
        assert np.array_equal(labels, v["full_antipodal"].astype(np.int32)) and labels.sum() > 0
        # a "ground truth" cloud: every second point, other normals' neighbourhoods -> different labels
        gt_xyz, gt_n = cl["xyz"][::2].copy(), cl["normals"][::2].copy()
        ctx.upload_cloud(gt_xyz, gt_n)
        mixed = hands[:1500].copy()  # valid and invalid records alike (index -1 -> label 0)
        labels, out = ctx.reevaluate(mixed)
        wl, wout = oracle_mod.reevaluate(p, gt_xyz, gt_n, mixed)
        assert np.array_equal(labels, wl) and out.tobytes() == wout.tobytes()
        assert 0 < labels.sum() < len(labels)
        assert not np.array_equal(labels, mixed["full_antipodal"].astype(np.int32))
        # the search buffers were reused: imaging hands of the earlier search must fail loudly
        with pytest.raises(api.GpdHipError):
            ctx.images(hands.reshape(-1, 8)[:4])
        l0, h0 = ctx.reevaluate(mixed[:0])
        assert len(l0) == 0
    finally:
        ctx.close()
