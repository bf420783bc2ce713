"""The realistic density stressor: tutorials/table_mug.pcd as shipped, NOT voxelised (104 444 points; the reference
voxelises to 3 mm by default but does not have to: candidates_generator.cpp:24-26).  0.03 m normal neighbourhoods of
up to 3987 points (beyond the fast normals kernel), 0.11 m hand neighbourhoods of 26 k on average and 43 k at most
(the global-memory lists of the search), image boxes with thousands of points (the large points kernel).  Normals,
records, images and scores against the oracle."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unvoxelised_table_mug(oracle_mod, lenet15_real):
    xyz = np.load(os.path.join(ROOT, "tests", "golden", "table_mug_xyz.npz"))["xyz"]
    P = len(xyz)
    assert P == 104444
    cam = np.ones((1, P), np.int32)
    vp = np.zeros((1, 3))
    rng = np.random.RandomState(2)
    si = rng.choice(P, 64, replace=False).astype(np.int32)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(lenet15_real)
        ctx.upload_cloud(xyz, np.zeros_like(xyz), cam, vp)
        nrm = ctx.estimate_normals(0.03)                       # Cloud::calculateNormals on the device (retry with the large lists)
        onrm = oracle_mod.estimate_normals(xyz, cam, vp, 0.03)
        assert np.array_equal(nrm, onrm)
        hands, n_cand = ctx.detect(si)
        p = oracle_mod.default_params(15)
        ohands, on_cand, _ = oracle_mod.detect(p, xyz, onrm, cam, vp, si, lenet15_real)
        assert n_cand == on_cand and n_cand > 40
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, xyz, onrm, cam, vp, fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
    finally:
        ctx.close()
