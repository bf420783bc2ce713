"""BASELINE.json configs[0]: tutorials/krylon.pcd + cfg/eigen_params.cfg with num_samples = 500,
15 channels.  Preprocessing as CandidatesGenerator::preprocessPointCloud does it
(candidates_generator.cpp:14-37): voxelise(0.003) -> normals(radius 0.03) -> subsample."""
import os

import numpy as np
import pytest

from gpd_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def krylon():
    return np.load(os.path.join(GOLD, "krylon_xyz.npz"))["xyz"]


def _samples(n, m, seed=500):
    return np.ascontiguousarray(np.random.RandomState(seed).permutation(n)[:m].astype(np.int32))


def test_voxelize_known_answer(oracle_mod, krylon):
    """cloud.cpp:286-348 under libstdc++ std::set keeps 3366 of the 4467 points, the
    first ones from input indices 4466, 4464, 4459, 4458, 4457."""
    v, src = oracle_mod.voxelize(krylon, 0.003)
    assert len(v) == 3366
    assert src[:5].tolist() == [4466, 4464, 4459, 4458, 4457]
    mn = krylon.min(0)
    cell = np.floor((krylon[src] - mn) / np.float32(0.003))
    assert np.array_equal(v, mn + np.float32(0.003) * cell.astype(np.float32))
    assert len(np.unique(cell, axis=0)) == 2373  # the true number of occupied voxels (the comparator is not an ordering)


def test_normals_against_numpy_pca(oracle_mod, krylon):
    v, _ = oracle_mod.voxelize(krylon, 0.003)
    n = oracle_mod.estimate_normals(v)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    # every normal faces the camera at the origin: n . (p - vp) < 0
    assert np.all(np.einsum("ij,ij->i", n.astype(np.float64), v.astype(np.float64)) < 0)
    from scipy.spatial import cKDTree
    tree = cKDTree(v.astype(np.float64))
    for i in (0, 17, 1234, 3365):
        nb = v[tree.query_ball_point(v[i].astype(np.float64), 0.03)].astype(np.float64)
        w, U = np.linalg.eigh(np.cov(nb.T, bias=True))
        assert abs(abs(np.dot(U[:, 0], n[i])) - 1.0) < 1e-4


def test_config1_cpu_plumbing(oracle_mod, krylon, lenet15_real):
    """The reference's own CPU-runnable case, end to end on the CPU oracle."""
    v, _ = oracle_mod.voxelize(krylon, 0.003)
    n = oracle_mod.estimate_normals(v)
    si = _samples(len(v), 500)
    p = oracle_mod.default_params(15)
    hands, n_cand, times = oracle_mod.detect(p, v, n, np.ones((1, len(v)), np.int32), np.zeros((1, 3)), si, lenet15_real)
    assert hands.shape == (500, 8) and n_cand > 100
    sc = hands["score"][hands["valid"].astype(bool)]
    assert np.isfinite(sc).all() and sc.std() > 0


@pytest.mark.gpu
def test_config1_on_gpu_matches_oracle(oracle_mod, krylon, lenet15_real):
    from gpd_amd import api
    v, _ = oracle_mod.voxelize(krylon, 0.003)
    want_n = oracle_mod.estimate_normals(v)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(lenet15_real)
        ctx.upload_cloud(v, np.zeros_like(v))
        got_n = ctx.estimate_normals(0.03)
        assert np.array_equal(got_n, want_n), "normals differ in %d of %d components" % ((got_n != want_n).sum(), got_n.size)
        si = _samples(len(v), 500)
        hands, n_cand = ctx.detect(si)  # uses the normals left on the device
        p = oracle_mod.default_params(15)
        oh, on, _ = oracle_mod.detect(p, v, want_n, np.ones((1, len(v)), np.int32), np.zeros((1, 3)), si, lenet15_real)
        assert n_cand == on and np.array_equal(hands["valid"], oh["valid"])
        vmask = oh["valid"].astype(bool)
        assert np.array_equal(hands["finger_placement_index"][vmask], oh["finger_placement_index"][vmask])
        assert np.abs(hands["score"][vmask] - oh["score"][vmask]).max() <= 1e-4
    finally:
        ctx.close()


@pytest.mark.gpu
def test_normals_on_synthetic_cloud(oracle_mod, cloud30k):
    from gpd_amd import api
    ctx = api.Context(api.default_params(3))
    try:
        ctx.upload_cloud(cloud30k["xyz"], np.zeros_like(cloud30k["xyz"]))
        got = ctx.estimate_normals(0.03)
        want = oracle_mod.estimate_normals(cloud30k["xyz"])
        assert np.array_equal(got, want)
        # sanity: close to the analytic normals on the table plane
        tab = ~cloud30k["is_object"]
        assert np.abs(np.abs(np.einsum("ij,ij->i", got[tab], cloud30k["normals"][tab])) - 1).mean() < 0.05
    finally:
        ctx.close()
