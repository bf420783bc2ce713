"""Clustering::findClusters (clustering.cpp:5-105): the device kernel (gpd_hip_find_clusters), reached through the C-ABI
and through the host mirror's Clustering class, vs the oracle vs an independent numpy form."""
import numpy as np
import pytest

from gpd_amd import api, hostlib, synth

pytestmark = pytest.mark.gpu


def _numpy_clusters(hands, scores, min_inliers, remove_inliers):
    pos = hands["position"].astype(np.float64)
    axis = hands["frame"].reshape(-1, 3, 3)[:, :, 2]
    n = len(hands)
    used = np.zeros(n, bool)
    out = []
    for i in range(n):
        inl = []
        for j in range(n):
            if i == j or (remove_inliers and used[j]):
                continue
            d = pos[i] - pos[j]
            proj = d - axis[i] * np.dot(axis[i], d)
            if (abs(np.dot(axis[i], axis[j])) > np.cos(np.deg2rad(12.0)) and np.linalg.norm(d) <= 0.05
                    and np.linalg.norm(proj) <= 0.005):
                inl.append(j)
                if remove_inliers:
                    used[j] = True
        if len(inl) >= min_inliers and len(inl) > 0:
            s = scores[inl]
            lb = s.mean() - 2.576 * s.std() / np.sqrt(len(inl))
            out.append((i, pos[inl].mean(0), lb))
    return out


@pytest.fixture(scope="module")
def scored_hands(oracle_mod):
    small_cloud = synth.make_cloud(77, 6000)
    p = oracle_mod.default_params(15)
    si = synth.sample_indices(small_cloud, 60)
    hands = oracle_mod.search(p, small_cloud["xyz"], small_cloud["normals"], si).reshape(-1)
    hands = hands[hands["valid"].astype(bool)]
    # duplicate some hands with small offsets along their own axis so that clusters exist
    rng = np.random.RandomState(3)
    extra = hands[rng.randint(0, len(hands), 80)].copy()
    ax = extra["frame"].reshape(-1, 3, 3)[:, :, 2]
    extra["position"] += ax * rng.uniform(-0.03, 0.03, (len(extra), 1)) + rng.normal(0, 0.001, (len(extra), 3))
    hands = np.concatenate([hands, extra])
    scores = rng.normal(0, 3, len(hands))
    return hands, scores


@pytest.mark.parametrize("min_inliers,remove", [(1, False), (2, False), (1, True), (3, True)])
def test_host_clusters_match_oracle(oracle_mod, scored_hands, min_inliers, remove):
    hands, scores = scored_hands
    want, wsc, wsrc = oracle_mod.find_clusters(hands, scores, min_inliers, remove)
    assert len(want) > 3
    ctx = api.Context(api.default_params(15))
    try:
        routes = [hostlib.find_clusters(hands, scores, min_inliers, remove),  # Clustering class of the host mirror -> C-ABI
                  ctx.find_clusters(hands, scores, min_inliers, remove)]      # the C-ABI directly
    finally:
        ctx.close()
    for got, gsc, gsrc in routes:
        assert np.array_equal(gsrc, wsrc)
        assert gsc.tobytes() == wsc.tobytes()
        assert got["position"].tobytes() == want["position"].tobytes()
        assert np.array_equal(got["frame"], hands["frame"][wsrc])
        assert np.array_equal(got["score"], wsc.astype(np.float32))
    ref = _numpy_clusters(hands, scores, min_inliers, remove)
    assert [r[0] for r in ref] == list(wsrc)
    assert np.allclose(np.array([r[1] for r in ref]), want["position"], atol=1e-12)
    assert np.allclose(np.array([r[2] for r in ref]), wsc, atol=1e-9)


def test_cluster_edge_cases(oracle_mod, scored_hands):
    hands, scores = scored_hands
    got, gsc, gsrc = hostlib.find_clusters(hands[:0], scores[:0], 1)
    assert len(got) == 0
    got, gsc, gsrc = hostlib.find_clusters(hands[:1], scores[:1], 1)  # a lone hand has no inliers
    assert len(got) == 0
    # two identical hands are each other's single inlier: position unchanged, sd = 0 -> score = the other's score
    two = np.concatenate([hands[:1], hands[:1]])
    got, gsc, gsrc = hostlib.find_clusters(two, np.array([1.5, -2.0]), 1)
    assert list(gsrc) == [0, 1] and list(gsc) == [-2.0, 1.5]
    assert np.array_equal(got["position"], two["position"])


def test_clusters_of_many_hands(oracle_mod):
    """More hands than one 256-lane chunk and more than one 1024-entry round of the output scan (pruneGraspCandidates
    hands every valid grasp to the clustering: sequential_importance_sampling.cpp:178)."""
    rng = np.random.RandomState(9)
    cl = synth.make_cloud(12, 12000)
    p = oracle_mod.default_params(15)
    hands = oracle_mod.search(p, cl["xyz"], cl["normals"], synth.sample_indices(cl, 400)).reshape(-1)
    hands = hands[hands["valid"].astype(bool)]
    assert len(hands) > 1100
    extra = hands[rng.randint(0, len(hands), 600)].copy()
    ax = extra["frame"].reshape(-1, 3, 3)[:, :, 2]
    extra["position"] += ax * rng.uniform(-0.03, 0.03, (len(extra), 1)) + rng.normal(0, 0.001, (len(extra), 3))
    hands = np.concatenate([hands, extra])[rng.permutation(len(hands) + len(extra))]
    scores = rng.normal(0, 3, len(hands))
    ctx = api.Context(api.default_params(15))
    try:
        for min_inliers, remove in ((1, False), (2, True)):
            want, wsc, wsrc = oracle_mod.find_clusters(hands, scores, min_inliers, remove)
            got, gsc, gsrc = ctx.find_clusters(hands, scores, min_inliers, remove)
            assert len(want) > 200 and np.array_equal(gsrc, wsrc)
            assert gsc.tobytes() == wsc.tobytes() and got["position"].tobytes() == want["position"].tobytes()
    finally:
        ctx.close()
