"""Forms that share no code with either the oracle or the HIP path: numpy Rodrigues rotations, numpy `eigh`
(incl. degenerate and nearly degenerate matrices), scipy's k-d tree.  The oracle is checked in the CPU suite, the
product (through the C-ABI) in the GPU suite — so a misreading of Eigen's AngleAxis / SelfAdjointEigenSolver
semantics that the oracle and the kernels had in common (the eigensolver in search.hip is a port of the oracle's,
host_math.cpp's angle_axis has the oracle's form) would show up here."""
import math

import numpy as np
import pytest

from gpd_amd import synth


def rodrigues(angle, axis):
    """R = I + sin(a) K + (1 - cos(a)) K^2, K the cross-product matrix of the unit axis — textbook form, numpy only."""
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1.0 - math.cos(angle)) * (K @ K)


def test_oracle_angle_axis_against_rodrigues(oracle_mod):
    angles = [-math.pi / 2 + i * math.pi / 8 for i in range(8)] + [math.pi, 0.3, -2.0]
    for ang in angles:
        for ax in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
            R = np.asarray(oracle_mod.angle_axis(ang, ax)).reshape(3, 3)
            assert np.abs(R - rodrigues(ang, ax)).max() <= 4e-16, (ang, ax)
    # hand_set.cpp:52-53: AngleAxisd(M_PI, UnitY) keeps its +-sin(pi) entries
    R = np.asarray(oracle_mod.angle_axis(math.pi, [0, 1, 0])).reshape(3, 3)
    assert R[0, 2] == math.sin(math.pi) and R[2, 0] == -math.sin(math.pi) and R[0, 0] == -1.0


def test_oracle_eigen3_degenerate_and_near_degenerate(oracle_mod):
    rng = np.random.RandomState(11)
    Q, _ = np.linalg.qr(rng.randn(3, 3))
    cases = []
    for lam in ([1.0, 1.0, 1.0], [2.0, 2.0, 5.0], [1.0, 3.0, 3.0], [1.0, 1.0 + 1e-13, 4.0], [1e-18, 1.0, 1.0 + 1e-9],
                [0.0, 0.0, 7.0], [1e-300, 1e-300, 1e-300], [3.0, 3.0 * (1 + 2 ** -50), 3.0 * (1 + 2 ** -49)]):
        cases.append(Q @ np.diag(lam) @ Q.T)
        cases.append(np.diag(lam))
    for M in cases:
        M = 0.5 * (M + M.T)
        ev, V = oracle_mod.eigen3(M)
        w = np.linalg.eigvalsh(M)
        scale = max(abs(w).max(), 1e-300)
        assert np.all(np.diff(ev) >= 0)
        assert np.abs(ev - w).max() <= 1e-13 * scale
        assert np.abs(V.T @ V - np.eye(3)).max() <= 1e-12          # orthonormal even inside a degenerate eigenspace
        assert np.abs(V @ np.diag(ev) @ V.T - M).max() <= 1e-13 * scale


def _numpy_frames(cl, si, radius=0.01):
    """local_frame.cpp:14-41 with numpy.linalg.eigh and scipy's k-d tree; the sign of the curvature axis is
    whatever the eigensolver leaves (Eigen does not normalise it), so both signs are returned."""
    from scipy.spatial import cKDTree
    xyz = cl["xyz"]
    tree = cKDTree(xyz.astype(np.float64))
    out = []
    r2 = np.float32(radius * radius)
    for s in si:
        q = xyz[s]
        cand = np.array(tree.query_ball_point(q.astype(np.float64), radius * 1.01))
        d = (q[None, :] - xyz[cand]).astype(np.float32)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        nbr = cand[d2 < r2]
        N = cl["normals"][nbr].astype(np.float64).T
        w, U = np.linalg.eigh(N @ N.T)
        normal, curv = U[:, 2], U[:, 0]
        if np.dot(N.sum(1), normal) < 0:
            normal = -normal
        out.append((q.astype(np.float64), normal, curv, (w[1] - w[0]) / max(w[2], 1e-300), (w[2] - w[1]) / max(w[2], 1e-300)))
    return out


@pytest.mark.gpu
def test_product_frames_against_numpy_eigh_and_rodrigues(cloud30k):
    """gpd_hip_search's hand frames = [normal | curvature x normal | curvature] * Ry(pi) * Rz(angle) (hand_set.cpp:39-40,
    68-73) rebuilt from numpy.linalg.eigh + the Rodrigues form above; gpd_hand.frame to 1e-9 (eigenvectors of a
    well-separated spectrum), up to the joint sign of curvature and binormal."""
    from gpd_amd import api
    cl = cloud30k
    si = synth.sample_indices(cl, 200)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands = ctx.search(si)
    finally:
        ctx.close()
    assert hands.shape[0] == len(si)
    RB = rodrigues(math.pi, [0, 1, 0])
    checked = 0
    for s, (q, normal, curv, gap_lo, gap_hi) in enumerate(_numpy_frames(cl, si)):
        assert np.array_equal(hands[s, 0]["sample"], q)
        if min(gap_lo, gap_hi) < 1e-3:   # nearly degenerate spectrum: the eigenvectors themselves are ill-conditioned
            continue
        for j in range(8):
            R = rodrigues(-math.pi / 2 + j * math.pi / 8, [0, 0, 1])
            got = hands[s, j]["frame"].reshape(3, 3)
            ok = False
            for sgn in (1.0, -1.0):
                c = sgn * curv
                F = np.stack([normal, np.cross(c, normal), c], 1)
                ok |= np.abs(got - F @ RB @ R).max() <= 1e-9
            assert ok, (s, j)
        checked += 1
    assert checked > 100
