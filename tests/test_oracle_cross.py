"""Cross-checks of the CPU oracle against independent numpy / scipy / torch re-derivations
(SURVEY.md §8c "independent cross-checks available in-container")."""
import numpy as np
import pytest
import torch

from gpd_amd import synth

import pyref


@pytest.fixture(scope="module")
def small_cloud():
    return synth.make_cloud(77, 6000)


def test_radius_search_matches_kdtree_and_is_sorted(oracle_mod, cloud30k):
    from scipy.spatial import cKDTree
    xyz = cloud30k["xyz"]
    tree = cKDTree(xyz.astype(np.float64))
    rng = np.random.RandomState(1)
    for q in rng.choice(len(xyz), 12, replace=False):
        for r in (0.01, 0.10, 0.11):
            idx, d2 = oracle_mod.radius_search(xyz, xyz[q], r)
            assert np.all(d2 < np.float32(r * r))
            key = list(zip(d2.tolist(), idx.tolist()))
            assert key == sorted(key), "not in (d2, index) order"
            assert idx[0] == q and d2[0] == 0
            # same set as an exact search, up to points within float rounding of the sphere
            exact = set(tree.query_ball_point(xyz[q].astype(np.float64), r * (1 - 1e-5)))
            loose = set(tree.query_ball_point(xyz[q].astype(np.float64), r * (1 + 1e-5)))
            assert exact <= set(idx.tolist()) <= loose
            assert np.array_equal(idx, pyref.radius_neighbours(xyz, xyz[q], r))


def test_empty_and_tiny_neighbourhoods(oracle_mod):
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0.005, 0, 0]], np.float32)
    idx, d2 = oracle_mod.radius_search(xyz, np.array([5, 5, 5], np.float32), 0.01)
    assert len(idx) == 0
    idx, _ = oracle_mod.radius_search(xyz, xyz[0], 0.01)
    assert idx.tolist() == [0, 2]
    # strictness: a point at exactly d2 == r2 is not a neighbour
    r2 = np.float32(0.01 * 0.01)
    x = np.float32(np.sqrt(np.float64(r2)))
    if np.float32(x * x) == r2:
        idx, _ = oracle_mod.radius_search(np.array([[0, 0, 0], [x, 0, 0]], np.float32), np.zeros(3, np.float32), 0.01)
        assert idx.tolist() == [0]


def test_eigen3_against_numpy(oracle_mod):
    rng = np.random.RandomState(3)
    for k in range(50):
        N = rng.randn(3, 5 + k)
        if k % 7 == 0:
            N[2] = 0  # planar normals: one zero eigenvalue
        M = N @ N.T
        ev, V = oracle_mod.eigen3(M)
        w, U = np.linalg.eigh(M)
        assert np.all(np.diff(ev) >= 0)
        assert np.allclose(ev, w, rtol=1e-10, atol=1e-12 * abs(w).max())
        assert np.allclose(V @ np.diag(ev) @ V.T, M, atol=1e-10 * abs(M).max())
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-12)
    ev, V = oracle_mod.eigen3(np.diag([3.0, 1.0, 2.0]))
    assert ev.tolist() == [1.0, 2.0, 3.0]
    ev, V = oracle_mod.eigen3(np.zeros((3, 3)))
    assert ev.tolist() == [0, 0, 0] and np.array_equal(V, np.eye(3))


def test_frames_are_right_handed_and_face_the_normals(oracle_mod, cloud30k):
    p = oracle_mod.default_params()
    si = synth.sample_indices(cloud30k, 64)
    fr, has = oracle_mod.frames(p, cloud30k["xyz"], cloud30k["normals"], si)
    assert has.all()
    for f, s in zip(fr, si):
        n, b, c = f[3:6], f[6:9], f[9:12]
        assert np.allclose(f[:3], cloud30k["xyz"][s].astype(np.float64))
        assert np.allclose(np.cross(c, n), b, atol=1e-15)
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and abs(np.dot(n, c)) < 1e-12
        idx, _ = oracle_mod.radius_search(cloud30k["xyz"], cloud30k["xyz"][s], 0.01)
        assert np.dot(cloud30k["normals"][idx].astype(np.float64).sum(0), n) >= 0


def _torch_lenet(img, w, C):
    x = torch.from_numpy(img.astype(np.float64)).permute(0, 3, 1, 2)
    t = lambda k, *s: torch.from_numpy(w[k].astype(np.float64)).view(*s)
    h = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(x, t("c1w", 20, C, 5, 5), t("c1b", 20)), 2)
    h = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(h, t("c2w", 50, 20, 5, 5), t("c2b", 50)), 2)
    f = h.permute(0, 2, 3, 1).reshape(len(img), 7200)  # pixel-major, channel-minor flatten
    a = torch.relu(f @ t("f1w", 7200, 500) + t("f1b", 500))
    y = a @ t("f2w", 500, 2) + t("f2b", 2)
    return (y[:, 1] - y[:, 0]).numpy()


@pytest.mark.parametrize("C", [15, 3])
def test_lenet_against_torch_fp64(oracle_mod, C):
    import os
    real = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)))
    w = synth.lenet_weights(C, real=real)
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(6, 60, 60, C)).astype(np.uint8)
    got = oracle_mod.lenet(img, w)
    want = _torch_lenet(img, w, C)
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_workspace_filter_matches_numpy(oracle_mod, cloud30k):
    import bench
    p = oracle_mod.default_params()
    si = synth.sample_indices(cloud30k, 100)
    hands = oracle_mod.search(p, cloud30k["xyz"], cloud30k["normals"], si)
    a = oracle_mod.filter_workspace(p, hands.copy())
    b = hands.copy()
    bench._filter_workspace(b, p)
    assert np.array_equal(a["valid"], b["valid"])
    assert a["valid"].sum() < hands["valid"].sum()  # the 0.085 aperture removes some


@pytest.mark.parametrize("C", [15, 12, 3, 1])
def test_images_against_python_reference(oracle_mod, small_cloud, C):
    cl = small_cloud
    p = oracle_mod.default_params(C)
    si = synth.sample_indices(cl, 6)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    assert len(cand) >= 3
    flat = hands.reshape(-1)
    rng = pyref.Lcg(0)
    k = 0
    checked = 0
    for s in range(hands.shape[0]):
        if not hands[s]["valid"].any():
            continue
        nbr = pyref.radius_neighbours(cl["xyz"], hands[s, 0]["sample"].astype(np.float32), 0.10)
        vox = pyref.shadow_voxels(cl["xyz"][nbr], cl["view_points"][0], rng) if C == 15 else None
        for j in range(hands.shape[1]):
            if not hands[s, j]["valid"]:
                continue
            assert cand[k] == s * hands.shape[1] + j
            if checked < 4:
                want = pyref.grasp_image(p, flat[cand[k]], cl["xyz"], cl["normals"], nbr, vox)
                assert np.array_equal(img[k], want), "candidate %d: %d pixels differ" % (k, (img[k] != want).sum())
                checked += 1
            k += 1
        if checked >= 4 and C != 15:
            break
    assert checked >= 3


def test_image_channel_properties(oracle_mod, small_cloud):
    cl = small_cloud
    p = oracle_mod.default_params(15)
    si = synth.sample_indices(cl, 10)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
    img, _ = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    assert len(img) > 5
    for im in img:
        for pr in range(3):
            n = im[..., pr * 5:pr * 5 + 3]
            assert n.max() in (0, 255)         # min-max normalised over the 3 normal channels jointly
            for ch in (3, 4):
                c = im[..., pr * 5 + ch]
                assert c.max() in (0, 255) and c.min() == 0


def test_search_invariants(oracle_mod, cloud30k):
    p = oracle_mod.default_params()
    si = synth.sample_indices(cloud30k, 120)
    hands = oracle_mod.search(p, cloud30k["xyz"], cloud30k["normals"], si)
    assert hands.shape == (120, 8)
    v = hands["valid"].astype(bool)
    assert v.sum() > 50
    depths = [0.01, 0.015, 0.02, 0.025, 0.030000000000000002, 0.035, 0.04, 0.045, 0.049999999999999996,
              0.05499999999999999, 0.05999999999999999]
    assert set(np.unique(hands["top"][v])) <= set(depths)
    assert np.all(hands["bottom"][v] == hands["top"][v] - 0.06)
    assert np.all((hands["finger_placement_index"][v] >= 0) & (hands["finger_placement_index"][v] < 10))
    assert np.all(hands["grasp_width"][v] >= 0) and np.all(hands["grasp_width"][v] < 0.11)
    assert np.all(hands["full_antipodal"] <= hands["half_antipodal"])
    F = hands["frame"].reshape(-1, 3, 3)
    assert np.allclose(np.einsum("nij,nkj->nik", F, F), np.eye(3), atol=1e-12)
    # all 8 orientations of a set share the hand axis up to sign (rotation about axis 2)
    ax = hands["frame"].reshape(120, 8, 3, 3)[:, :, :, 2]
    assert np.allclose(np.abs(np.einsum("sjk,sk->sj", ax, ax[:, 0])), 1.0, atol=1e-12)
    assert np.array_equal(hands["slot"], np.tile(np.arange(8), (120, 1)))
    assert np.array_equal(hands["set_index"], np.repeat(np.arange(120), 8).reshape(120, 8))


def test_openmp_matches_single_thread(oracle_mod, small_cloud, lenet15_real):
    cl = small_cloud
    p = oracle_mod.default_params(15)
    si = synth.sample_indices(cl, 12)
    a, na, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real, threads=1)
    b, nb, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real, threads=4)
    assert na == nb and a.tobytes() == b.tobytes()


def test_multi_camera_shadow_against_python_reference(oracle_mod):
    """Two and three cameras with partial visibility (one set where camera 0 sees nothing): the oracle's
    15-channel images against the independent python rasteriser + camera-set intersection."""
    cl = synth.make_cloud(77, 5000)
    P = len(cl["xyz"])
    rng0 = np.random.RandomState(5)
    for n_cams in (2, 3):
        cam = (rng0.rand(n_cams, P) < 0.7).astype(np.int32)
        vp = np.array([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1], [-0.25, 0.3, 0.05]])[:n_cams]
        p = oracle_mod.default_params(15)
        si = synth.sample_indices(cl, 5)
        hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
        # blind camera 0 around the second live sample
        live = [s for s in range(len(si)) if hands[s]["valid"].any()]
        assert len(live) >= 2
        blind = pyref.radius_neighbours(cl["xyz"], hands[live[1], 0]["sample"].astype(np.float32), 0.10)
        cam[0, blind] = 0
        cam[1, blind[0]] = 1
        img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cam, vp, hands)
        flat = hands.reshape(-1)
        rng = pyref.Lcg(0)
        k = 0
        for s in live:
            nbr = pyref.radius_neighbours(cl["xyz"], hands[s, 0]["sample"].astype(np.float32), 0.10)
            vox = pyref.shadow_voxels_cameras(cl["xyz"][nbr], cam[:, nbr], vp, rng)
            first = True
            for j in range(hands.shape[1]):
                if not hands[s, j]["valid"]:
                    continue
                if first:  # one image per set is enough (the python rasteriser is slow)
                    want = pyref.grasp_image(p, flat[cand[k]], cl["xyz"], cl["normals"], nbr, vox)
                    assert np.array_equal(img[k], want), (n_cams, s, j, (img[k] != want).sum())
                    if s == live[1]:
                        assert not want[..., 4].any()  # camera 0 sees nothing here: empty intersection, no shadow
                    first = False
                k += 1
        assert k == len(cand)


def test_shadow_lcg_stream_for_another_image_volume(oracle_mod, small_cloud):
    """HandSet::calculateShadowForCamera draws N * num_shadow_points values per camera and hand set from ONE
    LCG stream (hand_set.cpp:202-212, 263-266), num_shadow_points = floor(shadow_length / 0.003): 33 for the
    default 0.10 m volume, 26 for a 0.08 m one.  The oracle (and the HIP path) place every set in the stream by
    jump-ahead; here the SECOND and later sets are checked against a plain sequential run of the generator."""
    cl = small_cloud
    p = oracle_mod.default_params(15)
    p.volume_width, p.volume_depth, p.volume_height = 0.08, 0.05, 0.03
    assert int(np.floor(0.08 / 0.003)) == 26
    si = synth.sample_indices(cl, 8)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    flat = hands.reshape(-1)
    rng = pyref.Lcg(0)
    k = 0
    live = 0
    checked = 0
    for s in range(hands.shape[0]):
        if not hands[s]["valid"].any():
            continue
        nbr = pyref.radius_neighbours(cl["xyz"], hands[s, 0]["sample"].astype(np.float32), 0.08)
        vox = pyref.shadow_voxels(cl["xyz"][nbr], cl["view_points"][0], rng, shadow_length=0.08)
        first = True
        for j in range(hands.shape[1]):
            if not hands[s, j]["valid"]:
                continue
            if live >= 1 and first and checked < 3:  # sets after the first: their stream offset matters
                want = pyref.grasp_image(p, flat[cand[k]], cl["xyz"], cl["normals"], nbr, vox)
                assert want[..., 4].any()
                assert np.array_equal(img[k], want), "set %d: %d pixels differ" % (s, (img[k] != want).sum())
                checked += 1
            first = False
            k += 1
        live += 1
    assert checked >= 2
