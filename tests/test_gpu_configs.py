"""BASELINE.json configs on the GPU against the oracle (configs[1..4]) plus edge cases and
error behaviour of the C-ABI.  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest

import ref_cases as rcs
from gpd_amd import api, synth

pytestmark = pytest.mark.gpu


def _weights(C):
    import os
    g = os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


def _full_compare(oracle_mod, cl, si, C, max_cand=None):
    w = _weights(C)
    ctx = api.Context(api.default_params(C))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands, n_cand = ctx.detect(si)
        p = oracle_mod.default_params(C)
        ohands, on_cand, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, w)
        assert n_cand == on_cand
        assert np.array_equal(hands["valid"], ohands["valid"])
        v = ohands["valid"].astype(bool)
        for f in ("finger_placement_index", "half_antipodal", "full_antipodal"):
            assert np.array_equal(hands[f][v], ohands[f][v]), f
        for f in ("frame", "position", "top", "bottom", "center", "grasp_width", "sample"):
            assert np.array_equal(hands[f][v], ohands[f][v]), f  # f64 record fields bit for bit, as the pin tests hold them
        err = np.abs(hands["score"][v] - ohands["score"][v]).max()
        assert err <= 1e-4, err
        return n_cand, float(err)
    finally:
        ctx.close()


def test_config2_30k_cloud_5000_candidates(oracle_mod, cloud30k):
    """configs[1]: single 30k-point cloud, ~5000 candidates, 15 channels, full oracle comparison."""
    si = synth.sample_indices(cloud30k, 2200)
    n, err = _full_compare(oracle_mod, cloud30k, si, 15)
    assert n > 4500


def test_config2_off_lattice(oracle_mod, cloud30k):
    """configs[1]'s scene with sensor-like coordinates (synth.off_lattice, bench.py --config 2o): every point off the 3 mm
    lattice — no distance ties, no point exactly on a decision plane — ~5000 candidates, full oracle comparison.  The small
    version of this cloud is pinned against the reference's own sources (ref_pin_offlattice_*.npz)."""
    cl = synth.off_lattice(cloud30k)
    si = synth.sample_indices(cl, 2200)
    n, err = _full_compare(oracle_mod, cl, si, 15)
    assert n > 4000


@pytest.mark.parametrize("seed", range(int(os.environ.get("GPD_FULLSIZE_SEEDS", "1"))))
def test_config2_other_scenes(oracle_mod, seed):
    """configs[1]'s size on OTHER synthetic scenes (seed 7000 + k; odd seeds off the lattice): ~5000 candidates each, full oracle
    comparison.  GPD_FULLSIZE_SEEDS=N widens the draw (12 soaked in round 6: profiles/r06_soak.txt)."""
    cl = synth.make_cloud(7000 + seed, 30000, clutter=bool(seed % 4 == 2))
    if seed % 2:
        cl = synth.off_lattice(cl, seed=seed)
    si = synth.sample_indices(cl, 2200, seed=seed)
    n, err = _full_compare(oracle_mod, cl, si, 15)
    assert n > 3000


@pytest.mark.parametrize("C", [1, 3, 12])
def test_config3_other_image_geometries(oracle_mod, cloud30k, C):
    """configs[2]: 3- and 12-channel geometries on the same cloud."""
    si = synth.sample_indices(cloud30k, 500)
    n, err = _full_compare(oracle_mod, cloud30k, si, C)
    assert n > 1000


def test_config4_dense_clutter_300k(oracle_mod):
    """configs[3] at its stated size: 300k-point clutter cloud, at least 50000 candidates through the fused
    device path (the 16384-image LeNet chunk loop, large neighbourhoods), every one compared with the oracle."""
    cl = synth.make_cloud(1234, 300000, clutter=True)
    si = synth.sample_indices(cl, 19000)
    n, err = _full_compare(oracle_mod, cl, si, 15)
    assert n >= 50000, n


def test_config5_batch_of_clouds_one_context(oracle_mod):
    """configs[4] on one GPU: independent clouds through one context, results independent of order."""
    w = _weights(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        p = oracle_mod.default_params(15)
        first = None
        for cid in (0, 1, 2, 0):
            cl = synth.make_cloud(1234 + cid, 30000)
            si = synth.sample_indices(cl, 120)
            ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
            hands, n = ctx.detect(si)
            oh, on, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, w)
            assert n == on and np.array_equal(hands["valid"], oh["valid"])
            v = oh["valid"].astype(bool)
            assert np.abs(hands["score"][v] - oh["score"][v]).max() <= 1e-4
            if cid == 0:
                if first is None:
                    first = hands.copy()
                else:  # the shadow LCG restarts per cloud: same cloud -> same bytes
                    assert first.tobytes() == hands.tobytes()
    finally:
        ctx.close()


def test_replay_is_idempotent_and_matches_detect(cloud30k):
    w = _weights(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
        si = synth.sample_indices(cloud30k, 300)
        hands, n = ctx.detect(si)
        want = hands["score"][hands["valid"].astype(bool)]
        for _ in range(3):
            ctx.replay(3)
        _, _, launches, sc = ctx.replay_times(n_scores=n)
        assert launches == 3 and np.array_equal(sc, want)
    finally:
        ctx.close()


def test_fc1_tile_shapes_agree_with_the_oracle_checked_one(oracle_mod):
    """ip1 runs 128 x (16 .. 128)-wide tiles picked per launch (lenet.hip fc1_pick_nt): every tile width, the
    two-round case and the ragged last tile must give each image the score it gets in a small batch (the 16-wide
    shape, which is the one compared with the oracle image by image here and in test_lenet_scores_match_oracle)."""
    w = _weights(15)
    rng = np.random.RandomState(5)
    base = (rng.randint(0, 256, (640, 60, 60, 15)) * (rng.rand(640, 60, 60, 15) < 0.3)).astype(np.uint8)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.set_lenet_mode(api.LENET_F32_CHAIN)  # this test walks the f32-chain kernels' tile shapes (the split path's: test_gpu_lenet_fast.py)
        ref = ctx.score(base)  # n = 640 -> the 16-wide tile
        assert np.array_equal(ref[:48], oracle_mod.lenet(base[:48], w))
        assert len(np.unique(ref)) > 600
        # (the small and odd counts walk conv1's persistent grid: one workgroup with 1-3 images, 256 workgroups with 2 / 3
        #  images each, image counts that do not divide by the grid)
        for n in (1, 2, 3, 5, 17, 255, 256, 257, 511, 512, 513, 767, 1000, 1025, 2100, 3100, 4200, 5000, 5200, 6200, 7200, 8192, 10000):
            idx = rng.randint(0, len(base), n)
            got = ctx.score(base[idx])
            assert np.array_equal(got, ref[idx]), n
    finally:
        ctx.close()


def test_edge_cases_and_errors(oracle_mod, cloud30k):
    w = _weights(15)
    ctx = api.Context(api.default_params(15))
    try:
        with pytest.raises(api.GpdHipError):  # no cloud yet
            ctx.search(np.array([0], np.int32))
        with pytest.raises(api.GpdHipError):  # no weights yet
            ctx.score(np.zeros((1, 60, 60, 15), np.uint8))
        ctx.set_lenet_weights(w)
        assert len(ctx.score(np.zeros((0, 60, 60, 15), np.uint8))) == 0
        # all-zero / all-255 images
        img = np.zeros((2, 60, 60, 15), np.uint8)
        img[1] = 255
        rcs.assert_scores(ctx, img, oracle_mod.lenet(img, w))
        # conv1's zero skipping on structured sparsity: single channels, single rows / columns / pixels set,
        # an odd image count (the second image of the last pair is a phantom)
        rng = np.random.RandomState(11)
        img = np.zeros((37, 60, 60, 15), np.uint8)
        for i in range(37):
            kind = i % 6
            if kind == 0:
                img[i, :, :, rng.randint(15)] = rng.randint(1, 256, (60, 60))
            elif kind == 1:
                img[i, rng.randint(60), :, :] = rng.randint(0, 256, (60, 15))
            elif kind == 2:
                img[i, :, rng.randint(60), rng.randint(15)] = rng.randint(1, 256, 60)
            elif kind == 3:
                img[i, rng.randint(60), rng.randint(60), rng.randint(15)] = 255
            elif kind == 4:
                img[i, 30:, 28:40, ::2] = rng.randint(0, 256, (30, 12, 8))
            else:
                img[i] = rng.randint(0, 256, (60, 60, 15)) * (rng.rand(60, 60, 15) < 0.05)
        rcs.assert_scores(ctx, img, oracle_mod.lenet(img, w))
        # image counts around the pair size of a conv1 workgroup (the pair's pixels share 64-lane chunks)
        for n in (1, 2, 3):
            rcs.assert_scores(ctx, img[:n], oracle_mod.lenet(img[:n], w))
        # one lit pixel at the image corners and at the seams of conv1's 7-column strips / 64-pixel chunks
        spots = [(0, 0), (59, 59), (0, 59), (59, 0), (17, 13), (18, 14), (35, 27), (36, 28), (19, 41), (20, 42)]
        img = np.zeros((len(spots), 60, 60, 15), np.uint8)
        for i, (y, x) in enumerate(spots):
            img[i, y, x, (0, 14)[i % 2]] = 255 - i
        rcs.assert_scores(ctx, img, oracle_mod.lenet(img, w))
        # the skipping is exact for finite weights only, and so are the split path's pieces: anything else is refused
        for key, at in (("c1w", 7), ("c2w", 123), ("f1w", 99999)):
            bad = {k: v.copy() for k, v in w.items()}
            bad[key][at] = np.inf if key != "f1w" else np.nan
            with pytest.raises(api.GpdHipError):
                ctx.set_lenet_weights(bad)
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
        assert ctx.search(np.zeros(0, np.int32)).shape == (0, 8)
        with pytest.raises(api.GpdHipError):  # index out of range
            ctx.search(np.array([len(cloud30k["xyz"])], np.int32))
        # a table point has no graspable geometry: a set without valid hands yields no image
        tab = np.flatnonzero(~cloud30k["is_object"])[:3].astype(np.int32)
        hands = ctx.search(tab)
        oh = oracle_mod.search(oracle_mod.default_params(15), cloud30k["xyz"], cloud30k["normals"], tab)
        assert np.array_equal(hands["valid"], oh["valid"])
        hands["valid"] = 0
        img, cand = ctx.images(hands)
        assert img.shape[0] == 0 and len(cand) == 0
        # a single sample
        one = synth.sample_indices(cloud30k, 1)
        h1, n1 = ctx.detect(one)
        o1, on1, _ = oracle_mod.detect(oracle_mod.default_params(15), cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"],
                                       cloud30k["view_points"], one, w)
        assert n1 == on1 and np.array_equal(h1["valid"], o1["valid"])
        # weights of the wrong geometry are refused
        with pytest.raises(AssertionError):
            ctx.set_lenet_weights(_weights(3))
    finally:
        ctx.close()


def test_tiny_cloud(oracle_mod):
    """A cloud smaller than one workgroup's stride and a sample whose neighbourhood is the whole cloud."""
    rng = np.random.RandomState(3)
    xyz = (np.round(rng.rand(50, 3) * 0.02 / 0.003) * 0.003).astype(np.float32)
    xyz = np.unique(xyz, axis=0)
    nrm = rng.randn(len(xyz), 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    ctx = api.Context(api.default_params(12))
    try:
        ctx.upload_cloud(xyz, nrm)
        si = np.arange(min(5, len(xyz)), dtype=np.int32)
        hands = ctx.search(si)
        oh = oracle_mod.search(oracle_mod.default_params(12), xyz, nrm, si)
        assert np.array_equal(hands["valid"], oh["valid"])
        assert np.allclose(hands["frame"], oh["frame"], rtol=1e-12, atol=1e-15)
    finally:
        ctx.close()


def test_two_cameras_shadow_intersection(oracle_mod, cloud30k):
    """Two view points: the shadow is the intersection of the per-camera voxel sets
    (hand_set.cpp:159-172), cameras draw from the LCG in order."""
    cl = cloud30k
    P = len(cl["xyz"])
    cam = np.zeros((2, P), np.int32)
    cam[0] = cl["xyz"][:, 0] < 0.08      # camera 0 misses one side, camera 1 the other
    cam[1] = cl["xyz"][:, 0] > -0.08
    vp = np.array([[0.0, 0.0, 0.0], [0.35, 0.1, 0.05]])
    si = synth.sample_indices(cl, 150)
    w = _weights(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], cl["normals"], cam, vp)
        p = oracle_mod.default_params(15)
        hands = oracle_mod.filter_workspace(p, ctx.search(si))
        got, gidx = ctx.images(hands)
        want, widx = oracle_mod.images(p, cl["xyz"], cl["normals"], cam, vp, hands)
        assert np.array_equal(gidx, widx) and len(gidx) > 100
        assert np.array_equal(got, want), "differing pixels: %d" % (got != want).sum()
        # the shadow channels are not trivially empty and differ from the one-camera result
        assert got[..., 4].any()
        ctx.upload_cloud(cl["xyz"], cl["normals"], cam[:1], vp[:1])
        h1 = oracle_mod.filter_workspace(p, ctx.search(si))
        one, _ = ctx.images(h1)
        assert not np.array_equal(one[..., 4], got[..., 4])
    finally:
        ctx.close()


def test_twelve_cameras(oracle_mod):
    """More cameras than any rig has (the reference's camera_source is an n_cams x N matrix without a bound,
    cloud.h:330-350; the kernels carry the seeing cameras as a 32-bit mask): normals flipped towards the first
    seeing camera, one shadow voxel set per seeing camera in camera order, their intersection, the LCG draws of
    every camera consumed — normals, records, images and scores against the oracle.  A 33rd camera is refused."""
    cl = synth.make_cloud(77, 12000)
    P = len(cl["xyz"])
    rng = np.random.RandomState(12)
    n_cams = 12
    cam = (rng.uniform(size=(n_cams, P)) < 0.93).astype(np.int32)
    cam[5] = 0                                  # a camera that sees nothing: no voxel set, no draws
    cam[7] = cl["xyz"][:, 0] < 0.1              # a camera that misses one side of the scene
    none = np.flatnonzero(cam.sum(axis=0) == 0)
    cam[0, none] = 1
    vp = np.array([[0.0, 0.0, 0.0]]) + rng.uniform(-0.04, 0.04, (n_cams, 3))
    vp[9] = [0.0, 0.0, 1.7]                     # one camera on the far side: normals of points only it sees flip
    only9 = rng.choice(P, 300, replace=False)
    cam[:, only9] = 0
    cam[9, only9] = 1
    w = _weights(15)
    p = oracle_mod.default_params(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], np.zeros_like(cl["xyz"]), cam, vp)
        nrm = ctx.estimate_normals(0.03)
        want_n = oracle_mod.estimate_normals(cl["xyz"], cam, vp, 0.03)
        assert np.array_equal(nrm, want_n)
        assert not np.array_equal(nrm, oracle_mod.estimate_normals(cl["xyz"], cam[:1], vp[:1], 0.03))
        si = synth.sample_indices(cl, 120)
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(p, cl["xyz"], nrm, cam, vp, si, w)
        assert n_cand == on_cand and n_cand > 100
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, cl["xyz"], nrm, cam, vp, fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
        assert img[..., 4].any() and img[..., 9].any()  # the intersection of eleven voxel sets is not empty
        big = np.ones((33, P), np.int32)
        with pytest.raises(api.GpdHipError, match="cameras"):
            ctx.upload_cloud(cl["xyz"], cl["normals"], big, np.zeros((33, 3)))
    finally:
        ctx.close()


def _locally_dense_cloud(copies):
    """The 30k cloud with `copies` jittered duplicates of everything within 6 cm of one object point."""
    from scipy.spatial import cKDTree
    cl = synth.make_cloud(1234, 30000)
    obj = np.flatnonzero(cl["is_object"])
    t = cKDTree(cl["xyz"].astype(np.float64))
    centre = cl["xyz"][obj[len(obj) // 2]].astype(np.float64)
    ball = np.array(t.query_ball_point(centre, 0.06))
    rng = np.random.RandomState(3)
    extra = [(cl["xyz"][ball] + rng.uniform(-0.0012, 0.0012, (len(ball), 3))).astype(np.float32) for _ in range(copies)]
    xyz = np.concatenate([cl["xyz"]] + extra)
    nrm = np.concatenate([cl["normals"]] + [cl["normals"][ball]] * copies)
    near = np.array([i for i in t.query_ball_point(centre, 0.045) if cl["is_object"][i]], np.int32)
    return cl, xyz, nrm, near


def test_dense_cloud_overflow_fallbacks(oracle_mod):
    """Neighbourhoods beyond 8192 points (the 16384-entry bitonic retry of the search) on a cloud of twice the
    density, and image boxes with more than 2048 in-box points (the global-scratch instantiation of the points
    kernel) on a locally five-fold cloud — both fallbacks against the oracle."""
    w = _weights(15)
    p = oracle_mod.default_params(15)
    cl, xyz, nrm, near = _locally_dense_cloud(5)
    cam = np.ones((1, len(xyz)), np.int32)
    si = near[np.random.RandomState(9).choice(len(near), min(60, len(near)), replace=False)]
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(xyz, nrm, cam, cl["view_points"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(p, xyz, nrm, cam, cl["view_points"], si, w)
        assert n_cand == on_cand and n_cand > 50
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, xyz, nrm, cam, cl["view_points"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
        # some box really holds more than 2048 points: count them for the valid hands (hand-frame box test)
        worst = 0
        flat = fw.reshape(-1)
        for h in flat[cand][:: max(1, len(cand) // 40)]:
            F = h["frame"].reshape(3, 3)
            t = (xyz.astype(np.float64) - h["sample"]) @ F
            inb = (t[:, 0] > h["bottom"]) & (t[:, 0] < h["bottom"] + 0.06) & (np.abs(t[:, 1] - h["center"]) < 0.05) & (np.abs(t[:, 2]) < 0.02)
            worst = max(worst, int(inb.sum()))
        assert worst > 2048, worst
    finally:
        ctx.close()
    # twice the density everywhere: neighbourhoods of more than 8192 points
    cl = synth.make_cloud(1234, 30000)
    rng = np.random.RandomState(3)
    jit = (cl["xyz"] + rng.uniform(-0.0012, 0.0012, cl["xyz"].shape)).astype(np.float32)
    xyz = np.concatenate([cl["xyz"], jit])
    nrm = np.concatenate([cl["normals"]] * 2)
    cam = np.ones((1, len(xyz)), np.int32)
    obj = np.flatnonzero(cl["is_object"])
    si = obj[np.random.RandomState(9).choice(len(obj), 400, replace=False)].astype(np.int32)
    w = _weights(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(xyz, nrm, cam, cl["view_points"])
        hands, n_cand = ctx.detect(si)
        p = oracle_mod.default_params(15)
        ohands, on_cand, _ = oracle_mod.detect(p, xyz, nrm, cam, cl["view_points"], si, w)
        assert n_cand == on_cand and n_cand > 100
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, xyz, nrm, cam, cl["view_points"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
        # the retry really ran: some neighbourhood holds more than 8192 points
        from scipy.spatial import cKDTree
        t = cKDTree(xyz.astype(np.float64))
        assert max(len(x) for x in t.query_ball_point(xyz[si[:40]].astype(np.float64), 0.11)) > 8192
        assert (img[..., 3] > 0).any()
    finally:
        ctx.close()


def test_neighbourhoods_beyond_every_lds_capacity(oracle_mod):
    """An un-voxelised scan is several times denser than the 3 mm clouds the LDS capacities were sized for (the
    reference has no limit: hand_search.cpp:178).  Four times the density: neighbourhoods of more than 16384 points
    take the global-memory bucket sort of the search, boxes with thousands of points the large points kernel —
    records, images and scores against the oracle."""
    cl = synth.make_cloud(1234, 30000)
    rng = np.random.RandomState(5)
    parts = [cl["xyz"]] + [(cl["xyz"] + rng.uniform(-0.0012, 0.0012, cl["xyz"].shape)).astype(np.float32) for _ in range(3)]
    xyz = np.concatenate(parts)
    nrm = np.concatenate([cl["normals"]] * 4)
    cam = np.ones((1, len(xyz)), np.int32)
    obj = np.flatnonzero(cl["is_object"])
    si = obj[np.random.RandomState(9).choice(len(obj), 48, replace=False)].astype(np.int32)
    from scipy.spatial import cKDTree
    t = cKDTree(xyz.astype(np.float64))
    sizes = [len(x) for x in t.query_ball_point(xyz[si].astype(np.float64), 0.11)]
    assert max(sizes) > 16384 + 1000, max(sizes)
    w = _weights(15)
    p = oracle_mod.default_params(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(xyz, nrm, cam, cl["view_points"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(p, xyz, nrm, cam, cl["view_points"], si, w)
        assert n_cand == on_cand and n_cand > 50
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, xyz, nrm, cam, cl["view_points"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
        # the unfused search retries through the capacities on its own, too
        h2 = ctx.search(si)
        o2 = oracle_mod.search(p, xyz, nrm, si)
        assert h2.tobytes() == o2.tobytes()
    finally:
        ctx.close()


def test_two_contexts_with_different_constants_on_one_device(oracle_mod):
    """The device constant blocks (image geometry + view points, hand constants) are one per device: two
    contexts with DIFFERENT geometry and cameras on the same device must not see each other's values —
    neither when their calls interleave (replay after the other context ran) nor from two host threads."""
    import threading
    w = _weights(15)
    clA, clB = synth.make_cloud(11, 12000), synth.make_cloud(12, 12000)
    clB = dict(clB)
    clB["view_points"] = np.array([[0.3, -0.2, 0.9]])  # another camera: other shadows, other constants
    pA, pB = api.default_params(15), api.default_params(15)
    pB.volume_width, pB.volume_depth, pB.volume_height = 0.08, 0.05, 0.03
    pB.finger_width, pB.num_finger_placements = 0.015, 6
    opA, opB = oracle_mod.default_params(15), oracle_mod.default_params(15)
    opB.volume_width, opB.volume_depth, opB.volume_height = 0.08, 0.05, 0.03
    opB.finger_width, opB.num_finger_placements = 0.015, 6
    siA, siB = synth.sample_indices(clA, 60), synth.sample_indices(clB, 60)
    wantA = oracle_mod.detect(opA, clA["xyz"], clA["normals"], clA["cam_source"], clA["view_points"], siA, w)
    wantB = oracle_mod.detect(opB, clB["xyz"], clB["normals"], clB["cam_source"], clB["view_points"], siB, w)
    A, B = api.Context(pA), api.Context(pB)
    try:
        for c, cl in ((A, clA), (B, clB)):
            c.set_lenet_weights(w)
            c.set_lenet_mode(api.LENET_F32_CHAIN)  # scores compared bit for bit with the oracle's below
            c.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])

        def same(got, want):
            h, n = got
            return n == want[1] and np.array_equal(h["valid"], want[0]["valid"]) and np.array_equal(h["score"], want[0]["score"])

        assert same(A.detect(siA), wantA) and same(B.detect(siB), wantB) and wantA[1] > 20 and wantB[1] > 20
        # A's candidate list is still on the device; B ran in between: the replay must reload A's constants
        hA = A.search(siA)
        hA = oracle_mod.filter_workspace(opA, hA)
        imgA, candA = A.images(hA)
        scoresA = A.score(imgA)
        assert same(B.detect(siB), wantB)
        A.replay(3)
        _, _, _, rs = A.replay_times(len(candA))
        assert np.array_equal(rs, scoresA)
        # two host threads, one context each
        bad = []

        def run(c, si, want):
            for _ in range(6):
                if not same(c.detect(si), want):
                    bad.append(1)

        ts = [threading.Thread(target=run, args=(A, siA, wantA)), threading.Thread(target=run, args=(B, siB, wantB))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not bad
    finally:
        A.close()
        B.close()


def test_bench_headline_list_exactly(oracle_mod, cloud30k):
    """The list bench.py times (configs[1] as quoted by `value`): cloud seed 1234, 2564 samples, the FIRST 5000 valid
    candidates after the workspace filter.  All 5000 images byte for byte and all 5000 scores against the oracle, with the
    benchmark's weights (|score| ~ 1000: bit-identical, relative 8e-6 of float64) and with the trained-magnitude set
    (north_star's absolute 1e-4).  Scores are also read back from gpd_hip_replay — the call the timed region is made of."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cl = cloud30k
    n_samples = min(int(5000 / 2.0) + 64, int(cl["is_object"].sum()))
    assert n_samples == 2564
    si = synth.sample_indices(cl, n_samples)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands = ctx.search(si)
        hf = hands.copy()
        bench._filter_workspace(hf, ctx.params)
        flat = hf.reshape(-1)
        vidx = np.flatnonzero(flat["valid"])
        assert len(vidx) > 5000
        flat["valid"][vidx[5000:]] = 0
        p = oracle_mod.default_params(15)
        oh = oracle_mod.search(p, cl["xyz"], cl["normals"], si)
        ohf = oracle_mod.filter_workspace(p, oh.copy())
        of = ohf.reshape(-1)
        ov = np.flatnonzero(of["valid"])
        of["valid"][ov[5000:]] = 0
        assert np.array_equal(hf["valid"], ohf["valid"])  # the numpy filter of bench.py == the oracle's == the reference's (pins)
        img, cand = ctx.images(hf, download=True)
        oimg, ocand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], ohf)
        assert len(cand) == 5000 and np.array_equal(cand, ocand)
        assert np.array_equal(img, oimg)
        for tm in (False, True):
            w = synth.lenet_weights(15, real=dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lenet15_params.npz"))),
                                    trained_magnitude=tm)
            ctx.set_lenet_weights(w)
            want = oracle_mod.lenet(oimg, w)
            ctx.set_lenet_mode(api.LENET_F32_CHAIN)
            ctx.replay(3)
            _, _, launches, sc = ctx.replay_times(n_scores=5000)
            assert launches == 1 and np.array_equal(sc, want), (tm, float(np.abs(sc - want).max()))
            ctx.set_lenet_mode(api.LENET_SPLIT)  # the default, and what bench.py times
            ctx.replay(3)
            _, _, launches, sp = ctx.replay_times(n_scores=5000)
            assert launches == 1
            f64 = bench._lenet_f64(oimg, w)
            if tm:
                # north_star's bar on all 5000 timed candidates, and the split path at least as close to float64 as the chain
                assert np.abs(f64).max() < 20.0 and np.abs(sc - f64).max() <= 1e-4 and np.abs(sp - f64).max() <= 1e-4
                assert np.abs(sp - want).max() <= 1e-4
                print("5000 timed candidates, trained magnitude: max |split - f64| = %.3g, max |f32 chain - f64| = %.3g"
                      % (np.abs(sp - f64).max(), np.abs(sc - f64).max()))
                assert np.abs(sp - f64).max() <= np.abs(sc - f64).max()
            else:
                assert np.abs(sp - f64).max() <= 2e-5 * np.abs(f64).max()
    finally:
        ctx.close()


def test_normals_of_neighbourhoods_beyond_every_lds_capacity(oracle_mod):
    """Cloud::calculateNormals has no size limit in the reference (cloud.cpp:497-535).  A 24k-point blob of 5 cm radius
    searched with radius 0.06: the central points have more than 16 000 neighbours (> the 512-entry wave sort, > the
    8192-entry LDS sort of the big kernel: the keys are sorted in the point's own row of the list array), next to a sparse
    shell whose points take the wave path — normals bit for bit against the oracle on all of them."""
    rng = np.random.RandomState(11)
    d = rng.randn(24000, 3)
    blob = (d / np.linalg.norm(d, axis=1, keepdims=True) * (0.05 * rng.rand(24000, 1) ** (1.0 / 3.0))).astype(np.float32)
    s = rng.randn(3000, 3)
    shell = (s / np.linalg.norm(s, axis=1, keepdims=True) * 0.6).astype(np.float32)
    xyz = np.concatenate([blob, shell]) + np.float32(0.7)
    cam = np.ones((1, len(xyz)), np.int32)
    vp = np.zeros((1, 3))
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(xyz, np.zeros_like(xyz), cam, vp)
        got = ctx.estimate_normals(0.06)
        want = oracle_mod.estimate_normals(xyz, cam, vp, 0.06)
        assert np.array_equal(got, want)
        got2 = ctx.estimate_normals(0.02)  # same context, smaller lists: the scratch array is reused
        assert np.array_equal(got2, oracle_mod.estimate_normals(xyz, cam, vp, 0.02))
    finally:
        ctx.close()


def test_neighbourhoods_beyond_65535_points(oracle_mod):
    """hand_search.cpp:178 takes whatever radiusSearch returns.  Sixteen times the density of the 3 mm benchmark cloud:
    0.11 m neighbourhoods of 70-100 thousand points — beyond the 16-bit neighbour ranks of hand_eval_kernel's LDS table
    (it walks the full list then), global-memory lists of > 65535 entries, image boxes with ten to twenty thousand points
    (the large points kernel, 32768 entries) — records, images and scores against the oracle."""
    cl = synth.make_cloud(1234, 30000)
    rng = np.random.RandomState(6)
    parts = [cl["xyz"]] + [(cl["xyz"] + rng.uniform(-0.0013, 0.0013, cl["xyz"].shape)).astype(np.float32) for _ in range(15)]
    xyz = np.concatenate(parts)
    nrm = np.concatenate([cl["normals"]] * 16)
    cam = np.ones((1, len(xyz)), np.int32)
    obj = np.flatnonzero(cl["is_object"])
    si = obj[np.random.RandomState(10).choice(len(obj), 16, replace=False)].astype(np.int32)
    from scipy.spatial import cKDTree
    t = cKDTree(xyz.astype(np.float64))
    sizes = [len(x) for x in t.query_ball_point(xyz[si].astype(np.float64), 0.11)]
    assert max(sizes) > 65535 + 2000, max(sizes)
    w = _weights(15)
    p = oracle_mod.default_params(15)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(xyz, nrm, cam, cl["view_points"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(p, xyz, nrm, cam, cl["view_points"], si, w)
        assert n_cand == on_cand and n_cand > 6
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(p, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(p, xyz, nrm, cam, cl["view_points"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
    finally:
        ctx.close()


def test_normals_of_a_cloud_beyond_one_tile_of_points(oracle_mod):
    """The normals kernels work through a cloud in tiles of 65536 points that share one lists array (search.hip kNormalsTile):
    a 150k-point cloud takes three passes, the last one ragged — normals bit for bit against the oracle, then a small cloud on
    the same context."""
    cl = synth.make_cloud(31, 150000)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(cl["xyz"], np.zeros_like(cl["xyz"]), cl["cam_source"], cl["view_points"])
        got = ctx.estimate_normals(0.01)
        want = oracle_mod.estimate_normals(cl["xyz"], cl["cam_source"], cl["view_points"], 0.01)
        assert np.array_equal(got, want)
        small = synth.make_cloud(32, 5000)
        ctx.upload_cloud(small["xyz"], np.zeros_like(small["xyz"]), small["cam_source"], small["view_points"])
        assert np.array_equal(ctx.estimate_normals(0.03), oracle_mod.estimate_normals(small["xyz"], small["cam_source"], small["view_points"], 0.03))
    finally:
        ctx.close()


def test_batch_after_a_dense_cloud_and_with_little_memory_to_presize(oracle_mod, cloud30k):
    """(1) the large neighbourhood lists one dense cloud needed are not carried into the next cloud / the next batch
    (they were: 44 bytes x list capacity x the lane's reserved samples); (2) a batch whose up-front sizing does not fit goes on
    with on-demand growth."""
    w = _weights(15)
    rng = np.random.RandomState(3)
    d = rng.randn(30000, 3)
    blob = (d / np.linalg.norm(d, axis=1, keepdims=True) * (0.04 * rng.rand(30000, 1) ** (1.0 / 3.0))).astype(np.float32) + np.float32(0.5)
    nb = blob / np.maximum(np.linalg.norm(blob - 0.5, axis=1, keepdims=True), 1e-6)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(blob, nb.astype(np.float32))
        ctx.search(np.arange(4, dtype=np.int32))  # neighbourhoods of ~30000 points: the global-memory lists
        assert ctx.fallbacks()["neighbourhood_list_capacity"] > 16384
        si = synth.sample_indices(cloud30k, 400)
        res = ctx.detect_batch([cloud30k, cloud30k], [si, si[:50]], 0)
        assert ctx.fallbacks()["neighbourhood_list_capacity"] == 8192
        p = oracle_mod.default_params(15)
        oh, on, _ = oracle_mod.detect(p, cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"], si, w)
        assert res[0][2] == on
        got = np.sort(res[0][0]["score"])
        want = np.sort(oh["score"][oh["valid"].astype(bool)])
        assert np.abs(got - want).max() <= 1e-4
    finally:
        ctx.close()
