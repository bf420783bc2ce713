"""SURVEY §8f rank 2 on the device: Cloud::filterWorkspace (cloud.cpp:243-266) + Cloud::voxelizeCloud
(cloud.cpp:286-348) through gpd_hip_preprocess_cloud, against the oracle's restatement — which IS libstdc++'s
std::set under the reference's "differs" comparator — and numpy for the workspace cut."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(api.default_params(15))
    yield c
    c.close()


def _inside(xyz, ws):
    x = xyz.astype(np.float64)
    return (x[:, 0] > ws[0]) & (x[:, 0] < ws[1]) & (x[:, 1] > ws[2]) & (x[:, 1] < ws[3]) & (x[:, 2] > ws[4]) & (x[:, 2] < ws[5])


def _check(ctx, oracle_mod, xyz, cam, ws, cell):
    """gpd_hip_preprocess_cloud (cut + keys + gather on the device, the voxeliser's sequential chain as a spine walk on
    one host core) against the oracle — which IS std::set under the reference's comparator."""
    got_xyz, got_cam, got_src, ms = ctx.preprocess_cloud(xyz, cam, ws, cell)
    keep = np.flatnonzero(_inside(xyz, ws)) if ws is not None else np.arange(len(xyz))
    if cell > 0:
        want_xyz, src = oracle_mod.voxelize(xyz[keep], cell) if len(keep) else (np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
        want_src = keep[src]
        want_cam = (cam[:, want_src] == 1).astype(np.int32)
    else:
        want_xyz, want_src = xyz[keep], keep
        want_cam = cam[:, keep]
    assert got_xyz.shape == want_xyz.shape
    assert np.array_equal(got_src, want_src)
    assert got_xyz.tobytes() == want_xyz.tobytes()
    assert np.array_equal(got_cam, want_cam)
    return len(got_xyz), ms


def test_krylon_known_answer(ctx, oracle_mod):
    """4467 points in 2373 voxels -> the std::set keeps 3366, first from input 4466, 4464, 4459."""
    xyz = np.load(os.path.join(GOLD, "krylon_xyz.npz"))["xyz"]
    cam = np.ones((1, len(xyz)), np.int32)
    n, _ = _check(ctx, oracle_mod, xyz, cam, None, 0.003)
    assert n == 3366
    _, _, src, _ = ctx.preprocess_cloud(xyz, cam, None, 0.003)
    assert src[:5].tolist() == [4466, 4464, 4459, 4458, 4457]
    # the reference's default workspace ([-1, 1]^3) keeps every point of this scan; a tight one cuts before the voxeliser
    assert _check(ctx, oracle_mod, xyz, cam, np.array([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0]), 0.003)[0] == 3366
    lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    mid = 0.5 * (lo + hi)
    n_cut, _ = _check(ctx, oracle_mod, xyz, cam, np.array([lo[0], mid[0], lo[1] - 1, hi[1] + 1, lo[2] - 1, hi[2] + 1]), 0.003)
    assert 0 < n_cut < 3366


def test_table_mug_scan(ctx, oracle_mod):
    """tutorials/table_mug.pcd as shipped: 104 444 points -> 35 788, two cameras' worth of source flags."""
    xyz = np.load(os.path.join(GOLD, "table_mug_xyz.npz"))["xyz"]
    rng = np.random.RandomState(3)
    cam = rng.randint(0, 3, (2, len(xyz))).astype(np.int32)  # values other than 0 / 1 exercise the == 1 rule
    n, ms = _check(ctx, oracle_mod, xyz, cam, None, 0.003)
    assert n == 35788
    print("table_mug voxelise: %.2f ms of kernels" % ms)
    # no voxeliser: the workspace cut alone, order and camera columns preserved
    ws = np.array([-0.2, 0.15, -0.3, 0.1, 0.0, 1.2])
    n_ws, _ = _check(ctx, oracle_mod, xyz, cam, ws, 0.0)
    assert 0 < n_ws < len(xyz)
    _check(ctx, oracle_mod, xyz, cam, ws, 0.003)
    _check(ctx, oracle_mod, xyz, cam, ws, 0.01)


def test_fuzz_against_std_set(ctx, oracle_mod):
    """Point orders of every kind — scan lines, shuffled, heavy repetition, all in one voxel, sizes around the 64-point
    batches of the tree walk and the 256-point blocks of the compaction."""
    rng = np.random.RandomState(17)
    for trial, n in enumerate((1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 30000, 131072, 300000)):
        kind = trial % 4
        if kind == 0:      # a raster scan with noise: neighbours in the sequence are neighbours in space
            t = np.arange(n)
            xyz = np.stack([(t % 517) * 0.0011, (t // 517) * 0.0013, 0.4 + 0.01 * np.sin(t * 0.01)], 1) + rng.randn(n, 3) * 2e-4
        elif kind == 1:    # uniformly random in a small box: many revisits of the same voxels, far apart in the sequence
            xyz = rng.rand(n, 3) * [0.05, 0.04, 0.03]
        elif kind == 2:    # a handful of distinct voxels
            xyz = rng.randint(0, 3, (n, 3)) * 0.003 + 1e-4
        else:              # a synthetic scene
            xyz = synth.make_cloud(40 + trial, max(n, 1000))["xyz"][:n]
        xyz = np.ascontiguousarray(xyz, np.float32)
        cam = rng.randint(0, 2, (1 + trial % 3, n)).astype(np.int32)
        cell = (0.003, 0.005, 0.0007)[trial % 3]
        got = _check(ctx, oracle_mod, xyz, cam, None, cell)[0]
        assert 1 <= got <= n
        if n >= 1000:
            lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
            ws = np.array([lo[0] + 0.1 * (hi[0] - lo[0]), hi[0], lo[1], hi[1] - 0.2 * (hi[1] - lo[1]), lo[2] - 1, hi[2] + 1])
            _check(ctx, oracle_mod, xyz, cam, ws, cell)
            _check(ctx, oracle_mod, xyz, cam, ws, 0.0)


def test_edges_and_errors(ctx, oracle_mod):
    xyz = np.array([[0.1, 0.2, 0.3], [0.1, 0.2, 0.3], [0.5, 0.5, 0.5]], np.float32)
    cam = np.array([[1, 0, 2]], np.int32)
    # nothing inside the workspace -> an empty cloud, with or without the voxeliser
    far = np.array([5.0, 6.0, 5.0, 6.0, 5.0, 6.0])
    for cell in (0.003, 0.0):
        out, c, s, _ = ctx.preprocess_cloud(xyz, cam, far, cell)
        assert out.shape == (0, 3) and c.shape == (1, 0) and len(s) == 0
    # the bounds are strict (cloud.cpp:246): a point on the face is outside
    ws = np.array([float(np.float32(0.1)), 1.0, 0.0, 1.0, 0.0, 1.0])
    out, _, s, _ = ctx.preprocess_cloud(xyz, cam, ws, 0.0)
    assert s.tolist() == [2]
    # an empty input, and no camera rows at all
    out, c, s, _ = ctx.preprocess_cloud(np.zeros((0, 3), np.float32), None, None, 0.003)
    assert out.shape == (0, 3)
    out, c, s, _ = ctx.preprocess_cloud(xyz, None, None, 0.003)
    assert len(out) == 2 and c.shape == (0, 2) and s.tolist() == [2, 0]
    # NaN / Inf coordinates are the loader's job (cloud.cpp:154-164): refused here, as by gpd_hip_upload_cloud
    bad = xyz.copy()
    bad[1, 2] = np.nan
    with pytest.raises(api.GpdHipError, match="non-finite"):
        ctx.preprocess_cloud(bad, cam, None, 0.003)
    # a voxel size far too small for the extent: the voxel index would leave the int32 range
    wide = np.array([[0, 0, 0], [3000.0, 0, 0]], np.float32)
    with pytest.raises(api.GpdHipError, match="32 bits"):
        ctx.preprocess_cloud(wide, None, None, 1e-7)
    # the context's own cloud is untouched by all of this
    cl = synth.make_cloud(7, 12000)
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    before = ctx.search(synth.sample_indices(cl, 20)).tobytes()
    ctx.preprocess_cloud(cl["xyz"], cl["cam_source"], None, 0.003)
    assert ctx.search(synth.sample_indices(cl, 20)).tobytes() == before


def test_config1_preprocessing_on_the_device(ctx, oracle_mod):
    """candidates_generator.cpp:19-31 for configs[0] without a host pass: cut + voxelise + normals on the device
    equal the oracle's voxelise + normals."""
    xyz = np.load(os.path.join(GOLD, "krylon_xyz.npz"))["xyz"]
    cam = np.ones((1, len(xyz)), np.int32)
    v, c, _, _ = ctx.preprocess_cloud(xyz, cam, np.array([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0]), 0.003)
    ctx.upload_cloud(v, np.zeros_like(v), c, np.zeros((1, 3)))
    n = ctx.estimate_normals(0.03)
    ov, _ = oracle_mod.voxelize(xyz, 0.003)
    assert np.array_equal(n, oracle_mod.estimate_normals(ov))


def _stepwise(ctx, scan, samples_xyz, ws, cell, radius):
    """the raw scan through the single calls: preprocess_cloud -> upload -> estimate_normals -> upload -> detect by coordinates"""
    vox, cam, _, _ = ctx.preprocess_cloud(scan["xyz"], scan["cam_source"], ws, cell)
    ctx.upload_cloud(vox, np.zeros_like(vox), cam, scan["view_points"])
    nrm = ctx.estimate_normals(radius)
    ctx.upload_cloud(vox, nrm, cam, scan["view_points"])
    hands, n_cand = ctx.detect_samples(samples_xyz)
    flat = hands.reshape(-1)
    return flat[flat["valid"].astype(bool)], len(vox), n_cand


def test_raw_scans_through_the_batch_entry(oracle_mod):
    """gpd_detect_job.raw: workspace cut + voxeliser + normals inside gpd_hip_detect_batch, the voxelised cloud never leaving
    the device — five raw scans (two cameras, different sizes, one without a workspace) against the same scans through the single
    calls, byte for byte; then the stepwise route itself against the oracle on one of them."""
    rng = np.random.RandomState(12)
    scans, samples, wss = [], [], []
    for k, n in enumerate((60000, 90000, 40000, 120000, 75000)):
        cl = synth.make_cloud(500 + k, 30000)
        # a raw scan: every lattice point four times, moved by up to 1 mm (the voxeliser's input), shuffled
        parts = [(cl["xyz"] + rng.uniform(-1e-3, 1e-3, cl["xyz"].shape)).astype(np.float32) for _ in range(max(1, n // 30000))]
        xyz = np.concatenate(parts)[:n]
        perm = rng.permutation(len(xyz))
        xyz = np.ascontiguousarray(xyz[perm])
        cam = np.ones((2, len(xyz)), np.int32)
        cam[1] = rng.rand(len(xyz)) < 0.5
        cam[0, cam[1] == 1] = rng.rand(int(cam[1].sum())) < 0.5
        cam[0, (cam[0] == 0) & (cam[1] == 0)] = 1
        vp = np.array([[0.0, 0.0, 0.0], [0.25, -0.1, 0.05]])
        scans.append(dict(xyz=xyz, cam_source=cam, view_points=vp))
        obj = np.flatnonzero(cl["is_object"])
        samples.append(cl["xyz"][rng.choice(obj, 150, replace=False)].astype(np.float64))
        wss.append(None if k == 2 else np.array([-0.4, 0.4, -0.4, 0.4, -1.0, 1.0]))
    w = synth.lenet_weights(15, real=dict(np.load(os.path.join(GOLD, "lenet15_params.npz"))), trained_magnitude=True)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        want = [_stepwise(ctx, sc, sm, ws, 0.003, 0.03) for sc, sm, ws in zip(scans, samples, wss)]
        assert all(x[2] > 100 for x in want)
        for ws_batch, idx in ((wss[0], [0, 1, 3, 4]), (None, [2])):
            jobs, keep = ctx.raw_batch([scans[i] for i in idx], [samples[i] for i in idx], ws_batch, 0.003, 0.03)
            ctx._check(api.lib().gpd_hip_detect_batch(ctx._h, jobs, len(jobs)))
            for j, k, i in zip(jobs, keep, idx):
                hands = k[5][: j.num_hands]
                assert j.status == 0 and j.num_points_processed == want[i][1] and j.num_candidates == want[i][2]
                assert hands.tobytes() == want[i][0].tobytes(), i
        # the stepwise route against the oracle (scan 2: no workspace)
        sc, sm = scans[2], samples[2]
        vox, src = oracle_mod.voxelize(sc["xyz"], 0.003)
        cam = (sc["cam_source"][:, src] == 1).astype(np.int32)
        nrm = oracle_mod.estimate_normals(vox, cam, sc["view_points"], 0.03)
        p = oracle_mod.default_params(15)
        oh, on, _ = oracle_mod.detect(p, vox, nrm, cam, sc["view_points"], None, w, samples_xyz=sm) if "samples_xyz" in oracle_mod.detect.__code__.co_varnames else (None, None, None)
        if oh is not None:
            v = oh.reshape(-1)[oh.reshape(-1)["valid"].astype(bool)]
            assert on == want[2][2] and np.array_equal(v["position"], want[2][0]["position"]) and np.abs(v["score"] - want[2][0]["score"]).max() <= 1e-4
    finally:
        ctx.close()


def test_raw_krylon_through_the_batch_entry_matches_the_reference():
    """configs[0]'s cloud as a raw job: tutorials/krylon.pcd's points, voxeliser 0.003, normals 0.03, the pin's samples by coordinates,
    selectGrasps(50) — against what the reference's own detectGrasps returned (pin krylon_e2e_hands)."""
    import ref_cases as rcs
    pin = rcs.load_pin("extras")
    H = api.HAND_DTYPE
    rh = pin["krylon_e2e_hands"].view(H).reshape(-1)
    xyz = np.load(os.path.join(GOLD, "krylon_xyz.npz"))["xyz"].astype(np.float32)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(rcs.weights(15))
        ctx.set_lenet_mode(api.LENET_F32_CHAIN)  # the reference's scores under the fma definition, bit for bit
        vox = ctx.preprocess_cloud(xyz, voxel_size=0.003)[0]
        sm = vox[pin["krylon_e2e_samples"]].astype(np.float64)
        scan = dict(xyz=xyz, cam_source=np.ones((1, len(xyz)), np.int32), view_points=np.zeros((1, 3)))
        jobs, keep = ctx.raw_batch([scan, scan], [sm, sm[:40]], None, 0.003, 0.03, num_selected=50)
        ctx._check(api.lib().gpd_hip_detect_batch(ctx._h, jobs, 2))
        sel = keep[0][5][: jobs[0].num_hands]
        sel = sel[np.argsort(-sel["score"].astype(np.float64), kind="stable")]
        assert jobs[0].num_points_processed == len(vox) == 3366 and len(sel) == len(rh)
        assert np.array_equal(sel["score"], rh["score"]) and np.array_equal(sel["position"], rh["position"]) and np.array_equal(sel["frame"], rh["frame"])
        assert jobs[1].status == 0 and jobs[1].num_hands > 0
    finally:
        ctx.close()


def test_raw_job_cuts_the_samples_and_drops_nan_rows():
    """preprocessPointCloud starts with Cloud::removeNans (candidates_generator.cpp:17) and Cloud::filterWorkspace cuts
    the SAMPLES with the cloud (cloud.cpp:225-237, strict double comparisons).  A raw job whose samples straddle the workspace and
    whose scan carries NaN / Inf rows (an organised sensor scan's missing depth) must equal the stepwise route on the cleaned scan
    with the samples cut by hand — a sample just outside still finds neighbours, and its hand sets would shift every later set's
    shadow stream."""
    rng = np.random.RandomState(77)
    cl = synth.make_cloud(640, 30000)
    xyz = (cl["xyz"] + rng.uniform(-1e-3, 1e-3, cl["xyz"].shape)).astype(np.float32)
    obj = np.flatnonzero(cl["is_object"])
    pick = rng.choice(obj, 200, replace=False)
    sm = cl["xyz"][pick].astype(np.float64)
    # the cut runs through the middle of the sampled object along x; one sample sits exactly ON the bound (strict: dropped)
    xcut = float(np.median(sm[:, 0]))
    on_bound = int(np.argmin(np.abs(sm[:, 0] - xcut)))
    xcut = float(sm[on_bound, 0])
    ws = np.array([-1.0, xcut, -1.0, 1.0, -1.0, 1.0])
    inside = (sm[:, 0] > ws[0]) & (sm[:, 0] < ws[1]) & (sm[:, 1] > ws[2]) & (sm[:, 1] < ws[3]) & (sm[:, 2] > ws[4]) & (sm[:, 2] < ws[5])
    assert 40 < inside.sum() < 160 and not inside[on_bound]
    # NaN / Inf rows scattered through the scan
    dirty = xyz.copy()
    bad = rng.choice(len(dirty), 500, replace=False)
    dirty[bad[:300]] = np.nan
    dirty[bad[300:400], 2] = np.inf
    dirty[bad[400:], 0] = -np.inf
    clean_rows = np.ones(len(dirty), bool)
    clean_rows[bad] = False
    cam = np.ones((1, len(xyz)), np.int32)
    vp = np.zeros((1, 3))
    w = synth.lenet_weights(15, real=dict(np.load(os.path.join(GOLD, "lenet15_params.npz"))), trained_magnitude=True)
    ctx = api.Context(api.default_params(15))
    try:
        ctx.set_lenet_weights(w)
        for ws_job in (ws, None):
            keep_s = inside if ws_job is not None else np.ones(len(sm), bool)
            want, n_vox, n_cand = _stepwise(ctx, dict(xyz=np.ascontiguousarray(xyz[clean_rows]), cam_source=cam[:, clean_rows], view_points=vp),
                                            sm[keep_s], ws_job, 0.003, 0.03)
            jobs, keep = ctx.raw_batch([dict(xyz=dirty, cam_source=cam, view_points=vp)], [sm], ws_job, 0.003, 0.03)
            ctx._check(api.lib().gpd_hip_detect_batch(ctx._h, jobs, 1))
            j = jobs[0]
            assert j.status == 0 and j.num_samples_processed == int(keep_s.sum()) and j.num_points_processed == n_vox
            assert j.num_candidates == n_cand and n_cand > 50
            assert keep[0][5][: j.num_hands].tobytes() == want.tobytes()
        # the standalone entry still refuses a non-finite cloud (it is filterWorkspace + voxelizeCloud, not removeNans)
        with pytest.raises(api.GpdHipError):
            ctx.preprocess_cloud(dirty, cam, None, 0.003)
    finally:
        ctx.close()
