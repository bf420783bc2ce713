"""The C++ host layer (gpd_amd/host): GraspDetector / ConfigFile / Cloud / detect_grasps CLI."""
import os
import subprocess

import numpy as np
import pytest

from gpd_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gpd_amd", "host", "detect_grasps")


def _write_case(tmp, cl, w, num_samples, num_selected, channels=15, min_inliers=0, extra=""):
    params = tmp / "params"
    params.mkdir()
    names = dict(c1w="conv1_weights", c1b="conv1_biases", c2w="conv2_weights", c2b="conv2_biases", f1w="ip1_weights",
                 f1b="ip1_biases", f2w="ip2_weights", f2b="ip2_biases")
    for k, v in names.items():
        np.asarray(w[k], "<f4").tofile(str(params / (v + ".bin")))
    (tmp / "hand_geometry.cfg").write_text("# hand\nfinger_width = 0.01\nhand_outer_diameter = 0.12\nhand_depth = 0.06\n"
                                           "hand_height = 0.02\ninit_bite = 0.01\n")
    (tmp / "image_geometry.cfg").write_text("volume_width = 0.10\nvolume_depth = 0.06\nvolume_height = 0.02\nimage_size = 60  \n"
                                            "image_num_channels = %d  # channels\n" % channels)
    cfg = tmp / "params.cfg"
    cfg.write_text("# test config in the format of cfg/eigen_params.cfg\n"
                   "hand_geometry_filename = hand_geometry.cfg\nimage_geometry_filename = image_geometry.cfg\n"
                   "weights_file = params/\ndevice = 1\ncamera_position = 0 0 0\nuse_file_normals = 1\n"
                   "num_samples = %d\nnum_threads = 4\nnn_radius = 0.01\nnum_orientations = 8\nnum_finger_placements = 10\n"
                   "hand_axes = 2\ndeepen_hand = 1\nfriction_coeff = 20\nmin_viable = 6\nmin_aperture = 0.0\nmax_aperture = 0.085\n"
                   "workspace_grasps = -1 1 -1 1 -1 1\nmin_inliers = %d\nnum_selected = %d\nplot_normals = 0\n%s"
                   % (num_samples, min_inliers, num_selected, extra))
    pcd = tmp / "cloud.pcd"
    P = len(cl["xyz"])
    with open(str(pcd), "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\n"
                "SIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (P, P))
        for p, n in zip(cl["xyz"], cl["normals"]):
            f.write("%.9g %.9g %.9g %.9g %.9g %.9g\n" % (p[0], p[1], p[2], n[0], n[1], n[2]))
    return cfg, pcd


def _subsample_indices(n, m, seed=0):
    """util::Cloud::subsample of the host layer (xorshift Fisher-Yates), restated."""
    idx = list(range(n))
    s = 0x9E3779B97F4A7C15 ^ seed
    M = (1 << 64) - 1
    m = min(m, n)
    for i in range(m):
        s ^= (s << 13) & M
        s ^= s >> 7
        s ^= (s << 17) & M
        j = i + s % (n - i)
        idx[i], idx[j] = idx[j], idx[i]
    return np.array(idx[:m], np.int32)


def test_cli_usage_and_no_gpu_fails_loudly(tmp_path):
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    out = subprocess.run([CLI], capture_output=True, text=True)
    assert out.returncode != 0 and "Usage: detect_grasps CONFIG_FILE PCD_FILE" in out.stdout
    import torch
    if torch.cuda.is_available():
        return
    cl = synth.make_cloud(5, 2000)
    cfg, pcd = _write_case(tmp_path, cl, synth.lenet_weights(15), 10, 5)
    out = subprocess.run([CLI, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode != 0 and "ERROR" in out.stdout  # no device -> error, not a CPU result


@pytest.mark.gpu
def test_detect_grasps_cli_matches_oracle(tmp_path, oracle_mod, lenet15_real):
    cl = synth.make_cloud(99, 12000)
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 150, 20)
    out = subprocess.run([CLI, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "======== RUNTIMES ========" in out.stdout
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("GRASP ")]
    assert len(got) == 20
    # the PCD is ASCII with 9 significant digits: float32 round-trips exactly
    si = _subsample_indices(len(cl["xyz"]), 150)
    p = oracle_mod.default_params(15)
    hands, n, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real)
    v = hands[hands["valid"].astype(bool)]
    order = np.argsort(-v["score"], kind="stable")[:20]
    want = v[order]
    gs = np.array([float(g[1]) for g in got])
    assert np.abs(gs - want["score"]).max() <= 1e-4
    assert np.all(np.diff(gs) <= 0)
    gp = np.array([[float(x) for x in g[2:5]] for g in got])
    assert np.allclose(gp, want["position"], rtol=1e-12, atol=1e-15)
    assert [int(g[6]) for g in got] == want["finger_placement_index"].tolist()


@pytest.mark.gpu
def test_cli_direction_filter_and_clustering(tmp_path, oracle_mod, lenet15_real):
    """detectGrasps with filter_approach_direction = 1 and min_inliers = 1 (grasp_detector.cpp:247-255,
    283-303): search -> workspace filter -> direction filter -> images -> scores -> top-k -> clusters."""
    cl = synth.make_cloud(99, 12000)
    direction, thresh, k = np.array([0.0, 0.0, -1.0]), 1.2, 120
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 300, k, min_inliers=1,
                           extra="filter_approach_direction = 1\ndirection = 0 0 -1\nthresh_rad = %g\n" % thresh)
    out = subprocess.run([CLI, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("GRASP ")]
    si = _subsample_indices(len(cl["xyz"]), 300)
    p = oracle_mod.default_params(15)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
    app = hands["frame"].reshape(hands.shape + (3, 3))[..., :, 0]
    with np.errstate(invalid="ignore"):
        ang = np.arccos(app @ direction)  # |dot| a hair above 1 -> NaN -> `angle > thresh` is false: kept, as in the reference
    n_before = int(hands["valid"].sum())
    hands["valid"] &= (~(ang > thresh)).astype(np.uint8)
    assert 0 < hands["valid"].sum() < n_before           # the filter removed some, not all
    lines = [l for l in out.stdout.splitlines() if l.startswith("Number of grasp candidates")]
    assert ("Number of grasp candidates with correct approach direction: %d" % hands["valid"].sum()) in lines, lines
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    sc = oracle_mod.lenet(img, lenet15_real)
    flat = hands.reshape(-1)[cand]
    order = np.argsort(-sc, kind="stable")[:k]
    sel, ssc = flat[order], sc[order].astype(np.float64)
    clusters, csc, _ = oracle_mod.find_clusters(sel, ssc, 1, False)
    assert len(clusters) > 3
    o2 = np.argsort(-csc, kind="stable")
    clusters, csc = clusters[o2], csc[o2]
    assert len(got) == len(clusters)
    gs = np.array([float(g[1]) for g in got])
    assert np.abs(gs - csc).max() <= 2e-4                 # bound of up to 120 float scores within 1e-4 each
    gp = np.array([[float(x) for x in g[2:5]] for g in got])
    assert np.allclose(gp, clusters["position"], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_host_selftest_samples_prune_groundtruth(tmp_path, lenet15_real):
    """GraspDetector members the CLI does not reach: Cloud::setSamples, pruneGraspCandidates, evalGroundTruth."""
    exe = os.path.join(ROOT, "gpd_amd", "host", "host_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    cl = synth.make_cloud(99, 12000)
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 120, 4000)
    out = subprocess.run([exe, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    last = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    assert out.returncode == 0 and last.startswith("SELFTEST OK"), last + out.stderr[-1000:]
    assert int(last.split()[4]) > 50  # candidates re-evaluated against the "ground truth" cloud


@pytest.mark.gpu
def test_cli_on_raw_krylon_pcd(tmp_path, oracle_mod, lenet15_real):
    """configs[0] through the product CLI: an x y z PCD without normals -> voxelise, GPU normals,
    subsample(500), detect.  Compared with the oracle run on the same preprocessing."""
    xyz = np.load(os.path.join(ROOT, "tests", "golden", "krylon_xyz.npz"))["xyz"]
    cl = dict(xyz=xyz, normals=np.zeros_like(xyz))
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 500, 5)
    with open(str(pcd), "w") as f:  # rewrite as a plain x y z cloud like tutorials/krylon.pcd
        f.write("# .PCD v.7 - Point Cloud Data file format\nVERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                "WIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA ascii\n" % (len(xyz), len(xyz)))
        for p in xyz:
            f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
    out = subprocess.run([CLI, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Voxelized cloud: 3366" in out.stdout
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("GRASP ")]
    assert len(got) == 5
    v, _ = oracle_mod.voxelize(xyz, 0.003)
    n = oracle_mod.estimate_normals(v)
    si = _subsample_indices(len(v), 500)
    p = oracle_mod.default_params(15)
    hands, _, _ = oracle_mod.detect(p, v, n, np.ones((1, len(v)), np.int32), np.zeros((1, 3)), si, lenet15_real)
    vv = hands[hands["valid"].astype(bool)]
    want = vv[np.argsort(-vv["score"], kind="stable")[:5]]
    assert np.abs(np.array([float(g[1]) for g in got]) - want["score"]).max() <= 1e-4


CEM = os.path.join(ROOT, "gpd_amd", "host", "cem_detect_grasps")


@pytest.mark.gpu
@pytest.mark.parametrize("method,min_inliers", [(0, 0), (1, 0), (0, 1)])
def test_cem_detect_grasps_matches_oracle_replay(tmp_path, oracle_mod, lenet15_real, method, min_inliers):
    """SequentialImportanceSampling::detectGrasps (sequential_importance_sampling.cpp:54-187) through the
    product CLI; the samples it drew are dumped and replayed through the oracle: initial candidates,
    three sampling rounds by coordinates, ONE classification of all collected hand sets, clustering."""
    assert os.path.exists(CEM), "run __graft_entry__.build()"
    cl = synth.make_cloud(99, 12000)
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 100, 50, min_inliers=min_inliers,
                           extra="num_init_samples = 40\nnum_iterations = 3\nnum_samples_per_iteration = 40\nprob_rand_samples = 0.3\n"
                                 "standard_deviation = 0.02\nsampling_method = %d\nmin_score = -300\nrandom_seed = 7\n" % method)
    dump = tmp_path / "sis_samples.txt"
    env = dict(os.environ, GPD_SIS_DUMP=str(dump))
    out = subprocess.run([CEM, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("GRASP ")]
    # ---- replay
    lines = dump.read_text().splitlines()
    n_init = int(lines[0].split()[1])
    idx = np.array([int(x) for x in lines[1:1 + n_init]], np.int32)
    rounds, k = [], 1 + n_init
    while k < len(lines):
        n = int(lines[k].split()[2])
        rounds.append(np.array([[float(v) for v in l.split()] for l in lines[k + 1:k + 1 + n]], np.float64))
        k += 1 + n
    assert len(rounds) == 3 and all(len(r) == 40 for r in rounds)
    if method == 0:  # 28 Gaussian samples around known hand sets + 12 cloud points per round
        assert all(np.isin(r[28:].astype(np.float32), cl["xyz"][idx]).all() for r in rounds)
    p = oracle_mod.default_params(15)

    def live_sets(h):
        h = oracle_mod.filter_workspace(p, h)
        return h[h["valid"].any(axis=1)]

    sets = [live_sets(oracle_mod.search(p, cl["xyz"], cl["normals"], idx))]
    for r in rounds:
        sets.append(live_sets(oracle_mod.search_xyz(p, cl["xyz"], cl["normals"], r)))
    allh = np.concatenate(sets)
    assert len(allh) > 40
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], allh)
    sc = oracle_mod.lenet(img, lenet15_real)
    keep = sc > -300
    want, wsc = allh.reshape(-1)[cand][keep], sc[keep].astype(np.float64)
    assert len(want) > 5, (len(want), np.sort(sc)[::-1][:10])
    if min_inliers > 0:
        want, wsc, _ = oracle_mod.find_clusters(want, wsc, min_inliers, False)
    assert len(got) == len(want)
    gs = np.array([float(g[1]) for g in got])
    assert np.abs(gs - wsc).max() <= 2e-4
    gp = np.array([[float(x) for x in g[2:5]] for g in got])
    assert np.allclose(gp, want["position"], rtol=1e-9, atol=1e-12)
    assert [int(g[6]) for g in got] == want["finger_placement_index"].tolist()


@pytest.mark.gpu
def test_generate_candidates_cli_matches_oracle(tmp_path, oracle_mod, lenet15_real):
    exe = os.path.join(ROOT, "gpd_amd", "host", "generate_candidates")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    cl = synth.make_cloud(99, 12000)
    cfg, pcd = _write_case(tmp_path, cl, lenet15_real, 80, 5)
    out = subprocess.run([exe, str(cfg), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    got = [l.split() for l in out.stdout.splitlines() if l.startswith("CANDIDATE ")]
    si = _subsample_indices(len(cl["xyz"]), 80)
    hands = oracle_mod.search(oracle_mod.default_params(15), cl["xyz"], cl["normals"], si)
    want = [(s, j) for s in range(hands.shape[0]) for j in range(hands.shape[1]) if hands[s, j]["valid"]]
    assert [(int(g[1]), int(g[2])) for g in got] == want and len(want) > 50
    assert ("Generated %d grasp candidates." % len(want)) in out.stdout
    for g, (s, j) in zip(got, want):
        h = hands[s, j]
        assert int(g[3]) == h["finger_placement_index"] and int(g[8]) == h["half_antipodal"] and int(g[9]) == h["full_antipodal"]
        assert np.array_equal(np.array([float(x) for x in g[4:7]]), h["position"]) and float(g[7]) == h["grasp_width"]


@pytest.mark.gpu
def test_label_grasps_cli_matches_oracle(tmp_path, oracle_mod, lenet15_real):
    """label_grasps CONFIG PCD MESH: candidates on one cloud, every candidate checked again against the "mesh"
    cloud (here the same scene, so the labels are the search's own full-antipodal flags — rare, but present).
    Both files carry normals, which the tool negates like the reference does (so the files hold the inward
    normals, and the tool ends up with the outward ones)."""
    exe = os.path.join(ROOT, "gpd_amd", "host", "label_grasps")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    full = synth.make_cloud(1234, 30000)
    neg = dict(full)
    neg["normals"] = -full["normals"]
    cfg, pcd = _write_case(tmp_path, neg, lenet15_real, 1000, 5)
    out = subprocess.run([exe, str(cfg), str(pcd), str(pcd)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    got = [int(l.split()[-1]) for l in out.stdout.splitlines() if l.startswith("(") and "label:" in l]
    p = oracle_mod.default_params(15)
    si = _subsample_indices(len(full["xyz"]), 1000)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, full["xyz"], full["normals"], si))
    v = hands.reshape(-1)[hands.reshape(-1)["valid"].astype(bool)]
    labels, _ = oracle_mod.reevaluate(p, full["xyz"], full["normals"], v)
    assert np.array_equal(labels, v["full_antipodal"].astype(np.int32)) and labels.sum() > 0
    assert got == labels.tolist()
    assert ("LABELS %d / %d" % (labels.sum(), len(labels))) in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("channels,axes", [(15, None), (3, None), (15, [0, 1, 2])])
def test_test_grasp_image_tool_on_krylon_sample_3456(tmp_path, oracle_mod, channels, axes):
    """The reference's only built test program and its documented invocation (src/tests/test_grasp_image.cpp,
    README: `test_grasp_image ../tutorials/krylon.pcd 3456 1 ...`): raw krylon cloud, normals with radius 0.03
    then negated, sample index 3456, ONE orientation about axis 2 (or the axes given), an image per valid hand.
    The tool's frame / position / flags / images against the oracle on the same steps."""
    exe = os.path.join(ROOT, "gpd_amd", "host", "test_grasp_image")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    xyz = np.load(os.path.join(ROOT, "tests", "golden", "krylon_xyz.npz"))["xyz"]
    pcd = tmp_path / "krylon.pcd"
    with open(str(pcd), "w") as f:
        f.write("# .PCD v.7 - Point Cloud Data file format\nVERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                "WIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA ascii\n" % (len(xyz), len(xyz)))
        for p in xyz:
            f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
    dump = tmp_path / "images.bin"
    argv = [exe, str(pcd), "3456", "0", str(channels)] + [str(a) for a in (axes or [])]
    out = subprocess.run(argv, capture_output=True, text=True, cwd=str(tmp_path), timeout=300,
                         env=dict(os.environ, GPD_DUMP_IMAGES=str(dump)))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = out.stdout.splitlines()
    assert ("hand_axes: " + "".join("%d " % a for a in (axes or [2]))) in lines
    # ---- the oracle on the same steps
    p = oracle_mod.default_params(channels)
    p.num_orientations = 1
    p.num_hand_axes = len(axes or [2])
    for i, a in enumerate(axes or [2]):
        p.hand_axes[i] = a
    normals = -oracle_mod.estimate_normals(xyz, radius=0.03)
    cam, vp = np.ones((1, len(xyz)), np.int32), np.zeros((1, 3))
    hands = oracle_mod.search(p, xyz, normals, np.array([3456], np.int32))
    assert hands.shape == (1, len(axes or [2]))
    h = hands[0, 0]
    k = lines.index("grasp orientation:")
    frame = np.array([[float(v) for v in lines[k + 1 + r].split()] for r in range(3)])
    assert np.array_equal(frame, h["frame"].reshape(3, 3))
    assert np.array_equal(np.array([float(v) for v in lines[k - 1].split()[1:]]), h["sample"])
    assert np.array_equal(np.array([float(v) for v in lines[k + 4].split()[2:]]), h["position"])
    oimg, ocand = oracle_mod.images(p, xyz, normals, cam, vp, hands.copy())
    valid = [hands[0, j] for j in range(hands.shape[1]) if hands[0, j]["valid"]]
    got = [l.split() for l in lines if l.startswith("IMAGE ")]
    assert len(got) == len(ocand) == len(valid) >= 1
    raw = np.fromfile(str(dump), np.uint8).reshape(len(got), 60, 60, channels)
    assert np.array_equal(raw, oimg)
    for g, v in zip(got, valid):
        assert (int(g[3]), int(g[4]), int(g[5]), int(g[6])) == (v["slot"], v["finger_placement_index"], v["half_antipodal"], v["full_antipodal"])
    anti = [l for l in lines if l.startswith("Antipodal:")][0].split()[1:]
    assert [int(x) for x in anti] == [int(v["full_antipodal"]) for v in valid]
