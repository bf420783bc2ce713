"""bench.py's host-side helpers (no GPU): the ip1 tile rule it mirrors from lenet.hip, conv1's live-pair statistic,
the staleness check of the committed PMC numbers, the cloud -> rank assignment of the batch mode."""
import json

import pytest
import os

import numpy as np

import bench
from gpd_amd import dist as gdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fc1_tile_rule_matches_the_kernel_launcher():
    # lenet_fast.hip fc1f_pick_nt: image tiles of 32 NT rows (NT <= 5), 8 workgroups each, filling 256 CUs in whole rounds
    assert [bench._fc1_tile(n) for n in (1, 37, 1024, 1025, 5000, 6077, 8192, 10000, 16384, 50000)] == [1, 1, 1, 2, 5, 3, 4, 5, 4, 5]
    src = open(os.path.join(ROOT, "gpd_amd", "csrc", "lenet_fast.hip")).read()
    assert "(n + 32 * r * 32 - 1) / (32 * r * 32)" in src  # the rule restated above is the one in the launcher


def test_executed_mfma_work_matches_the_kernels_constants():
    """bench.py prices the split path's kernels by the MFMA operations they execute: the tile / step counts it multiplies
    are the kernels' own constants."""
    src = open(os.path.join(ROOT, "gpd_amd", "csrc", "lenet_fast.hip")).read()
    for needle in ("F1_TILES = 28 * 7", "F1_KS = 7, F1_MT = 5", "wave < 4 ? 18 : 12;", "F2_LUNITS = 24", "for (int step = 0; step < 8; step++)", "for (int ks = 0; ks < 16; ks++)", "kLenetXld"):
        assert needle in src, needle
    w = bench.lenet_mfma_work(15)
    assert w["conv1_i8_kernel"]["executed"] == 196 * 7 * 5 * 16 * 16 * 64 * 2
    # (round 6) waves 0-3: 18 pixel tiles (group 2: 20 + 16), waves 4-6: 12, each x 16 k-steps x 6 piece products; wave 7: 24 units x 8 steps x 6
    assert w["conv2_bf16_kernel"]["executed"] == ((4 * 18 + 3 * 12) * 16 * 6 + 24 * 8 * 6) * 16 * 16 * 32 * 2
    assert w["fc1_bf16_kernel"]["executed"] == 2.0 * 512 * 7296 * 6
    for v in w.values():
        assert 0.7 < v["algorithmic_split"] / v["executed"] <= 1.0


def test_bench_launches_itself_for_several_gpus(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command): bench.py re-runs itself
    under torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1, and prints ONE line.  GPD_BENCH_DRYRUN stops
    each rank where it would create its context (gloo instead of RCCL): launch, rendezvous, barrier, max / sum over ranks."""
    import subprocess
    import sys
    env = dict(os.environ, GPD_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 2, "world": 2, "max": 2.0, "sum": 30.0, "batch_clouds": 32}
    # a launcher that disagrees with --gpus is refused with a message, not an AssertionError
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_pmc_numbers_are_dropped_when_a_kernel_source_changes(tmp_path, monkeypatch):
    good = {"source_hashes": bench.source_hashes(),
            "kernels": {"void gpd::conv1_i8_kernel<15>(x)": {"hbm_bytes_per_launch": 8e8},
                        "void gpd::fc1_bf16_kernel<5>(x)": {"hbm_bytes_per_launch": 2.7e8},
                        "void gpd::fc1_bf16_kernel<3>(x)": {"hbm_bytes_per_launch": 3.0e8}}}
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps(good))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    t = bench._pmc_traffic(5000)
    assert t["conv1_i8"] == 8e8 and t["fc1_bf16"] == 2.7e8      # the ip1 instantiation this n runs with
    assert "default workload" in bench._pmc_traffic(6077)["note"] and "default workload" in bench._pmc_traffic(5000, 12)["note"]
    good["source_hashes"]["gpd_amd/csrc/lenet.hip"] = "0" * 40
    f.write_text(json.dumps(good))
    t = bench._pmc_traffic(5000)
    assert "conv1_i8" not in t and "stale" in t["note"]


def test_committed_profiles_belong_to_the_committed_kernels():
    """profiles/r05_traffic.json / r05_pmc_sq.json are only reported while the kernel sources are the ones they were
    measured on; a commit that changes a kernel has to re-collect them (profiles/collect_r05.sh, profiles/pmc_sq.sh)."""
    for name in ("r05_traffic.json", "r05_pmc_sq.json"):
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            pytest.skip(name + " not collected yet")
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        if d["source_hashes"] != bench.source_hashes():
            # bench.py drops the numbers of a stale file by itself (previous test); a kernel change between two collection
            # passes is work in progress, not a failure of the suite — but it is reported
            pytest.skip(name + " is stale: a kernel source changed since it was collected; bench.py reports no PMC numbers until it is re-collected")


def test_traffic_totals_take_every_kernel_of_a_stage():
    """The per-stage totals pick their kernels by name: with the committed profile every kernel of the image stage, of
    LeNet and of the search must be found (the shadow image kernel once dropped out of `image` when it gained a second
    template argument: 505 instead of 623 MB), and a live measurement reports the committed figures next to its own."""
    path = os.path.join(ROOT, "profiles", "r05_traffic.json")
    if not os.path.exists(path):
        pytest.skip("no committed traffic profile")
    d = json.load(open(path))["kernels"]
    t = bench._traffic_totals(d, 5000, "x")

    def one(sub):
        hits = [v["hbm_bytes_per_launch"] for k, v in d.items() if sub in k]
        assert len(hits) == 1, (sub, [k for k in d if sub in k])
        return hits[0]

    img = one("grasp_image_kernel<false>") + one("shadow_image_kernel<6144") + one("shadow_set_kernel")
    assert t["image"] == pytest.approx(img) and one("shadow_image_kernel<6144") > 5e7
    assert t["lenet"] == pytest.approx(one("conv1_i8") + one("conv2_bf16") + one("fc1_bf16_kernel<5>") + one("fc1_combine") + one("fc2_score"))
    assert t["search"] == pytest.approx(one("neighbourhood_kernel<false>") + one("hand_eval_kernel") + one("plan_kernel") + one("centre_kernel"))


def test_live_pmc_falls_back_to_the_committed_profile(monkeypatch):
    """--live-pmc without a working rocprofv3 (no GPU here): the committed, stamped profile is reported instead, and an
    exception inside the measurement does not take the bench line down."""
    monkeypatch.setattr(bench, "_live_pmc_kernels", lambda: None)
    a = bench._pmc_traffic(5000, live=True)
    b = bench._pmc_traffic(5000, live=False)
    assert a == b

    def boom():
        raise RuntimeError("no counters")
    monkeypatch.setattr(bench, "_live_pmc_kernels", boom)
    assert bench._pmc_traffic(5000, live=True) == b
    fake = {"void gpd::conv1_i8_kernel<15>(x)": {"hbm_bytes_per_launch": 7e8}}
    monkeypatch.setattr(bench, "_live_pmc_kernels", lambda: fake)
    c = bench._pmc_traffic(5000, live=True)
    assert c["conv1_i8"] == 7e8 and c["source"].startswith("live")
    if "conv1_i8" in b:
        assert c["committed_file"]["conv1_i8"] == b["conv1_i8"]


def test_batch_mode_cloud_assignment():
    for world in (1, 2, 4, 8):
        seen = sorted(c for r in range(world) for c in gdist.clouds_of_rank(256, r, world))
        assert seen == list(range(256))
        assert all(len(gdist.clouds_of_rank(256, r, world)) == 256 // world for r in range(world))


def test_float64_reference_of_the_score_leg_is_chunk_independent(lenet15_real, oracle_mod):
    """bench._lenet_f64 walks the list 500 images at a time (bounded host memory): a list that straddles two chunk
    boundaries gives what each image gives alone (to float64 summation order), and agrees with the oracle's f32 chain to f32 summation noise."""
    rng = np.random.RandomState(5)
    imgs = (rng.randint(0, 256, (1003, 60, 60, 15)) * (rng.rand(1003, 60, 60, 15) < 0.2)).astype(np.uint8)
    got = bench._lenet_f64(imgs, lenet15_real)
    assert got.shape == (1003,) and got.dtype == np.float64
    for i in (0, 499, 500, 999, 1000, 1002):
        assert abs(got[i] - bench._lenet_f64(imgs[i:i + 1], lenet15_real)[0]) < 1e-12  # torch's f64 GEMM order moves with the batch size
    chain = oracle_mod.lenet(imgs, lenet15_real)
    assert np.abs(got - chain).max() < 1e-4
    assert bench._lenet_f64(imgs[:0], lenet15_real).shape == (0,)


def test_roofline_object_reproduces_the_committed_line():
    """lenet_kernel_entry / lenet_roofline (the functions bench.py's main builds `kernels` and `roofline` with) fed the launch
    time of the committed line of the final round-6 kernels give that line's numbers, and the f32-equivalent block carries both peaks."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")))
    rf, n = line["roofline"], line["config"]["candidates_per_gpu"]
    dom = rf["kernel"]
    work = bench.lenet_mfma_work(15)
    kd = bench.lenet_kernel_entry(rf["launch_ms"] / 1e3, work[dom]["algorithmic"], n, work[dom])
    for k in ("ms", "algorithmic_flops", "achieved_TFLOPs", "frac_f32", "executed_ops", "executed_Tops", "frac_pipe"):
        assert kd[k] == pytest.approx(line["kernels"][dom][k], rel=1e-12), k
    got = bench.lenet_roofline(dom, kd, work[dom], rf["traffic"])
    for k in ("kernel", "bound", "peak", "unit", "pipe", "traffic"):
        assert got[k] == rf[k], k
    for k in ("achieved", "frac", "ops_per_launch", "launch_ms", "algorithmic_flops_per_launch", "useful_share_of_executed"):
        assert got[k] == pytest.approx(rf[k], rel=1e-12), k
    fe = got["f32_equivalent"]
    assert fe["achieved_TFLOPs"] == pytest.approx(rf["f32_equivalent"]["achieved_TFLOPs"], rel=1e-12)
    assert fe["frac_of_f32_peak"] == pytest.approx(fe["achieved_TFLOPs"] / 157.3, rel=1e-12)
    assert fe["frac_of_pipe_peak"] == pytest.approx(fe["achieved_TFLOPs"] / got["peak"], rel=1e-12)
    assert fe["frac_of_pipe_peak"] / fe["ceiling_frac_of_pipe_peak"] == pytest.approx(got["frac"], rel=1e-12)  # = the pipe's utilisation
    # ip2 has no matrix instructions: a plain entry
    e = bench.lenet_kernel_entry(25e-6, 2000.0, n, None)
    assert "pipe" not in e and e["achieved_TFLOPs"] == pytest.approx(2000.0 * n / 25e-6 / 1e12)
    json.dumps(got)
