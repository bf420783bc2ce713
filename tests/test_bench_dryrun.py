"""bench.py end to end without a GPU: bench.main() itself — argument handling, the legs' bookkeeping, everything that turns
measurements into the ONE JSON line — executed against an oracle-backed stand-in for the context (tests/fake_context.py), so
that an edit to bench.py cannot cost a round its bench result.  The numbers are not measurements; the contract's keys, their
types and the arithmetic between them are checked."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, **extra_env):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dryrun_harness.py")] + list(args), capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]  # stdout carries exactly one line
    return json.loads(lines[0])


def _finite(x):
    return isinstance(x, (int, float)) and math.isfinite(x)


def test_default_line_assembles(oracle_mod):
    d = _run("--points", "6000", "--candidates", "160", "--steps", "2", "--warmup", "1", "--batch-clouds", "2", "--batch-passes", "2",
             "--batch-samples", "30", "--cpu-samples", "24")
    # the driver's contract
    assert d["metric"].startswith("15-ch grasp candidates scored/sec") and d["unit"] == "candidates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "bf16" in d["dtype"] and "int8" in d["dtype"]
    assert _finite(d["value"]) and d["value"] > 0 and _finite(d["ms_per_step"]) and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["candidates_per_gpu"] == 160
    n = d["config"]["candidates_per_gpu"]
    # roofline: the dominant LeNet kernel, its numbers consistent with each other and with `kernels`
    r = d["roofline"]
    assert r["kernel"] == "conv2_bf16_kernel" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and r["traffic"] is None
    k = d["kernels"][r["kernel"]]
    assert r["frac"] == r["achieved"] / r["peak"] and r["achieved"] == k["executed_Tops"] and r["launch_ms"] == k["ms"]
    assert abs(r["ops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12 - r["achieved"]) < 1e-9 * r["achieved"]
    fe = r["f32_equivalent"]
    assert abs(fe["achieved_TFLOPs"] - r["algorithmic_flops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) < 1e-9 * fe["achieved_TFLOPs"]
    assert abs(fe["frac_of_pipe_peak"] / fe["ceiling_frac_of_pipe_peak"] - r["frac"]) < 1e-12
    assert r["algorithmic_flops_per_launch"] == 2.0 * 50 * 500 * 24 * 24 * n
    for name in ("grasp_image_kernel", "lenet_forward", "conv1_i8_kernel", "conv2_bf16_kernel", "fc1_bf16_kernel", "fc2_score_kernel", "search"):
        assert _finite(d["kernels"][name]["ms"]) and d["kernels"][name]["ms"] > 0, name
    # the stand-in's stage times are the committed line's, scaled to the list: the same fractions come out
    assert abs(r["frac"] - (3 * 36 * 96 + 24 * 48) * 16384.0 * 5000 / 0.72e-3 / 1e12 / 2500.0) < 1e-9
    # the legs around the headline
    s = d["scores_timed_list"]
    assert s["images"] == n and s["within_1e-4"] is True and s["f32_chain_mode_bit_identical_to_oracle"] is True
    assert s["timed_scores_reproduced_by_gpd_hip_score"] is True and s["max_abs_oracle_chain_minus_float64"] < 1e-4
    # (round 6) the same images under SURVEY 8d's own weight set: relative to max |score| (~1000: one f32 ulp is 6e-5 there)
    v = s["survey_8d_weights"]
    assert v["max_abs_score"] > 100 and v["max_rel_split_minus_float64"] < 1e-5 and v["max_rel_oracle_chain_minus_float64"] < 1e-5
    assert v["f32_chain_mode_bit_identical_to_oracle"] is True
    assert d["detect_end_to_end"]["candidates"] > n and d["search"]["samples"] == d["config"]["samples"]
    assert d["preprocess"]["points"] == 120000 and 0 < d["preprocess"]["kept"] <= 120000
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "candidates/s" and c["cores"] >= 1 and _finite(c["value"]) and c["value"] > 0 and "samples" in c["sample"]
    assert _finite(c["preprocess_ms"]) and _finite(c["normals_ms"])
    # the reference's own sources beside it: timed in the run where oracle/_ref is in the tree (the build container; the GPU box
    # when the snapshot carries it), else the committed figure
    rs = c["reference_sources"]
    from oracle import ref
    if ref.available():
        assert rs["kind"] == "reference" and rs["cores"] == 1 and 0 < rs["value"] < 1000 and "measured in this run" in rs["sample"]
        assert rs["committed_full_list"]["value"] > 0
    else:
        assert "NOT measured in this run" in rs["note"]
    # the batch legs (two clouds per pass here): gpd_hip_detect_batch through the binding's own job arrays
    b = d["batch_end_to_end"]
    assert b["clouds"] == 4 and b["passes"]["n"] == 2 and b["passes"]["clouds_per_pass"] == 2 and b["candidates"] > 0
    assert b["cand_per_s"] == b["candidates"] / b["wall_s"] and b["passes"]["buffer_growths_in_timed_passes"] == 0
    assert abs(b["passes"]["host_ms_slowest_pass"]["total"] - 8.8) < 1e-3  # the stand-in's stamps through _host_split
    br = d["batch_raw_end_to_end"]
    assert br["clouds_per_pass"] == 2 and br["raw_points_per_cloud"] == 120000 and br["passes"] == 3 and len(br["wall_ms_per_pass"]) == 3
    assert all(119000 < m <= 120000 for m in br["points_after_preprocessing"]) and br["ms_per_cloud"]["min"] <= br["ms_per_cloud"]["median"]
    assert d["fallbacks"]["lenet_passes"] == 1


def test_config_line_without_the_cpu_legs(oracle_mod):
    """--config 3b (12 channels) with --cpu-samples 0: the line of a parity config — no accuracy leg, no CPU baseline, no preprocess leg."""
    d = _run("--config", "3b", "--points", "5000", "--candidates", "100", "--steps", "1", "--warmup", "0", "--batch-clouds", "0", "--cpu-samples", "0")
    assert d["metric"].startswith("12-ch") and d["config"]["channels"] == 12 and d["config"]["candidates_per_gpu"] == 100
    assert "cpu_baseline" not in d and "scores_timed_list" not in d and "preprocess" not in d
    assert "batch_end_to_end" not in d and "batch_raw_end_to_end" not in d  # --batch-clouds 0
    assert d["roofline"]["kernel"] in ("conv2_bf16_kernel", "conv1_i8_kernel") and 0 < d["roofline"]["frac"] < 1
    assert d["kernels"]["conv1_i8_kernel"]["algorithmic_flops"] == 2.0 * 20 * 25 * 12 * 56 * 56 * 100


def test_off_lattice_side_config(oracle_mod):
    """--config 2o: configs[1]'s scene with sensor-like coordinates; the workload says so."""
    d = _run("--config", "2o", "--points", "5000", "--candidates", "100", "--steps", "1", "--warmup", "0", "--batch-clouds", "0", "--cpu-samples", "0")
    assert "off the 3 mm lattice" in d["config"]["workload"] and d["config"]["channels"] == 15 and d["config"]["candidates_per_gpu"] == 100


def test_accuracy_leg_is_bounded_on_a_long_list(oracle_mod):
    """configs[3] times 50 000 candidates; the score-accuracy leg then looks at the first ACCURACY_LEG_MAX of them only (here 100 of
    160): the images it asks for and the timed scores it compares them with are the same candidates, in the same order."""
    d = _run("--points", "6000", "--candidates", "160", "--steps", "1", "--warmup", "0", "--batch-clouds", "0", "--cpu-samples", "24",
             GPD_DRYRUN_ACCURACY_MAX="100")
    s = d["scores_timed_list"]
    assert d["config"]["candidates_per_gpu"] == 160 and s["images"] == 100
    assert s["max_abs_hip_minus_oracle_chain"] == 0.0 and s["timed_scores_reproduced_by_gpd_hip_score"] is True  # same candidates, same order


def test_batch_mode_line(oracle_mod):
    """--mode batch (configs[4]): clouds over the ranks, the line of the batch entry."""
    d = _run("--mode", "batch", "--clouds", "3", "--steps", "2", "--warmup", "1", "--batch-samples", "30")
    assert d["metric"].startswith("15-ch grasp candidates generated+scored/sec") and d["scaling"] == "strong" and d["steps"] == 2
    b = d["batch_end_to_end"]
    assert b["clouds"] == 6 and d["value"] == b["cand_per_s"] and abs(d["ms_per_step"] - b["wall_s"] / 2 * 1e3) < 1e-9
    assert d["roofline"]["bound"] == "mfma" and _finite(d["roofline"]["frac"]) and d["config"]["clouds"] == 3
    # (round 6) the stage is priced against the peaks of the pipes it runs on: a utilisation, never above 1
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["unit"] == "TOP/s" and d["roofline"]["peak"] > 2500


def test_two_ranks_under_the_drivers_launcher(oracle_mod):
    """The driver's N > 1 command line (torch.distributed.run, one rank per GPU) on the stand-in, gloo instead of RCCL: rank 0 prints
    the one line, `value` is the whole job's (both ranks' candidates over the slowest rank's time), the CPU legs stay out."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "bench_dryrun_harness.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--points", "6000", "--candidates", "160", "--steps", "2", "--warmup", "1", "--batch-clouds", "2", "--batch-passes", "2",
                        "--batch-samples", "30"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["candidates_per_gpu"] == 160
    assert abs(d["value"] - 2 * 160 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]  # both ranks' candidates / max time
    assert "cpu_baseline" not in d and "scores_timed_list" not in d and "batch_raw_end_to_end" not in d
    assert d["batch_end_to_end"]["clouds"] == 8 and "host_binding_rank0" in d  # 2 ranks x 2 clouds x 2 passes


def test_smoke_entry_runs_against_the_stand_in(oracle_mod):
    """__graft_entry__.smoke() (the driver's round-end check on cuda:0) executed against the stand-in: its own bookkeeping holds."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpd_amd import api; import fake_context; api.Context = fake_context.FakeContext\n"
            "import __graft_entry__ as g; g.smoke()\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0 and "smoke ok: 60 candidates" in p.stdout, p.stdout[-500:] + p.stderr[-2000:]


def test_a_failing_reference_timing_does_not_cost_the_line(oracle_mod):
    """cpu_baseline.reference_sources is timed in a child process: a reference library that cannot even be loaded (here: a file
    that is not one) leaves the line intact, the committed figure in its place and the failure on record."""
    d = _run("--points", "6000", "--candidates", "120", "--steps", "1", "--warmup", "0", "--batch-clouds", "0", "--cpu-samples", "16",
             GPD_REF_LIB="/bin/true")
    rs = d["cpu_baseline"]["reference_sources"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert "live_attempt_failed" in rs and "NOT measured in this run" in rs["note"] and rs["value"] > 0
