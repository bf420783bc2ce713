"""The N > 1 path of bench.py on the ONE GPU of the test box: two ranks under torch.distributed.run, both on device 0
(`--devices 0,0`; gloo carries the barrier / reductions because RCCL refuses two ranks on one device), configs[4]'s mode
(`--mode batch`: cloud i -> rank i mod N, no collective on the data path).  What the driver's 8-GPU run relies on is
checked here: the self-launch, the rendezvous, the sharding, the whole-job sums and the per-rank spread in rank 0's ONE
line — against the oracle's candidate counts cloud by cloud."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gpd_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_batch_mode_two_ranks_on_one_device(oracle_mod):
    clouds, samples, steps = 8, 300, 2
    env = dict(os.environ)
    env.pop("GPD_BENCH_DRYRUN", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "batch", "--clouds", str(clouds),
                          "--batch-samples", str(samples), "--steps", str(steps), "--warmup", "1", "--devices", "0,0", "--dist-backend", "gloo"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "strong" and d["unit"] == "candidates/s"
    b = d["batch_end_to_end"]
    # the oracle's candidate count of every cloud of the job (seed 1234 + i, the list bench.py builds)
    p = oracle_mod.default_params(15)
    want = 0
    for i in range(clouds):
        cl = synth.make_cloud(1234 + i, 30000)
        si = synth.sample_indices(cl, samples)
        h = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
        want += int(h["valid"].sum())
    assert b["clouds"] == clouds * steps and b["clouds_per_rank"] == clouds // 2 * steps
    assert b["candidates"] == want * steps
    assert b["rank_clouds_per_s"]["min"] > 0 and b["rank_clouds_per_s"]["max"] >= b["rank_clouds_per_s"]["min"]
    assert b["passes"]["buffer_growths_in_timed_passes"] == 0 and len(b["passes"]["wall_ms_rank0"]) == steps
    assert abs(d["value"] - b["candidates"] / b["wall_s"]) <= 1e-6 * d["value"]
    assert "host_binding_rank0" in d
