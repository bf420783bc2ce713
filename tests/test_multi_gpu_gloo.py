"""The N>1 path on CPU: world_size 2, gloo.  Each rank owns cloud i with i mod 2 == rank,
runs the (oracle) candidate search on it, and the host-side gather in cloud order must equal
the single-process result — the sharding is collective-free and order-preserving."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from gpd_amd import synth, dist as gdist
import oracle
dist.init_process_group("gloo")
rank, world = gdist.rank_world()
NUM = 5
mine = gdist.clouds_of_rank(NUM, rank, world)
p = oracle.default_params(3)
res = []
for c in mine:
    cl = synth.make_cloud(1234 + c, 3000)
    si = synth.sample_indices(cl, 6)
    h = oracle.search(p, cl["xyz"], cl["normals"], si)
    res.append([int(h["valid"].sum()), float(h["grasp_width"].sum())])
allres = gdist.gather_in_cloud_order(dist, NUM, rank, world, res)
tmax, usum = gdist.reduce_timing(dist, 1.0 + rank, len(mine))
if rank == 0:
    print("RESULT " + json.dumps({"res": allres, "tmax": tmax, "usum": usum}))
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    import json
    got = json.loads(line[7:])
    assert got["tmax"] == 2.0 and got["usum"] == 5.0
    import oracle
    from gpd_amd import synth
    p = oracle.default_params(3)
    want = []
    for c in range(5):
        cl = synth.make_cloud(1234 + c, 3000)
        h = oracle.search(p, cl["xyz"], cl["normals"], synth.sample_indices(cl, 6))
        want.append([int(h["valid"].sum()), float(h["grasp_width"].sum())])
    assert got["res"] == want


def test_round_robin_assignment():
    from gpd_amd import dist as gdist
    assert gdist.clouds_of_rank(256, 3, 8) == list(range(3, 256, 8))
    allc = sorted(c for r in range(8) for c in gdist.clouds_of_rank(256, r, 8))
    assert allc == list(range(256))
    assert all(len(gdist.clouds_of_rank(256, r, 8)) == 32 for r in range(8))
