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


SHARD_WORKER = r'''
import os, sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from gpd_amd import synth, dist as gdist
import oracle
dist.init_process_group("gloo")
rank, world = gdist.rank_world()
# ONE cloud, its samples cut into contiguous ranges, one per rank (gpd_hip_detect_sharded's scheme, restated with the oracle):
cl = synth.make_cloud(1234, 6000)
si = synth.sample_indices(cl, 60)
lo, hi = len(si) * rank // world, len(si) * (rank + 1) // world
p = oracle.default_params(15)
hands = oracle.filter_workspace(p, oracle.search(p, cl["xyz"], cl["normals"], si[lo:hi]))
# phase 1: the range's draw total (the plan summary's total_draws on the device); host-side exclusive scan over the ranks
oracle.set_lcg_base(0)
oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands, want_images=False)
mine = torch.tensor([oracle.last_lcg_draws()], dtype=torch.int64)
allv = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allv, mine)
base = int(sum(int(v[0]) for v in allv[:rank]))
# phase 2: the images with the stream continued
oracle.set_lcg_base(base)
img, cand = oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
digest = hashlib.sha1(img.tobytes()).hexdigest()
out = [None] * world
dist.all_gather_object(out, dict(rank=rank, n=int(len(img)), base=base, draws=int(mine[0]), digest=digest, path=None))
np.save(os.path.join(sys.argv[1], "img_%%d.npy" %% rank), img)
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_one_cloud_sharded_by_sample_range(tmp_path):
    """The scheme of gpd_hip_detect_sharded on the CPU: two ranks, one cloud, contiguous sample ranges, the ranges' shadow-draw
    totals exchanged (two numbers) and scanned on the host, every range continuing the cloud's ONE LCG stream where the ranges
    before it stopped (hand_set.cpp:268-283) — the concatenated images are byte for byte the single process's; without the
    scan (every range from draw 0) they are not."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script), str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    import oracle
    from gpd_amd import synth
    cl = synth.make_cloud(1234, 6000)
    si = synth.sample_indices(cl, 60)
    p = oracle.default_params(15)
    oracle.set_lcg_base(0)
    hands = oracle.filter_workspace(p, oracle.search(p, cl["xyz"], cl["normals"], si))
    want, _ = oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    total = oracle.last_lcg_draws()
    assert got[0]["base"] == 0 and got[1]["base"] == got[0]["draws"] and got[0]["draws"] + got[1]["draws"] == total
    parts = [np.load(tmp_path / ("img_%d.npy" % r)) for r in range(2)]
    assert sum(len(x) for x in parts) == len(want) and len(parts[1]) > 20
    assert np.array_equal(np.concatenate(parts), want)
    # the second range from draw 0: its shadow channels differ
    oracle.set_lcg_base(0)
    h2 = oracle.filter_workspace(p, oracle.search(p, cl["xyz"], cl["normals"], si[len(si) // 2:]))
    alone, _ = oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], h2)
    assert not np.array_equal(alone, parts[1])
