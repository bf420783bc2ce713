#!/bin/bash
# Multi-GPU readiness on the round's kernels, on a ONE-GPU box — not a scaling measurement:
#  1. the driver's N = 8 command line with all eight ranks on device 0 (gloo carries the reductions: RCCL refuses several ranks on
#     one device): launch, rendezvous, NUMA binding, whole-job sums; the ranks' candidate counts against eight single-rank runs;
#  2. gpd_hip_detect_sharded over EIGHT contexts on the 300k-point clutter cloud against the single call, byte for byte;
#  3. what a context costs: pinned host memory and device memory of one rank's context after the batch leg.
#   profiles/r06_sim8.sh   ->  gpurun_out/r06sim8/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06sim8
mkdir -p $OUT
cd $ROOT
G="python profiles/memguard.py --rss-gb 48 --seconds 900"
HSA_ENABLE_IPC_MODE_LEGACY=0 $G -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 \
  bench.py --gpus 8 --steps 5 --warmup 2 --devices 0 --dist-backend gloo > $OUT/bench8.json 2> $OUT/bench8.err
echo "rc=$?" > $OUT/summary.txt; grep memguard $OUT/bench8.err >> $OUT/summary.txt
python - >> $OUT/summary.txt 2>&1 <<P
import json, sys, os, time
sys.path.insert(0, "$ROOT")
import numpy as np
d = json.loads([l for l in open("$OUT/bench8.json") if l.startswith("{")][-1])
b = d.get("batch_end_to_end") or {}
print("1. bench.py --gpus 8, eight ranks on device 0 (gloo): n_gpus %d  value %.0f cand/s  ms/step %.3f  scaling %s" % (d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"]))
print("   batch leg: clouds %s  candidates %s  cand/s %.0f  rank spread %s  buffer growths in timed passes %s" % (b.get("clouds"), b.get("candidates"), b.get("cand_per_s", 0), b.get("rank_clouds_per_s"), (b.get("passes") or {}).get("buffer_growths_in_timed_passes")))
print("   host binding of rank 0:", d.get("host_binding_rank0"))
from gpd_amd import api, synth
import bench
w = synth.lenet_weights(15, real=dict(np.load(os.path.join("$ROOT", "tests", "golden", "lenet15_params.npz"))), trained_magnitude=True)
# the headline's per-rank candidate counts, one rank at a time: cloud seed 1234 + rank, first 5000 valid candidates
tot = 0
ctx = api.Context(api.default_params(15)); ctx.set_lenet_weights(w)
for r in range(8):
    cl = synth.make_cloud(1234 + r, 30000)
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    si = synth.sample_indices(cl, min(2564, int(cl["is_object"].sum())))
    h = ctx.search(si); bench._filter_workspace(h, ctx.params)
    tot += min(int(h["valid"].sum()), 5000)
print("   candidates per step: 8-rank line %d, eight single-rank runs %d -> %s" % (d["config"]["candidates_per_gpu"] * 8 if False else round(d["value"] * d["ms_per_step"] / 1e3), tot, "EQUAL" if round(d["value"] * d["ms_per_step"] / 1e3) == tot else "DIFFERENT"))
# 2. one 300k-point cloud over eight contexts
cl = synth.make_cloud(1234, 300000, clutter=True)
si = synth.sample_indices(cl, 6000)
ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
t0 = time.perf_counter(); want, n_cand = ctx.detect(si); t1 = time.perf_counter() - t0
flat = want.reshape(-1); want = flat[flat["valid"].astype(bool)]
others = []
for g in range(7):
    c = api.Context(api.default_params(15)); c.set_lenet_weights(w); others.append(c)
got, info = ctx.detect_sharded(others, cl, si)
t0 = time.perf_counter(); got, info = ctx.detect_sharded(others, cl, si); t2 = time.perf_counter() - t0
print("2. gpd_hip_detect_sharded, 300k-point clutter cloud, %d samples over 8 contexts on one device: %d candidates, records %s the single call's (%d candidates); shard candidates %s; draws %s" % (len(si), len(got), "BYTE FOR BYTE" if got.tobytes() == want.tobytes() else "DIFFER FROM", n_cand, [i[1] for i in info], [i[3] for i in info]))
print("   wall: single call %.1f ms, eight shards on the one device %.1f ms" % (t1 * 1e3, t2 * 1e3))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def mem():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value, t.value
free, total = mem()
print("3. device memory in use with 8 contexts after the 300k cloud: %.1f GB of %.1f GB" % ((total - free) / 2**30, total / 2**30))
for c in others: c.close()
free2, _ = mem()
print("   one context after a 300k-point cloud + the 5000-candidate lists: %.2f GB of device memory (7 closed: %.1f GB returned)" % ((free2 - free) / 7 / 2**30, (free2 - free) / 2**30))
ctx.close()
P
cat $OUT/summary.txt
