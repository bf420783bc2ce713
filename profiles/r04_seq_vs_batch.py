"""Is the two-lane batch worth it?  The same 24 clouds through gpd_hip_detect_batch in one call (two clouds in flight) and
one call per cloud (one lane, the host waits for every cloud), alternating, four passes each.  Run on the GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpd_amd import api, synth

C = 15
real = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lenet15_params.npz")))
ctx = api.Context(api.default_params(C))
ctx.set_lenet_weights(synth.lenet_weights(C, real=real))
n = int(os.environ.get("CLOUDS", "24"))
clouds = [synth.make_cloud(1234 + i, 30000) for i in range(n)]
samples = [synth.sample_indices(cl, 2564) for cl in clouds]
whole = ctx.batch(clouds, samples, 0)
singles = [ctx.batch([cl], [si], 0) for cl, si in zip(clouds, samples)]
pairs = [ctx.batch(clouds[i:i + 2], samples[i:i + 2], 0) for i in range(0, n, 2)]

def run_whole():
    return sum(nc for _, _, nc, _ in ctx.run_batch(whole))
def run_singles():
    return sum(ctx.run_batch(b)[0][2] for b in singles)
def run_pairs():
    return sum(nc for b in pairs for _, _, nc, _ in ctx.run_batch(b))

for f in (run_whole, run_singles, run_pairs):
    f()
res = {"whole": [], "singles": [], "pairs": []}
for _ in range(4):
    for name, f in (("whole", run_whole), ("singles", run_singles), ("pairs", run_pairs)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        nc = f()
        dt = time.perf_counter() - t
        res[name].append(nc / dt)
for k, v in res.items():
    print("%-8s cand/s: %s  median %.0f" % (k, " ".join("%.0f" % x for x in v), sorted(v)[len(v) // 2]))
