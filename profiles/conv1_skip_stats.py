"""Zero-skip statistics of conv1 on the benchmark images (run on the GPU box): live (chunk, channel)
pairs per image for different numberings of the 28x28 pooled pixels into chunks of 64 lanes —
the measurement behind the strip-major order of conv1_mfma_kernel."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpd_amd import api, synth
import bench
C = 15
cloud = synth.make_cloud(1234, 30000)
ctx = api.Context(api.default_params(C), device=0)
ctx.upload_cloud(cloud["xyz"], cloud["normals"], cloud["cam_source"], cloud["view_points"])
si = synth.sample_indices(cloud, 2564)
hands = ctx.search(si)
hf = hands.copy(); bench._filter_workspace(hf, ctx.params)
flat = hf.reshape(-1); vidx = np.flatnonzero(flat["valid"]); flat["valid"][vidx[5000:]] = 0
imgs, cand = ctx.images(hf, download=True)
imgs = np.asarray(imgs).reshape(len(cand), 60, 60, C)[::3]
nz = imgs != 0
n = len(imgs)
# winnz[n, py, px, c]: any nonzero in rows 2py..2py+5, cols 2px..2px+5
win = np.zeros((n, 28, 28, C), bool)
for dy in range(6):
    for dx in range(6):
        win |= nz[:, dy:dy + 56:2, dx:dx + 56:2, :]
print("per-pixel live fraction %.3f" % win.mean())
def cost(order, name):
    order = np.asarray(order)
    tot = 0; nch = 0
    for s in range(0, len(order), 64):
        ps = order[s:s + 64]
        live = win[:, ps // 28, ps % 28, :].any(axis=1)
        tot += live.sum(); nch += 1
    print("%-22s chunks %2d  live-chunk-equiv %.2f" % (name, nch, tot / n / C))
cost(np.arange(784), "row-major")
for W in (2, 4, 7, 14):
    o = [r * 28 + c for s in range(0, 28, W) for r in range(28) for c in range(s, s + W)]
    cost(o, "vstrip W=%d" % W)
for H in (2, 4, 7, 14):
    o = [r * 28 + c for s in range(0, 28, H) for c in range(28) for r in range(s, s + H)]
    cost(o, "hstrip H=%d" % H)
def morton(y, x):
    z = 0
    for b in range(5):
        z |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return z
o = sorted(range(784), key=lambda p: morton(p // 28, p % 28))
cost(o, "morton")
o = sorted(range(784), key=lambda p: morton(p % 28, p // 28))
cost(o, "morton-T")
# channel-wise: which channels are live most
print("live per channel (vstrip7):", np.round(win.any(axis=(1,2)).mean(axis=0), 2))
print("per-pixel live per channel:", np.round(win.mean(axis=(0,1,2)), 2))
# ---- data-dependent orderings
import itertools
def cost_dyn(keyfn, name):
    tot = 0
    for i in range(n):
        m = win[i].reshape(784, C)
        order = keyfn(m)
        for s in range(0, 784, 64):
            tot += m[order[s:s + 64]].any(axis=0).sum()
    print("%-28s live-chunk-equiv %.2f" % (name, tot / n / C))
dens = win.mean(axis=(0, 1, 2))
rank = np.argsort(-dens)           # most dense channel first
wts_hi = np.zeros(C, np.int64); wts_hi[rank] = 1 << np.arange(C)[::-1]    # densest channel = most significant bit
wts_lo = np.zeros(C, np.int64); wts_lo[rank] = 1 << np.arange(C)          # densest = least significant
n = 300; 
cost_dyn(lambda m: np.argsort(m @ wts_hi, kind="stable"), "sort mask (dense=msb)")
cost_dyn(lambda m: np.argsort(m @ wts_lo, kind="stable"), "sort mask (dense=lsb)")
cost_dyn(lambda m: np.argsort(m.sum(axis=1), kind="stable"), "sort popcount")
vs = np.array([r * 28 + c for s in range(0, 28, 7) for r in range(28) for c in range(s, s + 7)])
def dead_removed(m):
    o = vs[m[vs].any(axis=1)]
    d = vs[~m[vs].any(axis=1)]
    return np.concatenate([o, d])
cost_dyn(dead_removed, "vstrip7, dead pixels last")
print("fully dead pixel fraction %.3f" % (1 - win.any(axis=3).mean()))
