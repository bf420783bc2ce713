"""What a tile-mask ordering of conv1's chunks would skip, on the benchmark's own images — CPU only:
the images come from the oracle.  conv1 skips a (chunk of 64 pooled pixels, channel) pair when every byte of the chunk's
input windows is zero.  Shipped: chunks of 64 consecutive pixels in strip-major order (strips of 8, 8, 8, 4 columns).
Alternative: chunks of four 4x4-pixel tiles chosen per image by their 15-bit channel masks."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from gpd_amd import synth

C, N = 15, int(os.environ.get("N_IMAGES", "400"))
cloud = synth.make_cloud(1234, 30000)
si = synth.sample_indices(cloud, 2564)[:: max(1, 2564 * 2 // N)]  # samples spread over the benchmark's list
p = oracle.default_params(C)
h = oracle.filter_workspace(p, oracle.search(p, cloud["xyz"], cloud["normals"], si))
imgs, cand = oracle.images(p, cloud["xyz"], cloud["normals"], cloud["cam_source"], cloud["view_points"], h)
imgs = np.asarray(imgs).reshape(len(cand), 60, 60, C)[:N]
n = len(imgs)
nz = imgs != 0
win = np.zeros((n, 28, 28, C), bool)  # pooled pixel (py, px), channel c: any nonzero in rows 2py..2py+5, cols 2px..2px+5
for dy in range(6):
    for dx in range(6):
        win |= nz[:, dy:dy + 56:2, dx:dx + 56:2, :]
print("%d images of the benchmark's list; per-pixel live fraction %.3f; fully dead pixels %.3f" % (n, win.mean(), 1 - win.any(axis=3).mean()))


def chunks_cost(order_of_image, name):
    """order_of_image(i) -> the 784 pooled-pixel numbers of image i in chunk order; chunks straddle images as in the kernel"""
    live = total = 0
    carry = np.zeros((0, C), bool)
    for i in range(n):
        m = win[i].reshape(784, C)[order_of_image(i)]
        m = np.concatenate([carry, m])
        k = len(m) // 64
        blk = m[: 64 * k].reshape(k, 64, C).any(axis=1)
        live += blk.sum()
        total += k * C
        carry = m[64 * k:]
    print("%-58s live (chunk, channel) pairs %.3f" % (name, live / total))
    return live / total


strip = np.array([r * 28 + c for s, w in ((0, 8), (8, 8), (16, 8), (24, 4)) for r in range(28) for c in range(s, s + w)])
base = chunks_cost(lambda i: strip, "shipped: strip-major chunks of 64 consecutive pixels")

tiles = [(ty, tx) for ty in range(7) for tx in range(7)]  # 49 tiles of 4x4 pooled pixels
tile_pix = np.array([[(4 * ty + y) * 28 + 4 * tx + x for y in range(4) for x in range(4)] for ty, tx in tiles])  # [49][16]
dens = win.mean(axis=(0, 1, 2))
rank = np.argsort(-dens)
w_lo = np.zeros(C, np.int64); w_lo[rank] = 1 << np.arange(C)          # densest channel = least significant bit
w_hi = np.zeros(C, np.int64); w_hi[rank] = 1 << np.arange(C)[::-1]


def tile_masks(i):
    return win[i].reshape(784, C)[tile_pix].any(axis=1)  # [49][C]


chunks_cost(lambda i: tile_pix.reshape(-1), "tiles of 4x4, raster order")
chunks_cost(lambda i: tile_pix[np.argsort(tile_masks(i) @ w_lo, kind="stable")].reshape(-1), "tiles sorted by mask (densest channel = lsb)")
chunks_cost(lambda i: tile_pix[np.argsort(tile_masks(i) @ w_hi, kind="stable")].reshape(-1), "tiles sorted by mask (densest channel = msb)")
chunks_cost(lambda i: tile_pix[np.argsort(tile_masks(i).sum(axis=1), kind="stable")].reshape(-1), "tiles sorted by popcount")


def greedy(i):
    """start a chunk with the sparsest remaining tile, add the three tiles whose masks add the fewest channels"""
    m = tile_masks(i)
    left = list(np.argsort(m.sum(axis=1), kind="stable"))
    order = []
    while left:
        cur = [left.pop(0)]
        u = m[cur[0]].copy()
        while len(cur) < 4 and left:
            j = min(range(len(left)), key=lambda q: (m[left[q]] & ~u).sum())
            u |= m[left[j]]
            cur.append(left.pop(j))
        order += cur
    return tile_pix[order].reshape(-1)


chunks_cost(greedy, "tiles grouped greedily (fewest added channels)")
# the bound of any grouping of tiles: every tile alone
t = np.stack([tile_masks(i) for i in range(n)])
print("%-58s live (tile, channel) pairs     %.3f" % ("bound: every 4x4 tile its own chunk", t.mean()))
# 2x2-pixel tiles (an MFMA block of 16 lanes = four of them): 196 tiles per image, 16 per chunk
t2 = np.array([[(2 * ty + y) * 28 + 2 * tx + x for y in range(2) for x in range(2)] for ty in range(14) for tx in range(14)])  # [196][4]
chunks_cost(lambda i: t2[np.argsort(win[i].reshape(784, C)[t2].any(axis=1) @ w_lo, kind="stable")].reshape(-1), "tiles of 2x2 sorted by mask")
# rows of 4 pixels (what conv2's staging moves with one 16-byte LDS store): 196 per image
r4 = np.array([[py * 28 + 4 * q + x for x in range(4)] for py in range(28) for q in range(7)])
chunks_cost(lambda i: r4[np.argsort(win[i].reshape(784, C)[r4].any(axis=1) @ w_lo, kind="stable")].reshape(-1), "row segments of 4 pixels sorted by mask")
pm = win.reshape(n, 784, C)
chunks_cost(lambda i: np.argsort(pm[i] @ w_lo, kind="stable"), "for comparison: single PIXELS sorted by mask")
