#!/usr/bin/env python
"""shadow_set_kernel's LDS atomics, modelled on the CPU: are its bank conflicts (66 % of its LDS cycles, profiles/r04_pmc_sq.txt)
lanes hitting the SAME bitset word — which a merge inside the wave before the atomic would remove — or
lanes hitting DIFFERENT words that share a bank?

The kernel (gpd_amd/csrc/images.hip, shadow_set_kernel<0>): one workgroup of 1024 threads per hand set; thread t takes the
neighbours t, t + 1024, ... of the set's 0.10 m neighbourhood (sorted by distance from the sample) and walks each one's 33 shadow
draws p + t_k * vec (t_k from the reference's LCG, vec = 0.10 m along camera -> centroid, hand_set.cpp:147-233); a draw inside the
86^3-voxel region of the set becomes one `ds_or_b32` on word (vx * 86 + vy) * 86 + vz >> 5.  So at every step k the 64 lanes of a
wave issue up to 64 atomics for 64 DIFFERENT rays.  This script replays exactly that address stream for hand sets of the
benchmark's cloud (the oracle's neighbour lists and draws) and counts per wave instruction: active lanes, distinct words (what a
merge could save), and the largest number of distinct words that fall into one of the 64 banks (the cycles the instruction takes).

    python profiles/shadow_set_bank_model.py > profiles/r05_shadow_set_bank_model.txt          (CPU only)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SD, SR, VOX, NSH, THREADS = 86, 43, 0.003, 33, 1024


def main():
    import oracle
    from gpd_amd import synth
    from scipy.spatial import cKDTree
    cloud = synth.make_cloud(1234, 30000)
    xyz = cloud["xyz"].astype(np.float64)
    vp = np.asarray(cloud["view_points"], np.float64).reshape(-1, 3)[0]
    si = synth.sample_indices(cloud, 48)
    tree = cKDTree(xyz)
    draws = oracle.fastrand(33 * 8192 * 48 + 64).astype(np.float64) / 32767.0  # the reference's stream from its start
    pos = 0
    tot = dict(instr=0, active=0, words=0, cycles=0, cycles_merged=0, dup=0)
    rows = []
    for s in si:
        c = xyz[s]
        idx = np.array(tree.query_ball_point(c, 0.10), np.int64)
        d2 = ((xyz[idx].astype(np.float32) - c.astype(np.float32)) ** 2).sum(1)
        idx = idx[np.lexsort((idx, d2))]            # the list's order: (d^2, index)
        P = xyz[idx]
        N = len(P)
        vec = P.mean(0) - vp
        vec = 0.10 * vec / np.linalg.norm(vec)
        org = np.floor(c / VOX).astype(np.int64) - SR
        t = draws[pos:pos + NSH * N].reshape(N, NSH)   # draw k of neighbour i sits at offset i * 33 + k of the set's stretch
        pos += NSH * N
        q = P[:, None, :] + t[:, :, None] * vec[None, None, :]
        v = (q / VOX).astype(np.int64) - org[None, None, :]   # (int) truncation: the coordinates are positive here
        inside = np.all((v >= 0) & (v < SD), axis=2)
        word = ((v[:, :, 0] * SD + v[:, :, 1]) * SD + v[:, :, 2]) >> 5
        st = dict(instr=0, active=0, words=0, cycles=0, cycles_merged=0)
        for r0 in range(0, N, THREADS):                      # the round: neighbours r0 + tid
            for w0 in range(r0, min(r0 + THREADS, N), 64):   # a wave's 64 neighbours
                sl = slice(w0, min(w0 + 64, N))
                for k in range(NSH):
                    a = inside[sl, k]
                    n = int(a.sum())
                    if n == 0:
                        continue
                    ws = word[sl, k][a]
                    uniq = np.unique(ws)
                    # an LDS atomic is a read-modify-write per lane: lanes on one bank take turns, same word or not;
                    # after a merge only the distinct words would be left
                    cyc = np.bincount(ws % 64, minlength=64).max()
                    cyc_m = np.bincount(uniq % 64, minlength=64).max()
                    st["instr"] += 1
                    st["active"] += n
                    st["words"] += len(uniq)
                    st["cycles"] += int(cyc)
                    st["cycles_merged"] += int(cyc_m)
        rows.append((int(s), N, st))
        for k in st:
            tot[k] += st[k]
    print("# shadow_set_kernel<0>: the wave-level address stream of its ds_or_b32, replayed on the CPU (profiles/shadow_set_bank_model.py)")
    print("# cloud: the benchmark's (seed 1234, 30k points), %d hand sets; per set: neighbours, wave instructions with an active lane," % len(si))
    print("# mean active lanes, share of active lanes whose word another lane of the instruction also hits, cycles per instruction")
    print("# (largest number of lanes on one of the 64 banks) as issued and if equal words were merged first\n")
    print("%8s %6s %7s %8s %10s %12s %14s" % ("sample", "N_i", "instr", "active", "same word", "cycles/instr", "after merging"))
    for s, N, st in rows[:12]:
        print("%8d %6d %7d %8.1f %9.1f%% %12.2f %14.2f" % (s, N, st["instr"], st["active"] / st["instr"], 100.0 * (1 - st["words"] / st["active"]),
                                                           st["cycles"] / st["instr"], st["cycles_merged"] / st["instr"]))
    print("   ... (%d sets)" % len(rows))
    a, i = tot["active"], tot["instr"]
    print("\nall sets: %.1f active lanes per instruction; %.2f %% of the active lanes share their word with another lane;"
          % (a / i, 100.0 * (1 - tot["words"] / a)))
    print("          %.2f cycles per instruction as issued -> conflict share 1 - 1 / %.2f = %.0f %% (measured: 66 %% of LDS cycles, r04_pmc_sq.txt);"
          % (tot["cycles"] / i, tot["cycles"] / i, 100.0 * (1 - i / tot["cycles"])))
    print("          %.2f cycles per instruction with equal words merged: %.1f %% fewer"
          % (tot["cycles_merged"] / i, 100.0 * (1 - tot["cycles_merged"] / tot["cycles"])))
    rng = np.random.RandomState(0)
    lanes = int(round(a / i))
    mx = np.mean([np.bincount(rng.randint(0, 64, lanes), minlength=64).max() for _ in range(20000)])
    print("          %d lanes on uniformly random banks: %.2f cycles per instruction — the stream behaves like unrelated addresses" % (lanes, mx))


if __name__ == "__main__":
    main()
