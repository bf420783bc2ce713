"""BASELINE.json configs[4] on ONE rank: the clouds a rank of an 8-GPU job owns (cloud i -> rank i mod 8,
32 of the 256 clouds), each detected end to end through gpd_hip_detect (upload, search, filter, images,
LeNet, scored hands back on the host).  Reports clouds/s and candidates/s with one context and with
two contexts in flight on two host threads (the second hides the host hops of the first).

usage: python profiles/batch_config5.py [--clouds 32] [--world 8] [--rank 0]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpd_amd import api, dist, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=256)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--limit", type=int, default=32, help="at most this many of the rank's clouds")
    args = ap.parse_args()
    ids = dist.clouds_of_rank(args.clouds, args.rank, args.world)[: args.limit]
    w = synth.lenet_weights(15, real=dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lenet15_params.npz"))))
    clouds = []
    for i in ids:
        cl = synth.make_cloud(1234 + i, 30000)
        clouds.append((cl, synth.sample_indices(cl, 2564)))

    def run(ctx, items, out):
        for cl, si in items:
            ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
            hands, n = ctx.detect(si)
            out.append(n)

    for n_ctx in (1, 2):
        ctxs = [api.Context(api.default_params(15)) for _ in range(n_ctx)]
        for c in ctxs:
            c.set_lenet_weights(w)
            run(c, clouds[:1], [])  # warm-up: allocations
        outs = [[] for _ in range(n_ctx)]
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(ctxs[k], clouds[k::n_ctx], outs[k])) for k in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        total = sum(sum(o) for o in outs)
        print("contexts in flight %d: %d clouds, %d candidates in %.3f s -> %.1f clouds/s, %.0f candidates/s end to end"
              % (n_ctx, len(clouds), total, dt, len(clouds) / dt, total / dt))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
