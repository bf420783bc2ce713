"""Stress of the LeNet path (persistent conv1 ring, fc1 tile shapes): random batch sizes and sparsities, every score
against the same image scored in a small batch.  profiles/stress_lenet.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpd_amd import api, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(123)
for C in (15, 3, 12, 1):
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lenet%d_params.npz" % C)
    w = synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None)
    ctx = api.Context(api.default_params(C))
    ctx.set_lenet_weights(w)
    base = np.zeros((512, 60, 60, C), np.uint8)
    for i in range(512):
        dens = rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])
        base[i] = rng.randint(0, 256, (60, 60, C)) * (rng.rand(60, 60, C) < dens)
        if rng.rand() < 0.3:
            base[i, :, :, rng.randint(C):] = 0      # whole channels empty
    ref = np.concatenate([ctx.score(base[i:i + 64]) for i in range(0, 512, 64)])
    t0 = time.time()
    for it in range(iters if C == 15 else iters // 3):
        n = int(rng.choice([rng.randint(1, 40), rng.randint(40, 700), rng.randint(700, 6000)]))
        idx = rng.randint(0, 512, n)
        if rng.rand() < 0.3:
            idx = np.sort(idx)                      # runs of equal / similar images
        got = ctx.score(base[idx])
        assert np.array_equal(got, ref[idx]), (C, it, n)
    print("C=%d ok (%.1f s)" % (C, time.time() - t0))
    ctx.close()
