"""Stress of the persistent conv1 (two-slot image ring in LDS, slots released and refilled by whichever wave finishes a
chunk last — lenet.hip) and of the ip1 tile shapes behind it: random batch sizes from 1 to 20 000, random sparsities
including all-zero and all-255 images, a few hundred launches back to back, two contexts at once from two host threads —
every score against the same image scored in a small batch.  Plus the watchdog: a slot that is never published must
surface as GPD_ERR_HIP and leave the context usable (it used to be a __builtin_trap(), which lost the HIP context)."""
import os
import threading

import numpy as np
import pytest

from gpd_amd import api, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _weights(C):
    g = os.path.join(GOLD, "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


def _pool(rng, C, n=512):
    base = np.zeros((n, 60, 60, C), np.uint8)
    for i in range(n):
        dens = rng.choice([0.0, 0.0, 0.01, 0.05, 0.3, 1.0])
        base[i] = rng.randint(0, 256, (60, 60, C)) * (rng.rand(60, 60, C) < dens)
        if rng.rand() < 0.3:
            base[i, :, :, rng.randint(C):] = 0      # whole channels empty
        if rng.rand() < 0.05:
            base[i] = 255                            # saturated image: nothing to skip
    return base


def _run(ctx, base, ref, rng, launches, sizes):
    for it in range(launches):
        lo, hi = sizes[rng.randint(len(sizes))]
        n = int(rng.randint(lo, hi))
        idx = rng.randint(0, len(base), n)
        if rng.rand() < 0.3:
            idx = np.sort(idx)                      # runs of equal images
        got = ctx.score(base[idx])
        assert np.array_equal(got, ref[idx]), (it, n)


@pytest.mark.parametrize("C", [15, 12, 3, 1])
def test_random_batches_back_to_back(C):
    rng = np.random.RandomState(100 + C)
    ctx = api.Context(api.default_params(C))
    try:
        ctx.set_lenet_weights(_weights(C))
        base = _pool(rng, C)
        ref = np.concatenate([ctx.score(base[i:i + 64]) for i in range(0, len(base), 64)])
        assert np.isfinite(ref).all()
        sizes = [(1, 40), (40, 700), (700, 6000)]
        _run(ctx, base, ref, rng, 200 if C == 15 else 60, sizes)
        if C == 15:
            _run(ctx, base, ref, rng, 3, [(12000, 20001)])
    finally:
        ctx.close()


def test_two_contexts_from_two_threads():
    """Two contexts on the one device, each driven by its own host thread: the persistent workgroups of two conv1
    launches share the CUs, and every score still equals the small-batch score."""
    C = 15
    errors = []

    def worker(seed):
        try:
            rng = np.random.RandomState(seed)
            ctx = api.Context(api.default_params(C))
            try:
                ctx.set_lenet_weights(_weights(C))
                base = _pool(rng, C, 256)
                ref = np.concatenate([ctx.score(base[i:i + 64]) for i in range(0, len(base), 64)])
                _run(ctx, base, ref, rng, 60, [(1, 40), (40, 700), (700, 4000)])
            finally:
                ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(s,)) for s in (7, 8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


_WATCHDOG_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from gpd_amd import api, synth
assert api.LIB_PATH.endswith("libgpd_hip_prof.so"), api.LIB_PATH
C = 15
g = os.path.join(sys.argv[1], "tests", "golden", "lenet15_params.npz")
w = synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None)
rng = np.random.RandomState(5)
ctx = api.Context(api.default_params(C))
ctx.set_lenet_weights(w)
ctx.set_lenet_mode(api.LENET_F32_CHAIN)  # the slot protocol (and its watchdog) is the f32-chain conv1's; the split path's conv1 has one barrier per image
base = np.maximum(rng.randint(0, 256, (128, 60, 60, C)) * (rng.rand(128, 60, 60, C) < 0.3), 1).astype(np.uint8)  # no empty images
good = ctx.score(base[:64])
os.environ["GPD_C1_FAULT"] = "1"
try:
    ctx.score(np.concatenate([base] * 8))  # 1024 images: four per workgroup, so slots must be refilled
    print("NO ERROR")
    sys.exit(2)
except api.GpdHipError as e:
    assert "slot" in str(e), str(e)
del os.environ["GPD_C1_FAULT"]
again = ctx.score(base[:64])
assert np.array_equal(again, good)
ctx.close()
print("WATCHDOG OK")
"""


def test_slot_watchdog_reports_an_error_and_keeps_the_context():
    """The fault injector lives in the profiling build only (libgpd_hip_prof.so, -DGPD_PROFILING; the release library
    reads no environment variable), so this test runs in a process of its own with that build: GPD_C1_FAULT=1 makes
    workgroup 0 never publish the slots it refills.  Its waves time out (~0.5 s), raise the launch's error word and
    leave; the call returns GPD_ERR_HIP with a text — and the next call on the same context is correct."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "gpd_amd", "libgpd_hip_prof.so")
    assert os.path.exists(prof), "libgpd_hip_prof.so is not built (make -C gpd_amd/csrc prof)"
    env = dict(os.environ, GPD_HIP_LIB=prof)
    env.pop("GPD_C1_FAULT", None)
    r = subprocess.run([sys.executable, "-c", _WATCHDOG_SCRIPT, root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WATCHDOG OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
