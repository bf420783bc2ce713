"""Runs bench.main() with gpd_amd.api.Context replaced by the oracle-backed stand-in (tests/fake_context.py): test infrastructure,
started by tests/test_bench_dryrun.py as `python tests/bench_dryrun_harness.py <bench.py arguments>`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

torch.cuda.synchronize = lambda *a, **k: None  # no device here; bench.py brackets its timed regions with it
torch.cuda.set_device = lambda *a, **k: None

from gpd_amd import api  # noqa: E402
import fake_context  # noqa: E402

api.Context = fake_context.FakeContext
_proxy = fake_context.FakeLib(api.lib())
api.lib = lambda: _proxy

import bench  # noqa: E402

if os.environ.get("GPD_DRYRUN_ACCURACY_MAX"):  # the leg's bound at a size the CPU oracle reaches
    bench.ACCURACY_LEG_MAX = int(os.environ["GPD_DRYRUN_ACCURACY_MAX"])
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
