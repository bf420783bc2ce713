#!/bin/bash
# One GPU-box pass: the -m gpu suite, the default bench line, the batch line (run through gpurun).
#   profiles/gpu_check.sh <tag> [pytest-args]
set -u
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
grep detect-timing $OUT/bench.err | tail -2
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("value %.0f cand/s  ms/step %.3f" % (d["value"], d["ms_per_step"]))
    for k, v in d["kernels"].items():
        print("  %-22s %.3f ms" % (k, v["ms"]))
    print("  detect", d["detect_end_to_end"])
    print("  batch", d.get("batch_end_to_end"))
    print("  roofline", {k: v for k, v in d["roofline"].items() if k != "note"})
    print("  cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
PY
