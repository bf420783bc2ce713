#!/bin/bash
# Quick GPU-box pass while iterating on a kernel: parity tests of the touched stage + a kernel-trace of the default bench.
#   profiles/gpu_quick.sh <tag> "<pytest -k expression or test files>"
set -u
TAG=${1:-q}
SEL=${2:-tests/test_gpu_parity.py tests/test_gpu_resident.py}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest $SEL -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-samples 0 --batch-clouds 0 > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.0f cand/s  ms/step %.3f" % (d["value"], d["ms_per_step"]))
for k, v in d["kernels"].items():
    print("  %-22s %.3f ms" % (k, v["ms"]))
print("  detect", d["detect_end_to_end"]["wall_ms"], d["detect_end_to_end"]["kernel_ms"])
PY
python profiles/timeline.py $(find $OUT/trace -name "*.db" | head -1) --last-seconds 0.045 | tail -16
