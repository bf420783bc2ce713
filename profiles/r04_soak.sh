#!/bin/bash
# a one-off soak of the differential fuzz tests (HIP path against the oracle) over many more seeds than the suite runs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04soak
mkdir -p $OUT
cd $ROOT
GPD_FUZZ_DETECT=${1:-96} GPD_FUZZ_WIDE=${2:-24} GPD_FUZZ_GEOMETRY=${3:-40} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1
echo "soak rc=$?"; tail -5 $OUT/pytest.log
