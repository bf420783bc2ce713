#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   profiles/run_profile.sh <tag>   ->  gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}
# Kernel-trace/stats and each PMC counter set are separate runs (PMC is never combined
# with sys/runtime tracing).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-samples 0 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -50
