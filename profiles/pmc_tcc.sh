#!/bin/bash
# The two TCC counter passes (HBM traffic per kernel) of the default bench line — their own script since round 5: a collection
# pass that contained them lost its GPU node twice in a row on 2026-09-25.  Run only when a node can be risked.
#   profiles/pmc_tcc.sh <tag>  ->  gpurun_out/<tag>/prof/pmc_{fetch,write}, then  python profiles/summarize.py <dir> --traffic <json>
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=$ROOT/gpurun_out/$TAG/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/pmc_write.log 2>&1
cd $ROOT
python profiles/summarize.py $P --traffic $ROOT/gpurun_out/$TAG/traffic.json
