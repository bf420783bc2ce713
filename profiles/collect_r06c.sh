#!/bin/bash
# Round 6, collection leg C (its own gpurun call, the riskiest one: round 5 suspected these passes of taking nodes down, wrongly
# as it turned out — the culprit was a host-memory bug of bench.py): the two TCC counter passes of the default line, FETCH_SIZE and
# WRITE_SIZE, each its own rocprofv3 run with --pmc only (MI355X_MICROARCH.md: they do not fit one pass; reads x2 on gfx950).
#   profiles/collect_r06c.sh <tag>   ->  gpurun_out/<tag>/{traffic.json,pmc_summary.txt}
set -u
TAG=${1:-r06c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=$ROOT/gpurun_out/$TAG/prof
G="python $ROOT/profiles/memguard.py --rss-gb 24 --seconds 300"
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
$G -- rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/pmc_fetch.log 2>&1
$G -- rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/pmc_write.log 2>&1
cd $ROOT
python profiles/summarize.py $P > $ROOT/gpurun_out/$TAG/pmc_summary.txt 2>&1
python profiles/summarize.py $P --traffic $ROOT/gpurun_out/$TAG/traffic.json
python - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/$TAG/traffic.json"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:16]:
    print("%-60s read %8.1f MB  write %8.1f MB" % (k[:60], v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6))
PY
