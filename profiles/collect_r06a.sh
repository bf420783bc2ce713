#!/bin/bash
# Round 6, collection leg A (one gpurun call; nothing here has ever taken a box down): the GPU suite, the default bench line, the
# kernel trace of the default line, the SQ counter pass (MfmaUtil, LDS, waits, clock) and the VALU lane-occupancy pass of the image
# kernels.  Every command runs under profiles/memguard.py (host RSS + wall-clock limits).
#   profiles/collect_r06a.sh <tag> [skip-tests]   ->  gpurun_out/<tag>/
set -u
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
G="python $ROOT/profiles/memguard.py --rss-gb 24"
mkdir -p $OUT
cd $ROOT
if [ "${2:-}" != "skip-tests" ]; then
  $G --seconds 900 -- python -m pytest tests -m gpu -x -q > $OUT/gputests.txt 2>&1
  tail -3 $OUT/gputests.txt
fi
$G --seconds 400 -- python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof
mkdir -p $P
$G --seconds 300 -- rocprofv3 --kernel-trace --stats -d $P/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/stats.log 2>&1
cd $ROOT
python profiles/summarize.py $P > $OUT/rocprof_summary.txt 2>&1
rocprofv3 -L > $OUT/counters_available.txt 2>&1
bash $ROOT/profiles/pmc_sq.sh $TAG > $OUT/pmc_sq.log 2>&1
cp $ROOT/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_sq.txt 2>/dev/null
cp $ROOT/gpurun_out/pmc_$TAG/summary.json $OUT/pmc_sq.json 2>/dev/null
bash $ROOT/profiles/pmc_lanes.sh $TAG > $OUT/pmc_lanes.log 2>&1
cp $ROOT/gpurun_out/lanes_$TAG/summary.txt $OUT/pmc_lanes.txt 2>/dev/null
python - "$OUT/bench_default.json" <<PY
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.0f  ms/step %.3f  roofline %s %.3f" % (d["value"], d["ms_per_step"], r.get("kernel", "")[:24], r.get("frac") or 0))
print({k: round(v["ms"], 4) for k, v in d["kernels"].items() if "ms" in v})
PY
tail -25 $OUT/rocprof_summary.txt | cut -c1-150
cat $OUT/pmc_sq.txt | cut -c1-220
cat $OUT/pmc_lanes.txt | cut -c1-220
