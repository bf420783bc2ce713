#!/bin/bash
# Round 6, collection leg B (one gpurun call): the large configurations on the current kernels — config 4 (300k points / 50 000
# candidates), configs[4] on one GPU (batch of 64 clouds), the side configs 3a / 3b, and the kernel trace of a batch run for the
# timeline.  Round 5 lost its boxes to an unbounded host-side leg of bench.py on config 4: every command here runs under
# profiles/memguard.py (24 GB of resident host memory, a wall-clock limit), config 4 last.
#   profiles/collect_r06b.sh <tag>   ->  gpurun_out/<tag>/
set -u
TAG=${1:-r06b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
G="python $ROOT/profiles/memguard.py --rss-gb 24"
mkdir -p $OUT
cd $ROOT
$G --seconds 300 -- python bench.py --config 3a --cpu-samples 0 > $OUT/bench_config3a.json 2> $OUT/bench_config3a.err
$G --seconds 300 -- python bench.py --config 3b --cpu-samples 0 > $OUT/bench_config3b.json 2> $OUT/bench_config3b.err
$G --seconds 400 -- python bench.py --mode batch --clouds 64 --steps 3 --warmup 1 > $OUT/bench_batch64.json 2> $OUT/bench_batch64.err
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof
mkdir -p $P
$G --seconds 300 -- rocprofv3 --kernel-trace --stats -d $P/batch -o batch -- python $ROOT/bench.py --mode batch --clouds 16 --steps 2 --warmup 1 > $P/batch.log 2>&1
cd $ROOT
python profiles/timeline.py $(find $P/batch -name "*.db" | head -1) --last-seconds 0.25 > $OUT/batch_timeline.txt 2>&1
$G --seconds 600 -- python bench.py --config 4 --steps 5 --warmup 1 --cpu-samples 600 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
for f in $OUT/bench_*.json; do python - "$f" <<PY
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1].split("/")[-1], "no line:", e); raise SystemExit
r = d.get("roofline", {})
print("%-28s value %10.0f  ms/step %8.3f  roofline %s %.3f  cpu %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel", "")[:24], r.get("frac") or 0, (d.get("cpu_baseline") or {}).get("value")))
if "kernels" in d: print("   ", {k: round(v["ms"], 4) for k, v in d["kernels"].items() if "ms" in v})
PY
done
grep memguard $OUT/*.err | cut -c1-200
head -12 $OUT/batch_timeline.txt
