#!/bin/bash
# Round-5 evidence, one GPU-box pass (run through gpurun): bench lines of every BASELINE config, the rocprofv3
# kernel trace + the two PMC passes of the default line, the batch timeline.  Everything lands under
# gpurun_out/<tag>/; copy the summaries into profiles/.
#   profiles/collect_r05.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 400 python bench.py > $OUT/bench_config2.json 2> $OUT/bench_config2.err
timeout 400 python bench.py --config 3a > $OUT/bench_config3a.json 2> $OUT/bench_config3a.err
timeout 400 python bench.py --config 3b > $OUT/bench_config3b.json 2> $OUT/bench_config3b.err
timeout 600 python bench.py --config 4 --steps 5 --warmup 1 --cpu-samples 600 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
timeout 400 python bench.py --mode batch --clouds 64 --steps 3 --warmup 1 > $OUT/bench_batch64.json 2> $OUT/bench_batch64.err
cd /tmp && export TMPDIR=/tmp
P=$OUT/prof
mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-samples 0 --batch-clouds 0 --no-live-pmc > $P/stats.log 2>&1
# (the TCC passes — rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE — are NOT run any more: this script lost its GPU node twice in a
#  row on 2026-09-25, about when they start; profiles/pmc_tcc.sh has them for a day on which a node can be risked)
timeout 300 rocprofv3 --kernel-trace --stats -d $P/batch -o batch -- python $ROOT/bench.py --mode batch --clouds 16 --steps 2 --warmup 1 > $P/batch.log 2>&1
cd $ROOT
python profiles/summarize.py $P > $OUT/rocprof_summary.txt 2>&1
python profiles/summarize.py $P --traffic $OUT/traffic.json
python profiles/timeline.py $(find $P/batch -name "*.db" | head -1) --last-seconds 0.25 > $OUT/batch_timeline.txt 2>&1
for f in $OUT/bench_*.json; do python - "$f" <<PY
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%-28s value %10.0f  ms/step %8.3f  roofline %s %.3f  cpu %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel", "")[:24], r.get("frac") or 0, (d.get("cpu_baseline") or {}).get("value")))
PY
done
# SQ counters (MfmaUtil, LDS, waits) and the instruction mix, each its own PMC run
bash $ROOT/profiles/pmc_sq.sh $TAG > $OUT/pmc_sq.log 2>&1
bash $ROOT/profiles/pmc_mix.sh $TAG > $OUT/pmc_mix.log 2>&1
tail -20 $OUT/rocprof_summary.txt | cut -c1-160
cat $OUT/batch_timeline.txt | head -8
