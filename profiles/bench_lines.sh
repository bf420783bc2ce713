#!/bin/bash
# The committed bench lines of the round (run through gpurun after profiles/r02_traffic.json / r02_pmc_sq.json are in place,
# so that the default line carries the PMC traffic of the committed kernel sources).
#   profiles/bench_lines.sh <tag>  ->  gpurun_out/<tag>/bench_{default,config2,config3a,config3b,config4,batch64}.json
TAG=${1:-lines}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --config 2 > $OUT/bench_config2.json 2> $OUT/bench_config2.err
timeout 400 python bench.py --config 3a > $OUT/bench_config3a.json 2> $OUT/bench_config3a.err
timeout 400 python bench.py --config 3b > $OUT/bench_config3b.json 2> $OUT/bench_config3b.err
timeout 600 python bench.py --config 4 --steps 5 --warmup 1 --cpu-samples 600 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
timeout 400 python bench.py --mode batch --clouds 64 --steps 2 --warmup 1 > $OUT/bench_batch64.json 2> $OUT/bench_batch64.err
for f in $OUT/bench_*.json; do python - "$f" <<PY
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%-28s value %10.0f  ms/step %8.3f  roofline %s frac %.3f traffic %s  cpu %s  pre %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel", "")[:24], r.get("frac") or 0, r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"), (d.get("preprocess") or {}).get("kernel_ms")))
PY
done
