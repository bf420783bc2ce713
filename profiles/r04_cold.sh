#!/bin/bash
# One fresh-lease pass as the driver does it: the default bench line FIRST (cold box), then the -m gpu suite, then the
# bench line again (warm box).   gpurun -- bash profiles/r04_cold.sh <tag> [pytest-args]
TAG=${1:-r04a}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_cold.json 2> $OUT/bench_cold.err
echo "bench cold rc=$?"
timeout 2400 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_warm.json 2> $OUT/bench_warm.err
echo "bench warm rc=$?"
python - $OUT <<'PY'
import json, sys
for nm in ("bench_cold", "bench_warm"):
    try:
        d = json.loads(open("%s/%s.json" % (sys.argv[1], nm)).read().strip().splitlines()[-1])
    except Exception as e:
        print(nm, "unreadable", e); continue
    b = d.get("batch_end_to_end") or {}
    print(nm, "value %.0f  ms/step %.3f  detect %.2f ms  batch %.0f cand/s  passes %s  growths %s" % (
        d["value"], d["ms_per_step"], d["detect_end_to_end"]["wall_ms"], b.get("cand_per_s", 0),
        json.dumps(b.get("passes", {}).get("cand_per_s_rank0")), b.get("passes", {}).get("buffer_growths_in_timed_passes")))
    print("   pass walls", [round(x, 1) for x in b.get("passes", {}).get("wall_ms_rank0", [])], "slowest host", b.get("passes", {}).get("host_ms_slowest_pass"))
PY
