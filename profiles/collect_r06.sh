#!/bin/bash
# Round 6, the final collection on the final kernels: legs A (GPU suite, default line, kernel trace, SQ counters, lane occupancy),
# B (side configurations, batch64, batch timeline, config 4) and C (the two TCC passes) in one gpurun call — each leg ran on its own
# first (collect_r06{a,b,c}.sh) and none took a box down; every command under profiles/memguard.py.
#   profiles/collect_r06.sh   ->  gpurun_out/r06{a,b,c}/ ; copy the summaries into profiles/ with profiles/install_r06.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash profiles/collect_r06a.sh r06a
G="python $ROOT/profiles/memguard.py --rss-gb 24"
mkdir -p gpurun_out/r06b
$G --seconds 300 -- python bench.py --config 2o --cpu-samples 0 > gpurun_out/r06b/bench_config2o.json 2> gpurun_out/r06b/bench_config2o.err
bash profiles/collect_r06b.sh r06b
bash profiles/collect_r06c.sh r06c
