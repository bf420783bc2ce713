#!/bin/bash
# copies what profiles/collect_r06.sh left under gpurun_out/ into profiles/ (the files the docs cite)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
A=gpurun_out/r06a; B=gpurun_out/r06b; C=gpurun_out/r06c
cp $A/bench_default.json profiles/r06_bench_default.json
cp $A/rocprof_summary.txt profiles/r06_rocprof_summary.txt
cp $A/pmc_sq.txt profiles/r06_pmc_sq.txt; cp $A/pmc_sq.json profiles/r06_pmc_sq.json
cp $A/pmc_lanes.txt profiles/r06_pmc_lanes.txt
grep -E "passed|failed" $A/gputests.txt | tail -2 > profiles/r06_gputests_tail.txt
for f in config2o config3a config3b config4 batch64; do cp $B/bench_$f.json profiles/r06_bench_$f.json; done
cp $B/batch_timeline.txt profiles/r06_batch_timeline.txt
cp $C/traffic.json profiles/r06_traffic.json; cp $C/pmc_summary.txt profiles/r06_pmc_tcc.txt
ls -la profiles/r06_*
