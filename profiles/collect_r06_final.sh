#!/bin/bash
# Round 6, the collection on the FINAL kernels of the round's second session (ip1 by LDS-DMA on blocked operands, ip2 with the
# combine pass inside, conv2 without its fourth column tile).  Leg C (the two TCC passes) runs first, as its own gpurun call, and
# its traffic.json is installed as profiles/r06_traffic.json — then this script: legs A and B as in collect_r06.sh, and at the end
# the default line once more with the fresh SQ-counter file in place, so that the committed line is the one the driver's run
# will produce (roofline.traffic and the LDS figures read from the stamped profile files).
#   profiles/collect_r06_final.sh   ->  gpurun_out/r06{a,b}/ ; then profiles/install_r06.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash profiles/collect_r06a.sh r06a
G="python $ROOT/profiles/memguard.py --rss-gb 24"
mkdir -p gpurun_out/r06b
$G --seconds 300 -- python bench.py --config 2o --cpu-samples 0 > gpurun_out/r06b/bench_config2o.json 2> gpurun_out/r06b/bench_config2o.err
bash profiles/collect_r06b.sh r06b
cp gpurun_out/r06a/pmc_sq.json profiles/r06_pmc_sq.json
$G --seconds 400 -- python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06a/bench_default.json").read().strip().splitlines()[-1])
print("final default line: value %.0f  ms/step %.3f  roofline.frac %.3f  traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
PY
