#!/bin/bash
# A/B on one box: the default replay (image stage, then LeNet, one stream) against GPD_REPLAY_PIPE=1 (the image stage of
# replay k + 1 beside the LeNet stage of replay k: two image buffers, two streams).  DESIGN §8.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-pipe}
mkdir -p $OUT
cd $ROOT
export GPD_HIP_LIB=${GPD_HIP_LIB:-$ROOT/gpd_amd/libgpd_hip_prof.so}  # the measurement switches exist in the profiling build only (make -C gpd_amd/csrc prof)
for rep in 1 2 3; do
  for mode in 0 1; do
    GPD_REPLAY_PIPE=$mode timeout 300 python bench.py --steps 40 --warmup 5 --cpu-samples 0 --batch-clouds 0 > $OUT/b_${mode}_$rep.json 2> $OUT/b_${mode}_$rep.err
    python - $OUT/b_${mode}_$rep.json $mode <<PY
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("pipe=%s value %9.0f cand/s  ms/step %.3f | images %.3f lenet %.3f conv1 %.3f conv2 %.3f fc1 %.3f" % (sys.argv[2], d["value"], d["ms_per_step"],
      k["grasp_image_kernel"]["ms"], k["lenet_forward"]["ms"], k["conv1_mfma_kernel"]["ms"], k["conv2_mfma_kernel"]["ms"], k["fc1_mfma_kernel"]["ms"]))
PY
  done
done
