#!/bin/bash
# A/B timing on ONE GPU box: ab/libgpd_hip_A.so (a build of another revision) against the in-tree library.
#   profiles/ab.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
show() { python -c "
import json,sys
d=json.load(open(sys.argv[1]))
k=d['kernels']
print('%-4s value %.0f  search %.3f  images %.3f  lenet %.3f  conv1 %.3f conv2 %.3f fc1 %.3f  detect %.2f ms' % (sys.argv[2], d['value'], d['search']['kernel_ms'], k['grasp_image_kernel']['ms'], k['lenet_forward']['ms'], k['conv1_i8_kernel']['ms'], k['conv2_bf16_kernel']['ms'], k['fc1_bf16_kernel']['ms'], d['detect_end_to_end']['wall_ms']))
" $1 $2; }
for rep in 1 2; do
  GPD_HIP_LIB=$ROOT/ab/libgpd_hip_A.so python bench.py --cpu-samples 0 --batch-clouds 0 "$@" > gpurun_out/ab_A.json 2> gpurun_out/ab_A.err; show gpurun_out/ab_A.json A
  python bench.py --cpu-samples 0 --batch-clouds 0 "$@" > gpurun_out/ab_B.json 2> gpurun_out/ab_B.err; show gpurun_out/ab_B.json B
done
