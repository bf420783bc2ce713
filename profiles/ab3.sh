#!/bin/bash
# step / image-stage time of several library builds, alternating, on ONE box:  profiles/ab3.sh NAME [NAME ...]  ("tree" = in-tree)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in "$@"; do
  lib=$ROOT/ab/libgpd_hip_$v.so; [ "$v" = tree ] && lib=$ROOT/gpd_amd/libgpd_hip.so
  GPD_HIP_LIB=$lib python $ROOT/bench.py --cpu-samples 0 --batch-clouds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-8s step %.3f ms  images %.3f  conv1 %.3f conv2 %.3f fc1 %.3f  search %.3f  detect %.2f' % ('$v', d['ms_per_step'], k['grasp_image_kernel']['ms'], k['conv1_i8_kernel']['ms'], k['conv2_bf16_kernel']['ms'], k['fc1_bf16_kernel']['ms'], d['search']['kernel_ms'], d['detect_end_to_end']['wall_ms']))"
done; done
