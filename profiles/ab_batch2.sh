#!/bin/bash
# step time and batch throughput of several library builds, alternating, on ONE box:  profiles/ab_batch2.sh NAME [NAME ...]  ("tree" = in-tree)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in "$@"; do
  lib=$ROOT/ab/libgpd_hip_$v.so; [ "$v" = tree ] && lib=$ROOT/gpd_amd/libgpd_hip.so
  GPD_HIP_LIB=$lib python $ROOT/bench.py --cpu-samples 0 --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; b=d['batch_end_to_end']; print('%-8s step %.3f ms  conv1 %.3f conv2 %.3f fc1 %.3f images %.3f | detect %.2f ms | batch %.0f cand/s (median pass %.0f)' % ('$v', d['ms_per_step'], k['conv1_mfma_kernel']['ms'], k['conv2_mfma_kernel']['ms'], k['fc1_mfma_kernel']['ms'], k['grasp_image_kernel']['ms'], d['detect_end_to_end']['wall_ms'], b['cand_per_s'], b['passes']['cand_per_s_rank0']['median']))"
done; done
