#!/bin/bash
# (round 5: the C2_EXP switches this script compiled with were removed from the shipped sources — the experiment is closed, its
#  results are in the .txt beside this file; a new one of the kind: profiles/mkpatched.sh NAME FILE 'sed-script')
# What bounds conv2_mfma_kernel (MfmaUtil 74 %, 117 of the 155 TFLOP/s the pipe sustains)?  Timing-only builds (wrong
# results): 1 = without the VALU tail of filters 48, 49; 2 = additionally without the LDS operand reads in the loop (the
# operands of step 0 reused: what the MFMA issue structure alone takes); 3 = without the sched_barriers.
#   gpurun -- bash profiles/r04_conv2_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04c2
mkdir -p $OUT $ROOT/ab
cd $ROOT
for v in 0 1 2 3; do
  T=$(mktemp -d); mkdir -p $T/gpd_amd/csrc $T/include
  cp gpd_amd/csrc/*.hip gpd_amd/csrc/*.h gpd_amd/csrc/*.cpp gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp include/*.h $T/include/
  X=""; [ $v != 0 ] && X="-DC2_EXP=$v"
  make -s -C $T/gpd_amd/csrc -j16 EXTRA="$X" ../libgpd_hip.so > /dev/null 2>&1
  cp $T/gpd_amd/libgpd_hip.so ab/libgpd_hip_c2_$v.so; rm -rf $T
done
for rep in 1 2; do
for v in 0 1 2 3; do
  GPD_HIP_LIB=$ROOT/ab/libgpd_hip_c2_$v.so python bench.py --cpu-samples 0 --batch-clouds 0 --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('exp $v: step %.3f ms  conv1 %.3f  conv2 %.3f  ip1 %.3f  images %.3f' % (d['ms_per_step'], k['conv1_mfma_kernel']['ms'], k['conv2_mfma_kernel']['ms'], k['fc1_mfma_kernel']['ms'], k['grasp_image_kernel']['ms']))"
done
done | tee $OUT/ab.txt
