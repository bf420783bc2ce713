#!/bin/bash
# hand_eval_kernel's LDS table: 2048 entries (four workgroups per CU) against 1664 (five)   gpurun -- bash profiles/r04_he_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/ab $ROOT/gpurun_out/r04he
cd $ROOT
for v in 2048 1664 1536; do
  T=$(mktemp -d); mkdir -p $T/gpd_amd/csrc $T/include
  cp gpd_amd/csrc/*.hip gpd_amd/csrc/*.h gpd_amd/csrc/*.cpp gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp include/*.h $T/include/
  make -s -C $T/gpd_amd/csrc -j16 EXTRA="-DHE_COMPACT_N=$v" ../libgpd_hip.so > /dev/null 2>&1
  cp $T/gpd_amd/libgpd_hip.so ab/libgpd_hip_he$v.so; rm -rf $T
done
for rep in 1 2 3; do for v in 2048 1664 1536; do
  GPD_HIP_LIB=$ROOT/ab/libgpd_hip_he$v.so python bench.py --cpu-samples 0 --batch-clouds 0 --no-live-pmc --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HE_COMPACT $v: search %.4f ms  detect %.3f ms (search %.3f)' % (d['search']['kernel_ms'], d['detect_end_to_end']['wall_ms'], d['detect_end_to_end']['kernel_ms']['search']))"
done; done | tee $ROOT/gpurun_out/r04he/ab.txt
