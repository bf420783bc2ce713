#!/bin/bash
# per-kernel "alone" durations (image kernels serialised: GPD_IMG_SERIAL) of several library builds on ONE box:
#   profiles/ab_kernels.sh NAME [NAME ...]     (ab/libgpd_hip_NAME.so; "tree" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  lib=$ROOT/ab/libgpd_hip_$v.so; [ "$v" = tree ] && lib=$ROOT/gpd_amd/libgpd_hip_prof.so  # GPD_IMG_SERIAL needs a -DGPD_PROFILING build (mkvariant.sh builds the variants that way)
  echo "== $v"
  GPD_HIP_LIB=$lib GPD_IMG_SERIAL=1 bash $ROOT/profiles/kernel_times.sh --batch-clouds 0 2>&1 | grep -E "image_kernel<|set_kernel|^[0-9]"
  GPD_HIP_LIB=$lib python $ROOT/bench.py --cpu-samples 0 --batch-clouds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   step %.3f ms  image stage %.3f ms' % (d['ms_per_step'], d['kernels']['grasp_image_kernel']['ms']))"
done
