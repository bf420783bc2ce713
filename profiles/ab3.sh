ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in head gather stream2; do
  GPD_HIP_LIB=$ROOT/ab/libgpd_hip_$v.so python $ROOT/bench.py --cpu-samples 0 --batch-clouds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v   step %.3f ms  image stage %.3f ms' % (d['ms_per_step'], d['kernels']['grasp_image_kernel']['ms']))"
done; done
