#!/bin/bash
# A/B of the batch entry on ONE GPU box: ab/libgpd_hip_A.so against the in-tree library.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
show() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); b=d['batch_end_to_end']
print('%-4s batch %.0f cand/s  %.1f clouds/s  wall %.3f s  (%d clouds, %d cand)' % (sys.argv[2], b['cand_per_s'], b['clouds_per_s'], b['wall_s'], b['clouds'], b['candidates']))
" $1 $2; }
for rep in 1 2; do
  GPD_HIP_LIB=$ROOT/ab/libgpd_hip_A.so python bench.py --mode batch --clouds 24 --steps 2 --warmup 1 > gpurun_out/abb_A.json 2> gpurun_out/abb_A.err; show gpurun_out/abb_A.json A
  python bench.py --mode batch --clouds 24 --steps 2 --warmup 1 > gpurun_out/abb_B.json 2> gpurun_out/abb_B.err; show gpurun_out/abb_B.json B
done
