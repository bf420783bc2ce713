#!/bin/bash
# second soak: the LeNet queue protocol (conv1's two-slot ring + image counter) and the batch entries 12 times over, then the
# differential fuzz over a few hundred further seeds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04soak2
mkdir -p $OUT
cd $ROOT
fail=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_lenet_stress.py tests/test_gpu_resident.py -m gpu -q -x > $OUT/stress_$i.log 2>&1 || { fail=1; echo "stress pass $i FAILED"; tail -20 $OUT/stress_$i.log; break; }
  tail -1 $OUT/stress_$i.log
done
echo "stress fail=$fail"
GPD_FUZZ_DETECT=${1:-360} GPD_FUZZ_WIDE=${2:-60} GPD_FUZZ_GEOMETRY=${3:-160} timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 6 > $OUT/fuzz.log 2>&1
echo "fuzz rc=$?"; tail -4 $OUT/fuzz.log
