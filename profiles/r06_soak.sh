#!/bin/bash
# Round-6 soak on the final kernels (centre sums inside the gather, narrow conv1, conv2's tile prefetch): the LeNet queue protocol and
# the batch / resident / preprocessing entries six times over, then the differential fuzz of the HIP path against the oracle over a few
# hundred further seeds (a third of them off the lattice).  Everything under profiles/memguard.py.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06soak
mkdir -p $OUT
cd $ROOT
G="python profiles/memguard.py --rss-gb 40"
fail=0
for i in $(seq 1 6); do
  $G --seconds 700 -- python -m pytest tests/test_gpu_lenet_stress.py tests/test_gpu_lenet_fast.py tests/test_gpu_resident.py tests/test_gpu_preprocess.py tests/test_centre_certificate.py -m gpu -q -x > $OUT/stress_$i.log 2>&1 || { fail=1; echo "stress pass $i FAILED"; tail -20 $OUT/stress_$i.log; break; }
  tail -1 $OUT/stress_$i.log
done
echo "stress fail=$fail"
GPD_FUZZ_DETECT=${1:-360} GPD_FUZZ_WIDE=${2:-60} GPD_FUZZ_GEOMETRY=${3:-160} $G --seconds 2400 -- python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 6 > $OUT/fuzz.log 2>&1
echo "fuzz rc=$?"; tail -4 $OUT/fuzz.log
