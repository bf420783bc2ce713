#!/bin/bash
# the round-end checks in one call: smoke(), the -m gpu suite, the RCCL path of bench.py on one GPU (world size 1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04final
mkdir -p $OUT
cd $ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
GPD_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-samples 0 --no-live-pmc --batch-clouds 6 --batch-passes 2 > $OUT/bench_nccl.json 2> $OUT/bench_nccl.err
echo "bench nccl rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_nccl.json').read().strip().splitlines()[-1]); print('nccl path: value %.0f, batch %.0f cand/s, rank spread %s' % (d['value'], d['batch_end_to_end']['cand_per_s'], d['batch_end_to_end']['rank_clouds_per_s']))"
