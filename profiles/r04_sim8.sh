#!/bin/bash
# the driver's N = 8 command line with all eight ranks on the ONE GPU of this box (gloo carries the reductions: RCCL refuses
# several ranks on one device): the launch, the rendezvous, the NUMA binding, the whole-job sums — not a scaling measurement
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04sim8
mkdir -p $OUT
cd $ROOT
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 \
  bench.py --gpus 8 --steps 5 --warmup 2 --devices 0 --dist-backend gloo > $OUT/bench8.json 2> $OUT/bench8.err
echo "rc=$?"; tail -3 $OUT/bench8.err
python - <<P
import json
d=json.loads([l for l in open('$OUT/bench8.json') if l.startswith('{')][-1])
b=d.get('batch_end_to_end') or {}
print('n_gpus', d['n_gpus'], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'scaling', d['scaling'])
print('batch: clouds', b.get('clouds'), 'cand/s %.0f' % b.get('cand_per_s', 0), 'rank spread', b.get('rank_clouds_per_s'), 'growths', (b.get('passes') or {}).get('buffer_growths_in_timed_passes'))
print('binding', d.get('host_binding_rank0'))
P
