#!/bin/bash
# MFMA utilisation / stall counters per kernel (own rocprofv3 run, PMC only + kernel trace).
# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); SQ_* wave counters are
# fractions of SQ_WAVE_CYCLES.
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 > $OUT/log.txt 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("$OUT/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(counters_collection)")]
nm="kernel_name" if "kernel_name" in cols else "name"
rows=c.execute("select %s,counter_name,avg(value) from counters_collection group by %s,counter_name"%(nm,nm)).fetchall()
dur=dict(c.execute("select name,avg(end-start) from kernels group by name").fetchall())
d={}
for k,cn,v in rows: d.setdefault(k,{})[cn]=v
for k,v in sorted(d.items(), key=lambda kv:-dur.get(kv[0],0)):
    if dur.get(k,0)<2e4: continue
    g=v.get("GRBM_GUI_ACTIVE",0)/8.0  # the counter is summed over the 8 XCDs

    print("%-44s dur %8.1f us  clk %.2f GHz  MfmaUtil %5.1f%%  wave_cyc %.3g  wait_inst %.1f%%  wait_any %.1f%%  active %.1f%%  lds_conf/idx %.1f%%"%(k[:44],dur[k]/1e3,g/dur[k] if dur.get(k) else 0,100*v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g*1024) if g else 0,v.get("SQ_WAVE_CYCLES",0),100*v.get("SQ_WAIT_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),100*v.get("SQ_WAIT_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),100*v.get("SQ_ACTIVE_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),100*v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
