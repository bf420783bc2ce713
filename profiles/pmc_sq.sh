#!/bin/bash
# SQ counters per kernel of the default bench line (own rocprofv3 run: PMC + kernel trace only):
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
#   LdsUtil  = SQ_LDS_IDX_ACTIVE (all LDS-array cycles, MI355X_MICROARCH.md §LDS) / (GRBM_GUI_ACTIVE / 8 * 256 CUs):
#              the fraction of the chip's LDS-array cycles that moved data — the LDS roofline of the latency-bound
#              image kernels (peak 256 B/clk/CU); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
#   the SQ_* wave counters are fractions of SQ_WAVE_CYCLES.
#   profiles/pmc_sq.sh <tag>  ->  gpurun_out/pmc_<tag>/{summary.txt,summary.json}
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > $OUT/log.txt 2>&1
cd $ROOT
python - <<PY
import sqlite3, glob, json, sys
sys.path.insert(0, "$ROOT")
import bench
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
nm = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute("select %s,counter_name,avg(value) from counters_collection group by %s,counter_name" % (nm, nm)).fetchall()
dur = dict(c.execute("select name,avg(end-start) from kernels group by name").fetchall())
d = {}
for k, cn, v in rows:
    d.setdefault(k, {})[cn] = v
out = {}
lines = []
for k, v in sorted(d.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if dur.get(k, 0) < 2e4:
        continue
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0  # the counter is summed over the 8 XCDs
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    rec = dict(dur_us=dur[k] / 1e3, clk_ghz=g / dur[k] if dur.get(k) else 0,
               mfma_util=v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024) if g else 0,
               lds_util=v.get("SQ_LDS_IDX_ACTIVE", 0) / (g * 256) if g else 0,
               lds_conflict=v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
               wait_inst=v.get("SQ_WAIT_INST_ANY", 0) / wc, wait_any=v.get("SQ_WAIT_ANY", 0) / wc, active=v.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    out[k] = rec
    lines.append("%-46s dur %8.1f us  clk %.2f GHz  MfmaUtil %5.1f%%  LdsUtil %5.1f%% (conflict %4.1f%% of it)  wait_inst %4.1f%%  wait_any %4.1f%%  active %4.1f%%"
                 % (k[:46], rec["dur_us"], rec["clk_ghz"], 100 * rec["mfma_util"], 100 * rec["lds_util"], 100 * rec["lds_conflict"],
                    100 * rec["wait_inst"], 100 * rec["wait_any"], 100 * rec["active"]))
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump({"source_hashes": bench.source_hashes(), "kernels": out}, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
print("\n".join(lines))
PY
