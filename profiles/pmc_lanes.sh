#!/bin/bash
# VALU lane occupancy of the image / search kernels: of the 64 lanes of a wave, how many are switched on while
# a vector instruction executes.  One PMC pass (SQ counters + kernel trace only), kernels serialised with GPD_IMG_SERIAL (profiling build):
#   lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU   (thread-cycles per wave-cycle of VALU execution: 64 = every lane on; the units of
#           the two counters are calibrated on the LeNet kernels of the same pass, whose vector instructions run with all 64 lanes on)
#   profiles/pmc_lanes.sh <tag>  ->  gpurun_out/lanes_<tag>/summary.txt
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/lanes_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPD_HIP_LIB=${GPD_HIP_LIB:-$ROOT/gpd_amd/libgpd_hip_prof.so}
export GPD_IMG_SERIAL=1
python $ROOT/profiles/memguard.py --rss-gb 24 --seconds 300 -- rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/p1 -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > $OUT/log1.txt 2>&1
if ! ls $OUT/p1/*/*.db > /dev/null 2>&1 && ! ls $OUT/p1/*.db > /dev/null 2>&1; then
  # a counter of the list may not exist on this rocprofv3: the three that matter alone
  python $ROOT/profiles/memguard.py --rss-gb 24 --seconds 300 -- rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/p1 -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > $OUT/log1b.txt 2>&1
fi
cd $ROOT
python - <<PY
import sqlite3, glob
d, dur = {}, {}
dbs = glob.glob("$OUT/p1/**/*.db", recursive=True)
if not dbs:
    raise SystemExit("no counter database: see $OUT/log1.txt")
c = sqlite3.connect(dbs[0])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
nm = "kernel_name" if "kernel_name" in cols else "name"
for k, cn, v in c.execute("select %s,counter_name,avg(value) from counters_collection group by %s,counter_name" % (nm, nm)):
    d.setdefault(k, {})[cn] = v
for k, v in c.execute("select name,avg(end-start) from kernels group by name"):
    dur[k] = v
lines = []
for k, v in sorted(d.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if dur.get(k, 0) < 2e4:
        continue
    f = lambda n: v.get(n, 0.0)
    act = max(f("SQ_ACTIVE_INST_VALU"), 1.0)
    ins = max(f("SQ_INSTS_VALU"), 1.0)
    lines.append("%-46s %8.1f us | SQ_THREAD_CYCLES_VALU %.4g  SQ_ACTIVE_INST_VALU %.4g  SQ_INSTS_VALU %.4g | thread-cycles per active quad-cycle %7.2f | per VALU instruction %7.2f | active quad-cycles per instruction %5.2f"
                 % (k[:46], dur[k] / 1e3, f("SQ_THREAD_CYCLES_VALU"), f("SQ_ACTIVE_INST_VALU"), f("SQ_INSTS_VALU"),
                    f("SQ_THREAD_CYCLES_VALU") / act, f("SQ_THREAD_CYCLES_VALU") / ins, act / ins))
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
