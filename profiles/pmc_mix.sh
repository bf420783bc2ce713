#!/bin/bash
# Instruction mix of the image / search kernels (two PMC passes, kernels serialised with GPD_IMG_SERIAL):
#   per kernel: wave-instructions by class, and the share of the SIMD cycles each class keeps a wave "executing"
#   (SQ_ACTIVE_INST_* are per-wave quad-cycles; / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / 4) = average waves per SIMD in that state)
#   profiles/pmc_mix.sh <tag>  ->  gpurun_out/mix_<tag>/summary.txt
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPD_HIP_LIB=${GPD_HIP_LIB:-$ROOT/gpd_amd/libgpd_hip_prof.so}  # the measurement switches exist in the profiling build only (make -C gpd_amd/csrc prof)
export GPD_IMG_SERIAL=1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p1 -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM -d $OUT/p2 -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > $OUT/log2.txt 2>&1
cd $ROOT
python - <<PY
import sqlite3, glob
d = {}
dur = {}
for p in ("p1", "p2"):
    dbs = glob.glob("$OUT/%s/**/*.db" % p, recursive=True)
    if not dbs:
        print("no db for", p); continue
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    nm = "kernel_name" if "kernel_name" in cols else "name"
    for k, cn, v in c.execute("select %s,counter_name,avg(value) from counters_collection group by %s,counter_name" % (nm, nm)):
        d.setdefault(k, {})[cn] = v
    for k, v in c.execute("select name,avg(end-start) from kernels group by name"):
        dur[k] = v
lines = []
for k, v in sorted(d.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if dur.get(k, 0) < 1e5 or "mfma" in k or "conv1_i8" in k or "_bf16_kernel" in k:
        continue
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    simd_quads = g * 1024 / 4.0
    f = lambda n: v.get(n, 0)
    lines.append("%-44s %7.1f us | Minst: VALU %6.2f (f64 fma %5.2f add %5.2f mul %5.2f, cvt %5.2f, int32 %5.2f) SALU %6.2f LDS %5.2f (atomic %4.2f) VMEM %4.2f | waves/SIMD executing: VALU %.2f LDS %.2f SCA %.2f VMEM %.2f | resident %.2f"
                 % (k[:44], dur[k] / 1e3, f("SQ_INSTS_VALU") / 1e6, f("SQ_INSTS_VALU_FMA_F64") / 1e6, f("SQ_INSTS_VALU_ADD_F64") / 1e6, f("SQ_INSTS_VALU_MUL_F64") / 1e6,
                    f("SQ_INSTS_VALU_CVT") / 1e6, f("SQ_INSTS_VALU_INT32") / 1e6, f("SQ_INSTS_SALU") / 1e6, f("SQ_INSTS_LDS") / 1e6, f("SQ_INSTS_LDS_ATOMIC") / 1e6, f("SQ_INSTS_VMEM") / 1e6,
                    f("SQ_ACTIVE_INST_VALU") / simd_quads, f("SQ_ACTIVE_INST_LDS") / simd_quads, f("SQ_ACTIVE_INST_SCA") / simd_quads, f("SQ_ACTIVE_INST_VMEM") / simd_quads,
                    f("SQ_WAVE_CYCLES") / simd_quads))
open("$OUT/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
