#!/bin/bash
# instruction counts / pipe occupancy of the normals kernels (one PMC pass around profiles/normals_times.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04npmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p1 -o b -- python $ROOT/profiles/normals_times.py > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT -d $OUT/p2 -o b -- python $ROOT/profiles/normals_times.py > $OUT/log2.txt 2>&1
python - $OUT <<'PY'
import sqlite3, glob, sys
d = {}; dur = {}
for p in ("p1", "p2"):
    dbs = glob.glob(sys.argv[1] + "/%s/**/*.db" % p, recursive=True)
    if not dbs:
        print("no db for", p); continue
    c = sqlite3.connect(dbs[0])
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    nm = "kernel_name" if "kernel_name" in cols else "name"
    for k, cn, v in c.execute("select %s,counter_name,min(value) from counters_collection group by %s,counter_name" % (nm, nm)):
        d.setdefault(k, {})[cn] = v   # min over dispatches = the 30k-point cloud
    for k, v in c.execute("select name,min(end-start) from kernels group by name"):
        dur[k] = v
for k, v in sorted(d.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if "normals" not in k:
        continue
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    sq = g * 1024 / 4.0
    f = lambda n: v.get(n, 0)
    print("%-40s %7.1f us | Minst VALU %6.2f (f64 add %5.2f mul %5.2f cvt %5.2f) SALU %6.2f LDS %5.2f VMEM %5.2f | waves %d | per SIMD: resident %.2f, executing VALU %.2f LDS %.2f SCA %.2f VMEM %.2f, waiting any %.2f inst %.2f"
          % (k[:40], dur[k] / 1e3, f("SQ_INSTS_VALU") / 1e6, f("SQ_INSTS_VALU_ADD_F64") / 1e6, f("SQ_INSTS_VALU_MUL_F64") / 1e6, f("SQ_INSTS_VALU_CVT") / 1e6,
             f("SQ_INSTS_SALU") / 1e6, f("SQ_INSTS_LDS") / 1e6, f("SQ_INSTS_VMEM") / 1e6, f("SQ_WAVES"), f("SQ_WAVE_CYCLES") / sq, f("SQ_ACTIVE_INST_VALU") / sq,
             f("SQ_ACTIVE_INST_LDS") / sq, f("SQ_ACTIVE_INST_SCA") / sq, f("SQ_ACTIVE_INST_VMEM") / sq, f("SQ_WAIT_ANY") / sq, f("SQ_WAIT_INST_ANY") / sq))
PY
