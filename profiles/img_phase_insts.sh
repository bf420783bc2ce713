#!/bin/bash
# Cumulative INSTRUCTION counts of the image kernels' phases (same exits as profiles/img_phases.sh, instrumented build):
# wave-instructions per candidate by class, from SQ_INSTS_* of one launch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export GPD_HIP_LIB=$ROOT/ab/libgpd_hip_exits.so GPD_IMG_SERIAL=1
cd /tmp && export TMPDIR=/tmp
for k in 11 30 31 32 12 0; do
  rm -rf /tmp/pk; GPD_IMG_EXIT=$k rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d /tmp/pk -o b -- python $ROOT/bench.py --steps 2 --warmup 1 --cpu-samples 0 --batch-clouds 0 > /dev/null 2>&1
  python - "$k" <<PY
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob("/tmp/pk/**/*.db", recursive=True)[0])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
nm = "kernel_name" if "kernel_name" in cols else "name"
d = {}
for k, cn, v in c.execute("select %s,counter_name,avg(value) from counters_collection where %s like '%%image_kernel<%%false>%%' group by %s,counter_name" % (nm, nm, nm)):
    d.setdefault("shadow" if "shadow" in k and "6144" in k else ("points" if "grasp_image_kernel<false" in k else "other"), {})[cn] = v / 5000.0
out = []
for kn in ("shadow", "points"):
    v = d.get(kn, {})
    out.append("%s VALU %7.0f SALU %6.0f LDS %5.0f VMEM %4.0f" % (kn, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM", 0)))
print("exit %2s: %s" % (sys.argv[1], "   ".join(out)))
PY
done
