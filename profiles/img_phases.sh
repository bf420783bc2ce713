#!/bin/bash
# Cumulative cost of the image kernels' phases, each kernel alone on the chip, on the instrumented build
# (EXTRA=-DGPD_IMG_EXITS profiles/mkpatched.sh exits images.hip @profiles/img_exits.patch -> ab/libgpd_hip_exits.so: the early
# returns live in that patch, not in the shipped source; they cost registers, so the absolute times are 10-20 % above the shipped
# kernels': read the differences).  GPD_IMG_EXIT=k leaves both kernels after phase k —
# shadow: 1 extract, 2 list, 3 count/place, 4 non-empty list, 5 walks, 6 first projection done, 7 second;
# points: 11 collect, 12 count/place, 13 non-empty list, 14 walks, 21 live groups listed, 22 live groups dilated,
# 23 min/max + scale, 15 first projection done, 16 second; 0 = whole kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export GPD_HIP_LIB=$ROOT/ab/libgpd_hip_exits.so GPD_IMG_SERIAL=1
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 11 12 13 14 21 22 23 15 16 0; do
  rm -rf /tmp/pk; GPD_IMG_EXIT=$k rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $ROOT/bench.py --steps 5 --warmup 1 --cpu-samples 0 --batch-clouds 0 > /dev/null 2>&1
  python - "$k" <<PY
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob("/tmp/pk/**/*.db", recursive=True)[0])
r = dict(c.execute("select name, avg(end-start)/1e3 from kernels where name like '%image_kernel<%false>%' group by name").fetchall())
sh = [v for k, v in r.items() if "shadow_image_kernel<6144" in k]
pt = [v for k, v in r.items() if "grasp_image_kernel<false" in k]
print("exit %2s: shadow %7.1f us   points %7.1f us" % (sys.argv[1], sh[0] if sh else -1, pt[0] if pt else -1))
PY
done
