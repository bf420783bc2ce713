#!/bin/bash
# gpurun -- bash profiles/r04_micro.sh <tag>: the co-residency / MFMA-peak microbenchmark and the normals kernel times
TAG=${1:-r04m}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/corun.hip -o /tmp/corun && timeout 300 /tmp/corun > $OUT/corun.txt 2>&1
echo "corun rc=$?"; cat $OUT/corun.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/nrm -o n -- python $ROOT/profiles/normals_times.py > $OUT/normals.log 2>&1
echo "normals rc=$?"; tail -3 $OUT/normals.log
python - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/nrm/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(sys.argv[1] + "/normals_kernels.txt", "w") as o:
        for r in rows:
            if "normals" in r["Name"] or "split_soa" in r["Name"]:
                line = "%-60s calls %5s  avg %10.1f us  total %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
                print(line); o.write(line + "\n")
PY
