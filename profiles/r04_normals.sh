#!/bin/bash
# gpurun -- bash profiles/r04_normals.sh <tag>: the parity tests that go through gpd_hip_estimate_normals, then its kernel times
TAG=${1:-r04n}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest -x -q -m gpu tests/test_config1_krylon.py tests/test_golden.py tests/test_gpu_dense_scan.py tests/test_gpu_fuzz.py tests/test_gpu_preprocess.py "tests/test_gpu_configs.py" "tests/test_ref_pin.py" -k "normals or krylon or golden or table_mug or fuzz or preprocess or extras or two_cameras or direction" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/nrm -o n -- python $ROOT/profiles/normals_times.py > $OUT/normals.log 2>&1
echo "normals rc=$?"; grep points $OUT/normals.log
python - $OUT <<'PY'
import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/nrm/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
q = "select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start) from %s k join %s s on k.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kt, sym)
with open(sys.argv[1] + "/normals_kernels.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace of profiles/normals_times.py: 6 calls on a 30k-point cloud (the min column), 6 on a raw 120k-point scan (the max column)\n")
    for name, n, avg, mn, mx in c.execute(q):
        if "normals" in name or "split_soa" in name:
            line = "%-62s calls %4d  avg %9.1f us  min %9.1f  max %9.1f" % (name[:62], n, avg / 1e3, mn / 1e3, mx / 1e3)
            print(line); o.write(line + "\n")
PY
