#!/bin/bash
# (round 5: the NL_SKIP_SORT / NL_SKIP_STORE switches this script compiled with were removed from the shipped sources — the experiment is closed, its
#  results are in the .txt beside this file; a new one of the kind: profiles/mkpatched.sh NAME FILE 'sed-script')
# where the time of normals_list_kernel goes: the shipped kernel against builds without the sort / without the gather + store
# (wrong results, timing only).  gpurun -- bash profiles/r04_normals_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04nab
mkdir -p $OUT $ROOT/ab
cd $ROOT
for v in base nosort nostore; do
  T=$(mktemp -d); mkdir -p $T/gpd_amd/csrc $T/include
  cp gpd_amd/csrc/*.hip gpd_amd/csrc/*.h gpd_amd/csrc/*.cpp gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp include/*.h $T/include/
  X=""; [ $v = nosort ] && X="-DNL_SKIP_SORT"; [ $v = nostore ] && X="-DNL_SKIP_STORE"
  make -s -C $T/gpd_amd/csrc -j16 EXTRA="$X" ../libgpd_hip.so > /dev/null 2>&1
  cp $T/gpd_amd/libgpd_hip.so ab/libgpd_hip_$v.so; rm -rf $T
done
cd /tmp && export TMPDIR=/tmp
for v in base nosort nostore; do
  rm -rf /tmp/nab; GPD_HIP_LIB=$ROOT/ab/libgpd_hip_$v.so rocprofv3 --kernel-trace --stats -d /tmp/nab -o n -- python $ROOT/profiles/normals_times.py > /dev/null 2>&1
  python - $v <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob("/tmp/nab/**/*.db", recursive=True)[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
for name, n, mn, mx in c.execute("select s.kernel_name, count(*), min(k.end-k.start), max(k.end-k.start) from %s k join %s s on k.kernel_id=s.id group by s.kernel_name" % (kt, sym)):
    if "normals_list_kernel" in name or "normals_finish" in name:
        print("%-8s %-40s 30k points %7.1f us   120k points %7.1f us" % (sys.argv[1], name[8:40], mn / 1e3, mx / 1e3))
PY
done | tee $OUT/ab.txt
