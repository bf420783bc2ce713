cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pw; rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o b -- python $R/bench.py --steps 3 --warmup 1 --cpu-samples 0 --batch-clouds 0 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob("/tmp/pw/**/*.db", recursive=True)[0])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
nm = "kernel_name" if "kernel_name" in cols else "name"
for r in c.execute("select %s, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by %s order by avg(value) desc limit 4" % (nm, nm)): print("%-50s %.1f MB written per launch" % (r[0][:50], r[1] * 1024 / 1e6))
PY
