#!/bin/bash
# per-kernel average durations of one bench run (rocprofv3 kernel trace), printed as a table
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-samples 0 "$@" > /tmp/pk.json 2>/dev/null
python - <<PY
import sqlite3,glob,json
print(json.loads(open("/tmp/pk.json").read())["value"])
db=glob.glob("/tmp/pk/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
for r in c.execute("select name,count(*),avg(end-start)/1e3 from kernels group by name order by sum(end-start) desc limit 7"): print("%-60s %3d %10.1f us"%(r[0][:60],r[1],r[2]))
PY
