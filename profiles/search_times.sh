#!/bin/bash
# per-kernel average durations of the search stage (rocprofv3 kernel trace): 30k-point cloud, 2564 samples, through
# gpd_hip_search (x4) and gpd_hip_detect (x4: adds plan_kernel and the image / LeNet kernels of 6077 candidates)
cat > /tmp/search_only.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from gpd_amd import api, synth
cl = synth.make_cloud(1234, 30000)
si = synth.sample_indices(cl, 2564)
ctx = api.Context(api.default_params(15))
ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
ctx.set_lenet_weights(synth.lenet_weights(15))
for _ in range(4):
    ctx.search(si)
for _ in range(4):
    ctx.detect(si)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o b -- python /tmp/search_only.py > /tmp/pk.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pk/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
for r in c.execute("select name,count(*),avg(end-start)/1e3 from kernels group by name order by sum(end-start) desc limit 14"): print("%-60s %3d %10.1f us"%(r[0][:60],r[1],r[2]))
PY
