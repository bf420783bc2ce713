import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from gpd_amd import api, synth
cl = synth.make_cloud(1234, 30000)
si = synth.sample_indices(cl, 2564)
w = synth.lenet_weights(15)
ctx = api.Context(api.default_params(15))
ctx.set_lenet_weights(w)
def t(f, n=5):
    f(); ts=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f(); ts.append((time.perf_counter()-t0)*1e3)
    return min(ts), r
print("upload  %.2f ms" % t(lambda: ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"]))[0])
ms, hands = t(lambda: ctx.search(si)); print("search  %.2f ms (kernels %.2f)" % (ms, ctx.stage_ms()[0]))
ms, (h2, nc) = t(lambda: ctx.detect(si)); sm = ctx.stage_ms(); print("detect  %.2f ms  stages %s  cands %d" % (ms, sm, nc))
hands = ctx.search(si)
from gpd_amd.api import HAND_DTYPE
ms, _ = t(lambda: ctx.images(hands, download=False)); print("images(no download) %.2f ms (kernels %.2f)" % (ms, ctx.stage_ms()[1]))
