import sys, os
os.environ.setdefault("GPD_HIP_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpd_amd", "libgpd_hip_prof.so"))  # the timing switches exist in the profiling build only
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpd_amd import api, synth
cl = synth.make_cloud(1234, 30000)
si = synth.sample_indices(cl, 2564)
ctx = api.Context(api.default_params(15))
ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
ctx.set_lenet_weights(synth.lenet_weights(15))
for _ in range(4):
    ctx.detect(si)
