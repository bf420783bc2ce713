"""Per-phase cycle counters of the image kernels (GPD_IMG_TIMING=1) on the benchmark's 5000-candidate list."""
import os, sys
os.environ.setdefault("GPD_HIP_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpd_amd", "libgpd_hip_prof.so"))  # the timing switches exist in the profiling build only
os.environ["GPD_IMG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpd_amd import api, synth
cloud = synth.make_cloud(1234, 30000)
ctx = api.Context(api.default_params(15))
ctx.upload_cloud(cloud["xyz"], cloud["normals"], cloud["cam_source"], cloud["view_points"])
si = synth.sample_indices(cloud, 2564)
hands = ctx.search(si)
hf = hands.copy(); bench._filter_workspace(hf, ctx.params)
flat = hf.reshape(-1); vidx = np.flatnonzero(flat["valid"]); flat["valid"][vidx[5000:]] = 0
ctx.images(hf, download=False)
print("---- second pass", file=sys.stderr)
ctx.images(hf, download=False)
