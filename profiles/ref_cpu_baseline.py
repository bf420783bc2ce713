"""The reference's OWN code as the CPU baseline: oracle/_ref/libgpd_ref.so — the translation units of
/root/reference compiled unmodified through the test-only Eigen / PCL / OpenCV interface subsets of oracle/shim — timed on
bench.py's candidate list: cloud seed 1234, 30k points, 2564 samples.  ONE thread (the reference's OpenMP loops are racy,
SURVEY §9-Q9; oracle/build_ref.sh builds without -fopenmp).  Runs in the BUILD CONTAINER only (the library needs
/root/reference); the result is committed as profiles/r04_ref_cpu_baseline.json and bench.py prints it inside
`cpu_baseline` next to the OpenMP oracle's figure measured on the GPU box.

The shim's Eigen is plain loops (no SIMD, no blocking): the real Eigen build of the reference is faster per core, so this
is a LOWER bound of the reference's own single-thread rate, on another machine's core — stated in the file.

    python profiles/ref_cpu_baseline.py [classify_images]
"""
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpd_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    n_cls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    assert ref.available(), "oracle/_ref/libgpd_ref.so is missing (bash oracle/build_ref.sh A)"
    cl = synth.make_cloud(1234, 30000)
    si = synth.sample_indices(cl, 2564)
    real = dict(np.load(os.path.join(ROOT, "tests", "golden", "lenet15_params.npz")))
    w = synth.lenet_weights(15, real=real)
    p = orc.default_params(15)
    det = ref.Detector(p, weights=w)
    rc = ref.Cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    rc.set_sample_indices(si)
    t0 = time.perf_counter()
    hands = det.generate(rc, len(si))
    t_search = time.perf_counter() - t0
    t0 = time.perf_counter()
    valid = det.filter_workspace()
    t_filter = time.perf_counter() - t0
    n_valid = int(valid.sum())
    t0 = time.perf_counter()
    img, cand = det.images(rc, n_valid + 16)
    t_img = time.perf_counter() - t0
    assert len(img) == n_valid
    # the oracle on the same list: the bytes must agree (this is the pin, re-checked on the bench's own list)
    oh = orc.filter_workspace(p, orc.search(p, cl["xyz"], cl["normals"], si))
    assert np.array_equal(oh["valid"], valid)
    oimg, ocand = orc.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], oh)
    assert np.array_equal(ocand, cand) and np.array_equal(oimg, img)
    n_cls = min(n_cls, n_valid)
    t0 = time.perf_counter()
    sc = det.classify(img[:n_cls])
    t_cls = time.perf_counter() - t0
    per_cand = t_img / n_valid + t_cls / n_cls
    out = {
        "kind": "reference sources through oracle/shim, 1 thread",
        "value": 1.0 / per_cand, "unit": "candidates/s", "cores": 1,
        "sample": "cloud seed 1234 (30k points), 2564 samples -> %d hand sets, %d candidates after filterGraspsWorkspace (bench.py's list is their "
                  "first 5000): ImageGenerator::createImages on all of them %.2f s, Classifier::classifyImages on the first %d %.2f s; "
                  "generateGraspCandidates %.2f s and the filter %.3f s are not counted (as in `value`)" % (len(hands), n_valid, t_img, n_cls, t_cls, t_search, t_filter),
        "ms_per_candidate": {"images": t_img / n_valid * 1e3, "classify": t_cls / n_cls * 1e3},
        "search_s": t_search,
        "host": {"machine": platform.machine(), "cpu_count": os.cpu_count(), "where": "build container (not the GPU box)"},
        "caveat": "the shim's Eigen / OpenCV / PCL subsets are plain loops: the reference built on the real libraries is faster per core; "
                  "images verified byte for byte against the oracle on this very list",
        "scores_checksum": float(np.asarray(sc, np.float64).sum()),
    }
    with open(os.path.join(ROOT, "profiles", "r04_ref_cpu_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
