"""Writes tests/golden/ref_pin_*.npz: what the REFERENCE's own code returns on the cases of tests/ref_cases.py.

Needs oracle/_ref/libgpd_ref.so (oracle/build_ref.sh A: the reference's translation units, unmodified, through the
test-only third-party subsets of oracle/shim) and therefore the reference tree — it runs in the build container only.
The files it writes are committed and travel to the GPU box; tests/test_ref_pin.py compares the oracle (CPU suite) and the
HIP path (GPU suite) with them.  Nothing in here calls the oracle: the pins are the reference's outputs alone.

    python tests/golden/make_ref_pins.py            # all
    python tests/golden/make_ref_pins.py variants   # or: extras
    python tests/golden/make_ref_pins.py variants twelve_cameras   # these cases only
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_cases as rcs  # noqa: E402
from gpd_amd import synth  # noqa: E402
from oracle import ref  # noqa: E402
from oracle.oracle import default_params  # noqa: E402  (the parameter block only: a ctypes struct, no oracle arithmetic)

REFERENCE = os.environ.get("REF", "/root/reference")


def _stamp():
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
    except Exception:
        rev = "unknown"
    return np.array("reference sources at %s through oracle/shim (oracle/build_ref.sh A), repository %s" % (REFERENCE, rev))


def variants(only=None):
    for name in rcs.VARIANTS:
        if only and name not in only:
            continue
        p, cl, si, cam, vp = rcs.case_inputs(name, default_params)
        C = p.image_num_channels
        det = ref.Detector(p, weights=rcs.weights(15) if C == 15 else None)
        rc = ref.Cloud(cl["xyz"], cl["normals"], cam, vp)
        rc.set_sample_indices(si)
        hands = det.generate(rc, len(si))
        valid_f = det.filter_workspace()
        img, cand = det.images(rc, int(valid_f.sum()) + 1)
        out = dict(hands=hands.view(np.uint8), valid_filtered=valid_f, cand=cand, digests=rcs.image_digests(img), made_by=_stamp())
        if name in rcs.FULL_IMAGES:
            out["images"] = img
        if C == 15:
            # EigenClassifier on the reference's own images; mode 1: the shim's products as k-ascending fma chains (pins
            # im2col / flatten / weight index order bit for bit), mode 2: long double accumulation (an order-free yardstick)
            for tag, tm in (("", False), ("_trained", True)):
                det.load_weights(rcs.weights(15, trained_magnitude=tm), "params" + tag)
                for mode, key in ((1, "scores_fma"), (2, "scores_ld"), (0, "scores_plain")):
                    ref.set_product_mode(mode)
                    out[key + tag] = det.classify(img)
            ref.set_product_mode(0)
        det.close()
        rc.close()
        np.savez_compressed(os.path.join(HERE, "ref_pin_%s.npz" % name), **out)
        print("%-28s %3d valid, %3d images%s" % (name, int(hands["valid"].sum()), len(img), ", scores %s" % out["scores_fma"][:3] if C == 15 else ""))


def extras():
    out = dict(made_by=_stamp())
    # --- Cloud::voxelizeCloud on the tutorial clouds (4467 -> 3366, 104444 -> 35788) and normals after it
    for nm in ("krylon", "table_mug"):
        rc = ref.Cloud(pcd=os.path.join(REFERENCE, "tutorials", nm + ".pcd"))
        n_in = rc.size()
        rc.voxelize(0.003)
        vox, _ = rc.get()
        out[nm + "_n_in"] = np.array(n_in)
        out[nm + "_vox_count"] = np.array(len(vox))
        out[nm + "_vox_digest"] = rcs.digest(vox)
        out[nm + "_vox_head"] = vox[:16].copy()
        if nm == "krylon":
            rc.calculate_normals(0.03)
            _, nrm = rc.get()
            out["krylon_normals"] = nrm.astype(np.float32)
        rc.close()
    # --- workspace cut + voxelise + normals on a synthetic raw scan, two cameras with overlapping visibility
    cl = synth.make_cloud(77, 6000)
    cam, vp = rcs._cams(2, len(cl["xyz"]), seed=5)
    rc = ref.Cloud(cl["xyz"], None, cam, vp)
    ws = np.array([-0.2, 0.25, -0.25, 0.2, -1.0, 1.0])
    rc.filter_workspace(ws)
    cut, _ = rc.get()
    rc.calculate_normals(0.03)
    _, nrm = rc.get()
    out["cut_workspace"] = ws
    out["cut_count"] = np.array(len(cut))
    out["cut_digest"] = rcs.digest(cut)
    out["normals_two_cameras"] = nrm.astype(np.float32)
    rc.close()
    # --- the default case's candidates for select / cluster / re-evaluation
    p, cl, si, cam, vp = rcs.case_inputs("default_c15", default_params)
    w = rcs.weights(15)
    det = ref.Detector(p, weights=w, num_selected=25)
    rc = ref.Cloud(cl["xyz"], cl["normals"], cam, vp)
    rc.set_sample_indices(synth.sample_indices(cl, 120))
    hands = det.generate(rc, 120)
    valid_f = det.filter_workspace()
    img, cand = det.images(rc, int(valid_f.sum()) + 1)
    ref.set_product_mode(1)
    scores = det.classify(img)
    ref.set_product_mode(0)
    flat = hands.reshape(-1)[cand]
    out["sel_hands"] = flat.view(np.uint8)
    out["sel_scores"] = scores
    for i, sc in enumerate((scores, np.round(scores / 200).astype(np.float32), np.zeros(50, np.float32))):
        out["select_in_%d" % i] = sc
        out["select_out_%d" % i] = det.select(sc)
    for rm in (0, 1):
        for mi in (1, 3):
            c, cs = det.find_clusters(flat, scores.astype(np.float64), mi, bool(rm))
            out["clusters_%d_%d_pos" % (rm, mi)] = c["position"]
            out["clusters_%d_%d_score" % (rm, mi)] = cs
            out["clusters_%d_%d_full" % (rm, mi)] = c["full_antipodal"]
    gt = synth.make_cloud(4243, 8000)
    rc_gt = ref.Cloud(gt["xyz"], gt["normals"], gt["cam_source"], gt["view_points"])
    for tag, c in (("other", rc_gt), ("same", rc)):
        lab, hh = det.reevaluate(c, flat)
        out["reeval_%s_labels" % tag] = lab
        out["reeval_%s_half" % tag] = hh["half_antipodal"]
        out["reeval_%s_full" % tag] = hh["full_antipodal"]
    # --- samples given by coordinates (Cloud::setSamples, frame_estimator.cpp:38-65)
    sm = cl["xyz"][si].astype(np.float64) + np.random.default_rng(3).normal(0, 0.002, (len(si), 3))
    rc.set_samples(sm)
    out["xyz_samples"] = sm
    out["xyz_hands"] = det.generate(rc, len(sm)).view(np.uint8)
    det.close()
    rc.close()
    rc_gt.close()
    # --- ConvLayer::forward alone (conv_layer.cpp:26-98) on a small random layer, plain products
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256, (3, 9, 8)).astype(np.float32)
    cw = rng.normal(0, 0.1, (4, 3, 5, 5)).astype(np.float32)
    cb = rng.normal(0, 0.1, 4).astype(np.float32)
    ref.set_product_mode(1)
    out["conv_x"], out["conv_w"], out["conv_b"], out["conv_y"] = x, cw, cb, ref.conv_forward(x, cw, cb)
    ref.set_product_mode(0)
    # --- configs[0]: tutorials/krylon.pcd through the reference's own preprocessing and detectGrasps with the values of
    #     cfg/eigen_params.cfg (voxelize 0.003, workspace +-1, 8 orientations, axis 2, 10 placements), 500 seeded samples
    for tag, min_inliers, nsel in (("krylon_e2e", 0, 50), ("krylon_e2e_clustered", 1, 200)):
        p = default_params(15)
        det = ref.Detector(p, weights=rcs.weights(15), num_selected=nsel, min_inliers=min_inliers)
        rc = ref.Cloud(pcd=os.path.join(REFERENCE, "tutorials", "krylon.pcd"))
        rc.filter_workspace([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0])
        rc.voxelize(0.003)
        rc.calculate_normals(0.03)
        n = rc.size()
        samples = np.random.RandomState(3456).permutation(n)[:500].astype(np.int32)
        rc.set_sample_indices(samples)
        ref.reset_shadow_seed()
        ref.set_product_mode(1)
        final = det.detect(rc, 4096)
        ref.set_product_mode(0)
        out[tag + "_samples"] = samples
        out[tag + "_hands"] = final.view(np.uint8)
        print(tag, len(final), "grasps, best scores", final["score"][:3])
        det.close()
        rc.close()
    # --- detectGrasps with the approach-direction filter and clustering (grasp_detector.cpp:247-255, 283-303, 422-453)
    cl = synth.make_cloud(99, 12000)
    si = synth.sample_indices(cl, 300)
    p = default_params(15)
    det = ref.Detector(p, weights=rcs.weights(15, trained_magnitude=True), num_selected=120, min_inliers=1, filter_approach_direction=1,
                       direction=(0.0, 0.0, -1.0), thresh_rad=1.2)
    rc = ref.Cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    rc.set_sample_indices(si)
    ref.reset_shadow_seed()
    ref.set_product_mode(1)
    final = det.detect(rc, 4096)
    ref.set_product_mode(0)
    out["dirfilter_hands"] = final.view(np.uint8)
    print("dirfilter_e2e", len(final), "grasps")
    det.close()
    rc.close()
    np.savez_compressed(os.path.join(HERE, "ref_pin_extras.npz"), **out)
    print("extras:", sorted(out))


if __name__ == "__main__":
    if not ref.available() and not ref.build():
        sys.exit("oracle/_ref/libgpd_ref.so cannot be built here (no reference tree): the committed pins stay as they are")
    what = sys.argv[1:] or ["variants", "extras"]
    if "variants" in what:
        variants(only=[w for w in what if w in rcs.VARIANTS])
    if "extras" in what:
        extras()
