#!/usr/bin/env python
"""How much of "third-party numerics unpinned" can matter: the reference's own translation units (oracle/_ref, tier A) re-run
on the 20 pinned cases with ONE boundary of the third-party subsets perturbed at a time, against the committed pins.

The pins (tests/golden/ref_pin_*.npz) were produced through oracle/shim's restatements of FLANN / Eigen / OpenCV.  Where
the real libraries may legitimately differ from those restatements — tie order, a different backward-stable eigen-solver, a
different summation order, a different rounding of a half, one division instead of reciprocal-times-range — this script builds
the reference with exactly that difference (oracle/build_ref.sh A with EXTRA=-DGPD_SHIM_PERTURB=N into oracle/_ref/perturbN)
and counts the DISCRETE outputs that change: hand validity, finger placement, candidate lists, image bytes.  That is an
estimate of how likely the real binary is to agree bit for bit with the pins (and so with the oracle and the HIP path), and it
names which library behaviour has to be confirmed first on a machine that has the libraries (tier B).

CPU only; needs /root/reference (build container).   python profiles/thirdparty_sensitivity.py > profiles/r05_thirdparty_sensitivity.txt
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PERTURBATIONS = {
    1: "FLANN: neighbours at equal distance returned in the opposite order (hand_search.cpp:178, frame_estimator via KdTree::radiusSearch)",
    2: "Eigen: SelfAdjointEigenSolver's vectors from a cyclic Jacobi iteration instead of the tridiagonal QR (local_frame.cpp:17-21)",
    3: "Eigen: double-precision matrix products summed pairwise instead of sequentially (M = N N^T in local_frame.cpp:17, rotations)",
    4: "OpenCV: saturate_cast<uchar> rounding halves up (floor(x + 0.5)) instead of to even (image_strategy.cpp:116-119, 144-153)",
    5: "OpenCV: normalize's scale as (b - a) / (max - min) instead of (b - a) * (1 / (max - min)) (image_strategy.cpp:116-119)",
}


def worker():
    """runs inside a process whose GPD_REF_LIB points at one build; prints one JSON line per case"""
    import ref_cases as rcs
    from oracle import ref
    from oracle.oracle import HAND_DTYPE, default_params
    for name in sorted(rcs.VARIANTS):
        pin = rcs.load_pin(name)
        p, cl, si, cam, vp = rcs.case_inputs(name, default_params)
        det = ref.Detector(p, weights=None)
        rc = ref.Cloud(cl["xyz"], cl["normals"], cam, vp)
        rc.set_sample_indices(si)
        hands = det.generate(rc, len(si))
        valid_f = det.filter_workspace()
        img, cand = det.images(rc, int(valid_f.sum()) + 1)
        det.close()
        rc.close()
        rh = pin["hands"].view(HAND_DTYPE).reshape(hands.shape)
        both = rh["valid"].astype(bool) & hands["valid"].astype(bool)
        out = dict(case=name, slots=int(hands.size), valid_pin=int(rh["valid"].sum()),
                   validity_changed=int((rh["valid"] != hands["valid"]).sum()),
                   placement_changed=int((rh["finger_placement_index"][both] != hands["finger_placement_index"][both]).sum()),
                   filtered_changed=int((pin["valid_filtered"] != valid_f).sum()),
                   records_bitwise_changed=int(sum(hands.reshape(-1)[i].tobytes() != rh.reshape(-1)[i].tobytes() for i in np.flatnonzero(both.reshape(-1)))),
                   max_frame_delta=float(np.abs(hands["frame"][both] - rh["frame"][both]).max()) if both.any() else 0.0,
                   candidates_pin=int(len(pin["cand"])), candidates_same=bool(np.array_equal(cand, pin["cand"])))
        if out["candidates_same"]:
            dg = rcs.image_digests(img)
            out["images_changed"] = int((dg != pin["digests"]).any(axis=1).sum())
            if "images" in pin:
                diff = img != pin["images"]
                out["bytes_changed"] = int(diff.sum())
                out["bytes_total"] = int(diff.size)
                out["max_byte_delta"] = int(np.abs(img.astype(np.int16) - pin["images"].astype(np.int16)).max())
        else:
            out["images_changed"] = None
        print("@@" + json.dumps(out))
        sys.stdout.flush()
    # the same default case on a cloud WITHOUT the 3 mm lattice (every coordinate moved by up to 0.3 mm, as a real scan's are):
    # no pin exists for it — the raw outputs go to a file and main() compares each perturbed build with the unperturbed one
    dump = os.environ.get("GPD_SENS_DUMP")
    if dump:
        p, cl, si, cam, vp = rcs.case_inputs("default_c15", default_params)
        rng = np.random.RandomState(2025)
        xyz = (cl["xyz"] + rng.uniform(-3e-4, 3e-4, cl["xyz"].shape)).astype(np.float32)
        si = rcs.synth.sample_indices(cl, 120)
        det = ref.Detector(p, weights=None)
        rc = ref.Cloud(xyz, cl["normals"], cam, vp)
        rc.set_sample_indices(si)
        hands = det.generate(rc, len(si))
        valid_f = det.filter_workspace()
        img, cand = det.images(rc, int(valid_f.sum()) + 1)
        det.close()
        rc.close()
        np.savez(dump, hands=hands.view(np.uint8), shape=np.array(hands.shape), valid_f=valid_f, img=img, cand=cand)


def _compare_dumps(base, other):
    from oracle.oracle import HAND_DTYPE
    b, o = np.load(base), np.load(other)
    hb = b["hands"].view(HAND_DTYPE).reshape(tuple(b["shape"]))
    ho = o["hands"].view(HAND_DTYPE).reshape(tuple(o["shape"]))
    both = hb["valid"].astype(bool) & ho["valid"].astype(bool)
    same_c = np.array_equal(b["cand"], o["cand"])
    return dict(slots=int(hb.size), valid=int(hb["valid"].sum()), validity=int((hb["valid"] != ho["valid"]).sum()),
                placement=int((hb["finger_placement_index"][both] != ho["finger_placement_index"][both]).sum()),
                max_dframe=float(np.abs(hb["frame"][both] - ho["frame"][both]).max()) if both.any() else 0.0,
                cand_same=bool(same_c), images=int(len(b["cand"])),
                images_changed=int((b["img"] != o["img"]).reshape(len(b["img"]), -1).any(axis=1).sum()) if same_c else None,
                bytes_changed=int((b["img"] != o["img"]).sum()) if same_c else None)


def main():
    ref_root = os.environ.get("REF", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "src", "gpd")):
        sys.exit("needs the reference tree (build container)")
    print("# third-party sensitivity of the pinned cases (profiles/thirdparty_sensitivity.py; reference sources through oracle/shim, one boundary")
    print("# perturbed per row group; the unperturbed build reproduces every pin, first block)\n")
    totals = {}
    for n in [0] + sorted(PERTURBATIONS):
        out_dir = os.path.join(ROOT, "oracle", "_ref") if n == 0 else os.path.join(ROOT, "oracle", "_ref", "perturb%d" % n)
        env = dict(os.environ, OUT=out_dir, EXTRA="" if n == 0 else "-DGPD_SHIM_PERTURB=%d" % n)
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh"), "A"], env=env, stdout=subprocess.DEVNULL)
        env = dict(os.environ, GPD_REF_LIB=os.path.join(out_dir, "libgpd_ref.so"), OMP_NUM_THREADS="1", GPD_SENS_DUMP="/tmp/gpd_sens_%d.npz" % n)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, check=True)
        rows = [json.loads(l[2:]) for l in r.stdout.splitlines() if l.startswith("@@")]
        assert len(rows) == 20, (len(rows), r.stderr[-2000:])
        title = "unperturbed (must reproduce the pins)" if n == 0 else "perturbation %d — %s" % (n, PERTURBATIONS[n])
        print("## " + title)
        print("%-30s %6s %6s | %9s %9s %9s %9s %11s | %6s %8s %10s" % ("case", "slots", "valid", "validity", "placement", "filtered", "rec.bits", "max dframe",
                                                                     "cand", "images", "bytes"))
        t = dict(slots=0, valid=0, validity=0, placement=0, filtered=0, recbits=0, cand_cases=0, images=0, images_total=0, bytes=0, bytes_total=0)
        for o in rows:
            print("%-30s %6d %6d | %9d %9d %9d %9d %11.3g | %6s %8s %10s"
                  % (o["case"], o["slots"], o["valid_pin"], o["validity_changed"], o["placement_changed"], o["filtered_changed"],
                     o["records_bitwise_changed"], o["max_frame_delta"], "same" if o["candidates_same"] else "DIFF",
                     "-" if o["images_changed"] is None else "%d/%d" % (o["images_changed"], o["candidates_pin"]),
                     "%d" % o["bytes_changed"] if "bytes_changed" in o else "-"))
            t["slots"] += o["slots"]
            t["valid"] += o["valid_pin"]
            t["validity"] += o["validity_changed"]
            t["placement"] += o["placement_changed"]
            t["filtered"] += o["filtered_changed"]
            t["recbits"] += o["records_bitwise_changed"]
            t["cand_cases"] += 0 if o["candidates_same"] else 1
            if o["images_changed"] is not None:
                t["images"] += o["images_changed"]
                t["images_total"] += o["candidates_pin"]
            t["bytes"] += o.get("bytes_changed", 0)
            t["bytes_total"] += o.get("bytes_total", 0)
        totals[n] = t
        print("total: %d of %d slots change validity (%.2e), %d of %d valid hands change placement (%.2e), %d records differ in some bit; "
              "%d of 20 candidate lists differ; %d of %d images change (%.2e)%s\n"
              % (t["validity"], t["slots"], t["validity"] / t["slots"], t["placement"], t["valid"], t["placement"] / max(t["valid"], 1), t["recbits"],
                 t["cand_cases"], t["images"], t["images_total"], t["images"] / max(t["images_total"], 1),
                 ", %d of %d image bytes (%.2e)" % (t["bytes"], t["bytes_total"], t["bytes"] / t["bytes_total"]) if t["bytes_total"] else ""))
        if n == 0:
            assert t["validity"] == t["placement"] == t["recbits"] == t["images"] == t["cand_cases"] == 0, "the unperturbed build does not reproduce the pins"
    print("## the default case on a cloud without the lattice (coordinates moved by up to 0.3 mm; 120 samples): each perturbed build against the unperturbed one")
    for n in sorted(PERTURBATIONS):
        c = _compare_dumps("/tmp/gpd_sens_0.npz", "/tmp/gpd_sens_%d.npz" % n)
        print("%d  validity %d / %d slots  placement %d / %d valid  max dframe %.3g  candidates %s  images changed %s / %d  bytes changed %s"
              % (n, c["validity"], c["slots"], c["placement"], c["valid"], c["max_dframe"], "same" if c["cand_same"] else "DIFF", c["images_changed"],
                 c["images"], c["bytes_changed"]))
    print()
    print("## summary: fraction of discrete outputs that change per boundary (the 20 pinned cases: synthetic clouds on a 3 mm lattice)")
    for n in sorted(PERTURBATIONS):
        t = totals[n]
        print("%d  validity %.2e  placement %.2e  images %.2e   %s" % (n, t["validity"] / t["slots"], t["placement"] / max(t["valid"], 1),
                                                                      t["images"] / max(t["images_total"], 1), PERTURBATIONS[n]))


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker()
    else:
        main()
