#!/usr/bin/env python
"""BASELINE.json: "LeNet scores within 1e-4 of the Eigen path".  The reference's OWN classifier code (EigenClassifier / ConvLayer /
DenseLayer compiled from /root/reference through oracle/_ref, plain float products summed in ascending k, the same in long double)
against the DEFINITION of the library's default scoring mode (tests/test_lenet_split_model.py: int8 digit planes for conv1, three
bf16 pieces and six piece products for conv2 / ip1 — what gpd_amd/csrc/lenet_fast.hip computes up to the order inside a block of
32 k; tests/test_gpu_lenet_fast.py holds the kernels against this model on the device) and against the oracle's f32 fma chain (the
library's other mode, bit for bit), on grasp images of random clouds — more of them than the 20 pinned cases hold.

CPU only, build container (needs oracle/_ref).      python profiles/split_vs_reference_scores.py [IMAGES] > profiles/r05_split_vs_reference_scores.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle
    import ref_cases as rcs
    import test_lenet_split_model as model
    from gpd_amd import synth
    from oracle import ref
    if not ref.available():
        sys.exit("needs oracle/_ref (build container)")
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    w = rcs.weights(15, trained_magnitude=True)
    p = oracle.default_params(15)
    det = ref.Detector(p, weights=w)
    rows, done, seed = [], 0, 0
    all_split, all_chain, all_plain, all_ld = [], [], [], []
    while done < want:
        cl = synth.make_cloud(7000 + seed, 8000 + 500 * (seed % 9), clutter=bool(seed % 2))
        si = synth.sample_indices(cl, 40, seed=seed)
        hands = oracle.filter_workspace(p, oracle.search(p, cl["xyz"], cl["normals"], si))
        img, _ = oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
        img = img[: min(len(img), want - done, 150)]
        ref.set_product_mode(0)
        plain = det.classify(img)
        ref.set_product_mode(2)
        ld = det.classify(img)
        ref.set_product_mode(0)
        chain = oracle.lenet(img, w)
        split = np.array([model._forward(im, w) for im in img])
        all_split.append(split)
        all_chain.append(chain)
        all_plain.append(plain)
        all_ld.append(ld)
        rows.append((seed, len(cl["xyz"]), len(img), float(np.abs(plain).max()), float(np.abs(split - plain).max()), float(np.abs(chain - plain).max()),
                     float(np.abs(split - ld).max()), float(np.abs(chain - ld).max()), float(np.abs(plain - ld).max())))
        done += len(img)
        seed += 1
    det.close()
    split, chain, plain, ld = [np.concatenate(a) for a in (all_split, all_chain, all_plain, all_ld)]
    print("# scores of the default scoring mode's DEFINITION (split operands) and of the f32 fma chain against the reference's own classifier code")
    print("# (oracle/_ref: plain float products / long double), %d grasp images of %d random clouds, trained-magnitude weights" % (len(split), len(rows)))
    print("# (profiles/split_vs_reference_scores.py)\n")
    print("%5s %7s %7s %9s | %13s %13s | %13s %13s %13s" % ("cloud", "points", "images", "max|score|", "|split-plain|", "|chain-plain|", "|split-ld|", "|chain-ld|", "|plain-ld|"))
    for r in rows[:10]:
        print("%5d %7d %7d %9.2f | %13.3g %13.3g | %13.3g %13.3g %13.3g" % r)
    print("  ... (%d clouds)" % len(rows))

    def stats(name, e):
        e = np.abs(e)
        print("%-44s max %.3g   99.9th %.3g   99th %.3g   median %.3g" % (name, e.max(), np.percentile(e, 99.9), np.percentile(e, 99), np.median(e)))
    print()
    stats("|split definition - reference plain float|", split - plain)
    stats("|f32 fma chain    - reference plain float|", chain - plain)
    stats("|split definition - reference long double|", split - ld)
    stats("|f32 fma chain    - reference long double|", chain - ld)
    stats("|reference plain  - reference long double|", plain - ld)
    print("\nmax |score| %.2f; BASELINE's tolerance 1e-4: split %s, chain %s" % (np.abs(plain).max(), "met" if np.abs(split - plain).max() <= 1e-4 else "NOT met",
                                                                                    "met" if np.abs(chain - plain).max() <= 1e-4 else "NOT met"))
    print("sign agreement with the reference (score > 0): split %d / %d, chain %d / %d"
          % (int(((split > 0) == (plain > 0)).sum()), len(plain), int(((chain > 0) == (plain > 0)).sum()), len(plain)))


if __name__ == "__main__":
    main()
