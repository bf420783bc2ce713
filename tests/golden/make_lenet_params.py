"""Packs the reference's released LeNet parameters into a small fixture.

Run in the build container (needs /root/reference):  python tests/golden/make_lenet_params.py
Source files: models/lenet/{15,3}channels/params/{conv1,conv2,ip2}_{weights,biases}.bin and
ip1_biases.bin — raw little-endian float32 exactly as EigenClassifier reads them
(net/eigen_classifier.cpp:28-50, 185-204).  ip1_weights.bin (500x7200) is not in the
reference snapshot (.MISSING_LARGE_BLOBS) and is synthesised by gpd_amd.synth.lenet_weights.
"""
import os
import sys

import numpy as np

REF = "/root/reference/models/lenet"
OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = dict(c1w="conv1_weights", c1b="conv1_biases", c2w="conv2_weights", c2b="conv2_biases", f1b="ip1_biases",
             f2w="ip2_weights", f2b="ip2_biases")

for ch in (15, 3):
    d = os.path.join(REF, "%dchannels" % ch, "params")
    if not os.path.isdir(d):
        sys.exit("reference not present: " + d)
    arrs = {k: np.fromfile(os.path.join(d, v + ".bin"), "<f4") for k, v in NAMES.items()}
    np.savez_compressed(os.path.join(OUT, "lenet%d_params.npz" % ch), **arrs)
    print(ch, {k: a.shape for k, a in arrs.items()})
