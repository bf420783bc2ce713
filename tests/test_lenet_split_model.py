"""The NUMERICS of the split LeNet path, stated in numpy and held against the reference pins on the CPU (no GPU, no oracle):

  conv1   integer dot products of the u8 image with round(w * 2^s) (|.| < 2^30 per filter), one rounding, 2^-s, + bias, pool;
  conv2,  every f32 operand cut into three round-to-nearest bf16 pieces, the six piece products h*h + h*m + m*h + h*l + l*h + m*m
  ip1     summed exactly per block of 32 k (what one MFMA instruction adds up) and accumulated in f32 from block to block;
  ip2     two f32 fma chains.

gpd_amd/csrc/lenet_fast.hip computes this up to the order inside and between the blocks (tests/test_gpu_lenet_fast.py holds the
kernels against float64 and the pins on the device; tests/test_lenet_fast_tables.py their index arithmetic).  Here: the definition
itself is within BASELINE.json's 1e-4 of what the reference's own code returned with its plain float products, and closer to the
long-double yardstick than the k-ascending f32 chain is; five products instead of six are not (why there are six)."""
import os

import numpy as np
import pytest

import ref_cases as rcs


def _bf16_rne(x):
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def _split3(x):
    x = np.ascontiguousarray(x, np.float32)
    h = _bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = _bf16_rne(r1)
    return h, m, _bf16_rne((r1 - m).astype(np.float32))


def _im2col(x, k=5):
    C, H, W = x.shape
    oh, ow = H - k + 1, W - k + 1
    cols = np.empty((C * k * k, oh * ow), x.dtype)
    i = 0
    for c in range(C):
        for a in range(k):
            for b in range(k):
                cols[i] = x[c, a:a + oh, b:b + ow].reshape(-1)
                i += 1
    return cols


def _pool(h, side):
    F = h.shape[0]
    return h.reshape(F, side // 2, 2, side // 2, 2).max(axis=(2, 4)).reshape(F, -1)


def _gemm_split(A, B, terms=6):
    Ah, Am, Al = [a.astype(np.float64) for a in _split3(A)]
    Bh, Bm, Bl = [b.astype(np.float64) for b in _split3(B)]
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k0 in range(0, A.shape[1], 32):
        s = slice(k0, k0 + 32)
        parts = [Al[:, s] @ Bh[s], Ah[:, s] @ Bl[s], Am[:, s] @ Bm[s], Am[:, s] @ Bh[s], Ah[:, s] @ Bm[s], Ah[:, s] @ Bh[s]]
        for c in parts[6 - terms:]:
            acc = (acc.astype(np.float64) + c).astype(np.float32)
    return acc


def _forward(img, w, terms=6):
    x = np.transpose(img, (2, 0, 1)).astype(np.int64)
    W1 = w["c1w"].reshape(20, -1)
    cols = _im2col(x)
    h1 = np.empty((20, 56 * 56), np.float32)
    for f in range(20):
        s = 30 - int(np.frexp(float(np.abs(W1[f]).max()))[1])
        Wi = np.rint(W1[f].astype(np.float64) * 2.0 ** s).astype(np.int64)
        h1[f] = np.ldexp((Wi @ cols).astype(np.float32), -s)  # exact integer sum, ONE rounding
    p1 = (_pool(h1, 56) + w["c1b"][:, None]).astype(np.float32)
    h2 = _gemm_split(w["c2w"].reshape(50, 500), _im2col(p1.reshape(20, 28, 28)), terms)
    p2 = (_pool(h2, 24) + w["c2b"][:, None]).astype(np.float32)
    y = _gemm_split(p2.T.reshape(1, -1), w["f1w"].reshape(7200, 500), terms)
    y = np.maximum((y[0] + w["f1b"]).astype(np.float32), 0)
    o = []
    for which in (0, 1):
        acc = np.float32(0)
        ww = w["f2w"][which::2]
        for j in range(500):
            acc = np.float32(np.float64(ww[j]) * np.float64(y[j]) + np.float64(acc))
        o.append(np.float32(acc + w["f2b"][which]))
    return np.float32(o[1] - o[0])


def test_split_definition_against_the_reference_pins():
    pin = rcs.load_pin("default_c15")
    assert pin is not None and "images" in pin
    w = rcs.weights(15, trained_magnitude=True)
    n = 12
    six = np.array([_forward(pin["images"][i], w) for i in range(n)])
    plain, ld, chain = pin["scores_plain_trained"][:n], pin["scores_ld_trained"][:n], pin["scores_fma_trained"][:n]
    assert np.abs(six - plain).max() <= 1e-4 and np.abs(six - ld).max() <= 1e-4
    e_split, e_chain = float(np.abs(six - ld).max()), float(np.abs(chain - ld).max())
    print("12 pin images: max |split - long double| = %.3g, |f32 chain - long double| = %.3g" % (e_split, e_chain))
    assert e_split <= e_chain
    five = np.array([_forward(pin["images"][i], w, terms=5) for i in range(4)])  # without the smallest kept product (l*h)
    assert np.abs(five - ld[:4]).max() > 3 * e_split
