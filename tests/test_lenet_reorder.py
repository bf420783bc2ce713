"""How much do the LeNet scores move when the float32 summation order changes?

The reference's EigenClassifier sums its dot products in whatever order Eigen's GEMM / GEMV kernels use
(conv_layer.cpp:54, dense_layer.cpp:12) — unknown here (Eigen is absent) and build-dependent.  The oracle and the
HIP kernels define ONE order (k-ascending fmaf chains).  "Scores within 1e-4 of the reference" can therefore only
be argued through the sensitivity of the scores to the order: three other float32 orders — torch's conv2d /
linear (MKL-DNN blocking), a K-blocked numpy GEMM (BLAS order inside 64-wide K blocks, blocks added in order) and
a plain k-DESCENDING multiply-add chain (no FMA) — are compared with the oracle on real grasp images,
unnormalised 0..255 inputs and all.  The float64 result is the yardstick.

Finding (recorded in DESIGN.md §2): with the SYNTHETIC ip1 of this repository (N(0, 0.005^2): the reference's
trained ip1_weights.bin is not in the snapshot) the logits are not O(10) as for a trained net but reach |score| ~ 1000,
where one float32 ulp is 6e-5: no two float32 summation orders agree to 1e-4 there — the k-ascending chain (the
least accurate order: 7200 sequential adds in ip1) sits 2e-3 from float64, BLAS / blocked orders 1.4e-4 .. 4e-4.
What holds, and is asserted, is the RELATIVE statement |score - oracle| <= 8e-6 * max(12.5, |score|) (worst: the opposite, k-descending chain; BLAS-like orders 4.6e-6), i.e. the 1e-4 bar
for every score up to |12.5|; and with ip1 scaled so that the logits have the size a trained net produces (|score| < 20),
all orders are within 1e-4 absolute.  (ip1 + ip2 in float64 would bring the oracle itself to 2e-6 relative — the
convolutions' chains remain — which roughly doubles that range, at ~0.2 ms per 5000 images on the f64
matrix pipe; not done.)"""
import numpy as np
import pytest

from gpd_amd import synth


def _layers(w, C):
    W1 = w["c1w"].reshape(20, C * 25)
    W2 = w["c2w"].reshape(50, 500)
    F1 = w["f1w"].reshape(7200, 500)  # column-major 500 x 7200: W[u, j] = w[j * 500 + u]
    F2 = w["f2w"].reshape(500, 2)
    return W1, W2, F1, F2


def _im2col(x, k=5):  # x [N, C, H, W] -> [N, C*k*k, OH*OW], row order channel -> kh -> kw (conv_layer.cpp:71-73)
    N, Cc, H, Wd = x.shape
    OH, OW = H - k + 1, Wd - k + 1
    cols = np.empty((N, Cc, k, k, OH, OW), x.dtype)
    for kh in range(k):
        for kw in range(k):
            cols[:, :, kh, kw] = x[:, :, kh:kh + OH, kw:kw + OW]
    return cols.reshape(N, Cc * k * k, OH * OW)


def _pool(x):  # [N, F, H, W] max 2x2 stride 2
    return np.maximum(np.maximum(x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2]), np.maximum(x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]))


def _forward(images, w, matmul, dtype):
    """LeNet forward with `matmul(A [M,K], B [N,K,P]) -> [N,M,P]` as the only place sums are formed."""
    C = images.shape[3]
    W1, W2, F1, F2 = _layers({k: v.astype(dtype) for k, v in w.items()}, C)
    x = images.transpose(0, 3, 1, 2).astype(dtype)
    h = matmul(W1, _im2col(x)).reshape(-1, 20, 56, 56) + w["c1b"].astype(dtype)[None, :, None, None]
    h = _pool(h)
    h = matmul(W2, _im2col(h)).reshape(-1, 50, 24, 24) + w["c2b"].astype(dtype)[None, :, None, None]
    h = _pool(h)                                            # [N, 50, 12, 12]
    flat = h.transpose(0, 2, 3, 1).reshape(len(h), 7200)   # pixel-major, channel-minor (eigen_classifier.cpp:103-107)
    y = matmul(F1.T.copy(), flat[:, :, None])[:, :, 0] + w["f1b"].astype(dtype)
    y = np.maximum(y, 0)
    z = matmul(F2.T.copy(), y[:, :, None])[:, :, 0] + w["f2b"].astype(dtype)
    return (z[:, 1] - z[:, 0]).astype(np.float64)


def _mm_blas(A, B):
    return np.matmul(A[None], B)


def _mm_blocked(A, B, kb=64):
    out = np.zeros((B.shape[0], A.shape[0], B.shape[2]), A.dtype)
    for k0 in range(0, A.shape[1], kb):
        out = out + np.matmul(A[None, :, k0:k0 + kb], B[:, k0:k0 + kb])
    return out


def _mm_descending(A, B):
    out = np.zeros((B.shape[0], A.shape[0], B.shape[2]), A.dtype)
    for k in range(A.shape[1] - 1, -1, -1):  # separate multiply and add, highest k first
        out = out + A[None, :, k, None] * B[:, None, k, :]
    return out


@pytest.fixture(scope="module")
def grasp_images(oracle_mod, lenet15_real):
    cl = synth.make_cloud(77, 14000)
    p = oracle_mod.default_params(15)
    si = synth.sample_indices(cl, 30)
    hands = oracle_mod.filter_workspace(p, oracle_mod.search(p, cl["xyz"], cl["normals"], si))
    img, cand = oracle_mod.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands)
    assert len(cand) >= 40
    return img[:64]


def _deviations(oracle_mod, images, w):
    ref = oracle_mod.lenet(images, w).astype(np.float64)
    f64 = _forward(images, w, _mm_blas, np.float64)
    out = {"oracle (k-ascending fmaf chains) vs float64": (ref, f64)}
    orders = {"BLAS float32": _mm_blas, "K-blocked float32": _mm_blocked, "k-descending float32, no FMA": _mm_descending}
    for name, mm in orders.items():
        out[name + " vs oracle"] = (_forward(images, w, mm, np.float32), ref)
    import torch
    import torch.nn.functional as Fn
    with torch.no_grad():
        C = 15
        x = torch.from_numpy(images.transpose(0, 3, 1, 2).astype(np.float32))
        h = Fn.max_pool2d(Fn.conv2d(x, torch.from_numpy(w["c1w"].reshape(20, C, 5, 5)), torch.from_numpy(w["c1b"])), 2)
        h = Fn.max_pool2d(Fn.conv2d(h, torch.from_numpy(w["c2w"].reshape(50, 20, 5, 5)), torch.from_numpy(w["c2b"])), 2)
        flat = h.permute(0, 2, 3, 1).reshape(len(h), 7200)
        y = torch.relu(Fn.linear(flat, torch.from_numpy(w["f1w"].reshape(7200, 500).T.copy()), torch.from_numpy(w["f1b"])))
        z = Fn.linear(y, torch.from_numpy(w["f2w"].reshape(500, 2).T.copy()), torch.from_numpy(w["f2b"]))
        out["torch float32 conv2d/linear vs oracle"] = ((z[:, 1] - z[:, 0]).double().numpy(), ref)
    return ref, out


def _synthetic_magnitude(w):
    """the conftest fixture carries ip1 / 128 (trained-net magnitudes); the synthetic set is that times 128 (exact: a power of two)"""
    w = {k: v.copy() for k, v in w.items()}
    w["f1w"] = (w["f1w"] * np.float32(synth.TRAINED_IP1_DIVISOR)).astype(np.float32)
    return w


def test_scores_under_other_summation_orders(oracle_mod, lenet15_real, grasp_images):
    """The synthetic ip1 at its full magnitude (|score| ~ 1000): the relative bar."""
    ref, dev = _deviations(oracle_mod, grasp_images, _synthetic_magnitude(lenet15_real))
    print("\nscore range [%.3f, %.3f]" % (ref.min(), ref.max()))
    for k, (a, b) in dev.items():
        rel = np.abs(a - b) / np.maximum(12.5, np.abs(b))
        print("  %-52s max abs %.3g   max |d| / max(12.5, |score|) %.3g" % (k, np.abs(a - b).max(), rel.max()))
        assert rel.max() <= 8e-6, k  # = 1e-4 absolute up to |score| = 12.5
    assert np.abs(ref).max() > 100.0  # the synthetic ip1 really produces logits far beyond a trained net's


def test_scores_under_other_summation_orders_at_trained_net_magnitude(oracle_mod, lenet15_real, grasp_images):
    """ip1 scaled so that |score| stays below 20, as for a trained LeNet: every order within 1e-4 absolute."""
    ref, dev = _deviations(oracle_mod, grasp_images, lenet15_real)  # the fixture is the trained-magnitude set
    assert 1.0 < np.abs(ref).max() < 20.0, np.abs(ref).max()
    print("\nscore range [%.3f, %.3f]" % (ref.min(), ref.max()))
    for k, (a, b) in dev.items():
        print("  %-52s max abs %.3g" % (k, np.abs(a - b).max()))
        assert np.abs(a - b).max() <= 1e-4, k
