"""The split LeNet path's operand tables and index arithmetic, replayed on the CPU (no GPU, no oracle).

gpd_hip_lenet_fast_tables (host only) hands out the tables exactly as gpd_hip_set_lenet_weights uploads them; the functions
below restate what conv1_i8_kernel / conv2_bf16_kernel (gpd_amd/csrc/lenet_fast.hip) do with them lane by lane — LDS
layouts, the k-slot tables, the MFMA fragment layouts (cdna_hip_programming.md §3: A / B lane (row or column l & 15, k group
l >> 4), D lane (column l & 15, rows 4 (l >> 4) + r)), the pool over a lane quad / a lane's registers — and compare with a
plain convolution in numpy.  What it cannot check is the hardware's side of those layouts; the GPU tests do that.
"""
import ctypes as C
import os

import numpy as np
import pytest

from gpd_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables(channels, w):
    L = api.lib()
    L.gpd_hip_lenet_fast_tables.argtypes = [C.c_int] + [C.c_void_p] * 6
    atab = np.zeros((7, 5, 64, 16), np.int8)
    corr = np.zeros(20, np.float64)
    shift = np.zeros(20, np.int32)
    btab = np.zeros((4, 3, 16, 64, 8), np.uint16)  # [slot][piece][k-step][lane][8]
    c1w = np.ascontiguousarray(w["c1w"], np.float32)
    c2w = np.ascontiguousarray(w["c2w"], np.float32)
    rc = L.gpd_hip_lenet_fast_tables(channels, c1w.ctypes.data, c2w.ctypes.data, atab.ctypes.data, corr.ctypes.data, shift.ctypes.data,
                                     btab.ctypes.data)
    assert rc == 0
    return atab, corr, shift, btab


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _bf16_rne(x):
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7FFF + ((b >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _split3(x):
    x = np.ascontiguousarray(x, np.float32)
    h = _bf16_rne(x)
    r1 = (x - _bf16_to_f32(h)).astype(np.float32)
    m = _bf16_rne(r1)
    r2 = (r1 - _bf16_to_f32(m)).astype(np.float32)
    return h, m, _bf16_rne(r2)


def _mfma(a, b):
    """D[i][j] = sum over lane groups g and the bytes / elements e of A[lane 16 g + i][e] * B[lane 16 g + j][e];
    returned per lane: acc[lane][r] = D[4 (lane >> 4) + r][lane & 15]."""
    a4 = a.reshape(4, 16, -1).astype(np.float64)
    b4 = b.reshape(4, 16, -1).astype(np.float64)
    d = np.einsum("gie,gje->ij", a4, b4)
    lanes = np.arange(64)
    return np.stack([d[4 * (lanes >> 4) + r, lanes & 15] for r in range(4)], 1)


def _conv_valid(x, w):
    """x [C, H, W], w [F, C, 5, 5] -> [F, H - 4, W - 4] in float64 / int64"""
    F = w.shape[0]
    H, W = x.shape[1] - 4, x.shape[2] - 4
    out = np.zeros((F, H, W), x.dtype if x.dtype == np.int64 else np.float64)
    for ky in range(5):
        for kx in range(5):
            out += np.einsum("fc,chw->fhw", w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W])
    return out


def _pool(h):
    F, H, W = h.shape
    return h.reshape(F, H // 2, 2, W // 2, 2).max(axis=(2, 4))


@pytest.mark.parametrize("channels", [15, 12, 3, 1])
def test_conv1_tables_and_index_arithmetic(channels):
    rng = np.random.RandomState(5)
    w = synth.lenet_weights(channels, seed=7)
    # a filter with a wide dynamic range and one with a single huge weight: the fixed-point position is per filter
    c1w = w["c1w"].reshape(20, channels * 25).copy()
    c1w[3] *= np.exp(rng.uniform(-12, 0, c1w.shape[1])).astype(np.float32)
    c1w[7, 5] = 1e4
    c1w[11] = 0
    w["c1w"] = c1w.ravel()
    atab, corr, shift, _ = _tables(channels, w)
    # the digits reassemble the fixed-point weights, which are the weights rounded at 2^-shift
    K = channels * 25
    for f in range(20):
        mx = np.abs(c1w[f]).max()
        Wi = np.rint(c1w[f].astype(np.float64) * 2.0 ** int(shift[f])).astype(np.int64)
        assert np.abs(Wi).max() < 2 ** 30 and (mx == 0 or np.abs(Wi).max() >= 2 ** 28)
        assert corr[f] == 128.0 * Wi.sum()
    img = rng.randint(0, 256, (channels, 60, 60)).astype(np.uint8)
    img[:, 10:30, 5:50] = 0
    img[min(2, channels - 1)] = 255
    v = (img.astype(np.int16) - 128).astype(np.int8)  # == x ^ 0x80 as int8
    lanes = np.arange(64)
    j, q = lanes & 15, lanes >> 4
    m_row, m_x = (j >> 1) & 1, 2 * (j >> 2) + (j & 1)
    narrow = channels <= 4
    pack12 = channels == 12
    if narrow:
        # LDS image (round 6, C <= 4): four-byte pixels, 68 per row, FOUR copies, copy s shifted by s pixels; what lies beyond column 59
        # (and the padding bytes) is never initialised by the kernel: poison it, its weights must be zero
        P, COPY = 68, 60 * 68 * 4
        hwc = np.full((4 * COPY + 4096,), 77, np.int8)
        for sft in range(4):
            for c in range(channels):
                for y in range(60):
                    hwc[sft * COPY + (y * P + np.arange(60) + sft) * 4 + c] = v[c, y]
        n_shift = (4 - (m_x & 3)) & 3
        lane_off = n_shift * COPY + (m_row * P + m_x + n_shift) * 4

        def slot(ks, g):  # lenet_fast.hip f1n_slot
            sl = 4 * ks + g
            return 2 * sl if sl < 5 else (2 * (sl - 5) + 1 if sl < 10 else -1)
        n_slot = np.zeros((3, 64), np.int64)
        for ks in range(3):
            for g in range(4):
                sl = slot(ks, g)
                n_slot[ks, q == g] = 0 if sl < 0 else ((sl >> 1) * P + 4 * (sl & 1)) * 4
        assert (atab[3:] == 0).all()
        n_ks = 3
    elif pack12:
        # LDS image (round 6, C = 12): twelve-byte pixels, 736 bytes per row, TWO copies, copy 1 shifted by 4 bytes; a k-step is a kernel
        # row: 5 taps x 12 channels = 60 bytes + 4 that must carry zero weights (poisoned here)
        ROWB, COPY = 736, 60 * 736
        hwc = np.full((2 * COPY + 4096,), 77, np.int8)
        for sft in range(2):
            for c in range(12):
                for y in range(60):
                    hwc[sft * COPY + y * ROWB + 12 * np.arange(60) + c + 4 * sft] = v[c, y]
        lane_off = (m_x & 1) * (COPY + 4) + m_row * ROWB + 12 * m_x + 16 * q
        assert (atab[5:] == 0).all()
        n_ks = 5
    else:
        # LDS image: pixel-major, 72 pixels per row, 16 bytes per pixel, x ^ 0x80
        P = 72
        hwc = np.zeros((60 * P * 16 + 4096,), np.int8)
        for c in range(channels):
            for y in range(60):
                hwc[(y * P + np.arange(60)) * 16 + c] = v[c, y]
        kyg = np.array([0, 2, 1, 3])[q]
        lane_off = (m_row * P + m_x) * 16
        n_ks = 7
    out = np.zeros((28, 28, 20), np.float32)
    bias = w["c1b"]
    for t in range(196):
        trow, tcol = divmod(t, 7)
        if narrow:
            base = ((2 * trow) * P + 8 * tcol) * 4 + lane_off
            assert (base % 16 == 0).all()  # every window is 16-byte aligned in the lane's copy
            addr = [base + n_slot[ks] for ks in range(3)]
        elif pack12:
            base = (2 * trow) * ROWB + 12 * (8 * tcol) + lane_off
            assert (base % 8 == 0).all()  # every fragment is two 8-byte aligned reads in the lane's copy
            addr = [base + ks * ROWB for ks in range(5)]
        else:
            base = ((2 * trow) * P + 8 * tcol) * 16 + lane_off
            pa, pb, pc = base + kyg * P * 16, base + 4 * P * 16 + q * 16, base + 4 * P * 16 + 64
            addr = [pa + ks * 16 for ks in range(5)] + [pb, pc]
        acc = np.zeros((5, 64, 4))
        for ks in range(n_ks):
            B = np.stack([hwc[a:a + 16] for a in addr[ks]])
            for mt in range(5):
                acc[mt] += _mfma(atab[ks, mt], B)  # weights are the A operand (rows), pixels the B operand (columns)
        # epilogue: digits -> exact sum, max over the lane quad, + corr, one rounding, 2^-shift, + bias
        for mt in range(5):
            a = acc[mt].astype(np.int64)
            assert np.abs(a).max() < 2 ** 24
            hi, lo = a[:, 3] * 256 + a[:, 2], a[:, 1] * 256 + a[:, 0]
            assert np.abs(hi).max() < 2 ** 31 and np.abs(lo).max() < 2 ** 31
            s = hi * 65536 + lo
            s = s.reshape(16, 4).max(axis=1).repeat(4)  # quads of consecutive lanes
            f = 4 * mt + q
            val = np.ldexp((s + corr[f]).astype(np.float32), -shift[f]).astype(np.float32) + bias[f]
            p, wdw = j & 3, j >> 2
            keep = (p == mt) if mt < 4 else (p == 0)
            out[trow, 4 * tcol + wdw[keep], f[keep]] = val[keep]
    # plain integer convolution with the fixed-point weights
    ref = np.zeros((20, 28, 28), np.float32)
    x64 = img.astype(np.int64)
    for f in range(20):
        Wi = np.rint(c1w[f].astype(np.float64) * 2.0 ** int(shift[f])).astype(np.int64).reshape(1, channels, 5, 5)
        h = _pool(_conv_valid(x64, Wi))[0]
        ref[f] = np.ldexp(h.astype(np.float32), -int(shift[f])).astype(np.float32) + bias[f]
    assert np.array_equal(out, np.transpose(ref, (1, 2, 0)))
    # and that is the convolution with the float weights to within the fixed-point step
    exact = _pool(_conv_valid(img.astype(np.float64), c1w.reshape(20, channels, 5, 5).astype(np.float64))) + bias[:, None, None]
    tol = (np.abs(c1w).max(axis=1) * 2.0 ** -30 * K * 255 + 1e-30)[:, None, None] + np.abs(exact) * 2.0 ** -22
    assert (np.abs(np.transpose(out, (2, 0, 1)) - exact) <= tol).all()


def test_conv2_tables_and_index_arithmetic():
    rng = np.random.RandomState(9)
    w = synth.lenet_weights(15, seed=3)
    _, _, _, btab = _tables(15, w)
    c2w = w["c2w"].reshape(50, 20, 5, 5)
    pool1 = (rng.randn(28, 28, 20) * 30).astype(np.float32)  # [row][column][channel], conv1's output layout
    # LDS: [row][piece][column][20 channels] bf16
    PP, RS = 28 * 40, 3 * 28 * 40
    lds = np.zeros(28 * RS // 2 + 4096, np.uint16)
    pieces = _split3(pool1)
    for pc in range(3):
        for r in range(28):
            for c in range(28):
                o = (r * RS + pc * PP + c * 40) // 2
                lds[o:o + 20] = pieces[pc][r, c]
    lanes = np.arange(64)
    j, q = lanes & 15, lanes >> 4
    m_row, m_x = (j >> 1) & 1, 2 * (j >> 2) + (j & 1)
    lane_off = m_row * RS + m_x * 40
    off_x, off_y = lane_off + 40 * q, lane_off + q * RS + 160
    off_z, off_w = lane_off + 4 * RS + 160 + 8 * q, lane_off + 4 * RS + 160 + 32 + 0 * q
    flat = np.full((144, 50), np.nan, np.float64)
    terms = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
    # the seven waves of the full filter groups (round 6 roles: waves 0-3 group 1 + wave / 2, a half each — group 2 cut 20 : 16 —;
    # waves 4-6 group 0, a third each): every (group, tile) pair exactly once
    roles = [(1, 0, 18), (1, 18, 18), (2, 0, 20), (2, 20, 16)] + [(0, 12 * (w - 4), 12) for w in range(4, 7)]
    seen = set()
    for grp, t0, tcnt in roles:
        for tt in range(tcnt):
            T = t0 + tt
            assert (grp, T) not in seen
            seen.add((grp, T))
            rp, xt = divmod(T, 3)
            base = (2 * rp) * RS + (8 * xt) * 40
            acc = np.zeros((64, 4))
            for ks in range(16):
                frag = []
                for pc in range(3):
                    halves = []
                    for h in range(2):
                        e = ks + 16 * h
                        a = (base + off_x + (e // 5) * RS + (e % 5) * 8) if e < 25 else (base + off_y + (e - 25) * 8) if e < 30 else \
                            (base + off_z) if e == 30 else (base + off_w)
                        a = (a + pc * PP) // 2
                        halves.append(np.stack([lds[x:x + 4] for x in a]))
                    frag.append(_bf16_to_f32(np.concatenate(halves, 1)))
                for pa, pw in terms:
                    acc += _mfma(frag[pa], _bf16_to_f32(btab[grp, pw, ks]))  # pixels are the A operand, filters B
            flat[rp * 12 + 4 * xt + q, 16 * grp + j] = acc.max(axis=1)
    assert len(seen) == 3 * 36
    # the eighth wave: filters 48 and 49 as (filter, kernel column) rows x 16 input columns of a conv row, the five shifted
    # rows added across lanes — the lane arithmetic of conv2_bf16_kernel's second role
    for unit in range(24):
        yp, xh = unit >> 1, unit & 1
        ubase = (2 * yp) * RS + (12 * xh) * 40
        conv_row = []
        for rr in range(2):
            acc = np.zeros((64, 4))
            for ks in range(4):
                idx = np.minimum(4 * ks + q, 14)
                ky, c8 = idx // 3, idx % 3
                lo = ky * RS + j * 40 + c8 * 16
                hi = np.where(c8 == 2, lo, lo + 8)
                frag = []
                for pc in range(3):
                    a = ubase + rr * RS + pc * PP
                    frag.append(_bf16_to_f32(np.concatenate([np.stack([lds[(a + x) // 2:(a + x) // 2 + 4] for x in lo]),
                                                             np.stack([lds[(a + x) // 2:(a + x) // 2 + 4] for x in hi])], 1)))
                for pa, pw in terms:
                    acc += _mfma(_bf16_to_f32(btab[3, pw, ks]), frag[pa])  # weights are the A operand here
            shl = lambda v, n: np.where((lanes & 15) + n < 16, v[np.minimum(lanes + n, 63)], 0.0)  # DPP row_shl:n, bound_ctrl
            v = ((acc[:, 0] + shl(acc[:, 1], 1)) + shl(acc[:, 2], 2)) + shl(acc[:, 3], 3)
            e4 = shl(acc[:, 0], 4)
            conv_row.append(v + e4[(lanes + 32) & 63])  # lanes 0..31 <- lanes 32..63 (the upper lanes' sums are not used)
        pv = np.maximum(conv_row[0], conv_row[1])
        pv = np.maximum(pv, shl(pv, 1))
        ok = (q < 2) & (j % 2 == 0) & (j < 12)
        flat[(yp * 12 + 6 * xh + (j >> 1))[ok], (48 + (q & 1))[ok]] = pv[ok]
    assert not np.isnan(flat).any()
    # reference: the same six piece products, summed in float64
    wp = [_bf16_to_f32(p).astype(np.float64) for p in _split3(c2w)]
    xp = [np.transpose(_bf16_to_f32(p).astype(np.float64), (2, 0, 1)) for p in pieces]
    ref = sum(_conv_valid(xp[pa], wp[pw]) for pa, pw in terms)
    ref = np.transpose(_pool(ref), (1, 2, 0)).reshape(144, 50)
    assert np.allclose(flat, ref, rtol=1e-12, atol=1e-9)
    # and the six products are the f32 product to 2^-23
    full = np.transpose(_pool(_conv_valid(np.transpose(pool1, (2, 0, 1)).astype(np.float64), c2w.astype(np.float64))), (1, 2, 0)).reshape(144, 50)
    scale = np.abs(full).max()
    assert np.abs(flat - full).max() <= scale * 2.0 ** -20


def test_pieces_reassemble_exactly():
    """h + m + l == a for every f32 whose pieces do not underflow: three round-to-nearest bf16 pieces hold 24 bits."""
    rng = np.random.RandomState(2)
    a = (rng.randn(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    h, m, l = _split3(a)
    s = _bf16_to_f32(h).astype(np.float64) + _bf16_to_f32(m).astype(np.float64) + _bf16_to_f32(l).astype(np.float64)
    assert np.array_equal(s.astype(np.float32), a) and np.abs(s - a.astype(np.float64)).max() == 0
