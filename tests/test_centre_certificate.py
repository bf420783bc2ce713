"""The centre of a sample's image neighbourhood (HandSet::calculateShadow, hand_set.cpp:131-133: points.rowwise().sum() / size)
is an fp64 sum of float coordinates.  The oracle sums sequentially in neighbour order, Eigen reduces in packets; the neighbourhood
kernel takes an order-free sum and certifies it (search.hip centre_exact): every addend is a multiple of 2^(emin - 150) and below
2^(emax - 126), so with ceil(log2 n) + emax - emin <= 28 every partial sum of every order is exact in fp64.  CPU: the certificate's
arithmetic as a property of numpy floats.  GPU: both routes against the oracle, the serial-chain fallback forced by points
micrometres from a coordinate plane."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth


def centre_exact(x):
    """the certificate of gpd_amd/csrc/search.hip centre_exact, on a float32 vector"""
    b = x.view(np.uint32)
    nz = (b & 0x7FFFFFFF) != 0
    if not nz.any():
        return True
    e = ((b[nz] >> 23) & 0xFF).astype(np.int64)
    e = np.maximum(e, 1)
    if e.max() >= 255:
        return False
    lg = 0
    while (1 << lg) < len(x):
        lg += 1
    return lg + int(e.max()) - int(e.min()) <= 28


def test_certified_sums_are_the_same_in_every_order():
    rng = np.random.RandomState(5)
    certified = refused = 0
    for case in range(400):
        n = int(rng.choice([1, 7, 300, 2600, 9000]))
        scale = 10.0 ** rng.uniform(-3, 1)
        x = (rng.uniform(-1, 1, n) * scale).astype(np.float32)
        if case % 3 == 0:  # a few addends far below the others: a certificate that must refuse at some point
            k = rng.randint(1, 4)
            x[rng.choice(n, min(k, n), replace=False)] = (rng.uniform(-1, 1, min(k, n)) * scale * 2.0 ** -rng.randint(5, 40)).astype(np.float32)
        if case % 7 == 0:
            x[rng.choice(n, max(n // 10, 1), replace=False)] = 0.0
        seq = 0.0
        for v in x:  # the oracle's chain
            seq = seq + float(v)
        if centre_exact(x):
            certified += 1
            from fractions import Fraction
            assert Fraction(seq) == sum(Fraction(float(v)) for v in x)  # the chain never rounded ...
            for _ in range(3):  # ... nor does any other order: pairwise tree over a random permutation, eight partial sums
                p = x[rng.permutation(n)].astype(np.float64)
                parts = [float(np.add.reduce(c)) for c in np.array_split(p, 8)]
                assert float(np.add.reduce(np.array(parts))) == seq
        else:
            refused += 1
    assert certified > 150 and refused > 30


def test_certificate_edges():
    assert centre_exact(np.zeros(5, np.float32))
    assert not centre_exact(np.array([1.0, np.inf], np.float32))
    assert not centre_exact(np.array([1.0, np.nan], np.float32))
    assert centre_exact(np.array([1e-40, 2e-40], np.float32))  # denormals alone: multiples of 2^-149
    assert not centre_exact(np.array([1.0, 1e-40], np.float32))
    x = np.full(4096, 0.5, np.float32)
    x[0] = 2.0 ** -17  # 12 + 126 - 110 = 28: still certified
    assert centre_exact(x)
    x[0] = 2.0 ** -18
    assert not centre_exact(x)


@pytest.mark.gpu
def test_centres_on_both_routes_match_the_oracle(oracle_mod):
    """Images (whose shadow channels hang on the centre through the shadow direction) and records byte for byte against the
    oracle: once on a cloud where every sum is certified (no serial chain runs), once with points micrometres and less from the
    coordinate planes inside the sampled neighbourhoods (the fallback must run, and give the sequential result)."""
    cl = synth.make_cloud(4242, 20000)
    xyz = cl["xyz"].copy()
    obj = np.flatnonzero(cl["is_object"])
    rng = np.random.RandomState(9)
    si = rng.choice(obj, 64, replace=False).astype(np.int32)
    w = synth.lenet_weights(15, real=dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lenet15_params.npz"))), trained_magnitude=True)
    p, op = api.default_params(15), oracle_mod.default_params(15)
    ctx = api.Context(p)
    try:
        ctx.set_lenet_weights(w)
        for dirty in (False, True):
            x = xyz.copy()
            if dirty:
                # move the cloud so that the sampled object straddles the x = 0 and y = 0 planes, then put neighbours of the samples
                # (lattice points a few steps away) at tiny / denormal / zero coordinates
                o = x[si[0]].copy()
                x[:, 0] -= o[0]
                x[:, 1] -= o[1]
                d = np.linalg.norm(x - x[si[0]], axis=1)
                near = np.flatnonzero((d < 0.05) & (d > 0.004))  # inside the first sample's 0.10 m neighbourhood (and its neighbours')
                assert len(near) > 20
                pick = rng.choice(near, 12, replace=False)
                x[pick[:4], 0] = np.float32([3e-9, -7e-12, 1e-40, 0.0])
                x[pick[4:8], 1] = np.float32([5e-10, -2e-13, -3e-41, 0.0])
                x[pick[8:], 2] += np.float32(1e-7)  # (z stays decimetres from its plane: certified)
            ctx.upload_cloud(x, cl["normals"], cl["cam_source"], cl["view_points"])
            hands, n_cand = ctx.detect(si)
            chains = ctx.centre_chains()
            ohands, on_cand, _ = oracle_mod.detect(op, x, cl["normals"], cl["cam_source"], cl["view_points"], si, w)
            assert n_cand == on_cand and n_cand > 30
            a, b = hands.copy(), ohands.copy()
            assert np.abs(a["score"] - b["score"]).max() <= 1e-4
            a["score"] = 0
            b["score"] = 0
            assert a.tobytes() == b.tobytes()
            fw = oracle_mod.filter_workspace(op, ohands.copy())
            img, cand = ctx.images(fw)
            oimg, ocand = oracle_mod.images(op, x, cl["normals"], cl["cam_source"], cl["view_points"], fw)
            assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
            if dirty:
                assert 0 < chains < 3 * len(si), chains
            else:
                assert chains == 0, chains
    finally:
        ctx.close()
