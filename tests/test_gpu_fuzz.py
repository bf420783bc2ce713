"""Differential fuzz of the HIP path against the oracle: random clouds (sizes, clutter, off-lattice
jitter, random normals, one to three cameras with random per-point visibility), random samples by
index and by coordinate, random parameter draws.  Records and images byte for byte, scores to 1e-4."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth


def _weights(C):
    g = os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    cl = synth.make_cloud(5000 + seed, int(rng.choice([3000, 9000, 20000])), clutter=bool(rng.randint(2)))
    xyz, nrm = cl["xyz"].copy(), cl["normals"].copy()
    if seed % 3 == 1:  # off the 3 mm lattice: no exact distance ties, arbitrary floats
        xyz += rng.uniform(-0.0012, 0.0012, xyz.shape).astype(np.float32)
    if seed % 4 == 2:  # noisy normals (not unit length, like a sloppy estimator's)
        nrm = (nrm + rng.normal(0, 0.2, nrm.shape)).astype(np.float32)
    n_cams = 1 + seed % 3
    cam = (rng.rand(n_cams, len(xyz)) < 0.8).astype(np.int32)
    cam[0] |= (cam.sum(0) == 0).astype(np.int32)  # every point seen by somebody
    vp = np.array([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1], [-0.25, 0.3, 0.05]])[:n_cams]
    kw = {}
    if seed % 2:
        kw["num_orientations"] = int(rng.choice([3, 5, 8]))
        kw["friction_coeff"] = float(rng.choice([10.0, 20.0, 30.0]))
        kw["min_viable"] = int(rng.choice([1, 6, 12]))
        kw["nn_radius_frames"] = float(rng.choice([0.008, 0.01, 0.015]))
    C = int(rng.choice([15, 15, 12, 3, 1]))
    return dict(xyz=xyz, normals=nrm, cam=cam, vp=vp, kw=kw, C=C, obj=cl["is_object"], rng=rng)


# GPD_FUZZ_DETECT / _WIDE / _GEOMETRY widen the seed ranges for a soak run (profiles/r04_soak.sh)
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("GPD_FUZZ_DETECT", "12"))))
def test_fuzz_detect_matches_oracle(oracle_mod, seed):
    c = _case(seed)
    C = c["C"]
    w = _weights(C)
    gp, op = api.default_params(C), oracle_mod.default_params(C)
    for k, v in c["kw"].items():
        setattr(gp, k, v)
        setattr(op, k, v)
    obj = np.flatnonzero(c["obj"])
    si = c["rng"].choice(obj, size=min(120, len(obj)), replace=False).astype(np.int32)
    ctx = api.Context(gp)
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(c["xyz"], c["normals"], c["cam"], c["vp"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(op, c["xyz"], c["normals"], c["cam"], c["vp"], si, w)
        assert hands.shape == ohands.shape and n_cand == on_cand
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        fw = oracle_mod.filter_workspace(op, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(op, c["xyz"], c["normals"], c["cam"], c["vp"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
        # the same samples by coordinate, nudged off the cloud
        sm = c["xyz"][si[:40]].astype(np.float64) + c["rng"].uniform(-0.003, 0.003, (40, 3))
        got = ctx.search_samples(sm)
        want = oracle_mod.search_xyz(op, c["xyz"], c["normals"], sm)
        assert got.shape == want.shape and got.tobytes() == want.tobytes()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 4, 7])
def test_fuzz_normals_match_oracle(oracle_mod, seed):
    """Cloud::calculateNormals on random clouds (off-lattice, several cameras): float32 normals bit for bit."""
    c = _case(seed)
    xyz = c["xyz"][:6000]
    cam = c["cam"][:, :6000]
    radius = [0.02, 0.03, 0.015][seed % 3]
    ctx = api.Context(api.default_params(15))
    try:
        ctx.upload_cloud(xyz, np.zeros_like(xyz), cam, c["vp"])
        got = ctx.estimate_normals(radius)
        want = oracle_mod.estimate_normals(xyz, cam, c["vp"], radius)
        assert got.shape == want.shape and got.tobytes() == want.tobytes()
        assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    finally:
        ctx.close()


def _geometry(seed, wide=False):
    """Every knob of the hand / image geometry and of the search at once, drawn as arbitrary doubles (the shipped cfg files
    hold round decimals): threshold tables, the correctly-rounded division by the box extents, the finger lookup table, the
    deepen steps and the voxel windows all depend on them."""
    rng = np.random.RandomState(77000 + seed)
    fw = rng.uniform(0.005, 0.02)
    od = rng.uniform(0.08, 0.14)
    kw = dict(finger_width=fw, hand_outer_diameter=od, hand_depth=rng.uniform(0.04, 0.08), hand_height=rng.uniform(0.01, 0.03),
              init_bite=rng.uniform(0.005, 0.015), volume_width=rng.uniform(0.06, 0.125), volume_depth=rng.uniform(0.04, 0.08),
              volume_height=rng.uniform(0.01, 0.03), nn_radius_frames=rng.uniform(0.008, 0.02),
              friction_coeff=rng.uniform(5.0, 40.0), min_viable=int(rng.randint(1, 12)), num_orientations=int(rng.randint(1, 9)),
              num_finger_placements=int(rng.randint(4, 13)), deepen_hand=int(rng.rand() < 0.8))
    if rng.rand() < 0.3:
        kw["min_aperture"], kw["max_aperture"] = rng.uniform(0.0, 0.03), rng.uniform(0.05, 0.1)
    axes = [int(a) for a in rng.permutation(3)[: rng.randint(1, 4)]]
    if wide:
        # beyond the tuned kernels' windows and tables (round 4: none of this is refused any more): image volumes up to a box
        # diagonal of ~0.28 m (the general shadow kernel), fingers of up to 0.16 m (more than 32 deepening steps), and the
        # approach-direction filter of the fused entry
        kw["volume_width"] = rng.uniform(0.12, 0.22)
        kw["volume_depth"] = rng.uniform(0.06, 0.16)
        kw["hand_depth"] = rng.uniform(0.06, 0.20)
        if rng.rand() < 0.6:
            d = rng.randn(3)
            kw["direction"] = [float(x) for x in d / np.linalg.norm(d)]
            kw["thresh_rad"] = rng.uniform(0.8, 2.4)
    return kw, axes, rng


def _set_all(p, kw, axes):
    for k, v in kw.items():
        if k == "direction":
            p.filter_approach_direction = 1
            for i, a in enumerate(v):
                p.direction[i] = a
        else:
            setattr(p, k, v)
    p.num_hand_axes = len(axes)
    for i, a in enumerate(axes):
        p.hand_axes[i] = a


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("GPD_FUZZ_WIDE", "5"))))
def test_fuzz_wide_geometry_matches_oracle(oracle_mod, seed):
    """The same with geometries the round-3 kernels refused (GPD_ERR_CAPACITY): wide and deep image volumes, long fingers,
    plus the device-side approach-direction filter.  GPD_FUZZ_WIDE=N widens the draw."""
    kw, axes, rng = _geometry(1000 + seed, wide=True)
    c = _case(seed + 11)
    if c["C"] != 15:
        c["C"] = 15  # the shadow channels are what the wide volumes stress
    w = _weights(15)
    gp, op = api.default_params(15), oracle_mod.default_params(15)
    _set_all(gp, kw, axes)
    _set_all(op, kw, axes)
    obj = np.flatnonzero(c["obj"])
    si = rng.choice(obj, size=min(60, len(obj)), replace=False).astype(np.int32)
    ctx = api.Context(gp)
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(c["xyz"], c["normals"], c["cam"], c["vp"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(op, c["xyz"], c["normals"], c["cam"], c["vp"], si, w)
        assert hands.shape == ohands.shape and n_cand == on_cand, (kw, axes)
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4, (kw, axes)
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes(), (kw, axes)
        fw = oracle_mod.filter_workspace(op, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(op, c["xyz"], c["normals"], c["cam"], c["vp"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg), (kw, axes)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("GPD_FUZZ_GEOMETRY", "6"))))
def test_fuzz_geometry_matches_oracle(oracle_mod, seed):
    """Random hand / image geometry on random clouds: records and images byte for byte, scores to 1e-4.
    GPD_FUZZ_GEOMETRY=N widens the draw (soak runs; `profiles/README.md`)."""
    kw, axes, rng = _geometry(seed)
    c = _case(seed + 3)
    C = c["C"]
    w = _weights(C)
    gp, op = api.default_params(C), oracle_mod.default_params(C)
    for p in (gp, op):
        for k, v in kw.items():
            setattr(p, k, v)
        p.num_hand_axes = len(axes)
        for i, a in enumerate(axes):
            p.hand_axes[i] = a
    obj = np.flatnonzero(c["obj"])
    si = rng.choice(obj, size=min(100, len(obj)), replace=False).astype(np.int32)
    ctx = api.Context(gp)
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(c["xyz"], c["normals"], c["cam"], c["vp"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(op, c["xyz"], c["normals"], c["cam"], c["vp"], si, w)
        assert hands.shape == ohands.shape and n_cand == on_cand, (kw, axes)
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4, (kw, axes)
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes(), (kw, axes)
        fw = oracle_mod.filter_workspace(op, ohands.copy())
        img, cand = ctx.images(fw)
        oimg, ocand = oracle_mod.images(op, c["xyz"], c["normals"], c["cam"], c["vp"], fw)
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg), (kw, axes)
    finally:
        ctx.close()
