"""The cases on which the oracle (and the HIP path) are pinned against the REFERENCE's own code.

tests/golden/make_ref_pins.py runs every case through oracle/_ref/libgpd_ref.so — the reference's translation units
compiled unmodified through oracle/shim (build container only: the reference tree does not travel) — and commits what
the reference returned as tests/golden/ref_pin_*.npz.  tests/test_ref_pin.py recomputes each case with the oracle (CPU
suite) and with the HIP path through the C-ABI (GPU suite) and compares with those files, bit for bit."""
import hashlib
import os

import numpy as np

from gpd_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLOUD_SEED, CLOUD_POINTS, NUM_SAMPLES = 4242, 8000, 24
RECORD_FIELDS = ("sample", "frame", "position", "top", "bottom", "center", "grasp_width", "finger_placement_index", "half_antipodal",
                 "full_antipodal")


def set_params(p, **kw):
    for k, v in kw.items():
        if k == "hand_axes":
            p.num_hand_axes = len(v)
            for i, a in enumerate(v):
                p.hand_axes[i] = a
        elif k == "workspace_grasps":
            for i, a in enumerate(v):
                p.workspace_grasps[i] = a
        elif k == "direction":  # filterGraspsDirection: setting a direction switches the filter on
            p.filter_approach_direction = 1
            for i, a in enumerate(v):
                p.direction[i] = a
        else:
            setattr(p, k, v)
    return p


def _cams(n, P, seed=9):
    rng = np.random.default_rng(seed)
    if n == 2:
        cam = np.zeros((2, P), np.int32)
        cam[0] = rng.random(P) < 0.7
        cam[1] = (rng.random(P) < 0.6) | (cam[0] == 0)
        return cam, np.array([[0.0, 0.0, 0.8], [0.5, -0.4, 0.3]])
    if n > 3:
        # a rig of n cameras around the first view point: most see most of the scene, one sees nothing, one only one side
        cam = (rng.random((n, P)) < 0.93).astype(np.int32)
        cam[n // 2] = 0
        cam[0, cam.sum(axis=0) == 0] = 1
        vp = np.array([[0.0, 0.0, 0.8]]) + rng.uniform(-0.05, 0.05, (n, 3))
        vp[n - 1] = [0.4, -0.3, 0.5]
        return cam, vp
    cam = np.zeros((3, P), np.int32)
    cam[0] = rng.random(P) < 0.5
    cam[1] = rng.random(P) < 0.5
    cam[2] = (cam[0] + cam[1]) == 0
    return cam, np.array([[0.0, 0.0, 0.8], [0.5, -0.4, 0.3], [-0.3, 0.3, 0.6]])


# name -> (channels, parameter overrides, number of cameras)
VARIANTS = {
    "default_c15": (15, {}, 1),
    "default_c12": (12, {}, 1),
    "default_c3": (3, {}, 1),
    "default_c1": (1, {}, 1),
    "three_axes": (15, dict(hand_axes=[0, 1, 2]), 1),
    "axis1_four_orientations": (15, dict(hand_axes=[1], num_orientations=4), 1),
    "no_deepen": (15, dict(deepen_hand=0), 1),
    "twelve_orientations": (15, dict(num_orientations=12), 1),
    "six_placements_wide_fingers": (15, dict(num_finger_placements=6, finger_width=0.015), 1),
    "small_hand": (15, dict(hand_outer_diameter=0.09, hand_depth=0.045, hand_height=0.015, init_bite=0.008), 1),
    "deep_hand": (15, dict(hand_depth=0.08), 1),
    "other_image_volume": (15, dict(volume_width=0.08, volume_depth=0.05, volume_height=0.03), 1),
    "wide_image_volume": (15, dict(volume_width=0.16), 1),  # beyond the default voxel windows of the shadow kernels (Vox<WIDE>)
    "huge_image_volume": (15, dict(volume_width=0.16, volume_depth=0.10), 1),  # beyond the wide windows too: the general shadow kernel
    "friction_viable_aperture": (15, dict(friction_coeff=35.0, min_viable=2, min_aperture=0.02, max_aperture=0.07), 1),
    "tight_workspace": (15, dict(workspace_grasps=[-0.1, 0.1, -0.1, 0.12, -1.0, 1.0]), 1),
    "frame_radius": (15, dict(nn_radius_frames=0.02), 1),
    "two_cameras": (15, {}, 2),
    "three_cameras": (15, {}, 3),
    "twelve_cameras": (15, {}, 12),  # the reference's camera_source has no bound on its rows (cloud.h); the kernels take 32
    # (round 6) the same scene with sensor-like coordinates (synth.off_lattice: every point moved by a seeded sub-voxel offset): no
    # distance ties, no point exactly on a decision plane — where the unpinned third-party behaviours (FLANN's tie order, ulp-level
    # eigen-solver differences) stop deciding outputs (profiles/r05_thirdparty_sensitivity.txt)
    "offlattice_c15": (15, {}, 1),
    "offlattice_c3": (3, {}, 1),
    "offlattice_two_cameras": (15, {}, 2),
    "offlattice_three_axes": (15, dict(hand_axes=[0, 1, 2]), 1),
}
FULL_IMAGES = ("default_c15", "default_c12", "default_c3", "default_c1", "two_cameras", "offlattice_c15")  # the others are pinned by digest


def cloud(name=""):
    cl = synth.make_cloud(CLOUD_SEED, CLOUD_POINTS)
    return synth.off_lattice(cl) if name.startswith("offlattice") else cl


def case_inputs(name, default_params):
    """(params, cloud dict, sample indices, cam_source, view_points) of a variant; `default_params(channels)` builds the
    parameter block of whoever runs the case (oracle.default_params / api.default_params: same fields)."""
    C, over, ncam = VARIANTS[name]
    cl = cloud(name)
    si = synth.sample_indices(cl, NUM_SAMPLES)
    p = set_params(default_params(C), **over)
    if ncam == 1:
        cam, vp = cl["cam_source"], cl["view_points"]
    else:
        cam, vp = _cams(ncam, len(cl["xyz"]))
    return p, cl, si, cam, vp


def weights(C, trained_magnitude=False):
    g = os.path.join(GOLD, "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=trained_magnitude)


def image_digests(images):
    """One SHA-1 per image (20 bytes): pins every byte of an image list at 0.04 % of its size."""
    out = np.zeros((len(images), 20), np.uint8)
    for i, im in enumerate(images):
        out[i] = np.frombuffer(hashlib.sha1(np.ascontiguousarray(im).tobytes()).digest(), np.uint8)
    return out


def digest(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def records_equal(a, b, where=None):
    """Names of the record fields in which two hand arrays differ (over `where`, a boolean mask; default: everywhere)."""
    bad = []
    for f in RECORD_FIELDS:
        x, y = (a[f], b[f]) if where is None else (a[f][where], b[f][where])
        if not np.array_equal(x, y):
            bad.append(f)
    return bad


def load_pin(name):
    path = os.path.join(GOLD, "ref_pin_%s.npz" % name)
    return dict(np.load(path)) if os.path.exists(path) else None


def assert_scores(ctx, img, want, tol=1e-4, rel=0.0):
    """Classifier::classifyImages in both modes of the library: the f32-chain mode reproduces `want` (the oracle's k-ascending
    fmaf chains) bit for bit; the default split mode (int8 / bf16 matrix pipes on exactly split operands) is within `tol`
    absolute (BASELINE.json's 1e-4 at trained-net magnitudes) or `rel` of the largest |score| (weight sets that drive the
    logits to ~1000, where one f32 ulp is 6e-5).  Returns the split-mode scores; the context is left in split mode."""
    from gpd_amd import api
    ctx.set_lenet_mode(api.LENET_F32_CHAIN)
    got = ctx.score(img)
    assert np.array_equal(got, want), float(np.abs(got - want).max()) if len(want) else 0.0
    ctx.set_lenet_mode(api.LENET_SPLIT)
    got = ctx.score(img)
    if len(want):
        bound = max(tol, rel * float(np.abs(want).max()))
        assert np.abs(got - want).max() <= bound, (float(np.abs(got - want).max()), bound)
    return got
