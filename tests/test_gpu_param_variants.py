"""The HIP path against the oracle under non-default parameters: every knob of cfg/eigen_params.cfg
and the hand / image geometry files that the path reads (grasp_detector.cpp:44-161)."""
import os

import numpy as np
import pytest

from gpd_amd import api, synth


def _weights(C):
    g = os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


def _set(p, **kw):
    for k, v in kw.items():
        if k == "hand_axes":
            p.num_hand_axes = len(v)
            for i, a in enumerate(v):
                p.hand_axes[i] = a
        elif k == "workspace_grasps":
            for i, a in enumerate(v):
                p.workspace_grasps[i] = a
        else:
            setattr(p, k, v)
    return p


VARIANTS = {
    "three_axes": dict(hand_axes=[0, 1, 2]),                       # 24 slots per sample
    "four_orientations_axis1": dict(num_orientations=4, hand_axes=[1]),
    "no_deepen": dict(deepen_hand=0),
    "six_placements_wide_fingers": dict(num_finger_placements=6, finger_width=0.015),
    "small_hand": dict(hand_outer_diameter=0.09, hand_depth=0.045, hand_height=0.015, init_bite=0.008),
    "other_image_volume": dict(volume_width=0.08, volume_depth=0.05, volume_height=0.03),
    "wide_image_volume": dict(volume_width=0.16),  # box diagonal 0.176 m: the 64^3 / 128^3 voxel windows (images.hip Vox<WIDE>)
    "huge_image_volume": dict(volume_width=0.16, volume_depth=0.10),  # box diagonal 0.193 m: beyond the 64-voxel windows, the general shadow kernel
    "long_fingers": dict(hand_depth=0.20),  # 38 deepening steps (two rounds of 32), 0.2 m hand neighbourhoods
    "friction_viable_aperture": dict(friction_coeff=35.0, min_viable=2, min_aperture=0.02, max_aperture=0.07),
    "tight_workspace": dict(workspace_grasps=[-0.1, 0.1, -0.1, 0.12, -1.0, 1.0]),
    "frame_radius": dict(nn_radius_frames=0.02),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_detect_matches_oracle_under_parameter_variant(oracle_mod, name):
    cl = synth.make_cloud(4242, 20000)
    si = synth.sample_indices(cl, 160)
    w = _weights(15)
    gp = _set(api.default_params(15), **VARIANTS[name])
    op = _set(oracle_mod.default_params(15), **VARIANTS[name])
    ctx = api.Context(gp)
    try:
        ctx.set_lenet_weights(w)
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        hands, n_cand = ctx.detect(si)
        ohands, on_cand, _ = oracle_mod.detect(op, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, w)
        assert hands.shape == ohands.shape and n_cand == on_cand
        if name != "tight_workspace":
            assert n_cand > 50, n_cand
        # the whole record, byte for byte, except the float score (asserted to 1e-4, observed identical)
        a, b = hands.copy(), ohands.copy()
        assert np.abs(a["score"] - b["score"]).max() <= 1e-4
        a["score"] = 0
        b["score"] = 0
        assert a.tobytes() == b.tobytes()
        img, cand = ctx.images(oracle_mod.filter_workspace(op, ohands.copy()))
        oimg, ocand = oracle_mod.images(op, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"],
                                        oracle_mod.filter_workspace(op, ohands.copy()))
        assert np.array_equal(cand, ocand) and np.array_equal(img, oimg)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_image_volume_beyond_every_window_is_refused_not_truncated(oracle_mod):
    """The reference has no limit on the image volume (hand_set.cpp:138).  The tuned shadow kernels take boxes up to
    ~0.18 m of diagonal, the general one (`huge_image_volume` above: 0.16 x 0.10 m, parity with the oracle and with the
    reference's own code) windows of 256 voxels = 0.77 m; an image volume beyond THAT must come back as GPD_ERR_CAPACITY —
    never as images that silently lost shadow voxels."""
    cl = synth.make_cloud(4242, 20000)
    si = synth.sample_indices(cl, 40)
    gp = _set(api.default_params(15), volume_width=0.70, volume_depth=0.40)
    ctx = api.Context(gp)
    try:
        ctx.set_lenet_weights(_weights(15))
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        with pytest.raises(api.GpdHipError, match="capacity"):
            ctx.detect(si)
    finally:
        ctx.close()
