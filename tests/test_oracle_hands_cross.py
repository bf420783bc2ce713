"""The oracle's candidate evaluation against tests/pyref_hands.py, an independent numpy restatement
that materialises cropByHandHeight's padded list instead of modelling it as a ghost multiplicity."""
import numpy as np
import pytest

import pyref_hands
from gpd_amd import synth


def _set(p, **kw):
    for k, v in kw.items():
        if k == "hand_axes":
            p.num_hand_axes = len(v)
            for i, a in enumerate(v):
                p.hand_axes[i] = a
        else:
            setattr(p, k, v)
    return p


@pytest.mark.parametrize("variant", [dict(), dict(deepen_hand=0), dict(hand_axes=[0, 2], num_orientations=5),
                                     dict(num_finger_placements=6, friction_coeff=35.0, min_viable=2)])
def test_eval_hand_set_matches_independent_restatement(oracle_mod, variant):
    cl = synth.make_cloud(31, 9000)
    p = _set(oracle_mod.default_params(15), **variant)
    si = synth.sample_indices(cl, 45)
    hands = oracle_mod.search(p, cl["xyz"], cl["normals"], si)
    frames, has = oracle_mod.frames(p, cl["xyz"], cl["normals"], si)
    assert has.all() and hands.shape[0] == len(si)
    radius = max(p.hand_outer_diameter - p.finger_width, p.hand_depth, p.hand_height / 2.0)
    n_valid = n_full = n_ghost_sensitive = 0
    for s in range(len(si)):
        nbr, _ = oracle_mod.radius_search(cl["xyz"], cl["xyz"][si[s]], radius)
        ref = pyref_hands.eval_hand_set(p, cl["xyz"], cl["normals"], nbr, frames[s], oracle_mod.angle_axis)
        for j, r in enumerate(ref):
            h = hands[s, j]
            assert bool(h["valid"]) == r["valid"], (s, j)
            assert np.allclose(h["frame"].reshape(3, 3), r["frame"], rtol=0, atol=1e-15)
            if r["valid"]:
                n_valid += 1
                n_full += r["full"]
                assert int(h["finger_placement_index"]) == r["fidx"]
                assert bool(h["half_antipodal"]) == r["half"] and bool(h["full_antipodal"]) == r["full"]
                for f, key in (("top", "top"), ("bottom", "bottom"), ("center", "center"), ("grasp_width", "width")):
                    assert h[f] == r[key], (f, h[f], r[key])
                assert np.array_equal(h["position"], r["position"])
            else:
                # invalid hands keep what Hand() saw before deepening
                assert int(h["finger_placement_index"]) == r["fidx"] and h["top"] == r["top"] and h["bottom"] == r["bottom"]
    assert n_valid > 40 and n_full > 0
