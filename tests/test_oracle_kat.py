"""Known-answer tests of the CPU oracle (the reference has no golden vectors)."""
import math

import numpy as np


def test_finger_spacing_table(oracle_mod):
    # finger_hand.cpp:13-18 with od=0.12, fw=0.01, n=10
    sp = oracle_mod.finger_spacing()
    right = [0, 0.012222222222222223, 0.024444444444444446, 0.03666666666666667, 0.04888888888888889,
             0.061111111111111116, 0.07333333333333333, 0.08555555555555557, 0.09777777777777778, 0.11]
    left = [-0.11, -0.09777777777777778, -0.08555555555555555, -0.07333333333333333, -0.061111111111111095,
            -0.04888888888888888, -0.03666666666666666, -0.02444444444444443, -0.012222222222222212, 5.204170427930421e-18]
    assert list(sp[10:]) == right
    assert list(sp[:10]) == left
    assert sp[9] != 0.0


def test_orientation_angles(oracle_mod):
    # hand_search.cpp:151-155
    want = [-1.5707963267948966, -1.1780972450961724, -0.7853981633974483, -0.39269908169872414, 0.0,
            0.39269908169872414, 0.7853981633974483, 1.1780972450961724]
    assert list(oracle_mod.angles(8)) == want


def test_fastrand_stream_and_jump(oracle_mod):
    # hand_set.cpp:263-266 from seed 0 (the MSVC rand() LCG)
    assert list(oracle_mod.fastrand(8)) == [38, 7719, 21238, 2437, 8855, 11797, 8365, 32285]
    seq = oracle_mod.fastrand(5000)
    for off in (0, 1, 33, 4999, 1234):
        assert oracle_mod.fastrand_at(off) == seq[off]
    # jump-ahead far beyond what a loop would reach in a test
    s = 0
    a, c = 214013, 2531011
    n = 10 ** 6 + 7
    for _ in range(n + 1):
        s = (a * s + c) & 0xFFFFFFFF
    assert oracle_mod.fastrand_at(n) == (s >> 16) & 0x7FFF


def test_rot_binormal(oracle_mod):
    R = oracle_mod.angle_axis(math.pi, [0, 1, 0])
    assert R[0, 0] == -1.0 and R[1, 1] == 1.0 and R[2, 2] == -1.0
    assert R[0, 2] == 1.2246467991473532e-16 and R[2, 0] == -1.2246467991473532e-16
    assert R[0, 1] == 0 and R[1, 0] == 0 and R[1, 2] == 0 and R[2, 1] == 0


def test_angle_axis_is_rotation(oracle_mod):
    for ang in oracle_mod.angles(8):
        R = oracle_mod.angle_axis(ang, [0, 0, 1])
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)
        assert np.allclose(R[:2, :2], [[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])


def test_float_radii():
    # (float)(r*r) as pcl::KdTreeFLANN::radiusSearch passes to FLANN
    assert float(np.float32(0.01 * 0.01)) == 9.999999747378752e-05
    assert float(np.float32(0.10 * 0.10)) == 0.009999999776482582
    assert float(np.float32(0.11 * 0.11)) == 0.01209999993443489


def test_conv_layer_kat(oracle_mod):
    # src/tests/test_conv_layer.cpp:11-16 (CS231n example), expected [4 3 4 2 4 3 2 3 4]
    x = np.array([[1, 1, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 1, 1, 1], [0, 0, 1, 1, 0], [0, 1, 1, 0, 0]], np.float32)[None]
    w = np.array([[1, 0, 1], [0, 1, 0], [1, 0, 1]], np.float32)[None, None]
    y = oracle_mod.conv_generic(x, w, np.zeros(1, np.float32))
    assert y.reshape(-1).tolist() == [4, 3, 4, 2, 4, 3, 2, 3, 4]


def test_struct_sizes_agree(oracle_mod):
    from gpd_amd import api
    assert api.HAND_DTYPE.itemsize == oracle_mod.HAND_DTYPE.itemsize == 176
    import ctypes
    assert ctypes.sizeof(api.Params) == ctypes.sizeof(oracle_mod.Params)
