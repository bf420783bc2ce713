"""ctypes binding of oracle/libgpd_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# numpy mirror of `gpd_hand` (include/gpd_hip.h)
HAND_DTYPE = np.dtype([
    ("sample", "<f8", (3,)), ("frame", "<f8", (9,)), ("position", "<f8", (3,)),
    ("top", "<f8"), ("bottom", "<f8"), ("center", "<f8"), ("grasp_width", "<f8"),
    ("score", "<f4"), ("finger_placement_index", "<i4"), ("set_index", "<i4"), ("slot", "<i4"),
    ("valid", "u1"), ("half_antipodal", "u1"), ("full_antipodal", "u1"), ("pad_", "u1", (5,)),
], align=False)


class Params(C.Structure):
    """Mirror of `gpd_params` (include/gpd_hip.h)."""
    _fields_ = [
        ("finger_width", C.c_double), ("hand_outer_diameter", C.c_double), ("hand_depth", C.c_double),
        ("hand_height", C.c_double), ("init_bite", C.c_double), ("volume_width", C.c_double),
        ("volume_depth", C.c_double), ("volume_height", C.c_double), ("nn_radius_frames", C.c_double),
        ("friction_coeff", C.c_double), ("min_aperture", C.c_double), ("max_aperture", C.c_double),
        ("workspace_grasps", C.c_double * 6), ("image_size", C.c_int32), ("image_num_channels", C.c_int32),
        ("num_orientations", C.c_int32), ("num_finger_placements", C.c_int32), ("num_hand_axes", C.c_int32),
        ("hand_axes", C.c_int32 * 3), ("deepen_hand", C.c_int32), ("min_viable", C.c_int32),
        ("filter_approach_direction", C.c_int32), ("reserved_", C.c_int32), ("direction", C.c_double * 3), ("thresh_rad", C.c_double),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgpd_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.gpd_oracle_fastrand_at.argtypes = [C.c_uint64]
        assert _LIB.gpd_oracle_sizeof_hand() == HAND_DTYPE.itemsize, (_LIB.gpd_oracle_sizeof_hand(), HAND_DTYPE.itemsize)
        assert _LIB.gpd_oracle_sizeof_params() == C.sizeof(Params)
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def default_params(channels=15):
    p = Params()
    lib().gpd_oracle_default_params(C.byref(p))
    p.image_num_channels = channels
    return p


def radius_search(xyz, query, radius, cap=1 << 20):
    xyz = np.ascontiguousarray(xyz, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    idx = np.empty(cap, np.int32)
    d2 = np.empty(cap, np.float32)
    n = lib().gpd_oracle_radius_search(_p(xyz), len(xyz), _p(q), C.c_double(radius), _p(idx), _p(d2), cap)
    assert n <= cap
    return idx[:n].copy(), d2[:n].copy()


def eigen3(M):
    M = np.ascontiguousarray(M, np.float64)
    ev = np.empty(3)
    V = np.empty((3, 3))
    lib().gpd_oracle_eigen3(_p(M), _p(ev), _p(V))
    return ev, V


def finger_spacing(fw=0.01, od=0.12, depth=0.06, n=10):
    out = np.empty(2 * n)
    lib().gpd_oracle_finger_spacing(C.c_double(fw), C.c_double(od), C.c_double(depth), n, _p(out))
    return out


def angles(n=8):
    out = np.empty(n)
    lib().gpd_oracle_angles(n, _p(out))
    return out


def fastrand(n):
    out = np.empty(n, np.int32)
    lib().gpd_oracle_fastrand(n, _p(out))
    return out


def fastrand_at(offset):
    return lib().gpd_oracle_fastrand_at(offset)


def angle_axis(angle, axis):
    R = np.empty((3, 3))
    ax = np.ascontiguousarray(axis, np.float64)
    lib().gpd_oracle_angle_axis(C.c_double(angle), _p(ax), _p(R))
    return R


def frames(params, xyz, normals, sample_idx):
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    si = np.ascontiguousarray(sample_idx, np.int32)
    out = np.zeros((len(si), 12))
    has = np.zeros(len(si), np.uint8)
    lib().gpd_oracle_frames(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(si), len(si), _p(out), _p(has))
    return out, has


def search(params, xyz, normals, sample_idx):
    """HandSearch::searchHands -> (hands[n_sets, n_slots], n_sets)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    si = np.ascontiguousarray(sample_idx, np.int32)
    n_slots = params.num_hand_axes * params.num_orientations
    hands = np.zeros((len(si), n_slots), HAND_DTYPE)
    n_sets = C.c_int(0)
    lib().gpd_oracle_search(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(si), len(si), _p(hands), C.byref(n_sets))
    return hands[: n_sets.value].copy()


def filter_workspace(params, hands):
    hands = np.ascontiguousarray(hands)
    lib().gpd_oracle_filter(C.byref(params), _p(hands), hands.shape[0])
    return hands


def set_lcg_base(base):
    """Shadow draws consumed before the next images() call's first hand set (0: a whole cloud, the default; a later sample range
    of a sharded cloud: the draws of the ranges before it).  Stays set until changed."""
    lib().gpd_oracle_set_lcg_base(C.c_uint64(int(base)))


def last_lcg_draws():
    """Shadow draws the hand sets of the last images() call consumed."""
    L = lib()
    L.gpd_oracle_last_lcg_draws.restype = C.c_uint64
    return int(L.gpd_oracle_last_lcg_draws())


def images(params, xyz, normals, cam_source, view_points, hands, want_images=True):
    """ImageGenerator::createImages -> (images[n,60,60,C] u8, cand_index[n])."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    cam = np.ascontiguousarray(cam_source, np.int32).reshape(-1, len(xyz))
    vp = np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
    hands = np.ascontiguousarray(hands)
    n_valid = int(hands["valid"].astype(bool).sum())
    Cn = params.image_num_channels
    img = np.zeros((n_valid, 60, 60, Cn), np.uint8) if want_images else None
    cand = np.zeros(n_valid, np.int32)
    n = C.c_int(0)
    lib().gpd_oracle_images(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(cam), cam.shape[0], _p(vp), _p(hands),
                            hands.shape[0], _p(img), _p(cand), C.byref(n))
    assert n.value == n_valid
    return img, cand


def lenet(images_u8, weights):
    """EigenClassifier::classifyImages.  weights: dict c1w,c1b,c2w,c2b,f1w,f1b,f2w,f2b."""
    img = np.ascontiguousarray(images_u8, np.uint8)
    n, Cn = img.shape[0], img.shape[-1]
    w = [np.ascontiguousarray(weights[k], np.float32) for k in ("c1w", "c1b", "c2w", "c2b", "f1w", "f1b", "f2w", "f2b")]
    out = np.zeros(n, np.float32)
    lib().gpd_oracle_lenet(_p(img), n, Cn, *[_p(a) for a in w], _p(out))
    return out


def conv_generic(x, w, b):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    Cn, H, W = x.shape
    F, _, K, _ = w.shape
    out = np.zeros((F, H - K + 1, W - K + 1), np.float32)
    lib().gpd_oracle_conv_generic(_p(x), Cn, H, W, _p(w), _p(b), F, K, _p(out))
    return out


def detect(params, xyz, normals, cam_source, view_points, sample_idx, weights, max_cand=0, threads=None):
    """detectGrasps steps 1-4; returns (hands, n_cand, times[3] seconds)."""
    L = lib()
    if threads:
        L.gpd_oracle_set_num_threads(threads)
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    cam = np.ascontiguousarray(cam_source, np.int32).reshape(-1, len(xyz))
    vp = np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
    si = np.ascontiguousarray(sample_idx, np.int32)
    n_slots = params.num_hand_axes * params.num_orientations
    hands = np.zeros((len(si), n_slots), HAND_DTYPE)
    w = [np.ascontiguousarray(weights[k], np.float32) for k in ("c1w", "c1b", "c2w", "c2b", "f1w", "f1b", "f2w", "f2b")]
    wp = (C.c_void_p * 8)(*[a.ctypes.data for a in w])
    n_sets, n_cand = C.c_int(0), C.c_int(0)
    times = np.zeros(3)
    L.gpd_oracle_detect(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(cam), cam.shape[0], _p(vp), _p(si), len(si), wp,
                        _p(hands), C.byref(n_sets), C.byref(n_cand), None, int(max_cand), _p(times))
    return hands[: n_sets.value].copy(), n_cand.value, times


def num_threads():
    return lib().gpd_oracle_num_threads()


def voxelize(xyz, cell_size=0.003):
    """Cloud::voxelizeCloud -> (voxel points f32 [n,3], kept input index [n])."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros((len(xyz), 3), np.float32)
    src = np.zeros(len(xyz), np.int32)
    n = lib().gpd_oracle_voxelize(_p(xyz), len(xyz), C.c_float(cell_size), _p(out), _p(src))
    return out[:n].copy(), src[:n].copy()


def estimate_normals(xyz, cam_source=None, view_points=None, radius=0.03):
    """Cloud::calculateNormals (OMP radius PCA + reverseNormals) -> f32 [P,3]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    P = len(xyz)
    cam = np.ones((1, P), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, P)
    vp = np.zeros((1, 3)) if view_points is None else np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
    out = np.zeros((P, 3), np.float32)
    lib().gpd_oracle_normals(_p(xyz), P, _p(cam), cam.shape[0], _p(vp), C.c_double(radius), _p(out))
    return out


def find_clusters(hands, scores, min_inliers=1, remove_inliers=False):
    """Clustering::findClusters -> (cluster records, double scores, seed index)."""
    hands = np.ascontiguousarray(hands).reshape(-1)
    scores = np.ascontiguousarray(scores, np.float64)
    n = len(hands)
    out = np.zeros(max(n, 1), hands.dtype)
    osc = np.zeros(max(n, 1), np.float64)
    src = np.zeros(max(n, 1), np.int32)
    k = lib().gpd_oracle_find_clusters(_p(hands), _p(scores), n, int(min_inliers), int(bool(remove_inliers)), _p(out), _p(osc), _p(src))
    return out[:k].copy(), osc[:k].copy(), src[:k].copy()


def select(scores, num_selected):
    """GraspDetector::selectGrasps over a score list -> indices of the kept hands, in the kept order."""
    sc = np.ascontiguousarray(scores, np.float32)
    out = np.zeros(max(min(len(sc), num_selected), 1), np.int32)
    k = lib().gpd_oracle_select(_p(sc), len(sc), int(num_selected), _p(out))
    return out[:k].copy()


def search_xyz(params, xyz, normals, samples):
    """HandSearch::searchHands for samples given by coordinates (f64 [S,3]) -> hands [n_sets, n_slots]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    sm = np.ascontiguousarray(samples, np.float64).reshape(-1, 3)
    n_slots = params.num_hand_axes * params.num_orientations
    hands = np.zeros((len(sm), n_slots), HAND_DTYPE)
    ns = C.c_int(0)
    lib().gpd_oracle_search_xyz(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(sm), len(sm), _p(hands), C.byref(ns))
    return hands[:ns.value].copy()


def reevaluate(params, xyz, normals, hands):
    """HandSearch::reevaluateHypotheses on this cloud -> (labels int32 [n], hands with rewritten antipodal flags)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    h = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1).copy()
    labels = np.zeros(len(h), np.int32)
    lib().gpd_oracle_reevaluate(C.byref(params), _p(xyz), _p(normals), len(xyz), _p(h), len(h), _p(labels))
    return labels, h
