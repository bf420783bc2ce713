"""ctypes access to the flat entry points of the host mirror (gpd_amd/host/libgpd_host.so).

The C++ classes (GraspDetector, Clustering, ...) are the product surface; this module only exposes
the `extern "C"` helpers for Python callers and tests.  No fallback: a missing library raises.
"""
import ctypes as C
import os

import numpy as np

from .api import HAND_DTYPE

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libgpd_host.so")
        if not os.path.exists(path):
            raise OSError("libgpd_host.so is not built (run `make -C gpd_amd/host` or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.gpd_host_find_clusters.restype = C.c_int
    return _LIB


def find_clusters(hands, scores, min_inliers=1, remove_inliers=False):
    """Clustering::findClusters (clustering.cpp:5-105) on POD records -> (records, scores f64, seed index)."""
    hands = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1)
    scores = np.ascontiguousarray(scores, np.float64)
    assert len(scores) == len(hands)
    n = len(hands)
    out = np.zeros(max(n, 1), HAND_DTYPE)
    osc = np.zeros(max(n, 1), np.float64)
    src = np.zeros(max(n, 1), np.int32)
    k = lib().gpd_host_find_clusters(hands.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p), n, int(min_inliers),
                                     int(bool(remove_inliers)), out.ctypes.data_as(C.c_void_p), osc.ctypes.data_as(C.c_void_p),
                                     src.ctypes.data_as(C.c_void_p))
    return out[:k].copy(), osc[:k].copy(), src[:k].copy()


def load_pcd(path, cap=1 << 22):
    """util::Cloud(filename): ASCII or uncompressed binary PCD -> (xyz f32 [n,3], normals f32 [n,3] or None)."""
    xyz = np.zeros((cap, 3), np.float32)
    nrm = np.zeros((cap, 3), np.float32)
    has = C.c_int(0)
    L = lib()
    L.gpd_host_load_pcd.restype = C.c_int
    n = L.gpd_host_load_pcd(str(path).encode(), xyz.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p), cap, C.byref(has))
    n = min(n, cap)
    return xyz[:n].copy(), (nrm[:n].copy() if has.value else None)
