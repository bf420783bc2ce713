"""ctypes binding of oracle/_ref/libgpd_ref.so — the REFERENCE's own translation units, compiled unmodified through the
test-only third-party subsets of oracle/shim/ (oracle/build_ref.sh tier A).  TEST INFRASTRUCTURE ONLY.

The library exists only where /root/reference exists (the build container); tests/golden/make_ref_pins.py runs it there
and commits what it returns as tests/golden/ref_pin_*.npz, which travel to the GPU box.  `available()` says whether the
live library can be used; tests that need it skip otherwise and fall back on the committed pins."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from .oracle import HAND_DTYPE, Params, _p

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("GPD_REF_LIB") or os.path.join(_HERE, "_ref", "libgpd_ref.so")  # (GPD_REF_LIB: a perturbed build of profiles/thirdparty_sensitivity.py)
_LIB = None


def build():
    """oracle/build_ref.sh A — needs the reference tree; returns True when the library exists afterwards."""
    subprocess.call(["bash", os.path.join(_HERE, "build_ref.sh"), "A"])
    return os.path.exists(_PATH)


def available():
    return os.path.exists(_PATH)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("oracle/_ref/libgpd_ref.so is missing: run oracle/build_ref.sh A where /root/reference exists")
        L = C.CDLL(_PATH)
        for name in ("gpd_ref_create", "gpd_ref_cloud_create", "gpd_ref_cloud_load"):
            getattr(L, name).restype = C.c_void_p
        assert L.gpd_ref_abi() == HAND_DTYPE.itemsize
        _LIB = L
    return _LIB


class _Quiet:
    """The reference prints its parameters and timings on stdout; keep it off the test logs (GPD_REF_VERBOSE=1 shows it)."""

    def __enter__(self):
        if os.environ.get("GPD_REF_VERBOSE"):
            self.saved = None
            return self
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            C.CDLL(None).fflush(None)
            os.dup2(self.saved, 1)
            os.close(self.saved)
            os.close(self.null)
        return False


def set_product_mode(mode):
    """0: a*b summed in ascending k, unfused (default); 1: fmaf chain in ascending k; 2: long double accumulation."""
    with _Quiet():
        lib().gpd_ref_set_product_mode(int(mode))


def reset_shadow_seed():
    """HandSet::seed_ = 0: the state of a fresh process (the oracle restarts the shadow LCG for every cloud)."""
    lib().gpd_ref_reset_shadow_seed()


def _fmt(v):
    return repr(float(v))


def write_cfg(path, params, weights_dir="", num_samples=100000, num_selected=100, min_inliers=0, workspace=None, voxelize=0,
              voxel_size=0.003, normals_radius=0.03, filter_approach_direction=0, direction=(1, 0, 0), thresh_rad=2.0,
              num_threads=1):
    """A cfg file in the format of the reference's cfg/eigen_params.cfg, hand and image geometry in the same file
    (hand_geometry_filename = 0, grasp_detector.cpp:12-15,123-125), every plot_* key cleared."""
    p = params
    ws = workspace if workspace is not None else (-1.0, 1.0, -1.0, 1.0, -1.0, 1.0)
    axes = " ".join(str(p.hand_axes[i]) for i in range(p.num_hand_axes))
    lines = [
        "hand_geometry_filename = 0", "image_geometry_filename = 0",
        "finger_width = " + _fmt(p.finger_width), "hand_outer_diameter = " + _fmt(p.hand_outer_diameter),
        "hand_depth = " + _fmt(p.hand_depth), "hand_height = " + _fmt(p.hand_height), "init_bite = " + _fmt(p.init_bite),
        "volume_width = " + _fmt(p.volume_width), "volume_depth = " + _fmt(p.volume_depth), "volume_height = " + _fmt(p.volume_height),
        "image_size = %d" % p.image_size, "image_num_channels = %d" % p.image_num_channels,
        "voxelize = %d" % voxelize, "voxel_size = " + _fmt(voxel_size), "remove_outliers = 0",
        "workspace = " + " ".join(_fmt(v) for v in ws), "sample_above_plane = 0", "normals_radius = " + _fmt(normals_radius),
        "num_samples = %d" % num_samples, "num_threads = %d" % num_threads, "nn_radius = " + _fmt(p.nn_radius_frames),
        "num_orientations = %d" % p.num_orientations, "num_finger_placements = %d" % p.num_finger_placements, "hand_axes = " + axes,
        "deepen_hand = %d" % p.deepen_hand, "friction_coeff = " + _fmt(p.friction_coeff), "min_viable = %d" % p.min_viable,
        "min_aperture = " + _fmt(p.min_aperture), "max_aperture = " + _fmt(p.max_aperture),
        "workspace_grasps = " + " ".join(_fmt(p.workspace_grasps[i]) for i in range(6)),
        "filter_approach_direction = %d" % filter_approach_direction, "direction = " + " ".join(_fmt(v) for v in direction),
        "thresh_rad = " + _fmt(thresh_rad), "min_inliers = %d" % min_inliers, "num_selected = %d" % num_selected,
        "plot_normals = 0", "plot_samples = 0", "plot_candidates = 0", "plot_filtered_candidates = 0", "plot_valid_grasps = 0",
        "plot_clustered_grasps = 0", "plot_selected_grasps = 0",
    ]
    if weights_dir:
        lines.append("weights_file = " + (weights_dir if weights_dir.endswith("/") else weights_dir + "/"))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def write_weights_dir(path, weights):
    """The eight raw float32 files EigenClassifier reads (eigen_classifier.cpp:28-50); returns the directory with a
    trailing slash."""
    os.makedirs(path, exist_ok=True)
    names = {"c1w": "conv1_weights", "c1b": "conv1_biases", "c2w": "conv2_weights", "c2b": "conv2_biases", "f1w": "ip1_weights",
             "f1b": "ip1_biases", "f2w": "ip2_weights", "f2b": "ip2_biases"}
    for k, n in names.items():
        np.ascontiguousarray(weights[k], "<f4").tofile(os.path.join(path, n + ".bin"))
    return path if path.endswith("/") else path + "/"


class Cloud:
    """util::Cloud (reference include/gpd/util/cloud.h)."""

    def __init__(self, xyz=None, normals=None, cam_source=None, view_points=None, pcd=None):
        L = lib()
        if pcd is not None:
            with _Quiet():
                self.h = C.c_void_p(L.gpd_ref_cloud_load(pcd.encode()))
        else:
            xyz = np.ascontiguousarray(xyz, np.float32)
            P = len(xyz)
            cam = np.ones((1, P), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, P)
            vp = np.zeros((1, 3)) if view_points is None else np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
            assert len(vp) == cam.shape[0]
            nrm = None if normals is None else np.ascontiguousarray(normals, np.float32)
            with _Quiet():
                self.h = C.c_void_p(L.gpd_ref_cloud_create(_p(xyz), P, _p(nrm), _p(cam), cam.shape[0], _p(vp)))

    def close(self):
        if self.h:
            with _Quiet():
                lib().gpd_ref_cloud_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sample_indices(self, idx):
        idx = np.ascontiguousarray(idx, np.int32)
        with _Quiet():
            lib().gpd_ref_cloud_set_sample_indices(self.h, _p(idx), len(idx))

    def set_samples(self, xyz):
        s = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        with _Quiet():
            lib().gpd_ref_cloud_set_samples(self.h, _p(s), len(s))

    def sample_indices(self):
        with _Quiet():
            n = lib().gpd_ref_cloud_sample_indices(self.h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        with _Quiet():
            lib().gpd_ref_cloud_sample_indices(self.h, _p(out), n)
        return out[:n]

    def size(self):
        with _Quiet():
            return lib().gpd_ref_cloud_size(self.h)

    def get(self):
        """(xyz f32 [P,3], normals f64 [N,3], cam_source i32 [cams,P]) of the processed cloud."""
        L = lib()
        P = L.gpd_ref_cloud_size(self.h)
        N = L.gpd_ref_cloud_num_normals(self.h)
        xyz = np.zeros((P, 3), np.float32)
        nrm = np.zeros((N, 3), np.float64)
        L.gpd_ref_cloud_get(self.h, _p(xyz), _p(nrm) if N else None, None)
        return xyz, nrm

    def filter_workspace(self, ws):
        w = np.ascontiguousarray(ws, np.float64)
        with _Quiet():
            lib().gpd_ref_cloud_filter_workspace(self.h, _p(w))

    def voxelize(self, cell=0.003):
        with _Quiet():
            lib().gpd_ref_cloud_voxelize(self.h, C.c_float(cell))

    def calculate_normals(self, radius=0.03):
        with _Quiet():
            lib().gpd_ref_cloud_calculate_normals(self.h, C.c_double(radius))


class Detector:
    """gpd::GraspDetector built from a cfg file written from `params` (+ the glue's own ImageGenerator / Classifier)."""

    def __init__(self, params, weights=None, **cfg):
        self._tmp = tempfile.TemporaryDirectory(prefix="gpd_ref_")
        wdir = ""
        if weights is not None:
            wdir = write_weights_dir(os.path.join(self._tmp.name, "params"), weights)
        path = os.path.join(self._tmp.name, "ref.cfg")
        write_cfg(path, params, weights_dir=wdir, **cfg)
        self.params = params
        with _Quiet():
            self.h = C.c_void_p(lib().gpd_ref_create(path.encode()))
        with _Quiet():
            self.n_slots = lib().gpd_ref_num_slots(self.h)
        self.n_sets = 0
        if wdir:
            with _Quiet():
                assert lib().gpd_ref_classifier_load(self.h, wdir.encode()) == 0

    def close(self):
        if self.h:
            with _Quiet():
                lib().gpd_ref_destroy(self.h)
            self.h = None
        self._tmp.cleanup()

    def load_weights(self, weights, name="params2"):
        """Classifier::create on another parameter directory (the detector's own classifier is not touched)."""
        wdir = write_weights_dir(os.path.join(self._tmp.name, name), weights)
        with _Quiet():
            assert lib().gpd_ref_classifier_load(self.h, wdir.encode()) == 0

    def preprocess(self, cloud):
        with _Quiet():
            lib().gpd_ref_preprocess(self.h, cloud.h)

    def generate(self, cloud, cap_sets):
        """generateGraspCandidates -> records [n_sets, n_slots]."""
        reset_shadow_seed()
        hands = np.zeros((cap_sets, self.n_slots), HAND_DTYPE)
        n = C.c_int(0)
        with _Quiet():
            lib().gpd_ref_generate(self.h, cloud.h, _p(hands), cap_sets, C.byref(n))
        assert n.value <= cap_sets
        self.n_sets = n.value
        return hands[: n.value].copy()

    def filter_workspace(self, ws=None):
        """filterGraspsWorkspace on the current list -> valid flags [n_sets, n_slots] (generate's numbering)."""
        w = np.ascontiguousarray(ws if ws is not None else [self.params.workspace_grasps[i] for i in range(6)], np.float64)
        valid = np.zeros((self.n_sets, self.n_slots), np.uint8)
        with _Quiet():
            lib().gpd_ref_filter_workspace(self.h, _p(w), self.n_sets, _p(valid))
        return valid

    def filter_direction(self, direction, thresh_rad):
        d = np.ascontiguousarray(direction, np.float64)
        valid = np.zeros((self.n_sets, self.n_slots), np.uint8)
        with _Quiet():
            lib().gpd_ref_filter_direction(self.h, _p(d), C.c_double(thresh_rad), self.n_sets, _p(valid))
        return valid

    def images(self, cloud, cap):
        """ImageGenerator::createImages on the current list -> (images [n,S,S,C] u8, cand index [n])."""
        S, Cn = self.params.image_size, self.params.image_num_channels
        img = np.zeros((cap, S, S, Cn), np.uint8)
        cand = np.zeros(cap, np.int32)
        n = C.c_int(0)
        with _Quiet():
            rc = lib().gpd_ref_images(self.h, cloud.h, _p(img), _p(cand), cap, C.byref(n))
        assert rc == 0 and n.value <= cap, (rc, n.value, cap)
        return img[: n.value].copy(), cand[: n.value].copy()

    def classify(self, images):
        img = np.ascontiguousarray(images, np.uint8)
        n, S, _, Cn = img.shape
        out = np.zeros(n, np.float32)
        with _Quiet():
            assert lib().gpd_ref_classify(self.h, _p(img), n, S, Cn, _p(out)) == 0
        return out

    def select(self, scores):
        sc = np.ascontiguousarray(scores, np.float32)
        out = np.zeros(max(len(sc), 1), np.int32)
        with _Quiet():
            k = lib().gpd_ref_select(self.h, _p(sc), len(sc), _p(out))
        return out[:k].copy()

    def find_clusters(self, hands, scores, min_inliers=1, remove_inliers=False):
        h = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1)
        sc = np.ascontiguousarray(scores, np.float64)
        out = np.zeros(max(len(h), 1), HAND_DTYPE)
        osc = np.zeros(max(len(h), 1), np.float64)
        with _Quiet():
            k = lib().gpd_ref_find_clusters(self.h, _p(h), _p(sc), len(h), int(min_inliers), int(bool(remove_inliers)), _p(out), _p(osc))
        return out[:k].copy(), osc[:k].copy()

    def reevaluate(self, cloud, hands):
        h = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1).copy()
        labels = np.zeros(len(h), np.int32)
        with _Quiet():
            lib().gpd_ref_reevaluate(self.h, cloud.h, _p(h), len(h), _p(labels))
        return labels, h

    def detect(self, cloud, cap=4096):
        out = np.zeros(cap, HAND_DTYPE)
        with _Quiet():
            n = lib().gpd_ref_detect(self.h, cloud.h, _p(out), cap)
        assert n <= cap
        return out[:n].copy()


def conv_forward(x, w, b):
    """net::ConvLayer::forward (conv_layer.cpp:26-98): x [C,H,W], w [F,C,K,K] -> [F,H-K+1,W-K+1]."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    Cn, H, W = x.shape
    F, _, K, _ = w.shape
    out = np.zeros((F, H - K + 1, W - K + 1), np.float32)
    with _Quiet():
        lib().gpd_ref_conv_forward(_p(x), Cn, H, W, _p(w), _p(b), F, K, _p(out))
    return out
