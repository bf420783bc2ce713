"""ctypes binding of libgpd_hip.so (include/gpd_hip.h) — the product path.

There is no CPU fallback: if the HIP library is missing or a call fails, this
raises.  The Python layer only marshals numpy arrays into the C-ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPD_HIP_LIB: another build of the same library (A/B timing of a kernel change on one GPU box)
LIB_PATH = os.environ.get("GPD_HIP_LIB") or os.path.join(_HERE, "libgpd_hip.so")
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


class DetectJob(C.Structure):
    """Mirror of `gpd_detect_job` (include/gpd_hip.h)."""
    _fields_ = [
        ("xyz", C.c_void_p), ("normals", C.c_void_p), ("cam_source", C.c_void_p), ("view_points", C.c_void_p),
        ("sample_indices", C.c_void_p), ("hands", C.c_void_p),
        ("num_points", C.c_int32), ("num_cams", C.c_int32), ("num_samples", C.c_int32), ("num_selected", C.c_int32),
        ("hands_capacity", C.c_int32), ("num_sets", C.c_int32), ("num_candidates", C.c_int32), ("num_hands", C.c_int32),
        ("status", C.c_int32), ("stage_ms", C.c_float * 3), ("host_ms", C.c_float * 5), ("allocs", C.c_int32),
        ("reserved_", C.c_int32), ("lcg_base", C.c_uint64), ("lcg_draws", C.c_uint64),
        ("raw", C.c_int32), ("voxel_size", C.c_float), ("workspace", C.c_void_p), ("normals_radius", C.c_double), ("sample_xyz", C.c_void_p),
        ("num_points_processed", C.c_int32), ("num_samples_processed", C.c_int32),
    ]


class GpdHipError(RuntimeError):
    pass


# gpd_hip_set_lenet_mode (include/gpd_hip.h)
LENET_SPLIT, LENET_F32_CHAIN = 0, 1


def bind_host_thread(device):
    """gpd_hip_bind_host_thread: the calling thread onto the CPUs of the device's NUMA node -> (node or -1, cpus)."""
    n = C.c_int(0)
    node = lib().gpd_hip_bind_host_thread(int(device), C.byref(n))
    return int(node), int(n.value)


EXPORTS = ["gpd_hip_default_params", "gpd_hip_create", "gpd_hip_destroy", "gpd_hip_last_error",
           "gpd_hip_set_lenet_weights", "gpd_hip_score", "gpd_hip_upload_cloud", "gpd_hip_search",
           "gpd_hip_images", "gpd_hip_detect", "gpd_hip_last_stage_ms", "gpd_hip_replay", "gpd_hip_replay_times", "gpd_hip_last_images_stats", "gpd_hip_estimate_normals",
           "gpd_hip_search_samples", "gpd_hip_detect_samples", "gpd_hip_reevaluate", "gpd_hip_replay_kernel_ms", "gpd_hip_last_centre_chains",
           "gpd_hip_detect_select", "gpd_hip_detect_batch", "gpd_hip_detect_batch_multi", "gpd_hip_conv1_stats", "gpd_hip_last_fallbacks", "gpd_hip_preprocess_cloud", "gpd_hip_find_clusters", "gpd_hip_reserve", "gpd_hip_bind_host_thread",
           "gpd_hip_set_lenet_mode", "gpd_hip_lenet_debug", "gpd_hip_lenet_fast_tables", "gpd_hip_detect_sharded"]


def build(prof=True):
    """Compile libgpd_hip.so for gfx950 (hipcc cross-compiles without a GPU) and, with `prof`, the profiling build
    libgpd_hip_prof.so (needs the roctx headers; the release library does not)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])
    if prof:
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc"), "prof"])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GpdHipError("libgpd_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                              "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.gpd_hip_last_error.restype = C.c_char_p
        L.gpd_hip_create.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.gpd_hip_destroy.argtypes = [C.c_void_p]
        L.gpd_hip_destroy.restype = None
        L.gpd_hip_set_lenet_weights.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.gpd_hip_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.gpd_hip_upload_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.gpd_hip_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.gpd_hip_search_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.gpd_hip_detect_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gpd_hip_reevaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.gpd_hip_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.gpd_hip_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gpd_hip_detect_select.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gpd_hip_detect_batch.argtypes = [C.c_void_p, C.POINTER(DetectJob), C.c_int]
        L.gpd_hip_detect_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(DetectJob), C.c_int]
        L.gpd_hip_detect_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(DetectJob)]
        L.gpd_hip_last_fallbacks.argtypes = [C.c_void_p, C.c_void_p]
        L.gpd_hip_last_centre_chains.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.gpd_hip_last_stage_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.gpd_hip_replay.argtypes = [C.c_void_p, C.c_int]
        L.gpd_hip_estimate_normals.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.gpd_hip_find_clusters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int)]
        L.gpd_hip_preprocess_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.gpd_hip_last_images_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.gpd_hip_replay_kernel_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.gpd_hip_conv1_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.gpd_hip_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gpd_hip_bind_host_thread.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.gpd_hip_set_lenet_mode.argtypes = [C.c_void_p, C.c_int]
        L.gpd_hip_lenet_debug.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.gpd_hip_replay_times.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        _LIB = L
    return _LIB


def default_params(channels=15):
    p = Params()
    lib().gpd_hip_default_params(C.byref(p))
    p.image_num_channels = channels
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One gpd_hip_ctx: owns the device copy of the cloud, the LeNet weights and all scratch."""

    def __init__(self, params=None, device=0):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        self._check(lib().gpd_hip_create(int(device), C.byref(self.params), C.byref(self._h)))
        self.n_slots = self.params.num_hand_axes * self.params.num_orientations

    def _check(self, rc):
        if rc != 0:
            raise GpdHipError("libgpd_hip error %d: %s" % (rc, lib().gpd_hip_last_error().decode()))

    def close(self):
        if self._h:
            lib().gpd_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lenet_weights(self, w):
        arrs = [np.ascontiguousarray(w[k], np.float32) for k in ("c1w", "c1b", "c2w", "c2b", "f1w", "f1b", "f2w", "f2b")]
        ch = self.params.image_num_channels
        assert arrs[0].size == 20 * ch * 25, "conv1 weights do not match image_num_channels"
        self._check(lib().gpd_hip_set_lenet_weights(self._h, ch, *[_ptr(a) for a in arrs]))

    def set_lenet_mode(self, mode):
        """LENET_SPLIT (default: int8 / bf16 matrix pipes on exactly split operands) or LENET_F32_CHAIN (the oracle's
        k-ascending fmaf chains, bit-identical, 1/16 of the matrix rate)."""
        self._check(lib().gpd_hip_set_lenet_mode(self._h, int(mode)))

    def lenet_debug(self, which, n):
        """Test hook: pool1 (0), the flattened pool2 as bf16 planes (1) or ip1 transposed (2) of the last score() pass."""
        out = {0: np.zeros((n, 15680), np.float32), 1: np.zeros((3, n, 7200), np.uint16), 2: np.zeros((500, n), np.float32)}[which]
        self._check(lib().gpd_hip_lenet_debug(self._h, int(which), int(n), _ptr(out)))
        return out

    def score(self, images=None, n=None):
        """Classifier::classifyImages; images [n,60,60,C] u8, or None to score the device images."""
        if images is not None:
            images = np.ascontiguousarray(images, np.uint8)
            n = images.shape[0]
        out = np.zeros(n, np.float32)
        self._check(lib().gpd_hip_score(self._h, _ptr(images), n, _ptr(out)))
        return out

    def upload_cloud(self, xyz, normals, cam_source=None, view_points=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        normals = np.ascontiguousarray(normals, np.float32)
        P = len(xyz)
        cam = np.ones((1, P), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, P)
        vp = np.zeros((1, 3)) if view_points is None else np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
        self._check(lib().gpd_hip_upload_cloud(self._h, _ptr(xyz), _ptr(normals), P, _ptr(cam), cam.shape[0], _ptr(vp)))
        self._num_points = P

    def search(self, sample_indices):
        """generateGraspCandidateSets -> hands[n_sets, n_slots]."""
        si = np.ascontiguousarray(sample_indices, np.int32)
        hands = np.empty((len(si), self.n_slots), HAND_DTYPE)  # rows [0, num_sets) are written by the call
        ns = C.c_int(0)
        self._check(lib().gpd_hip_search(self._h, _ptr(si), len(si), _ptr(hands), C.byref(ns)))
        return hands[: ns.value].copy()

    def search_samples(self, samples_xyz):
        """generateGraspCandidateSets for samples given by coordinates (f64 [S,3])."""
        sm = np.ascontiguousarray(samples_xyz, np.float64).reshape(-1, 3)
        hands = np.empty((len(sm), self.n_slots), HAND_DTYPE)
        ns = C.c_int(0)
        self._check(lib().gpd_hip_search_samples(self._h, _ptr(sm), len(sm), _ptr(hands), C.byref(ns)))
        return hands[: ns.value].copy()

    def detect_samples(self, samples_xyz):
        sm = np.ascontiguousarray(samples_xyz, np.float64).reshape(-1, 3)
        hands = np.empty((len(sm), self.n_slots), HAND_DTYPE)
        ns, nc = C.c_int(0), C.c_int(0)
        self._check(lib().gpd_hip_detect_samples(self._h, _ptr(sm), len(sm), _ptr(hands), C.byref(ns), C.byref(nc)))
        return hands[: ns.value].copy(), nc.value

    def reevaluate(self, hands):
        """HandSearch::reevaluateHypotheses on the uploaded cloud -> (labels int32 [n], rewritten hands [n])."""
        h = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1).copy()
        labels = np.zeros(len(h), np.int32)
        self._check(lib().gpd_hip_reevaluate(self._h, _ptr(h), len(h), _ptr(labels)))
        return labels, h

    def images(self, hands, download=True):
        """ImageGenerator::createImages -> (images[n,60,60,C] or None, cand_index[n])."""
        hands = np.ascontiguousarray(hands)
        nv = int(hands["valid"].astype(bool).sum())
        Cn = self.params.image_num_channels
        img = np.zeros((nv, 60, 60, Cn), np.uint8) if download else None
        cand = np.zeros(max(nv, 1), np.int32)
        n = C.c_int(0)
        self._check(lib().gpd_hip_images(self._h, _ptr(hands), hands.shape[0], _ptr(img), _ptr(cand), C.byref(n)))
        assert n.value == nv
        return img, cand[:nv]

    def detect(self, sample_indices):
        """detectGrasps steps 1-4 -> (hands[n_sets, n_slots] with scores, n_candidates)."""
        si = np.ascontiguousarray(sample_indices, np.int32)
        hands = np.empty((len(si), self.n_slots), HAND_DTYPE)  # rows [0, num_sets) are written by the call
        ns, nc = C.c_int(0), C.c_int(0)
        self._check(lib().gpd_hip_detect(self._h, _ptr(si), len(si), _ptr(hands), C.byref(ns), C.byref(nc)))
        return hands[: ns.value], nc.value

    def detect_select(self, sample_indices, num_selected=0):
        """detectGrasps steps 1-4 (+ selectGrasps when num_selected > 0) -> (hands[k], n_sets, n_candidates):
        only the scored candidates (or the num_selected best, score descending) come back."""
        si = np.ascontiguousarray(sample_indices, np.int32)
        cap = len(si) * self.n_slots if num_selected == 0 else min(num_selected, len(si) * self.n_slots)
        hands = np.empty(max(cap, 1), HAND_DTYPE)
        ns, nc, nh = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(lib().gpd_hip_detect_select(self._h, _ptr(si), len(si), int(num_selected), _ptr(hands), cap,
                                                C.byref(ns), C.byref(nc), C.byref(nh)))
        return hands[: nh.value], ns.value, nc.value

    def _jobs(self, clouds, samples, num_selected):
        jobs = (DetectJob * len(clouds))()
        keep = []
        for j, cl, si in zip(jobs, clouds, samples):
            xyz = np.ascontiguousarray(cl["xyz"], np.float32)
            nrm = np.ascontiguousarray(cl["normals"], np.float32)
            P = len(xyz)
            cam = np.ascontiguousarray(cl["cam_source"], np.int32).reshape(-1, P)
            vp = np.ascontiguousarray(cl["view_points"], np.float64).reshape(-1, 3)
            si = np.ascontiguousarray(si, np.int32)
            cap = len(si) * self.n_slots if num_selected == 0 else min(num_selected, len(si) * self.n_slots)
            hands = np.empty(max(cap, 1), HAND_DTYPE)
            keep.append((xyz, nrm, cam, vp, si, hands))
            j.xyz, j.normals, j.cam_source, j.view_points = _ptr(xyz), _ptr(nrm), _ptr(cam), _ptr(vp)
            j.sample_indices, j.hands = _ptr(si), _ptr(hands)
            j.num_points, j.num_cams, j.num_samples = P, cam.shape[0], len(si)
            j.num_selected, j.hands_capacity = int(num_selected), cap
        return jobs, keep

    def raw_batch(self, scans, samples_xyz, workspace=None, voxel_size=0.003, normals_radius=0.03, num_selected=0):
        """The job array of gpd_hip_detect_batch for RAW scans (dicts with xyz, cam_source, view_points): preprocessPointCloud
        (workspace cut, voxeliser, normals) on the device, search at the given sample coordinates (one f64 [S, 3] array per scan)."""
        jobs = (DetectJob * len(scans))()
        keep = []
        ws = None if workspace is None else np.ascontiguousarray(workspace, np.float64)
        for j, cl, sm in zip(jobs, scans, samples_xyz):
            xyz = np.ascontiguousarray(cl["xyz"], np.float32)
            P = len(xyz)
            cam = np.ascontiguousarray(cl["cam_source"], np.int32).reshape(-1, P)
            vp = np.ascontiguousarray(cl["view_points"], np.float64).reshape(-1, 3)
            sm = np.ascontiguousarray(sm, np.float64).reshape(-1, 3)
            cap = len(sm) * self.n_slots if num_selected == 0 else min(num_selected, len(sm) * self.n_slots)
            hands = np.empty(max(cap, 1), HAND_DTYPE)
            keep.append((xyz, None, cam, vp, sm, hands, ws))
            j.xyz, j.cam_source, j.view_points, j.hands = _ptr(xyz), _ptr(cam), _ptr(vp), _ptr(hands)
            j.sample_xyz, j.workspace = _ptr(sm), _ptr(ws)
            j.num_points, j.num_cams, j.num_samples = P, cam.shape[0], len(sm)
            j.num_selected, j.hands_capacity = int(num_selected), cap
            j.raw, j.voxel_size, j.normals_radius = 1, float(voxel_size), float(normals_radius)
        return jobs, keep

    def batch(self, clouds, samples, num_selected=0):
        """The job array of gpd_hip_detect_batch with its input views and output buffers, built once and reusable: a
        caller that runs batch after batch (bench.py's passes) allocates nothing per call."""
        return self._jobs(clouds, samples, num_selected)

    def run_batch(self, batch):
        """gpd_hip_detect_batch on a prepared batch -> list of (hands[k], n_sets, n_candidates, stage_ms[3]) in cloud
        order; the hands are views into the batch's own buffers (overwritten by the next run)."""
        jobs, keep = batch
        self._check(lib().gpd_hip_detect_batch(self._h, jobs, len(jobs)))
        # where the host was per cloud (ms since entry) and the buffer growths booked on it: kept for the caller that asks
        self.last_batch_timeline = [([float(x) for x in j.host_ms], int(j.allocs)) for j in jobs]
        return [(k[5][: j.num_hands], j.num_sets, j.num_candidates, [float(x) for x in j.stage_ms])
                for j, k in zip(jobs, keep)]

    def detect_batch(self, clouds, samples, num_selected=0):
        """detect_grasps over independent clouds (two in flight per context).  clouds: dicts with xyz, normals,
        cam_source, view_points; samples: one int32 index array per cloud.
        -> list of (hands[k], n_sets, n_candidates, stage_ms[3]) in cloud order."""
        return self.run_batch(self.batch(clouds, samples, num_selected))

    def reserve(self, max_points, max_cams=1, max_samples=0, max_candidates=0, max_selected=0):
        """gpd_hip_reserve: size every buffer of the context once (no allocation in later calls within these sizes)."""
        self._check(lib().gpd_hip_reserve(self._h, int(max_points), int(max_cams), int(max_samples), int(max_candidates), int(max_selected)))

    def detect_batch_multi(self, others, clouds, samples, num_selected=0):
        """gpd_hip_detect_batch_multi over this context and `others` (one host thread per context, cloud i ->
        context i mod G).  Same return value as detect_batch."""
        ctxs = [self] + list(others)
        jobs, keep = self._jobs(clouds, samples, num_selected)
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        self._check(lib().gpd_hip_detect_batch_multi(arr, len(ctxs), jobs, len(clouds)))
        return [(k[5][: j.num_hands], j.num_sets, j.num_candidates, [float(x) for x in j.stage_ms])
                for j, k in zip(jobs, keep)]

    def detect_sharded(self, others, cloud, samples, split=None):
        """gpd_hip_detect_sharded: ONE cloud, its samples cut into len(others) + 1 contiguous ranges (at the indices `split`, default
        equal shares), range g on context g.  -> (hands of all ranges concatenated, per-shard (n_sets, n_candidates, lcg_base,
        lcg_draws))."""
        ctxs = [self] + list(others)
        G = len(ctxs)
        si = np.ascontiguousarray(samples, np.int32)
        cuts = [0] + (list(split) if split is not None else [len(si) * g // G for g in range(1, G)]) + [len(si)]
        parts = [si[cuts[g]:cuts[g + 1]] for g in range(G)]
        jobs, keep = self._jobs([cloud] * G, parts, 0)
        arr = (C.c_void_p * G)(*[c._h for c in ctxs])
        self._check(lib().gpd_hip_detect_sharded(arr, G, jobs))
        hands = np.concatenate([k[5][: j.num_hands] for j, k in zip(jobs, keep)])
        return hands, [(j.num_sets, j.num_candidates, int(j.lcg_base), int(j.lcg_draws)) for j in jobs]

    def stage_ms(self):
        ms = np.zeros(3, np.float32)
        self._check(lib().gpd_hip_last_stage_ms(self._h, _ptr(ms)))
        return ms

    def replay(self, stages=3):
        """Re-run images (1) and/or LeNet (2) on the device-resident candidate list (async)."""
        self._check(lib().gpd_hip_replay(self._h, int(stages)))

    def replay_times(self, n_scores=0):
        """Synchronise -> (image_ms_sum, lenet_ms_sum, launches, scores or None)."""
        ms = np.zeros(2, np.float32)
        n = C.c_int(0)
        sc = np.zeros(n_scores, np.float32) if n_scores else None
        self._check(lib().gpd_hip_replay_times(self._h, _ptr(ms), C.byref(n), _ptr(sc)))
        return float(ms[0]), float(ms[1]), n.value, sc

    def replay_kernel_ms(self):
        """Summed HIP-event time of conv1, conv2, ip1, ip2 over the replays of the last replay_times()."""
        ms = np.zeros(4, np.float32)
        self._check(lib().gpd_hip_replay_kernel_ms(self._h, _ptr(ms)))
        return [float(x) for x in ms]

    def conv1_stats(self, reset=True):
        """(executed, looked-at) (chunk, channel) pairs of conv1's launches since the last reset."""
        out = np.zeros(2, np.uint64)
        self._check(lib().gpd_hip_conv1_stats(self._h, _ptr(out), int(bool(reset))))
        return int(out[0]), int(out[1])

    def images_stats(self):
        out = np.zeros(4, np.int64)
        self._check(lib().gpd_hip_last_images_stats(self._h, _ptr(out)))
        return dict(candidates=int(out[0]), sets=int(out[1]), sum_set_ni=int(out[2]), sum_cand_ni=int(out[3]))

    def fallbacks(self):
        """Slow paths of the last search / image stage: list capacity, candidates redone by the large shadow / points
        kernels, LeNet passes."""
        out = np.zeros(4, np.int64)
        self._check(lib().gpd_hip_last_fallbacks(self._h, _ptr(out)))
        return dict(neighbourhood_list_capacity=int(out[0]), large_shadow_kernel_candidates=int(out[1]),
                    large_points_kernel_candidates=int(out[2]), lenet_passes=int(out[3]))

    def centre_chains(self):
        """(sample, coordinate) pairs of the last search whose neighbourhood centre took the serial fp64 chain (the order-free sum
        inside the neighbourhood kernel could not be certified exact)."""
        out = C.c_longlong(0)
        self._check(lib().gpd_hip_last_centre_chains(self._h, C.byref(out)))
        return int(out.value)

    def find_clusters(self, hands, scores, min_inliers=1, remove_inliers=False):
        """Clustering::findClusters on the device -> (cluster records, scores f64, seed index)."""
        hands = np.ascontiguousarray(hands, HAND_DTYPE).reshape(-1)
        scores = np.ascontiguousarray(scores, np.float64)
        assert len(scores) == len(hands)
        n = len(hands)
        out = np.zeros(max(n, 1), HAND_DTYPE)
        osc = np.zeros(max(n, 1), np.float64)
        src = np.zeros(max(n, 1), np.int32)
        k = C.c_int(0)
        self._check(lib().gpd_hip_find_clusters(self._h, _ptr(hands), _ptr(scores), n, int(min_inliers), int(bool(remove_inliers)), _ptr(out),
                                                _ptr(osc), _ptr(src), C.byref(k)))
        return out[: k.value].copy(), osc[: k.value].copy(), src[: k.value].copy()

    def preprocess_cloud(self, xyz, cam_source=None, workspace=None, voxel_size=0.003):
        """Cloud::filterWorkspace (points) + Cloud::voxelizeCloud on the device ->
        (xyz f32 [M,3], cam_source i32 [cams,M], input index per output point i32 [M], kernel ms)."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        P = len(xyz)
        cam = np.zeros((0, P), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, P)
        ws = None if workspace is None else np.ascontiguousarray(workspace, np.float64)
        assert ws is None or ws.shape == (6,)
        out = np.zeros((P, 3), np.float32)
        cam_out = np.zeros((cam.shape[0], P), np.int32).reshape(-1)
        src = np.zeros(P, np.int32)
        n, ms = C.c_int(0), C.c_float(0)
        self._check(lib().gpd_hip_preprocess_cloud(self._h, _ptr(xyz) if P else None, _ptr(cam) if cam.size else None, P, cam.shape[0],
                                                   _ptr(ws) if ws is not None else None, float(voxel_size), _ptr(out) if P else None,
                                                   _ptr(cam_out) if cam.size else None, _ptr(src) if P else None, C.byref(n), C.byref(ms)))
        M = n.value
        return out[:M].copy(), cam_out[: cam.shape[0] * M].reshape(cam.shape[0], M).copy(), src[:M].copy(), ms.value

    def estimate_normals(self, radius=0.03):
        """Cloud::calculateNormals on the uploaded cloud -> f32 [P,3] (also kept on the device)."""
        P = self._num_points
        out = np.zeros((P, 3), np.float32)
        self._check(lib().gpd_hip_estimate_normals(self._h, float(radius), _ptr(out)))
        return out
