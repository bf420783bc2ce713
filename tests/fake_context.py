"""A stand-in for gpd_amd.api.Context that answers from the CPU oracle — TEST INFRASTRUCTURE for tests/test_bench_dryrun.py only.

bench.py cannot run without a GPU, and a slip in the part of it that turns measurements into the JSON line would cost a round its
bench result.  The dry run executes bench.main() unchanged against this class: same calls, same shapes and dtypes of what comes
back (computed by the oracle, so the legs that compare scores see consistent numbers), made-up stage times in the proportions of
the committed round-5 line.  Nothing here is a product path: the product has no CPU fallback (tests/test_capi.py).
"""
import ctypes as C

import numpy as np

import oracle
from gpd_amd import api

STEP_MS = dict(images=1.07, conv1=0.41, conv2=0.72, ip1=0.22, ip2=0.03)  # per 5000 candidates, profiles/r05_bench_default.json


def _view(addr, dtype, shape):
    """numpy view of the caller's buffer behind a job's pointer field"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_char * n).from_address(addr), dtype=dtype).reshape(shape)


class FakeLib:
    """libgpd_hip.so with gpd_hip_detect_batch answered by the oracle (everything else: the real library's host-side symbols,
    e.g. gpd_hip_default_params)."""

    def __init__(self, real):
        self._real = real
        self._cache = {}

    def __getattr__(self, name):
        return getattr(self._real, name)

    def gpd_hip_detect_batch(self, handle, jobs, n):
        ctx = FakeContext.live[int(handle)]
        tl = 0.0
        for i in range(n):
            j = jobs[i]
            P, cams, S = j.num_points, j.num_cams, j.num_samples
            xyz = _view(j.xyz, np.float32, (P, 3))
            if j.raw:  # counters only: the voxeliser's count is the oracle's, the candidate count a plausible number
                key = ("raw", j.xyz)
                if key not in self._cache:
                    self._cache[key] = len(oracle.voxelize(xyz, j.voxel_size)[0])
                j.num_points_processed = self._cache[key]
                j.num_sets, j.num_candidates, j.num_hands = S, 2 * S, 2 * S
            else:
                key = (j.xyz, j.sample_indices, S)
                if key not in self._cache:
                    hands, nc, _ = oracle.detect(ctx.p, xyz, _view(j.normals, np.float32, (P, 3)), _view(j.cam_source, np.int32, (cams, P)),
                                                 _view(j.view_points, np.float64, (cams, 3)), _view(j.sample_indices, np.int32, (S,)), ctx.w)
                    flat = hands.reshape(-1)
                    self._cache[key] = (flat[flat["valid"].astype(bool)].copy(), len(hands), nc)
                scored, ns, nc = self._cache[key]
                assert j.num_selected == 0 and len(scored) <= j.hands_capacity
                _view(j.hands, api.HAND_DTYPE, (j.hands_capacity,))[: len(scored)] = scored
                j.num_sets, j.num_candidates, j.num_hands = ns, nc, len(scored)
            j.status, j.allocs = 0, 0
            for k, v in enumerate((0.55, 1.4, 1.8)):
                j.stage_ms[k] = v
            for k, v in enumerate((0.4, 1.0, 1.3, 4.0, 4.3)):  # begin done, plan arrived, middle enqueued, results arrived, records copied
                j.host_ms[k] = tl + v
            tl += 4.5
        return 0


class FakeContext:
    live = {}  # handle -> context, for FakeLib

    def __init__(self, params=None, device=0):
        self.params = params if params is not None else api.default_params()
        self.C = self.params.image_num_channels
        self.p = oracle.default_params(self.C)
        self.n_slots = self.params.num_hand_axes * self.params.num_orientations
        self.mode = api.LENET_SPLIT
        self.w = None
        self.cl = None
        self._imgs = None
        self._launches = 0
        self._last_launches = 0
        self._detect_cache = {}
        self.last_batch_timeline = []
        self._h = 1000 + len(FakeContext.live)
        FakeContext.live[self._h] = self

    # the job arrays and their bookkeeping are the product binding's own host code
    _jobs = api.Context._jobs
    batch = api.Context.batch
    raw_batch = api.Context.raw_batch
    run_batch = api.Context.run_batch

    def _check(self, rc):
        assert rc == 0, rc

    # -- state
    def close(self):
        FakeContext.live.pop(self._h, None)

    def set_lenet_weights(self, w):
        assert np.asarray(w["c1w"]).size == 20 * self.C * 25
        self.w = w

    def set_lenet_mode(self, mode):
        assert mode in (api.LENET_SPLIT, api.LENET_F32_CHAIN)
        self.mode = mode

    def upload_cloud(self, xyz, normals, cam_source=None, view_points=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        P = len(xyz)
        cam = np.ones((1, P), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, P)
        vp = np.zeros((1, 3)) if view_points is None else np.ascontiguousarray(view_points, np.float64).reshape(-1, 3)
        self.cl = dict(xyz=xyz, normals=np.ascontiguousarray(normals, np.float32), cam=cam, vp=vp)
        self._num_points = P
        self._detect_cache = {}

    # -- the path
    def search(self, sample_indices):
        return oracle.search(self.p, self.cl["xyz"], self.cl["normals"], np.ascontiguousarray(sample_indices, np.int32))

    def stage_ms(self):
        return np.array([0.55, 1.4, 1.8], np.float32)

    def detect(self, sample_indices):
        si = np.ascontiguousarray(sample_indices, np.int32)
        key = si.tobytes()
        if key not in self._detect_cache:
            hands, n, _ = oracle.detect(self.p, self.cl["xyz"], self.cl["normals"], self.cl["cam"], self.cl["vp"], si, self.w)
            self._detect_cache[key] = (hands, n)
        hands, n = self._detect_cache[key]
        return hands.copy(), n

    def images(self, hands, download=True):
        hands = np.ascontiguousarray(hands)
        img, cand = oracle.images(self.p, self.cl["xyz"], self.cl["normals"], self.cl["cam"], self.cl["vp"], hands)
        self._imgs = img
        self._sets = int(hands["valid"].astype(bool).any(axis=1).sum())
        return (img if download else None), cand

    def images_stats(self):
        n = len(self._imgs)
        return dict(candidates=n, sets=self._sets, sum_set_ni=3000 * self._sets, sum_cand_ni=3000 * n)

    def fallbacks(self):
        return dict(neighbourhood_list_capacity=0, large_shadow_kernel_candidates=0, large_points_kernel_candidates=0, lenet_passes=1)

    def score(self, images=None, n=None):
        imgs = self._imgs if images is None else np.ascontiguousarray(images, np.uint8)
        return oracle.lenet(imgs, self.w)

    def replay(self, stages=3):
        assert stages in (1, 2, 3) and self._imgs is not None
        self._launches += 1

    def replay_times(self, n_scores=0):
        L, self._launches = self._launches, 0
        self._last_launches = L
        scale = len(self._imgs) / 5000.0
        sc = self.score()[:n_scores] if n_scores else None
        net = STEP_MS["conv1"] + STEP_MS["conv2"] + STEP_MS["ip1"] + STEP_MS["ip2"]
        return STEP_MS["images"] * scale * L, net * scale * L, L, sc

    def replay_kernel_ms(self):
        scale = len(self._imgs) / 5000.0 * self._last_launches
        return [STEP_MS[k] * scale for k in ("conv1", "conv2", "ip1", "ip2")]

    def conv1_stats(self, reset=True):
        return 0, 0  # the split conv1 executes every tile: no live-pair counters

    # -- the rows before the path
    def preprocess_cloud(self, xyz, cam_source=None, workspace=None, voxel_size=0.003):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        ws = np.asarray(workspace, np.float64)
        inside = np.all((xyz >= ws[0::2]) & (xyz <= ws[1::2]), axis=1)
        assert inside.all()  # the synthetic scans of bench.py lie inside the cfg workspace
        v, src = oracle.voxelize(xyz, voxel_size)
        cam = np.zeros((0, len(xyz)), np.int32) if cam_source is None else np.ascontiguousarray(cam_source, np.int32).reshape(-1, len(xyz))
        return v, cam[:, src].copy(), src, 0.68

    def estimate_normals(self, radius=0.03):
        return oracle.estimate_normals(self.cl["xyz"], self.cl["cam"], self.cl["vp"], radius)
