"""Independent pure-Python/numpy restatement of the grasp-image rasteriser, written from
the reference sources (descriptor/image_strategy.cpp:32-243, image_15_channels_strategy.cpp,
candidate/hand_set.cpp:118-283) without sharing code with oracle/gpd_oracle.cpp.  Slow;
used on a handful of candidates to cross-check the C++ oracle."""
import math

import numpy as np

f32 = np.float32


def radius_neighbours(xyz, q, radius):
    """FLANN semantics: float d2 accumulated over x,y,z; strict <; sorted by (d2, index)."""
    d = (q[None, :].astype(f32) - xyz.astype(f32))
    d2 = f32(0) + d[:, 0] * d[:, 0]
    d2 = d2 + d[:, 1] * d[:, 1]
    d2 = d2 + d[:, 2] * d[:, 2]
    r2 = f32(radius * radius)
    idx = np.flatnonzero(d2 < r2)
    order = np.lexsort((idx, d2[idx]))
    return idx[order]


def _to_unit(F, sample, bottom, center, pts, P):
    depth, width, height = P.volume_depth, P.volume_width, P.volume_height
    half = width / 2.0
    out_idx, out_u = [], []
    for i in range(len(pts)):
        c = pts[i] - sample
        t = [F[0, r] * c[0] + F[1, r] * c[1] + F[2, r] * c[2] for r in range(3)]
        if (t[0] > bottom and t[0] < bottom + depth and t[1] > center - half and t[1] < center + half
                and t[2] > -1.0 * height and t[2] < height):
            out_idx.append(i)
            out_u.append([(t[0] - bottom) / depth, (t[1] - (center - half)) / width, (t[2] + height) / (2.0 * height)])
    return out_idx, np.array(out_u, np.float64).reshape(-1, 3)


def _cells(ua, ub):
    cs = 1.0 / 60.0
    return [min(int(math.floor(b / cs)), 59) + 60 * min(int(math.floor(a / cs)), 59) for a, b in zip(ua, ub)]


def _post(img):
    """3x3 max dilate (border ignored), min-max normalise, u8 (round half even)."""
    H = img.reshape(60, 60, -1).astype(f32)
    pad = np.full((62, 62, H.shape[2]), -np.inf, f32)
    pad[1:61, 1:61] = H
    d = np.max(np.stack([pad[1 + dr:61 + dr, 1 + dc:61 + dc] for dr in (-1, 0, 1) for dc in (-1, 0, 1)]), axis=0)
    smin, smax = float(d.min()), float(d.max())
    scale = 1.0 * (1.0 / (smax - smin) if (smax - smin) > np.finfo(np.float64).eps else 0.0)
    shift = 0.0 - smin * scale
    v = d * f32(scale) + f32(shift)
    u = v * f32(255.0) + f32(0.0)
    return np.clip(np.rint(u), 0, 255).astype(np.uint8)


def _normals_image(nrm, cells):
    img = np.zeros((3600, 3), f32)
    for i, idx in enumerate(cells):
        row, col = 59 - idx // 60, idx % 60
        v = img[row * 60 + col]
        a = np.array([f32(abs(nrm[i][0])), f32(abs(nrm[i][1])), f32(abs(nrm[i][2]))], f32)
        if v[0] == 0 and v[1] == 0 and v[2] == 0:
            img[row * 60 + col] = a
        else:
            s = np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
            inv = 1.0 / float(s)
            for c in range(3):
                dlt = f32(a[c] - v[c])
                img[row * 60 + col, c] = f32(v[c] + f32(float(dlt) * inv))
    return _post(img)


def _depth_image(depth, cells):
    img = np.zeros(3600, f32)
    avgs = np.zeros(3600, f32)
    cnt = np.zeros(3600, f32)
    for i, idx in enumerate(cells):
        row, col = 59 - idx // 60, idx % 60
        cnt[idx] = f32(float(cnt[idx]) + 1.0)
        avgs[idx] = f32(float(avgs[idx]) + (depth[i] - float(avgs[idx])) * (1.0 / float(cnt[idx])))
        img[row * 60 + col] = f32(1.0 - float(avgs[idx]))
    return _post(img)[..., 0]


def _shadow_image(depth, cells):
    img = np.zeros(3600, f32)
    cnt = np.zeros(3600, f32)
    nz = np.zeros(3600, bool)
    for i, idx in enumerate(cells):
        p = (59 - idx // 60) * 60 + idx % 60
        cnt[idx] = f32(float(cnt[idx]) + 1.0)
        img[p] = f32(float(img[p]) + (depth[i] - float(img[p])) * (1.0 / float(cnt[idx])))
        nz[p] = True
    mx = float(img[nz].max()) if nz.any() else 0.0
    img = np.where(nz, f32(mx), f32(0)).astype(f32) - img
    return _post(img)[..., 0]


class Lcg:
    def __init__(self, s=0):
        self.s = s

    def next(self):
        self.s = (214013 * self.s + 2531011) & 0xFFFFFFFF
        return (self.s >> 16) & 0x7FFF


def shadow_voxels(pts, view_point, rng, shadow_length=0.10):
    """calculateShadow for one camera -> lexicographically sorted unique voxel array."""
    pts = np.asarray(pts, np.float64)
    center = np.zeros(3)
    for p in pts:
        center = center + p
    center = center / float(len(pts))
    vec = center - np.asarray(view_point, np.float64)
    vec = shadow_length * vec / math.sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2])
    n_sh = int(math.floor(shadow_length / 0.003))
    mult = 1.0 / 0.003
    out = set()
    for i in range(len(pts) * n_sh):
        t = float(rng.next()) * (1.0 / 32767.0)
        q = (pts[i // n_sh] + t * vec) * mult
        out.add((int(q[0]), int(q[1]), int(q[2])))  # int() truncates toward zero like .cast<int>()
    return sorted(out)


def shadow_voxels_cameras(pts, cam_rows, view_points, rng, shadow_length=0.10):
    """HandSet::calculateShadow for several cameras (hand_set.cpp:118-185): every camera that sees at
    least one neighbourhood point casts the shadow of ALL the points from its view point (in camera
    order, consuming LCG draws); one camera: its set; several: camera 0's set (empty when camera 0
    sees nothing) intersected with the sets of the other seeing cameras."""
    cam_rows = np.asarray(cam_rows)
    n_cams = cam_rows.shape[0]
    sets = []
    for c in range(n_cams):
        sets.append(set(shadow_voxels(pts, view_points[c], rng, shadow_length)) if cam_rows[c].sum() >= 1 else set())
    if n_cams == 1:
        return sorted(sets[0])
    allv = sets[0]
    for c in range(1, n_cams):
        if cam_rows[c].sum() >= 1:
            allv = allv & sets[c]
    return sorted(allv)


def grasp_image(P, hand, xyz, normals, nbr_idx, shadow_vox=None):
    """One candidate's image [60,60,C] from its 0.10 m neighbourhood (in neighbour order)."""
    C = P.image_num_channels
    F = np.asarray(hand["frame"], np.float64).reshape(3, 3)
    sample = np.asarray(hand["sample"], np.float64)
    pts = xyz[nbr_idx].astype(np.float64)
    idx, u = _to_unit(F, sample, float(hand["bottom"]), float(hand["center"]), pts, P)
    nrm = []
    for i in idx:
        n = normals[nbr_idx[i]].astype(np.float64)
        nrm.append([F[0, r] * n[0] + F[1, r] * n[1] + F[2, r] * n[2] for r in range(3)])
    su = np.zeros((0, 3))
    if C == 15:
        sp = np.array(shadow_vox, np.float64).reshape(-1, 3) * 0.003
        _, su = _to_unit(F, sample, float(hand["bottom"]), float(hand["center"]), sp, P)
    perm = [(0, 1, 2), (2, 1, 0), (2, 0, 1)]
    nproj = 1 if C <= 3 else 3
    per = {15: 5, 12: 4, 3: 3, 1: 1}[C]
    img = np.zeros((60, 60, C), np.uint8)
    for pr in range(nproj):
        a, b, d = perm[pr]
        cells = _cells(u[:, a], u[:, b]) if len(u) else []
        if C == 1:  # image_1_channels_strategy.cpp:25-40: the depth image alone
            img[..., 0] = _depth_image(u[:, d] if len(u) else [], cells)
            break
        img[..., pr * per:pr * per + 3] = _normals_image(nrm, cells)
        if C >= 12:
            img[..., pr * per + 3] = _depth_image(u[:, d] if len(u) else [], cells)
        if C == 15:
            sc = _cells(su[:, a], su[:, b]) if len(su) else []
            img[..., pr * per + 4] = _shadow_image(su[:, d] if len(su) else [], sc)
    return img
