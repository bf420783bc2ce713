"""Independent numpy restatement of the candidate evaluation, written from the reference sources:
HandSet::evalHandSet / evalHands (candidate/hand_set.cpp:31-116), modifyCandidate / labelHypothesis
(:235-261), PointList::transformToHandFrame / cropByHandHeight (util/point_list.cpp:22-55),
FingerHand (candidate/finger_hand.cpp:6-184), Hand::construct (candidate/hand.cpp:24-45) and
Antipodal::evaluateGrasp (candidate/antipodal.cpp:10-96).

Unlike the oracle it MATERIALISES the list cropByHandHeight really returns — N columns, the k
in-height points followed by N-k copies of column 0 — instead of a ghost point with a
multiplicity, so it checks that modelling too.  Test infrastructure only."""
import numpy as np


def _mat3mul(a, b):
    c = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            c[i, j] = a[i, 0] * b[0, j] + a[i, 1] * b[1, j] + a[i, 2] * b[2, j]
    return c


class FingerHand:
    def __init__(self, fw, od, depth, n):
        self.fw, self.depth, self.n = fw, depth, n
        step = ((od - fw) - 0.0) / (n - 1)  # Eigen LinSpaced: low + i * step, the last entry is the upper bound
        half = np.array([0.0 + i * step for i in range(n)])
        half[-1] = od - fw  # Eigen LinSpaced hits the upper bound exactly
        self.spacing = np.concatenate([half - od + fw, half])
        self.fingers = np.zeros(2 * n, bool)
        self.hand = np.zeros(n, bool)
        self.top = self.bottom = self.center = 0.0

    def copy(self):
        o = FingerHand.__new__(FingerHand)
        o.__dict__ = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()}
        return o

    def evaluate_fingers(self, pts, bite, idx=-1):
        self.top, self.bottom, self.center = bite, bite - self.depth, 0.0
        self.fingers[:] = False
        cropped = []
        for i in range(pts.shape[1]):
            if pts[0, i] < bite:
                if pts[0, i] < self.bottom:
                    return
                cropped.append(i)
        if not cropped:
            return
        y = pts[1, cropped]

        def gap_free(s):
            return not np.any((y > self.spacing[s]) & (y < self.spacing[s] + self.fw))

        if idx == -1:
            for s in range(2 * self.n):
                self.fingers[s] = gap_free(s)
        else:
            self.fingers[idx] = gap_free(idx)
            self.fingers[self.n + idx] = gap_free(self.n + idx)

    def evaluate_hand(self):
        self.hand = self.fingers[: self.n] & self.fingers[self.n:]

    def choose_middle(self):
        idx = np.flatnonzero(self.hand)
        return -1 if len(idx) == 0 else int(idx[int(np.ceil(len(idx) / 2.0)) - 1])

    def deepen(self, pts, min_depth, max_depth):
        mid = self.choose_middle()
        new, last = self.copy(), self.copy()
        d = min_depth + 0.005
        while d <= max_depth:
            new.evaluate_fingers(pts, d, mid)
            if not new.fingers[mid] or not new.fingers[self.n + mid]:
                break
            self.hand[mid] = True
            last = new.copy()
            d += 0.005
        self.__dict__ = last.copy().__dict__
        self.hand = np.zeros(self.n, bool)
        self.hand[mid] = True
        return mid

    def closing_region(self, pts, idx):
        left, right = self.spacing[idx] + self.fw, self.spacing[self.n + idx]
        self.center = 0.5 * (left + right)
        return np.flatnonzero((pts[0] > self.bottom) & (pts[0] < self.top) & (pts[1] > left) & (pts[1] < right))


def antipodal(pts, nrm, friction_deg, min_viable):
    cosf = np.cos(friction_deg * np.pi / 180.0)
    min_x, max_x = pts[1].min() + 0.003, pts[1].max() - 0.003
    left = np.flatnonzero((-nrm[1] > cosf) & (pts[1] < min_x))
    right = np.flatnonzero((nrm[1] > cosf) & (pts[1] > max_x))
    res = 1 if (len(left) or len(right)) else 0
    if len(left) and len(right):
        L, R = pts[:, left], pts[:, right]
        ty, by = min(L[0].max(), R[0].max()), max(L[0].min(), R[0].min())
        tz, bz = min(L[2].max(), R[2].max()), max(L[2].min(), R[2].min())
        nl = int(np.sum((L[0] >= by) & (L[0] <= ty) & (L[2] >= bz) & (L[2] <= tz)))
        nr = int(np.sum((R[0] >= by) & (R[0] <= ty) & (R[2] >= bz) & (R[2] <= tz)))
        if nl >= min_viable and nr >= min_viable:
            res = 2
    return res


def eval_hand_set(P, xyz, normals, nbr_idx, frame12, angle_axis):
    """-> list over slots of dict(valid, fidx, top, bottom, center, width, half, full, position, frame)."""
    sample = np.asarray(frame12[:3], np.float64)
    F = np.stack([frame12[3:6], frame12[6:9], frame12[9:12]], axis=1)  # columns normal | binormal | curvature
    RB = angle_axis(np.pi, [0.0, 1.0, 0.0])
    n_or = P.num_orientations
    angles = [-np.pi / 2.0 + i * (np.pi / n_or) for i in range(n_or)]
    axes = np.eye(3)
    c = xyz[nbr_idx].astype(np.float64) - sample
    nd = normals[nbr_idx].astype(np.float64)
    out = []
    for ai in range(P.num_hand_axes):
        fh = FingerHand(P.finger_width, P.hand_outer_diameter, P.hand_depth, P.num_finger_placements)
        for oi in range(n_or):
            FR = _mat3mul(_mat3mul(F, RB), angle_axis(angles[oi], axes[P.hand_axes[ai]]))
            pf = np.stack([FR[0, r] * c[:, 0] + FR[1, r] * c[:, 1] + FR[2, r] * c[:, 2] for r in range(3)])
            nf = np.stack([FR[0, r] * nd[:, 0] + FR[1, r] * nd[:, 1] + FR[2, r] * nd[:, 2] for r in range(3)])
            # cropByHandHeight: `indices` has N entries, only the first k are written
            N = pf.shape[1]
            inh = np.flatnonzero((pf[2] > -1.0 * P.hand_height) & (pf[2] < P.hand_height))
            idx = np.zeros(N, int)
            idx[: len(inh)] = inh
            pts, nrm = pf[:, idx], nf[:, idx]
            fh.evaluate_fingers(pts, P.init_bite)
            fh.evaluate_hand()

            def construct():
                first = np.flatnonzero(fh.hand)
                pos = np.array([FR[r, 0] * fh.bottom + FR[r, 1] * fh.center + FR[r, 2] * 0.0 for r in range(3)]) + sample
                return dict(top=fh.top, bottom=fh.bottom, center=fh.center, fidx=int(first[0]) if len(first) else -1, position=pos)

            h = dict(valid=False, width=0.0, half=False, full=False, frame=FR.copy(), **construct())
            if fh.hand.any():
                fidx = fh.deepen(pts, P.init_bite, P.hand_depth) if P.deepen_hand else fh.choose_middle()
                closing = fh.closing_region(pts, fidx)
                if len(closing):
                    h.update(construct())
                    h["valid"] = True
                    h["width"] = pts[1, closing].max() - pts[1, closing].min()
                    lab = antipodal(pts[:, closing], nrm[:, closing], P.friction_coeff, P.min_viable)
                    h["half"], h["full"] = lab >= 1, lab == 2
            out.append(h)
    return out
