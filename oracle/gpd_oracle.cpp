/*
 * gpd_oracle.cpp — CPU restatement of the GPD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (libgpd_hip.so) never links or calls it.
 *
 * PARITY STATUS: the IN-TREE logic of the reference is pinned — the reference's own translation units
 * (candidate/, descriptor/, net/, util/cloud, clustering, grasp_detector) are compiled UNMODIFIED through
 * test-only interface subsets of Eigen / PCL / OpenCV / Boost (oracle/shim, oracle/build_ref.sh tier A ->
 * oracle/_ref/libgpd_ref.so); what they return is committed as tests/golden/ref_pin_*.npz and this file must
 * reproduce it bit for bit (tests/test_ref_pin.py).  THIRD-PARTY NUMERICS STAY UNPINNED: the real PCL / FLANN,
 * Eigen, OpenCV and Boost are absent here, the shim restates their semantics from memory, and so does this
 * file (FLANN result order, Eigen's eigensolver and product summation order, OpenCV's rounding); the
 * reference BINARY never ran here (tier B of build_ref.sh needs the real libraries), and the reference
 * ships no golden vectors.  Also checked: the known-answer constants of SURVEY.md §9-K and
 * independent numpy/scipy/torch re-derivations (tests/test_oracle_*.py).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree).  Third-party semantics (FLANN radiusSearch, Eigen
 * SelfAdjointEigenSolver / AngleAxis / LinSpaced, OpenCV dilate / normalize /
 * convertTo) are restated from their published algorithms; the choices that the
 * reference leaves open are fixed here and are part of the oracle's definition:
 *   - every fp expression is evaluated left to right, unfused (-ffp-contract=off);
 *   - M = N*N^T and other reductions are sequential sums in neighbour order;
 *   - shadow voxels are visited in lexicographic (x,y,z) order, no jitter
 *     (hand_set.cpp:191-199 draws from std::random_device: irreproducible);
 *   - the shadow LCG (hand_set.cpp:263-266) starts at seed 0 for every cloud;
 *   - LeNet dot products are fmaf chains in ascending k, bias added last.
 *
 * Build: see oracle/Makefile (g++ -O3 -fopenmp -ffp-contract=off -mfma -mavx2).
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gpd_hip.h"

namespace {

static double now_s() {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

// ---------------------------------------------------------------------------
// Radius search — pcl::KdTreeFLANN::radiusSearch semantics (call sites
// hand_search.cpp:178, frame_estimator.cpp:74, image_generator.cpp:61).
// FLANN L2_Simple<float>: d2 accumulated in float over x,y,z; neighbour iff
// d2 < (float)(r*r); result sorted by (d2, index).
// ---------------------------------------------------------------------------
struct Grid {
  float lo[3];
  float cell;
  int dim[3];
  std::vector<int> start;  // dim[0]*dim[1]*dim[2] + 1
  std::vector<int> items;
  const float *xyz;
  int P;

  void build(const float *pts, int n, float cell_size) {
    xyz = pts;
    P = n;
    cell = cell_size;
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    lo[0] = lo[1] = lo[2] = FLT_MAX;
    for (int i = 0; i < n; i++)
      for (int c = 0; c < 3; c++) {
        lo[c] = std::min(lo[c], pts[3 * i + c]);
        hi[c] = std::max(hi[c], pts[3 * i + c]);
      }
    if (n == 0) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.f;
    for (int c = 0; c < 3; c++) dim[c] = std::max(1, (int)std::floor((hi[c] - lo[c]) / cell) + 1);
    size_t nc = (size_t)dim[0] * dim[1] * dim[2];
    start.assign(nc + 1, 0);
    std::vector<int> cid(n);
    for (int i = 0; i < n; i++) {
      cid[i] = cellOf(pts + 3 * i);
      start[cid[i] + 1]++;
    }
    for (size_t i = 0; i < nc; i++) start[i + 1] += start[i];
    items.resize(n);
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; i++) items[fill[cid[i]]++] = i;
  }
  int coord(float v, int c) const {
    int k = (int)std::floor((v - lo[c]) / cell);
    return std::min(std::max(k, 0), dim[c] - 1);
  }
  int cellOf(const float *p) const {
    return (coord(p[0], 0) * dim[1] + coord(p[1], 1)) * dim[2] + coord(p[2], 2);
  }
};

struct Neighbour {
  float d2;
  int idx;
};
static inline bool nbLess(const Neighbour &a, const Neighbour &b) {
  return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx);
}

static inline float dist2f(const float *q, const float *p) {
  float r = 0.f;
  float d = q[0] - p[0];
  r += d * d;
  d = q[1] - p[1];
  r += d * d;
  d = q[2] - p[2];
  r += d * d;
  return r;
}

static void radiusSearch(const Grid &g, const float q[3], double radius, std::vector<Neighbour> &out) {
  out.clear();
  const float r2 = (float)(radius * radius);
  const float margin = (float)radius * 1.001f + 1e-5f;
  int c0[3], c1[3];
  for (int c = 0; c < 3; c++) {
    c0[c] = g.coord(q[c] - margin, c);
    c1[c] = g.coord(q[c] + margin, c);
  }
  for (int x = c0[0]; x <= c1[0]; x++)
    for (int y = c0[1]; y <= c1[1]; y++)
      for (int z = c0[2]; z <= c1[2]; z++) {
        size_t cell = ((size_t)x * g.dim[1] + y) * g.dim[2] + z;
        for (int k = g.start[cell]; k < g.start[cell + 1]; k++) {
          int i = g.items[k];
          float d2 = dist2f(q, g.xyz + 3 * i);
          if (d2 < r2) out.push_back({d2, i});
        }
      }
  std::sort(out.begin(), out.end(), nbLess);
}

// ---------------------------------------------------------------------------
// 3x3 helpers.  Row-major double[9].  Products are sum_k a(i,k)*b(k,j) left to
// right, unfused (SURVEY §9-T "Eigen small products").
// ---------------------------------------------------------------------------
static void mat3mul(const double *a, const double *b, double *c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// Eigen::AngleAxisd(angle, axis).toRotationMatrix() (Rodrigues form, Eigen
// Geometry/AngleAxis.h) — hand_set.cpp:52-53, 68-69.
static void angleAxis(double angle, const double ax[3], double *R) {
  double s = std::sin(angle), c = std::cos(angle);
  double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  double c1[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double t;
  t = c1[0] * ax[1];
  R[1] = t - sa[2];
  R[3] = t + sa[2];
  t = c1[0] * ax[2];
  R[2] = t + sa[1];
  R[6] = t - sa[1];
  t = c1[1] * ax[2];
  R[5] = t - sa[0];
  R[7] = t + sa[0];
  R[0] = c1[0] * ax[0] + c;
  R[4] = c1[1] * ax[1] + c;
  R[8] = c1[2] * ax[2] + c;
}

// ---------------------------------------------------------------------------
// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (iterative path) —
// local_frame.cpp:17-21.  Scale, closed-form 3x3 tridiagonalisation, implicit
// symmetric QR with Wilkinson shift, ascending selection sort.  No sign fix.
// evec is row-major, columns are eigenvectors.
// ---------------------------------------------------------------------------
static void givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0 ? 1.0 : -1.0;
  } else if (std::fabs(p) > std::fabs(q)) {
    double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

static double hypotPos(double x, double y) {
  x = std::fabs(x);
  y = std::fabs(y);
  double p = std::max(x, y);
  if (p == 0.0) return 0.0;
  double qp = std::min(y, x) / p;
  return p * std::sqrt(1.0 + qp * qp);
}

static void selfAdjointEigen3(const double M[9], double eval[3], double evec[9]) {
  // lower triangle, scaled to [-1,1]
  double m00 = M[0], m10 = M[3], m11 = M[4], m20 = M[6], m21 = M[7], m22 = M[8];
  double scale = std::max({std::fabs(m00), std::fabs(m10), std::fabs(m11), std::fabs(m20), std::fabs(m21), std::fabs(m22)});
  if (scale == 0.0) scale = 1.0;
  m00 /= scale;
  m10 /= scale;
  m11 /= scale;
  m20 /= scale;
  m21 /= scale;
  m22 /= scale;
  double diag[3], sub[2];
  double Q[9];
  const double tol = DBL_MIN;
  diag[0] = m00;
  double v1norm2 = m20 * m20;
  if (v1norm2 <= tol) {
    diag[1] = m11;
    diag[2] = m22;
    sub[0] = m10;
    sub[1] = m21;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = 1; Q[5] = 0; Q[6] = 0; Q[7] = 0; Q[8] = 1;
  } else {
    double beta = std::sqrt(m10 * m10 + v1norm2);
    double invBeta = 1.0 / beta;
    double m01 = m10 * invBeta;
    double m02 = m20 * invBeta;
    double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = m01; Q[5] = m02; Q[6] = 0; Q[7] = m02; Q[8] = -m01;
  }
  const int n = 3;
  int end = n - 1, start = 0, iter = 0;
  const int maxIter = 30;
  const double considerAsZero = DBL_MIN;
  const double precision_inv = 1.0 / DBL_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; i++) {
      if (std::fabs(sub[i]) < considerAsZero) {
        sub[i] = 0.0;
      } else {
        double ss = precision_inv * sub[i];
        if (ss * ss <= (std::fabs(diag[i]) + std::fabs(diag[i + 1]))) sub[i] = 0.0;
      }
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > maxIter * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    // one implicit QR step with Wilkinson shift on [start, end]
    double td = (diag[end - 1] - diag[end]) * 0.5;
    double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
      mu -= std::fabs(e);
    } else if (e != 0.0) {
      double e2 = e * e;
      double h = hypotPos(td, e);
      if (e2 == 0.0)
        mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
      else
        mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end && z != 0.0; k++) {
      double c, s;
      givens(x, z, c, s);
      double sdk = s * diag[k] + c * sub[k];
      double dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      x = sub[k];
      if (k < end - 1) {
        z = -s * sub[k + 1];
        sub[k + 1] = c * sub[k + 1];
      }
      // Q = Q * G : columns k, k+1
      for (int i = 0; i < 3; i++) {
        double xi = Q[3 * i + k], yi = Q[3 * i + k + 1];
        Q[3 * i + k] = c * xi - s * yi;
        Q[3 * i + k + 1] = s * xi + c * yi;
      }
    }
  }
  // ascending selection sort with column swaps
  for (int i = 0; i < n - 1; i++) {
    int k = 0;
    for (int j = 1; j < n - i; j++)
      if (diag[i + j] < diag[i + k]) k = j;
    if (k > 0) {
      std::swap(diag[i], diag[k + i]);
      for (int r = 0; r < 3; r++) std::swap(Q[3 * r + i], Q[3 * r + k + i]);
    }
  }
  for (int i = 0; i < 3; i++) eval[i] = diag[i] * scale;
  std::memcpy(evec, Q, sizeof(Q));
}

// ---------------------------------------------------------------------------
// LocalFrame::findAverageNormalAxis — local_frame.cpp:14-41.
// normals: k neighbours (double, in (d2,idx) order).  Output frame = 12 doubles:
// sample(3) normal(3) binormal(3) curvature(3).
// ---------------------------------------------------------------------------
static void averageNormalAxis(const std::vector<double> &nn, double *normal, double *binormal, double *curv) {
  const int k = (int)nn.size() / 3;
  double M[9] = {0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int n = 0; n < k; n++) s += nn[3 * n + i] * nn[3 * n + j];
      M[3 * i + j] = s;
    }
  double eval[3], evec[9];
  selfAdjointEigen3(M, eval, evec);
  // minCoeff / maxCoeff: first occurrence of the extreme value
  int mn = 0, mx = 0;
  for (int i = 1; i < 3; i++) {
    if (eval[i] < eval[mn]) mn = i;
    if (eval[i] > eval[mx]) mx = i;
  }
  for (int r = 0; r < 3; r++) {
    curv[r] = evec[3 * r + mn];
    normal[r] = evec[3 * r + mx];
  }
  double avg[3] = {0, 0, 0};
  for (int n = 0; n < k; n++)
    for (int r = 0; r < 3; r++) avg[r] += nn[3 * n + r];
  double nrm = std::sqrt(avg[0] * avg[0] + avg[1] * avg[1] + avg[2] * avg[2]);
  for (int r = 0; r < 3; r++) avg[r] /= nrm;
  double dot = avg[0] * normal[0] + avg[1] * normal[1] + avg[2] * normal[2];
  if (dot < 0)
    for (int r = 0; r < 3; r++) normal[r] *= -1.0;
  binormal[0] = curv[1] * normal[2] - curv[2] * normal[1];
  binormal[1] = curv[2] * normal[0] - curv[0] * normal[2];
  binormal[2] = curv[0] * normal[1] - curv[1] * normal[0];
}

// ---------------------------------------------------------------------------
// FingerHand — finger_hand.cpp:6-184 — on a neighbourhood already in the hand
// frame.  Points are given as the *cropped* list of PointList::cropByHandHeight
// (point_list.cpp:44-55), which (Q1) holds the k in-height points followed by
// N-k copies of column 0; we keep the ghost as one entry with a multiplicity.
// ---------------------------------------------------------------------------
struct FramePts {
  std::vector<double> x, y, z, nx, ny, nz;
  std::vector<int> mult;  // multiplicity (1, or N-k for the ghost)
  size_t size() const { return x.size(); }
  void clear() {
    x.clear(); y.clear(); z.clear(); nx.clear(); ny.clear(); nz.clear(); mult.clear();
  }
};

struct FingerHand {
  int n;  // placements
  double fw, depth;
  std::vector<double> spacing;  // 2n
  std::vector<char> fingers, hand;
  double top, bottom, center, left, right, surface;

  FingerHand(double finger_width, double od, double hand_depth, int num) : n(num), fw(finger_width), depth(hand_depth) {
    // finger_hand.cpp:13-18; LinSpaced(n, lo, hi) = lo + i*(hi-lo)/(n-1), last = hi
    std::vector<double> half(n);
    double lo = 0.0, hi = od - finger_width;
    double step = (n > 1) ? (hi - lo) / (double)(n - 1) : 0.0;
    for (int i = 0; i < n; i++) half[i] = (i == n - 1) ? hi : lo + (double)i * step;
    spacing.resize(2 * n);
    for (int i = 0; i < n; i++) {
      spacing[i] = (half[i] - od) + finger_width;
      spacing[n + i] = half[i];
    }
    fingers.assign(2 * n, 0);
    hand.assign(n, 0);
    top = bottom = center = left = right = surface = 0.0;
  }

  bool gapFree(const FramePts &p, const std::vector<int> &cropped, int idx) const {
    // finger_hand.cpp:173-184
    for (int i : cropped) {
      double y = p.y[i];
      if (y > spacing[idx] && y < spacing[idx] + fw) return false;
    }
    return true;
  }

  // finger_hand.cpp:26-73
  void evaluateFingers(const FramePts &p, double bite, int idx = -1) {
    top = bite;
    bottom = bite - depth;
    center = 0.0;
    std::fill(fingers.begin(), fingers.end(), 0);
    std::vector<int> cropped;
    for (size_t i = 0; i < p.size(); i++) {
      if (p.x[i] < bite) {
        if (p.x[i] < bottom) return;
        cropped.push_back((int)i);
      }
    }
    if (cropped.empty()) return;
    if (idx == -1) {
      for (int i = 0; i < 2 * n; i++)
        if (gapFree(p, cropped, i)) fingers[i] = 1;
    } else {
      if (gapFree(p, cropped, idx)) fingers[idx] = 1;
      if (gapFree(p, cropped, n + idx)) fingers[n + idx] = 1;
    }
  }
  void evaluateHand() {  // finger_hand.cpp:75-81
    for (int i = 0; i < n; i++) hand[i] = fingers[i] && fingers[n + i];
  }
  void evaluateHand(int idx) {  // finger_hand.cpp:83-87
    std::fill(hand.begin(), hand.end(), 0);
    hand[idx] = fingers[idx] && fingers[n + idx];
  }
  bool any() const {
    for (char h : hand)
      if (h) return true;
    return false;
  }
  int chooseMiddleHand() const {  // finger_hand.cpp:89-105
    std::vector<int> idx;
    for (int i = 0; i < n; i++)
      if (hand[i]) idx.push_back(i);
    if (idx.empty()) return -1;
    return idx[(int)std::ceil(idx.size() / 2.0) - 1];
  }
  int deepenHand(const FramePts &p, double min_depth, double max_depth) {  // finger_hand.cpp:107-139
    int mid = chooseMiddleHand();
    int opp = n + mid;
    const double STEP = 0.005;
    FingerHand nh = *this;
    FingerHand last = nh;
    for (double d = min_depth + STEP; d <= max_depth; d += STEP) {
      nh.evaluateFingers(p, d, mid);
      if (!nh.fingers[mid] || !nh.fingers[opp]) break;
      hand[mid] = 1;
      last = nh;
    }
    *this = last;
    std::fill(hand.begin(), hand.end(), 0);
    hand[mid] = 1;
    return mid;
  }
  // finger_hand.cpp:141-171; returns indices into p
  std::vector<int> closingRegion(const FramePts &p, int idx) {
    if (idx == -1)
      for (int i = 0; i < n; i++)
        if (hand[i]) {
          idx = i;
          break;
        }
    left = spacing[idx] + fw;
    right = spacing[n + idx];
    center = 0.5 * (left + right);
    surface = DBL_MAX;
    for (size_t i = 0; i < p.size(); i++) surface = std::min(surface, p.y[i]);
    std::vector<int> out;
    for (size_t i = 0; i < p.size(); i++)
      if (p.x[i] > bottom && p.x[i] < top && p.y[i] > left && p.y[i] < right) out.push_back((int)i);
    return out;
  }
};

// Antipodal::evaluateGrasp(point_list, 0.003, lateral=1, forward=0, vertical=2)
// — antipodal.cpp:10-96.  0 none, 1 half, 2 full.  Ghost multiplicities count.
static int antipodalLabel(const FramePts &p, const std::vector<int> &idx, double friction_coeff, int min_viable) {
  const double extremal = 0.003;
  double cosf = std::cos(friction_coeff * M_PI / 180.0);
  double mn = DBL_MAX, mx = -DBL_MAX;
  for (int i : idx) {
    mn = std::min(mn, p.y[i]);
    mx = std::max(mx, p.y[i]);
  }
  double min_x = mn + extremal, max_x = mx - extremal;
  std::vector<int> L, R;
  for (int i : idx) {
    // l = (0,-1,0), r = (0,1,0): l^T n = 0*nx + (-1)*ny + 0*nz
    double ln = 0.0 * p.nx[i] + -1.0 * p.ny[i] + 0.0 * p.nz[i];
    double rn = 0.0 * p.nx[i] + 1.0 * p.ny[i] + 0.0 * p.nz[i];
    bool lc = ln > cosf, rc = rn > cosf;
    bool le = p.y[i] < min_x, re = p.y[i] > max_x;
    if (lc && le) L.push_back(i);
    if (rc && re) R.push_back(i);
  }
  int result = 0;
  if (!L.empty() || !R.empty()) result = 1;
  if (!L.empty() && !R.empty()) {
    auto mm = [&](const std::vector<int> &v, const std::vector<double> &a, double &lo, double &hi) {
      lo = DBL_MAX;
      hi = -DBL_MAX;
      for (int i : v) {
        lo = std::min(lo, a[i]);
        hi = std::max(hi, a[i]);
      }
    };
    double lxlo, lxhi, rxlo, rxhi, lzlo, lzhi, rzlo, rzhi;
    mm(L, p.x, lxlo, lxhi);
    mm(R, p.x, rxlo, rxhi);
    mm(L, p.z, lzlo, lzhi);
    mm(R, p.z, rzlo, rzhi);
    double top_y = std::min(lxhi, rxhi), bot_y = std::max(lxlo, rxlo);
    double top_z = std::min(lzhi, rzhi), bot_z = std::max(lzlo, rzlo);
    long nl = 0, nr = 0;
    for (int i : L)
      if (p.x[i] >= bot_y && p.x[i] <= top_y && p.z[i] >= bot_z && p.z[i] <= top_z) nl += p.mult[i];
    for (int i : R)
      if (p.x[i] >= bot_y && p.x[i] <= top_y && p.z[i] >= bot_z && p.z[i] <= top_z) nr += p.mult[i];
    if (nl >= min_viable && nr >= min_viable) result = 2;
  }
  return result;
}

// orientation angles — hand_search.cpp:151-155: LinSpaced(n+1, -pi/2, pi/2).head(n)
static std::vector<double> orientationAngles(int n) {
  std::vector<double> a(n);
  double lo = -1.0 * M_PI / 2.0, hi = M_PI / 2.0;
  double step = (hi - lo) / (double)n;  // (n+1)-1 intervals
  for (int i = 0; i < n; i++) a[i] = lo + (double)i * step;
  return a;
}

// transformToHandFrame (point_list.cpp:22-33) with rotation = FR^T, then cropByHandHeight
// (point_list.cpp:44-55, quirk Q1: the N-k rejected columns come back as copies of column 0).
static void handFramePoints(const gpd_params &P, const float *xyz, const float *normals, const std::vector<Neighbour> &nbr,
                            const double *sample, const double *FR, FramePts &pts) {
  const int N = (int)nbr.size();
  pts.clear();
  double g[6] = {0, 0, 0, 0, 0, 0};
  int k = 0;
  for (int i = 0; i < N; i++) {
    const float *pp = xyz + 3 * nbr[i].idx;
    const float *nn = normals + 3 * nbr[i].idx;
    double c[3] = {(double)pp[0] - sample[0], (double)pp[1] - sample[1], (double)pp[2] - sample[2]};
    double nd[3] = {(double)nn[0], (double)nn[1], (double)nn[2]};
    double t[3], tn[3];
    for (int r = 0; r < 3; r++) {
      t[r] = FR[0 + r] * c[0] + FR[3 + r] * c[1] + FR[6 + r] * c[2];
      tn[r] = FR[0 + r] * nd[0] + FR[3 + r] * nd[1] + FR[6 + r] * nd[2];
    }
    if (i == 0) {
      g[0] = t[0]; g[1] = t[1]; g[2] = t[2]; g[3] = tn[0]; g[4] = tn[1]; g[5] = tn[2];
    }
    if (t[2] > -1.0 * P.hand_height && t[2] < P.hand_height) {
      pts.x.push_back(t[0]); pts.y.push_back(t[1]); pts.z.push_back(t[2]);
      pts.nx.push_back(tn[0]); pts.ny.push_back(tn[1]); pts.nz.push_back(tn[2]);
      pts.mult.push_back(1);
      k++;
    }
  }
  if (N - k > 0) {  // ghosts: N-k copies of column 0
    pts.x.push_back(g[0]); pts.y.push_back(g[1]); pts.z.push_back(g[2]);
    pts.nx.push_back(g[3]); pts.ny.push_back(g[4]); pts.nz.push_back(g[5]);
    pts.mult.push_back(N - k);
  }
}

// HandSet::evalHandSet / evalHands — hand_set.cpp:31-116, 235-261.
// nbr: the nn_radius (0.11) neighbourhood of the sample in (d2,idx) order.
static void evalHandSet(const gpd_params &P, const float *xyz, const float *normals, const std::vector<Neighbour> &nbr,
                        const double *frame12, int set_index, gpd_hand *out) {
  const double *sample = frame12;
  // frame_ << normal, binormal, curvature (columns) — hand_set.cpp:39-40
  double F[9];
  for (int r = 0; r < 3; r++) {
    F[3 * r + 0] = frame12[3 + r];
    F[3 * r + 1] = frame12[6 + r];
    F[3 * r + 2] = frame12[9 + r];
  }
  const double UY[3] = {0, 1, 0};
  double RB[9];
  angleAxis(M_PI, UY, RB);
  double FRB[9];
  mat3mul(F, RB, FRB);
  std::vector<double> angles = orientationAngles(P.num_orientations);
  const double AX[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  FramePts pts;
  for (int ai = 0; ai < P.num_hand_axes; ai++) {
    FingerHand fh(P.finger_width, P.hand_outer_diameter, P.hand_depth, P.num_finger_placements);
    for (int oi = 0; oi < P.num_orientations; oi++) {
      const int slot = ai * P.num_orientations + oi;
      gpd_hand &H = out[slot];
      std::memset(&H, 0, sizeof(H));
      double R[9], FR[9];
      angleAxis(angles[oi], AX[P.hand_axes[ai]], R);
      mat3mul(FRB, R, FR);
      handFramePoints(P, xyz, normals, nbr, sample, FR, pts);
      fh.evaluateFingers(pts, P.init_bite);
      fh.evaluateHand();
      // Hand(sample, frame_rot, finger_hand, 0.0) — hand_set.cpp:89-90, hand.cpp:24-45
      auto construct = [&](gpd_hand &h) {
        h.top = fh.top;
        h.bottom = fh.bottom;
        h.center = fh.center;
        for (int r = 0; r < 3; r++)
          h.position[r] = (FR[3 * r + 0] * h.bottom + FR[3 * r + 1] * fh.center + FR[3 * r + 2] * 0.0) + sample[r];
        h.finger_placement_index = -1;
        for (int i = 0; i < fh.n; i++)
          if (fh.hand[i]) {
            h.finger_placement_index = i;
            break;
          }
      };
      for (int r = 0; r < 3; r++) H.sample[r] = sample[r];
      std::memcpy(H.frame, FR, sizeof(FR));
      H.set_index = set_index;
      H.slot = slot;
      construct(H);
      if (fh.any()) {
        int fidx = P.deepen_hand ? fh.deepenHand(pts, P.init_bite, P.hand_depth) : fh.chooseMiddleHand();
        std::vector<int> closing = fh.closingRegion(pts, fidx);
        if (closing.empty()) continue;
        H.valid = 1;
        construct(H);  // modifyCandidate: hand.construct(finger_hand)
        double mn = DBL_MAX, mx = -DBL_MAX;
        for (int i : closing) {
          mn = std::min(mn, pts.y[i]);
          mx = std::max(mx, pts.y[i]);
        }
        H.grasp_width = mx - mn;
        int label = antipodalLabel(pts, closing, P.friction_coeff, P.min_viable);
        H.half_antipodal = label >= 1;
        H.full_antipodal = label == 2;
      }
    }
  }
}

static double nnRadiusHands(const gpd_params &P) {  // hand_search.cpp:10-17
  return std::max({P.hand_outer_diameter - P.finger_width, P.hand_depth, P.hand_height / 2.0});
}
static double nnRadiusImages(const gpd_params &P) {  // image_generator.cpp:43-46
  return std::max({P.volume_depth, P.volume_height / 2.0, P.volume_width});
}

// ---------------------------------------------------------------------------
// Shadow — hand_set.cpp:118-283 (15 channels only).
// ---------------------------------------------------------------------------
static uint64_t g_lcg_base = 0, g_lcg_draws = 0;  // gpd_oracle_set_lcg_base / gpd_oracle_last_lcg_draws (below)
struct Lcg {  // hand_set.cpp:263-266 ("fastrand"); jump() = n steps at once
  uint32_t s;
  int next() {
    s = 214013u * s + 2531011u;
    return (int)(((int32_t)s >> 16) & 0x7FFF);
  }
  void jump(uint64_t n) {
    uint32_t a = 214013u, c = 2531011u, A = 1u, C = 0u;
    while (n) {
      if (n & 1) {
        A = a * A;
        C = a * C + c;
      }
      c = (a + 1u) * c;
      a = a * a;
      n >>= 1;
    }
    s = A * s + C;
  }
};

struct Voxel {
  int v[3];
  bool operator<(const Voxel &o) const {
    if (v[0] != o.v[0]) return v[0] < o.v[0];
    if (v[1] != o.v[1]) return v[1] < o.v[1];
    return v[2] < o.v[2];
  }
  bool operator==(const Voxel &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};

// calculateShadow: sorted, de-duplicated voxel list (lexicographic order).
static void calculateShadow(const float *xyz, int P, const int32_t *cam_source, int n_cams, const double *view_points,
                            const std::vector<Neighbour> &nbr, double shadow_length, Lcg &rng, std::vector<Voxel> &out) {
  const double voxel = 0.003;
  const int num_shadow = (int)std::floor(shadow_length / voxel);
  const double mult = 1.0 / voxel;
  const double maxinv = 1.0 / 32767.0;
  const int N = (int)nbr.size();
  double center[3] = {0, 0, 0};
  for (int i = 0; i < N; i++)
    for (int r = 0; r < 3; r++) center[r] += (double)xyz[3 * nbr[i].idx + r];
  for (int r = 0; r < 3; r++) center[r] /= (double)N;
  std::vector<std::vector<Voxel>> sets(n_cams);
  std::vector<char> seen(n_cams, 0);
  for (int c = 0; c < n_cams; c++) {
    long s = 0;
    for (int i = 0; i < N; i++) s += cam_source[(size_t)c * P + nbr[i].idx];
    seen[c] = s >= 1;
    if (!seen[c]) continue;
    double vec[3];
    for (int r = 0; r < 3; r++) vec[r] = center[r] - view_points[3 * c + r];
    double nrm = std::sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
    for (int r = 0; r < 3; r++) vec[r] = shadow_length * vec[r] / nrm;
    std::vector<Voxel> &S = sets[c];
    S.reserve((size_t)N * num_shadow);
    const int n = N * num_shadow;
    for (int i = 0; i < n; i++) {
      const float *pp = xyz + 3 * nbr[i / num_shadow].idx;
      double t = (double)rng.next() * maxinv;
      Voxel v;
      for (int r = 0; r < 3; r++) v.v[r] = (int)(((double)pp[r] + t * vec[r]) * mult);
      S.push_back(v);
    }
    std::sort(S.begin(), S.end());
    S.erase(std::unique(S.begin(), S.end()), S.end());
  }
  if (n_cams == 1) {
    out = sets[0];
    return;
  }
  out = sets[0];
  for (int c = 1; c < n_cams; c++) {
    if (!seen[c]) continue;
    std::vector<Voxel> tmp;
    std::set_intersection(out.begin(), out.end(), sets[c].begin(), sets[c].end(), std::back_inserter(tmp));
    out.swap(tmp);
  }
}

// ---------------------------------------------------------------------------
// Rasterisers — image_strategy.cpp:92-243.
// ---------------------------------------------------------------------------
static const int IS = 60;
static const int NPIX = IS * IS;

static inline uint8_t toU8(float v) {  // convertTo(CV_8U, 255.0): round-half-even, saturate
  float u = v * 255.0f + 0.0f;
  long r = std::lrintf(u);
  return (uint8_t)std::min(255L, std::max(0L, r));
}

// 3x3 rect max dilate (border ignored), NORM_MINMAX to [0,1], u8; nch interleaved.
static void dilateNormalizeU8(const float *img, int nch, uint8_t *out, int out_stride, int out_off) {
  std::vector<float> d((size_t)NPIX * nch);
  for (int r = 0; r < IS; r++)
    for (int c = 0; c < IS; c++)
      for (int ch = 0; ch < nch; ch++) {
        float m = -FLT_MAX;
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= IS || cc < 0 || cc >= IS) continue;
            m = std::max(m, img[(rr * IS + cc) * nch + ch]);
          }
        d[(r * IS + c) * nch + ch] = m;
      }
  double smin = DBL_MAX, smax = -DBL_MAX;
  for (float v : d) {
    smin = std::min(smin, (double)v);
    smax = std::max(smax, (double)v);
  }
  double scale = 1.0 * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
  double shift = 0.0 - smin * scale;
  const float fs = (float)scale, fb = (float)shift;
  for (int p = 0; p < NPIX; p++)
    for (int ch = 0; ch < nch; ch++) {
      float v = d[p * nch + ch] * fs + fb;
      out[(size_t)p * out_stride + out_off + ch] = toU8(v);
    }
}

// findCellIndices — image_strategy.cpp:92-102
static inline int cellIndex(double u0, double u1) {
  const double cellsize = 1.0 / (double)IS;
  int v = std::min((int)std::floor(u0 / cellsize), IS - 1);
  int h = std::min((int)std::floor(u1 / cellsize), IS - 1);
  return h + v * IS;
}

// createNormalsImage — image_strategy.cpp:124-156
static void normalsImage(const std::vector<double> &nrm, const std::vector<int> &cells, uint8_t *out, int stride, int off) {
  std::vector<float> img((size_t)NPIX * 3, 0.f);
  for (size_t i = 0; i < cells.size(); i++) {
    int idx = cells[i];
    int row = IS - 1 - idx / IS, col = idx % IS;
    float *v = &img[(row * IS + col) * 3];
    float a[3] = {(float)std::fabs(nrm[3 * i]), (float)std::fabs(nrm[3 * i + 1]), (float)std::fabs(nrm[3 * i + 2])};
    if (v[0] == 0 && v[1] == 0 && v[2] == 0) {
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2];
    } else {
      float s = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      double inv = 1.0 / (double)s;
      for (int c = 0; c < 3; c++) {
        float d = a[c] - v[c];
        float t = (float)((double)d * inv);
        v[c] = v[c] + t;
      }
    }
  }
  dilateNormalizeU8(img.data(), 3, out, stride, off);
}

// createDepthImage — image_strategy.cpp:158-190.  depth = row 2 of the projected points.
static void depthImage(const std::vector<double> &depth, const std::vector<int> &cells, uint8_t *out, int stride, int off) {
  std::vector<float> img(NPIX, 0.f), avgs(NPIX, 0.f), counts(NPIX, 0.f);
  for (size_t i = 0; i < cells.size(); i++) {
    int idx = cells[i];
    int row = IS - 1 - idx / IS, col = idx % IS;
    counts[idx] = (float)((double)counts[idx] + 1.0);
    avgs[idx] = (float)((double)avgs[idx] + (depth[i] - (double)avgs[idx]) * (1.0 / (double)counts[idx]));
    img[row * IS + col] = (float)(1.0 - (double)avgs[idx]);
  }
  dilateNormalizeU8(img.data(), 1, out, stride, off);
}

// createShadowImage — image_strategy.cpp:192-233
static void shadowImage(const std::vector<double> &depth, const std::vector<int> &cells, uint8_t *out, int stride, int off) {
  std::vector<float> img(NPIX, 0.f), counts(NPIX, 0.f);
  std::vector<uint8_t> nz(NPIX, 0);
  for (size_t i = 0; i < cells.size(); i++) {
    int idx = cells[i];
    int row = IS - 1 - idx / IS, col = idx % IS;
    counts[idx] = (float)((double)counts[idx] + 1.0);
    float &v = img[row * IS + col];
    v = (float)((double)v + (depth[i] - (double)v) * (1.0 / (double)counts[idx]));
    nz[row * IS + col] = 1;
  }
  // minMaxLoc with mask: max over masked pixels, 0 when the mask is empty
  bool any = false;
  float mxf = -FLT_MAX;
  for (int p = 0; p < NPIX; p++)
    if (nz[p]) {
      any = true;
      mxf = std::max(mxf, img[p]);
    }
  double mx = any ? (double)mxf : 0.0;
  for (int p = 0; p < NPIX; p++) {
    float m = nz[p] ? (float)mx : 0.0f;
    img[p] = m - img[p];
  }
  dilateNormalizeU8(img.data(), 1, out, stride, off);
}

struct UnitPts {
  std::vector<double> u;  // 3*n unit-cube coords
  std::vector<double> n;  // 3*n normals in hand frame (unpermuted)
};

// transformToUnitImage / findPointsInUnitImage / transformPointsToUnitImage —
// image_strategy.cpp:32-90.  src(i, out[3]) yields the i-th world point.
template <class Src>
static void toUnitImage(const gpd_params &P, const gpd_hand &H, int N, Src src, bool with_normals, UnitPts &out) {
  out.u.clear();
  out.n.clear();
  const double half_od = P.volume_width / 2.0;
  const double dbl_h = 2.0 * P.volume_height;
  const double *F = H.frame;
  for (int i = 0; i < N; i++) {
    double w[3], nn[3];
    src(i, w, nn);
    double c[3] = {w[0] - H.sample[0], w[1] - H.sample[1], w[2] - H.sample[2]};
    double t[3];
    for (int r = 0; r < 3; r++) t[r] = F[0 + r] * c[0] + F[3 + r] * c[1] + F[6 + r] * c[2];
    if ((t[0] > H.bottom) && (t[0] < H.bottom + P.volume_depth) && (t[1] > H.center - half_od) && (t[1] < H.center + half_od) &&
        (t[2] > -1.0 * P.volume_height) && (t[2] < P.volume_height)) {
      out.u.push_back((t[0] - H.bottom) / P.volume_depth);
      out.u.push_back((t[1] - (H.center - half_od)) / P.volume_width);
      out.u.push_back((t[2] + P.volume_height) / dbl_h);
      if (with_normals)
        for (int r = 0; r < 3; r++) out.n.push_back(F[0 + r] * nn[0] + F[3 + r] * nn[1] + F[6 + r] * nn[2]);
    }
  }
}

// Image{15,12,3,1}ChannelsStrategy::createImage/calculateImage/calculateChannels —
// image_15_channels_strategy.cpp:27-105, image_12…:27-86, image_3…:27-42,
// image_1_channels_strategy.cpp:25-49 (one channel = the depth image of projection 0).
static void createImage(const gpd_params &P, const gpd_hand &H, const float *xyz, const float *normals, const std::vector<Neighbour> &nbr,
                        const std::vector<Voxel> &shadow, uint8_t *img) {
  const int C = P.image_num_channels;
  UnitPts pts, sh;
  toUnitImage(P, H, (int)nbr.size(),
              [&](int i, double *w, double *nn) {
                const float *pp = xyz + 3 * nbr[i].idx;
                const float *q = normals + 3 * nbr[i].idx;
                for (int r = 0; r < 3; r++) {
                  w[r] = (double)pp[r];
                  nn[r] = (double)q[r];
                }
              },
              true, pts);
  if (C == 15)
    toUnitImage(P, H, (int)shadow.size(),
                [&](int i, double *w, double *) {
                  for (int r = 0; r < 3; r++) w[r] = (double)shadow[i].v[r] * 0.003;
                },
                false, sh);
  const int np = (int)pts.u.size() / 3, ns = (int)sh.u.size() / 3;
  // projections by cumulative row swaps (0<->2 then 1<->2): (x,y,z),(z,y,x),(z,x,y)
  static const int perm[3][3] = {{0, 1, 2}, {2, 1, 0}, {2, 0, 1}};
  const int nproj = (C <= 3) ? 1 : 3;
  const int per = (C == 15) ? 5 : (C == 12 ? 4 : C);
  std::vector<int> cells(np), scells(ns);
  std::vector<double> depth(np), sdepth(ns);
  for (int pr = 0; pr < nproj; pr++) {
    for (int i = 0; i < np; i++) {
      cells[i] = cellIndex(pts.u[3 * i + perm[pr][0]], pts.u[3 * i + perm[pr][1]]);
      depth[i] = pts.u[3 * i + perm[pr][2]];
    }
    if (C == 1) {
      depthImage(depth, cells, img, C, 0);
      break;
    }
    normalsImage(pts.n, cells, img, C, pr * per);
    if (C >= 12) depthImage(depth, cells, img, C, pr * per + 3);
    if (C == 15) {
      for (int i = 0; i < ns; i++) {
        scells[i] = cellIndex(sh.u[3 * i + perm[pr][0]], sh.u[3 * i + perm[pr][1]]);
        sdepth[i] = sh.u[3 * i + perm[pr][2]];
      }
      shadowImage(sdepth, scells, img, C, pr * per + 4);
    }
  }
}

// ---------------------------------------------------------------------------
// LeNet — eigen_classifier.cpp:81-183, conv_layer.cpp:26-98, dense_layer.cpp:6-15.
// ---------------------------------------------------------------------------
static void convForward(const float *x, int C, int H, int W, const float *w, const float *b, int F, float *out) {
  const int OH = H - 4, OW = W - 4;
  const int K = C * 25;
  std::vector<float> acc(OW);
  for (int f = 0; f < F; f++)
    for (int oy = 0; oy < OH; oy++) {
      std::fill(acc.begin(), acc.end(), 0.f);
      for (int k = 0; k < K; k++) {
        int c = k / 25, kh = (k % 25) / 5, kw = k % 5;
        const float wk = w[(size_t)f * K + k];
        const float *row = x + ((size_t)c * H + oy + kh) * W + kw;
        for (int ox = 0; ox < OW; ox++) acc[ox] = fmaf(wk, row[ox], acc[ox]);
      }
      for (int ox = 0; ox < OW; ox++) out[((size_t)f * OH + oy) * OW + ox] = acc[ox] + b[f];
    }
}
static void pool2(const float *x, int C, int H, int W, float *out) {  // eigen_classifier.cpp:151-183
  const int OH = H / 2, OW = W / 2;
  for (int c = 0; c < C; c++)
    for (int y = 0; y < OH; y++)
      for (int xx = 0; xx < OW; xx++) {
        const float *p = x + ((size_t)c * H + 2 * y) * W + 2 * xx;
        out[((size_t)c * OH + y) * OW + xx] = std::max(std::max(p[0], p[1]), std::max(p[W], p[W + 1]));
      }
}

struct LeNetW {
  int C;
  const float *c1w, *c1b, *c2w, *c2b, *f1w, *f1b, *f2w, *f2b;
};

static float lenetForward(const LeNetW &w, const uint8_t *img_hwc) {
  const int C = w.C;
  std::vector<float> x((size_t)C * NPIX);
  for (int c = 0; c < C; c++)  // imageToArray: HWC u8 -> CHW float
    for (int p = 0; p < NPIX; p++) x[(size_t)c * NPIX + p] = (float)img_hwc[(size_t)p * C + c];
  std::vector<float> h1(20 * 56 * 56), p1(20 * 28 * 28), h2(50 * 24 * 24), p2(50 * 12 * 12);
  convForward(x.data(), C, 60, 60, w.c1w, w.c1b, 20, h1.data());
  pool2(h1.data(), 20, 56, 56, p1.data());
  convForward(p1.data(), 20, 28, 28, w.c2w, w.c2b, 50, h2.data());
  pool2(h2.data(), 50, 24, 24, p2.data());
  // flatten column-major of 50 x 144: j = pixel*50 + channel (eigen_classifier.cpp:103-107)
  std::vector<float> f(7200), a1(500, 0.f);
  for (int pix = 0; pix < 144; pix++)
    for (int ch = 0; ch < 50; ch++) f[pix * 50 + ch] = p2[ch * 144 + pix];
  for (int j = 0; j < 7200; j++) {
    const float xj = f[j];
    const float *col = w.f1w + (size_t)j * 500;
    for (int u = 0; u < 500; u++) a1[u] = fmaf(col[u], xj, a1[u]);
  }
  for (int u = 0; u < 500; u++) a1[u] = std::max(a1[u] + w.f1b[u], 0.f);
  float y[2] = {0.f, 0.f};
  for (int j = 0; j < 500; j++) {
    y[0] = fmaf(w.f2w[j * 2 + 0], a1[j], y[0]);
    y[1] = fmaf(w.f2w[j * 2 + 1], a1[j], y[1]);
  }
  y[0] += w.f2b[0];
  y[1] += w.f2b[1];
  return y[1] - y[0];
}

// GraspDetector::filterGraspsWorkspace — grasp_detector.cpp:334-398 (Q6 typo kept).
static void filterWorkspace(const gpd_params &P, gpd_hand *hands, int n_sets, int n_slots) {
  for (int s = 0; s < n_sets; s++)
    for (int j = 0; j < n_slots; j++) {
      gpd_hand &h = hands[(size_t)s * n_slots + j];
      if (!h.valid) continue;
      double half_width = 0.5 * P.hand_outer_diameter;
      double lb[3], rb[3], lt[3], rt[3], ap[3];
      for (int r = 0; r < 3; r++) {
        double bin = h.frame[3 * r + 1], app = h.frame[3 * r + 0];
        lb[r] = h.position[r] + half_width * bin;
        rb[r] = h.position[r] - half_width * bin;
        lt[r] = lb[r] + P.hand_depth * app;
        rt[r] = lb[r] + P.hand_depth * app;
        ap[r] = h.position[r] - 0.05 * app;
      }
      bool ok = h.grasp_width >= P.min_aperture && h.grasp_width <= P.max_aperture;
      for (int r = 0; r < 3 && ok; r++) {
        double mn = std::min({lb[r], rb[r], lt[r], rt[r], ap[r]});
        double mx = std::max({lb[r], rb[r], lt[r], rt[r], ap[r]});
        ok = mn >= P.workspace_grasps[2 * r] && mx <= P.workspace_grasps[2 * r + 1];
      }
      h.valid = ok ? 1 : 0;
    }
}

// GraspDetector::filterGraspsDirection — grasp_detector.cpp:423-456: angle = acos(direction^T * approach), dropped when
// angle > thresh_rad.  acos is not clamped: a dot product a hair outside [-1, 1] gives NaN, and NaN > thresh is false.
static void filterDirection(const gpd_params &P, gpd_hand *hands, int n_sets, int n_slots) {
  for (size_t i = 0; i < (size_t)n_sets * n_slots; i++) {
    gpd_hand &h = hands[i];
    if (!h.valid) continue;
    const double angle = std::acos(P.direction[0] * h.frame[0] + P.direction[1] * h.frame[3] + P.direction[2] * h.frame[6]);
    if (angle > P.thresh_rad) h.valid = 0;
  }
}

}  // namespace

// ===========================================================================
// C entry points (ctypes)
// ===========================================================================
extern "C" {

void gpd_oracle_default_params(gpd_params *p) {
  std::memset(p, 0, sizeof(*p));
  p->finger_width = 0.01;
  p->hand_outer_diameter = 0.12;
  p->hand_depth = 0.06;
  p->hand_height = 0.02;
  p->init_bite = 0.01;
  p->volume_width = 0.10;
  p->volume_depth = 0.06;
  p->volume_height = 0.02;
  p->nn_radius_frames = 0.01;
  p->friction_coeff = 20.0;
  p->min_aperture = 0.0;
  p->max_aperture = 0.085;
  const double ws[6] = {-1, 1, -1, 1, -1, 1};
  std::memcpy(p->workspace_grasps, ws, sizeof(ws));
  p->image_size = 60;
  p->image_num_channels = 15;
  p->num_orientations = 8;
  p->num_finger_placements = 10;
  p->num_hand_axes = 1;
  p->hand_axes[0] = 2;
  p->deepen_hand = 1;
  p->min_viable = 6;
  p->filter_approach_direction = 0;  // cfg/eigen_params.cfg:60-62
  p->direction[0] = 1.0;
  p->thresh_rad = 2.0;
}

int gpd_oracle_sizeof_hand() { return (int)sizeof(gpd_hand); }
int gpd_oracle_sizeof_params() { return (int)sizeof(gpd_params); }

// sorted radius search of one query; returns count (may exceed cap; only cap written)
int gpd_oracle_radius_search(const float *xyz, int P, const float *query, double radius, int32_t *idx, float *d2, int cap) {
  Grid g;
  g.build(xyz, P, 0.02f);
  std::vector<Neighbour> nb;
  radiusSearch(g, query, radius, nb);
  for (int i = 0; i < (int)nb.size() && i < cap; i++) {
    idx[i] = nb[i].idx;
    d2[i] = nb[i].d2;
  }
  return (int)nb.size();
}

void gpd_oracle_eigen3(const double *M, double *eval, double *evec) { selfAdjointEigen3(M, eval, evec); }

void gpd_oracle_finger_spacing(double fw, double od, double depth, int n, double *out) {
  FingerHand fh(fw, od, depth, n);
  for (int i = 0; i < 2 * n; i++) out[i] = fh.spacing[i];
}
void gpd_oracle_angles(int n, double *out) {
  std::vector<double> a = orientationAngles(n);
  for (int i = 0; i < n; i++) out[i] = a[i];
}
void gpd_oracle_fastrand(int n, int32_t *out) {
  Lcg r{0};
  for (int i = 0; i < n; i++) out[i] = r.next();
}
int gpd_oracle_fastrand_at(uint64_t offset) {  // draw number `offset` (0-based) via jump-ahead
  Lcg r{0};
  r.jump(offset);
  return r.next();
}
void gpd_oracle_angle_axis(double angle, const double *axis, double *R) { angleAxis(angle, axis, R); }

// frames: out 12 doubles per sample (sample, normal, binormal, curvature), has[i]=0 if no neighbour
// FrameEstimator::calculateLocalFrames for samples by index (frame_estimator.cpp:6-36) or by
// coordinates (:38-65; sample_xyz != NULL): the kd-tree query is the float cast of the sample
// (eigenVectorToPcl, frame_estimator.cpp:88-95), the frame keeps the double.
static void localFrames(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *sample_idx,
                        const double *sample_xyz, int S, double *frames, uint8_t *has) {
  Grid g;
  g.build(xyz, np, 0.02f);
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < S; i++) {
    std::vector<Neighbour> nb;
    float qf[3];
    double qd[3];
    for (int r = 0; r < 3; r++) {
      qd[r] = sample_xyz ? sample_xyz[3 * i + r] : (double)xyz[3 * sample_idx[i] + r];
      qf[r] = (float)qd[r];
    }
    const float *q = qf;
    radiusSearch(g, q, P->nn_radius_frames, nb);
    has[i] = !nb.empty();
    if (nb.empty()) continue;
    std::vector<double> nn(3 * nb.size());
    for (size_t k = 0; k < nb.size(); k++)
      for (int r = 0; r < 3; r++) nn[3 * k + r] = (double)normals[3 * nb[k].idx + r];
    double *f = frames + 12 * i;
    for (int r = 0; r < 3; r++) f[r] = qd[r];
    averageNormalAxis(nn, f + 3, f + 6, f + 9);
  }
}
void gpd_oracle_frames(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *sample_idx, int S,
                       double *frames, uint8_t *has) {
  localFrames(P, xyz, normals, np, sample_idx, nullptr, S, frames, has);
}

// HandSearch::searchHands (hand_search.cpp:24-64, 144-188), samples by index.
// hands: S*n_slots records; returns n_sets via pointer.
static int searchHands(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *sample_idx,
                       const double *sample_xyz, int S, gpd_hand *hands, int *num_sets) {
  const int n_slots = P->num_hand_axes * P->num_orientations;
  std::vector<double> frames((size_t)12 * S);
  std::vector<uint8_t> has(S);
  localFrames(P, xyz, normals, np, sample_idx, sample_xyz, S, frames.data(), has.data());
  std::vector<int> kept;
  for (int i = 0; i < S; i++)
    if (has[i]) kept.push_back(i);
  Grid g;
  g.build(xyz, np, 0.02f);
  const double radius = nnRadiusHands(*P);
#pragma omp parallel for schedule(dynamic, 4)
  for (int s = 0; s < (int)kept.size(); s++) {
    const double *f = &frames[(size_t)12 * kept[s]];
    float q[3] = {(float)f[0], (float)f[1], (float)f[2]};
    std::vector<Neighbour> nb;
    radiusSearch(g, q, radius, nb);
    gpd_hand *out = hands + (size_t)s * n_slots;
    if (nb.empty()) {  // cannot happen for samples by index; keep records defined
      std::memset(out, 0, sizeof(gpd_hand) * n_slots);
      continue;
    }
    evalHandSet(*P, xyz, normals, nb, f, s, out);
  }
  *num_sets = (int)kept.size();
  return 0;
}
int gpd_oracle_search(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *sample_idx, int S, gpd_hand *hands,
                      int *num_sets) {
  return searchHands(P, xyz, normals, np, sample_idx, nullptr, S, hands, num_sets);
}
// the same for samples given by coordinates (Cloud::getSamples, hand_search.cpp:37-39)
int gpd_oracle_search_xyz(const gpd_params *P, const float *xyz, const float *normals, int np, const double *sample_xyz, int S,
                          gpd_hand *hands, int *num_sets) {
  return searchHands(P, xyz, normals, np, nullptr, sample_xyz, S, hands, num_sets);
}

// HandSearch::reevaluateHypotheses / reevaluateHypothesis / labelHypothesis (hand_search.cpp:66-134,
// 190-228): every hand is checked again against this (ground-truth) cloud with its own frame, depth
// (top) and finger placement; labels[i] = 1 for a full antipodal grasp, the hands' half/full flags are
// rewritten.  A hand without a finger placement (index < 0) would index out of bounds in the
// reference; here it is labelled 0.
void gpd_oracle_reevaluate(const gpd_params *P, const float *xyz, const float *normals, int np, gpd_hand *hands, int n, int32_t *labels) {
  Grid g;
  g.build(xyz, np, 0.02f);
  const double radius = nnRadiusHands(*P);
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < n; i++) {
    gpd_hand &H = hands[i];
    labels[i] = 0;
    H.half_antipodal = 0;
    H.full_antipodal = 0;
    const int idx = H.finger_placement_index;
    if (idx < 0 || idx >= P->num_finger_placements) continue;
    float q[3] = {(float)H.sample[0], (float)H.sample[1], (float)H.sample[2]};
    std::vector<Neighbour> nb;
    radiusSearch(g, q, radius, nb);
    if (nb.empty()) continue;
    FramePts pts;
    handFramePoints(*P, xyz, normals, nb, H.sample, H.frame, pts);
    FingerHand fh(P->finger_width, P->hand_outer_diameter, P->hand_depth, P->num_finger_placements);
    fh.evaluateFingers(pts, H.top, idx);
    fh.evaluateHand(idx);
    if (!fh.any()) continue;
    std::vector<int> closing = fh.closingRegion(pts, -1);
    if (closing.empty()) continue;
    const int label = antipodalLabel(pts, closing, P->friction_coeff, P->min_viable);
    if (label == 2) {
      labels[i] = 1;
      H.full_antipodal = 1;
    } else if (label == 1) {
      H.half_antipodal = 1;
    }
  }
}

// detectGrasps step 2 (grasp_detector.cpp:236-255): the workspace / aperture filter, then the approach-direction
// filter when the cfg asks for it
void gpd_oracle_filter(const gpd_params *P, gpd_hand *hands, int n_sets) {
  filterWorkspace(*P, hands, n_sets, P->num_hand_axes * P->num_orientations);
  if (P->filter_approach_direction) filterDirection(*P, hands, n_sets, P->num_hand_axes * P->num_orientations);
}

// ImageGenerator::createImages (image_generator.cpp:17-99).  Sets without a
// valid hand are skipped (filterGraspsWorkspace drops them before this stage,
// so they consume no shadow draws).
int gpd_oracle_images(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *cam_source, int n_cams,
                      const double *view_points, const gpd_hand *hands, int n_sets, uint8_t *images, int32_t *cand_index, int *num_cand) {
  const int n_slots = P->num_hand_axes * P->num_orientations;
  const int C = P->image_num_channels;
  const size_t img_bytes = (size_t)NPIX * C;
  Grid g;
  g.build(xyz, np, 0.02f);
  const double radius = nnRadiusImages(*P);
  // pass 1 (serial, cheap): which sets are live, image offsets, LCG offsets
  std::vector<int> live;
  std::vector<size_t> img_off;
  size_t n_img = 0;
  for (int s = 0; s < n_sets; s++) {
    int nv = 0;
    for (int j = 0; j < n_slots; j++) nv += hands[(size_t)s * n_slots + j].valid ? 1 : 0;
    if (!nv) continue;
    live.push_back(s);
    img_off.push_back(n_img);
    for (int j = 0; j < n_slots; j++)
      if (hands[(size_t)s * n_slots + j].valid) {
        if (cand_index) cand_index[n_img] = s * n_slots + j;
        n_img++;
      }
  }
  std::vector<std::vector<Neighbour>> nbrs(live.size());
#pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < (int)live.size(); k++) {
    const gpd_hand &h0 = hands[(size_t)live[k] * n_slots];
    float q[3] = {(float)h0.sample[0], (float)h0.sample[1], (float)h0.sample[2]};
    radiusSearch(g, q, radius, nbrs[k]);
  }
  // (g_lcg_base: the draws consumed before this call's first hand set — 0 for a whole cloud, as the reference's seed_ = 0 of a
  //  fresh process; a sample RANGE of a cloud continues the one stream of hand_set.cpp:268-283 where the ranges before it stopped)
  std::vector<uint64_t> lcg_off(live.size() + 1, g_lcg_base);
  const int num_shadow = (int)std::floor(radius / 0.003);  // as calculateShadow evaluates it
  if (C == 15)
    for (size_t k = 0; k < live.size(); k++) {
      uint64_t d = 0;
      const int N = (int)nbrs[k].size();
      for (int c = 0; c < n_cams; c++) {
        long seen = 0;
        for (int i = 0; i < N; i++) seen += cam_source[(size_t)c * np + nbrs[k][i].idx];
        // calculateShadowForCamera draws N * num_shadow_points values (hand_set.cpp:202-212),
        // num_shadow_points = floor(shadow_length / 0.003): 33 only for the default 0.10 m volume
        if (seen >= 1) d += (uint64_t)N * (uint64_t)num_shadow;
      }
      lcg_off[k + 1] = lcg_off[k] + d;
    }
  g_lcg_draws = lcg_off.back() - g_lcg_base;
  if (images) {
#pragma omp parallel for schedule(dynamic, 2)
    for (int k = 0; k < (int)live.size(); k++) {
      std::vector<Voxel> shadow;
      if (C == 15 && !nbrs[k].empty()) {
        Lcg rng{0};
        rng.jump(lcg_off[k]);
        calculateShadow(xyz, np, cam_source, n_cams, view_points, nbrs[k], radius, rng, shadow);  // shadow_length_ (image_15_channels_strategy.h:70-75)
      }
      size_t o = img_off[k];
      for (int j = 0; j < n_slots; j++) {
        const gpd_hand &h = hands[(size_t)live[k] * n_slots + j];
        if (!h.valid) continue;
        std::memset(images + o * img_bytes, 0, img_bytes);
        createImage(*P, h, xyz, normals, nbrs[k], shadow, images + o * img_bytes);
        o++;
      }
    }
  }
  *num_cand = (int)n_img;
  return 0;
}

// EigenClassifier::classifyImages — eigen_classifier.cpp:59-79
void gpd_oracle_lenet(const uint8_t *images, int n, int C, const float *c1w, const float *c1b, const float *c2w, const float *c2b,
                      const float *f1w, const float *f1b, const float *f2w, const float *f2b, float *scores) {
  LeNetW w{C, c1w, c1b, c2w, c2b, f1w, f1b, f2w, f2b};
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < n; i++) scores[i] = lenetForward(w, images + (size_t)i * NPIX * C);
}

// ConvLayer::forward on one CHW float image (test_conv_layer.cpp KAT); 5x5 kernels only
// are used by the net, so the KAT (3x3) goes through a generic direct form here.
void gpd_oracle_conv_generic(const float *x, int C, int H, int W, const float *w, const float *b, int F, int K, float *out) {
  const int OH = H - K + 1, OW = W - K + 1;
  for (int f = 0; f < F; f++)
    for (int oy = 0; oy < OH; oy++)
      for (int ox = 0; ox < OW; ox++) {
        float acc = 0.f;
        for (int c = 0; c < C; c++)
          for (int kh = 0; kh < K; kh++)
            for (int kw = 0; kw < K; kw++) acc = fmaf(w[((f * C + c) * K + kh) * K + kw], x[(c * H + oy + kh) * W + ox + kw], acc);
        out[(f * OH + oy) * OW + ox] = acc + b[f];
      }
}

// detectGrasps steps 1-4 (grasp_detector.cpp:222-273) with stage timers; the CPU
// baseline of bench.py.  images_out may be NULL (then a scratch buffer is used).
// times: [0] candidates, [1] images, [2] classify (seconds).
int gpd_oracle_detect(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *cam_source, int n_cams,
                      const double *view_points, const int32_t *sample_idx, int S, const float *const *weights, gpd_hand *hands, int *num_sets,
                      int *num_cand, uint8_t *images_out, int max_cand, double *times) {
  double t0 = now_s();
  gpd_oracle_search(P, xyz, normals, np, sample_idx, S, hands, num_sets);
  gpd_oracle_filter(P, hands, *num_sets);
  double t1 = now_s();
  const int n_slots = P->num_hand_axes * P->num_orientations;
  size_t nv = 0;
  for (size_t i = 0; i < (size_t)(*num_sets) * n_slots; i++) nv += hands[i].valid;
  if (max_cand > 0 && nv > (size_t)max_cand) {  // truncate to the first max_cand candidates (set,slot order)
    size_t seen = 0;
    for (size_t i = 0; i < (size_t)(*num_sets) * n_slots; i++)
      if (hands[i].valid && ++seen > (size_t)max_cand) hands[i].valid = 0;
    nv = max_cand;
  }
  const size_t img_bytes = (size_t)NPIX * P->image_num_channels;
  std::vector<uint8_t> scratch;
  uint8_t *images = images_out;
  if (!images) {
    scratch.resize(nv * img_bytes);
    images = scratch.data();
  }
  std::vector<int32_t> cand(nv);
  gpd_oracle_images(P, xyz, normals, np, cam_source, n_cams, view_points, hands, *num_sets, images, cand.data(), num_cand);
  double t2 = now_s();
  std::vector<float> scores(nv);
  gpd_oracle_lenet(images, (int)nv, P->image_num_channels, weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6],
                   weights[7], scores.data());
  for (size_t i = 0; i < nv; i++) hands[cand[i]].score = scores[i];
  double t3 = now_s();
  if (times) {
    times[0] = t1 - t0;
    times[1] = t2 - t1;
    times[2] = t3 - t2;
  }
  return 0;
}

// Cloud::voxelizeCloud (util/cloud.cpp:286-348) without normals.  The reference keys a std::set
// with UniqueVector4First3Comparator (util/cloud.h:105-122), which returns "the first three
// elements differ" — not a strict weak ordering, so which points survive depends on the
// red-black tree of libstdc++'s std::set.  Restated by using std::set with the same predicate
// (SURVEY §9-K KAT: tutorials/krylon.pcd 4467 -> 3366 points).  out_src: input index kept per voxel.
struct VoxelDiffers {
  bool operator()(const std::array<int, 4> &a, const std::array<int, 4> &b) const {
    for (int i = 0; i < 3; i++)
      if (a[i] != b[i]) return true;
    return false;
  }
};
int gpd_oracle_voxelize(const float *xyz, int P, float cell_size, float *out_xyz, int32_t *out_src) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (int i = 0; i < P; i++)
    for (int c = 0; c < 3; c++) mn[c] = std::min(mn[c], xyz[3 * i + c]);
  std::set<std::array<int, 4>, VoxelDiffers> bins;
  for (int i = 0; i < P; i++) {
    std::array<int, 4> v;
    for (int c = 0; c < 3; c++) v[c] = (int)std::floor((xyz[3 * i + c] - mn[c]) / cell_size);
    v[3] = i;
    bins.insert(v);
  }
  int n = 0;
  for (const auto &v : bins) {
    for (int c = 0; c < 3; c++) out_xyz[3 * n + c] = mn[c] + cell_size * (float)v[c];
    if (out_src) out_src[n] = v[3];
    n++;
  }
  return n;
}

// Cloud::calculateNormals (util/cloud.cpp:451-476): pcl::NormalEstimationOMP with a radius search
// (:497-535) followed by Cloud::reverseNormals (:573-604).  PCL is absent; its semantics restated:
// neighbours of the point within `radius` among ALL points (FLANN order), covariance of the
// neighbour coordinates about their centroid, eigenvector of the smallest eigenvalue, flipped
// towards the view point of the camera that sees the point (flipNormalTowardsViewpoint), stored
// as float.  Oracle choices (PCL's float accumulators and closed-form eigen33 are unpinned):
// centroid and covariance as sequential fp64 sums in neighbour order, the same 3x3 QR eigensolver
// as the local frames.
void gpd_oracle_normals(const float *xyz, int P, const int32_t *cam_source, int n_cams, const double *view_points, double radius,
                        float *normals_out) {
  Grid g;
  g.build(xyz, P, 0.02f);
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < P; i++) {
    std::vector<Neighbour> nb;
    radiusSearch(g, xyz + 3 * i, radius, nb);
    const int k = (int)nb.size();
    double c[3] = {0, 0, 0};
    for (int n = 0; n < k; n++)
      for (int r = 0; r < 3; r++) c[r] += (double)xyz[3 * nb[n].idx + r];
    for (int r = 0; r < 3; r++) c[r] /= (double)k;
    double M[9] = {0};
    for (int n = 0; n < k; n++) {
      const double d0 = (double)xyz[3 * nb[n].idx] - c[0], d1 = (double)xyz[3 * nb[n].idx + 1] - c[1], d2 = (double)xyz[3 * nb[n].idx + 2] - c[2];
      M[0] += d0 * d0; M[3] += d1 * d0; M[4] += d1 * d1; M[6] += d2 * d0; M[7] += d2 * d1; M[8] += d2 * d2;
    }
    M[1] = M[3]; M[2] = M[6]; M[5] = M[7];
    double ev[3], V[9];
    selfAdjointEigen3(M, ev, V);
    int mn = 0;
    for (int q = 1; q < 3; q++)
      if (ev[q] < ev[mn]) mn = q;
    double nrm[3] = {V[mn], V[3 + mn], V[6 + mn]};
    // every point is estimated once, with the view point of the FIRST camera that sees it
    // (convertCameraSourceMatrixToLists, cloud.cpp:606-620: `== 1` and a break); setViewPoint takes floats (cloud.cpp:513)
    for (int cam = 0; cam < n_cams; cam++) {
      if (cam_source[(size_t)cam * P + i] != 1) continue;
      double t[3] = {nrm[0], nrm[1], nrm[2]};
      const double *vp = view_points + 3 * cam;
      const double dot = ((double)(float)vp[0] - (double)xyz[3 * i]) * t[0] + ((double)(float)vp[1] - (double)xyz[3 * i + 1]) * t[1] +
                         ((double)(float)vp[2] - (double)xyz[3 * i + 2]) * t[2];
      if (dot < 0)
        for (int r = 0; r < 3; r++) t[r] = -t[r];
      for (int r = 0; r < 3; r++) normals_out[3 * i + r] = (float)t[r];
      break;
    }
    // reverseNormals: reverse unless some seeing camera has normal . (p - vp) < 0
    bool needs_reverse = true;
    for (int cam = 0; cam < n_cams && needs_reverse; cam++) {
      if (cam_source[(size_t)cam * P + i] != 1) continue;
      const double *vp = view_points + 3 * cam;
      const double d = (double)normals_out[3 * i] * ((double)xyz[3 * i] - vp[0]) + (double)normals_out[3 * i + 1] * ((double)xyz[3 * i + 1] - vp[1]) +
                       (double)normals_out[3 * i + 2] * ((double)xyz[3 * i + 2] - vp[2]);
      if (d < 0) needs_reverse = false;
    }
    if (needs_reverse)
      for (int r = 0; r < 3; r++) normals_out[3 * i + r] = (float)((double)normals_out[3 * i + r] * -1.0);
  }
}

int gpd_oracle_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
// sample-range sharding of ONE cloud: the shadow draws consumed before the next gpd_oracle_images call's first hand
// set, and the draws its hand sets consumed (tests/test_multi_gpu_gloo.py: two ranks, a host-side scan of the totals)
void gpd_oracle_set_lcg_base(uint64_t base) { g_lcg_base = base; }
uint64_t gpd_oracle_last_lcg_draws(void) { return g_lcg_draws; }

void gpd_oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}


// Clustering::findClusters — clustering.cpp:5-105 (step 6 of detectGrasps, grasp_detector.cpp:283-303).
// hands: n records whose frame column 2 is the hand axis and `position` the hand position; scores
// as doubles (Hand::score_ is a double in the reference).  Output record k is a copy of seed
// src[k] with position moved to the inliers' mean position and out_scores[k] = lower 99 %
// confidence bound; returns the number of clusters.
int gpd_oracle_find_clusters(const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers, gpd_hand *out,
                             double *out_scores, int32_t *out_src) {
  const double AXIS_ALIGN_ANGLE_THRESH = 12.0 * M_PI / 180.0;  // clustering.cpp:9
  const double AXIS_ALIGN_DIST_THRESH = 0.005;                 // :10
  const double MAX_DIST_THRESH = 0.05;                         // :12
  std::vector<char> used(n, 0);
  int n_out = 0;
  auto axis = [&](int i, int r) { return hands[i].frame[3 * r + 2]; };
  for (int i = 0; i < n; i++) {
    int num_inliers = 0;
    double pos_sum[3] = {0, 0, 0};
    double outer[9];  // axis * axis^T (:29-30)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) outer[3 * r + c] = axis(i, r) * axis(i, c);
    double mean = 0.0, sd = 0.0;
    for (int j = 0; j < n; j++) {
      if (i == j || (remove_inliers && used[j])) continue;
      double dot = 0.0;  // :39-42
      for (int r = 0; r < 3; r++) dot += axis(i, r) * axis(j, r);
      const bool aligned = std::fabs(dot) > std::cos(AXIS_ALIGN_ANGLE_THRESH);
      double delta[3], mag2 = 0.0;  // :45-48
      for (int r = 0; r < 3; r++) {
        delta[r] = hands[i].position[r] - hands[j].position[r];
        mag2 += delta[r] * delta[r];
      }
      const bool near = std::sqrt(mag2) <= MAX_DIST_THRESH;
      double pm2 = 0.0;  // :53-58: (I - a a^T) * delta
      for (int r = 0; r < 3; r++) {
        double v = 0.0;
        for (int c = 0; c < 3; c++) v += ((r == c ? 1.0 : 0.0) - outer[3 * r + c]) * delta[c];
        pm2 += v * v;
      }
      const bool on_axis = std::sqrt(pm2) <= AXIS_ALIGN_DIST_THRESH;
      if (aligned && near && on_axis) {  // :60-76
        num_inliers++;
        for (int r = 0; r < 3; r++) pos_sum[r] += hands[j].position[r];
        const double old_mean = mean;
        mean += (scores[j] - mean) / (double)num_inliers;
        sd += (scores[j] - mean) * (scores[j] - old_mean);
        if (remove_inliers) used[j] = 1;
      }
    }
    if (num_inliers >= min_inliers) {  // :79-101
      const double dn = (double)num_inliers;
      double pd[3];
      for (int r = 0; r < 3; r++) pd[r] = pos_sum[r] / dn - hands[i].position[r];
      sd /= dn;
      if (sd != 0) sd = std::sqrt(sd);
      const double conf_lb = mean - 2.576 * sd / std::sqrt((double)num_inliers);
      out[n_out] = hands[i];
      for (int r = 0; r < 3; r++) out[n_out].position[r] = hands[i].position[r] + pd[r];
      out[n_out].score = (float)conf_lb;
      out_scores[n_out] = conf_lb;
      out_src[n_out] = i;
      n_out++;
    }
  }
  return n_out;
}

// GraspDetector::selectGrasps — grasp_detector.cpp:405-420: std::partial_sort of the hands by
// isScoreGreater (grasp_detector.h: hand1->getScore() > hand2->getScore()), the first
// min(num_selected, n) kept.  The algorithm only sees comparator outcomes, so sorting (score, index)
// pairs in the hands' order reproduces the arrangement the reference's libstdc++ build leaves — also
// among equal scores.  out_idx receives the indices of the kept hands, in the kept order.
int gpd_oracle_select(const float *scores, int n, int num_selected, int32_t *out_idx) {
  std::vector<std::pair<float, int32_t>> v((size_t)n);
  for (int i = 0; i < n; i++) v[i] = {scores[i], i};
  const int middle = std::min(n, num_selected);
  std::partial_sort(v.begin(), v.begin() + middle, v.end(),
                    [](const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) { return a.first > b.first; });
  for (int i = 0; i < middle; i++) out_idx[i] = v[i].second;
  return middle;
}

}  // extern "C"
