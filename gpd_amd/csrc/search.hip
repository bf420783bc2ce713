// Candidate search on gfx950: radius neighbourhoods, local frames, hand evaluation.
//
// Replaces HandSearch::searchHands / evalHands (candidate/hand_search.cpp:24-64,
// 144-188), FrameEstimator::calculateFrame (candidate/frame_estimator.cpp:66-86),
// LocalFrame::findAverageNormalAxis (candidate/local_frame.cpp:14-41),
// HandSet::evalHandSet/evalHands (candidate/hand_set.cpp:31-116), FingerHand
// (candidate/finger_hand.cpp:6-184), Hand::construct (candidate/hand.cpp:24-45),
// HandSet::modifyCandidate/labelHypothesis (hand_set.cpp:235-261) and
// Antipodal::evaluateGrasp (candidate/antipodal.cpp:10-96).
//
// neighbourhood_kernel   one workgroup per sample: visits the cells of a uniform 2 cm grid
//     around the sample (instead of walking PCL's k-d tree), collects the points with
//     d2 < r^2 (r = the largest of the hand / image / frame radii) in LDS and sorts them
//     by (d2, index) — FLANN's order — with a 1024-bucket counting sort whose buckets are
//     ordered through registers (bitonic sort in place for the 16384-entry retry); writes
//     the sorted neighbourhood gathered as SoA.  The three neighbourhoods are prefixes of
//     this list (same query point, same float d2, same sort).  Lane 0 then forms
//     M = sum n n^T in neighbour order and runs the 3x3 symmetric QR eigensolver
//     (Eigen's SelfAdjointEigenSolver algorithm, fp64, unfused).
// centre_kernel          centre of the image neighbourhood, one lane per serial fp64 chain.
// hand_eval_kernel       one workgroup per (sample, orientation): one transform pass compacts
//     the in-height points into LDS, then order-free reductions over them (finger
//     collision masks, deepen masks, closing region, antipodal extremes, antipodal
//     counts).  The reference's cropByHandHeight quirk (point_list.cpp:44-55 pads with
//     copies of column 0) is reproduced by a ghost point with multiplicity N-k.
// reeval_kernel          HandSearch::reevaluateHypotheses (hand_search.cpp:66-134, 190-228).
// normals_*_kernel       Cloud::calculateNormals (util/cloud.cpp:451-604): count, scan, lists, finish.
//
// All fp64 expressions are evaluated unfused, left to right (-ffp-contract=off),
// exactly as oracle/gpd_oracle.cpp defines them.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include <mutex>

#include "gpd_internal.h"

namespace gpd {

#define HIP_RET(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return GPD_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

// ---------------------------------------------------------------------------
// Cloud upload: AoS (caller layout) -> SoA planes.
// ---------------------------------------------------------------------------
// pxyz.w carries the first camera's flag of the point (bits of the int): the gather of neighbourhood_kernel then needs no
// third random access per neighbour (that phase is bound by the address unit: one lane per cycle and load)
__global__ void split_soa_kernel(const float *__restrict__ xyz, const float *__restrict__ nrm, const int32_t *__restrict__ cam0, int n,
                                 float *px, float *py, float *pz, float *nx, float *ny, float *nz, float4 *pxyz, float4 *pnrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  const float a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
  px[i] = x;
  py[i] = y;
  pz[i] = z;
  nx[i] = a;
  ny[i] = b;
  nz[i] = c;
  pxyz[i] = make_float4(x, y, z, __int_as_float(cam0[i]));
  pnrm[i] = make_float4(a, b, c, 0.f);
}

// ---- uniform grid ----------------------------------------------------------------------
__device__ inline int grid_coord(const GridView &g, int axis, float v) {
  int k = (int)floorf((v - g.lo[axis]) / g.cell);
  return k < 0 ? 0 : (k > g.dim[axis] - 1 ? g.dim[axis] - 1 : k);
}
__global__ void grid_count_kernel(GridView g, const float *px, const float *py, const float *pz, int n, int32_t *counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (grid_coord(g, 0, px[i]) * g.dim[1] + grid_coord(g, 1, py[i])) * g.dim[2] + grid_coord(g, 2, pz[i]);
  atomicAdd(&counts[c], 1);
}
// exclusive scan of counts[0..n) into start[0..n], single workgroup with a running carry
__global__ __launch_bounds__(1024) void grid_scan_kernel(const int32_t *counts, int32_t *start, int32_t *cursor, int n) {
  __shared__ int s_part[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(incl, o);
      if (lane >= o) incl += x;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int off = s_carry;
    for (int w = 0; w < wave; w++) off += s_part[w];
    if (i < n) {
      start[i] = off + incl - v;
      cursor[i] = off + incl - v;
    }
    __syncthreads();
    if (tid == 1023) s_carry = off + incl;
    __syncthreads();
  }
  if (tid == 0) start[n] = s_carry;
}
__global__ void grid_scatter_kernel(GridView g, const float *px, const float *py, const float *pz, int n, int32_t *cursor,
                                    float4 *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (grid_coord(g, 0, px[i]) * g.dim[1] + grid_coord(g, 1, py[i])) * g.dim[2] + grid_coord(g, 2, pz[i]);
  const int pos = atomicAdd(&cursor[c], 1);
  out[pos] = make_float4(px[i], py[i], pz[i], __int_as_float(i));
}

// Visits every point of the cells overlapping the cube of half-edge `reach` around q: calls
// visit(original index, x, y, z).  Wave w takes (x, y) cell columns w, w + nwaves, ...; a column's
// cells are contiguous in memory (z fastest), its lanes stride the point range.  All lanes of a
// wave run the same number of iterations, so `visit` may use wave-wide operations.
template <class Visit>
__device__ inline void grid_visit(const GridView &g, float qx, float qy, float qz, float reach, int wave, int nwaves, int lane,
                                  Visit visit) {
  const int x0 = grid_coord(g, 0, qx - reach), x1 = grid_coord(g, 0, qx + reach);
  const int y0 = grid_coord(g, 1, qy - reach), y1 = grid_coord(g, 1, qy + reach);
  const int z0 = grid_coord(g, 2, qz - reach), z1 = grid_coord(g, 2, qz + reach);
  const int ny = y1 - y0 + 1;
  const int ncol = (x1 - x0 + 1) * ny;
  for (int col = wave; col < ncol; col += nwaves) {
    const int cx = x0 + col / ny, cy = y0 + col % ny;
    const int cbase = (cx * g.dim[1] + cy) * g.dim[2];
    const int b = g.start[cbase + z0], e = g.start[cbase + z1 + 1];
    for (int t0 = b; t0 < e; t0 += 64) {
      const int t = t0 + lane;
      const bool in = t < e;
      const int tt = in ? t : b;
      const float4 p = g.p[tt];
      visit(in, __float_as_int(p.w), p.x, p.y, p.z);
    }
  }
}

void cloud_free(Cloud &c) {
  normals_free(c.normals);
  void *ptrs[] = {c.px, c.py, c.pz, c.nx, c.ny, c.nz, c.cam_source, c.staging, c.g_start, c.g_cursor, c.g_p, c.pxyz, c.pnrm};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (c.h_pin) (void)hipHostFree(c.h_pin);
  c = Cloud();
}

// Device + pinned staging buffers for clouds of up to n points / num_cams cameras (grows with 25 % slack, never
// shrinks).  A growth is hipFree + hipMalloc, i.e. a device stall: gpd_hip_reserve / gpd_hip_detect_batch size the lanes
// once, ahead of the first cloud.
int cloud_reserve(Cloud &c, int n, int num_cams) {
  if (n <= c.capacity && num_cams <= c.cap_cams) return GPD_OK;
  note_alloc(__func__);
  const uint64_t gen = c.generation;
  const int cap = n > c.capacity ? n + n / 4 : c.capacity;  // slack: clouds of a batch differ by a few points
  const int cams = num_cams > c.cap_cams ? num_cams : c.cap_cams;
  const int cells_cap = c.g_cells_cap;
  cloud_free(c);  // hipFree waits for the device: kernels of an earlier cloud on this stream are done
  c.generation = gen;
  float **planes[] = {&c.px, &c.py, &c.pz, &c.nx, &c.ny, &c.nz};
  for (float **p : planes) HIP_RET(hipMalloc(p, (size_t)cap * sizeof(float)));
  float4 **quads[] = {&c.g_p, &c.pxyz, &c.pnrm};
  for (float4 **p : quads) HIP_RET(hipMalloc(p, (size_t)cap * sizeof(float4)));
  HIP_RET(hipMalloc(&c.cam_source, (size_t)cap * cams * sizeof(int32_t)));
  HIP_RET(hipMalloc(&c.staging, ((size_t)cap * 6 + 8) * sizeof(float)));  // + 8: the bounds of a cloud handed over on the device
  HIP_RET(hipHostMalloc(reinterpret_cast<void **>(&c.h_pin), (size_t)cap * (6 * sizeof(float) + cams * sizeof(int32_t)) + 32, 0));
  c.capacity = cap;
  c.cap_cams = cams;
  if (cells_cap > 0) {  // the grid tables keep their size across a growth of the point buffers
    HIP_RET(hipMalloc(&c.g_start, (size_t)(cells_cap + 1) * sizeof(int32_t)));
    HIP_RET(hipMalloc(&c.g_cursor, (size_t)cells_cap * sizeof(int32_t)));
    c.g_cells_cap = cells_cap;
  }
  return GPD_OK;
}

// The uniform grid's tables for scenes of up to `cells` cells of 2 cm (a 2 x 2 x 1 m scene has 500 000); a larger scene grows them.
int cloud_reserve_grid(Cloud &c, int cells) {
  if (cells <= c.g_cells_cap) return GPD_OK;
  note_alloc(__func__);
  if (c.g_start) (void)hipFree(c.g_start);
  if (c.g_cursor) (void)hipFree(c.g_cursor);
  c.g_start = nullptr;
  c.g_cursor = nullptr;
  c.g_cells_cap = 0;
  HIP_RET(hipMalloc(&c.g_start, (size_t)(cells + 1) * sizeof(int32_t)));
  HIP_RET(hipMalloc(&c.g_cursor, (size_t)cells * sizeof(int32_t)));
  c.g_cells_cap = cells;
  return GPD_OK;
}

// The caller's arrays are copied into a pinned staging buffer (one pass that also takes the bounds of the
// uniform grid and rejects non-finite coordinates), so the three host-to-device copies are truly
// asynchronous: with sync == false nothing here waits for the device and the upload of the next cloud
// overlaps the kernels of the current one (gpd_hip_detect_batch).
int cloud_upload(Cloud &c, const float *xyz, const float *normals, int n, const int32_t *cam_source, int num_cams,
                 const double *view_points, hipStream_t stream, bool sync) {
  if (num_cams > kMaxCams) {
    set_error("upload_cloud: at most %d cameras are supported", kMaxCams);
    return GPD_ERR_INVALID;
  }
  {
    const int rc = cloud_reserve(c, n, num_cams);
    if (rc) return rc;
  }
  float *hx = reinterpret_cast<float *>(c.h_pin), *hn = hx + (size_t)3 * n;
  int32_t *hc = reinterpret_cast<int32_t *>(hx + (size_t)6 * n);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool finite = true;
  for (int i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) {
      const float v = xyz[3 * (size_t)i + a];
      hx[3 * (size_t)i + a] = v;
      finite &= std::isfinite(v);
      lo[a] = v < lo[a] ? v : lo[a];
      hi[a] = v > hi[a] ? v : hi[a];
    }
  if (!finite) {  // pcl::removeNaNFromPointCloud runs before the path (candidates_generator.cpp:17); an Inf would
                  // make the grid dimensions undefined
    set_error("upload_cloud: the cloud holds non-finite coordinates (remove NaN/Inf points first)");
    return GPD_ERR_INVALID;
  }
  std::memcpy(hn, normals, (size_t)n * 3 * sizeof(float));
  std::memcpy(hc, cam_source, (size_t)n * num_cams * sizeof(int32_t));
  c.num_points = n;
  c.num_cams = num_cams;
  c.generation++;
  std::memcpy(c.view_points, view_points, sizeof(double) * 3 * num_cams);
  HIP_RET(hipMemcpyAsync(c.staging, hx, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, stream));
  HIP_RET(hipMemcpyAsync(c.cam_source, hc, (size_t)n * num_cams * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  split_soa_kernel<<<(n + 255) / 256, 256, 0, stream>>>(c.staging, c.staging + (size_t)n * 3, c.cam_source, n, c.px, c.py, c.pz, c.nx,
                                                         c.ny, c.nz, c.pxyz, c.pnrm);
  HIP_RET(hipGetLastError());
  // uniform grid: bounds from the pass above, counting sort on the device
  c.g_cell = 0.02f;
  for (;;) {  // at most 256 cells per axis
    bool ok = true;
    for (int a = 0; a < 3; a++) {
      c.g_lo[a] = lo[a];
      c.g_dim[a] = (int)std::floor((hi[a] - lo[a]) / c.g_cell) + 1;
      if (c.g_dim[a] > 256 || c.g_dim[a] < 1) ok = false;
    }
    if (ok) break;
    c.g_cell *= 2.f;
  }
  const int cells = c.g_dim[0] * c.g_dim[1] * c.g_dim[2];
  if (cells > c.g_cells_cap) {
    const int rc = cloud_reserve_grid(c, cells + cells / 2);
    if (rc) return rc;
  }
  GridView g = grid_view(c);
  HIP_RET(hipMemsetAsync(c.g_cursor, 0, (size_t)cells * sizeof(int32_t), stream));
  grid_count_kernel<<<(n + 255) / 256, 256, 0, stream>>>(g, c.px, c.py, c.pz, n, c.g_cursor);
  grid_scan_kernel<<<1, 1024, 0, stream>>>(c.g_cursor, c.g_start, c.g_cursor, cells);
  grid_scatter_kernel<<<(n + 255) / 256, 256, 0, stream>>>(g, c.px, c.py, c.pz, n, c.g_cursor, c.g_p);
  HIP_RET(hipGetLastError());
  if (sync) HIP_RET(hipStreamSynchronize(stream));
  return GPD_OK;
}

// min / max of n points [n][3] on the device -> out[6] (one workgroup: the clouds handed over on the device are the voxelised ones)
__global__ __launch_bounds__(1024) void bounds_kernel(const float *__restrict__ xyz, int n, float *__restrict__ out) {
  __shared__ float s_lo[16][3], s_hi[16][3];
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = threadIdx.x; i < n; i += 1024)
    for (int a = 0; a < 3; a++) {
      const float v = xyz[3 * (size_t)i + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  for (int a = 0; a < 3; a++)
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++) {
      s_lo[threadIdx.x >> 6][a] = lo[a];
      s_hi[threadIdx.x >> 6][a] = hi[a];
    }
  __syncthreads();
  if (threadIdx.x < 3) {
    float l = FLT_MAX, h = -FLT_MAX;
    for (int w = 0; w < 16; w++) {
      l = fminf(l, s_lo[w][threadIdx.x]);
      h = fmaxf(h, s_hi[w][threadIdx.x]);
    }
    out[threadIdx.x] = l;
    out[3 + threadIdx.x] = h;
  }
}

// The cloud from arrays that are ALREADY on the device (the output of the preprocessing kernels: finite by construction):
// d_xyz [n][3], d_cam [cams][n]; normals zero until normals_run fills them.  One small wait: the bounds of the grid.
int cloud_from_device(Cloud &c, const float *d_xyz, const int32_t *d_cam, int n, int num_cams, const double *view_points, hipStream_t stream) {
  if (num_cams > kMaxCams || num_cams < 1 || n < 1) {
    set_error("cloud_from_device: bad argument");
    return GPD_ERR_INVALID;
  }
  {
    const int rc = cloud_reserve(c, n, num_cams);
    if (rc) return rc;
  }
  c.num_points = n;
  c.num_cams = num_cams;
  c.generation++;
  std::memcpy(c.view_points, view_points, sizeof(double) * 3 * num_cams);
  HIP_RET(hipMemcpyAsync(c.staging, d_xyz, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
  HIP_RET(hipMemsetAsync(c.staging + (size_t)n * 3, 0, (size_t)n * 3 * sizeof(float), stream));
  HIP_RET(hipMemcpyAsync(c.cam_source, d_cam, (size_t)n * num_cams * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  float *hb = reinterpret_cast<float *>(c.h_pin);  // the pinned staging is free: nothing of this cloud came through it
  bounds_kernel<<<1, 1024, 0, stream>>>(c.staging, n, c.staging + (size_t)c.capacity * 6);
  HIP_RET(hipMemcpyAsync(hb, c.staging + (size_t)c.capacity * 6, 6 * sizeof(float), hipMemcpyDeviceToHost, stream));
  split_soa_kernel<<<(n + 255) / 256, 256, 0, stream>>>(c.staging, c.staging + (size_t)n * 3, c.cam_source, n, c.px, c.py, c.pz, c.nx,
                                                         c.ny, c.nz, c.pxyz, c.pnrm);
  HIP_RET(hipGetLastError());
  HIP_RET(hipStreamSynchronize(stream));
  const float *lo = hb, *hi = hb + 3;
  c.g_cell = 0.02f;
  for (;;) {  // at most 256 cells per axis
    bool ok = true;
    for (int a = 0; a < 3; a++) {
      c.g_lo[a] = lo[a];
      c.g_dim[a] = (int)std::floor((hi[a] - lo[a]) / c.g_cell) + 1;
      if (c.g_dim[a] > 256 || c.g_dim[a] < 1) ok = false;
    }
    if (ok) break;
    c.g_cell *= 2.f;
  }
  const int cells = c.g_dim[0] * c.g_dim[1] * c.g_dim[2];
  if (cells > c.g_cells_cap) {
    const int rc = cloud_reserve_grid(c, cells + cells / 2);
    if (rc) return rc;
  }
  GridView g = grid_view(c);
  HIP_RET(hipMemsetAsync(c.g_cursor, 0, (size_t)cells * sizeof(int32_t), stream));
  grid_count_kernel<<<(n + 255) / 256, 256, 0, stream>>>(g, c.px, c.py, c.pz, n, c.g_cursor);
  grid_scan_kernel<<<1, 1024, 0, stream>>>(c.g_cursor, c.g_start, c.g_cursor, cells);
  grid_scatter_kernel<<<(n + 255) / 256, 256, 0, stream>>>(g, c.px, c.py, c.pz, n, c.g_cursor, c.g_p);
  HIP_RET(hipGetLastError());
  return GPD_OK;
}

// ---------------------------------------------------------------------------
// 3x3 symmetric eigensolver, fp64: scale, closed-form tridiagonalisation, implicit
// QR with Wilkinson shift, ascending sort — the algorithm of Eigen's
// SelfAdjointEigenSolver<Matrix3d>::compute (used at local_frame.cpp:17-21).
// Q is row-major, columns are eigenvectors.
// ---------------------------------------------------------------------------
__device__ inline void givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

__device__ inline double hypot_pos(double x, double y) {
  x = fabs(x);
  y = fabs(y);
  double p = fmax(x, y);
  if (p == 0.0) return 0.0;
  double qp = fmin(y, x) / p;
  return p * sqrt(1.0 + qp * qp);
}

__device__ void eigen3(double m00, double m10, double m11, double m20, double m21, double m22, double *eval, double *Q) {
  double scale = fmax(fmax(fmax(fabs(m00), fabs(m10)), fmax(fabs(m11), fabs(m20))), fmax(fabs(m21), fabs(m22)));
  if (scale == 0.0) scale = 1.0;
  m00 /= scale;
  m10 /= scale;
  m11 /= scale;
  m20 /= scale;
  m21 /= scale;
  m22 /= scale;
  double diag[3], sub[2];
  diag[0] = m00;
  const double v1norm2 = m20 * m20;
  if (v1norm2 <= DBL_MIN) {
    diag[1] = m11;
    diag[2] = m22;
    sub[0] = m10;
    sub[1] = m21;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = 1; Q[5] = 0; Q[6] = 0; Q[7] = 0; Q[8] = 1;
  } else {
    const double beta = sqrt(m10 * m10 + v1norm2);
    const double invBeta = 1.0 / beta;
    const double m01 = m10 * invBeta;
    const double m02 = m20 * invBeta;
    const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = m01; Q[5] = m02; Q[6] = 0; Q[7] = m02; Q[8] = -m01;
  }
  int end = 2, start = 0, iter = 0;
  const double precision_inv = 1.0 / DBL_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; i++) {
      if (fabs(sub[i]) < DBL_MIN) {
        sub[i] = 0.0;
      } else {
        const double ss = precision_inv * sub[i];
        if (ss * ss <= (fabs(diag[i]) + fabs(diag[i + 1]))) sub[i] = 0.0;
      }
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > 90) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
      mu -= fabs(e);
    } else if (e != 0.0) {
      const double e2 = e * e;
      const double h = hypot_pos(td, e);
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
      const double sdk = s * diag[k] + c * sub[k];
      const double dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      x = sub[k];
      if (k < end - 1) {
        z = -s * sub[k + 1];
        sub[k + 1] = c * sub[k + 1];
      }
      for (int i = 0; i < 3; i++) {
        const double xi = Q[3 * i + k], yi = Q[3 * i + k + 1];
        Q[3 * i + k] = c * xi - s * yi;
        Q[3 * i + k + 1] = s * xi + c * yi;
      }
    }
  }
  for (int i = 0; i < 2; i++) {
    int k = 0;
    for (int j = 1; j < 3 - i; j++)
      if (diag[i + j] < diag[i + k]) k = j;
    if (k > 0) {
      double t = diag[i];
      diag[i] = diag[k + i];
      diag[k + i] = t;
      for (int r = 0; r < 3; r++) {
        t = Q[3 * r + i];
        Q[3 * r + i] = Q[3 * r + k + i];
        Q[3 * r + k + i] = t;
      }
    }
  }
  for (int i = 0; i < 3; i++) eval[i] = diag[i] * scale;
}

// ---------------------------------------------------------------------------
// neighbourhood_kernel
// ---------------------------------------------------------------------------
// Is the fp64 sum of n floats exact in every order?  emin / emax: smallest / largest biased exponent among the non-zero addends
// (denormals count as exponent 1; no non-zero addend: emin = 255 > emax = 0).  Every addend is a multiple of 2^(emin - 150) and
// below 2^(emax - 126) in magnitude, so every partial sum of every subset is a multiple of the former below n 2^(emax - 126);
// fp64 holds every multiple of 2^L below 2^(L + 53): exact when ceil(log2 n) + emax - 126 <= emin - 150 + 53.  One more bit is
// left as margin; infinities / NaNs (emax = 255) never pass.
__host__ __device__ inline bool centre_exact(int emin, int emax, int n) {
  if (emax == 0) return true;  // nothing but zeros
  if (emax >= 255) return false;
  int lg = 0;
  while ((1ll << lg) < (long long)n) lg++;
  return lg + emax - emin <= 28;
}

struct NbParams {
  const float *px, *py, *pz, *nx, *ny, *nz;
  const float4 *pxyz, *pnrm;  // AoS copies: one 16-byte load per random access
  int num_points;
  const int32_t *sample_idx;
  const double *sample_xyz;  // non-null: samples by coordinates; the query is their float cast (eigenVectorToPcl)
  float r2_all, r2_hands, r2_images, r2_frames;  // r2_all = the largest: one search, the three neighbourhoods are prefixes
  int cap;  // list capacity; a power of two in bitonic mode
  int bucket;  // 1: bucket sort (keys + u16 slot table + 2 x 1024 counters in LDS), 0: in-place bitonic sort
  int32_t *counts;  // [S][8]: N_hands, N_images, k_frames, found, mask of cameras that see the image neighbourhood, -, -, -
  int32_t *nn_idx;
  float *nn;
  double *frames;
  double *centers;  // [S][3] mean of the image neighbourhood (hand_set.cpp:131-133)
  const int32_t *cam_source;
  int num_cams;
  GridView grid;
  float reach;  // half-edge of the cube of cells to visit (radius + margin)
  unsigned long long *dbg;  // profiling aid (GPD_NB_TIMING=1): per-phase cycle sums of wave 0
  // the height list of the hand search (the crop shared by a sample's orientations, see hl_note below), built by the
  // frame wave once the frame is known; hl == nullptr: not wanted (re-evaluation)
  float4 *hl;                       // [S][cap]: x, y, z, rank bits of the points that can be in-height for some orientation
  int hl_slots;
  double hl_col[GPD_MAX_SLOTS][3];  // third column of rot_binormal * rot[slot]: the slot's hand axis in the local frame
  double hl_height, hl_radius;
};

// ---- bucket sort of the (d2, index) keys -------------------------------------------------
// d2 < r2 maps monotonically to one of NB_BUCKETS buckets (about 3-5 keys each at the reference's
// radii); a counting sort groups the point indices bucket by bucket and every bucket is ordered
// through registers.  LDS holds two u32 arrays (visit order / bucket order of the indices; after
// the sort: d2 bits / indices, both sorted) — the same 64 KB as the u64 keys of the bitonic path,
// d2 is recomputed from the coordinates (bit-identical, L2 hits) instead of being stored twice.
// Five barriers instead of the 78 barrier-separated bitonic stages (648 of the kernel's 1320 us).
constexpr int NB_BUCKETS = 1024;
// 512 threads: sixteen waves per CU at two workgroups (72 KB of LDS each) — the kernel waits on dependent global
// loads (cell bounds -> points) and on its barriers, more waves in flight is what hides them
constexpr int NB_THREADS = 512, NB_WAVES = NB_THREADS / 64;
// The keys — (d2 bits << 32) | index, or ~0 for padding — are compared as DOUBLES: a d2 below r2 has float bits under
// 0x3f800000, so the key is the bit pattern of a non-negative finite (possibly denormal: f64 denormals are kept) double,
// and those order like the 64-bit integers; the padding must be a number too (a NaN would be dropped by both v_min_f64 and
// v_max_f64 in favour of the key, which would then appear twice): the largest finite double.  One compare-exchange = v_min_f64 + v_max_f64 instead of v_cmp_gt_u64 + four v_cndmask_b32.
template <int N>
__device__ inline void sort_regs64(unsigned long long (&k)[N]) {
  double d[N];
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = __longlong_as_double((long long)k[i]);
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int i = 0; i < N; i++) {
        const int j = i ^ stride;
        if (j > i) {
          const bool up = (i & size) == 0;
          const double lo = __builtin_fmin(d[i], d[j]), hi = __builtin_fmax(d[i], d[j]);
          d[i] = up ? lo : hi;
          d[j] = up ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) k[i] = (unsigned long long)__double_as_longlong(d[i]);
}
// GLOBAL: the slow path for neighbourhoods beyond the LDS list capacities (the reference has no limit:
// hand_search.cpp:178 takes whatever radiusSearch returns).  The same bucket sort with the two index arrays in
// global memory — they live in the output rows themselves (sorted indices in nn_idx, d2 bits in the last nn row until
// the gather overwrites it) — and, LDS being free, 8192 buckets.
template <bool GLOBAL>
__global__ __launch_bounds__(NB_THREADS) void neighbourhood_kernel(NbParams P) {
  constexpr int NBK = GLOBAL ? 8192 : NB_BUCKETS;  // buckets
  constexpr int PER = NBK / NB_THREADS;           // per lane in the scan
  // workgroup-wide hand-over of the index arrays: a barrier, plus a fence when they are global memory
  auto hand_over = [] {
    if constexpr (GLOBAL) __threadfence_block();
    __syncthreads();
  };
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
  __shared__ int s_count;
  __shared__ int s_bounds[3];
  __shared__ int s_seen;
  __shared__ double s_csum[NB_WAVES][3];  // centre of the image neighbourhood: the waves' partial sums ...
  __shared__ int s_cexp[NB_WAVES][6];     // ... and the smallest / largest biased exponent among their non-zero addends
  __shared__ double s_hl_axis[4];  // height list: hand axis of slot 0 and the limit h + margin (published by the frame wave)
  __shared__ int s_hl_ready, s_hl_next, s_hl_count;
  __shared__ int s_ncrowd;
  // the frame wave's own list of the frame neighbourhood (d2 < r2_frames): keys (d2 bits, index) in visit order, the point
  // indices in FLANN order; s_fkf = their number once the frame is done (more than FCAP: -1, the frame waits for the workgroup's sorted list)
  constexpr int FCAP = 160;
  __shared__ unsigned long long s_fk[FCAP];
  __shared__ int s_fidx[FCAP];
  __shared__ int s_fkf;
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  unsigned long long t_last = __builtin_readcyclecounter();
#define NTICK(k_)                                                    \
  do {                                                              \
    if (P.dbg && tid == 0) {                                        \
      const unsigned long long now_ = __builtin_readcyclecounter(); \
      atomicAdd(&P.dbg[k_], now_ - t_last);                         \
      t_last = now_;                                                \
    }                                                               \
  } while (0)
  float qx, qy, qz;
  double sx, sy, sz;  // the sample the hand frame keeps (frame_estimator.cpp:20-22, 53-54)
  if (P.sample_xyz) {
    sx = P.sample_xyz[3 * (size_t)s + 0];
    sy = P.sample_xyz[3 * (size_t)s + 1];
    sz = P.sample_xyz[3 * (size_t)s + 2];
    qx = (float)sx;
    qy = (float)sy;
    qz = (float)sz;
  } else {
    const int sidx = P.sample_idx[s];
    qx = P.px[sidx];
    qy = P.py[sidx];
    qz = P.pz[sidx];
    sx = (double)qx;
    sy = (double)qy;
    sz = (double)qz;
  }
  if (tid == 0) {
    s_count = 0;
    s_bounds[0] = 0;
    s_bounds[1] = 0;
    s_bounds[2] = 0;
    s_seen = 0;
    s_hl_ready = 0;
    s_hl_next = 0;
    s_hl_count = 0;
    s_ncrowd = 0;
    s_fkf = -1;
  }
  // bucket mode: two u32 arrays in the key storage, counters behind them
  uint32_t *s_a, *s_b;  // indices in visit order, then sorted d2 bits / indices in bucket order, then sorted
  int *s_hist;
  if constexpr (GLOBAL) {
    s_a = reinterpret_cast<uint32_t *>(P.nn + ((size_t)s * 6 + 5) * P.cap);
    s_b = reinterpret_cast<uint32_t *>(P.nn_idx + (size_t)s * P.cap);
    s_hist = reinterpret_cast<int *>(s_keys);
  } else {
    s_a = reinterpret_cast<uint32_t *>(s_keys);
    s_b = s_a + P.cap;
    s_hist = reinterpret_cast<int *>(s_keys + P.cap);
  }
  int *s_start = s_hist + NBK;  // NBK + 1 entries
  auto d2_of = [&](int i) {  // FLANN L2_Simple<float>, the same operation order as in the visit
    const float4 p = P.pxyz[i];
    float d = qx - p.x;
    float d2 = 0.f;
    d2 += d * d;
    d = qy - p.y;
    d2 += d * d;
    d = qz - p.z;
    d2 += d * d;
    return d2;
  };
  const float bscale = (float)NBK / P.r2_all;
  auto bucket_of = [&](float d2) {
    const int b = (int)(d2 * bscale);
    return b < NBK - 1 ? b : NBK - 1;
  };
  if (P.bucket)
    for (int i = tid; i < NBK; i += NB_THREADS) s_hist[i] = 0;
  __syncthreads();
  constexpr int VW = NB_WAVES - 1;  // visiting waves
  // frames (frame_estimator.cpp:66-86 + local_frame.cpp:14-41) from the kf nearest neighbours in FLANN order, idx_of(t) =
  // point index of the t-th; one whole wave.  M = sum n n^T and sum n are nine sequential fp64 chains in neighbour
  // order: the normals come in ONE round trip per 64 — lane t asks for the t-th neighbour's — and are handed to the
  // nine chain lanes through v_readlane (the same adds in the same order per chain); the wave's first lane runs the
  // eigensolver, stores the frame and, for the height list, publishes the hand axis of slot 0 and the limit.
  auto frame_of = [&](int kf, auto idx_of) {
    const int cp = lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 1 : lane == 3 ? 2 : lane == 4 ? 2 : lane == 5 ? 2 : lane == 6 ? 0 : lane == 7 ? 1 : 2;
    const int cq = lane == 0 ? 0 : lane == 1 ? 0 : lane == 2 ? 1 : lane == 3 ? 0 : lane == 4 ? 1 : lane == 5 ? 2 : -1;
    double acc = 0.0;
    for (int base = 0; base < kf; base += 64) {
      const int tl = base + lane;
      const float4 nl = P.pnrm[idx_of(tl < kf ? tl : base)];
      const int m = kf - base < 64 ? kf - base : 64;
      for (int u = 0; u < m; u++) {
        const float ux = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl.x), u));
        const float uy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl.y), u));
        const float uz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nl.z), u));
        const float xs = cp == 0 ? ux : (cp == 1 ? uy : uz);
        const float ys = cq == 0 ? ux : (cq == 1 ? uy : uz);
        const double x = (double)xs;
        const double y = cq >= 0 ? (double)ys : 1.0;
        if (lane < 9) acc += x * y;
      }
    }
    auto chain = [&](int c) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(acc);
      const unsigned lo = __shfl((unsigned)b, c), hi = __shfl((unsigned)(b >> 32), c);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    double m00 = chain(0), m10 = chain(1), m11 = chain(2), m20 = chain(3), m21 = chain(4), m22 = chain(5);
    double a0 = chain(6), a1 = chain(7), a2 = chain(8);
    double *f = P.frames + 12 * (size_t)s;
    if (lane == 0) {
      f[0] = sx;
      f[1] = sy;
      f[2] = sz;
      if (kf > 0) {
        double ev[3], Q[9];
        eigen3(m00, m10, m11, m20, m21, m22, ev, Q);
        int mn = 0, mx = 0;
        for (int i = 1; i < 3; i++) {
          if (ev[i] < ev[mn]) mn = i;
          if (ev[i] > ev[mx]) mx = i;
        }
        double curv[3], nor[3];
        for (int r = 0; r < 3; r++) {
          curv[r] = Q[3 * r + mn];
          nor[r] = Q[3 * r + mx];
        }
        const double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
        a0 /= nrm;
        a1 /= nrm;
        a2 /= nrm;
        const double dot = a0 * nor[0] + a1 * nor[1] + a2 * nor[2];
        if (dot < 0)
          for (int r = 0; r < 3; r++) nor[r] *= -1.0;
        f[3] = nor[0];
        f[4] = nor[1];
        f[5] = nor[2];
        f[6] = curv[1] * nor[2] - curv[2] * nor[1];
        f[7] = curv[2] * nor[0] - curv[0] * nor[2];
        f[8] = curv[0] * nor[1] - curv[1] * nor[0];
        f[9] = curv[0];
        f[10] = curv[1];
        f[11] = curv[2];
      }
    }
    // hl_note.  The height crop shared by the orientations of a sample: cropByHandHeight (point_list.cpp:35-55) keeps the
    // points whose coordinate along the hand frame's third axis lies in (-h, h); that axis is the one the orientations
    // rotate about, so the frames of a sample's slots have the same third column up to rounding.  The points with
    // |z| < h + margin for the axis of slot 0 — margin = (largest deviation of any slot's axis from it) x radius + 1e-12,
    // a superset of every orientation's exact crop, which hand_eval_kernel then decides with its own frame — are listed
    // in this kernel, by the gathering waves as they pass (coordinates in registers; order of the list: none, the entries
    // carry their rank).  This was a kernel of its own that read the gathered rows back from HBM (110 MB per 2564
    // samples), then a pass of its own behind the frame.
    if (P.hl && kf > 0) {
      auto bc = [&](double v) {  // lane 0's value to the wave
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      };
      // lane 0 holds the frame it has just stored: F = [normal | binormal | curvature] (hand_set.cpp:39-40)
      double F[9];
      __threadfence_block();
#pragma unroll
      for (int r = 0; r < 3; r++) {
        F[3 * r + 0] = bc(lane == 0 ? f[3 + r] : 0.0);
        F[3 * r + 1] = bc(lane == 0 ? f[6 + r] : 0.0);
        F[3 * r + 2] = bc(lane == 0 ? f[9 + r] : 0.0);
      }
      double axis[3] = {0.0, 0.0, 0.0}, dev = 0.0;
      for (int slot = 0; slot < P.hl_slots; slot++) {
        double a[3];
#pragma unroll
        for (int r = 0; r < 3; r++)
          a[r] = F[3 * r] * P.hl_col[slot][0] + F[3 * r + 1] * P.hl_col[slot][1] + F[3 * r + 2] * P.hl_col[slot][2];
        if (slot == 0) {
          axis[0] = a[0];
          axis[1] = a[1];
          axis[2] = a[2];
        }
        dev = fmax(dev, fabs(a[0] - axis[0]) + fabs(a[1] - axis[1]) + fabs(a[2] - axis[2]));
      }
      if (lane == 0) {
        s_hl_axis[0] = axis[0];
        s_hl_axis[1] = axis[1];
        s_hl_axis[2] = axis[2];
        s_hl_axis[3] = P.hl_height + dev * P.hl_radius + 1e-12;
      }
    }
  };
  auto early_frame = [&] {
    // the cells within the frame radius (same distance arithmetic as the visit), hits appended to s_fk in visit order
    const float rfr = sqrtf(P.r2_frames) * 1.001f + 1e-5f;
    int fn = 0;  // hits so far: the same in every lane (it only ever grows by a ballot's population)
    auto visit_f = [&](bool in, int i, float x, float y, float z) {
      float d = qx - x;
      float d2 = 0.f;
      d2 += d * d;
      d = qy - y;
      d2 += d * d;
      d = qz - z;
      d2 += d * d;
      const bool hit = in && d2 < P.r2_frames;
      const unsigned long long ballot = __ballot(hit);
      if (hit) {
        const int pos = fn + __popcll(ballot & ((1ull << lane) - 1ull));
        if (pos < FCAP) s_fk[pos] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)i;
      }
      fn += __popcll(ballot);
    };
    {
      const GridView &g = P.grid;
      const int x0 = grid_coord(g, 0, qx - rfr), x1 = grid_coord(g, 0, qx + rfr);
      const int y0 = grid_coord(g, 1, qy - rfr), y1 = grid_coord(g, 1, qy + rfr);
      const int z0 = grid_coord(g, 2, qz - rfr), z1 = grid_coord(g, 2, qz + rfr);
      const int ny = y1 - y0 + 1;
      const int ncol = (x1 - x0 + 1) * ny;
      if (ncol <= 64) {
        // the point ranges of all columns in one round trip (a lane per column), then four columns' points at a time
        int cbv = 0, cev = 0;
        if (lane < ncol) {
          const int cx = x0 + lane / ny, cy = y0 + lane % ny;
          const int cbase = (cx * g.dim[1] + cy) * g.dim[2];
          cbv = g.start[cbase + z0];
          cev = g.start[cbase + z1 + 1];
        }
        for (int c0 = 0; c0 < ncol; c0 += 4) {
          int cb[4], ce[4];
          float4 p[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int col = c0 + q < ncol ? c0 + q : c0;
            cb[q] = __builtin_amdgcn_readlane(cbv, col);
            ce[q] = c0 + q < ncol ? __builtin_amdgcn_readlane(cev, col) : cb[q];
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int t = cb[q] + lane;
            p[q] = g.p[t < ce[q] ? t : 0];
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (cb[q] < ce[q]) visit_f(cb[q] + lane < ce[q], __float_as_int(p[q].w), p[q].x, p[q].y, p[q].z);
#pragma unroll
          for (int q = 0; q < 4; q++)
            for (int t0 = cb[q] + 64; t0 < ce[q]; t0 += 64) {
              const int t = t0 + lane;
              const bool in = t < ce[q];
              const float4 pt = g.p[in ? t : cb[q]];
              visit_f(in, __float_as_int(pt.w), pt.x, pt.y, pt.z);
            }
        }
      } else {
        grid_visit(g, qx, qy, qz, rfr, 0, 1, lane, visit_f);
      }
    }
    const int kf = fn;
    if (kf > FCAP) return;  // left to the sorted list (the late path below)
    __threadfence_block();  // the keys were written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    // FLANN's order, (d2, index) ascending: the rank of a key is the number of smaller keys (they are distinct)
    constexpr int KPL = (FCAP + 63) / 64;
    unsigned long long mine[KPL];
    int rank[KPL];
#pragma unroll
    for (int q = 0; q < KPL; q++) {
      mine[q] = lane + 64 * q < kf ? s_fk[lane + 64 * q] : ~0ull;
      rank[q] = 0;
    }
    for (int u = 0; u < kf; u++) {
      const unsigned long long ku = s_fk[u];
#pragma unroll
      for (int q = 0; q < KPL; q++) rank[q] += ku < mine[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < KPL; q++)
      if (lane + 64 * q < kf) s_fidx[rank[q]] = (int)(unsigned)mine[q];
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    frame_of(kf, [&](int t) { return s_fidx[t]; });
    if (lane == 0) s_fkf = kf;
  };
  // 1. visit the grid cells around the sample; FLANN L2_Simple<float>: d2 accumulated over x,y,z, strict <.
  //    The point ranges of the (x, y) cell columns are fetched up front, one column per lane, into LDS (a column holds
  //    ~45 points: one dependent global round trip less per column and wave).
  auto visit = [&](bool in, int i, float x, float y, float z) {
    float d = qx - x;
    float d2 = 0.f;
    d2 += d * d;
    d = qy - y;
    d2 += d * d;
    d = qz - z;
    d2 += d * d;
    const bool hit = in && d2 < P.r2_all;
    const unsigned long long ballot = __ballot(hit);
    if (ballot) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(ballot));
      base = __builtin_amdgcn_readfirstlane(base);  // every lane is active here and lane 0 holds it
      if (hit) {
        const int pos = base + __popcll(ballot & ((1ull << lane) - 1ull));
        if (pos < P.cap) {
          if (P.bucket) {
            if constexpr (GLOBAL) {
              s_a[pos] = (uint32_t)i;  // d2 is recomputed by the scatter (a thread's entries do not fit its registers)
            } else {
              s_a[pos] = __float_as_uint(d2);  // visit order; the scatter below re-orders both arrays in place
              s_b[pos] = (uint32_t)i;
            }
            atomicAdd(&s_hist[bucket_of(d2)], 1);
          } else {
            s_keys[pos] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)i;
          }
        }
      }
    }
  };
  {
    const GridView &g = P.grid;
    const int x0 = grid_coord(g, 0, qx - P.reach), x1 = grid_coord(g, 0, qx + P.reach);
    const int y0 = grid_coord(g, 1, qy - P.reach), y1 = grid_coord(g, 1, qy + P.reach);
    const int z0 = grid_coord(g, 2, qz - P.reach), z1 = grid_coord(g, 2, qz + P.reach);
    const int ny = y1 - y0 + 1;
    const int ncol = (x1 - x0 + 1) * ny;
    __shared__ int s_cb[NB_THREADS], s_ce[NB_THREADS];
    if (ncol <= NB_THREADS) {
      if (tid < ncol) {
        const int cx = x0 + tid / ny, cy = y0 + tid % ny;
        const int cbase = (cx * g.dim[1] + cy) * g.dim[2];
        s_cb[tid] = g.start[cbase + z0];
        s_ce[tid] = g.start[cbase + z1 + 1];
      }
      __syncthreads();
      // four columns of the wave at a time: the first 64 points of each are requested together, then visited (a column per
      // iteration was one dependent global round trip per column, 18 in a row per wave: 30 of the kernel's ~110 kcycles);
      // the few columns with more than 64 points finish in the tail loop.  The visit order is free: the list is sorted.
      constexpr int CU_ = 4;
      if ((tid >> 6) < VW)
      for (int c0 = tid >> 6; c0 < ncol; c0 += CU_ * VW) {
        int cb[CU_], ce[CU_];
        float4 p[CU_];
#pragma unroll
        for (int q = 0; q < CU_; q++) {
          const int col = c0 + q * VW;
          cb[q] = col < ncol ? s_cb[col] : 0;
          ce[q] = col < ncol ? s_ce[col] : 0;
        }
#pragma unroll
        for (int q = 0; q < CU_; q++) {
          const int t = cb[q] + lane;
          p[q] = g.p[t < ce[q] ? t : 0];
        }
#pragma unroll
        for (int q = 0; q < CU_; q++)
          if (cb[q] < ce[q]) visit(cb[q] + lane < ce[q], __float_as_int(p[q].w), p[q].x, p[q].y, p[q].z);
#pragma unroll
        for (int q = 0; q < CU_; q++)
          for (int t0 = cb[q] + 64; t0 < ce[q]; t0 += 64) {
            const int t = t0 + lane;
            const bool in = t < ce[q];
            const float4 pt = g.p[in ? t : cb[q]];
            visit(in, __float_as_int(pt.w), pt.x, pt.y, pt.z);
          }
      }
    } else if ((tid >> 6) < VW) {
      grid_visit(g, qx, qy, qz, P.reach, tid >> 6, VW, lane, visit);
    }
  }
  // The last wave does not visit: it finds the frame neighbourhood on its own (the few cells within the frame radius,
  // ~40 points), orders it and computes the local frame while the other seven walk the 144 cell columns — the 3 x 3
  // eigensolver is a chain of ~2-3 k dependent f64 instructions on one lane, and started after the sort it kept the
  // gathering waves waiting at the end of the kernel (GPD_NB_TIMING: 7 of 108 kcycles) and the height list behind it.
  if ((tid >> 6) == VW) early_frame();
  hand_over();
  NTICK(0);
  const int found = s_count;
  const int n = found < P.cap ? found : P.cap;
  if (P.bucket) {
    // 2. bucket sort: exclusive scan of the counters (PER buckets per lane) ...
    __shared__ int s_wsum[NB_WAVES];
    int c4[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      c4[q] = s_hist[PER * tid + q];
      sum += c4[q];
    }
    int incl = sum;  // wave-wide inclusive prefix sum through DPP (row_shr 1 / 2 / 4 / 8, then the row totals by row_bcast)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
    if (lane == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (tid >> 6); w++) run += s_wsum[w];
#pragma unroll
    for (int q = 0; q < PER; q++) {
      s_start[PER * tid + q] = run;
      s_hist[PER * tid + q] = run;  // the scatter cursor
      run += c4[q];
    }
    if (tid == NB_THREADS - 1) s_start[NBK] = run;
    __syncthreads();
    NTICK(1);
    // ... entries grouped bucket by bucket ...
    if constexpr (GLOBAL) {
      for (int t = tid; t < n; t += NB_THREADS) {
        const uint32_t i = s_a[t];
        const int pos = atomicAdd(&s_hist[bucket_of(d2_of((int)i))], 1);
        if (pos < n) s_b[pos] = i;  // always true (the counters come from the same d2 values); keeps a corrupted table out of memory
      }
    } else {
      // in place: a thread takes its (at most 16) entries into registers, everybody waits, then they go to their bucket
      // positions in the same two arrays — no second look at the coordinates
      constexpr int EPT = 8192 / NB_THREADS;
      uint32_t rd[EPT], ri[EPT];
#pragma unroll
      for (int q = 0; q < EPT; q++) {
        const int t = tid + q * NB_THREADS;
        rd[q] = t < n ? s_a[t] : 0u;
        ri[q] = t < n ? s_b[t] : 0u;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; q++) {
        const int t = tid + q * NB_THREADS;
        if (t < n) {
          const int pos = atomicAdd(&s_hist[bucket_of(__uint_as_float(rd[q]))], 1);
          if (pos < n) {
            s_a[pos] = rd[q];
            s_b[pos] = ri[q];
          }
        }
      }
    }
    hand_over();
    NTICK(2);
    // ... and every bucket ordered by (d2 bits, index); non-negative floats order as unsigned.  The d2 bits are
    // in s_a already (LDS mode) or recomputed into it (GLOBAL).
    auto d2bits_of_entry = [&](int x) {
      if constexpr (GLOBAL) return __float_as_uint(d2_of((int)s_b[x]));
      else return s_a[x];
    };
    for (int b = tid; b < NBK; b += NB_THREADS) {
      const int st = s_start[b], nb = s_start[b + 1] - st;
      if (nb <= 0) continue;
      auto run = [&](auto tag) {
        constexpr int N = decltype(tag)::value;
        unsigned long long k[N];
#pragma unroll
        for (int q = 0; q < N; q++) {
          const int x = st + (q < nb ? q : 0);
          k[q] = q < nb ? ((unsigned long long)d2bits_of_entry(x) << 32) | s_b[x] : 0x7fefffffffffffffull;  // padding: the largest finite double
        }
        sort_regs64<N>(k);
#pragma unroll
        for (int q = 0; q < N; q++)
          if (q < nb) {
            s_a[st + q] = (uint32_t)(k[q] >> 32);
            s_b[st + q] = (uint32_t)k[q];
          }
      };
      if (nb <= 4)
        run(std::integral_constant<int, 4>());
      else if (nb <= 8)
        run(std::integral_constant<int, 8>());
      else if (nb <= 16)
        run(std::integral_constant<int, 16>());
      else {
        // crowded bucket (lattice clouds put dozens of points at exactly the same distance): left
        // to a whole wave below
        const int c = atomicAdd(&s_ncrowd, 1);
        if (c < NBK) s_hist[c] = b;  // the scatter cursors are dead: their storage lists the crowded buckets
      }
    }
    __syncthreads();
    NTICK(3);
    const int ncrowd = s_ncrowd < NBK ? s_ncrowd : NBK;
    for (int c = tid >> 6; c < ncrowd; c += NB_WAVES) {
      const int b = s_hist[c];
      const int st = s_start[b], nb = s_start[b + 1] - st;
      if (nb <= 64) {
        // one key per lane, bitonic network across the wave
        unsigned long long k = lane < nb ? ((unsigned long long)d2bits_of_entry(st + lane) << 32) | s_b[st + lane] : ~0ull;
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
          for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)k, stride), hi = __shfl_xor((unsigned)(k >> 32), stride);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool keep_min = ((lane & stride) == 0) == ((lane & size) == 0);
            k = keep_min ? (k < other ? k : other) : (k > other ? k : other);
          }
        }
        if (lane < nb) {
          s_a[st + lane] = (uint32_t)(k >> 32);
          s_b[st + lane] = (uint32_t)k;
        }
      } else {
        // more than 64 keys in one bucket (dense clouds; lattice clouds put many points at exactly the same distance):
        // the d2 column first, then a bitonic network over the two arrays by the whole wave.  Every comparison
        // sorts upwards (the first step of each merge pairs x with its mirror image), so positions past the end
        // would hold +inf and never move: those comparisons are simply left out.
        if constexpr (GLOBAL)
          for (int x = lane; x < nb; x += 64) s_a[st + x] = __float_as_uint(d2_of((int)s_b[st + x]));
        int m = 1;
        while (m < nb) m <<= 1;
        auto exchange = [&](int lo, int hi) {
          if (hi < nb) {
            const unsigned long long klo = ((unsigned long long)s_a[st + lo] << 32) | s_b[st + lo];
            const unsigned long long khi = ((unsigned long long)s_a[st + hi] << 32) | s_b[st + hi];
            if (klo > khi) {
              s_a[st + lo] = (uint32_t)(khi >> 32);
              s_b[st + lo] = (uint32_t)khi;
              s_a[st + hi] = (uint32_t)(klo >> 32);
              s_b[st + hi] = (uint32_t)klo;
            }
          }
        };
        auto wave_sync = [] {
          __threadfence_block();
          __builtin_amdgcn_wave_barrier();
        };
        for (int size = 2; size <= m; size <<= 1) {
          wave_sync();
          const int half = size >> 1;
          for (int t = lane; t < (m >> 1); t += 64) {
            const int blk = t / half, off = t - blk * half;
            exchange(blk * size + off, blk * size + size - 1 - off);
          }
          for (int stride = size >> 2; stride > 0; stride >>= 1) {
            wave_sync();
            for (int t = lane; t < (m >> 1); t += 64) {
              const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
              exchange(lo, lo + stride);
            }
          }
        }
      }
    }
    hand_over();
    NTICK(4);
  } else {
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = n + tid; i < m; i += NB_THREADS) s_keys[i] = ~0ull;
    __syncthreads();
    // 2. bitonic sort ascending by (d2 bits, index); non-negative floats order as unsigned
    for (int k = 2; k <= m; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (m >> 1); t += NB_THREADS) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const unsigned long long a = s_keys[lo], b = s_keys[hi];
          if ((a > b) == up) {
            s_keys[lo] = b;
            s_keys[hi] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  // sorted position t -> d2 bits / point index
  auto d2bits_at = [&](int t) { return P.bucket ? s_a[t] : (unsigned)(s_keys[t] >> 32); };
  auto index_at = [&](int t) { return P.bucket ? (int)s_b[t] : (int)(unsigned)(s_keys[t] & 0xffffffffull); };
  // 3. prefix lengths of the image, frame and hand-search neighbourhoods
  const unsigned ri = __float_as_uint(P.r2_images), rf = __float_as_uint(P.r2_frames), rh = __float_as_uint(P.r2_hands);
  for (int t = tid; t < n; t += NB_THREADS) {
    const unsigned d = d2bits_at(t);
    const unsigned dn = (t + 1 < n) ? d2bits_at(t + 1) : 0xffffffffu;
    if (d < ri && dn >= ri) s_bounds[0] = t + 1;
    if (d < rf && dn >= rf) s_bounds[1] = t + 1;
    if (d < rh && dn >= rh) s_bounds[2] = t + 1;
  }
  __syncthreads();
  NTICK(5);
  // 4. sorted index list + gathered SoA neighbourhood, and the height list of the hand search on the way
  // (the sorted index list itself has no reader on the device: nn_idx is the GLOBAL mode's scratch only — 43 MB of the
  //  303 MB this phase used to write per 2564 samples)
  float *on = P.nn + (size_t)s * 6 * P.cap;
  const int n_img = s_bounds[0], kf = s_bounds[1], Nh = s_bounds[2];
  // the frame wave's own neighbourhood must be the sorted list's prefix; when it is not there (more than FCAP points
  // inside the frame radius) or differs, the frame is computed from the sorted list after the gather, as it used to be
  const bool late = s_fkf != kf;
  if (tid == 0) {
    P.counts[8 * s + 0] = Nh;
    P.counts[8 * s + 1] = n_img;
    P.counts[8 * s + 2] = kf;
    P.counts[8 * s + 3] = found;
  }
  const bool list_here = P.hl && !late && kf > 0 && Nh > 0;
  const double ax0 = s_hl_axis[0], ax1 = s_hl_axis[1], ax2 = s_hl_axis[2], lim = s_hl_axis[3];  // (read by list_here only)
  float4 *hl_out = P.hl ? P.hl + (size_t)s * P.cap : nullptr;
  int seen = 0;
  // Two entries per thread and round, and the loads of round r + 1 are requested BEFORE the stores of round r go out
  // (two register sets, A and B): gfx9 counts loads and stores in one in-order counter, so a round that stores first and
  // then asks for the next points waits for its 14 store acknowledgements before it sees them.  The first camera's
  // flag comes with the coordinates (pxyz.w).
  constexpr int GT = NB_THREADS;  // every wave gathers
  // Centre of the image neighbourhood (HandSet::calculateShadow, hand_set.cpp:131-133: points.rowwise().sum() / size; the oracle
  // defines the sum sequentially in neighbour order, Eigen reduces in packets).  The addends are floats: when every addend is
  // a multiple of 2^L and n x max|x| < 2^(L + 53), EVERY partial sum of EVERY summation order is exact in fp64 — the
  // sequential chain, Eigen's, and the order-free one taken here on the way (each thread its entries, then the waves, then
  // the workgroup).  The certificate is two exponents per coordinate (see centre_exact); a sample that fails it — a point within
  // micrometres of a coordinate plane among points decimetres away — is flagged and gets its serial chain from centre_kernel.
  // (That kernel used to walk all 3 S chains from the gathered rows: 93 MB read back, 0.12-0.13 ms on a side stream.)
  double csum[3] = {0.0, 0.0, 0.0};
  int cemin[3] = {255, 255, 255}, cemax[3] = {0, 0, 0};
  auto centre_add = [&](const float4 &a) {
    const float v[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      csum[c] += (double)v[c];
      const int e = (int)((__float_as_uint(v[c]) >> 23) & 0xffu);
      const bool nz = (__float_as_uint(v[c]) & 0x7fffffffu) != 0u;
      const int el = e ? e : 1;  // a denormal's ulp is that of exponent 1
      cemin[c] = nz && el < cemin[c] ? el : cemin[c];
      cemax[c] = nz && el > cemax[c] ? el : cemax[c];
    }
  };
  struct Rnd {
    int t0, i0, i1, c0, c1;
    float4 a0, b0, a1, b1;
  };
  auto fetch = [&](int t0, Rnd &r) {
    r.t0 = t0;
    const int t1 = t0 + GT;
    r.i0 = index_at(t0 < n ? t0 : 0);
    r.i1 = index_at(t1 < n ? t1 : 0);
    r.a0 = P.pxyz[r.i0];
    r.b0 = P.pnrm[r.i0];
    r.a1 = P.pxyz[r.i1];
    r.b1 = P.pnrm[r.i1];
    r.c0 = __float_as_int(r.a0.w);  // the first camera's flag (split_soa_kernel)
    r.c1 = __float_as_int(r.a1.w);
  };
  auto put = [&](const Rnd &r) {  // (wave-uniform control flow around the ballots)
    const int t0 = r.t0, t1 = r.t0 + GT;
    const bool one = t0 < n, two = t1 < n;
    if (one) {
      on[0 * P.cap + t0] = r.a0.x;
      on[1 * P.cap + t0] = r.a0.y;
      on[2 * P.cap + t0] = r.a0.z;
      on[3 * P.cap + t0] = r.b0.x;
      on[4 * P.cap + t0] = r.b0.y;
      on[5 * P.cap + t0] = r.b0.z;
      if (t0 < n_img) {
        seen |= (int)(r.c0 != 0);
        centre_add(r.a0);
      }
    }
    if (two) {
      on[0 * P.cap + t1] = r.a1.x;
      on[1 * P.cap + t1] = r.a1.y;
      on[2 * P.cap + t1] = r.a1.z;
      on[3 * P.cap + t1] = r.b1.x;
      on[4 * P.cap + t1] = r.b1.y;
      on[5 * P.cap + t1] = r.b1.z;
      if (t1 < n_img) {
        seen |= (int)(r.c1 != 0);
        centre_add(r.a1);
      }
    }
    for (int cam = 1; cam < P.num_cams; cam++) {  // further cameras: not pipelined
      if (one && t0 < n_img) seen |= (int)((unsigned)(P.cam_source[(size_t)cam * P.num_points + r.i0] != 0) << cam);
      if (two && t1 < n_img) seen |= (int)((unsigned)(P.cam_source[(size_t)cam * P.num_points + r.i1] != 0) << cam);
    }
    if (list_here && (t0 & ~63) < Nh) {  // uniform: some entry of the wave's first row is a hand-search neighbour
      bool ina = false, inb = false;
      if (t0 < Nh) {
        const double za = ax0 * ((double)r.a0.x - sx) + ax1 * ((double)r.a0.y - sy) + ax2 * ((double)r.a0.z - sz);
        ina = za > -lim && za < lim;
      }
      if (t1 < Nh) {
        const double zb = ax0 * ((double)r.a1.x - sx) + ax1 * ((double)r.a1.y - sy) + ax2 * ((double)r.a1.z - sz);
        inb = zb > -lim && zb < lim;
      }
      const unsigned long long ba = __ballot(ina), bb = __ballot(inb);
      if (ba | bb) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_hl_count, __popcll(ba) + __popcll(bb));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ina) hl_out[base + __popcll(ba & below)] = make_float4(r.a0.x, r.a0.y, r.a0.z, __int_as_float(t0));
        if (inb) hl_out[base + __popcll(ba) + __popcll(bb & below)] = make_float4(r.a1.x, r.a1.y, r.a1.z, __int_as_float(t1));
      }
    }
  };
  if (n > 0) {
    Rnd A, B;
    fetch(tid, A);
    for (int t0 = tid; (t0 & ~63) < n; t0 += 4 * GT) {  // whole waves stay in the loop (the ballots in put)
      fetch(t0 + 2 * GT, B);
      __builtin_amdgcn_sched_barrier(0);
      put(A);
      __builtin_amdgcn_sched_barrier(0);
      fetch(t0 + 4 * GT, A);
      __builtin_amdgcn_sched_barrier(0);
      put(B);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (seen) atomicOr(&s_seen, seen);
  {  // the wave's share of the centre sums (exact under the certificate, discarded otherwise)
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        csum[c] += __shfl_xor(csum[c], o);
        cemin[c] = min(cemin[c], __shfl_xor(cemin[c], o));
        cemax[c] = max(cemax[c], __shfl_xor(cemax[c], o));
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s_csum[tid >> 6][c] = csum[c];
        s_cexp[tid >> 6][c] = cemin[c];
        s_cexp[tid >> 6][3 + c] = cemax[c];
      }
    }
  }
  NTICK(6);
  if (late) {
    // 5'. the frame from the sorted list (more than FCAP points inside the frame radius), then the height list as a pass
    //     of its own: every wave takes chunks of 128 hand-search neighbours until none is left
    if ((tid >> 6) == VW) {
      frame_of(kf, index_at);
      __threadfence_block();
      if (lane == 0) *(volatile int *)&s_hl_ready = 1;
    }
    if (P.hl) {
      while (*(volatile int *)&s_hl_ready == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      NTICK(8);
      const double lx0 = s_hl_axis[0], lx1 = s_hl_axis[1], lx2 = s_hl_axis[2], llim = s_hl_axis[3];
      while (kf > 0 && Nh > 0) {
        int chunk = 0;
        if (lane == 0) chunk = atomicAdd(&s_hl_next, 1);
        const int e0 = __builtin_amdgcn_readfirstlane(chunk) * 128;
        if (e0 >= Nh) break;
        const int ea = e0 + lane, eb = e0 + 64 + lane;  // two entries per lane: their loads are in flight together
        const float4 pa = P.pxyz[index_at(ea < Nh ? ea : Nh - 1)], pb = P.pxyz[index_at(eb < Nh ? eb : Nh - 1)];
        const double za = lx0 * ((double)pa.x - sx) + lx1 * ((double)pa.y - sy) + lx2 * ((double)pa.z - sz);
        const double zb = lx0 * ((double)pb.x - sx) + lx1 * ((double)pb.y - sy) + lx2 * ((double)pb.z - sz);
        const bool ina = ea < Nh && za > -llim && za < llim, inb = eb < Nh && zb > -llim && zb < llim;
        const unsigned long long ba = __ballot(ina), bb = __ballot(inb);
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_hl_count, __popcll(ba) + __popcll(bb));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ina) hl_out[base + __popcll(ba & below)] = make_float4(pa.x, pa.y, pa.z, __int_as_float(ea));
        if (inb) hl_out[base + __popcll(ba) + __popcll(bb & below)] = make_float4(pb.x, pb.y, pb.z, __int_as_float(eb));
      }
      NTICK(9);
    }
  }
  __syncthreads();
  NTICK(7);
  if (tid == 0) {
    P.counts[8 * s + 4] = s_seen;
    if (P.hl) P.counts[8 * s + 6] = s_hl_count;
  }
  if (tid < 3) {
    double acc = 0.0;
    int emin = 255, emax = 0;
    for (int w = 0; w < NB_WAVES; w++) {
      acc += s_csum[w][tid];
      emin = min(emin, s_cexp[w][tid]);
      emax = max(emax, s_cexp[w][tid + 3]);
    }
    const bool exact = centre_exact(emin, emax, n_img);
    if (exact) P.centers[3 * (size_t)s + tid] = n_img > 0 ? acc / (double)n_img : 0.0;
    // counts[8 s + 5]: bit c set -> coordinate c of the centre still needs its serial chain (centre_kernel)
    const unsigned long long todo = __ballot(!exact);
    if (tid == 0) P.counts[8 * s + 5] = (int)(todo & 7ull);
  }
#undef NTICK
}

// centre of the image neighbourhood (HandSet::calculateShadow, hand_set.cpp:131-133): sequential
// fp64 sums in neighbour order.  The chain is serial, so one LANE per (sample, coordinate) walks it
// and the whole batch runs side by side (inside neighbourhood_kernel the three chains of a sample
// kept a 256-thread workgroup waiting: 390 of its 1320 us).  Since round 6 only for the (sample, coordinate) pairs whose
// sum neighbourhood_kernel could not certify as exact in every order (counts[8 s + 5]): on the benchmark clouds none.
__global__ __launch_bounds__(64) void centre_kernel(const float *__restrict__ nn, const int32_t *__restrict__ counts, int cap,
                                                    int num_samples, double *__restrict__ centers) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= 3 * num_samples) return;
  const int s = g / 3, c = g - 3 * s;
  if (!((counts[8 * s + 5] >> c) & 1)) return;  // neighbourhood_kernel's order-free sum was exact: the centre is written
  const int n_img = counts[8 * s + 1];
  const float *row = nn + ((size_t)s * 6 + c) * cap;
  double acc = 0.0;
  int t = 0;
  for (; t + 16 <= n_img; t += 16) {  // four 16-byte loads in flight, adds in order
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const float4 *>(row + t + 4 * i);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc += (double)v[i].x;
      acc += (double)v[i].y;
      acc += (double)v[i].z;
      acc += (double)v[i].w;
    }
  }
  for (; t < n_img; t++) acc += (double)row[t];
  centers[g] = n_img > 0 ? acc / (double)n_img : 0.0;
}

// ---------------------------------------------------------------------------
// Cloud::calculateNormals (util/cloud.cpp:451-476): the radius-search normal estimation of calculateNormalsOMP
// (:497-535, pcl::NormalEstimationOMP) followed by reverseNormals (:573-604).  Per point: the neighbours within the
// radius in FLANN order (d2, index), centroid and covariance as sequential fp64 sums IN THAT ORDER (the oracle's
// definition: nine serial chains as long as the neighbourhood), Eigen's 3x3 eigensolver, eigenvector of the smallest
// eigenvalue, flipped towards the view point of the first camera that sees the point, then the reference's reversal rule.
//
// Rounds 1-3 gave every point a workgroup of its own that did all of this behind barriers: 1.40 ms per 30k points, 75 % of
// it three lanes walking the chains one LDS round trip per step, the eigensolver on one lane of 256.  Now every part runs
// where it is parallel:
//   normals_list_kernel    a WAVE per point (cell order: neighbouring waves share grid cells in L1): the hits' (d2 bits,
//                          index) keys go to the wave's LDS, then into registers — 4, 8 or 16 keys per lane — for a bitonic
//                          network that never touches memory (in-lane stages on registers, cross-lane stages through
//                          ds_bpermute); the sorted indices leave as 16-byte stores into the point's row.  No workgroup
//                          barrier anywhere.  A point with more than 1024 neighbours is queued
//   normals_list_big_kernel  the queued points (un-voxelised scans), a workgroup each: keys sorted in LDS (up to 8192) or
//                          in the point's own slice of an arena in global memory — ANY size: no capacity
//   normals_finish_kernel  a LANE per point: the nine chains of 64 points side by side in one wave (every lane useful, nine
//                          independent chains per lane against the add latency), coordinates gathered from the L2-resident
//                          point table with 32 rows in flight per lane, then 64 eigensolvers side by side, flips, store
// ---------------------------------------------------------------------------
constexpr int NL_CAP = 1024;       // neighbours a wave sorts in registers (voxelised clouds at radius 0.03: ~340, up to ~950)
constexpr int NL_WAVES = 4;        // points per workgroup of the wave-per-point kernel
constexpr int NL_BIG_LDS = 8192;   // keys the big kernel sorts in LDS (64 KB); longer lists are sorted in their arena slice
struct NormalsParams {
  GridView grid;
  const float4 *pxyz;              // [P] by original index
  int num_points;
  const int32_t *cam_source;
  int num_cams;
  double view_points[3 * kMaxCams];
  float r2;
  float reach;
  int32_t *count;                  // [P] by cell-order position
  int w0, w1;                      // the tile of this pass: cell-order positions [w0, w1), w0 a multiple of 64
  float4 *lists;                   // [(w1 - w0) / 64][NL_CAP][64] neighbour coordinates in FLANN order, transposed per block of 64 points
  int32_t *big;                    // [0] number of queued points, [1 ..] their cell-order positions
  long long *big_off;              // [P] slice of a queued point in the arena (8-byte units)
  unsigned long long *arena;       // sort space + sorted indices of the queued points
  unsigned long long *arena_top;   // bump allocator (8-byte units)
  long long arena_cap;
  int32_t *status;                 // bit 0: the arena is too small (the host grows it and runs again)
  float *out;                      // AoS [P][3] by original index
  double *cent;                    // [P][3] by cell-order position: the centroid of a point's list when its order-free sum is certified
                                   // exact (centre_exact), NaN where it is not (normals_finish_kernel then walks the chain)
};

// the hits of one query among the cells around it, a wave at work: hit(is a hit, index, d2) for every candidate, 64 at a time
template <typename Hit>
__device__ inline void normals_visit(const NormalsParams &P, float qx, float qy, float qz, int lane, Hit hit) {
  auto test = [&](bool in, int i, float x, float y, float z) {
    float d = qx - x;  // FLANN L2_Simple<float>: d2 accumulated over x, y, z, strict <
    float d2 = 0.f;
    d2 += d * d;
    d = qy - y;
    d2 += d * d;
    d = qz - z;
    d2 += d * d;
    hit(in && d2 < P.r2, i, d2);
  };
  const GridView &g = P.grid;
  const int x0 = grid_coord(g, 0, qx - P.reach), x1 = grid_coord(g, 0, qx + P.reach);
  const int y0 = grid_coord(g, 1, qy - P.reach), y1 = grid_coord(g, 1, qy + P.reach);
  const int z0 = grid_coord(g, 2, qz - P.reach), z1 = grid_coord(g, 2, qz + P.reach);
  const int ny = y1 - y0 + 1;
  const int ncol = (x1 - x0 + 1) * ny;
  if (ncol <= 64) {
    // the point ranges of all (x, y) cell columns in one round trip (a lane per column); then the first 64 points of
    // EIGHT columns are requested before any of them is tested — a column of a voxelised cloud holds ~45 points, so
    // a 4 x 4-column neighbourhood costs three dependent round trips instead of nine (the kernel waits for memory, not
    // for its ALUs: 30k waves of a few hundred instructions each)
    int cbv = 0, cev = 0;
    if (lane < ncol) {
      const int cx = x0 + lane / ny, cy = y0 + lane % ny;
      const int cbase = (cx * g.dim[1] + cy) * g.dim[2];
      cbv = g.start[cbase + z0];
      cev = g.start[cbase + z1 + 1];
    }
    for (int c0 = 0; c0 < ncol; c0 += 8) {
      int cb[8], cn[8];
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int c = c0 + u < ncol ? c0 + u : c0;
        cb[u] = __builtin_amdgcn_readlane(cbv, c);
        cn[u] = c0 + u < ncol ? __builtin_amdgcn_readlane(cev, c) - cb[u] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) p[u] = g.p[lane < cn[u] ? cb[u] + lane : 0];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (cn[u] > 0) test(lane < cn[u], __float_as_int(p[u].w), p[u].x, p[u].y, p[u].z);
#pragma unroll
      for (int u = 0; u < 8; u++)  // fuller columns (a wall seen edge-on, an un-voxelised scan): the rest, 64 at a time
        for (int t0 = 64; t0 < cn[u]; t0 += 64) {
          const bool in = t0 + lane < cn[u];
          const float4 pt = g.p[in ? cb[u] + t0 + lane : 0];
          test(in, __float_as_int(pt.w), pt.x, pt.y, pt.z);
        }
    }
  } else {
    grid_visit(g, qx, qy, qz, P.reach, 0, 1, lane, test);
  }
}

// Bitonic sort of 64 K keys held K per lane, element e = lane * K + r, ascending — in the formulation whose every
// compare-exchange puts the minimum at the lower index: a merge of width k starts with the "flip" step (partner e ^ (k - 1))
// and goes on with partners e ^ j, j = k / 4 .. 1.  Partners inside a lane exchange registers (v_min_f64 + v_max_f64: the keys
// are compared AS DOUBLES — sign bit clear, never a NaN, so the order is the order of the bit patterns — at the full f64 rate,
// where a 64-bit integer compare plus selects costs four times as much); partners in another lane come through ds_bpermute.
// (d2 bits, index) ascending = FLANN's result order; positive floats order as their bit patterns.
__device__ __forceinline__ double key_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));  // (no canonicalisation of the operands: they are never NaNs)
  return r;
}
__device__ __forceinline__ double key_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr unsigned long long NL_PAD_KEY = 0x7fefffffffffffffull;  // above every key, and a finite double
template <int K>
__device__ __forceinline__ void wave_sort_regs(double (&key)[K], int lane) {
  auto from_lane = [&](double v, int xor_mask) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __shfl_xor((unsigned)b, xor_mask), hi = __shfl_xor((unsigned)(b >> 32), xor_mask);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  };
#pragma unroll
  for (int k = 2; k <= 64 * K; k <<= 1) {
    // ---- the flip step: partner e ^ (k - 1)
    if (k <= K) {
#pragma unroll
      for (int r = 0; r < K; r++)
        if ((r & (k >> 1)) == 0) {  // the lower half of its block of k
          const int r2 = r ^ (k - 1);
          const double a = key[r], b = key[r2];
          key[r] = key_min(a, b);
          key[r2] = key_max(a, b);
        }
    } else {
      const int lm = k / K - 1;                       // lanes of a block of k elements: flip them all, and r within the lane
      const bool lower = (lane & (k / K / 2)) == 0;   // the lower half of the block holds the minima
      double other[K];
#pragma unroll
      for (int r = 0; r < K; r++) other[r] = from_lane(key[K - 1 - r], lm);
#pragma unroll
      for (int r = 0; r < K; r++) key[r] = ((other[r] < key[r]) == lower) ? other[r] : key[r];
    }
    // ---- partners e ^ j
#pragma unroll
    for (int j = k >> 2; j > 0; j >>= 1) {
      if (j >= K) {
        const int lj = j / K;
        const bool lower = (lane & lj) == 0;
#pragma unroll
        for (int r = 0; r < K; r++) {
          const double other = from_lane(key[r], lj);
          key[r] = ((other < key[r]) == lower) ? other : key[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < K; r++)
          if ((r & j) == 0) {
            const double a = key[r], b = key[r | j];
            key[r] = key_min(a, b);
            key[r | j] = key_max(a, b);
          }
      }
    }
  }
}

// sort the n keys of the wave's LDS row through registers, fetch the coordinates and store the list TRANSPOSED: entry e of
// the p-th point of a block of 64 points at [block][e][p], so that normals_finish_kernel — a lane per point — reads one
// contiguous kilobyte per step and wave
template <int K>
__device__ __forceinline__ void normals_sort_store(const NormalsParams &P, const unsigned long long *keys, int n, int lane, int w, const float4 &q) {
  double key[K];
#pragma unroll
  for (int r = 0; r < K; r++) key[r] = __longlong_as_double((long long)(lane * K + r < n ? keys[lane * K + r] : NL_PAD_KEY));
  wave_sort_regs<K>(key, lane);
  float4 *col = P.lists + (size_t)((w - P.w0) >> 6) * NL_CAP * 64 + (w & 63);
  // (round 6) the centroid on the way: the coordinates pass through this wave's registers anyway.  The sum of floats in fp64 is
  // the same in EVERY order when centre_exact's certificate holds (two exponents per coordinate), so the order-free sum taken
  // here — a lane its entries, then the wave — is the sequential one of the oracle; normals_finish_kernel then reads the lists
  // ONCE (covariance) instead of twice: it moved 447 MB per 30k points, 3.8 TB/s.  A list that fails the certificate (a neighbour
  // micrometres from a coordinate plane) gets NaN and its chain is walked there, as before.
  // The certificate's two exponents per coordinate: every neighbour lies within the search radius of the query, so when the
  // query is farther than that from the coordinate plane (wave-uniform: three neighbourhoods in four on the benchmark clouds)
  // |q| - r <= |x| <= |q| + r bounds them without looking at a single entry; only a neighbourhood that reaches a coordinate
  // plane tracks its smallest non-zero and its largest |x| entry by entry (as unsigned bit patterns: |x| orders like its bits,
  // and bits - 1 sends a zero to the far end of a minimum).
  const float qv[3] = {q.x, q.y, q.z};
  const float rad = sqrtf(P.r2) * 1.0001f;
  const bool track = !(fabsf(qv[0]) > 2.f * rad && fabsf(qv[1]) > 2.f * rad && fabsf(qv[2]) > 2.f * rad);
  double cs[3] = {0.0, 0.0, 0.0};
  unsigned umax[3] = {0u, 0u, 0u}, umin[3] = {~0u, ~0u, ~0u};
  auto take = [&](const float4 &v, bool on) {
    const float x[3] = {on ? v.x : 0.f, on ? v.y : 0.f, on ? v.z : 0.f};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      cs[a] += (double)x[a];
      if (track) {
        const unsigned b = __float_as_uint(x[a]) & 0x7fffffffu;
        umax[a] = max(umax[a], b);
        umin[a] = min(umin[a], b - 1u);
      }
    }
  };
#pragma unroll
  for (int r0 = 0; r0 < K; r0 += 4) {  // four gathers in flight, then their four stores
    const int e = lane * K + r0;
    const float4 v0 = P.pxyz[e + 0 < n ? (unsigned)__double_as_longlong(key[r0 + 0]) : 0u];
    const float4 v1 = P.pxyz[e + 1 < n ? (unsigned)__double_as_longlong(key[r0 + 1]) : 0u];
    const float4 v2 = P.pxyz[e + 2 < n ? (unsigned)__double_as_longlong(key[r0 + 2]) : 0u];
    const float4 v3 = P.pxyz[e + 3 < n ? (unsigned)__double_as_longlong(key[r0 + 3]) : 0u];
    if (e + 0 < n) col[(size_t)(e + 0) * 64] = v0;
    if (e + 1 < n) col[(size_t)(e + 1) * 64] = v1;
    if (e + 2 < n) col[(size_t)(e + 2) * 64] = v2;
    if (e + 3 < n) col[(size_t)(e + 3) * 64] = v3;
    take(v0, e + 0 < n);
    take(v1, e + 1 < n);
    take(v2, e + 2 < n);
    take(v3, e + 3 < n);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cs[a] += __shfl_xor(cs[a], o);
      if (track) {
        umax[a] = max(umax[a], (unsigned)__shfl_xor((int)umax[a], o));
        umin[a] = min(umin[a], (unsigned)__shfl_xor((int)umin[a], o));
      }
    }
  }
  if (lane < 3) {
    const double sum = lane == 0 ? cs[0] : (lane == 1 ? cs[1] : cs[2]);
    auto bexp = [](unsigned bits) {  // biased exponent of |x| given as bits, a denormal's as 1 (its ulp is that of exponent 1)
      const int e = (int)((bits >> 23) & 0xffu);
      return e ? e : 1;
    };
    int emin, emax;
    if (track) {
      const unsigned hi = lane == 0 ? umax[0] : (lane == 1 ? umax[1] : umax[2]), lo = lane == 0 ? umin[0] : (lane == 1 ? umin[1] : umin[2]);
      emax = hi ? bexp(hi) : 0;  // (no non-zero addend: emax = 0, the sum is exact)
      emin = hi ? bexp(lo + 1u) : 255;
    } else {
      const float aq = fabsf(lane == 0 ? qv[0] : (lane == 1 ? qv[1] : qv[2]));
      emax = bexp(__float_as_uint(aq + rad));       // |x| <= |q| + r: its exponent bounds every entry's from above ...
      emin = bexp(__float_as_uint(aq - rad)) - 1;   // ... |x| >= |q| - r > r from below (one binade of slack for the float subtraction)
    }
    P.cent[3 * (size_t)w + lane] = centre_exact(emin, emax, n) ? sum / (double)n : __longlong_as_double(0x7ff8000000000000ll);
  }
}

__global__ __launch_bounds__(64 * NL_WAVES) void normals_list_kernel(NormalsParams P) {
  __shared__ unsigned long long s_keys[NL_WAVES][NL_CAP];
  const int lane = threadIdx.x & 63;
  // the point of this wave, told to the compiler as what it is — wave-uniform: the cell ranges, column counts and loop
  // bounds derived from it then live in scalar registers and the visit's control flow is scalar branches (left as
  // per-lane values they became exec-mask regions with a vmcnt(0) at every join: the loads of a column batch were waited
  // for one by one)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = P.w0 + blockIdx.x * NL_WAVES + wv;
  if (w >= P.w1) return;
  unsigned long long *keys = s_keys[wv];
  const float4 q4 = P.grid.p[w];
  const float qx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q4.x))),
              qy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q4.y))),
              qz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q4.z)));
  int n = 0;  // wave-uniform: the hits so far
  normals_visit(P, qx, qy, qz, lane, [&](bool hit, int i, float d2) {
    const unsigned long long ballot = __ballot(hit);
    const int pos = n + __popcll(ballot & ((1ull << lane) - 1ull));
    if (hit && pos < NL_CAP) keys[pos] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)i;
    n += __popcll(ballot);
  });
  if (lane == 0) P.count[w] = n;
  if (n > NL_CAP) {  // a neighbourhood for the big kernel (it searches again)
    if (lane == 0) P.big[1 + atomicAdd(P.big, 1)] = w;
    return;
  }
  __threadfence_block();  // the keys were written by other lanes of this wave
  __builtin_amdgcn_wave_barrier();
  if (n <= 256)
    normals_sort_store<4>(P, keys, n, lane, w, q4);
  else if (n <= 512)
    normals_sort_store<8>(P, keys, n, lane, w, q4);
  else
    normals_sort_store<16>(P, keys, n, lane, w, q4);
}

// The queued points: a workgroup each (a persistent launch walks the queue, whose length stays on the device).  The list
// length is known from the first kernel; a slice of m = 2^ceil(log2 n) eight-byte units is drawn from the arena, the keys
// are sorted in LDS (m <= 8192) or in the slice itself, and the sorted indices are written over the slice's first 4 n
// bytes from the front: index t lands on key t / 2, which has been read by then.
__global__ __launch_bounds__(256) void normals_list_big_kernel(NormalsParams P) {
  __shared__ unsigned long long s_big[NL_BIG_LDS];
  __shared__ int s_count;
  __shared__ long long s_off;
  const int tid = threadIdx.x, lane = tid & 63;
  const int queued = P.big[0];
  for (int b = blockIdx.x; b < queued; b += gridDim.x) {
    const int w = P.big[1 + b];
    const int n = P.count[w];
    int m = 1;
    while (m < n) m <<= 1;
    __syncthreads();
    if (tid == 0) {
      s_count = 0;
      const long long off = (long long)atomicAdd(P.arena_top, (unsigned long long)m);
      s_off = off;
      P.big_off[w] = off;
      if (off + m > P.arena_cap) atomicOr(P.status, 1);
    }
    __syncthreads();
    const long long off = s_off;
    if (off + m > P.arena_cap) continue;  // the host grows the arena and runs again
    unsigned long long *slice = P.arena + off;
    unsigned long long *keys = m <= NL_BIG_LDS ? s_big : slice;
    const float4 q = P.grid.p[w];
    grid_visit(P.grid, q.x, q.y, q.z, P.reach, tid >> 6, 4, lane, [&](bool in, int i, float x, float y, float z) {
      float d = q.x - x;
      float d2 = 0.f;
      d2 += d * d;
      d = q.y - y;
      d2 += d * d;
      d = q.z - z;
      d2 += d * d;
      const bool hit = in && d2 < P.r2;
      const unsigned long long ballot = __ballot(hit);
      if (ballot) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_count, __popcll(ballot));
        base = __builtin_amdgcn_readfirstlane(base);
        if (hit) keys[base + __popcll(ballot & ((1ull << lane) - 1ull))] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)i;
      }
    });
    for (int t = n + tid; t < m; t += 256) keys[t] = ~0ull;
    __threadfence_block();
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (m >> 1); t += 256) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const unsigned long long ka = keys[lo], kb = keys[hi];
          if ((ka > kb) == up) {
            keys[lo] = kb;
            keys[hi] = ka;
          }
        }
        __threadfence_block();
        __syncthreads();
      }
    int32_t *row = reinterpret_cast<int32_t *>(slice);
    for (int t0 = 0; t0 < n; t0 += 256) {
      const int t = t0 + tid;
      const int idx = t < n ? (int)(unsigned)keys[t] : 0;
      __syncthreads();  // every key of this batch has been read
      if (t < n) row[t] = idx;
      __threadfence_block();
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(64) void normals_finish_kernel(NormalsParams P) {
  const int w = P.w0 + blockIdx.x * 64 + threadIdx.x;
  if (w >= P.w1) return;
  if (*P.status & 1) return;  // the arena was too small: this run is repeated
  const int n = P.count[w];
  const float4 q = P.grid.p[w];
  const int pi = __float_as_int(q.w);
  constexpr int B = 16;  // list entries per batch; the next batch is requested before this one is summed
  // centroid: three sequential sums in neighbour order; covariance entries 00, 10, 11, 20, 21, 22 about it: six more
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  double m00 = 0.0, m10 = 0.0, m11 = 0.0, m20 = 0.0, m21 = 0.0, m22 = 0.0;
  const float4 *col = P.lists + (size_t)((w - P.w0) >> 6) * NL_CAP * 64 + (w & 63);  // this point's column of its block's transposed lists
  int nmax = n;  // the longest list of this wave: the loop below is uniform, shorter lists are masked out arithmetically
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  if (nmax <= NL_CAP) {
    // The common case, straight-line: every lane's list is a column of the block's transposed array, so entry t of all 64
    // points is ONE contiguous kilobyte.  Two batches of 16 entries are in flight while one is summed; nothing in the loop
    // depends on a lane (addresses are clamped, a list that has ended adds +0.0 — exact: no sum here can be -0.0), so the
    // loads are counted, not waited for at branch joins.
    auto load = [&](int t0, float4 (&v)[B]) {
#pragma unroll
      for (int i = 0; i < B; i++) v[i] = col[(size_t)min(t0 + i, NL_CAP - 1) * 64];
    };
    // neighbouring points have neighbourhoods of similar size: up to the shortest list of the wave (rounded down to whole
    // rounds of three batches) no entry needs a mask — the masks are two v_cndmask per double, as many instructions as the
    // sums themselves, and this kernel is ONE wave per SIMD: its time is its instruction count
    int nmin = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmin = min(nmin, __shfl_xor(nmin, o));
    const int tfull = __builtin_amdgcn_readfirstlane(nmin) / (3 * B) * (3 * B);
    float4 ba[B], bb[B], bc[B];
    auto sum0 = [&](int t0, const float4 (&v)[B], bool masked) {
#pragma unroll
      for (int i = 0; i < B; i++) {
        const bool on = !masked || t0 + i < n;
        c0 += on ? (double)v[i].x : 0.0;
        c1 += on ? (double)v[i].y : 0.0;
        c2 += on ? (double)v[i].z : 0.0;
      }
    };
    auto sum1 = [&](int t0, const float4 (&v)[B], bool masked) {
#pragma unroll
      for (int i = 0; i < B; i++) {
        const bool on = !masked || t0 + i < n;
        const double d0 = on ? (double)v[i].x - c0 : 0.0, d1 = on ? (double)v[i].y - c1 : 0.0, d2 = on ? (double)v[i].z - c2 : 0.0;
        m00 += d0 * d0;
        m10 += d1 * d0;
        m11 += d1 * d1;
        m20 += d2 * d0;
        m21 += d2 * d1;
        m22 += d2 * d2;
      }
    };
#define NL_ROUND(SUM, MASKED)    \
  load(t0 + 2 * B, bc);          \
  SUM(t0, ba, MASKED);           \
  load(t0 + 3 * B, ba);          \
  SUM(t0 + B, bb, MASKED);       \
  load(t0 + 4 * B, bb);          \
  SUM(t0 + 2 * B, bc, MASKED);
    // (round 6) the centroid comes from normals_list_kernel, which summed it order-free under centre_exact's certificate; a lane
    // whose list failed it holds NaN, and only a wave with such a lane walks the three centroid chains (this pass read 223 of
    // the kernel's 447 MB per 30k points)
    c0 = P.cent[3 * (size_t)w + 0];
    c1 = P.cent[3 * (size_t)w + 1];
    c2 = P.cent[3 * (size_t)w + 2];
    int t0 = 0;
    if (__builtin_amdgcn_ballot_w64(c0 != c0 || c1 != c1 || c2 != c2) != 0ull) {
      const double k0 = c0, k1 = c1, k2 = c2;
      c0 = c1 = c2 = 0.0;
      load(0, ba);
      load(B, bb);
      for (; t0 < tfull; t0 += 3 * B) { NL_ROUND(sum0, false) }
      for (; t0 < nmax; t0 += 3 * B) { NL_ROUND(sum0, true) }
      c0 = k0 != k0 ? c0 / (double)n : k0;
      c1 = k1 != k1 ? c1 / (double)n : k1;
      c2 = k2 != k2 ? c2 / (double)n : k2;
    }
    load(0, ba);
    load(B, bb);
    for (t0 = 0; t0 < tfull; t0 += 3 * B) { NL_ROUND(sum1, false) }
    for (; t0 < nmax; t0 += 3 * B) { NL_ROUND(sum1, true) }
#undef NL_ROUND
  } else {
    // a wave with a queued point (more than NL_CAP neighbours): per-lane walks, each lane at its own pace
    const bool small = n <= NL_CAP;
    const int32_t *row = small ? nullptr : reinterpret_cast<const int32_t *>(P.arena + P.big_off[w]);
    auto at = [&](int t) { return small ? col[(size_t)t * 64] : P.pxyz[row[t]]; };
    for (int t = 0; t < n; t++) {
      const float4 v = at(t);
      c0 += (double)v.x;
      c1 += (double)v.y;
      c2 += (double)v.z;
    }
    c0 /= (double)n;
    c1 /= (double)n;
    c2 /= (double)n;
    for (int t = 0; t < n; t++) {
      const float4 v = at(t);
      const double d0 = (double)v.x - c0, d1 = (double)v.y - c1, d2 = (double)v.z - c2;
      m00 += d0 * d0;
      m10 += d1 * d0;
      m11 += d1 * d1;
      m20 += d2 * d0;
      m21 += d2 * d1;
      m22 += d2 * d2;
    }
  }
  double ev[3], Q[9];
  eigen3(m00, m10, m11, m20, m21, m22, ev, Q);
  int mn = 0;
  for (int k = 1; k < 3; k++)
    if (ev[k] < ev[mn]) mn = k;
  const double nx = mn == 0 ? Q[0] : (mn == 1 ? Q[1] : Q[2]), ny = mn == 0 ? Q[3] : (mn == 1 ? Q[4] : Q[5]),
               nz = mn == 0 ? Q[6] : (mn == 1 ? Q[7] : Q[8]);
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  // estimated once, with the view point of the FIRST camera that sees the point (convertCameraSourceMatrixToLists,
  // cloud.cpp:606-620: `== 1` and a break); NormalEstimation::setViewPoint takes floats (cloud.cpp:513)
  for (int cam = 0; cam < P.num_cams; cam++) {
    if (P.cam_source[(size_t)cam * P.num_points + pi] != 1) continue;
    const double *vp = P.view_points + 3 * cam;
    const double dot = ((double)(float)vp[0] - (double)q.x) * nx + ((double)(float)vp[1] - (double)q.y) * ny + ((double)(float)vp[2] - (double)q.z) * nz;
    o0 = (float)(dot < 0 ? -nx : nx);
    o1 = (float)(dot < 0 ? -ny : ny);
    o2 = (float)(dot < 0 ? -nz : nz);
    break;
  }
  bool needs_reverse = true;  // reverseNormals (cloud.cpp:573-604)
  for (int cam = 0; cam < P.num_cams && needs_reverse; cam++) {
    if (P.cam_source[(size_t)cam * P.num_points + pi] != 1) continue;
    const double *vp = P.view_points + 3 * cam;
    const double d = (double)o0 * ((double)q.x - vp[0]) + (double)o1 * ((double)q.y - vp[1]) + (double)o2 * ((double)q.z - vp[2]);
    if (d < 0) needs_reverse = false;
  }
  if (needs_reverse) {
    o0 = (float)((double)o0 * -1.0);
    o1 = (float)((double)o1 * -1.0);
    o2 = (float)((double)o2 * -1.0);
  }
  P.out[3 * (size_t)pi + 0] = o0;
  P.out[3 * (size_t)pi + 1] = o1;
  P.out[3 * (size_t)pi + 2] = o2;
}

void normals_free(NormalsScratch &s) {
  void *ptrs[] = {s.d_count, s.d_lists, s.d_big, s.d_big_off, s.d_arena, s.d_ctl, s.d_out, s.d_cent};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  s = NormalsScratch();
}

// Host wrapper: normals of the uploaded cloud, result to the host and into the context's device copy.  Nothing waits
// for the device between the three kernels; the arena of the large neighbourhoods grows when a cloud needs more (the
// status word comes back with the result, and the run is repeated once with the size the first one asked for).
constexpr int kNormalsTile = 65536;  // points per pass of the normals kernels (a multiple of 64): their transposed lists are 1 GB

int normals_run(Cloud &c, double radius, float *normals_out, hipStream_t stream) {
  NormalsScratch &s = c.normals;
  const int P = c.num_points;
  if (P > s.cap_points) {
    note_alloc(__func__);
    unsigned long long *arena = s.d_arena;
    const long long arena_cap = s.arena_cap;
    s.d_arena = nullptr;
    normals_free(s);
    s.d_arena = arena;
    s.arena_cap = arena_cap;
    const int cap = P + P / 4;
    HIP_RET(hipMalloc(&s.d_count, (size_t)cap * sizeof(int32_t)));
    // the transposed lists take 16 KB per point (1024 entries of 16 bytes; a point touches 16 B per neighbour): a tile of at
    // most kNormalsTile points at a time shares ONE such array (1 GB), whatever the size of the cloud — sized per point of
    // the whole cloud it was 10 GB for a 500k-point raw scan
    const size_t tile_pts = std::min<size_t>(((size_t)cap + 63) / 64 * 64, kNormalsTile);
    HIP_RET(hipMalloc(&s.d_lists, tile_pts * NL_CAP * sizeof(float4)));
    HIP_RET(hipMalloc(&s.d_big, ((size_t)cap + 1) * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_big_off, (size_t)cap * sizeof(long long)));
    HIP_RET(hipMalloc(&s.d_ctl, 4 * sizeof(unsigned long long)));  // [0] arena top, [1] status
    HIP_RET(hipMalloc(&s.d_out, (size_t)cap * 3 * sizeof(float)));
    HIP_RET(hipMalloc(&s.d_cent, (size_t)cap * 3 * sizeof(double)));
    s.cap_points = cap;
  }
  if (!s.d_arena) {
    note_alloc(__func__);
    const long long want = 1ll << 20;  // 8 MB: a voxelised cloud needs none of it
    HIP_RET(hipMalloc(&s.d_arena, (size_t)want * sizeof(unsigned long long)));
    s.arena_cap = want;
  }
  NormalsParams np;
  np.grid = grid_view(c);
  np.pxyz = c.pxyz;
  np.num_points = P;
  np.cam_source = c.cam_source;
  np.num_cams = c.num_cams;
  std::memcpy(np.view_points, c.view_points, sizeof(np.view_points));
  np.r2 = (float)(radius * radius);
  np.reach = (float)radius * 1.001f + 1e-5f;
  np.count = s.d_count;
  np.lists = s.d_lists;
  np.big = s.d_big;
  np.big_off = s.d_big_off;
  np.arena_top = s.d_ctl;
  np.status = reinterpret_cast<int32_t *>(s.d_ctl + 1);
  np.out = s.d_out;
  np.cent = s.d_cent;
  struct {
    unsigned long long top, status;
  } h = {0, 0};
  int32_t queued = 0;
  const int tiles = (P + kNormalsTile - 1) / kNormalsTile;
  std::vector<int32_t> tile_queued((size_t)tiles, 0);
  for (int attempt = 0; attempt < 2; attempt++) {
    np.arena = s.d_arena;
    np.arena_cap = s.arena_cap;
    HIP_RET(hipMemsetAsync(s.d_out, 0, (size_t)P * 3 * sizeof(float), stream));
    unsigned long long top_max = 0, status_any = 0;
    queued = 0;
    for (int t = 0; t < tiles; t++) {
      // one tile of points: its lists, the queue of its large neighbourhoods and their arena slices, its sums — the lists array
      // and the arena are the next tile's again
      np.w0 = t * kNormalsTile;
      np.w1 = std::min(P, np.w0 + kNormalsTile);
      const int tp = np.w1 - np.w0;
      HIP_RET(hipMemsetAsync(s.d_ctl, 0, 2 * sizeof(unsigned long long), stream));
      HIP_RET(hipMemsetAsync(s.d_big, 0, sizeof(int32_t), stream));
      normals_list_kernel<<<(tp + NL_WAVES - 1) / NL_WAVES, 64 * NL_WAVES, 0, stream>>>(np);
      normals_list_big_kernel<<<256, 256, 0, stream>>>(np);
      normals_finish_kernel<<<(tp + 63) / 64, 64, 0, stream>>>(np);
      HIP_RET(hipGetLastError());
      if (tiles > 1 || t == tiles - 1) {
        HIP_RET(hipMemcpyAsync(&h, s.d_ctl, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_RET(hipMemcpyAsync(&tile_queued[(size_t)t], s.d_big, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        if (t == tiles - 1 && normals_out) HIP_RET(hipMemcpyAsync(normals_out, s.d_out, (size_t)P * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (t == tiles - 1) {
          // the device copy of the cloud (planes nx, ny, nz) behind the last tile, before the ONE wait of the call (round 6: it
          // was a second launch + wait after this one); a run that has to be repeated for a larger arena rewrites it
          split_soa_kernel<<<(P + 255) / 256, 256, 0, stream>>>(c.staging, s.d_out, c.cam_source, P, c.px, c.py, c.pz, c.nx, c.ny, c.nz, c.pxyz,
                                                                 c.pnrm);
          HIP_RET(hipGetLastError());
        }
        HIP_RET(hipStreamSynchronize(stream));  // (several tiles: the control words are the next tile's)
        top_max = std::max(top_max, h.top);
        status_any |= h.status;
        queued += tile_queued[(size_t)t];
      }
    }
    h.top = top_max;
    h.status = status_any;
    if (!(h.status & 1)) break;
    // more / larger neighbourhoods beyond a wave's reach than the arena holds: grow it to what this run asked for
    note_alloc(__func__);
    (void)hipFree(s.d_arena);
    s.d_arena = nullptr;
    s.arena_cap = 0;
    const long long want = (long long)h.top + (long long)h.top / 8;
    if (attempt == 1 || hipMalloc(&s.d_arena, (size_t)want * sizeof(unsigned long long)) != hipSuccess) {
      (void)hipGetLastError();
      set_error("normals: the neighbour lists of this cloud (%llu sort slots within %.3f m) do not fit the device memory", h.top, radius);
      return GPD_ERR_CAPACITY;
    }
    s.arena_cap = want;
  }
  s.last_queued = queued;
  c.generation++;
  return GPD_OK;
}

// ---------------------------------------------------------------------------
// hand_eval_kernel
// ---------------------------------------------------------------------------
struct HandConsts {
  double rot[GPD_MAX_SLOTS][9];
  double rot_binormal[9];
  double spacing[32];
  double spfw[32];  // spacing[i] + finger width, as isGapFree adds them (finger_hand.cpp:173-184)
  // finger slots a lateral coordinate can fall into: cell q = (t1 - lut_base) * lut_inv of 256 cells over the slots'
  // extent (the outer cells reach to infinity) -> bit mask of the slots whose open interval touches the cell widened
  // by 1 % — a superset; the exact comparisons decide
  uint32_t slot_lut[256];
  double lut_base, lut_inv;
  double depths[128];  // deepenHand's steps (finger_hand.cpp:116-121), evaluated 32 at a time
  int num_deepen;
  int nfp;  // num_finger_placements
  int slots;
  int deepen;
  int min_viable;
  double cos_friction;
  double fw, hand_depth, hand_height, init_bite;
  FilterConsts filter;  // filterGraspsWorkspace (grasp_detector.cpp:334-398), applied to the valid hands
};

struct HandParams {
  const int32_t *counts;
  const float *nn;
  const double *frames;
  int cap;
  gpd_hand *hands;
  uint8_t *fvalid;  // hand_eval_kernel: [S][slots] is_valid after filterGraspsWorkspace
  float4 *hl;       // [S][cap] neighbourhood_kernel's gather -> hand_eval_kernel; its length per sample in counts[8 s + 6]
  double radius;    // of the hand-search neighbourhood (bounds |p - sample|)
  int num_samples;
  int32_t *labels;  // reeval_kernel only: [n][8] rows of the counts table, column 5
  unsigned long long *dbg;  // profiling aid (GPD_HE_TIMING=1): per-phase cycle sums of thread 0, [8] = sum of N, [9] = sum of k
};

__device__ inline void mat3mul(const double *a, const double *b, double *c) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// Wave-wide reductions through DPP (rows of 16 lanes: quad_perm x 2, row_half_mirror, row_mirror; then row_bcast:15 / :31;
// the total is read from lane 63): 7 VALU steps per 32-bit word where six __shfl_xor steps are 6 ds_bpermute + ~25 VALU —
// hand_eval_kernel is VALU bound (profiles/pmc_mix.sh) and runs up to nineteen block reductions per hand.
template <int CTRL, int RMASK>
__device__ inline unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, RMASK, 0xf, false);  // a lane without a source keeps its own value
}
template <int CTRL, int RMASK>
__device__ inline double dpp_f64(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = dpp_u32<CTRL, RMASK>((unsigned)b), hi = dpp_u32<CTRL, RMASK>((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL, int RMASK>
__device__ inline long long dpp_i64(long long x) {
  const unsigned long long b = (unsigned long long)x;
  const unsigned lo = dpp_u32<CTRL, RMASK>((unsigned)b), hi = dpp_u32<CTRL, RMASK>((unsigned)(b >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
#define GPD_WAVE_REDUCE(v, OP, DPP)            \
  v = OP(v, (DPP<0xB1, 0xf>(v)));  /* quad_perm [1,0,3,2] */ \
  v = OP(v, (DPP<0x4E, 0xf>(v)));  /* quad_perm [2,3,0,1] */ \
  v = OP(v, (DPP<0x141, 0xf>(v))); /* row_half_mirror */     \
  v = OP(v, (DPP<0x140, 0xf>(v))); /* row_mirror */          \
  v = OP(v, (DPP<0x142, 0xa>(v))); /* row_bcast:15 */        \
  v = OP(v, (DPP<0x143, 0xc>(v)))  /* row_bcast:31: lane 63 holds the total */
__device__ inline unsigned op_or(unsigned a, unsigned b) { return a | b; }
__device__ inline double op_min(double a, double b) { return fmin(a, b); }
__device__ inline double op_max(double a, double b) { return fmax(a, b); }
// The sum is not idempotent: a lane without a DPP source must add 0, so the steps that reach across (row_bcast) and the
// symmetric ones are replaced by the xor butterfly inside rows (every lane has a source) + two masked adds.
__device__ inline long long wave_sum_i64(long long v) {
  v += dpp_i64<0xB1, 0xf>(v);
  v += dpp_i64<0x4E, 0xf>(v);
  v += dpp_i64<0x141, 0xf>(v);
  v += dpp_i64<0x140, 0xf>(v);  // every lane: the sum of its row
  const unsigned long long b = (unsigned long long)v;
  const unsigned l0 = __builtin_amdgcn_readlane((int)(unsigned)b, 0), h0 = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 0);
  const unsigned l1 = __builtin_amdgcn_readlane((int)(unsigned)b, 16), h1 = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16);
  const unsigned l2 = __builtin_amdgcn_readlane((int)(unsigned)b, 32), h2 = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 32);
  const unsigned l3 = __builtin_amdgcn_readlane((int)(unsigned)b, 48), h3 = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 48);
  return (long long)((((unsigned long long)h0 << 32) | l0) + (((unsigned long long)h1 << 32) | l1) + (((unsigned long long)h2 << 32) | l2) +
                     (((unsigned long long)h3 << 32) | l3));
}
__device__ inline unsigned lane63_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ inline double lane63_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = lane63_u32((unsigned)b), hi = lane63_u32((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Block-wide helpers (256 threads = 4 waves).
__device__ inline unsigned block_or(unsigned v, unsigned *s_tmp) {
  GPD_WAVE_REDUCE(v, op_or, dpp_u32);
  v = lane63_u32(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_tmp[0] | s_tmp[1] | s_tmp[2] | s_tmp[3];
}
__device__ inline double block_min(double v, double *s_tmp) {
  GPD_WAVE_REDUCE(v, op_min, dpp_f64);
  v = lane63_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmin(fmin(s_tmp[0], s_tmp[1]), fmin(s_tmp[2], s_tmp[3]));
}
__device__ inline double block_max(double v, double *s_tmp) {
  GPD_WAVE_REDUCE(v, op_max, dpp_f64);
  v = lane63_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(s_tmp[0], s_tmp[1]), fmax(s_tmp[2], s_tmp[3]));
}
__device__ inline long long block_sum(long long v, long long *s_tmp) {
  v = wave_sum_i64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

// The list FingerHand sees: in-height points (z in (-h, h)) followed by the ghost
// (column 0) with multiplicity N-k.  Iterates entries e in [0, N]; e == N is the
// ghost.  Returns false when the entry is not part of the list.
struct ListCtx {
  const float *nn;
  int cap, N;
  double FR[9];
  double sample[3];
  double hand_height;
  int ghost_mult;  // N - k (valid after pass 0)
  // hand_eval_kernel: the in-height entries compacted into LDS by pass 0 — forward / lateral
  // coordinate and neighbour rank of each (ct0 == nullptr: not compacted, every pass transforms
  // all N points again)
  const double *ct0, *ct1;
  const unsigned short *ce;
  int kc;
};
__device__ inline bool list_entry(const ListCtx &L, int e, double t[3], double tn[3], int &mult) {
  const int i = (e == L.N) ? 0 : e;
  const double c0 = (double)L.nn[0 * L.cap + i] - L.sample[0];
  const double c1 = (double)L.nn[1 * L.cap + i] - L.sample[1];
  const double c2 = (double)L.nn[2 * L.cap + i] - L.sample[2];
  const double n0 = (double)L.nn[3 * L.cap + i], n1 = (double)L.nn[4 * L.cap + i], n2 = (double)L.nn[5 * L.cap + i];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    t[r] = L.FR[0 + r] * c0 + L.FR[3 + r] * c1 + L.FR[6 + r] * c2;
    tn[r] = L.FR[0 + r] * n0 + L.FR[3 + r] * n1 + L.FR[6 + r] * n2;
  }
  if (e == L.N) {
    mult = L.ghost_mult;
    return L.ghost_mult > 0;
  }
  mult = 1;
  return t[2] > -1.0 * L.hand_height && t[2] < L.hand_height;
}

// f(t0, t1, mult, rank) for every entry of the list FingerHand sees (in-height points, then the
// ghost with rank N); the passes are order-free reductions, so the compacted order is as good
template <class F>
__device__ inline void for_each_entry(const ListCtx &L, F f) {
  if (L.ct0) {
    for (int c = threadIdx.x; c < L.kc; c += 256) f(L.ct0[c], L.ct1[c], 1, (int)L.ce[c]);
    if (threadIdx.x == 255 && L.ghost_mult > 0) {
      double t[3], tn[3];
      int mult;
      list_entry(L, L.N, t, tn, mult);
      f(t[0], t[1], mult, L.N);
    }
  } else {
    for (int e = threadIdx.x; e <= L.N; e += 256) {
      double t[3], tn[3];
      int mult;
      if (list_entry(L, e, t, tn, mult)) f(t[0], t[1], mult, e);
    }
  }
}

__constant__ HandConsts c_hand;
constexpr int HE_COMPACT = 2048;  // in-height entries kept in LDS by hand_eval_kernel (36 KB: four workgroups per CU; 1664 entries = five per CU
                                  // measured slower, 0.566 vs 0.557 ms of search: more samples fall back to walking their full list)

// computePointsInClosingRegion (finger_hand.cpp:141-171) + grasp width (hand_set.cpp:235-245) +
// Antipodal::evaluateGrasp (antipodal.cpp:10-96; lateral 1, forward 0, vertical 2) over the list of
// L.  Returns false when the closing region is empty; label 0 none, 1 half, 2 full.
__device__ bool closing_region_label(const ListCtx &L, int N, double top, double bottom, double left, double right, const HandConsts &K,
                                     double &width, int &label, unsigned *s_u, double *s_d, long long *s_l) {
  double ymin = DBL_MAX, ymax = -DBL_MAX;
  unsigned any_c = 0;
  for_each_entry(L, [&](double t0, double t1, int, int) {
    if (t0 > bottom && t0 < top && t1 > left && t1 < right) {
      any_c = 1;
      ymin = fmin(ymin, t1);
      ymax = fmax(ymax, t1);
    }
  });
  any_c = block_or(any_c, s_u);
  label = 0;
  width = 0.0;
  if (!any_c) return false;
  ymin = block_min(ymin, s_d);
  ymax = block_max(ymax, s_d);
  width = ymax - ymin;
  const double min_x = ymin + 0.003, max_x = ymax - 0.003;
  double lx0 = DBL_MAX, lx1 = -DBL_MAX, lz0 = DBL_MAX, lz1 = -DBL_MAX;
  double rx0 = DBL_MAX, rx1 = -DBL_MAX, rz0 = DBL_MAX, rz1 = -DBL_MAX;
  unsigned lr = 0;
  for_each_entry(L, [&](double t0, double t1, int, int e) {
    if (!(t0 > bottom && t0 < top && t1 > left && t1 < right)) return;
    double t[3], tn[3];  // the vertical coordinate and the normal only of the few points between the fingers
    int mult;
    list_entry(L, e, t, tn, mult);
    const double ln = 0.0 * tn[0] + -1.0 * tn[1] + 0.0 * tn[2];
    const double rn = 0.0 * tn[0] + 1.0 * tn[1] + 0.0 * tn[2];
    if (ln > K.cos_friction && t[1] < min_x) {
      lr |= 1u;
      lx0 = fmin(lx0, t[0]);
      lx1 = fmax(lx1, t[0]);
      lz0 = fmin(lz0, t[2]);
      lz1 = fmax(lz1, t[2]);
    }
    if (rn > K.cos_friction && t[1] > max_x) {
      lr |= 2u;
      rx0 = fmin(rx0, t[0]);
      rx1 = fmax(rx1, t[0]);
      rz0 = fmin(rz0, t[2]);
      rz1 = fmax(rz1, t[2]);
    }
  });
  lr = block_or(lr, s_u);
  if (lr) label = 1;
  if (lr == 3u) {
    lx0 = block_min(lx0, s_d);
    lx1 = block_max(lx1, s_d);
    lz0 = block_min(lz0, s_d);
    lz1 = block_max(lz1, s_d);
    rx0 = block_min(rx0, s_d);
    rx1 = block_max(rx1, s_d);
    rz0 = block_min(rz0, s_d);
    rz1 = block_max(rz1, s_d);
    const double top_y = fmin(lx1, rx1), bot_y = fmax(lx0, rx0);
    const double top_z = fmin(lz1, rz1), bot_z = fmax(lz0, rz0);
    long long nl = 0, nr = 0;
    for_each_entry(L, [&](double t0, double t1, int mult, int e) {
      if (!(t0 > bottom && t0 < top && t1 > left && t1 < right)) return;
      double t[3], tn[3];
      int m1;
      list_entry(L, e, t, tn, m1);
      const double ln = 0.0 * tn[0] + -1.0 * tn[1] + 0.0 * tn[2];
      const double rn = 0.0 * tn[0] + 1.0 * tn[1] + 0.0 * tn[2];
      const bool inb = t[0] >= bot_y && t[0] <= top_y && t[2] >= bot_z && t[2] <= top_z;
      if (ln > K.cos_friction && t[1] < min_x && inb) nl += mult;
      if (rn > K.cos_friction && t[1] > max_x && inb) nr += mult;
    });
    nl = block_sum(nl, s_l);
    nr = block_sum(nr, s_l);
    if (nl >= K.min_viable && nr >= K.min_viable) label = 2;
  }
  return true;
}

// HandSearch::reevaluateHypotheses (hand_search.cpp:66-134, 190-228): one
// workgroup per hand checks it again against the uploaded (ground-truth) cloud with the hand's own
// frame, depth and finger placement: evaluateFingers(points, top, idx), evaluateHand(idx), closing
// region, antipodal label.  counts[8*i+5] receives the label (1 = full antipodal grasp).
__global__ __launch_bounds__(256) void reeval_kernel(HandParams P) {
  __shared__ unsigned s_u[4];
  __shared__ double s_d[4];
  __shared__ long long s_l[4];
  const HandConsts &K = c_hand;
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int N = P.counts[8 * s + 0];
  gpd_hand *H = P.hands + s;
  const int idx = H->finger_placement_index;
  int label = 0;
  if (N > 0 && idx >= 0 && idx < K.nfp) {
    ListCtx L;
    L.nn = P.nn + (size_t)s * 6 * P.cap;
    L.cap = P.cap;
    L.N = N;
    L.hand_height = K.hand_height;
#pragma unroll
    for (int r = 0; r < 3; r++) L.sample[r] = H->sample[r];
#pragma unroll
    for (int i = 0; i < 9; i++) L.FR[i] = H->frame[i];
    L.ghost_mult = 0;
    L.ct0 = nullptr;
    L.ct1 = nullptr;
    L.ce = nullptr;
    L.kc = 0;
    long long kin = 0;
    for (int e = tid; e < N; e += 256) {
      double t[3], tn[3];
      int mult;
      if (list_entry(L, e, t, tn, mult)) kin++;
    }
    L.ghost_mult = N - (int)block_sum(kin, s_l);
    const double top = H->top, bottom = top - K.hand_depth;
    const double sl = K.spacing[idx], sr = K.spacing[K.nfp + idx];
    unsigned flags = 0;  // bit0 some x<top, bit1 some x<bottom, bit2 a finger of the pair is blocked
    for (int e = tid; e <= N; e += 256) {
      double t[3], tn[3];
      int mult;
      if (!list_entry(L, e, t, tn, mult)) continue;
      if (t[0] < top) {
        flags |= 1u;
        if (t[0] < bottom) flags |= 2u;
        if ((t[1] > sl && t[1] < sl + K.fw) || (t[1] > sr && t[1] < sr + K.fw)) flags |= 4u;
      }
    }
    flags = block_or(flags, s_u);
    if (flags == 1u) {
      double width;
      closing_region_label(L, N, top, bottom, sl + K.fw, sr, K, width, label, s_u, s_d, s_l);
    }
  }
  if (tid == 0) {
    H->half_antipodal = label == 1;
    H->full_antipodal = label == 2;
    P.labels[8 * s + 5] = label == 2 ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void hand_eval_kernel(HandParams P) {
  __shared__ unsigned s_u[4];
  __shared__ double s_d[4];
  __shared__ long long s_l[4];
  const HandConsts &K = c_hand;
  // XCD-aware order (workgroup L runs on XCD L % 8, one L2 per XCD): all orientations of a sample
  // run on the same XCD, so its neighbourhood is fetched from HBM once, not once per XCD
  // (1.87 GB per launch before, 9x the algorithmic bytes).  L = 8 * (slots * g + slot) + xcd.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int s = (idx / K.slots) * 8 + xcd;
  const int slot = idx % K.slots;
  if (s >= P.num_samples) return;
  const int tid = threadIdx.x;
  const int N = P.counts[8 * s + 0];
  const int kf = P.counts[8 * s + 2];
  gpd_hand *H = P.hands + (size_t)s * K.slots + slot;
  const double *fr = P.frames + 12 * (size_t)s;
  // The 184-byte record is put together in LDS and stored by 46 lanes, one dword each: a gpd_hand on thread 0's
  // stack lives in scratch memory (184 B/lane), and writing it there and copying it out was a tenth of the kernel.
  constexpr int REC_DW = (int)(sizeof(gpd_hand) / 4);
  static_assert(sizeof(gpd_hand) % 4 == 0 && REC_DW <= 64, "record store");
  __shared__ gpd_hand s_rec;
  if (kf == 0 || N == 0) {  // no frame: the sample is dropped on the host
    if (tid < REC_DW) {
      uint32_t v = 0u;
      if (tid == (int)(offsetof(gpd_hand, finger_placement_index) / 4)) v = 0xffffffffu;  // -1
      if (tid == (int)(offsetof(gpd_hand, slot) / 4)) v = (uint32_t)slot;
      reinterpret_cast<uint32_t *>(H)[tid] = v;
    }
    if (tid == 0) P.fvalid[(size_t)s * K.slots + slot] = 0;
    return;
  }
  if (tid < REC_DW) reinterpret_cast<uint32_t *>(&s_rec)[tid] = 0u;  // ordered before thread 0's fields by the barriers below
  long long t_last = P.dbg ? clock64() : 0;
#define HTICK(k)                                  \
  if (P.dbg && tid == 0) {                        \
    const long long now_ = clock64();             \
    atomicAdd(&P.dbg[k], (unsigned long long)(now_ - t_last)); \
    t_last = now_;                                \
  }
  ListCtx L;
  L.nn = P.nn + (size_t)s * 6 * P.cap;
  L.cap = P.cap;
  L.N = N;
  L.hand_height = K.hand_height;
  // frame_ << normal, binormal, curvature (hand_set.cpp:39-40); frame_rot = frame_ * RB * R (:72)
  double F[9], FRB[9];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    L.sample[r] = fr[r];
    F[3 * r + 0] = fr[3 + r];
    F[3 * r + 1] = fr[6 + r];
    F[3 * r + 2] = fr[9 + r];
  }
  mat3mul(F, K.rot_binormal, FRB);
  mat3mul(FRB, K.rot[slot], L.FR);
  L.ghost_mult = 0;

  const int nfp = K.nfp;
  const double bite = K.init_bite;
  const double bottom0 = bite - K.hand_depth;
  // ---- pass 0: the one transform of all N points (PointList::transformToHandFrame +
  //      cropByHandHeight, point_list.cpp:22-55): in-height entries are compacted into LDS
  //      (forward / lateral coordinate, rank), every later pass reads those k entries instead of
  //      transforming the N points again (the kernel is fp64-VALU bound: k is ~0.3 N).
  __shared__ double s_t0[HE_COMPACT], s_t1[HE_COMPACT];
  __shared__ unsigned short s_e[HE_COMPACT];
  __shared__ int s_kc;
  __shared__ uint32_t s_lut[256];
  __shared__ double s_sp[32], s_spfw[32];
  if (tid == 0) s_kc = 0;
  s_lut[tid] = K.slot_lut[tid];
  if (tid < 32) {
    s_sp[tid] = K.spacing[tid];
    s_spfw[tid] = K.spfw[tid];
  }
  __syncthreads();
  L.ct0 = nullptr;
  L.ct1 = nullptr;
  L.ce = nullptr;
  L.kc = 0;
  // The points come from the sample's height list (written by neighbourhood_kernel's gather: the superset of every orientation's crop, one
  // 16-byte record per point); two rounds are in flight per thread.
  const float4 *hl = P.hl + (size_t)s * P.cap;
  const int NL = P.counts[8 * s + 6];
  constexpr int HE_R = 2;
  float4 cx[2][HE_R];
  auto request = [&](int i0, float4(&c)[HE_R]) {
#pragma unroll
    for (int q = 0; q < HE_R; q++) c[q] = hl[min(i0 + 256 * q + tid, NL - 1)];
  };
  auto consume = [&](int i0, const float4(&c)[HE_R]) {
#pragma unroll
    for (int q = 0; q < HE_R; q++) {
      const int i = i0 + 256 * q + tid;
      if (i0 + 256 * q >= NL) break;  // uniform
      double t[3] = {0, 0, 0};
      bool in = false;
      if (i < NL) {  // list_entry() on the listed coordinates: transformToHandFrame + the height test
        const double c0 = (double)c[q].x - L.sample[0], c1 = (double)c[q].y - L.sample[1], c2 = (double)c[q].z - L.sample[2];
#pragma unroll
        for (int r = 0; r < 3; r++) t[r] = L.FR[0 + r] * c0 + L.FR[3 + r] * c1 + L.FR[6 + r] * c2;
        in = t[2] > -1.0 * L.hand_height && t[2] < L.hand_height;
      }
      const unsigned long long ballot = __ballot(in);
      if (ballot) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&s_kc, __popcll(ballot));
        base = __builtin_amdgcn_readfirstlane(base);  // every lane is active here and lane 0 holds it
        if (in) {
          const int cpos = base + __popcll(ballot & ((1ull << (tid & 63)) - 1ull));
          if (cpos < HE_COMPACT) {
            s_t0[cpos] = t[0];
            s_t1[cpos] = t[1];
            s_e[cpos] = (unsigned short)__float_as_int(c[q].w);
          }
        }
      }
    }
  };
  if (NL > 0) {
    request(0, cx[0]);
    for (int i0 = 0; i0 < NL; i0 += 2 * 256 * HE_R) {
      if (i0 + 256 * HE_R < NL) request(i0 + 256 * HE_R, cx[1]);
      consume(i0, cx[0]);
      if (i0 + 256 * HE_R >= NL) break;
      if (i0 + 2 * 256 * HE_R < NL) request(i0 + 2 * 256 * HE_R, cx[0]);
      consume(i0 + 256 * HE_R, cx[1]);
    }
  }
  __syncthreads();
  HTICK(0);
  const int k = s_kc;
  L.ghost_mult = N - k;
  if (P.dbg && tid == 0) {
    atomicAdd(&P.dbg[8], (unsigned long long)N);
    atomicAdd(&P.dbg[9], (unsigned long long)k);
  }
  if (k <= HE_COMPACT && N <= 65535) {  // (the table's neighbour ranks are 16 bits wide: a longer list is walked in full)
    L.ct0 = s_t0;
    L.ct1 = s_t1;
    L.ce = s_e;
    L.kc = k;
  }
  // ---- pass 1: finger collision masks at init_bite (finger_hand.cpp:26-73)
  unsigned blocked = 0, flags = 0;  // flags bit0: some x<bite, bit1: some x<bottom
  // isGapFree for all 2 nfp slots: the slots an entry can block come from the lookup table (none, one, rarely
  // two), the reference's own comparisons decide — twenty pairs of comparisons per entry were 31 of the kernel's
  // 85 kcycles per workgroup
  for_each_entry(L, [&](double t0, double t1, int, int) {
    if (t0 < bite) {
      flags |= 1u;
      if (t0 < bottom0) flags |= 2u;
      const double u = (t1 - K.lut_base) * K.lut_inv;
      const int q = u < 0.0 ? 0 : (u >= 255.0 ? 255 : (int)u);
      unsigned m = s_lut[q] & ~blocked;
      while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1u;
        if (t1 > s_sp[i] && t1 < s_spfw[i]) blocked |= 1u << i;
      }
    }
  });
  blocked = block_or(blocked, s_u);
  flags = block_or(flags, s_u);
  HTICK(1);
  unsigned fingers = 0;
  if ((flags & 1u) && !(flags & 2u)) fingers = ~blocked & ((1u << (2 * nfp)) - 1u);
  unsigned hand = fingers & (fingers >> nfp) & ((1u << nfp) - 1u);

  double top = bite, bottom = bottom0, center = 0.0;
  int fidx = hand ? (__ffs(hand) - 1) : -1;
  bool valid = false;
  double width = 0.0;
  int label = 0;
  if (hand) {
    // chooseMiddleHand (finger_hand.cpp:89-105): idx[ceil(m/2) - 1]
    const int mcount = __popc(hand);
    const int want = (mcount + 1) / 2 - 1;
    int mid = -1, seen = 0;
    for (int i = 0; i < nfp; i++)
      if (hand & (1u << i)) {
        if (seen == want) {
          mid = i;
          break;
        }
        seen++;
      }
    if (K.deepen) {
      // ---- pass 2: deepenHand (finger_hand.cpp:107-139), 32 depths at once (the shipped geometry has ten; a longer
      //      finger simply takes another round of the loop — the reference has no bound on hand_depth)
      const double sl = K.spacing[mid], sr = K.spacing[nfp + mid];
      bool deeper = true;
      for (int j0 = 0; j0 < K.num_deepen && deeper; j0 += 32) {
        const int nj = min(32, K.num_deepen - j0);
        unsigned m_any = 0, m_back = 0, m_blk = 0;
        for_each_entry(L, [&](double t0, double t1, int, int) {
          const bool blk = (t1 > sl && t1 < sl + K.fw) || (t1 > sr && t1 < sr + K.fw);
          for (int j = 0; j < nj; j++) {
            const double d = K.depths[j0 + j];
            if (t0 < d) {
              m_any |= 1u << j;
              if (t0 < d - K.hand_depth) m_back |= 1u << j;
              if (blk) m_blk |= 1u << j;
            }
          }
        });
        m_any = block_or(m_any, s_u);
        m_back = block_or(m_back, s_u);
        m_blk = block_or(m_blk, s_u);
        for (int j = 0; j < nj; j++) {
          const bool ok = (m_any >> j & 1u) && !(m_back >> j & 1u) && !(m_blk >> j & 1u);
          if (!ok) {
            deeper = false;
            break;
          }
          top = K.depths[j0 + j];
          bottom = K.depths[j0 + j] - K.hand_depth;
        }
      }
      HTICK(2);
      hand = 1u << mid;
    }
    // ---- pass 3-5: closing region, width, antipodal label
    const double left = K.spacing[mid] + K.fw;
    const double right = K.spacing[nfp + mid];
    const double center_c = 0.5 * (left + right);
    if (closing_region_label(L, N, top, bottom, left, right, K, width, label, s_u, s_d, s_l)) {
      valid = true;
      center = center_c;
      fidx = __ffs(hand) - 1;
    } else {
      // closing region empty: the hand keeps what Hand() saw before deepening (hand_set.cpp:89-109)
      top = bite;
      bottom = bottom0;
      center = 0.0;
      fidx = __ffs(fingers & (fingers >> nfp) & ((1u << nfp) - 1u)) - 1;
    }
    HTICK(3);
  }
  if (tid == 0) {
    gpd_hand &h = s_rec;
#pragma unroll
    for (int r = 0; r < 3; r++) h.sample[r] = L.sample[r];
#pragma unroll
    for (int i = 0; i < 9; i++) h.frame[i] = L.FR[i];
    h.top = top;
    h.bottom = bottom;
    h.center = center;
    // Hand::calculateGraspPositions (hand.cpp:41-45)
#pragma unroll
    for (int r = 0; r < 3; r++) h.position[r] = (L.FR[3 * r + 0] * bottom + L.FR[3 * r + 1] * center + L.FR[3 * r + 2] * 0.0) + L.sample[r];
    h.grasp_width = width;
    h.finger_placement_index = fidx;
    h.set_index = s;
    h.slot = slot;
    h.valid = valid ? 1 : 0;
    h.half_antipodal = label >= 1;
    h.full_antipodal = label == 2;
    // detectGrasps step 2 (grasp_detector.cpp:238): the workspace / aperture filter only clears is_valid, the
    // record itself stays as the search produced it
    P.fvalid[(size_t)s * K.slots + slot] = (valid && workspace_ok(K.filter, h)) ? 1 : 0;
  }
  __syncthreads();
  if (tid < REC_DW) reinterpret_cast<uint32_t *>(H)[tid] = reinterpret_cast<const uint32_t *>(&s_rec)[tid];
  HTICK(4);
#undef HTICK
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
void search_free(SearchState &s) {
  void *ptrs[] = {s.d_sample_idx, s.d_sample_xyz, s.d_counts, s.d_nn_idx, s.d_nn, s.d_frames, s.d_centers, s.d_hands, s.d_fvalid, s.d_hl};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);  // hipFree waits for the device: the side stream is idle as well
  if (s.ev_fork) (void)hipEventDestroy(s.ev_fork);
  if (s.ev_join) (void)hipEventDestroy(s.ev_join);
  if (s.aux) (void)hipStreamDestroy(s.aux);
  s = SearchState();
}

static int search_reserve(SearchState &s, int S, int cap, int slots) {
  if (S <= s.capacity_samples && cap == s.nn_cap) return GPD_OK;
  int newS = S > s.capacity_samples ? S + S / 8 : s.capacity_samples;  // slack: the clouds of a batch differ a little
  // (a list-capacity change keeps the sample capacity the lane was sized for — while the lists are the LDS-sorted sizes: the
  //  global-memory lists of a dense scan cost 44 bytes per entry and sample, up to 46 MB per sample at 2^20 entries, and are
  //  sized for the call at hand only: 2884 reserved samples x 200k entries would be 25 GB)
  if (newS < s.min_samples && cap <= 16384) newS = s.min_samples;
  {
    // 44 bytes per list entry and sample (gathered rows, index scratch, height list): with the large lists of a dense
    // scan that is what bounds a call, and it should say so rather than fail inside hipMalloc
    size_t free_b = 0, total_b = 0;
    const size_t want = (size_t)newS * (size_t)cap * 44u;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b + (size_t)s.capacity_samples * (size_t)s.nn_cap * 44u) {
      set_error("search: %d samples with neighbourhood lists of %d entries need %.1f GB of device memory (%.1f GB free): pass fewer samples per call",
                newS, cap, want / 1e9, free_b / 1e9);
      return GPD_ERR_CAPACITY;
    }
  }
  note_alloc(__func__);
  const uint64_t seen = s.seen_generation;
  search_free(s);
  s.seen_generation = seen;  // (search_free resets the whole state; the cloud the lists belong to is still the same)
  HIP_RET(hipMalloc(&s.d_sample_idx, (size_t)newS * sizeof(int32_t)));
  HIP_RET(hipMalloc(&s.d_sample_xyz, (size_t)newS * 3 * sizeof(double)));
  HIP_RET(hipMalloc(&s.d_counts, (size_t)newS * 8 * sizeof(int32_t)));
  HIP_RET(hipMalloc(&s.d_nn_idx, (size_t)newS * cap * sizeof(int32_t)));
  HIP_RET(hipMalloc(&s.d_nn, (size_t)newS * 6 * cap * sizeof(float)));
  HIP_RET(hipMalloc(&s.d_frames, (size_t)newS * 12 * sizeof(double)));
  HIP_RET(hipMalloc(&s.d_centers, (size_t)newS * 3 * sizeof(double)));
  HIP_RET(hipMalloc(&s.d_hands, (size_t)newS * slots * sizeof(gpd_hand)));
  HIP_RET(hipMalloc(&s.d_fvalid, (size_t)newS * slots));
  HIP_RET(hipMalloc(&s.d_hl, (size_t)newS * cap * sizeof(float4)));
  s.capacity_samples = newS;
  s.nn_cap = cap;
  return GPD_OK;
}

int search_reserve_samples(SearchState &s, int S, int slots) { return search_reserve(s, S, s.nn_cap ? s.nn_cap : 8192, slots); }

static int run_neighbourhoods(const gpd_params &p, const Cloud &c, SearchState &s, const HostConsts &hc, int S, int cap, bool by_xyz,
                              int slots, bool want_height_list, hipStream_t stream, bool sync_counts) {
  NbParams np;
  np.px = c.px; np.py = c.py; np.pz = c.pz; np.nx = c.nx; np.ny = c.ny; np.nz = c.nz;
  np.pxyz = c.pxyz;
  np.pnrm = c.pnrm;
  np.num_points = c.num_points;
  np.sample_idx = s.d_sample_idx;
  np.sample_xyz = by_xyz ? s.d_sample_xyz : nullptr;
  // pcl::KdTreeFLANN::radiusSearch passes (float)(radius*radius) to FLANN
  const double r_all = std::fmax(hc.nn_radius_hands, std::fmax(hc.nn_radius_images, p.nn_radius_frames));
  np.r2_all = (float)(r_all * r_all);
  np.r2_hands = (float)(hc.nn_radius_hands * hc.nn_radius_hands);
  np.r2_images = (float)(hc.nn_radius_images * hc.nn_radius_images);
  np.r2_frames = (float)(p.nn_radius_frames * p.nn_radius_frames);
  np.cap = cap;
  np.counts = s.d_counts;
  np.nn_idx = s.d_nn_idx;
  np.nn = s.d_nn;
  np.frames = s.d_frames;
  np.centers = s.d_centers;
  np.cam_source = c.cam_source;
  np.num_cams = c.num_cams;
  np.grid = grid_view(c);
  np.reach = (float)r_all * 1.001f + 1e-5f;
  np.hl = want_height_list ? s.d_hl : nullptr;
  np.hl_slots = slots;
  for (int slot = 0; slot < slots && slot < GPD_MAX_SLOTS; slot++)
    for (int r = 0; r < 3; r++)  // third column of rot_binormal * rot[slot]
      np.hl_col[slot][r] = hc.rot_binormal[3 * r] * hc.rot[slot][2] + hc.rot_binormal[3 * r + 1] * hc.rot[slot][5] +
                           hc.rot_binormal[3 * r + 2] * hc.rot[slot][8];
  np.hl_height = p.hand_height;
  np.hl_radius = hc.nn_radius_hands * 1.001 + 1e-6;
  static unsigned long long *d_nbdbg = nullptr;
  np.dbg = nullptr;
  if (prof_env("GPD_NB_TIMING")) {
    if (!d_nbdbg) HIP_RET(hipMalloc(&d_nbdbg, 10 * sizeof(unsigned long long)));
    HIP_RET(hipMemsetAsync(d_nbdbg, 0, 10 * sizeof(unsigned long long), stream));
    np.dbg = d_nbdbg;
  }
  // 8192-entry lists are bucket-sorted (64 + 8 KB of LDS: two workgroups per CU); the 16384-entry
  // retry of an overfull neighbourhood sorts in place (bitonic, 128 KB); anything larger (up to kNnCapMax) is
  // bucket-sorted in global memory
  if (cap > 16384) {
    np.bucket = 1;
    const size_t lds = (size_t)(2 * 8192 + 1) * sizeof(int);
    HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void *>(neighbourhood_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    neighbourhood_kernel<true><<<S, NB_THREADS, lds, stream>>>(np);
  } else {
    np.bucket = cap < 16384 ? 1 : 0;
    const size_t lds = (size_t)cap * sizeof(unsigned long long) + (np.bucket ? (2 * NB_BUCKETS + 1) * sizeof(int) : 0);
    HIP_RET(hipFuncSetAttribute(reinterpret_cast<const void *>(neighbourhood_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    neighbourhood_kernel<false><<<S, NB_THREADS, lds, stream>>>(np);
  }
  HIP_RET(hipGetLastError());
  if (np.dbg) {
    unsigned long long h[10];
    HIP_RET(hipMemcpyAsync(h, d_nbdbg, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipStreamSynchronize(stream));
    static const char *names[10] = {"visit", "scan", "scatter", "bucket sort", "crowded buckets", "prefix lengths", "gather",
                                    "last barrier", "wait for the frame", "height list"};
    for (int i = 0; i < 10; i++) fprintf(stderr, "[nb-timing] %-16s %8.1f kcycles/sample\n", names[i], (double)h[i] / S / 1e3);
  }
  // the centre sums fork off to the side stream; search_join() brings them back
  if (!s.aux) {
    HIP_RET(hipStreamCreate(&s.aux));
    HIP_RET(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
    HIP_RET(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
  }
  HIP_RET(hipEventRecord(s.ev_fork, stream));
  HIP_RET(hipStreamWaitEvent(s.aux, s.ev_fork, 0));
  centre_kernel<<<(3 * S + 63) / 64, 64, 0, s.aux>>>(s.d_nn, s.d_counts, cap, S, s.d_centers);
  HIP_RET(hipGetLastError());
  HIP_RET(hipEventRecord(s.ev_join, s.aux));
  s.h_counts.clear();
  if (sync_counts) {
    s.h_counts.resize((size_t)S * 8);
    HIP_RET(hipMemcpyAsync(s.h_counts.data(), s.d_counts, (size_t)S * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipStreamSynchronize(stream));
  }
  return GPD_OK;
}

// the main stream waits for the centre sums of the last run_neighbourhoods
static int search_join(SearchState &s, hipStream_t stream) {
  if (s.ev_join) HIP_RET(hipStreamWaitEvent(stream, s.ev_join, 0));
  return GPD_OK;
}

// neighbourhoods of S samples (by index or by coordinates), list capacity grown once if needed
static int neighbourhoods(const gpd_params &p, const Cloud &c, SearchState &s, const HostConsts &hc, const int32_t *sample_idx,
                          const double *sample_xyz, int S, int slots, int *cap_out, hipStream_t stream, bool sync_counts = true,
                          bool want_height_list = true) {
  // a new cloud starts from the LDS-sorted list size again: the large lists one dense cloud needed are not carried into the
  // next one (they come back through the retry below if it needs them too)
  if (s.seen_generation != c.generation) {
    s.seen_generation = c.generation;
    if (s.nn_cap > 16384) {
      const int rcf = search_force_capacity(s, 8192);
      if (rcf) return rcf;
    }
  }
  int cap = s.nn_cap ? s.nn_cap : 8192;
  int rc = search_reserve(s, S, cap, slots);
  if (rc) return rc;
  for (;;) {
    if (sample_xyz)
      HIP_RET(hipMemcpyAsync(s.d_sample_xyz, sample_xyz, (size_t)S * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
    else
      HIP_RET(hipMemcpyAsync(s.d_sample_idx, sample_idx, (size_t)S * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    rc = run_neighbourhoods(p, c, s, hc, S, cap, sample_xyz != nullptr, slots, want_height_list, stream, sync_counts);
    if (rc) return rc;
    if (!sync_counts) break;  // the caller reads `worst found` from the plan summary and retries (search_next_capacity)
    int worst = 0;
    for (int i = 0; i < S; i++) worst = s.h_counts[8 * i + 3] > worst ? s.h_counts[8 * i + 3] : worst;
    if (worst <= cap) break;
    cap = search_next_capacity(s, worst);
    if (!cap) {
      set_error("search: a neighbourhood holds %d points, more than the list capacity %d", worst, kNnCapMax);
      return GPD_ERR_CAPACITY;
    }
    rc = search_reserve(s, S, cap, slots);
    if (rc) return rc;
  }
  *cap_out = cap;
  return GPD_OK;
}

// c_hand is ONE block per device, shared by every context of the process on that device.  A context
// whose constants differ from the loaded ones waits for the device (kernels of another context, on
// another stream, may still be reading the block) before it overwrites them; the lock is handed
// back to the caller, who keeps it until its kernels that read the block are enqueued.
static std::mutex g_hand_mutex;
static std::vector<unsigned char> g_hand_loaded[64];

static int upload_hand_consts(const gpd_params &p, const HostConsts &hc, int slots, hipStream_t stream,
                              std::unique_lock<std::mutex> &lock) {
  HandConsts hk;
  std::memset(&hk, 0, sizeof(hk));
  std::memcpy(hk.rot, hc.rot, sizeof(hk.rot));
  std::memcpy(hk.rot_binormal, hc.rot_binormal, sizeof(hk.rot_binormal));
  std::memcpy(hk.spacing, hc.finger_spacing, sizeof(hk.spacing));
  std::memcpy(hk.depths, hc.deepen_depths, sizeof(hk.depths));
  {
    const int nslots = 2 * p.num_finger_placements;
    double lo = DBL_MAX, hi = -DBL_MAX;
    for (int i = 0; i < nslots; i++) {
      hk.spfw[i] = hk.spacing[i] + p.finger_width;
      lo = std::fmin(lo, hk.spacing[i]);
      hi = std::fmax(hi, hk.spfw[i]);
    }
    const double w = (hi - lo) / 256.0;
    hk.lut_base = lo;
    hk.lut_inv = w > 0.0 ? 1.0 / w : 0.0;
    for (int c = 0; c < 256; c++) {
      const double c_lo = c == 0 ? -DBL_MAX : lo + (c - 0.01) * w - 1e-12;
      const double c_hi = c == 255 ? DBL_MAX : lo + (c + 1.01) * w + 1e-12;
      uint32_t m = 0;
      for (int i = 0; i < nslots; i++)
        if (hk.spacing[i] < c_hi && hk.spfw[i] > c_lo) m |= 1u << i;
      hk.slot_lut[c] = w > 0.0 ? m : (nslots >= 32 ? 0xffffffffu : (1u << nslots) - 1u);
    }
  }
  hk.num_deepen = hc.num_deepen;
  hk.nfp = p.num_finger_placements;
  hk.slots = slots;
  hk.deepen = p.deepen_hand;
  hk.min_viable = p.min_viable;
  hk.cos_friction = hc.cos_friction;
  hk.fw = p.finger_width;
  hk.hand_depth = p.hand_depth;
  hk.hand_height = p.hand_height;
  hk.init_bite = p.init_bite;
  hk.filter = filter_consts(p);
  lock = std::unique_lock<std::mutex>(g_hand_mutex);
  int dev = 0;
  HIP_RET(hipGetDevice(&dev));
  std::vector<unsigned char> &loaded = g_hand_loaded[dev & 63];
  const unsigned char *bytes = reinterpret_cast<const unsigned char *>(&hk);
  if (loaded.size() == sizeof(hk) && std::memcmp(loaded.data(), bytes, sizeof(hk)) == 0) return GPD_OK;
  if (!loaded.empty()) HIP_RET(hipDeviceSynchronize());
  HIP_RET(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_hand), &hk, sizeof(hk), 0, hipMemcpyHostToDevice, stream));
  // the block counts as loaded only once the copy has landed: a context with the same values skips
  // the copy and launches on ITS stream, which is not ordered after this one
  HIP_RET(hipStreamSynchronize(stream));
  loaded.assign(bytes, bytes + sizeof(hk));
  return GPD_OK;
}

int search_next_capacity(const SearchState &s, int worst) {
  if (worst <= s.nn_cap) return s.nn_cap;
  if (worst <= 16384) return 16384;  // in-place bitonic sort, 128 KB of LDS per workgroup
  if (worst > kNnCapMax) return 0;
  const int cap = (worst + 4095) / 4096 * 4096;  // global-memory lists
  return cap < kNnCapMax ? cap : kNnCapMax;
}
int search_force_capacity(SearchState &s, int cap) {
  if (cap == s.nn_cap) return GPD_OK;
  const int keep = s.capacity_samples > s.min_samples ? s.capacity_samples : s.min_samples;
  const uint64_t seen = s.seen_generation;
  search_free(s);
  s.seen_generation = seen;
  s.nn_cap = cap;  // search_reserve allocates on the next run (capacity_samples is 0 now) ...
  s.min_samples = keep;  // ... for at least as many samples as before: a lane sized by gpd_hip_reserve stays sized
  return GPD_OK;
}

int search_run(const gpd_params &p, const Cloud &c, SearchState &s, const int32_t *sample_idx, const double *sample_xyz, int S,
               hipStream_t stream, bool sync_counts) {
  const int slots = p.num_hand_axes * p.num_orientations;
  HostConsts hc;
  host_consts(p, hc);
  int cap = 0;
  int rc = neighbourhoods(p, c, s, hc, sample_idx, sample_xyz, S, slots, &cap, stream, sync_counts);
  if (rc) return rc;
  std::unique_lock<std::mutex> consts_lock;  // held until the kernels that read c_hand are enqueued
  rc = upload_hand_consts(p, hc, slots, stream, consts_lock);
  if (rc) return rc;
  HandParams hp;
  hp.counts = s.d_counts;
  hp.nn = s.d_nn;
  hp.frames = s.d_frames;
  hp.cap = cap;
  hp.hands = s.d_hands;
  hp.fvalid = s.d_fvalid;
  hp.labels = nullptr;
  hp.num_samples = S;
  static unsigned long long *d_hedbg = nullptr;
  hp.dbg = nullptr;
  if (prof_env("GPD_HE_TIMING")) {
    if (!d_hedbg) HIP_RET(hipMalloc(&d_hedbg, 16 * sizeof(unsigned long long)));
    HIP_RET(hipMemsetAsync(d_hedbg, 0, 16 * sizeof(unsigned long long), stream));
    hp.dbg = d_hedbg;
  }
  hp.hl = s.d_hl;
  hp.radius = hc.nn_radius_hands * 1.001 + 1e-6;
  hand_eval_kernel<<<((S + 7) / 8) * 8 * slots, 256, 0, stream>>>(hp);
  HIP_RET(hipGetLastError());
  if (hp.dbg) {
    unsigned long long h[16];
    HIP_RET(hipMemcpyAsync(h, d_hedbg, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipStreamSynchronize(stream));
    static const char *names[5] = {"transform+compact", "finger masks", "deepen", "closing region", "record"};
    const double wg = (double)S * slots;
    for (int i = 0; i < 5; i++) fprintf(stderr, "[he-timing] %-18s %8.2f kcycles/workgroup\n", names[i], (double)h[i] / wg / 1e3);
    fprintf(stderr, "[he-timing] mean N %.0f, mean in-height k %.0f\n", (double)h[8] / wg, (double)h[9] / wg);
  }
  rc = search_join(s, stream);
  if (rc) return rc;
  s.num_samples = S;
  s.cloud_generation = c.generation;
  return GPD_OK;
}

int reevaluate_run(const gpd_params &p, const Cloud &c, SearchState &s, gpd_hand *hands, int n, int32_t *labels, hipStream_t stream) {
  const int slots = p.num_hand_axes * p.num_orientations;
  HostConsts hc;
  host_consts(p, hc);
  s.num_samples = 0;  // the buffers are reused: hands of an earlier search can no longer be imaged
  s.h_set_sample.clear();
  s.h_samples.clear();
  std::vector<double> xyz((size_t)n * 3);
  for (int i = 0; i < n; i++)
    for (int r = 0; r < 3; r++) xyz[3 * (size_t)i + r] = hands[i].sample[r];
  int cap = 0;
  int rc = neighbourhoods(p, c, s, hc, nullptr, xyz.data(), n, slots, &cap, stream, true, /*want_height_list=*/false);
  if (rc) return rc;
  std::unique_lock<std::mutex> consts_lock;  // held until the kernels that read c_hand are enqueued
  rc = upload_hand_consts(p, hc, slots, stream, consts_lock);
  if (rc) return rc;
  HIP_RET(hipMemcpyAsync(s.d_hands, hands, (size_t)n * sizeof(gpd_hand), hipMemcpyHostToDevice, stream));
  HandParams hp;
  hp.counts = s.d_counts;
  hp.nn = s.d_nn;
  hp.frames = s.d_frames;
  hp.cap = cap;
  hp.hands = s.d_hands;
  hp.fvalid = nullptr;
  hp.labels = s.d_counts;
  hp.dbg = nullptr;
  hp.hl = nullptr;
  hp.radius = 0.0;
  hp.num_samples = n;
  reeval_kernel<<<n, 256, 0, stream>>>(hp);
  HIP_RET(hipGetLastError());
  rc = search_join(s, stream);
  if (rc) return rc;
  HIP_RET(hipMemcpyAsync(hands, s.d_hands, (size_t)n * sizeof(gpd_hand), hipMemcpyDeviceToHost, stream));
  HIP_RET(hipMemcpyAsync(s.h_counts.data(), s.d_counts, (size_t)n * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP_RET(hipStreamSynchronize(stream));
  for (int i = 0; i < n; i++) labels[i] = s.h_counts[8 * (size_t)i + 5];
  return GPD_OK;
}

int search_download(const gpd_params &p, SearchState &s, gpd_hand *hands, int *num_sets, hipStream_t stream) {
  const int slots = p.num_hand_axes * p.num_orientations;
  const int S = s.num_samples;
  std::vector<gpd_hand> tmp((size_t)S * slots);
  HIP_RET(hipMemcpyAsync(tmp.data(), s.d_hands, tmp.size() * sizeof(gpd_hand), hipMemcpyDeviceToHost, stream));
  HIP_RET(hipStreamSynchronize(stream));
  // drop samples without a frame neighbourhood before numbering sets (frame_estimator.cpp:24-29)
  s.h_set_sample.clear();
  s.h_samples.clear();
  int ns = 0;
  for (int i = 0; i < S; i++) {
    if (s.h_counts[8 * i + 2] == 0) continue;
    for (int j = 0; j < slots; j++) {
      gpd_hand h = tmp[(size_t)i * slots + j];
      h.set_index = ns;
      hands[(size_t)ns * slots + j] = h;
    }
    s.h_set_sample.push_back(i);
    for (int r = 0; r < 3; r++) s.h_samples.push_back(tmp[(size_t)i * slots].sample[r]);
    ns++;
  }
  *num_sets = ns;
  return GPD_OK;
}

}  // namespace gpd
