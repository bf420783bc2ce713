// Grasp-image generation on gfx950: one workgroup per candidate, image tiles in LDS.
//
// Replaces ImageGenerator::createImages / createImageList (descriptor/image_generator.cpp:17-99),
// ImageStrategy::{transformToUnitImage, findPointsInUnitImage, transformPointsToUnitImage,
// findCellIndices, createNormalsImage, createDepthImage, createShadowImage}
// (descriptor/image_strategy.cpp:32-243), Image{15,12,3}ChannelsStrategy::{createImage,
// calculateImage, calculateChannels} (image_15_channels_strategy.cpp:27-105,
// image_12_channels_strategy.cpp:27-86, image_3_channels_strategy.cpp:27-42) and
// HandSet::{calculateShadow, calculateShadowForCamera, shadowVoxelsToPoints, fastrand}
// (candidate/hand_set.cpp:118-283).
//
// The reference rasterises with per-pixel recurrences whose result depends on the order
// in which points hit a pixel (neighbour order; image_strategy.cpp:130-142, 166-174,
// 202-210).  Here every projection is a counting sort of the in-box points by pixel
// (LDS atomics, any order), after which the thread that owns a pixel sorts its short
// segment by neighbour rank and runs the recurrence sequentially — same arithmetic, same
// order, all lanes busy.  Shadow voxels (hand_set.cpp:202-233 hash set) become a bitset
// over the candidate's voxel AABB: set semantics for free, and walking the bits in index
// order is the lexicographic voxel order the oracle defines.
//
// The 33*N LCG draws of a hand set are regenerated per candidate by jump-ahead
// (hand_set.cpp:263-266 is an affine map mod 2^32), so no per-set voxel list is stored.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "gpd_internal.h"

namespace gpd {

#define HIP_RET(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GPD_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

constexpr int IMG_THREADS = 512;
constexpr int PLACE_CAP = 8192;  // in-box points / shadow voxels per candidate
constexpr int VDIM = 46;         // voxel AABB edge (box diagonal 0.1233 m / 3 mm + margins)
constexpr int VBITS = VDIM * VDIM * VDIM;
constexpr int VWORDS = (VBITS + 31) / 32;

struct ImgConsts {
  double vol_depth, vol_width, vol_height, half_od, dbl_h;
  int C, nproj, per;
  double view_point[3];
  double shadow_length, voxel, voxel_mult, rand_inv;
  int num_shadow;
  uint32_t stride_a, stride_c;  // LCG jump by IMG_THREADS * num_shadow draws
};
__constant__ ImgConsts c_img;

struct ImgParams {
  const float *nn;
  int cap;
  const double *centers;
  const gpd_hand *hands;  // one per candidate
  const int32_t *meta;    // [n][4]: sample slot, N_images, lcg offset lo, hi (hi < 0: no shadow)
  uint8_t *images;
  int32_t *status;
};

struct Box {
  double F[9];
  double sample[3];
  double bottom, center;
};

// ImageStrategy::transformToUnitImage / findPointsInUnitImage / transformPointsToUnitImage
// (image_strategy.cpp:32-90): rotate into the hand frame, strict box test, unit cube.
__device__ inline bool to_unit(const Box &B, double w0, double w1, double w2, double u[3]) {
  const ImgConsts &K = c_img;
  const double c0 = w0 - B.sample[0], c1 = w1 - B.sample[1], c2 = w2 - B.sample[2];
  const double t0 = B.F[0] * c0 + B.F[3] * c1 + B.F[6] * c2;
  const double t1 = B.F[1] * c0 + B.F[4] * c1 + B.F[7] * c2;
  const double t2 = B.F[2] * c0 + B.F[5] * c1 + B.F[8] * c2;
  if ((t0 > B.bottom) && (t0 < B.bottom + K.vol_depth) && (t1 > B.center - K.half_od) && (t1 < B.center + K.half_od) &&
      (t2 > -1.0 * K.vol_height) && (t2 < K.vol_height)) {
    u[0] = (t0 - B.bottom) / K.vol_depth;
    u[1] = (t1 - (B.center - K.half_od)) / K.vol_width;
    u[2] = (t2 + K.vol_height) / K.dbl_h;
    return true;
  }
  return false;
}

// ImageStrategy::findCellIndices (image_strategy.cpp:92-102)
__device__ inline int cell_index(double ua, double ub) {
  const double cellsize = 1.0 / (double)kImg;
  int v = (int)floor(ua / cellsize);
  int h = (int)floor(ub / cellsize);
  v = v < kImg - 1 ? v : kImg - 1;
  h = h < kImg - 1 ? h : kImg - 1;
  return h + v * kImg;
}

// projections by cumulative row swaps 0<->2 then 1<->2: (x,y,z), (z,y,x), (z,x,y)
__device__ inline void project(int pr, const double u[3], double &a, double &b, double &d) {
  if (pr == 0) {
    a = u[0]; b = u[1]; d = u[2];
  } else if (pr == 1) {
    a = u[2]; b = u[1]; d = u[0];
  } else {
    a = u[2]; b = u[0]; d = u[1];
  }
}

__device__ inline uint32_t lcg_step(uint32_t &s) {  // HandSet::fastrand (hand_set.cpp:263-266)
  s = 214013u * s + 2531011u;
  return (uint32_t)(((int32_t)s >> 16) & 0x7FFF);
}
__device__ inline uint32_t lcg_jump(uint32_t s, unsigned long long n) {
  uint32_t a = 214013u, c = 2531011u, A = 1u, Cc = 0u;
  while (n) {
    if (n & 1ull) {
      A = a * A;
      Cc = a * Cc + c;
    }
    c = (a + 1u) * c;
    a = a * a;
    n >>= 1;
  }
  return A * s + Cc;
}

template <class T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct Smem {
  uint32_t cells[kPix];        // (segment start << 16) | count
  float raster[kPix * 3];
  uint32_t place[PLACE_CAP];
  uint32_t bits[VWORDS];
  float red_f[2 * (IMG_THREADS / 64)];
  int red_i[IMG_THREADS / 64];
  int vorg[3];
  int flag;
};

// exclusive scan of cells[].count into the start field; returns the total
__device__ int scan_cells(Smem &S) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int PER = (kPix + IMG_THREADS - 1) / IMG_THREADS;  // 8
  int c[PER];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = tid * PER + k;
    c[k] = i < kPix ? (int)(S.cells[i] & 0xffffu) : 0;
    sum += c[k];
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  __syncthreads();
  if (lane == 63) S.red_i[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < IMG_THREADS / 64; w++) {
    if (w < wave) base += S.red_i[w];
    total += S.red_i[w];
  }
  int run = base + incl - sum;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = tid * PER + k;
    if (i < kPix) S.cells[i] = (uint32_t)(run < 0xffff ? run : 0xffff) << 16;
    run += c[k];
  }
  __syncthreads();
  return total;
}

__device__ inline void sort_segment(uint32_t *p, int n) {
  for (int i = 1; i < n; i++) {
    const uint32_t v = p[i];
    int j = i - 1;
    while (j >= 0 && p[j] > v) {
      p[j + 1] = p[j];
      j--;
    }
    p[j + 1] = v;
  }
}

// 3x3 rect max-dilate (border ignored), NORM_MINMAX to [0,1], u8 = round-half-even(v*255)
// (image_strategy.cpp:144-153, 178-187, 221-230; cv::dilate / cv::normalize / convertTo).
template <int NCH>
__device__ void finalize_channels(Smem &S, uint8_t *out, int C, int ch_off) {
  const int tid = threadIdx.x;
  constexpr int N = kPix * NCH;
  constexpr int PER = (N + IMG_THREADS - 1) / IMG_THREADS;
  float d[PER];
  float mn = FLT_MAX, mx = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int e = tid + k * IMG_THREADS;
    d[k] = 0.f;
    if (e < N) {
      const int pix = e / NCH, ch = e - pix * NCH;
      const int r = pix / kImg, c = pix - r * kImg;
      float m = -FLT_MAX;
#pragma unroll
      for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++) {
          const int rr = r + dr, cc = c + dc;
          if (rr >= 0 && rr < kImg && cc >= 0 && cc < kImg) m = fmaxf(m, S.raster[(rr * kImg + cc) * NCH + ch]);
        }
      d[k] = m;
      mn = fminf(mn, m);
      mx = fmaxf(mx, m);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    S.red_f[2 * (tid >> 6)] = mn;
    S.red_f[2 * (tid >> 6) + 1] = mx;
  }
  __syncthreads();
  mn = S.red_f[0];
  mx = S.red_f[1];
#pragma unroll
  for (int w = 1; w < IMG_THREADS / 64; w++) {
    mn = fminf(mn, S.red_f[2 * w]);
    mx = fmaxf(mx, S.red_f[2 * w + 1]);
  }
  const double smin = (double)mn, smax = (double)mx;
  const double scale = 1.0 * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
  const double shift = 0.0 - smin * scale;
  const float fs = (float)scale, fb = (float)shift;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int e = tid + k * IMG_THREADS;
    if (e < N) {
      const int pix = e / NCH, ch = e - pix * NCH;
      const float v = d[k] * fs + fb;
      const float u = v * 255.0f + 0.0f;
      float r = rintf(u);
      r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
      out[(size_t)pix * C + ch_off + ch] = (uint8_t)(int)r;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(IMG_THREADS) void grasp_image_kernel(ImgParams P) {
  __shared__ Smem S;
  const ImgConsts &K = c_img;
  const int cand = blockIdx.x;
  const int tid = threadIdx.x;
  const gpd_hand &H = P.hands[cand];
  const int slot_s = P.meta[4 * cand + 0];
  const int N = P.meta[4 * cand + 1];
  const uint32_t off_lo = (uint32_t)P.meta[4 * cand + 2];
  const int32_t off_hi = P.meta[4 * cand + 3];
  const float *nn = P.nn + (size_t)slot_s * 6 * P.cap;
  uint8_t *out = P.images + (size_t)cand * kPix * K.C;
  Box B;
#pragma unroll
  for (int i = 0; i < 9; i++) B.F[i] = H.frame[i];
#pragma unroll
  for (int i = 0; i < 3; i++) B.sample[i] = H.sample[i];
  B.bottom = H.bottom;
  B.center = H.center;
  if (tid == 0) S.flag = 0;

  // ---- shadow voxel bitset (15 channels): HandSet::calculateShadow for one camera ----
  const bool with_shadow = (K.C == 15);
  if (with_shadow) {
    for (int w = tid; w < VWORDS; w += IMG_THREADS) S.bits[w] = 0u;
    if (tid == 0) {
      // voxel AABB of the image box: corners sample + F * (bx, by, bz)
      double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
      for (int k = 0; k < 8; k++) {
        const double bx = (k & 1) ? B.bottom + K.vol_depth : B.bottom;
        const double by = (k & 2) ? B.center + K.half_od : B.center - K.half_od;
        const double bz = (k & 4) ? K.vol_height : -K.vol_height;
        for (int r = 0; r < 3; r++) {
          const double w = B.sample[r] + B.F[3 * r + 0] * bx + B.F[3 * r + 1] * by + B.F[3 * r + 2] * bz;
          lo[r] = fmin(lo[r], w);
        }
      }
      for (int r = 0; r < 3; r++) S.vorg[r] = (int)floor(lo[r] * K.voxel_mult) - 1;
    }
    __syncthreads();
    if (off_hi >= 0 && N > 0) {
      const int x0 = S.vorg[0], y0 = S.vorg[1], z0 = S.vorg[2];
      // shadow_vec = shadow_length * (center - view_point) / norm (hand_set.cpp:147-150)
      const double *cen = P.centers + 3 * (size_t)slot_s;
      double vec[3];
      for (int r = 0; r < 3; r++) vec[r] = cen[r] - K.view_point[r];
      const double nrm = sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
      for (int r = 0; r < 3; r++) vec[r] = K.shadow_length * vec[r] / nrm;
      const unsigned long long off = ((unsigned long long)(uint32_t)off_hi << 32) | off_lo;
      uint32_t state = lcg_jump(0u, off + (unsigned long long)tid * (unsigned)K.num_shadow);
      int bad = 0;
      for (int i = tid; i < N; i += IMG_THREADS) {
        const double p0 = (double)nn[0 * P.cap + i], p1 = (double)nn[1 * P.cap + i], p2 = (double)nn[2 * P.cap + i];
        uint32_t st = state;
        for (int k = 0; k < K.num_shadow; k++) {
          const double t = (double)(int)lcg_step(st) * K.rand_inv;
          const int vx = (int)((p0 + t * vec[0]) * K.voxel_mult);
          const int vy = (int)((p1 + t * vec[1]) * K.voxel_mult);
          const int vz = (int)((p2 + t * vec[2]) * K.voxel_mult);
          double u[3];
          if (to_unit(B, (double)vx * K.voxel, (double)vy * K.voxel, (double)vz * K.voxel, u)) {
            const int ix = vx - x0, iy = vy - y0, iz = vz - z0;
            if ((unsigned)ix < (unsigned)VDIM && (unsigned)iy < (unsigned)VDIM && (unsigned)iz < (unsigned)VDIM) {
              const int bit = (ix * VDIM + iy) * VDIM + iz;
              atomicOr(&S.bits[bit >> 5], 1u << (bit & 31));
            } else {
              bad = 1;
            }
          }
        }
        state = K.stride_a * state + K.stride_c;  // advance by IMG_THREADS * num_shadow draws
      }
      if (bad) atomicOr(&S.flag, 1);
    }
  }
  __syncthreads();

  for (int pr = 0; pr < K.nproj; pr++) {
    const int ch0 = pr * K.per;
    // ---- points: counting sort by pixel ----
    for (int c = tid; c < kPix; c += IMG_THREADS) S.cells[c] = 0u;
    __syncthreads();
    for (int i = tid; i < N; i += IMG_THREADS) {
      double u[3];
      if (to_unit(B, (double)nn[0 * P.cap + i], (double)nn[1 * P.cap + i], (double)nn[2 * P.cap + i], u)) {
        double a, b, d;
        project(pr, u, a, b, d);
        atomicAdd(&S.cells[cell_index(a, b)], 1u);
      }
    }
    __syncthreads();
    const int total = scan_cells(S);
    if (total > PLACE_CAP) {
      if (tid == 0) atomicOr(&S.flag, 2);
    }
    for (int i = tid; i < N; i += IMG_THREADS) {
      double u[3];
      if (to_unit(B, (double)nn[0 * P.cap + i], (double)nn[1 * P.cap + i], (double)nn[2 * P.cap + i], u)) {
        double a, b, d;
        project(pr, u, a, b, d);
        const uint32_t old = atomicAdd(&S.cells[cell_index(a, b)], 1u);
        const uint32_t slot = (old >> 16) + (old & 0xffffu);
        if (slot < PLACE_CAP) S.place[slot] = (uint32_t)i;
      }
    }
    __syncthreads();
    // ---- normals (image_strategy.cpp:124-156): pixel owner walks its segment in neighbour order
    for (int c = tid; c < kPix; c += IMG_THREADS) {
      const uint32_t w = S.cells[c];
      const int cnt = (int)(w & 0xffffu), start = (int)(w >> 16);
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (cnt > 0 && start + cnt <= PLACE_CAP) {
        sort_segment(&S.place[start], cnt);
        for (int e = 0; e < cnt; e++) {
          const int i = (int)S.place[start + e];
          const double n0 = (double)nn[3 * P.cap + i], n1 = (double)nn[4 * P.cap + i], n2 = (double)nn[5 * P.cap + i];
          const float a0 = (float)fabs(B.F[0] * n0 + B.F[3] * n1 + B.F[6] * n2);
          const float a1 = (float)fabs(B.F[1] * n0 + B.F[4] * n1 + B.F[7] * n2);
          const float a2 = (float)fabs(B.F[2] * n0 + B.F[5] * n1 + B.F[8] * n2);
          if (v0 == 0.f && v1 == 0.f && v2 == 0.f) {
            v0 = a0;
            v1 = a1;
            v2 = a2;
          } else {
            const float s = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
            const double inv = 1.0 / (double)s;
            const float d0 = a0 - v0, d1 = a1 - v1, d2 = a2 - v2;
            v0 = v0 + (float)((double)d0 * inv);
            v1 = v1 + (float)((double)d1 * inv);
            v2 = v2 + (float)((double)d2 * inv);
          }
        }
      }
      const int row = kImg - 1 - c / kImg, col = c % kImg;
      float *px = &S.raster[(row * kImg + col) * 3];
      px[0] = v0;
      px[1] = v1;
      px[2] = v2;
    }
    __syncthreads();
    finalize_channels<3>(S, out, K.C, ch0);
    // ---- depth (image_strategy.cpp:158-190)
    if (K.C >= 12) {
      for (int c = tid; c < kPix; c += IMG_THREADS) {
        const uint32_t w = S.cells[c];
        const int cnt = (int)(w & 0xffffu), start = (int)(w >> 16);
        float pix = 0.f;
        if (cnt > 0 && start + cnt <= PLACE_CAP) {
          float avg = 0.f, cn = 0.f;
          for (int e = 0; e < cnt; e++) {
            const int i = (int)S.place[start + e];
            double u[3], a, b, d;
            to_unit(B, (double)nn[0 * P.cap + i], (double)nn[1 * P.cap + i], (double)nn[2 * P.cap + i], u);
            project(pr, u, a, b, d);
            cn = (float)((double)cn + 1.0);
            avg = (float)((double)avg + (d - (double)avg) * (1.0 / (double)cn));
          }
          pix = (float)(1.0 - (double)avg);
        }
        const int row = kImg - 1 - c / kImg, col = c % kImg;
        S.raster[row * kImg + col] = pix;
      }
      __syncthreads();
      finalize_channels<1>(S, out, K.C, ch0 + 3);
    }
    // ---- shadow (image_strategy.cpp:192-233) over the voxel bitset in index order
    if (with_shadow) {
      const int x0 = S.vorg[0], y0 = S.vorg[1], z0 = S.vorg[2];
      for (int c = tid; c < kPix; c += IMG_THREADS) S.cells[c] = 0u;
      __syncthreads();
      for (int wd = tid; wd < VWORDS; wd += IMG_THREADS) {
        uint32_t bits = S.bits[wd];
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          const int lin = wd * 32 + b;
          const int ix = lin / (VDIM * VDIM), iy = (lin / VDIM) % VDIM, iz = lin % VDIM;
          double u[3], a, bb, d;
          to_unit(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, u);
          project(pr, u, a, bb, d);
          atomicAdd(&S.cells[cell_index(a, bb)], 1u);
        }
      }
      __syncthreads();
      const int tot_s = scan_cells(S);
      if (tot_s > PLACE_CAP) {
        if (tid == 0) atomicOr(&S.flag, 4);
      }
      for (int wd = tid; wd < VWORDS; wd += IMG_THREADS) {
        uint32_t bits = S.bits[wd];
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          const int lin = wd * 32 + b;
          const int ix = lin / (VDIM * VDIM), iy = (lin / VDIM) % VDIM, iz = lin % VDIM;
          double u[3], a, bb, d;
          to_unit(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, u);
          project(pr, u, a, bb, d);
          const uint32_t old = atomicAdd(&S.cells[cell_index(a, bb)], 1u);
          const uint32_t slot = (old >> 16) + (old & 0xffffu);
          if (slot < PLACE_CAP) S.place[slot] = (uint32_t)lin;
        }
      }
      __syncthreads();
      float lmax = -FLT_MAX;
      int lany = 0;
      for (int c = tid; c < kPix; c += IMG_THREADS) {
        const uint32_t w = S.cells[c];
        const int cnt = (int)(w & 0xffffu), start = (int)(w >> 16);
        float v = 0.f;
        if (cnt > 0 && start + cnt <= PLACE_CAP) {
          sort_segment(&S.place[start], cnt);
          float cn = 0.f;
          for (int e = 0; e < cnt; e++) {
            const int lin = (int)S.place[start + e];
            const int ix = lin / (VDIM * VDIM), iy = (lin / VDIM) % VDIM, iz = lin % VDIM;
            double u[3], a, bb, d;
            to_unit(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, u);
            project(pr, u, a, bb, d);
            cn = (float)((double)cn + 1.0);
            v = (float)((double)v + (d - (double)v) * (1.0 / (double)cn));
          }
          lmax = fmaxf(lmax, v);
          lany = 1;
        }
        const int row = kImg - 1 - c / kImg, col = c % kImg;
        S.raster[row * kImg + col] = v;
        S.raster[kPix + row * kImg + col] = cnt > 0 ? 1.f : 0.f;  // nonzero mask
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        lany |= __shfl_xor(lany, o);
      }
      __syncthreads();
      if ((tid & 63) == 0) {
        S.red_f[tid >> 6] = lmax;
        S.red_i[tid >> 6] = lany;
      }
      __syncthreads();
      float gmax = -FLT_MAX;
      int gany = 0;
#pragma unroll
      for (int w = 0; w < IMG_THREADS / 64; w++) {
        gmax = fmaxf(gmax, S.red_f[w]);
        gany |= S.red_i[w];
      }
      // minMaxLoc with mask -> max (0 if the mask is empty); image = max_img - image
      const double mxd = gany ? (double)gmax : 0.0;
      __syncthreads();
      for (int p = tid; p < kPix; p += IMG_THREADS) {
        const float m = S.raster[kPix + p] != 0.f ? (float)mxd : 0.0f;
        S.raster[p] = m - S.raster[p];
      }
      __syncthreads();
      finalize_channels<1>(S, out, K.C, ch0 + 4);
    }
  }
  if (tid == 0 && S.flag) atomicOr(P.status, S.flag);
}

// ---------------------------------------------------------------------------
void images_free(ImageState &im) {
  void *ptrs[] = {im.d_images, im.d_hands, im.d_cand_meta, im.d_status};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  im = ImageState();
}

int images_run(const gpd_params &p, const Cloud &c, const SearchState &s, ImageState &im, const gpd_hand *hands, int num_sets,
               int32_t *cand_index, hipStream_t stream) {
  const int slots = p.num_hand_axes * p.num_orientations;
  const int C = p.image_num_channels;
  if (s.cloud_generation != c.generation || s.num_samples == 0) {
    set_error("images: hands must come from gpd_hip_search on this context and cloud");
    return GPD_ERR_STATE;
  }
  if (C == 15 && c.num_cams != 1) {
    set_error("images: 15-channel shadow supports one camera (got %d)", c.num_cams);
    return GPD_ERR_INVALID;
  }
  if (num_sets > (int)s.h_set_sample.size()) {
    set_error("images: %d sets passed, search produced %zu", num_sets, s.h_set_sample.size());
    return GPD_ERR_INVALID;
  }
  // candidate list in set-major, slot-minor order; LCG offsets over live sets
  std::vector<gpd_hand> cand;
  std::vector<int32_t> meta;
  unsigned long long lcg = 0;
  im.stat_sets = 0;
  im.stat_sum_set_ni = 0;
  im.stat_sum_cand_ni = 0;
  for (int si = 0; si < num_sets; si++) {
    int nv = 0;
    for (int j = 0; j < slots; j++) nv += hands[(size_t)si * slots + j].valid ? 1 : 0;
    if (!nv) continue;
    const int samp = s.h_set_sample[si];
    const gpd_hand &h0 = hands[(size_t)si * slots];
    for (int r = 0; r < 3; r++)
      if (h0.sample[r] != s.h_samples[3 * (size_t)si + r]) {
        set_error("images: set %d does not match the last search (sample moved)", si);
        return GPD_ERR_STATE;
      }
    const int Ni = s.h_counts[8 * samp + 1];
    const bool seen = s.h_counts[8 * samp + 4] != 0;
    for (int j = 0; j < slots; j++) {
      const gpd_hand &h = hands[(size_t)si * slots + j];
      if (!h.valid) continue;
      if (cand_index) cand_index[cand.size()] = si * slots + j;
      cand.push_back(h);
      meta.push_back(samp);
      meta.push_back(Ni);
      meta.push_back((int32_t)(uint32_t)(lcg & 0xffffffffull));
      meta.push_back(seen ? (int32_t)(lcg >> 32) : -1);
    }
    if (C == 15 && seen) lcg += (unsigned long long)Ni * 33ull;
    im.stat_sets++;
    im.stat_sum_set_ni += Ni;
    im.stat_sum_cand_ni += (long long)Ni * nv;
  }
  const int n = (int)cand.size();
  im.num_candidates = n;
  if (n == 0) return GPD_OK;
  if (n > im.capacity) {
    if (im.d_images) (void)hipFree(im.d_images);
    if (im.d_hands) (void)hipFree(im.d_hands);
    if (im.d_cand_meta) (void)hipFree(im.d_cand_meta);
    im.d_images = nullptr;
    im.d_hands = nullptr;
    im.d_cand_meta = nullptr;
    im.capacity = 0;
    HIP_RET(hipMalloc(&im.d_images, (size_t)n * kPix * C));
    HIP_RET(hipMalloc(&im.d_hands, (size_t)n * sizeof(gpd_hand)));
    HIP_RET(hipMalloc(&im.d_cand_meta, (size_t)n * 4 * sizeof(int32_t)));
    im.capacity = n;
  }
  if (!im.d_status) HIP_RET(hipMalloc(&im.d_status, sizeof(int32_t)));
  HIP_RET(hipMemsetAsync(im.d_status, 0, sizeof(int32_t), stream));
  HIP_RET(hipMemcpyAsync(im.d_hands, cand.data(), (size_t)n * sizeof(gpd_hand), hipMemcpyHostToDevice, stream));
  HIP_RET(hipMemcpyAsync(im.d_cand_meta, meta.data(), (size_t)n * 4 * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  ImgConsts k;
  std::memset(&k, 0, sizeof(k));
  k.vol_depth = p.volume_depth;
  k.vol_width = p.volume_width;
  k.vol_height = p.volume_height;
  k.half_od = p.volume_width / 2.0;
  k.dbl_h = 2.0 * p.volume_height;
  k.C = C;
  k.nproj = (C == 3) ? 1 : 3;
  k.per = (C == 15) ? 5 : (C == 12 ? 4 : 3);
  for (int r = 0; r < 3; r++) k.view_point[r] = c.view_points[r];
  // shadow_length_ = max(volume_depth, volume_height/2, volume_width) (image_15_channels_strategy.h:70-75)
  k.shadow_length = std::fmax(std::fmax(p.volume_depth, p.volume_height / 2.0), p.volume_width);
  k.voxel = 0.003;
  k.voxel_mult = 1.0 / 0.003;
  k.rand_inv = 1.0 / 32767.0;
  k.num_shadow = (int)std::floor(k.shadow_length / k.voxel);
  {  // affine map of IMG_THREADS * num_shadow LCG steps
    uint32_t a = 214013u, cc = 2531011u, A = 1u, Cc = 0u;
    unsigned long long nsteps = (unsigned long long)IMG_THREADS * (unsigned)k.num_shadow;
    while (nsteps) {
      if (nsteps & 1ull) {
        A = a * A;
        Cc = a * Cc + cc;
      }
      cc = (a + 1u) * cc;
      a = a * a;
      nsteps >>= 1;
    }
    k.stride_a = A;
    k.stride_c = Cc;
  }
  HIP_RET(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_img), &k, sizeof(k), 0, hipMemcpyHostToDevice, stream));
  return images_launch(s, im, stream, true);
}

// Launches grasp_image_kernel over the candidate list resident on the device.
int images_launch(const SearchState &s, ImageState &im, hipStream_t stream, bool check) {
  const int n = im.num_candidates;
  if (n <= 0) return GPD_OK;
  ImgParams ip;
  ip.nn = s.d_nn;
  ip.cap = s.nn_cap;
  ip.centers = s.d_centers;
  ip.hands = im.d_hands;
  ip.meta = im.d_cand_meta;
  ip.images = im.d_images;
  ip.status = im.d_status;
  grasp_image_kernel<<<n, IMG_THREADS, 0, stream>>>(ip);
  HIP_RET(hipGetLastError());
  if (!check) return GPD_OK;
  int32_t status = 0;
  HIP_RET(hipMemcpyAsync(&status, im.d_status, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP_RET(hipStreamSynchronize(stream));
  if (status) {
    set_error("images: kernel capacity exceeded (flags %d: 1 voxel AABB, 2 in-box points > %d, 4 shadow voxels > %d)", status,
              PLACE_CAP, PLACE_CAP);
    return GPD_ERR_CAPACITY;
  }
  return GPD_OK;
}

}  // namespace gpd
