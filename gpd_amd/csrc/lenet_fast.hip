// LeNet forward on the fast matrix pipes of gfx950 — the default scoring path since round 5.
//
// Replaces EigenClassifier::forward (net/eigen_classifier.cpp:81-128), ConvLayer (net/conv_layer.cpp:26-98: im2col + GEMM,
// batch 1) and DenseLayer (net/dense_layer.cpp:6-15).  lenet.hip keeps the f32-input MFMA kernels whose every dot product is
// the oracle's k-ascending fmaf chain (bitwise); they run at 1/16 of the chip's bf16 matrix rate and 1/32 of its int8 rate.
// The reference itself fixes no summation order (Eigen's GEMM) and BASELINE's bar is |score - reference| <= 1e-4, so the
// kernels here compute the same sums at f32 accuracy or better on the fast pipes, by EXACT operand splitting:
//
//   conv1   the inputs are u8 grasp images — exact in int8 after x - 128.  Every f32 weight becomes a 32-bit fixed-point
//           integer per filter (W = round(w * 2^s), |W| <= 2^30: every weight within 2^8 of the filter's largest keeps all 24
//           bits, smaller ones are rounded at 2^-31 of the largest) cut into four balanced base-256 digits; the four digit planes
//           are four rows of an int8 MFMA tile (v_mfma_i32_16x16x64_i8, 4.4 POPS), the products and the sums are integers —
//           the dot product is EXACT — and the result is rounded to f32 once.
//   conv2,  f32 activations x f32 weights: each operand is the sum of three bf16 pieces (round to nearest, residual, again:
//   ip1     a = h + m + l to 2^-24 and better); h*h + h*m + m*h + h*l + l*h + m*m on v_mfma_f32_16x16x32_bf16 with f32
//           accumulation: every product exact, what is dropped (m*l, l*m, l*l) is below 2^-24 of the product.
//   ip2     two 500-long f32 chains, as before (lenet.hip fc2_score_kernel).
//
// Measured against float64 on the reference pins the scores of this path are closer than the f32 chain's
// (tests/test_gpu_lenet_fast.py), and within 1e-4 of the reference's plain-float Eigen path on every pin.
#include "gpd_internal.h"

#include <cstring>

namespace gpd {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// conv1 + pool1 on the int8 matrix pipe.
//
// GEMM view per image: D[(filter, digit)][pixel] = sum over (tap, channel) of digit(W[filter][channel][tap]) * (x - 128).
//   A operand (weights, register resident: 7 k-steps x 5 row tiles x 16 bytes per lane = 140 VGPRs):
//     row i of tile mt = digit (i & 3) of filter 4 mt + (i >> 2); the 64 k of a step = 4 taps (one per lane group) x 16 channels
//   B operand (pixels): the image sits in LDS pixel-major, 16 bytes per pixel (channels 0..C-1, rest padding), so the
//     fragment of a k-step is ONE ds_read_b128 per lane, no conversion: lane (pixel m = l & 15, tap slot g = l >> 4)
//   D (lane l, register r): pixel l & 15, row 4 (l >> 4) + r  ->  the four digits of ONE filter sit in the lane's four registers
// A tile is 8 x 2 conv pixels = four 2x2 pool windows, pixel m = 4 window + position: the pool is a max over a lane quad.
// LDS geometry: row pitch 72 pixels (= 8 mod 16) and the taps of lane groups (0, 1) and (2, 3) two image rows apart make every
// ds_read_b128 of the k-steps 0..4 bank-conflict free (MI355X_MICROARCH.md §LDS: 16-lane groups {0-3, 12-15, 20-27}, ...).
// Tap slots: k-step s < 5: column kx = s, rows ky = 0, 2, 1, 3 for the lane groups; step 5: (4, 0..3); step 6: (4, 4) + 3 empty.
// Epilogue: S = ((D3 * 256 + D2) * 256 + D1) * 256 + D0 exactly (f64), max over the window, + 128 * sum(W) (the x - 128
// shift), one rounding to f32, * 2^-s (v_ldexp_f32: exact), + bias.  max(a_i) + b == max(a_i + b): rounding is monotone.
// Persistent workgroups, one per CU: the next image streams into a raw LDS region one 16-byte piece per thread and tile,
// and is turned pixel-major between two barriers.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F1_THREADS = 512, F1_WAVES = 8;
constexpr int F1_PITCH = 72;                  // pixels per LDS row
constexpr int F1_ROWB = F1_PITCH * 16;        // 1152 bytes
constexpr int F1_HWC = kImg * F1_ROWB;        // 69120 bytes
constexpr int F1_TILES = 28 * 7;              // 28 row pairs x 7 tiles of 8 columns
constexpr int F1_KS = 7, F1_MT = 5;

// tap (ky * 5 + kx) of lane group g in k-step ks; -1: empty slot (zero weights, reads the slot of group 0)
__host__ __device__ constexpr int f1_tap(int ks, int g) {
  return ks < 5 ? ((g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 1 : 3) * 5 + ks) : ks == 5 ? 20 + g : (g == 0 ? 24 : -1);
}

__device__ inline double max_f64(double a, double b) {
  // v_max_f64 as is (fmax() adds a canonicalising max in front: the operands here are exact integers, never NaN)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline double dpp_quad_max(double v) {
  // max over the four lanes of a quad (quad_perm [1,0,3,2], then [2,3,0,1])
  int lo = __double2loint(v), hi = __double2hiint(v);
  double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true));
  v = max_f64(v, o);
  lo = __double2loint(v);
  hi = __double2hiint(v);
  o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true));
  return max_f64(v, o);
}

template <int C>
__global__ __launch_bounds__(F1_THREADS) void conv1_i8_kernel(const uint8_t *__restrict__ images, const uint4 *__restrict__ atab,
                                                              const double *__restrict__ corr, const int *__restrict__ shift,
                                                              const float *__restrict__ bias, float *__restrict__ pool1, int n,
                                                              int *__restrict__ queue) {
  constexpr int RAW = C * kPix, NV = RAW / 16;
  static_assert(RAW % 16 == 0 && C <= 16, "image bytes");
  __shared__ __attribute__((aligned(16))) uint8_t s_hwc[F1_HWC];
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[RAW];
  __shared__ int s_nxt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  int img = blockIdx.x;
  if (img >= n) return;
  // the weight fragments: resident for the whole launch
  i32x4 A[F1_KS][F1_MT];
#pragma unroll
  for (int ks = 0; ks < F1_KS; ks++)
#pragma unroll
    for (int mt = 0; mt < F1_MT; mt++) {
      const uint4 v = atab[(ks * F1_MT + mt) * 64 + lane];
      A[ks][mt] = i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  {  // the first image, by everybody
    const uint4 *src = reinterpret_cast<const uint4 *>(images + (size_t)img * RAW);
    uint4 *dst = reinterpret_cast<uint4 *>(s_raw);
    for (int i = tid; i < NV; i += F1_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  // raw planar [C][60][60] -> pixel-major [60][72][16] with x ^ 0x80 (u8 -> s8 of x - 128): a task is four consecutive pixels
  auto transpose = [&]() {
    for (int task = tid; task < kImg * (kImg / 4); task += F1_THREADS) {
      const int y = task / (kImg / 4), x0 = 4 * (task - y * (kImg / 4));
      uint32_t r[16];
#pragma unroll
      for (int c = 0; c < 16; c++) r[c] = c < C ? *reinterpret_cast<const uint32_t *>(s_raw + c * kPix + y * kImg + x0) : 0u;
      uint32_t o[4][4];  // [pixel][channel group]
#pragma unroll
      for (int cg = 0; cg < 4; cg++) {
        // 4 x 4 byte transpose: (channel, pixel) -> (pixel, channel)
        const uint32_t t0 = __builtin_amdgcn_perm(r[4 * cg + 1], r[4 * cg], 0x05010400u);      // c0.p0 c1.p0 c0.p1 c1.p1
        const uint32_t t1 = __builtin_amdgcn_perm(r[4 * cg + 1], r[4 * cg], 0x07030602u);      // c0.p2 c1.p2 c0.p3 c1.p3
        const uint32_t t2 = __builtin_amdgcn_perm(r[4 * cg + 3], r[4 * cg + 2], 0x05010400u);  // c2.p0 c3.p0 c2.p1 c3.p1
        const uint32_t t3 = __builtin_amdgcn_perm(r[4 * cg + 3], r[4 * cg + 2], 0x07030602u);
        o[0][cg] = __builtin_amdgcn_perm(t2, t0, 0x05040100u) ^ 0x80808080u;
        o[1][cg] = __builtin_amdgcn_perm(t2, t0, 0x07060302u) ^ 0x80808080u;
        o[2][cg] = __builtin_amdgcn_perm(t3, t1, 0x05040100u) ^ 0x80808080u;
        o[3][cg] = __builtin_amdgcn_perm(t3, t1, 0x07060302u) ^ 0x80808080u;
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
        *reinterpret_cast<uint4 *>(s_hwc + (y * F1_PITCH + x0 + e) * 16) = make_uint4(o[e][0], o[e][1], o[e][2], o[e][3]);
    }
  };
  transpose();
  // the lane's pixel inside a tile and its tap rows
  const int m_row = (j >> 1) & 1, m_x = 2 * (j >> 2) + (j & 1);
  const int kyg = q == 0 ? 0 : q == 1 ? 2 : q == 2 ? 1 : 3;
  const int lane_off = (m_row * F1_PITCH + m_x) * 16;
  const int p = j & 3, w = j >> 2;
  const double k_corr_own = corr[4 * p + q], k_corr_4 = corr[16 + q];
  const int k_shift_own = shift[4 * p + q], k_shift_4 = shift[16 + q];  // the filter's fixed-point position s: value = integer * 2^-s
  const float k_bias_own = bias[4 * p + q], k_bias_4 = bias[16 + q];
  for (;;) {
    if (tid == 0) s_nxt = (int)gridDim.x + atomicAdd(queue, 1);
    __syncthreads();  // the pixel-major image is complete, the raw region is free, s_nxt is visible
    const int nxt = s_nxt;
    const uint4 *nsrc = reinterpret_cast<const uint4 *>(images + (size_t)(nxt < n ? nxt : img) * RAW);
    int it = 0;
    for (int t = wave; t < F1_TILES; t += F1_WAVES, it++) {
      // one 16-byte piece of the next image per thread and tile: requested now, stored after the tile
      const int piece = it * F1_THREADS + tid;
      const bool has_piece = nxt < n && piece < NV;
      uint4 stage = make_uint4(0, 0, 0, 0);
      if (has_piece) stage = nsrc[piece];
      const int trow = t / 7, tcol = t - 7 * trow;
      const uint8_t *base = s_hwc + ((2 * trow) * F1_PITCH + 8 * tcol) * 16 + lane_off;
      const uint8_t *pa = base + kyg * F1_ROWB;
      const uint8_t *pb = base + 4 * F1_ROWB + q * 16;
      const uint8_t *pc = base + 4 * F1_ROWB + 4 * 16;
      i32x4 B[F1_KS];
#pragma unroll
      for (int ks = 0; ks < 5; ks++) B[ks] = *reinterpret_cast<const i32x4 *>(pa + ks * 16);
      B[5] = *reinterpret_cast<const i32x4 *>(pb);
      B[6] = *reinterpret_cast<const i32x4 *>(pc);
      i32x4 acc[F1_MT];
#pragma unroll
      for (int mt = 0; mt < F1_MT; mt++) acc[mt] = i32x4{0, 0, 0, 0};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < F1_KS; ks++)
#pragma unroll
        for (int mt = 0; mt < F1_MT; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][mt], B[ks], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // the 35 MFMAs stay one run: the SIMD's other wave has its epilogue meanwhile
      // epilogue: exact sum of the digit planes (f64), pool over the quad, one rounding.  The quad's four lanes end up with the
      // same pooled sums: lane p finishes tile p (filter 4 p + q), and every lane tile 4 (filter 16 + q, stored by p = 0)
      double s[F1_MT];
#pragma unroll
      for (int mt = 0; mt < F1_MT; mt++) {
        const int hi = (int)((unsigned)acc[mt][3] << 8) + acc[mt][2];  // |.| < 2^31: the top digit is within +-64
        const int lo = (int)((unsigned)acc[mt][1] << 8) + acc[mt][0];
        s[mt] = dpp_quad_max(__builtin_fma((double)hi, 65536.0, (double)lo));
      }
      const double s_own = p == 0 ? s[0] : p == 1 ? s[1] : p == 2 ? s[2] : s[3];
      const float v_own = ldexpf((float)(s_own + k_corr_own), -k_shift_own) + k_bias_own;
      const float v_4 = ldexpf((float)(s[4] + k_corr_4), -k_shift_4) + k_bias_4;
      float *dst = pool1 + ((size_t)img * 784 + trow * 28 + 4 * tcol + w) * 20;
      dst[4 * p + q] = v_own;
      if (p == 0) dst[16 + q] = v_4;
      if (has_piece) reinterpret_cast<uint4 *>(s_raw)[piece] = stage;
    }
    __syncthreads();  // every tile of the image is done, the next image's raw bytes are in LDS
    if (nxt >= n) break;
    transpose();
    img = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 pieces: a = h + m + l with h = bf16(a), m = bf16(a - h), l = bf16(a - h - m) (round to nearest even, v_cvt_pk_bf16_f32;
// the residuals are exact in f32).  |a - (h + m + l)| <= 2^-25 |a| or so; the three pieces of an f32 with 24 significant
// bits reproduce it exactly unless a piece underflows.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline bf16x8 as_bf16x8(const uint4 &v) {
  bf16x8 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}

struct Bf3 {
  unsigned short h, m, l;
};
__host__ __device__ inline unsigned short bf16_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __bf16 b = (__bf16)x;
  unsigned short u;
  __builtin_memcpy(&u, &b, 2);
  return u;
#else
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));  // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
#endif
}
__host__ __device__ inline float bf16_value(unsigned short b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_memcpy(&f, &u, 4);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}
__host__ __device__ inline Bf3 bf16_split3(float a) {
  Bf3 r;
  r.h = bf16_bits(a);
  const float r1 = a - bf16_value(r.h);
  r.m = bf16_bits(r1);
  const float r2 = r1 - bf16_value(r.m);
  r.l = bf16_bits(r2);
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// conv2 + pool2 on the bf16 matrix pipe: D[pixel][filter] = sum over the six piece products, k = (tap, channel).
//   v_mfma_f32_16x16x32_bf16: A (pixels): lane (pixel m = l & 15, k group g = l >> 4) holds 8 k = two 4-channel groups of
//   8 bytes; B (weights): lane (filter l & 15, k group g); D: lane (filter l & 15), register r = pixel 4 (l >> 4) + r — a tile
//   is 8 x 2 conv pixels = four pool windows, pixel m = 4 window + position, so the pool is a max over the lane's registers.
// Four waves per workgroup, ONE per SIMD, 512 registers each: wave (np, half) keeps the three bf16 planes of 32 filters
// (2 column tiles x 3 pieces x 16 k-steps x 4 VGPRs = 384 registers) for the whole launch and walks the 18 pixel tiles of its
// half of the image; the activations of a k-step (3 pieces x 2 reads of 8 bytes) feed 12 MFMAs.
// LDS: the image as bf16 pieces [row 28][piece 3][column 28][channel 20] (94 080 B) + the next image's raw f32 rows (62 720 B),
// which arrive one 16-byte piece per thread and tile and are split between two barriers.
// The 128 four-channel k-slots of the 16 k-steps hold the 125 (tap, channel group) pairs so that the two lane groups of a
// ds_read_b64 half read neighbouring columns or rows of ONE channel group (overlapping addresses broadcast: conflict free):
//   entry e = ks + 16 h: e < 25: ky = e / 5, cg = e % 5, kx = g       e in 25..29: cg = e - 25, kx = 4, ky = g
//                        e = 30: ky = kx = 4, cg = g                  e = 31: ky = kx = 4, cg = 4 (lane group 0 only)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F2_THREADS = 256;
constexpr int F2_PP = 28 * 40;         // bytes of one piece row: 28 pixels x 20 bf16
constexpr int F2_RS = 3 * F2_PP;       // bytes of one image row (three pieces)
constexpr int F2_IMG = 28 * F2_RS;     // 94080
constexpr int F2_RAW = 784 * 20 * 4;   // 62720
constexpr int F2_XLD = 7232;           // row length of the flat bf16 planes: 7200 + 32 zeros (ip1 walks K in steps of 64)

// (tap = ky * 5 + kx, channel group) of k-slot entry e for lane group g; tap -1: empty
__host__ __device__ constexpr int f2_slot_tap(int e, int g) {
  return e < 25 ? (e / 5) * 5 + g : e < 30 ? g * 5 + 4 : e == 30 ? 24 : (g == 0 ? 24 : -1);
}
__host__ __device__ constexpr int f2_slot_cg(int e, int g) { return e < 25 ? e % 5 : e < 30 ? e - 25 : e == 30 ? g : 4; }

__global__ __launch_bounds__(F2_THREADS) void conv2_bf16_kernel(const float *__restrict__ pool1, const uint4 *__restrict__ btab,
                                                                const float *__restrict__ bias, unsigned short *__restrict__ xs,
                                                                size_t xs_plane, int n, int *__restrict__ queue) {
  __shared__ __attribute__((aligned(16))) uint8_t s_img[F2_IMG];
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[F2_RAW];
  __shared__ int s_nxt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int np = wave & 1, half = wave >> 1;
  const int j = lane & 15, q = lane >> 4;
  int img = blockIdx.x;
  if (img >= n) return;
  // the weight fragments of the wave's 32 filters: [column tile][piece][k-step]
  bf16x8 W[2][3][16];
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        W[nt][pc][ks] = as_bf16x8(btab[(((np * 2 + nt) * 3 + pc) * 16 + ks) * 64 + lane]);
      }
  {  // the first image, by everybody
    const uint4 *src = reinterpret_cast<const uint4 *>(pool1 + (size_t)img * (784 * 20));
    uint4 *dst = reinterpret_cast<uint4 *>(s_raw);
    for (int i = tid; i < F2_RAW / 16; i += F2_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  // raw f32 [pixel][20] -> bf16 pieces [row][piece][column][20]; a task is one pixel's four-channel group
  auto split = [&]() {
    for (int task = tid; task < 784 * 5; task += F2_THREADS) {
      const int px = task / 5, cg = task - 5 * px;
      const int row = px / 28, col = px - 28 * row;
      const float4 a = *reinterpret_cast<const float4 *>(s_raw + px * 80 + cg * 16);
      const Bf3 s0 = bf16_split3(a.x), s1 = bf16_split3(a.y), s2 = bf16_split3(a.z), s3 = bf16_split3(a.w);
      uint8_t *d = s_img + row * F2_RS + col * 40 + cg * 8;
      *reinterpret_cast<uint2 *>(d) = make_uint2(s0.h | ((uint32_t)s1.h << 16), s2.h | ((uint32_t)s3.h << 16));
      *reinterpret_cast<uint2 *>(d + F2_PP) = make_uint2(s0.m | ((uint32_t)s1.m << 16), s2.m | ((uint32_t)s3.m << 16));
      *reinterpret_cast<uint2 *>(d + 2 * F2_PP) = make_uint2(s0.l | ((uint32_t)s1.l << 16), s2.l | ((uint32_t)s3.l << 16));
    }
  };
  split();
  const int m_row = (j >> 1) & 1, m_x = 2 * (j >> 2) + (j & 1);
  const int lane_off = m_row * F2_RS + m_x * 40;
  // the four address patterns of the k-slot table (see above): + 40 g (columns), + g rows, + 8 g (channel groups), none
  const int off_x = lane_off + 40 * q, off_y = lane_off + q * F2_RS + 4 * 40, off_z = lane_off + 4 * F2_RS + 4 * 40 + 8 * q,
            off_w = lane_off + 4 * F2_RS + 4 * 40 + 32;
  const float k_bias0 = (32 * np + j) < 50 ? bias[32 * np + j] : 0.f, k_bias1 = (32 * np + 16 + j) < 50 ? bias[32 * np + 16 + j] : 0.f;
  for (;;) {
    if (tid == 0) s_nxt = (int)gridDim.x + atomicAdd(queue, 1);
    __syncthreads();  // the pieces are complete, the raw region is free, s_nxt is visible
    const int nxt = s_nxt;
    const uint4 *nsrc = reinterpret_cast<const uint4 *>(pool1 + (size_t)(nxt < n ? nxt : img) * (784 * 20));
    for (int tt = 0; tt < 18; tt++) {
      const int piece = tt * F2_THREADS + tid;
      const bool has_piece = nxt < n && piece < F2_RAW / 16;
      uint4 stage = make_uint4(0, 0, 0, 0);
      if (has_piece) stage = nsrc[piece];
      const int T = half * 18 + tt, rp = T / 3, xt = T - 3 * rp;
      const uint8_t *base = s_img + (2 * rp) * F2_RS + (8 * xt) * 40;
      // the two 8-byte reads of k-step ks for piece pc
      auto frag = [&](int ks, int pc) -> bf16x8 {
        uint2 v[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int e = ks + 16 * h;
          const uint8_t *a = e < 25 ? base + off_x + (e / 5) * F2_RS + (e % 5) * 8
                             : e < 30 ? base + off_y + (e - 25) * 8
                             : e == 30 ? base + off_z
                                       : base + off_w;
          // (volatile: keeps the two halves two ds_read_b64 — merged into ds_read2_b64 they run at half the LDS rate and on
          //  the 32-bank rule, and their results have to be re-sorted into the operand registers)
          typedef const volatile __attribute__((address_space(3))) unsigned long long *lds_u64;
          const unsigned long long t = *(lds_u64)(a + pc * F2_PP);
          v[h] = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
        }
        return as_bf16x8(make_uint4(v[0].x, v[0].y, v[1].x, v[1].y));
      };
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      bf16x8 a_buf[2][3];  // the fragments of k-step ks live in a_buf[ks & 1]: the next step's are requested before this step's MFMAs
#pragma unroll
      for (int pc = 0; pc < 3; pc++) a_buf[0][pc] = frag(0, pc);
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        if (ks + 1 < 16) {
#pragma unroll
          for (int pc = 0; pc < 3; pc++) a_buf[(ks + 1) & 1][pc] = frag(ks + 1, pc);
        }
        // small terms first; the two column tiles alternate so that dependent MFMAs are two issues apart
#pragma unroll
        for (int term = 0; term < 6; term++) {
          // (activation piece, weight piece): l*h, h*l, m*m, m*h, h*m, h*h
          const int pa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
          const int pw = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_buf[ks & 1][pa], W[nt][pw][ks], acc[nt], 0, 0, 0);
        }
      }
      // pool over the lane's four registers, bias, split for ip1, store: flat index = pixel * 50 + filter (eigen_classifier.cpp:103-107)
      const int prow = rp, pcol = 4 * xt + q;
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const int f = 32 * np + 16 * nt + j;
        const float v = fmaxf(fmaxf(acc[nt][0], acc[nt][1]), fmaxf(acc[nt][2], acc[nt][3])) + (nt ? k_bias1 : k_bias0);
        if (f < 50) {
          const Bf3 sp = bf16_split3(v);
          unsigned short *o = xs + (size_t)img * F2_XLD + (prow * 12 + pcol) * 50 + f;
          o[0] = sp.h;
          o[xs_plane] = sp.m;
          o[2 * xs_plane] = sp.l;
        }
      }
      if (has_piece) reinterpret_cast<uint4 *>(s_raw)[piece] = stage;
    }
    __syncthreads();  // every tile of the image is done, the next image's raw rows are in LDS
    if (nxt >= n) break;
    split();
    img = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ip1 on the bf16 matrix pipe: D[image][unit] = sum_k X[image][k] W[k][unit], six piece products per k.
//   X: three bf16 planes [n][7232] written by conv2 (k = pixel * 50 + filter, then 32 zeros), W: three bf16 planes
//   [512][7232], unit-major (k contiguous), built once at gpd_hip_set_lenet_weights.
//   A (images): lane (row l & 15, k group l >> 4) = 16 contiguous bytes; B (units) likewise; D: lane (unit l & 15),
//   register r = image 4 (l >> 4) + r  ->  a float4 of the transposed output per lane.
// Workgroup tile 128 units x 16 NT images (NT picked per launch to fill the CUs in whole rounds), four waves, one per SIMD:
// wave w owns units 32 w .. 32 w + 31 x all images (2 NT accumulators).  K in steps of 64: two LDS buffers of
// 3 x (16 NT + 128) rows x 128 bytes, rows XOR-swizzled by (row >> 1) & 7 in 16-byte chunks so that every ds_read_b128 of a
// fragment is bank-conflict free; global -> registers one step ahead, registers -> LDS while the other buffer is multiplied.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F3_THREADS = 256, F3_BU = 128, F3_BK = 64, F3_STEPS = F2_XLD / F3_BK;  // 113
static_assert(F2_XLD % F3_BK == 0, "K steps");

template <int NT>
__global__ __launch_bounds__(F3_THREADS) void fc1_bf16_kernel(const unsigned short *__restrict__ xs, size_t xs_plane,
                                                             const unsigned short *__restrict__ wt, size_t wt_plane,
                                                             const float *__restrict__ bias, float *__restrict__ out_t, int n, int ld_out) {
  constexpr int BM = 16 * NT;
  constexpr int XB = BM * 128, WB = F3_BU * 128;  // bytes of one piece's tile
  constexpr int STAGE = 3 * (XB + WB);
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STAGE];
  static_assert(2 * STAGE <= 160 * 1024, "LDS");
  const int tid = threadIdx.x;
  const int L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3;
  const int u0 = (slot & 3) * F3_BU;
  const int m0 = ((slot >> 2) * 8 + xcd) * BM;
  if (m0 >= n) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane >> 4, j = lane & 15;
  // loader roles: a 16-byte chunk per thread and load, 8 consecutive lanes cover one row's 128 bytes; thread -> (row tid >> 3
  // + 32 i, chunk tid & 7): the row's swizzle (row >> 1) & 7 does not depend on i
  constexpr int W_PT = F3_BU / 32, X_PT = (BM + 31) / 32;  // loads per piece
  const int lrow = tid >> 3, lch = tid & 7;
  const unsigned short *wsrc = wt + (size_t)(u0 + lrow) * F2_XLD + lch * 8;
  const int ldst = lrow * 128 + ((lch ^ ((lrow >> 1) & 7)) << 4);
  const unsigned short *xsrc[X_PT];
#pragma unroll
  for (int i = 0; i < X_PT; i++) xsrc[i] = xs + (size_t)min(m0 + lrow + 32 * i, n - 1) * F2_XLD + lch * 8;
  u32x4 rw[3 * W_PT], rx[3 * X_PT];
  auto fetch = [&](int step, u32x4(&w2)[3 * W_PT], u32x4(&x2)[3 * X_PT]) {
    const int k0 = step * F3_BK;
#pragma unroll
    for (int pc = 0; pc < 3; pc++) {
#pragma unroll
      for (int i = 0; i < W_PT; i++) w2[pc * W_PT + i] = *reinterpret_cast<const u32x4 *>(wsrc + pc * wt_plane + (size_t)i * 32 * F2_XLD + k0);
#pragma unroll
      for (int i = 0; i < X_PT; i++) x2[pc * X_PT + i] = *reinterpret_cast<const u32x4 *>(xsrc[i] + pc * xs_plane + k0);
    }
  };
  auto stage = [&](int buf, const u32x4(&w2)[3 * W_PT], const u32x4(&x2)[3 * X_PT]) {
    uint8_t *b = smem + buf * STAGE + ldst;
#pragma unroll
    for (int pc = 0; pc < 3; pc++) {
#pragma unroll
      for (int i = 0; i < W_PT; i++) *reinterpret_cast<u32x4 *>(b + 3 * XB + pc * WB + i * 32 * 128) = w2[pc * W_PT + i];
#pragma unroll
      for (int i = 0; i < X_PT; i++)
        if (lrow + 32 * i < BM) *reinterpret_cast<u32x4 *>(b + pc * XB + i * 32 * 128) = x2[pc * X_PT + i];
    }
  };
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int ut = 0; ut < 2; ut++) acc[t][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses inside a piece's tile: row * 128 + ((4 kk + g) ^ ((row >> 1) & 7)) * 16; row & 15 = j for both operands
  const int sw = (j >> 1) & 7;
  const int xrow = j * 128, wrow = (32 * wave + j) * 128;
  auto compute = [&](int buf) {
    const uint8_t *b = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int coff = ((4 * kk + g) ^ sw) << 4;
      bf16x8 xa[NT][3], wb[2][3];
#pragma unroll
      for (int pc = 0; pc < 3; pc++) {
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          wb[ut][pc] = as_bf16x8(*reinterpret_cast<const uint4 *>(b + 3 * XB + pc * WB + wrow + ut * 16 * 128 + coff));
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
          xa[t][pc] = as_bf16x8(*reinterpret_cast<const uint4 *>(b + pc * XB + xrow + t * 16 * 128 + coff));
        }
      }
#pragma unroll
      for (int term = 0; term < 6; term++) {
        // (image piece, weight piece): l*h, h*l, m*m, m*h, h*m, h*h — small terms first
        const int pa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
        const int pw = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
          for (int ut = 0; ut < 2; ut++) acc[t][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][pa], wb[ut][pw], acc[t][ut], 0, 0, 0);
      }
    }
  };
  fetch(0, rw, rx);
  stage(0, rw, rx);
  fetch(1, rw, rx);
  __syncthreads();
  for (int t = 0; t < F3_STEPS; t++) {
    // buffer (t + 1) & 1 was read in step t - 1, whose barrier is behind us; the registers hold step t + 1 (the last step
    // restages itself into the buffer nobody reads again — no condition, so the compiler counts the loads exactly)
    stage((t + 1) & 1, rw, rx);
    fetch(min(t + 2, F3_STEPS - 1), rw, rx);
    compute(t & 1);
    __syncthreads();
  }
  // bias, ReLU (eigen_classifier.cpp:113), transposed store for ip2: a lane holds four consecutive images of one unit
#pragma unroll
  for (int ut = 0; ut < 2; ut++) {
    const int u = u0 + 32 * wave + 16 * ut + j;
    if (u >= kFc1Out) continue;
    const float bu = bias[u];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int m = m0 + 16 * t + 4 * g;
      float *o = out_t + (size_t)u * ld_out + m;
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (m + r < n) o[r] = fmaxf(acc[t][ut][r] + bu, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: the operand tables (built once per gpd_hip_set_lenet_weights) and the launch sequence.
// ---------------------------------------------------------------------------------------------------------------------
void lenet_fast_free(LeNetFast &f) {
  void *ps[] = {f.c1a, f.c1corr, f.c1shift, f.c2b, f.f1wt};
  for (void *p : ps)
    if (p) (void)hipFree(p);
  f = LeNetFast();
}

// conv1: per filter the fixed-point position s with |round(w 2^s)| < 2^30 for every weight, the four balanced base-256 digits
// of every weight laid out as the MFMA A fragments of conv1_i8_kernel, and 128 * sum(W) (the x - 128 shift of the inputs)
void lenet_fast_conv1_tables(int channels, const float *w, std::vector<uint8_t> &atab, std::vector<double> &corr, std::vector<int> &shift) {
  const int K = channels * 25;
  std::vector<long long> W((size_t)20 * K);
  corr.assign(20, 0.0);
  shift.assign(20, 0);
  for (int f = 0; f < 20; f++) {
    float mx = 0.f;
    for (int k = 0; k < K; k++) mx = std::fmax(mx, std::fabs(w[(size_t)f * K + k]));
    int s = 0;
    if (mx > 0.f) {
      int e;
      (void)std::frexp(mx, &e);  // mx = m 2^e, 0.5 <= m < 1  ->  mx < 2^e
      s = 30 - e;                // |w| 2^s < 2^30
    }
    shift[f] = s;
    long long sum = 0;
    for (int k = 0; k < K; k++) {
      const long long v = std::llrint(std::ldexp((double)w[(size_t)f * K + k], s));  // the scaling is exact, one rounding to integer
      W[(size_t)f * K + k] = v;
      sum += v;
    }
    corr[f] = 128.0 * (double)sum;  // exact: |sum| < 375 * 2^30
  }
  atab.assign((size_t)F1_KS * F1_MT * 64 * 16, 0);
  for (int ks = 0; ks < F1_KS; ks++)
    for (int mt = 0; mt < F1_MT; mt++)
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, g = lane >> 4;
        const int f = 4 * mt + (i >> 2), digit = i & 3;
        const int tap = f1_tap(ks, g);
        if (tap < 0) continue;
        for (int c = 0; c < channels; c++) {
          long long v = W[(size_t)f * K + c * 25 + tap];
          int d = 0;
          for (int q = 0; q <= digit; q++) {  // balanced digits, least significant first: d in [-128, 127]
            d = (int)(((v + 128) & 255) - 128);
            v = (v - d) >> 8;
          }
          atab[((size_t)(ks * F1_MT + mt) * 64 + lane) * 16 + c] = (uint8_t)(int8_t)d;
        }
      }
}

// conv2: the three bf16 pieces of every weight as the MFMA B fragments of conv2_bf16_kernel:
// [wave pair np][column tile nt][piece][k-step][lane] x 8 bf16; filter 32 np + 16 nt + (lane & 15), filters >= 50 are zero
void lenet_fast_conv2_tables(const float *w, std::vector<unsigned short> &btab) {
  btab.assign((size_t)2 * 2 * 3 * 16 * 64 * 8, 0);
  for (int np = 0; np < 2; np++)
    for (int nt = 0; nt < 2; nt++)
      for (int ks = 0; ks < 16; ks++)
        for (int lane = 0; lane < 64; lane++) {
          const int f = 32 * np + 16 * nt + (lane & 15), g = lane >> 4;
          if (f >= 50) continue;
          for (int h = 0; h < 2; h++) {
            const int e = ks + 16 * h, tap = f2_slot_tap(e, g), cg = f2_slot_cg(e, g);
            if (tap < 0) continue;
            for (int c = 0; c < 4; c++) {
              const Bf3 sp = bf16_split3(w[(size_t)f * 500 + (4 * cg + c) * 25 + tap]);
              const unsigned short pcs[3] = {sp.h, sp.m, sp.l};
              for (int pc = 0; pc < 3; pc++)
                btab[((((size_t)(np * 2 + nt) * 3 + pc) * 16 + ks) * 64 + lane) * 8 + 4 * h + c] = pcs[pc];
            }
          }
        }
}

// ip1: the reference's file layout is column-major 500 x 7200 == row-major [7200][500] (dense_layer.cpp:7); here unit-major
// bf16 planes [3][512][7232], zero beyond unit 499 / k 7199
void lenet_fast_ip1_tables(const float *w, std::vector<unsigned short> &wt) {
  const size_t plane = (size_t)512 * F2_XLD;
  wt.assign(3 * plane, 0);
  for (int k = 0; k < kFc1In; k++)
    for (int u = 0; u < kFc1Out; u++) {
      const Bf3 sp = bf16_split3(w[(size_t)k * kFc1Out + u]);
      const size_t o = (size_t)u * F2_XLD + k;
      wt[o] = sp.h;
      wt[plane + o] = sp.m;
      wt[2 * plane + o] = sp.l;
    }
}

}  // namespace gpd

// test hook (host only, no device): the operand tables as the kernels read them, so that a CPU test can replay the kernels'
// index arithmetic against a plain convolution (tests/test_lenet_fast_tables.py)
extern "C" int gpd_hip_lenet_fast_tables(int channels, const float *c1w, const float *c2w, uint8_t *atab, double *corr, int *shift,
                                         unsigned short *btab) {
  if (!c1w || !c2w || !atab || !corr || !shift || !btab || channels < 1 || channels > 16) return GPD_ERR_INVALID;
  std::vector<uint8_t> a;
  std::vector<double> c;
  std::vector<int> sh;
  std::vector<unsigned short> b;
  gpd::lenet_fast_conv1_tables(channels, c1w, a, c, sh);
  gpd::lenet_fast_conv2_tables(c2w, b);
  memcpy(atab, a.data(), a.size());
  memcpy(corr, c.data(), c.size() * sizeof(double));
  memcpy(shift, sh.data(), sh.size() * sizeof(int));
  memcpy(btab, b.data(), b.size() * sizeof(unsigned short));
  return GPD_OK;
}

namespace gpd {

hipError_t lenet_fast_prepare(LeNetFast &f, int channels, const float *c1w, const float *c2w, const float *f1w) {
  lenet_fast_free(f);
  std::vector<uint8_t> atab;
  std::vector<double> corr;
  std::vector<int> shift;
  std::vector<unsigned short> btab, wt;
  lenet_fast_conv1_tables(channels, c1w, atab, corr, shift);
  lenet_fast_conv2_tables(c2w, btab);
  lenet_fast_ip1_tables(f1w, wt);
  hipError_t e;
  auto up = [&](auto **dst, const void *src, size_t bytes) -> hipError_t {
    if ((e = hipMalloc(reinterpret_cast<void **>(dst), bytes)) != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  if ((e = up(&f.c1a, atab.data(), atab.size())) != hipSuccess) return e;
  if ((e = up(&f.c1corr, corr.data(), corr.size() * sizeof(double))) != hipSuccess) return e;
  if ((e = up(&f.c1shift, shift.data(), shift.size() * sizeof(int))) != hipSuccess) return e;
  if ((e = up(&f.c2b, btab.data(), btab.size() * sizeof(unsigned short))) != hipSuccess) return e;
  if ((e = up(&f.f1wt, wt.data(), wt.size() * sizeof(unsigned short))) != hipSuccess) return e;
  return hipSuccess;
}

template <int NT>
static void fc1f_launch(const LeNetWeights &w, LeNetScratch &s, int n, hipStream_t stream) {
  const int m_tiles = (n + 16 * NT - 1) / (16 * NT);
  const int groups = (m_tiles + 7) / 8;
  const size_t xs_plane = (size_t)s.capacity * F2_XLD, wt_plane = (size_t)512 * F2_XLD;
  fc1_bf16_kernel<NT><<<groups * 4 * 8, F3_THREADS, 0, stream>>>(s.xs, xs_plane, w.fast.f1wt, wt_plane, w.f1b, s.fc1t, n, s.capacity);
}
// image-tile height: the smallest multiple of 16 (at most 80: two LDS buffers) whose tiles fill the chip's 64 workgroup
// columns (256 CUs / 4 unit tiles) in r whole rounds, r as small as possible
static int fc1f_pick_nt(int n) {
  for (int r = 1;; r++) {
    const int nt = (n + 64 * r * 16 - 1) / (64 * r * 16);
    if (nt <= 5) return nt < 1 ? 1 : nt;
  }
}

hipError_t lenet_forward_fast(const LeNetWeights &w, LeNetScratch &s, const uint8_t *img, int m, float *d_scores, hipStream_t stream,
                              hipEvent_t *kernel_events, int *queue) {
  const int num_cus = s.num_cus;
  const int grid = m < num_cus ? m : num_cus;
  switch (w.channels) {
    case 15: conv1_i8_kernel<15><<<grid, F1_THREADS, 0, stream>>>(img, w.fast.c1a, w.fast.c1corr, w.fast.c1shift, w.c1b, s.pool1, m, queue); break;
    case 12: conv1_i8_kernel<12><<<grid, F1_THREADS, 0, stream>>>(img, w.fast.c1a, w.fast.c1corr, w.fast.c1shift, w.c1b, s.pool1, m, queue); break;
    case 3: conv1_i8_kernel<3><<<grid, F1_THREADS, 0, stream>>>(img, w.fast.c1a, w.fast.c1corr, w.fast.c1shift, w.c1b, s.pool1, m, queue); break;
    case 1: conv1_i8_kernel<1><<<grid, F1_THREADS, 0, stream>>>(img, w.fast.c1a, w.fast.c1corr, w.fast.c1shift, w.c1b, s.pool1, m, queue); break;
    default: return hipErrorInvalidValue;
  }
  if (kernel_events) (void)hipEventRecord(kernel_events[0], stream);
  conv2_bf16_kernel<<<grid, F2_THREADS, 0, stream>>>(s.pool1, w.fast.c2b, w.c2b, s.xs, (size_t)s.capacity * F2_XLD, m, queue + 1);
  if (kernel_events) (void)hipEventRecord(kernel_events[1], stream);
  switch (fc1f_pick_nt(m)) {
    case 1: fc1f_launch<1>(w, s, m, stream); break;
    case 2: fc1f_launch<2>(w, s, m, stream); break;
    case 3: fc1f_launch<3>(w, s, m, stream); break;
    case 4: fc1f_launch<4>(w, s, m, stream); break;
    default: fc1f_launch<5>(w, s, m, stream); break;
  }
  if (kernel_events) (void)hipEventRecord(kernel_events[2], stream);
  return hipGetLastError();
}

}  // namespace gpd
