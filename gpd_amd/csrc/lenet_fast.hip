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
//   ip2     two 500-long f32 chains, as before (lenet.hip fc2_score_kernel), which also adds ip1's four K quarters in order.
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
// ---- NARROW images (C <= 4 channels: the 3- and 1-channel geometries, cfg/image_geometry_3channels.cfg; round 6).  With 16 bytes
// per pixel a 3-channel image fills 3 of the 16 k-bytes of every tap: 7 k-steps for a fifth of the work.  Here a pixel is FOUR
// bytes (channels 0..C-1, then padding), so 16 contiguous bytes are four neighbouring pixels of a row = four taps of one kernel
// row.  A ds_read_b128 wants 16-byte alignment (MI355X_MICROARCH.md: an unaligned one is replayed at 64 cycles) and a window may
// start at any pixel, so the image lies in LDS four times, copy s shifted by s pixels: the window starting at pixel X is aligned
// in copy (-X) mod 4 — a constant of the lane (the lane's column inside its tile).  K-groups: (kernel row ky, taps kx 0..3) and
// (ky, taps kx 4..7: only kx = 4 is real, the others carry zero weights) = 10 groups -> 3 k-steps of 4 lane groups (2 slots
// empty): 15 MFMAs per tile instead of 35.
// ---- TWELVE channels (cfg/image_geometry_12channels.cfg; round 6): a pixel is 12 bytes, so the five taps of a kernel row are 60
// contiguous bytes = ONE k-step (4 padding bytes carry zero weights): 5 k-steps instead of 7, 25 MFMAs per tile instead of 35.  A
// lane group's 16 bytes of that row start at 12 px + 16 g — 4-byte aligned only.  Four shifted copies (as for the narrow images) do
// not fit; TWO do, shifted by 0 and 4 bytes: in the copy of its column's parity every fragment is 8-byte aligned and is read as
// two ds_read_b64.
constexpr int F1P_KS = 5;
constexpr int F1P_ROWB = 736;                       // bytes per LDS row: 60 x 12 + the bytes a window reaches beyond pixel 59
constexpr int F1P_COPY = kImg * F1P_ROWB;           // 44160 bytes per copy
constexpr int F1N_KS = 3;
constexpr int F1N_PITCH = 68;                       // pixels per row: 60 + 3 (shift) + 3 (window beyond column 59), multiple of 4
constexpr int F1N_COPY = kImg * F1N_PITCH * 4;      // 16320 bytes per copy
// slot (k-step ks, lane group g) -> ky * 2 + half (half 0: taps kx 0..3, half 1: kx 4..7), -1: empty
__host__ __device__ constexpr int f1n_slot(int ks, int g) {
  const int sl = 4 * ks + g;
  return sl < 5 ? 2 * sl : sl < 10 ? 2 * (sl - 5) + 1 : -1;
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
  constexpr bool NARROW = C <= 4;                       // four-byte pixels, four shifted copies (see f1n_slot)
  constexpr bool PACK12 = C == 12;                      // twelve-byte pixels, two shifted copies, a kernel row per k-step
  constexpr int KS = NARROW ? F1N_KS : PACK12 ? F1P_KS : F1_KS;
  __shared__ __attribute__((aligned(16))) uint8_t s_hwc[NARROW ? 4 * F1N_COPY : PACK12 ? 2 * F1P_COPY : F1_HWC];
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[RAW];
  __shared__ int s_nxt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  int img = blockIdx.x;
  if (img >= n) return;
  // waves w and w + 4 share a SIMD (MI355X_MICROARCH.md: a workgroup's waves go to the SIMDs cyclically): the first keeps a
  // higher priority for the whole launch, so its MFMA run always has the matrix pipe and the other wave's run falls into ITS
  // epilogue — complementary phases instead of lockstep
  if (wave < F1_WAVES / 2) __builtin_amdgcn_s_setprio(2);
  // the weight fragments: resident for the whole launch
  i32x4 A[KS][F1_MT];
#pragma unroll
  for (int ks = 0; ks < KS; ks++)
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
    if constexpr (NARROW) {
      // raw planar [C][60][60] -> four copies of [60][68] four-byte pixels (x ^ 0x80; the padding byte's weight is zero), copy s
      // shifted by s pixels; a task is four consecutive pixels: one 16-byte store into copy 0, four 4-byte stores into the others
      for (int task = tid; task < kImg * (kImg / 4); task += F1_THREADS) {
        const int y = task / (kImg / 4), x0 = 4 * (task - y * (kImg / 4));
        uint32_t r[4];
#pragma unroll
        for (int c = 0; c < 4; c++) r[c] = c < C ? *reinterpret_cast<const uint32_t *>(s_raw + c * kPix + y * kImg + x0) : 0x80808080u;
        const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
        const uint32_t o0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u) ^ 0x80808080u, o1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u) ^ 0x80808080u;
        const uint32_t o2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u) ^ 0x80808080u, o3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u) ^ 0x80808080u;
        uint8_t *d = s_hwc + (y * F1N_PITCH + x0) * 4;
        *reinterpret_cast<uint4 *>(d) = make_uint4(o0, o1, o2, o3);
#pragma unroll
        for (int sh = 1; sh < 4; sh++) {
          uint32_t *e = reinterpret_cast<uint32_t *>(d + sh * F1N_COPY + sh * 4);
          e[0] = o0;
          e[1] = o1;
          e[2] = o2;
          e[3] = o3;
        }
      }
      return;
    }
    if constexpr (PACK12) {
      // raw planar [12][60][60] -> two copies of [60][736 B] twelve-byte pixels (x ^ 0x80), copy 1 shifted by 4 bytes; a task is four
      // consecutive pixels = 48 contiguous bytes: three 16-byte stores into copy 0 (48 x0 / 4 is a multiple of 16), twelve 4-byte
      // stores into copy 1
      for (int task = tid; task < kImg * (kImg / 4); task += F1_THREADS) {
        const int y = task / (kImg / 4), x0 = 4 * (task - y * (kImg / 4));
        uint32_t o[4][3];  // [pixel][channel group]
#pragma unroll
        for (int cg = 0; cg < 3; cg++) {
          uint32_t r[4];
#pragma unroll
          for (int c = 0; c < 4; c++) r[c] = *reinterpret_cast<const uint32_t *>(s_raw + (4 * cg + c) * kPix + y * kImg + x0);
          const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
          const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
          o[0][cg] = __builtin_amdgcn_perm(t2, t0, 0x05040100u) ^ 0x80808080u;
          o[1][cg] = __builtin_amdgcn_perm(t2, t0, 0x07060302u) ^ 0x80808080u;
          o[2][cg] = __builtin_amdgcn_perm(t3, t1, 0x05040100u) ^ 0x80808080u;
          o[3][cg] = __builtin_amdgcn_perm(t3, t1, 0x07060302u) ^ 0x80808080u;
        }
        uint8_t *d = s_hwc + y * F1P_ROWB + 12 * x0;
        uint4 *d4 = reinterpret_cast<uint4 *>(d);
        d4[0] = make_uint4(o[0][0], o[0][1], o[0][2], o[1][0]);
        d4[1] = make_uint4(o[1][1], o[1][2], o[2][0], o[2][1]);
        d4[2] = make_uint4(o[2][2], o[3][0], o[3][1], o[3][2]);
        uint32_t *e = reinterpret_cast<uint32_t *>(d + F1P_COPY + 4);
#pragma unroll
        for (int px = 0; px < 4; px++)
#pragma unroll
          for (int cg = 0; cg < 3; cg++) e[3 * px + cg] = o[px][cg];
      }
      return;
    }
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
  // (narrow: the lane's copy is the one in which its windows are aligned: tile columns start at multiples of 8)
  const int n_shift = (4 - (m_x & 3)) & 3;
  // (twelve channels: the copy of the column's parity, shifted by four bytes for the odd columns)
  const int lane_off = NARROW   ? n_shift * F1N_COPY + (m_row * F1N_PITCH + m_x + n_shift) * 4
                       : PACK12 ? (m_x & 1) * (F1P_COPY + 4) + m_row * F1P_ROWB + 12 * m_x + 16 * q
                                : (m_row * F1_PITCH + m_x) * 16;
  // narrow: the byte offsets of the lane group's three slots (an empty slot reads slot 0's window: its weights are zero)
  int n_slot[F1N_KS];
#pragma unroll
  for (int ks = 0; ks < F1N_KS; ks++) {
    const int sl = q == 0 ? f1n_slot(ks, 0) : q == 1 ? f1n_slot(ks, 1) : q == 2 ? f1n_slot(ks, 2) : f1n_slot(ks, 3);
    n_slot[ks] = sl < 0 ? 0 : ((sl >> 1) * F1N_PITCH + 4 * (sl & 1)) * 4;
  }
  const int p = j & 3, w = j >> 2;
  const double k_corr_own = corr[4 * p + q], k_corr_4 = corr[16 + q];
  const int k_shift_own = shift[4 * p + q], k_shift_4 = shift[16 + q];  // the filter's fixed-point position s: value = integer * 2^-s
  const float k_bias_own = bias[4 * p + q], k_bias_4 = bias[16 + q];
  for (;;) {
    if (tid == 0) s_nxt = (int)gridDim.x + atomicAdd(queue, 1);
    __syncthreads();  // the pixel-major image is complete, the raw region is free, s_nxt is visible
    const int nxt = s_nxt;
    const uint4 *nsrc = reinterpret_cast<const uint4 *>(images + (size_t)(nxt < n ? nxt : img) * RAW);
    // A wave's tiles are wave, wave + 8, ...  Per tile: 35 MFMAs back to back (a VALU instruction of the SAME wave between two
    // MFMAs costs matrix-pipe time — interleaving this wave's epilogue with its own MFMAs by sched_group_barrier measured
    // 0.48 ms against 0.40 — an instruction of the SIMD's OTHER wave costs nothing), then the requests for the next tile's pixel
    // fragments (their registers are free now; the data arrives during the epilogue), then the epilogue.  The two waves of a SIMD
    // run at different priorities (see kernel head): with equal priorities they fall into lockstep — both in their MFMA runs,
    // sharing the pipe, then both in their epilogues with the pipe idle.
    auto tile_addr = [&](int t, const uint8_t *&pa, const uint8_t *&pb, const uint8_t *&pc) {
      const int tc = t < F1_TILES ? t : F1_TILES - 1;  // past the end: any valid address (the fragments are not used)
      const int trow = tc / 7, tcol = tc - 7 * trow;
      if constexpr (NARROW) {
        pa = pb = pc = s_hwc + ((2 * trow) * F1N_PITCH + 8 * tcol) * 4 + lane_off;
      } else if constexpr (PACK12) {
        pa = pb = pc = s_hwc + (2 * trow) * F1P_ROWB + 12 * (8 * tcol) + lane_off;
      } else {
        const uint8_t *base = s_hwc + ((2 * trow) * F1_PITCH + 8 * tcol) * 16 + lane_off;
        pa = base + kyg * F1_ROWB;
        pb = base + 4 * F1_ROWB + q * 16;
        pc = base + 4 * F1_ROWB + 4 * 16;
      }
    };
    auto frag_b = [&](int ks, const uint8_t *pa, const uint8_t *pb, const uint8_t *pc) {
      if constexpr (NARROW) return *reinterpret_cast<const i32x4 *>(pa + n_slot[ks < F1N_KS ? ks : 0]);
      else if constexpr (PACK12) {  // kernel row ks: two 8-byte reads (the window is 8-byte aligned in the lane's copy)
        typedef const volatile __attribute__((address_space(3))) unsigned long long *lds_u64;
        const unsigned long long lo = *(lds_u64)(pa + ks * F1P_ROWB), hi = *(lds_u64)(pa + ks * F1P_ROWB + 8);
        return i32x4{(int)(uint32_t)lo, (int)(uint32_t)(lo >> 32), (int)(uint32_t)hi, (int)(uint32_t)(hi >> 32)};
      }
      else return *reinterpret_cast<const i32x4 *>(ks < 5 ? pa + ks * 16 : ks == 5 ? pb : pc);
    };
    i32x4 B[KS];
    {
      const uint8_t *pa, *pb, *pc;
      tile_addr(wave, pa, pb, pc);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) B[ks] = frag_b(ks, pa, pb, pc);
    }
    int it = 0;
    for (int t = wave; t < F1_TILES; t += F1_WAVES, it++) {
      // one 16-byte piece of the next image per thread and tile: requested now, stored after the tile
      const int piece = it * F1_THREADS + tid;
      const bool has_piece = nxt < n && piece < NV;
      uint4 stage = make_uint4(0, 0, 0, 0);
      if (has_piece) stage = nsrc[piece];
      i32x4 acc[F1_MT];
#pragma unroll
      for (int mt = 0; mt < F1_MT; mt++) acc[mt] = i32x4{0, 0, 0, 0};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int mt = 0; mt < F1_MT; mt++) acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][mt], B[ks], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      {
        const uint8_t *pa, *pb, *pc;
        tile_addr(t + F1_WAVES, pa, pb, pc);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) B[ks] = frag_b(ks, pa, pb, pc);
      }
      __builtin_amdgcn_sched_barrier(0);
      // epilogue: exact sum of the digit planes (f64), pool over the quad, one rounding.  The quad's four lanes end up with the
      // same pooled sums: lane p finishes row tile p (filter 4 p + q), and every lane row tile 4 (filter 16 + q, stored by p = 0)
      double sm[F1_MT];
#pragma unroll
      for (int mt = 0; mt < F1_MT; mt++) {
        const int hi = (int)((unsigned)acc[mt][3] << 8) + acc[mt][2];  // |.| < 2^31: the top digit is within +-64
        const int lo = (int)((unsigned)acc[mt][1] << 8) + acc[mt][0];
        sm[mt] = dpp_quad_max(__builtin_fma((double)hi, 65536.0, (double)lo));
      }
      const double s_own = p == 0 ? sm[0] : p == 1 ? sm[1] : p == 2 ? sm[2] : sm[3];
      const float v_own = ldexpf((float)(s_own + k_corr_own), -k_shift_own) + k_bias_own;
      const float v_4 = ldexpf((float)(sm[4] + k_corr_4), -k_shift_4) + k_bias_4;
      const int trow = t / 7, tcol = t - 7 * trow;
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
// Eight waves per workgroup, two per SIMD: a wave keeps the three bf16 planes of 16 filters (3 pieces x 16 k-steps x 4 VGPRs
// = 192 registers) for the whole launch and walks pixel tiles of the image; the activations of a k-step (3 pieces x 2 reads of
// 8 bytes) feed 6 MFMAs on two alternating accumulators.  Seven waves share the 3 x 36 tiles of filters 0..47, the eighth
// computes filters 48 and 49 by another route (the roles: in the kernel; until round 6 every wave had 18 tiles of one of FOUR
// column tiles, the fourth 14 / 16 zeros).  (Round 5's first version ran four waves with 32 filters each in 512 registers, one
// per SIMD: every stall of the single wave idled the SIMD's matrix pipe, MfmaUtil 63 %.)
// LDS: the image as bf16 pieces [row 28][piece 3][column 28][channel 20] (94 080 B) + the next image's raw f32 rows (62 720 B),
// which arrive one 16-byte piece per thread and tile and are split between two barriers.
// The 128 four-channel k-slots of the 16 k-steps hold the 125 (tap, channel group) pairs so that the two lane groups of a
// ds_read_b64 half read neighbouring columns or rows of ONE channel group (overlapping addresses broadcast: conflict free):
//   entry e = ks + 16 h: e < 25: ky = e / 5, cg = e % 5, kx = g       e in 25..29: cg = e - 25, kx = 4, ky = g
//                        e = 30: ky = kx = 4, cg = g                  e = 31: ky = kx = 4, cg = 4 (lane group 0 only)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F2_THREADS = 512;
constexpr int F2_PP = 28 * 40;         // bytes of one piece row: 28 pixels x 20 bf16
constexpr int F2_RS = 3 * F2_PP;       // bytes of one image row (three pieces)
constexpr int F2_IMG = 28 * F2_RS;     // 94080
constexpr int F2_RAW = 784 * 20 * 4;   // 62720
constexpr int F2_LUNITS = 24;         // work units of the wave that computes filters 48, 49: 12 row pairs x 2 column halves
constexpr int F2_XLD = kLenetXld;      // row length of the flat bf16 planes: 7200 + 96 zeros (ip1 walks four K quarters in steps of 32)

// ip1's operands — X (the flattened pool2 of every image, written by conv2) and W (its weights, unit-major) — are kept BLOCKED
// the way ip1's loader takes them: 1 KB blocks [row block r / 16][k step k / 32 (228)][piece (3)] of 16 rows x 64 bytes, a
// row's four 16-byte k chunks XOR-swizzled as the LDS tile wants them.  One global_load_lds_dwordx4 copies one block, 1 KB of
// consecutive addresses = eight whole cache lines (rows of 7296 k, 64 bytes of each per instruction, measured 0.18 ms for ip1
// against 0.14 with blocks), and the blocks a workgroup streams for a row block lie behind one another.
constexpr int F3_KSTEPS = F2_XLD / 32;  // 228
__host__ __device__ inline size_t f3_blocked(int r, int k, int pc) {  // byte offset of (row r, element k, piece pc)
  const int rr = r & 15;
  return ((((size_t)(r >> 4) * F3_KSTEPS + (k >> 5)) * 3 + pc) << 10) + rr * 64 + ((((k >> 3) & 3) ^ ((4 - (rr >> 2)) & 3)) << 4) + (k & 7) * 2;
}

// (tap = ky * 5 + kx, channel group) of k-slot entry e for lane group g; tap -1: empty
__host__ __device__ constexpr int f2_slot_tap(int e, int g) {
  return e < 25 ? (e / 5) * 5 + g : e < 30 ? g * 5 + 4 : e == 30 ? 24 : (g == 0 ? 24 : -1);
}
__host__ __device__ constexpr int f2_slot_cg(int e, int g) { return e < 25 ? e % 5 : e < 30 ? e - 25 : e == 30 ? g : 4; }

__global__ __launch_bounds__(F2_THREADS) void conv2_bf16_kernel(const float *__restrict__ pool1, const uint4 *__restrict__ btab,
                                                                const float *__restrict__ bias, unsigned short *__restrict__ xs,
                                                                int n, int *__restrict__ queue) {
  __shared__ __attribute__((aligned(16))) uint8_t s_img[F2_IMG];
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[F2_RAW];
  __shared__ int s_nxt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int img0 = blockIdx.x;
  if (img0 >= n) return;
  {  // the first image, by everybody
    const uint4 *src = reinterpret_cast<const uint4 *>(pool1 + (size_t)img0 * (784 * 20));
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
  // one pooled value (+ bias) as its three pieces into ip1's blocked X: flat index = pixel * 50 + filter (eigen_classifier.cpp:103-107)
  auto store_x = [&](int img, int pixel, int f, float v) {
    const Bf3 sp = bf16_split3(v);
    uint8_t *o = reinterpret_cast<uint8_t *>(xs) + f3_blocked(img, pixel * 50 + f, 0);
    *reinterpret_cast<unsigned short *>(o) = sp.h;
    *reinterpret_cast<unsigned short *>(o + 1024) = sp.m;
    *reinterpret_cast<unsigned short *>(o + 2048) = sp.l;
  };
  // the frame every role runs per image: `passes` loop passes (the seven waves of the full groups: at least nine, each carries
  // 16 bytes per thread of the next image's raw rows from global memory into the LDS), two barriers
  auto image_loop = [&](int passes, auto carries, auto &&work) {  // carries: whether the role takes part in fetching the next image
    int img = img0;
    for (;;) {
      if (tid == 0) s_nxt = (int)gridDim.x + atomicAdd(queue, 1);
      __syncthreads();  // the pieces are complete, the raw region is free, s_nxt is visible
      const int nxt = s_nxt;
      const uint4 *nsrc = reinterpret_cast<const uint4 *>(pool1 + (size_t)(nxt < n ? nxt : img) * (784 * 20));
#pragma unroll 1
      for (int tt = 0; tt < passes; tt++) {
        // (waves 0-6 carry the next image, 448 pieces a pass, nine passes; the eighth wave's passes are too short for a global
        //  load to land in one: it waited for its pieces, and — gfx9 counts stores on vmcnt too — for its own stores)
        const int piece = tt * (7 * 64) + tid;
        const bool has_piece = decltype(carries)::value && nxt < n && piece < F2_RAW / 16;
        uint4 stage = make_uint4(0, 0, 0, 0);
        if (has_piece) stage = nsrc[piece];
        work(img, tt);
        if (has_piece) reinterpret_cast<uint4 *>(s_raw)[piece] = stage;
      }
      __syncthreads();  // every tile of the image is done, the next image's raw rows are in LDS
      if (nxt >= n) break;
      split();
      img = nxt;
    }
  };
  // Roles (round 6).  50 filters are three column tiles of 16 and two left over; until round 6 a fourth column tile carried the
  // two (14 of its 16 columns zero: a quarter of the kernel's matrix instructions for 4 % of its filters).  Now seven waves
  // share the 3 x 36 pixel tiles of the three full groups and the eighth computes filters 48 and 49 by another route;
  // waves w and w + 4 share a SIMD (a workgroup's waves go to the SIMDs cyclically), so every SIMD gets 2880 MFMAs per image:
  //   waves 0-3: group 1 + wave / 2, half wave & 1 (18 tiles x 96; group 2: 20 + 16);  waves 4-6: group 0, a third each (12 x 96);
  //   wave 7: the two filters (24 units x 48).  The two roles are two copies of the image loop, so that the eighth wave's
  //   registers are not those of a full group's weights.
  if (wave == 7) {
    // Filters 48 and 49 as a 16 x 16 tile of (filter, kernel column kx) rows x 16 INPUT columns of one conv row:
    //   D[(f, kx)][x] = sum over (ky, c) of W[f][c][ky][kx] * pool1[y + ky][x][c]        (K = 5 kernel rows x 20 channels)
    //   conv[f][y][x] = sum over kx of D[(f, kx)][x + kx]                                  (five shifted rows, added across lanes)
    // 16 input columns give 12 outputs, so a conv row is two such tiles; K is laid out as 15 chunks of 8 = (ky, 8 channels)
    // with the third chunk of a kernel row half empty (channels 16..19 + 4 zero weights): four k-steps of 32.  A work unit =
    // a pair of conv rows x 12 columns = 6 pooled pixels of both filters: 2 x 4 k-steps x 6 piece products = 48 MFMAs; 24
    // units per image = 1152 against the 3456 of a padded column tile.  The weights are the A operand here (rows (f, kx):
    // lane group 0 holds (48, kx 0..3) in its four registers, group 1 (49, kx 0..3), group 2 (48, 4), group 3 (49, 4)).
    __builtin_amdgcn_s_setprio(3);  // its chain of short steps is the longest of the SIMD's two: the pipe when it is ready (0.735 -> 0.68 ms)
    bf16x8 LW[3][4];
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) LW[pc][ks] = as_bf16x8(btab[((3 * 3 + pc) * 16 + ks) * 64 + lane]);
    int l_lo[4], l_hi[4];  // the lane's two 8-byte reads of k-step ks, relative to (conv row, first input column) of piece 0
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int idx = min(4 * ks + q, 14), ky = idx / 3, c8 = idx - 3 * ky;  // (chunk 15 is empty: zero weights, any finite data will do)
      l_lo[ks] = ky * F2_RS + j * 40 + c8 * 16;
      l_hi[ks] = c8 == 2 ? l_lo[ks] : l_lo[ks] + 8;  // (channels 20..23 do not exist: zero weights against a second copy of 16..19)
    }
    const int f_own = 48 + (q & 1);
    const float k_bias = bias[f_own];
    auto unit_base = [&](int u) { return s_img + (2 * (u >> 1)) * F2_RS + (12 * (u & 1)) * 40; };
    auto lfrag = [&](const uint8_t *ubase, int step, int pc) -> bf16x8 {  // step = conv row of the pair * 4 + k-step
      typedef const volatile __attribute__((address_space(3))) unsigned long long *lds_u64;
      const uint8_t *a = ubase + (step >> 2) * F2_RS + pc * F2_PP;
      const unsigned long long lo = *(lds_u64)(a + l_lo[step & 3]), hi = *(lds_u64)(a + l_hi[step & 3]);
      return as_bf16x8(make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)));
    };
    // a ring of four fragment sets, three steps ahead of the products — across the units too (the image's last unit asks for
    // its own first steps again: the same addresses the next image's first unit needs, requested anew behind the barriers)
    bf16x8 b_buf[4][3];
    auto shl = [](float v, auto n) {  // lane i of a row of 16 <- lane i + n (0 past the row's end)
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + decltype(n)::value, 0xf, 0xf, true));
    };
    asm volatile("" ::"v"(k_bias));  // (its load is waited for HERE, not by a vmcnt(0) in every unit)
    image_loop(F2_LUNITS, std::false_type(), [&](int img, int tt) {
      const uint8_t *ubase = unit_base(tt), *unext = unit_base(tt < F2_LUNITS - 1 ? tt + 1 : tt);
      if (tt == 0) {
#pragma unroll
        for (int st = 0; st < 3; st++)
#pragma unroll
          for (int pc = 0; pc < 3; pc++) b_buf[st][pc] = lfrag(ubase, st, pc);
      }
      f32x4 lacc[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
      for (int step = 0; step < 8; step++) {
#pragma unroll
        for (int pc = 0; pc < 3; pc++) b_buf[(step + 3) & 3][pc] = step + 3 < 8 ? lfrag(ubase, step + 3, pc) : lfrag(unext, step + 3 - 8, pc);
#pragma unroll
        for (int term = 0; term < 6; term++) {
          const int pa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
          const int pw = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
          lacc[step >> 2][term & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LW[pw][step & 3], b_buf[step & 3][pa], lacc[step >> 2][term & 1], 0, 0, 0);
        }
      }
      // the five shifted rows meet: lane (column x, group f) takes kx = 0..3 from its own registers' rows, shifted inside its
      // row of 16 lanes (DPP), and kx = 4 from group f + 2 (the upper half of the wave comes down through a permute)
      float conv_row[2];
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
        const f32x4 t = lacc[rr][0] + lacc[rr][1];
        const float v = ((t[0] + shl(t[1], std::integral_constant<int, 1>())) + shl(t[2], std::integral_constant<int, 2>())) + shl(t[3], std::integral_constant<int, 3>());
        // (ds_bpermute for the move between the wave's halves: v_permlane32_swap gave wrong sums here on the device, the same
        //  arithmetic with the permute passes — not pursued, the step is two instructions per unit)
        conv_row[rr] = v + __shfl(shl(t[0], std::integral_constant<int, 4>()), (lane + 32) & 63);
      }
      float pv = fmaxf(conv_row[0], conv_row[1]);
      pv = fmaxf(pv, shl(pv, std::integral_constant<int, 1>())) + k_bias;
      if (q < 2 && (j & 1) == 0 && j < 12) store_x(img, (tt >> 1) * 12 + 6 * (tt & 1) + (j >> 1), f_own, pv);
    });
    return;
  }
  const int grp = wave < 4 ? 1 + (wave >> 1) : 0;
  // (group 2 is cut 20 : 16, not 18 : 18: the SIMD that hosts the eighth wave runs its 2880 MFMAs ~11 % slower than the others
  //  — measured: without that wave's MFMAs, or with half its units, the kernel drops to the others' time — so its full-group
  //  wave gets two tiles fewer; 19 : 17 and 21 : 15 measured worse)
  const int t0 = wave == 3 ? 20 : wave < 4 ? 18 * (wave & 1) : 12 * (wave - 4), tcnt = wave == 2 ? 20 : wave == 3 ? 16 : wave < 4 ? 18 : 12;
  // the weight fragments of the wave's 16 filters: [piece][k-step]
  bf16x8 W[3][16];
#pragma unroll
  for (int pc = 0; pc < 3; pc++)
#pragma unroll
    for (int ks = 0; ks < 16; ks++) W[pc][ks] = as_bf16x8(btab[((grp * 3 + pc) * 16 + ks) * 64 + lane]);
  const int m_row = (j >> 1) & 1, m_x = 2 * (j >> 2) + (j & 1);
  const int lane_off = m_row * F2_RS + m_x * 40;
  // the four address patterns of the k-slot table (see above): + 40 g (columns), + g rows, + 8 g (channel groups), none
  const int off_x = lane_off + 40 * q, off_y = lane_off + q * F2_RS + 4 * 40, off_z = lane_off + 4 * F2_RS + 4 * 40 + 8 * q,
            off_w = lane_off + 4 * F2_RS + 4 * 40 + 32;
  const int f_own = 16 * grp + j;  // the lane's filter
  const float k_bias = bias[f_own];
  auto tile_base = [&](int tt) {
    const int T = t0 + tt, rp = T / 3, xt = T - 3 * rp;
    return s_img + (2 * rp) * F2_RS + (8 * xt) * 40;
  };
  // the two 8-byte reads of k-step ks for piece pc of the tile at `base`
  auto frag = [&](const uint8_t *base, int ks, int pc) -> bf16x8 {
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
  // the fragments of k-step ks live in a_buf[ks & 1]: the next step's are requested before this step's MFMAs — and behind a
  // tile's last step the FIRST step of the wave's next tile (an LDS round trip per tile otherwise, in which both waves of
  // the SIMD, running in step, leave the matrix pipe idle); the image's last tile asks for its own first step again, which is
  // what the next image's first tile needs: same addresses, other data — so the request is repeated behind the barriers
  bf16x8 a_buf[2][3];
  image_loop(tcnt, std::true_type(), [&](int img, int tt) {
    if (tt == 0) {
#pragma unroll
      for (int pc = 0; pc < 3; pc++) a_buf[0][pc] = frag(tile_base(0), 0, pc);
    }
    const int T = t0 + tt, rp = T / 3, xt = T - 3 * rp;
    const uint8_t *base = tile_base(tt), *base_next = tile_base(tt < tcnt - 1 ? tt + 1 : tt);  // (the last tile re-reads itself: unused)
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // two chains (even / odd terms), added at the end
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
#pragma unroll
      for (int pc = 0; pc < 3; pc++) a_buf[(ks + 1) & 1][pc] = ks + 1 < 16 ? frag(base, ks + 1, pc) : frag(base_next, 0, pc);
#pragma unroll
      for (int term = 0; term < 6; term++) {
        // (activation piece, weight piece): l*h, h*l, m*m, m*h, h*m, h*h — small terms first
        const int pa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
        const int pw = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
        acc[term & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_buf[ks & 1][pa], W[pw][ks], acc[term & 1], 0, 0, 0);
      }
    }
    // pool over the lane's four registers, bias, split for ip1, store
    const f32x4 t = acc[0] + acc[1];
    store_x(img, rp * 12 + 4 * xt + q, f_own, fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])) + k_bias);
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// ip1 on the bf16 matrix pipe: D[image][unit] = sum_k X[image][k] W[k][unit], six piece products per k.
//   X: the flattened pool2 of every image as three bf16 pieces, written by conv2 (k = pixel * 50 + filter, then 96 zeros);
//   W: the weights' three pieces, unit-major, built once at gpd_hip_set_lenet_weights; both BLOCKED (f3_blocked above).
//   A (images): lane (row l & 15, k group l >> 4) = 16 contiguous bytes; B (units) likewise; D: lane (unit l & 15),
//   register r = image 4 (l >> 4) + r.
// The GEMM is small against the chip (5000 x 512 x 7200) and its operands are fat (six bytes per element): with one
// 80 x 128 tile per CU the first version moved 2.3 GB through L2 for 0.11 ms of MFMA work (0.29 ms, MfmaUtil 30 %).  Now
// K is cut into four quarters: a workgroup owns a 32 NT x 256 tile of ONE quarter (NT <= 5: 160 x 256 x 1824 at n = 5000,
// 256 workgroups) and writes partial sums; ip2's kernel adds the four quarters in order (deterministic, and the same for
// every batch size), + bias, ReLU.  Tile traffic halves (1.15 GB), and workgroup L runs on XCD L % 8 = (unit half, quarter):
// every XCD's L2 holds ITS 2.8 MB slice of W for the whole launch.
// Eight waves, two per SIMD, as 2 (images) x 4 (units): a wave owns 16 NT images x 64 units = 4 NT accumulators.  K in steps
// of 32: two LDS stages of 3 x (32 NT + 256) rows x 64 bytes, the four 16-byte chunks of a row XOR-swizzled by
// (4 - (row >> 2)) & 3: every ds_read_b128 of a fragment is bank-conflict free.  Round 6: the stages are filled by LDS-DMA
// from the blocked operands, one step ahead (what the time was made of, by ablation: profiles/NOTES.md §I).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F3_THREADS = 512, F3_BN = 256, F3_BK = 32, F3_KQ = 4;
constexpr int F3_KLEN = F2_XLD / F3_KQ, F3_STEPS = F3_KLEN / F3_BK;  // 1824, 57
static_assert(F2_XLD % (F3_KQ * F3_BK) == 0 && F2_XLD >= kFc1In, "K quarters");
static_assert(F3_STEPS % 2 == 1, "the step loop below runs in pairs + one");

typedef const __attribute__((address_space(1))) void *glds_src_t;
typedef __attribute__((address_space(3))) void *glds_dst_t;

template <int NT>
__global__ __launch_bounds__(F3_THREADS) void fc1_bf16_kernel(const unsigned short *__restrict__ xs, const unsigned short *__restrict__ wt,
                                                             float *__restrict__ out_p, int n) {
  constexpr int BM = 32 * NT;
  constexpr int XB = BM * 64, WB = F3_BN * 64;  // bytes of one piece's tile
  constexpr int STAGE = 3 * (XB + WB);
  // two stages as two objects: the DMA into one and the fragment reads of the other are provably apart
  __shared__ __attribute__((aligned(16))) uint8_t smem_a[STAGE];
  __shared__ __attribute__((aligned(16))) uint8_t smem_b[STAGE];
  static_assert(2 * STAGE <= 160 * 1024, "LDS");
  const int tid = threadIdx.x;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int ut = xcd & 1, kq = xcd >> 1;
  const int m0 = (L >> 3) * BM;
  if (m0 >= n) return;
  const int u0 = ut * F3_BN, k_base = kq * F3_KLEN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 2, wu = wave & 3;
  const int g = lane >> 4, j = lane & 15;
  // Loader: the tiles go from global memory straight into the LDS (global_load_lds_dwordx4: no staging registers, no
  // ds_write pass — the register-staged version of round 5 spent a third of a step moving 98 KB from VGPRs into the LDS,
  // 13 cycles per ds_write_b128, which the matrix instructions of the same SIMD do not overlap).  One instruction copies
  // one 1 KB block of the blocked operands (f3_blocked) = 16 tile rows of 64 bytes, swizzle included: source and LDS image are
  // both lane-linear.  W: 16 row blocks x 3 pieces, a wave takes blocks wave and wave + 8 of every piece; X: 2 NT row blocks
  // x 3 pieces dealt round robin (row blocks past the last image re-read the last one: their sums are not stored).
  constexpr int XBLK = 6 * NT, X_PW = (XBLK + 7) / 8;
  constexpr size_t ROWBLK = (size_t)F3_KSTEPS * 3 * 1024;  // bytes of one row block
  const uint8_t *wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
    wsrc[i] = reinterpret_cast<const uint8_t *>(wt) + (size_t)(u0 / 16 + wave + 8 * i) * ROWBLK + (size_t)(k_base / 32) * 3072 + lane * 16;
  const uint8_t *xsrc[X_PW];
  int xdst[X_PW];
#pragma unroll
  for (int i = 0; i < X_PW; i++) {
    const int e = min(wave + 8 * i, XBLK - 1), rb = e / 3, pc = e - 3 * rb;
    xsrc[i] = reinterpret_cast<const uint8_t *>(xs) + (size_t)min(m0 / 16 + rb, (n - 1) / 16) * ROWBLK + (size_t)(k_base / 32) * 3072 + pc * 1024 + lane * 16;
    xdst[i] = pc * XB + rb * 1024;
  }
  auto fetch = [&](int step, uint8_t *b) {
    const int so = step * 3072;
#pragma unroll
    for (int pc = 0; pc < 3; pc++)
#pragma unroll
      for (int i = 0; i < 2; i++)
        __builtin_amdgcn_global_load_lds((glds_src_t)(wsrc[i] + so + pc * 1024), (glds_dst_t)(b + 3 * XB + pc * WB + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < X_PW; i++)
      if (wave + 8 * i < XBLK) __builtin_amdgcn_global_load_lds((glds_src_t)(xsrc[i] + so), (glds_dst_t)(b + xdst[i]), 16, 0, 0);
  };
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses inside a piece's tile: row * 64 + ((g ^ swizzle(row)) << 4); row & 15 = j for both operands
  const int foff = j * 64 + ((g ^ ((4 - (j >> 2)) & 3)) << 4);
  const int xoff = (wm * 16 * NT) * 64 + foff, woff = (wu * 64) * 64 + foff;
  auto compute = [&](const uint8_t *b) {
    bf16x8 wb[3][4], xa[2][NT];
    auto ldw = [&](int pc) {
#pragma unroll
      for (int c = 0; c < 4; c++) wb[pc][c] = as_bf16x8(*reinterpret_cast<const uint4 *>(b + 3 * XB + pc * WB + woff + c * 16 * 64));
    };
    auto ldx = [&](int s, int pa) {
#pragma unroll
      for (int t = 0; t < NT; t++) xa[s][t] = as_bf16x8(*reinterpret_cast<const uint4 *>(b + pa * XB + xoff + t * 16 * 64));
    };
    auto mul = [&](int s, int pw) {
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[s][t], wb[pw][c], acc[t][c], 0, 0, 0);
    };
    // by image piece, small terms first: l*h; m*m, m*h; h*l, h*m, h*h; the fragments of a group are requested a group ahead
    ldw(0);
    ldx(0, 2);
    ldw(1);
    ldx(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mul(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ldw(2);
    ldx(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mul(1, 1);
    mul(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mul(0, 2);
    mul(0, 1);
    mul(0, 0);
  };
  // Step t multiplies stage t & 1 while the DMA fills the other one with step t + 1; __syncthreads() waits for the wave's own
  // DMA (vmcnt(0): an LDS-DMA is a pending LDS write) before the barrier, so behind it every wave's part of the next stage
  // is in place, and the stage just read is free for step t + 2.
  fetch(0, smem_a);
  __syncthreads();
  for (int t = 0; t < F3_STEPS - 1; t += 2) {
    fetch(t + 1, smem_b);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem_a);
    __syncthreads();
    fetch(t + 2, smem_a);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem_b);
    __syncthreads();
  }
  compute(smem_a);
  // the quarter's partial sums for ip2's kernel, blocked [32 images][quarter][unit][32]: a lane holds four consecutive images
  // of one unit (16 bytes), the 16 units of a tile column are 16 consecutive rows of 128 bytes — and a workgroup of ip2
  // (32 images) reads ONE contiguous 256 KB (rows of the plain [unit][n] transpose lay 4 n bytes apart: 24 us for ip2)
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int u = u0 + 64 * wu + 16 * c + j;
    if (u >= kFc1Out) continue;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int m = m0 + 16 * NT * wm + 16 * t + 4 * g;
      if (m < n) *reinterpret_cast<f32x4 *>(out_p + fc1p_index(m, kq, u)) = acc[t][c];  // (rows past n inside the block: never read back)
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
  auto digit_of = [](long long v, int digit) {
    int d = 0;
    for (int q = 0; q <= digit; q++) {  // balanced digits, least significant first: d in [-128, 127]
      d = (int)(((v + 128) & 255) - 128);
      v = (v - d) >> 8;
    }
    return d;
  };
  const bool narrow = channels <= 4;  // conv1_i8_kernel's NARROW layout: byte 4 i + c of a slot = tap kx = 4 half + i, channel c
  const bool pack12 = channels == 12; // ... PACK12: k-step = kernel row, byte 16 g + b of it = tap kx = (16 g + b) / 12, channel (16 g + b) % 12
  for (int ks = 0; ks < (narrow ? F1N_KS : pack12 ? F1P_KS : F1_KS); ks++)
    for (int mt = 0; mt < F1_MT; mt++)
      for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, g = lane >> 4;
        const int f = 4 * mt + (i >> 2), digit = i & 3;
        uint8_t *row = &atab[((size_t)(ks * F1_MT + mt) * 64 + lane) * 16];
        if (narrow) {
          const int sl = f1n_slot(ks, g);
          if (sl < 0) continue;
          const int ky = sl >> 1, half = sl & 1;
          for (int px = 0; px < 4; px++) {
            const int kx = 4 * half + px;
            if (kx > 4) continue;
            for (int c = 0; c < channels; c++) row[4 * px + c] = (uint8_t)(int8_t)digit_of(W[(size_t)f * K + c * 25 + ky * 5 + kx], digit);
          }
          continue;
        }
        if (pack12) {
          for (int b = 0; b < 16; b++) {
            const int kb = 16 * g + b;
            if (kb >= 60) continue;
            row[b] = (uint8_t)(int8_t)digit_of(W[(size_t)f * K + (kb % 12) * 25 + ks * 5 + kb / 12], digit);
          }
          continue;
        }
        const int tap = f1_tap(ks, g);
        if (tap < 0) continue;
        for (int c = 0; c < channels; c++) row[c] = (uint8_t)(int8_t)digit_of(W[(size_t)f * K + c * 25 + tap], digit);
      }
}

// conv2: the three bf16 pieces of every weight as the MFMA fragments of conv2_bf16_kernel: [slot][piece][k-step][lane] x 8 bf16.
// Slots 0..2: the B fragments of filters 16 slot + (lane & 15).  Slot 3, k-steps 0..3: the A fragments of the wave that computes
// filters 48 and 49 — row lane & 15 = (filter, kernel column): 0..3 (48, kx), 4..7 (49, kx - 4), 8 (48, 4), 12 (49, 4), the
// rest zero; the lane's 8 k = chunk 4 ks + (lane >> 4) = (kernel row ky = chunk / 3, channels 8 (chunk % 3) ..+7), zero from
// channel 20 on and in chunk 15
void lenet_fast_conv2_tables(const float *w, std::vector<unsigned short> &btab) {
  btab.assign((size_t)4 * 3 * 16 * 64 * 8, 0);
  auto put = [&](int slot, int ks, int lane, int e, float v) {
    const Bf3 sp = bf16_split3(v);
    const unsigned short pcs[3] = {sp.h, sp.m, sp.l};
    for (int pc = 0; pc < 3; pc++) btab[((((size_t)slot * 3 + pc) * 16 + ks) * 64 + lane) * 8 + e] = pcs[pc];
  };
  for (int slot = 0; slot < 3; slot++)
    for (int ks = 0; ks < 16; ks++)
      for (int lane = 0; lane < 64; lane++) {
        const int f = 16 * slot + (lane & 15), g = lane >> 4;
        for (int h = 0; h < 2; h++) {
          const int e = ks + 16 * h, tap = f2_slot_tap(e, g), cg = f2_slot_cg(e, g);
          if (tap < 0) continue;
          for (int c = 0; c < 4; c++) put(slot, ks, lane, 4 * h + c, w[(size_t)f * 500 + (4 * cg + c) * 25 + tap]);
        }
      }
  for (int ks = 0; ks < 4; ks++)
    for (int lane = 0; lane < 64; lane++) {
      const int row = lane & 15, chunk = 4 * ks + (lane >> 4);
      if ((row > 8 && row != 12) || chunk > 14) continue;
      const int f = row < 8 ? 48 + (row >> 2) : 48 + ((row - 8) >> 2), kx = row < 8 ? (row & 3) : 4;
      const int ky = chunk / 3, c0 = 8 * (chunk % 3);
      for (int e = 0; e < 8 && c0 + e < 20; e++) put(3, ks, lane, e, w[(size_t)f * 500 + (c0 + e) * 25 + ky * 5 + kx]);
    }
}

// ip1: the reference's file layout is column-major 500 x 7200 == row-major [7200][500] (dense_layer.cpp:7); here unit-major
// bf16 planes [3][512][7232], zero beyond unit 499 / k 7199
void lenet_fast_ip1_tables(const float *w, std::vector<unsigned short> &wt) {
  wt.assign((size_t)3 * 512 * F2_XLD, 0);  // units 500..511 and k 7200..7295: zeros
  uint8_t *base = reinterpret_cast<uint8_t *>(wt.data());
  for (int k = 0; k < kFc1In; k++)
    for (int u = 0; u < kFc1Out; u++) {
      const Bf3 sp = bf16_split3(w[(size_t)k * kFc1Out + u]);
      const unsigned short pcs[3] = {sp.h, sp.m, sp.l};
      for (int pc = 0; pc < 3; pc++) memcpy(base + f3_blocked(u, k, pc), &pcs[pc], 2);
    }
}

// test hook: the blocked X operand (n images) back as three planes [3][n][7200]
void lenet_fast_unblock_x(const unsigned short *blocked, int n, unsigned short *planes) {
  const uint8_t *base = reinterpret_cast<const uint8_t *>(blocked);
  for (int pc = 0; pc < 3; pc++)
    for (int m = 0; m < n; m++)
      for (int k = 0; k < kFc1In; k++) memcpy(planes + ((size_t)pc * n + m) * kFc1In + k, base + f3_blocked(m, k, pc), 2);
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
  const int m_tiles = (n + 32 * NT - 1) / (32 * NT);
  fc1_bf16_kernel<NT><<<m_tiles * 8, F3_THREADS, 0, stream>>>(s.xs, w.fast.f1wt, s.fc1p, n);
}
// image-tile height 32 NT: the smallest (at most 160: two LDS buffers) whose tiles (8 workgroups each: 2 unit halves x 4 K
// quarters) fill the chip's 256 CUs in r whole rounds, r as small as possible
static int fc1f_pick_nt(int n) {
  for (int r = 1;; r++) {
    const int nt = (n + 32 * r * 32 - 1) / (32 * r * 32);
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
  conv2_bf16_kernel<<<grid, F2_THREADS, 0, stream>>>(s.pool1, w.fast.c2b, w.c2b, s.xs, m, queue + 1);
  if (kernel_events) (void)hipEventRecord(kernel_events[1], stream);
  switch (fc1f_pick_nt(m)) {
    case 1: fc1f_launch<1>(w, s, m, stream); break;
    case 2: fc1f_launch<2>(w, s, m, stream); break;
    case 3: fc1f_launch<3>(w, s, m, stream); break;
    case 4: fc1f_launch<4>(w, s, m, stream); break;
    default: fc1f_launch<5>(w, s, m, stream); break;
  }
  // (the four K quarters are added, in order, by ip2's kernel: lenet.hip fc2_score_kernel<true>)
  if (kernel_events) (void)hipEventRecord(kernel_events[2], stream);
  return hipGetLastError();
}

}  // namespace gpd
