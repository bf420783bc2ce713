// Grasp-image generation on gfx950: one workgroup per candidate, image tiles in LDS.
//
// Replaces ImageGenerator::createImages / createImageList (descriptor/image_generator.cpp:17-99),
// ImageStrategy::{transformToUnitImage, findPointsInUnitImage, transformPointsToUnitImage,
// findCellIndices, createNormalsImage, createDepthImage, createShadowImage}
// (descriptor/image_strategy.cpp:32-243), Image{15,12,3}ChannelsStrategy::{createImage,
// calculateImage, calculateChannels} (image_15_channels_strategy.cpp:27-105,
// image_12_channels_strategy.cpp:27-86, image_3_channels_strategy.cpp:27-42,
// image_1_channels_strategy.cpp:25-49) and
// HandSet::{calculateShadow, calculateShadowForCamera, shadowVoxelsToPoints, fastrand}
// (candidate/hand_set.cpp:118-283).
//
// The reference rasterises with per-pixel recurrences whose result depends on the order
// in which points hit a pixel (neighbour order; image_strategy.cpp:130-142, 166-174,
// 202-210).  Here every projection is a counting sort of the in-box points by pixel
// (LDS atomics, any order), after which the thread that owns a pixel orders its short
// segment by neighbour rank — through registers (bitonic network on 8/16/32 keys), the pixels
// handed out longest first — and runs the recurrence sequentially: same arithmetic, same order.  Shadow voxels (hand_set.cpp:202-233 hash set) become a bitset
// over the candidate's voxel AABB: set semantics for free, and walking the bits in index
// order is the lexicographic voxel order the oracle defines.
//
// The 33*N LCG draws of a hand set are generated once per set by shadow_set_kernel with
// jump-ahead (hand_set.cpp:263-266 is an affine map mod 2^32) into a bitset around the sample;
// each candidate extracts the voxels of its own box from it.
//
// Cell indices floor((x/len)/(1/60)) (image_strategy.cpp:92-102) are monotone in x, so
// they are looked up in a table of exact double thresholds computed on the host with the
// same IEEE divisions — bit-identical to dividing, without fp64 divisions in the kernel.
// Only the depth value that enters a pixel's running mean is actually divided.
//
// Device images are planar u8 [n][C][60][60]; the cv::Mat HWC layout of the reference
// interface is produced by planar_to_hwc_kernel when the caller asks for the pixels.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <mutex>
#include <type_traits>

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

constexpr int IMG_THREADS = 512;   // 256 VGPRs per lane: no spills (1024 threads measured equally fast but spilled)
constexpr int SET_THREADS = 1024;  // shadow_set_kernel
constexpr int IMG_WAVES = IMG_THREADS / 64;
constexpr int PT_CAP = 1024;  // in-box points per candidate on the two-per-CU instantiation (mean ~380 on the 3 mm benchmark clouds; entry
                              // indices are packed into 11 bits of the segment words); fuller boxes go to the large instantiation
constexpr int PT_CAP_BIG = 32768;  // fallback instantiation of the points kernel: storage in a global scratch row (its segment table, 64 KB, is what fills the CU's LDS)
constexpr int SH_CAP = 6144;      // in-box shadow voxels per candidate (two workgroups per CU)
constexpr int SH_CAP_BIG = 12288;  // fallback instantiation, one workgroup per CU
// Voxel windows of the shadow kernels.  Default geometry (image box diagonal 0.1233 m, every box of a hand set within
// 0.1233 m of the sample): a 46^3 window per candidate and an 86^3 region per set, the latter an LDS bitset.  WIDE — picked
// on the host for image volumes that do not fit those (images_run: up to ~0.18 m of box diagonal / 0.19 m of reach, e.g.
// volume_width 0.16): 64^3 / 128^3, the set region filled with global-memory atomics (256 KB per set and camera).  The
// reference has no such limits (hand_set.cpp:138); beyond WIDE the kernels still report GPD_ERR_CAPACITY.
template <bool WIDE>
struct Vox {
  static constexpr int VD = WIDE ? 64 : 46;   // candidate window edge (a window row = VD bits along world z, one 64-bit word)
  static constexpr int SD = WIDE ? 128 : 86;  // set region edge: offsets -SR .. SD - SR - 1 from the sample's voxel
  static constexpr int SR = WIDE ? 64 : 43;   // (default: 41.1 voxels of reach -> 42 whole ones, one more on the low side;
                                              //  86^3 bits = 79.5 KB: two shadow_set workgroups per CU)
  static constexpr int LB = 18;               // bits of a voxel position inside the window: iz | iy << 6 | ix << 12 (ascending = lexicographic)
  static constexpr int SETWORDS = (SD * SD * SD + 31) / 32;
};

struct ImgConsts {
  double vol_depth, vol_width, vol_height, half_od, dbl_h;
  int C, nproj, per;
  double shadow_length, voxel, voxel_mult, rand_inv;
  int num_shadow;
  uint32_t stride_a, stride_c;  // LCG jump by SET_THREADS * num_shadow draws
  double len[3];                // box extent per hand axis: vol_depth, vol_width, dbl_h
  double inv_cell[3];           // approximate cells per metre (first guess only)
  double thr[3][kImg + 1];      // thr[a][k] = smallest x with floor((x/len[a])/(1/60)) >= k
  double inv_len[3];            // RN(1/len[a]) for the division by a constant below
  int true_div;                 // 1: the host self-check of div_len failed for these extents, divide for real
  int set_sd, set_sr;           // edge / radius of the voxel region shadow_set_kernel fills around a sample: 86 / 43, 128 / 64 (WIDE) or,
                                // for image volumes beyond those windows, whatever the geometry needs (images_reserve)
};
__constant__ ImgConsts c_img;

struct ImgParams {
  const float *nn;
  int cap;
  const double *centers;
  const gpd_hand *hands;      // [S][slots] records of the search
  const int32_t *cand_hand;   // [n] record of a candidate (plan_kernel)
  const int32_t *meta;    // [n][4]: sample slot, N_images, first shadow bitset (< 0: no shadow), number of bitsets (cameras)
  const uint32_t *set_bits;  // [live sets][Vox::SETWORDS] shadow voxel bitsets (shadow_set_kernel)
  uint8_t *images;        // planar [n][C][3600]
  int32_t *status;
  int num_cand;              // candidates of the launch (the kernels that do not take a list)
  const int32_t *cand_list;  // large instantiations: the queued candidates (nullptr: all, in XCD-aware order) ...
  const int32_t *cand_count; // ... and how many there are (device side: nothing waits for the count)
  int32_t *overflow_list;    // shadow kernel: candidates whose box exceeds SHC voxels
  int32_t *overflow_count;
  unsigned long long *dbg;  // profiling aid (GPD_IMG_TIMING=1): per-phase cycle sums
  int32_t *pts_overflow_list;   // points kernel: candidates with more than PT_CAP in-box points
  int32_t *pts_overflow_count;
  char *pts_scratch;            // fallback instantiation: PTS_SCRATCH_BYTES per listed candidate
  int exit_after;               // read by the instrumented build only (profiles/img_exits.patch: leave the kernel after phase k)
  char *huge_scratch;           // shadow_image_any_kernel: HUGE_SCRATCH_BYTES per workgroup of its grid
};

struct Box {
  double F[9];
  double sample[3];
  double off[3];  // x_a = t_a - off[a]: bottom, center - half_od, -vol_height
  double lo[3], hi[3];
};

// points kernel: in-box points cached once, one projection's normals (3 planes) + depth at a time.
// BIG (the overflow fallback, up to PT_CAP_BIG in-box points): the point arrays live in a global
// scratch row instead of LDS.
constexpr size_t PTS_SCRATCH_BYTES = (size_t)PT_CAP_BIG * (3 * sizeof(double) + sizeof(float4) + sizeof(uint32_t));  // + the in-box points' neighbour indices, 32 bits wide
struct PointArrays {
  double t[3][PT_CAP];  // hand-frame coordinates of the in-box points
  float4 an[PT_CAP];    // |normal| in the hand frame (x, y, z) and, as bits in w, the cell key
                        // cx | cy << 6 | cz << 12: one 16-byte read per visit
};
struct NoPointArrays {};
// Under 80 KB for the small instantiation, so that two workgroups — two grasp_image ones, or one of them next to a
// shadow_image one of another stream — share a CU (the kernels wait on LDS round trips and barriers most of their
// time; the second resident workgroup fills those gaps).  What made it 155 KB before: three normal planes + a depth
// plane held at once (57.6 KB) and 2048-point arrays.  Now the walks leave their four values per NON-EMPTY pixel
// (at most one per in-box point) and the planes are rebuilt one after the other in the storage of the segment
// counters, dilated straight into registers.
template <bool BIG>
struct __attribute__((aligned(16))) SmemPts {
  typename std::conditional<BIG, NoPointArrays, PointArrays>::type p;
  __attribute__((aligned(16))) uint32_t cells[kPix];   // (segment start << 16) | count of a pixel; after the walks: the index raster, then the staged bytes
  uint16_t place[BIG ? PT_CAP_BIG : PT_CAP];  // segment table: entry numbers (entries are numbered in neighbour order)
  float4 nzv[(BIG ? kPix : PT_CAP) + 1];      // slot 0: the empty pixel (zeros); slot q + 1: the three normal values and the
                                              // depth value of the q-th non-empty pixel
  uint16_t nzlist[BIG ? kPix : PT_CAP];       // the non-empty pixels, longest segment first
  double thr[3][kImg + 1];
  double recip[128];  // 1.0 / k
  float red_f[4 * IMG_WAVES];
  int red_i[IMG_WAVES];
  int counter;
  int flag;
};
typedef SmemPts<false> Smem;
// shadow kernel: kept under 80 KB (SHC = 6144) so that two workgroups share a CU
template <int SHC, bool WIDE>
struct __attribute__((aligned(16))) SmemShadow {
  __attribute__((aligned(16))) float raster0[kPix];
  __attribute__((aligned(16))) uint32_t cells[kPix];
  uint32_t lin[SHC];  // set bits inside the box, ascending: voxel position (iz | iy << 6 | ix << 12) | cell x << LB | cell y << (LB + 6)
  union {
    struct {                        // while the list is built: per row of the voxel window (a line along world z) ...
      uint16_t rowbase[Vox<WIDE>::VD * Vox<WIDE>::VD];  // ... where its in-box voxels start in `lin`
      uint8_t rowcnt[Vox<WIDE>::VD * Vox<WIDE>::VD];    // ... and how many they are
    } rows;
    uint16_t place[SHC];  // afterwards: segment table of the counting sort
  } bp;
  uint16_t nz[kPix];
  uint8_t cz[SHC];  // cell z of the list entries
  double thr[3][kImg + 1];
  double recip[128];
  float red_f[4 * IMG_WAVES];
  int red_i[IMG_WAVES];
  int vorg[3];
  int flag;
};

// a workgroup-uniform double moved to scalar registers (the candidate's box is the same for
// all 1024 lanes; keeping its 21 doubles in SGPRs frees ~40 VGPRs per lane)
__device__ inline double uniform_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// wave-wide max / min of a float, the result in every lane: four DPP steps inside the rows of 16 lanes, two row
// broadcasts, one v_readlane — 7 VALU instructions where six __shfl_xor steps are 6 ds_bpermute + ~25 VALU.  (A lane
// whose DPP source is masked keeps its own value: max(v, v) = v.)
template <class OP>
__device__ inline float wave_reduce_f32(float v, OP op) {
  auto dpp = [](float x, auto ctrl, auto rmask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
  };
  using std::integral_constant;
  v = op(v, dpp(v, integral_constant<int, 0xB1>(), integral_constant<int, 0xf>()));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, integral_constant<int, 0x4E>(), integral_constant<int, 0xf>()));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, integral_constant<int, 0x141>(), integral_constant<int, 0xf>()));  // row_half_mirror
  v = op(v, dpp(v, integral_constant<int, 0x140>(), integral_constant<int, 0xf>()));  // row_mirror
  v = op(v, dpp(v, integral_constant<int, 0x142>(), integral_constant<int, 0xa>()));  // row_bcast:15 into rows 1, 3
  v = op(v, dpp(v, integral_constant<int, 0x143>(), integral_constant<int, 0xc>()));  // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline float wave_max_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ inline float wave_min_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fminf(a, b); }); }

// wave-wide inclusive prefix sum of an int: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then the row totals carried
// over by row_bcast:15 / :31 — the sequence LLVM's atomic optimiser emits for gfx9; lanes without a source add 0
__device__ inline int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// hand-frame coordinates of a world point: t = F^T (w - sample)  (image_strategy.cpp:36-40)
__device__ inline void to_hand(const Box &B, double w0, double w1, double w2, double t[3]) {
  const double c0 = w0 - B.sample[0], c1 = w1 - B.sample[1], c2 = w2 - B.sample[2];
  t[0] = B.F[0] * c0 + B.F[3] * c1 + B.F[6] * c2;
  t[1] = B.F[1] * c0 + B.F[4] * c1 + B.F[7] * c2;
  t[2] = B.F[2] * c0 + B.F[5] * c1 + B.F[8] * c2;
}
// findPointsInUnitImage (image_strategy.cpp:53-70): strict box test
__device__ inline bool in_box(const Box &B, const double t[3]) {
  return (t[0] > B.lo[0]) && (t[0] < B.hi[0]) && (t[1] > B.lo[1]) && (t[1] < B.hi[1]) && (t[2] > B.lo[2]) && (t[2] < B.hi[2]);
}
// min(floor(((t - off)/len)/(1/60)), 59) by exact thresholds (see file header)
template <class SM>
__device__ inline int cell_coord(const SM &S, int axis, double x) {
  int k = (int)(x * c_img.inv_cell[axis]);
  k = k < 0 ? 0 : (k > kImg - 1 ? kImg - 1 : k);
  // both neighbours of the guess at once (one LDS round trip; the guess is off by one at most, the loops are for the proof)
  const double t0 = S.thr[axis][k], t1 = S.thr[axis][k + 1];
  if (k > 0 && x < t0) {
    k--;
    while (k > 0 && x < S.thr[axis][k]) k--;
  } else if (k < kImg - 1 && x >= t1) {
    k++;
    while (k < kImg - 1 && x >= S.thr[axis][k + 1]) k++;
  }
  return k;
}
template <class SM>
__device__ inline uint32_t cells_of(const SM &S, const Box &B, const double t[3]) {
  const int cx = cell_coord(S, 0, t[0] - B.off[0]);
  const int cy = cell_coord(S, 1, t[1] - B.off[1]);
  const int cz = cell_coord(S, 2, t[2] - B.off[2]);
  return (uint32_t)cx | ((uint32_t)cy << 6) | ((uint32_t)cz << 12);
}
// projections by cumulative row swaps 0<->2 then 1<->2: (x,y,z), (z,y,x), (z,x,y);
// cell = horizontal + 60 * vertical (image_strategy.cpp:92-102)
__device__ inline int cell_of_key(uint32_t key, int pr) {
  const int cx = key & 63, cy = (key >> 6) & 63, cz = (key >> 12) & 63;
  const int v = pr == 0 ? cx : cz;
  const int h = pr == 2 ? cx : cy;
  return h + v * kImg;
}
__device__ inline int depth_axis(int pr) { return pr == 0 ? 2 : (pr == 1 ? 0 : 1); }

// x / len[axis], correctly rounded, without a division: q = x*y, r = x - len*q (exact, FMA),
// q' = q + r*y with y = RN(1/len) (Markstein).  Bit-identical to the IEEE quotient for the
// image extents; images_run() checks that on the host and falls back to a real division.
__device__ inline double div_len(double x, int axis) {
  const ImgConsts &K = c_img;
  if (K.true_div) return x / K.len[axis];
  const double y = K.inv_len[axis];
  const double q = x * y;
  const double r = __builtin_fma(-K.len[axis], q, x);
  return __builtin_fma(r, y, q);
}
// 1.0 / (double)count for the running means (image_strategy.cpp:170, 206)
template <int NTAB>
__device__ inline double recip_count(const double *tab, float fc) {
  const int k = (int)fc;
  return k < NTAB ? tab[k] : 1.0 / (double)fc;
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

// block-wide exclusive scan of one int per thread; returns the exclusive prefix, total in *total
template <class SM>
__device__ inline int block_excl_scan(SM &S, int v, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_incl_scan_i32(v);
  __syncthreads();
  if (lane == 63) S.red_i[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < IMG_WAVES; w++) {
    const int x = S.red_i[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

// exclusive scan of cells[].count into the start field; returns the total
template <class SM>
__device__ int scan_cells(SM &S) {
  const int tid = threadIdx.x;
  constexpr int PER = 8;  // consecutive cells per thread: two 16-byte LDS accesses each way (450 threads hold the 3600 cells)
  static_assert(kPix % PER == 0 && kPix / PER <= IMG_THREADS, "scan_cells: cells per thread");
  uint4 *c4 = reinterpret_cast<uint4 *>(S.cells);
  if (tid < 32) reinterpret_cast<int *>(S.red_f)[tid] = 0;  // the histogram of list_nonempty_cells (three barriers from here)
  uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
  if (tid < kPix / PER) {
    a = c4[2 * tid];
    b = c4[2 * tid + 1];
  }
  const int c[PER] = {(int)(a.x & 0xffffu), (int)(a.y & 0xffffu), (int)(a.z & 0xffffu), (int)(a.w & 0xffffu),
                      (int)(b.x & 0xffffu), (int)(b.y & 0xffffu), (int)(b.z & 0xffffu), (int)(b.w & 0xffffu)};
  int sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) sum += c[k];
  int total;
  int run = block_excl_scan(S, sum, &total);
  uint32_t o[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    o[k] = (uint32_t)(run < 0xffff ? run : 0xffff) << 16;
    run += c[k];
  }
  if (tid < kPix / PER) {
    c4[2 * tid] = make_uint4(o[0], o[1], o[2], o[3]);
    c4[2 * tid + 1] = make_uint4(o[4], o[5], o[6], o[7]);
  }
  __syncthreads();
  return total;
}

// list of the non-empty cells, ordered by their point count, descending (counting sort into 32
// buckets; counts >= 31 share the first).  The walks give one pixel to one lane, so a wave runs as
// long as its longest pixel: with equal lengths side by side the waves execute ~40 % fewer
// (mostly idle) iterations than in cell order.  Pixels are independent, their order is free.
// Returns the number of non-empty cells.
template <class SM>
__device__ int list_nonempty_cells(SM &S, uint16_t *nz) {
  const int tid = threadIdx.x;
  constexpr int PER = (kPix + IMG_THREADS - 1) / IMG_THREADS;
  int *hist = reinterpret_cast<int *>(S.red_f);  // 32 counters, cleared by scan_cells; red_f is idle until the final phase
  static_assert(PER == 8, "two 16-byte reads per thread");
  uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
  if (tid < kPix / PER) {
    a = reinterpret_cast<const uint4 *>(S.cells)[2 * tid];
    b = reinterpret_cast<const uint4 *>(S.cells)[2 * tid + 1];
  }
  const uint32_t cw[PER] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  int bucket[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int cn = (int)(cw[k] & 0xffffu);
    bucket[k] = cn ? 31 - (cn < 31 ? cn : 31) : -1;
    if (cn) atomicAdd(&hist[bucket[k]], 1);
  }
  __syncthreads();
  if (tid < 64) {
    const int v = tid < 32 ? hist[tid] : 0;
    const int incl = wave_incl_scan_i32(v);
    if (tid < 32) hist[tid] = incl - v;
    if (tid == 31) S.red_i[0] = incl;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++)
    if (bucket[k] >= 0) nz[atomicAdd(&hist[bucket[k]], 1)] = (uint16_t)(tid * PER + k);
  __syncthreads();
  return S.red_i[0];
}

// 3x3 rect max-dilate (border ignored), NORM_MINMAX to [0,1], u8 = round-half-even(v*255)
// (image_strategy.cpp:144-153, 178-187, 221-230; cv::dilate / cv::normalize / convertTo).
// Planes are in cell-index order (cell row = 59 - image row; the 3x3 window is symmetric).
// NPL planes are finished in one pass; planes with the same norm group share min/max (the three
// normal channels are normalised jointly, depth/shadow on their own).  Each thread owns groups of
// 4 consecutive pixels: three 16-byte LDS reads + two edge reads per row (indices clamped to the
// image: a clamped duplicate cannot change a max), one dword store.
// planes 0..2 (or 0 alone) start at p012 and are kPix apart; plane 3, if present, is p3 and is
// normalised on its own; output channels are consecutive planes starting at out.
template <int NPL, class SM>
__device__ void finalize_planes(SM &S, const float *p012, const float *p3, uint8_t *out) {
  const int tid = threadIdx.x;
  constexpr int GROUPS = NPL * 900;
  constexpr int PER = (GROUPS + IMG_THREADS - 1) / IMG_THREADS;
  float d[PER][4];
  float mn[2] = {FLT_MAX, FLT_MAX}, mx[2] = {-FLT_MAX, -FLT_MAX};
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int g = tid + k * IMG_THREADS;
    if (g < GROUPS) {
      const int ch = g / 900, rem = g - ch * 900;
      const int r = rem / 15, c0 = (rem - r * 15) * 4;
      const float *pl = (ch < 3) ? p012 + ch * kPix : p3;
      const int rm = r > 0 ? r - 1 : 0, rp = r < kImg - 1 ? r + 1 : kImg - 1;
      const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + 4 < kImg ? c0 + 4 : kImg - 1;
      const float4 a = *reinterpret_cast<const float4 *>(pl + rm * kImg + c0);
      const float4 b = *reinterpret_cast<const float4 *>(pl + r * kImg + c0);
      const float4 c = *reinterpret_cast<const float4 *>(pl + rp * kImg + c0);
      const float el = fmaxf(fmaxf(pl[rm * kImg + cl], pl[r * kImg + cl]), pl[rp * kImg + cl]);
      const float er = fmaxf(fmaxf(pl[rm * kImg + cr], pl[r * kImg + cr]), pl[rp * kImg + cr]);
      const float m0 = fmaxf(fmaxf(a.x, b.x), c.x), m1 = fmaxf(fmaxf(a.y, b.y), c.y);
      const float m2 = fmaxf(fmaxf(a.z, b.z), c.z), m3 = fmaxf(fmaxf(a.w, b.w), c.w);
      d[k][0] = fmaxf(fmaxf(el, m0), m1);
      d[k][1] = fmaxf(fmaxf(m0, m1), m2);
      d[k][2] = fmaxf(fmaxf(m1, m2), m3);
      d[k][3] = fmaxf(fmaxf(m2, m3), er);
      const float lo = fminf(fminf(d[k][0], d[k][1]), fminf(d[k][2], d[k][3]));
      const float hi = fmaxf(fmaxf(d[k][0], d[k][1]), fmaxf(d[k][2], d[k][3]));
      if (ch < 3) {
        mn[0] = fminf(mn[0], lo);
        mx[0] = fmaxf(mx[0], hi);
      } else {
        mn[1] = fminf(mn[1], lo);
        mx[1] = fmaxf(mx[1], hi);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    mn[q] = wave_min_f32(mn[q]);
    mx[q] = wave_max_f32(mx[q]);
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    S.red_f[4 * (tid >> 6) + 0] = mn[0];
    S.red_f[4 * (tid >> 6) + 1] = mx[0];
    S.red_f[4 * (tid >> 6) + 2] = mn[1];
    S.red_f[4 * (tid >> 6) + 3] = mx[1];
  }
  __syncthreads();
  float fs[2], fb[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    float a = S.red_f[2 * q], b = S.red_f[2 * q + 1];
#pragma unroll
    for (int w = 1; w < IMG_WAVES; w++) {
      a = fminf(a, S.red_f[4 * w + 2 * q]);
      b = fmaxf(b, S.red_f[4 * w + 2 * q + 1]);
    }
    const double smin = (double)a, smax = (double)b;
    const double scale = 1.0 * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
    const double shift = 0.0 - smin * scale;
    fs[q] = (float)scale;
    fb[q] = (float)shift;
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int g = tid + k * IMG_THREADS;
    if (g < GROUPS) {
      const int ch = g / 900, rem = g - ch * 900;
      const int r = rem / 15, c0 = (rem - r * 15) * 4;
      const int q = ch < 3 ? 0 : 1;
      uint32_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float v = d[k][j] * fs[q] + fb[q];
        const float u = v * 255.0f + 0.0f;
        float t = rintf(u);
        t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
        packed |= (uint32_t)(int)t << (8 * j);
      }
      // image row = 59 - cell row (image_strategy.cpp:128-129)
      *reinterpret_cast<uint32_t *>(out + (size_t)ch * kPix + (kImg - 1 - r) * kImg + c0) = packed;
    }
  }
  __syncthreads();
}

// ---- one plane at a time (grasp_image_kernel): the same arithmetic as finalize_planes, with the dilated values
//      of a thread's pixel groups kept in registers across the planes that are normalised together
constexpr int GPT = (900 + IMG_THREADS - 1) / IMG_THREADS;  // groups of 4 pixels per thread and plane
__device__ inline void dilate_plane(const float *pl, float (&d)[GPT][4], float &mn, float &mx) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < GPT; k++) {
    const int g = tid + k * IMG_THREADS;
    if (g < 900) {
      const int r = g / 15, c0 = (g - r * 15) * 4;
      const int rm = r > 0 ? r - 1 : 0, rp = r < kImg - 1 ? r + 1 : kImg - 1;
      const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + 4 < kImg ? c0 + 4 : kImg - 1;
      const float4 a = *reinterpret_cast<const float4 *>(pl + rm * kImg + c0);
      const float4 b = *reinterpret_cast<const float4 *>(pl + r * kImg + c0);
      const float4 c = *reinterpret_cast<const float4 *>(pl + rp * kImg + c0);
      const float el = fmaxf(fmaxf(pl[rm * kImg + cl], pl[r * kImg + cl]), pl[rp * kImg + cl]);
      const float er = fmaxf(fmaxf(pl[rm * kImg + cr], pl[r * kImg + cr]), pl[rp * kImg + cr]);
      const float m0 = fmaxf(fmaxf(a.x, b.x), c.x), m1 = fmaxf(fmaxf(a.y, b.y), c.y);
      const float m2 = fmaxf(fmaxf(a.z, b.z), c.z), m3 = fmaxf(fmaxf(a.w, b.w), c.w);
      d[k][0] = fmaxf(fmaxf(el, m0), m1);
      d[k][1] = fmaxf(fmaxf(m0, m1), m2);
      d[k][2] = fmaxf(fmaxf(m1, m2), m3);
      d[k][3] = fmaxf(fmaxf(m2, m3), er);
      mn = fminf(mn, fminf(fminf(d[k][0], d[k][1]), fminf(d[k][2], d[k][3])));
      mx = fmaxf(mx, fmaxf(fmaxf(d[k][0], d[k][1]), fmaxf(d[k][2], d[k][3])));
    }
  }
}
// block-wide min / max -> the scale and shift of cv::normalize(NORM_MINMAX) as floats (image_strategy.cpp:144-153)
template <class SM>
__device__ inline void minmax_scale(SM &S, float mn, float mx, float &fs, float &fb) {
  const int tid = threadIdx.x;
  mn = wave_min_f32(mn);
  mx = wave_max_f32(mx);
  __syncthreads();
  if ((tid & 63) == 0) {
    S.red_f[2 * (tid >> 6)] = mn;
    S.red_f[2 * (tid >> 6) + 1] = mx;
  }
  __syncthreads();
  float a = S.red_f[0], b = S.red_f[1];
#pragma unroll
  for (int w = 1; w < IMG_WAVES; w++) {
    a = fminf(a, S.red_f[2 * w]);
    b = fmaxf(b, S.red_f[2 * w + 1]);
  }
  const double smin = (double)a, smax = (double)b;
  const double scale = 1.0 * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
  const double shift = 0.0 - smin * scale;
  fs = (float)scale;
  fb = (float)shift;
}
__device__ inline void store_plane(const float (&d)[GPT][4], float fs, float fb, uint8_t *out) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < GPT; k++) {
    const int g = tid + k * IMG_THREADS;
    if (g < 900) {
      const int r = g / 15, c0 = (g - r * 15) * 4;
      uint32_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float v = d[k][j] * fs + fb;
        const float u = v * 255.0f + 0.0f;
        float t = rintf(u);
        t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
        packed |= (uint32_t)(int)t << (8 * j);
      }
      // image row = 59 - cell row (image_strategy.cpp:128-129)
      *reinterpret_cast<uint32_t *>(out + (kImg - 1 - r) * kImg + c0) = packed;
    }
  }
}

template <int N>
__device__ inline void sort_u16_regs(uint16_t *p, int n);
__device__ inline void sort_u16(uint16_t *p, int n) {
  if (n <= 8) return sort_u16_regs<8>(p, n);
  if (n <= 16) return sort_u16_regs<16>(p, n);
  if (n <= 32) return sort_u16_regs<32>(p, n);
  for (int i = 1; i < n; i++) {
    const uint16_t v = p[i];
    int j = i - 1;
    while (j >= 0 && p[j] > v) {
      p[j + 1] = p[j];
      j--;
    }
    p[j + 1] = v;
  }
}
// The same with at most 16 keys in registers at a time (grasp_image_kernel is held to 128 VGPRs; the 32-key network
// alone costs it 21 spilled registers = 0.2 GB of scratch traffic per 5000 candidates).  17..32 keys: both halves
// through the 16-key network, the cross step of the bitonic merge element by element through LDS, then each half —
// a bitonic sequence now — through the 16-key merge network.  Same 240 compare-exchanges, five LDS round trips
// instead of one, and only for the few pixels that hold more than 16 points.
template <int N>
__device__ inline void merge_regs(uint32_t (&k)[N]) {  // bitonic -> ascending
#pragma unroll
  for (int stride = N >> 1; stride > 0; stride >>= 1) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = i ^ stride;
      if (j > i) {
        const uint32_t lo = min(k[i], k[j]), hi = max(k[i], k[j]);
        k[i] = lo;
        k[j] = hi;
      }
    }
  }
}
template <int N>
__device__ inline void sort_regs(uint32_t (&k)[N]);
__device__ inline void sort_u16_lean(uint16_t *p, int n) {
  if (n <= 8) return sort_u16_regs<8>(p, n);
  if (n <= 16) return sort_u16_regs<16>(p, n);
  if (n > 32) return sort_u16(p, n);
  uint32_t k[16];
#pragma unroll
  for (int q = 0; q < 16; q++) k[q] = p[q];
  sort_regs<16>(k);
#pragma unroll
  for (int q = 0; q < 16; q++) p[q] = (uint16_t)k[q];
#pragma unroll
  for (int q = 0; q < 16; q++) k[q] = 16 + q < n ? (uint32_t)p[16 + q] : 0xffffffffu;
  sort_regs<16>(k);
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (16 + q < n) p[16 + q] = (uint16_t)k[q];
  // cross step: element i against element 31 - i (missing ones are +inf and stay where they are)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (31 - i < n) {
      const uint16_t a = p[i], b = p[31 - i];
      p[i] = a < b ? a : b;
      p[31 - i] = a < b ? b : a;
    }
  }
#pragma unroll
  for (int q = 0; q < 16; q++) k[q] = p[q];
  merge_regs<16>(k);
#pragma unroll
  for (int q = 0; q < 16; q++) p[q] = (uint16_t)k[q];
#pragma unroll
  for (int q = 0; q < 16; q++) k[q] = 16 + q < n ? (uint32_t)p[16 + q] : 0xffffffffu;
  merge_regs<16>(k);
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (16 + q < n) p[16 + q] = (uint16_t)k[q];
}
// N keys in registers, ascending (bitonic network, fully unrolled: 24 / 80 / 240 compare-exchanges
// for N = 8 / 16 / 32).  The walks order their short per-pixel segments through registers: N
// independent LDS reads, the network, N writes — one LDS round trip instead of the dependent
// read-compare-write chain of an insertion sort (~300 cycles per step, O(n^2) steps).
template <int N>
__device__ inline void sort_regs(uint32_t (&k)[N]) {
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int i = 0; i < N; i++) {
        const int j = i ^ stride;
        if (j > i) {
          const uint32_t lo = min(k[i], k[j]), hi = max(k[i], k[j]);
          const bool up = (i & size) == 0;
          k[i] = up ? lo : hi;
          k[j] = up ? hi : lo;
        }
      }
    }
  }
}
template <int N>
__device__ inline void sort_u16_regs(uint16_t *p, int n) {
  uint32_t k[N];
#pragma unroll
  for (int q = 0; q < N; q++) k[q] = q < n ? (uint32_t)p[q] : 0xffffffffu;
  sort_regs<N>(k);
#pragma unroll
  for (int q = 0; q < N; q++)
    if (q < n) p[q] = (uint16_t)k[q];
}
#define TICK(k)                                                     \
  do {                                                              \
    if (P.dbg && tid == 0) {                                        \
      const unsigned long long now_ = __builtin_readcyclecounter(); \
      atomicAdd(&P.dbg[k], now_ - t_last);                          \
      t_last = now_;                                                \
    }                                                               \
  } while (0)

// the candidate's box in scalar registers, bounds exactly as findPointsInUnitImage /
// transformPointsToUnitImage evaluate them (image_strategy.cpp:53-90)
__device__ inline void load_box(const gpd_hand &H, Box &B) {
  const ImgConsts &K = c_img;
#pragma unroll
  for (int i = 0; i < 9; i++) B.F[i] = uniform_f64(H.frame[i]);
#pragma unroll
  for (int i = 0; i < 3; i++) B.sample[i] = uniform_f64(H.sample[i]);
  const double hb = uniform_f64(H.bottom), hc = uniform_f64(H.center);
  B.off[0] = hb;
  B.off[1] = uniform_f64(hc - K.half_od);
  B.off[2] = uniform_f64(-K.vol_height);  // (t2 + height) == t2 - (-height)
  B.lo[0] = hb;
  B.hi[0] = uniform_f64(hb + K.vol_depth);
  B.lo[1] = B.off[1];
  B.hi[1] = uniform_f64(hc + K.half_od);
  B.lo[2] = uniform_f64(-1.0 * K.vol_height);
  B.hi[2] = uniform_f64(K.vol_height);
}

// ---------------------------------------------------------------------------
// shadow_image_kernel: the shadow channel of the three projections of one candidate
// (createShadowImage, image_strategy.cpp:192-233; 15 channels only).
// ---------------------------------------------------------------------------
// XCD-aware candidate order (workgroup L runs on XCD L % 8, one L2 per XCD): XCD x takes the x-th
// eighth of the candidate list, so the candidates of one hand set — which read the same neighbourhood
// rows / the same shadow bitset — meet in ONE L2 instead of being dealt out over all eight.
// Launch with 8 * ceil(n / 8) workgroups; returns -1 for the padding ones.
__device__ __forceinline__ int xcd_candidate(int n) {
  const int per = (n + 7) >> 3;
  const int L = blockIdx.x, slot = L >> 3;
  const int cand = (L & 7) * per + slot;
  return cand < n ? cand : -1;
}

template <int SHC, bool WIDE>
__device__ __forceinline__ void shadow_image_body(const ImgParams &P, SmemShadow<SHC, WIDE> &S, const int cand) {
  constexpr int VDIM = Vox<WIDE>::VD, SD = Vox<WIDE>::SD, SR = Vox<WIDE>::SR, LB = Vox<WIDE>::LB, SETWORDS = Vox<WIDE>::SETWORDS;
  unsigned long long t_last = __builtin_readcyclecounter();
  const ImgConsts &K = c_img;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int set_ord = P.meta[4 * cand + 2];
  const int set_nb = P.meta[4 * cand + 3];
  uint8_t *out = P.images + (size_t)cand * kPix * K.C;
  Box B;
  load_box(P.hands[P.cand_hand[cand]], B);
  for (int i = tid; i < 3 * (kImg + 1); i += IMG_THREADS) (&S.thr[0][0])[i] = (&K.thr[0][0])[i];
  for (int i = tid; i < 128; i += IMG_THREADS) S.recip[i] = i ? 1.0 / (double)i : 0.0;
  if (tid == 0) {
    S.flag = 0;
    // voxel AABB of the image box: corners sample + F * (bx, by, bz)
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const double bx = (k & 1) ? B.hi[0] : B.lo[0];
      const double by = (k & 2) ? B.hi[1] : B.lo[1];
      const double bz = (k & 4) ? B.hi[2] : B.lo[2];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double w = B.sample[a] + B.F[3 * a] * bx + B.F[3 * a + 1] * by + B.F[3 * a + 2] * bz;
        lo[a] = fmin(lo[a], w);
        hi[a] = fmax(hi[a], w);
      }
    }
    // capacity: the box must fit the VDIM^3 voxel window of this kernel and the SD^3 region that
    // shadow_set_kernel voxelised around the sample; a larger image volume is reported
    // (GPD_ERR_CAPACITY, flag 1) instead of silently losing shadow voxels
    int bad = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int v0 = (int)floor(lo[a] * K.voxel_mult) - 1, v1 = (int)floor(hi[a] * K.voxel_mult) + 1;
      const int o = (int)floor(B.sample[a] * K.voxel_mult) - SR;
      S.vorg[a] = v0;
      // (the window's one-voxel margins may stick out of the region: they hold no voxel of the box)
      if (v1 - v0 >= VDIM || v0 + 1 < o || v1 - 1 >= o + SD) bad = 1;
    }
    S.flag = bad;
  }
  __syncthreads();
  const int x0 = S.vorg[0], y0 = S.vorg[1], z0 = S.vorg[2];
  // ---- the set's shadow voxels (shadow_set_kernel) restricted to this candidate's box:
  //      walk the AABB rows of the set bitset(s), exact f64 box test per set voxel.  A thread keeps the in-box voxels of
  //      its rows (row = tid + k * 512: neighbouring rows go to neighbouring lanes) as bit masks in registers — no
  //      candidate bitset in LDS, no atomics, nothing to zero or to count again.
  constexpr int NROWS = VDIM * VDIM, RPT = (NROWS + IMG_THREADS - 1) / IMG_THREADS;
  unsigned long long mask[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) mask[k] = 0ull;
  if (set_ord >= 0) {
    // INVARIANT of the bitset rows: shadow_set_kernel<0> writes exactly the union of the windows floor(min) - 1 .. floor(max) + 1
    // that its candidates' boxes open (the same arithmetic as S.vorg above); the fixed VDIM x VDIM walk below can leave that
    // union, and what it reads there is zero (images_reserve clears a new allocation) or voxels of an earlier launch's set.
    // Neither can be kept: a bit survives only the exact f64 box test of its voxel, and no voxel outside
    // [floor(min), floor(max)] lies in the box.  A reader that counts or uses bits BEFORE that test must clamp to the union.
    const uint32_t *sb = P.set_bits + (size_t)set_ord * SETWORDS;
    const int ox = (int)floor(B.sample[0] * K.voxel_mult) - SR, oy = (int)floor(B.sample[1] * K.voxel_mult) - SR,
              oz = (int)floor(B.sample[2] * K.voxel_mult) - SR;
    const int zlo = z0 - oz;
    // t_a(z + 1) - t_a(z) = F[6 + a] * voxel, the same for every row
    double invBz[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double bz = B.F[6 + a] * K.voxel;
      invBz[a] = uniform_f64(fabs(bz) > 1e-9 ? 1.0 / bz : 0.0);
    }
    // 46 bits starting at rowbit + zlo, clipped to the row [0, SD): the same z range for every row
    const int za = zlo < 0 ? 0 : zlo, zb = (zlo + VDIM < SD) ? zlo + VDIM : SD;
    const int nbits = zb - za;
    // The row intervals below need the hand coordinates of a row's first voxel to ~1e-12 m only (EPS), so they come from
    // the affine form tb_a = T0_a + ix * Ax_a + iy * Ay_a and the slab faces in z-steps from one face + the slab's width
    // (24 f64 operations + two min / max per slab before); every number here is the same for the whole workgroup.
    double T0[3], Ax[3], Ay[3], Lf[3], Wd[3];
    {
      const double wx = (double)x0 * K.voxel - B.sample[0], wy = (double)y0 * K.voxel - B.sample[1],
                   wz = (double)(za - zlo + z0) * K.voxel - B.sample[2];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        T0[a] = uniform_f64(B.F[a] * wx + B.F[3 + a] * wy + B.F[6 + a] * wz);
        Ax[a] = uniform_f64(B.F[a] * K.voxel);
        Ay[a] = uniform_f64(B.F[3 + a] * K.voxel);
        Lf[a] = invBz[a] > 0.0 ? B.lo[a] : B.hi[a];                       // the face the row crosses first
        Wd[a] = uniform_f64(fabs((B.hi[a] - B.lo[a]) * invBz[a]));        // z-steps from that face to the other
      }
    }
    // Pass 1: the bit fields of all the thread's rows — straight-line loads at clamped addresses, 3 x RPT in flight
    // before the first use (inside the per-row loop below they were RPT dependent global round trips, one per row:
    // the bit loop of a row kept the next row's loads from being issued).
    unsigned long long fld[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) fld[k] = 0ull;
    // several cameras: the shadow is the intersection of their voxel sets (hand_set.cpp:159-172)
    for (int cb = 0; cb < set_nb; cb++) {
      const uint32_t *sc = sb + (size_t)cb * SETWORDS;
      uint32_t wa[RPT], wb[RPT], wc[RPT];
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int row = tid + k * IMG_THREADS;
        const int ix = row / VDIM, iy = row - ix * VDIM;
        const int sx = x0 + ix - ox, sy = y0 + iy - oy;
        const bool ok = row < NROWS && (unsigned)sx < (unsigned)SD && (unsigned)sy < (unsigned)SD && za < zb;
        const int w0 = ok ? ((sx * SD + sy) * SD + za) >> 5 : 0;
        wa[k] = sc[w0];
        wb[k] = sc[w0 + 1 < SETWORDS ? w0 + 1 : SETWORDS - 1];
        wc[k] = sc[w0 + 2 < SETWORDS ? w0 + 2 : SETWORDS - 1];
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int row = tid + k * IMG_THREADS;
        const int ix = row / VDIM, iy = row - ix * VDIM;
        const int sx = x0 + ix - ox, sy = y0 + iy - oy;
        const bool ok = row < NROWS && (unsigned)sx < (unsigned)SD && (unsigned)sy < (unsigned)SD && za < zb;
        const int b0 = ok ? (sx * SD + sy) * SD + za : 0;
        const int w0 = b0 >> 5, sh = b0 & 31;
        const unsigned long long lo64 = (unsigned long long)wa[k] | ((unsigned long long)(w0 + 1 < SETWORDS ? wb[k] : 0u) << 32);
        const unsigned long long hi = (w0 + 2 < SETWORDS) ? wc[k] : 0u;
        const unsigned long long f = (lo64 >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
        fld[k] = !ok ? 0ull : (cb == 0 ? f : (fld[k] & f));
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int row = tid + k * IMG_THREADS;
      if (row >= NROWS) continue;
      const int ix = row / VDIM, iy = row - ix * VDIM;
      unsigned long long field = fld[k];
      field &= (nbits >= 64) ? ~0ull : ((1ull << (nbits > 0 ? nbits : 0)) - 1ull);
      if (field) {
        // The row is a line along world z: its hand-frame coordinates are affine in z, t_a(z) = tb_a + z * bz_a, so per slab
        // lo_a < t_a < hi_a the voxels inside form an interval of z.  Voxels at least EPS steps inside all three intervals
        // are in the box (sure), voxels at least EPS outside one are not; only a voxel within EPS of a slab face goes
        // through the exact f64 test of the oracle below.  EPS = 1e-3 steps is >= 1e-12 m of hand coordinate (|bz| > 1e-9)
        // against ~1e-15 m of rounding in either evaluation, and ~1e-7 steps of error in the quotient.  (Every voxel of
        // the interval used to be tested exactly: ~35 VALU instructions per voxel and wave trip, a sixth of the kernel's
        // instructions — profiles/pmc_mix.sh: the kernel keeps the VALU pipe 56 % busy, it is instruction bound.)
        constexpr double EPS = 1e-3;
        const double fx = (double)ix, fy = (double)iy;
        double dmin = 0.0, dmax = (double)(nbits - 1);  // may be inside
        double smin = 0.0, smax = (double)(nbits - 1);  // surely inside
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const double tba = T0[a] + fx * Ax[a] + fy * Ay[a];
          if (invBz[a] != 0.0) {
            const double dl = (Lf[a] - tba) * invBz[a], dh = dl + Wd[a];
            dmin = fmax(dmin, dl - EPS);
            dmax = fmin(dmax, dh + EPS);
            smin = fmax(smin, dl + EPS);
            smax = fmin(smax, dh - EPS);
          } else if (tba < B.lo[a] - 1e-6 || tba > B.hi[a] + 1e-6) {
            dmax = -1.0;  // the row runs parallel to this slab (it moves < 1e-7 m over its 64 voxels), outside it
          } else if (!(tba > B.lo[a] + 1e-6 && tba < B.hi[a] - 1e-6)) {
            smax = -1.0;  // parallel and within a micrometre of a face: every voxel of the row is tested
          }
        }
        if (dmax < dmin) {
          field = 0ull;
        } else {
          const int t0 = (int)ceil(dmin), t1 = (int)floor(dmax);
          if (t0 > 0) field &= ~((1ull << t0) - 1ull);
          if (t1 < 63) field &= (2ull << t1) - 1ull;
          if (smax >= smin) {
            const int u0 = (int)ceil(smin), u1 = (int)floor(smax);
            if (u1 >= u0) {
              unsigned long long sure = field;
              if (u0 > 0) sure &= ~((1ull << u0) - 1ull);
              if (u1 < 63) sure &= (2ull << u1) - 1ull;
              mask[k] |= sure << (za - zlo);
              field &= ~sure;
            }
          }
        }
      }
      while (field) {
        const int t = __ffsll((long long)field) - 1;
        field &= field - 1;
        const int iz = za + t - zlo;
        double th[3];
        to_hand(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, th);
        if (in_box(B, th)) mask[k] |= 1ull << iz;
      }
    }
  }
  // ---- ordered list of the in-box voxels (ascending voxel index = row, then z): row counts -> exclusive prefix over
  //      the rows in row order (a thread sums RPT consecutive rows) -> every thread writes its rows' voxels
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int row = tid + k * IMG_THREADS;
    if (row < NROWS) S.bp.rows.rowcnt[row] = (uint8_t)__popcll(mask[k]);
  }
  __syncthreads();
  TICK(0);
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int row = tid * RPT + k;
    if (row < NROWS) cnt += S.bp.rows.rowcnt[row];
  }
  int n_sh;
  int pos = block_excl_scan(S, cnt, &n_sh);
  if (n_sh > SHC) {  // this instantiation cannot list the box: queue the candidate for the large one
    if (tid == 0) {
      if (P.overflow_list) {
        const int slot = atomicAdd(P.overflow_count, 1);
        P.overflow_list[slot] = cand;
      } else {
        atomicOr(P.status, 4);
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int row = tid * RPT + k;
    if (row < NROWS) {
      S.bp.rows.rowbase[row] = (uint16_t)pos;
      pos += S.bp.rows.rowcnt[row];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int row = tid + k * IMG_THREADS;
    if (row >= NROWS) continue;
    unsigned long long m = mask[k];
    int at = S.bp.rows.rowbase[row];
    const int rx = row / VDIM, ry = row - rx * VDIM;
    const uint32_t rowpos = ((uint32_t)rx << 12) | ((uint32_t)ry << 6);  // shifts instead of divisions wherever it is read
    while (m) {
      const int iz = __ffsll((long long)m) - 1;
      m &= m - 1;
      S.lin[at++] = rowpos | (uint32_t)iz;
    }
  }
  __syncthreads();  // from here on bp.place may overwrite the row tables
  const int ns = n_sh < SHC ? n_sh : SHC;
  if (P.dbg && tid == 0) {
    atomicMax(&P.dbg[28], (unsigned long long)n_sh);
    atomicAdd(&P.dbg[29], (unsigned long long)n_sh);
  }
  // the three cell coordinates of every entry, once: voxel -> hand frame -> exact threshold lookup
  for (int k = tid; k < ns; k += IMG_THREADS) {
    const int lin = (int)S.lin[k];
    const int ix = lin >> 12, iy = (lin >> 6) & 63, iz = lin & 63;
    double th[3];
    to_hand(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, th);
    const uint32_t c3 = cells_of(S, B, th);
    S.lin[k] = (uint32_t)lin | ((c3 & 0xfffu) << LB);
    S.cz[k] = (uint8_t)(c3 >> 12);
  }
  __syncthreads();
  TICK(1);
  auto cell_of_entry = [&](int k, int pr) {
    const uint32_t v = S.lin[k];
    return cell_of_key(((v >> LB) & 0xfffu) | ((uint32_t)S.cz[k] << 12), pr);
  };
  for (int pr = 0; pr < 3; pr++) {
    for (int c = tid; c < kPix / 4; c += IMG_THREADS) reinterpret_cast<uint4 *>(S.cells)[c] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int k = tid; k < ns; k += IMG_THREADS) atomicAdd(&S.cells[cell_of_entry(k, pr)], 1u);
    __syncthreads();
    scan_cells(S);
    for (int k = tid; k < ns; k += IMG_THREADS) {
      const uint32_t old = atomicAdd(&S.cells[cell_of_entry(k, pr)], 1u);
      S.bp.place[(old >> 16) + (old & 0xffffu)] = (uint16_t)k;
    }
    __syncthreads();
    TICK(2);
    const int da = depth_axis(pr);
    // column `da` of F and the matching offset, selected without indexing the register-resident box
    const double Fd0 = da == 0 ? B.F[0] : (da == 1 ? B.F[1] : B.F[2]);
    const double Fd1 = da == 0 ? B.F[3] : (da == 1 ? B.F[4] : B.F[5]);
    const double Fd2 = da == 0 ? B.F[6] : (da == 1 ? B.F[7] : B.F[8]);
    const double offd = da == 0 ? B.off[0] : (da == 1 ? B.off[1] : B.off[2]);
    float lmax = -FLT_MAX;
    int lany = 0;
    const int n_nz = list_nonempty_cells(S, S.nz);
    for (int c = tid; c < kPix / 4; c += IMG_THREADS) reinterpret_cast<float4 *>(S.raster0)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    TICK(9);
    for (int qn = tid; qn < n_nz; qn += IMG_THREADS) {
      const int c = S.nz[qn];
      const uint32_t w = S.cells[c];
      const int cn = (int)(w & 0xffffu), start = (int)(w >> 16);
      float v = 0.f, fc = 0.f;
      auto visit = [&](int k) {  // one set voxel, in ascending voxel order (the std::set order of the oracle)
        const int lin = (int)(S.lin[k] & ((1u << LB) - 1u));
        const int ix = lin >> 12, iy = (lin >> 6) & 63, iz = lin & 63;
        const double c0 = (double)(ix + x0) * K.voxel - B.sample[0], c1 = (double)(iy + y0) * K.voxel - B.sample[1],
                     c2 = (double)(iz + z0) * K.voxel - B.sample[2];
        const double td = Fd0 * c0 + Fd1 * c1 + Fd2 * c2;
        const double d = div_len(td - offd, da);
        fc = (float)((double)fc + 1.0);
        v = (float)((double)v + (d - (double)v) * recip_count<128>(S.recip, fc));
      };
      sort_u16(&S.bp.place[start], cn);
      for (int e = 0; e < cn; e++) visit((int)S.bp.place[start + e]);
      lmax = fmaxf(lmax, v);
      lany = 1;
      S.raster0[c] = v;
    }
    __syncthreads();
    TICK(10);
    lmax = wave_max_f32(lmax);
    lany = __ballot(lany != 0) != 0ull;
    if (lane == 0) {  // (red_f / red_i were last read before the barrier above)
      S.red_f[tid >> 6] = lmax;
      S.red_i[tid >> 6] = lany;
    }
    __syncthreads();
    float gmax = -FLT_MAX;
    int gany = 0;
#pragma unroll
    for (int w = 0; w < IMG_WAVES; w++) {
      gmax = fmaxf(gmax, S.red_f[w]);
      gany |= S.red_i[w];
    }
    // minMaxLoc with mask -> max (0 if the mask is empty); image = max_img - image
    const double mxd = gany ? (double)gmax : 0.0;
    const float mxf = (float)mxd;
    // (no barrier: red_f is rewritten behind the first barrier of finalize_planes only)
    for (int c = tid; c < kPix / 4; c += IMG_THREADS) {
      const uint4 w = reinterpret_cast<const uint4 *>(S.cells)[c];
      float4 r = reinterpret_cast<float4 *>(S.raster0)[c];
      r.x = ((w.x & 0xffffu) ? mxf : 0.0f) - r.x;
      r.y = ((w.y & 0xffffu) ? mxf : 0.0f) - r.y;
      r.z = ((w.z & 0xffffu) ? mxf : 0.0f) - r.z;
      r.w = ((w.w & 0xffffu) ? mxf : 0.0f) - r.w;
      reinterpret_cast<float4 *>(S.raster0)[c] = r;
    }
    __syncthreads();
    TICK(3);
    finalize_planes<1>(S, &S.raster0[0], nullptr, out + (size_t)(pr * K.per + 4) * kPix);
    TICK(4);
  }
  if (tid == 0 && S.flag) atomicOr(P.status, S.flag);
}

// LGRID workgroups of the large instantiations walk the queue of the small one; the queue length is read
// on the device, so the host enqueues them without waiting for it (an empty queue costs an empty launch)
constexpr int LGRID = 256;

template <int SHC, bool WIDE>
__global__ __launch_bounds__(IMG_THREADS, SHC <= SH_CAP ? 4 : 2) void shadow_image_kernel(ImgParams P) {
  __shared__ SmemShadow<SHC, WIDE> S;
  if constexpr (SHC > SH_CAP) {
    const int count = *P.cand_count;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
      __syncthreads();  // the previous candidate's LDS is dead
      shadow_image_body<SHC, WIDE>(P, S, P.cand_list[q]);
    }
  } else {
    const int cand = xcd_candidate(P.num_cand);
    if (cand < 0) return;
    shadow_image_body<SHC, WIDE>(P, S, cand);
  }
}

// ---------------------------------------------------------------------------
// shadow_image_any_kernel: the shadow channels of a candidate whose box the tuned kernels above cannot take — an image
// volume beyond their voxel windows (the host then sends EVERY candidate here), or more in-box voxels than the large
// instantiation lists (queued by it).  The reference has no size limit (hand_set.cpp:138, image_strategy.cpp:192-233);
// this is its device counterpart without one that matters: window edges up to 256 voxels (0.77 m), 65535 voxels in
// the box, a set region of run-time size.  Same arithmetic and the same order as shadow_image_body — the voxel list in
// lexicographic order, per projection a counting sort by pixel and one lane per pixel running the mean in list order —
// with the list, its cell keys and the segment table in a scratch row in global memory and none of the register tricks.
// A persistent launch: workgroup b takes entries b, b + grid, ... of its list.
// ---------------------------------------------------------------------------
constexpr int HUGE_CAP = 65535;       // in-box voxels (segment starts are 16 bits wide in the pixel table)
constexpr int HUGE_WIN = 256;         // window edge in voxels
constexpr size_t HUGE_SCRATCH_BYTES = (size_t)HUGE_WIN * HUGE_WIN * sizeof(int32_t) + (size_t)(HUGE_CAP + 1) * (2 * sizeof(uint32_t) + sizeof(uint16_t)) + 64;
struct __attribute__((aligned(16))) SmemAny {
  __attribute__((aligned(16))) float raster0[kPix];
  __attribute__((aligned(16))) uint32_t cells[kPix];
  uint16_t nz[kPix];
  double thr[3][kImg + 1];
  double recip[128];
  float red_f[4 * IMG_WAVES];
  int red_i[IMG_WAVES];
  int vorg[3], wdim[3];
  int flag;
};

__device__ void shadow_image_any_body(const ImgParams &P, SmemAny &S, const int cand, char *scratch) {
  const ImgConsts &K = c_img;
  const int SD = K.set_sd, SR = K.set_sr;
  const size_t SETWORDS = (size_t)(((long long)SD * SD * SD + 31) / 32);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int set_ord = P.meta[4 * cand + 2];
  const int set_nb = P.meta[4 * cand + 3];
  uint8_t *out = P.images + (size_t)cand * kPix * K.C;
  int32_t *rowtab = reinterpret_cast<int32_t *>(scratch);                              // [wx * wy]: count, then start of a window row
  uint32_t *lin = reinterpret_cast<uint32_t *>(rowtab + HUGE_WIN * HUGE_WIN);          // [n]: ix | iy << 8 | iz << 16, ascending = lexicographic
  uint32_t *ckey = lin + (HUGE_CAP + 1);                                               // [n]: cell x | y << 6 | z << 12
  uint16_t *place = reinterpret_cast<uint16_t *>(ckey + (HUGE_CAP + 1));               // [n]: segment table of the counting sort
  Box B;
  load_box(P.hands[P.cand_hand[cand]], B);
  for (int i = tid; i < 3 * (kImg + 1); i += IMG_THREADS) (&S.thr[0][0])[i] = (&K.thr[0][0])[i];
  for (int i = tid; i < 128; i += IMG_THREADS) S.recip[i] = i ? 1.0 / (double)i : 0.0;
  if (tid == 0) {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int k = 0; k < 8; k++) {
      const double bx = (k & 1) ? B.hi[0] : B.lo[0];
      const double by = (k & 2) ? B.hi[1] : B.lo[1];
      const double bz = (k & 4) ? B.hi[2] : B.lo[2];
      for (int a = 0; a < 3; a++) {
        const double w = B.sample[a] + B.F[3 * a] * bx + B.F[3 * a + 1] * by + B.F[3 * a + 2] * bz;
        lo[a] = fmin(lo[a], w);
        hi[a] = fmax(hi[a], w);
      }
    }
    int bad = 0;
    for (int a = 0; a < 3; a++) {
      const int v0 = (int)floor(lo[a] * K.voxel_mult) - 1, v1 = (int)floor(hi[a] * K.voxel_mult) + 1;
      const int o = (int)floor(B.sample[a] * K.voxel_mult) - SR;
      S.vorg[a] = v0;
      S.wdim[a] = v1 - v0 + 1;
      // (the window's one-voxel margins may stick out of the region: they hold no voxel of the box)
      if (v1 - v0 + 1 > HUGE_WIN || v0 + 1 < o || v1 - 1 >= o + SD) bad = 1;
    }
    S.flag = bad;
  }
  __syncthreads();
  if (S.flag) {  // beyond even this kernel (GPD_ERR_CAPACITY, flag 1): reported, never truncated
    if (tid == 0) atomicOr(P.status, 1);
    return;
  }
  const int x0 = S.vorg[0], y0 = S.vorg[1], z0 = S.vorg[2];
  const int wx = S.wdim[0], wy = S.wdim[1], wz = S.wdim[2];
  const int nrows = wx * wy;
  const int ox = (int)floor(B.sample[0] * K.voxel_mult) - SR, oy = (int)floor(B.sample[1] * K.voxel_mult) - SR,
            oz = (int)floor(B.sample[2] * K.voxel_mult) - SR;
  // the set voxels of window row (ix, iy) that lie in the box, in ascending z: emit(iz) for each.  A row is a run of wz bits
  // of the set region(s) — several cameras: the intersection of their voxel sets (hand_set.cpp:159-172) — and every set bit
  // takes the oracle's exact f64 box test
  auto walk_row = [&](int row, auto emit) {
    const int ix = row / wy, iy = row - ix * wy;
    const int sx = x0 + ix - ox, sy = y0 + iy - oy;
    if (set_ord < 0 || (unsigned)sx >= (unsigned)SD || (unsigned)sy >= (unsigned)SD) return;
    const int zlo = z0 - oz;  // region z of window z 0
    const int za = zlo < 0 ? 0 : zlo, zb = zlo + wz < SD ? zlo + wz : SD;
    if (za >= zb) return;
    const long long rowbit = ((long long)sx * SD + sy) * SD;
    for (long long w = (rowbit + za) >> 5; w <= (rowbit + zb - 1) >> 5; w++) {
      uint32_t bits = ~0u;
      for (int cb = 0; cb < set_nb; cb++) bits &= P.set_bits[((size_t)set_ord + cb) * SETWORDS + (size_t)w];
      const long long first = w << 5;  // region bit of this word's bit 0
      if (first < rowbit + za) bits &= ~0u << (int)(rowbit + za - first);
      if (first + 32 > rowbit + zb) bits &= ~0u >> (int)(first + 32 - (rowbit + zb));
      while (bits) {
        const int t = __ffs((int)bits) - 1;
        bits &= bits - 1;
        const int iz = (int)(first + t - rowbit) - zlo;
        double th[3];
        to_hand(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, th);
        if (in_box(B, th)) emit(iz);
      }
    }
  };
  for (int row = tid; row < nrows; row += IMG_THREADS) {
    int cnt = 0;
    walk_row(row, [&](int) { cnt++; });
    rowtab[row] = cnt;
  }
  __threadfence_block();
  __syncthreads();
  // exclusive prefix over the rows in row order: a thread sums a run of consecutive rows
  const int per = (nrows + IMG_THREADS - 1) / IMG_THREADS;
  int cnt = 0;
  for (int k = 0; k < per; k++) {
    const int row = tid * per + k;
    if (row < nrows) cnt += rowtab[row];
  }
  int n_sh;
  int pos = block_excl_scan(S, cnt, &n_sh);
  if (n_sh > HUGE_CAP) {
    if (tid == 0) atomicOr(P.status, 4);
    return;
  }
  for (int k = 0; k < per; k++) {
    const int row = tid * per + k;
    if (row < nrows) {
      const int c = rowtab[row];
      rowtab[row] = pos;
      pos += c;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int row = tid; row < nrows; row += IMG_THREADS) {
    int at = rowtab[row];
    const int ix = row / wy, iy = row - ix * wy;
    walk_row(row, [&](int iz) {
      double th[3];
      to_hand(B, (double)(ix + x0) * K.voxel, (double)(iy + y0) * K.voxel, (double)(iz + z0) * K.voxel, th);
      lin[at] = (uint32_t)ix | ((uint32_t)iy << 8) | ((uint32_t)iz << 16);
      ckey[at] = cells_of(S, B, th);
      at++;
    });
  }
  __threadfence_block();
  __syncthreads();
  const int ns = n_sh;
  for (int pr = 0; pr < 3; pr++) {
    for (int c = tid; c < kPix; c += IMG_THREADS) S.cells[c] = 0u;
    __syncthreads();
    for (int k = tid; k < ns; k += IMG_THREADS) atomicAdd(&S.cells[cell_of_key(ckey[k], pr)], 1u);
    __syncthreads();
    scan_cells(S);
    for (int k = tid; k < ns; k += IMG_THREADS) {
      const uint32_t old = atomicAdd(&S.cells[cell_of_key(ckey[k], pr)], 1u);
      place[(old >> 16) + (old & 0xffffu)] = (uint16_t)k;
    }
    __threadfence_block();
    __syncthreads();
    const int da = depth_axis(pr);
    const double Fd0 = da == 0 ? B.F[0] : (da == 1 ? B.F[1] : B.F[2]);
    const double Fd1 = da == 0 ? B.F[3] : (da == 1 ? B.F[4] : B.F[5]);
    const double Fd2 = da == 0 ? B.F[6] : (da == 1 ? B.F[7] : B.F[8]);
    const double offd = da == 0 ? B.off[0] : (da == 1 ? B.off[1] : B.off[2]);
    float lmax = -FLT_MAX;
    int lany = 0;
    const int n_nz = list_nonempty_cells(S, S.nz);
    for (int c = tid; c < kPix; c += IMG_THREADS) S.raster0[c] = 0.f;
    __syncthreads();
    for (int qn = tid; qn < n_nz; qn += IMG_THREADS) {
      const int c = S.nz[qn];
      const uint32_t w = S.cells[c];
      const int cn = (int)(w & 0xffffu), start = (int)(w >> 16);
      float v = 0.f, fc = 0.f;
      // the segment holds list positions in arrival order: ascending position = ascending voxel (the std::set order of the oracle)
      sort_u16(&place[start], cn);
      for (int e = 0; e < cn; e++) {
        const uint32_t lp = lin[place[start + e]];
        const int ix = lp & 255, iy = (lp >> 8) & 255, iz = lp >> 16;
        const double c0 = (double)(ix + x0) * K.voxel - B.sample[0], c1 = (double)(iy + y0) * K.voxel - B.sample[1],
                     c2 = (double)(iz + z0) * K.voxel - B.sample[2];
        const double td = Fd0 * c0 + Fd1 * c1 + Fd2 * c2;
        const double d = div_len(td - offd, da);
        fc = (float)((double)fc + 1.0);
        v = (float)((double)v + (d - (double)v) * recip_count<128>(S.recip, fc));
      }
      lmax = fmaxf(lmax, v);
      lany = 1;
      S.raster0[c] = v;
    }
    __syncthreads();
    lmax = wave_max_f32(lmax);
    lany = __ballot(lany != 0) != 0ull;
    if (lane == 0) {
      S.red_f[tid >> 6] = lmax;
      S.red_i[tid >> 6] = lany;
    }
    __syncthreads();
    float gmax = -FLT_MAX;
    int gany = 0;
    for (int w = 0; w < IMG_WAVES; w++) {
      gmax = fmaxf(gmax, S.red_f[w]);
      gany |= S.red_i[w];
    }
    // minMaxLoc with mask -> max (0 if the mask is empty); image = max_img - image
    const double mxd = gany ? (double)gmax : 0.0;
    const float mxf = (float)mxd;
    __syncthreads();  // red_f is rewritten by finalize_planes
    for (int c = tid; c < kPix; c += IMG_THREADS) S.raster0[c] = ((S.cells[c] & 0xffffu) ? mxf : 0.0f) - S.raster0[c];
    __syncthreads();
    finalize_planes<1>(S, &S.raster0[0], nullptr, out + (size_t)(pr * K.per + 4) * kPix);
  }
}

__global__ __launch_bounds__(IMG_THREADS) void shadow_image_any_kernel(ImgParams P) {
  __shared__ SmemAny S;
  char *scratch = P.huge_scratch + (size_t)blockIdx.x * HUGE_SCRATCH_BYTES;
  const int count = P.cand_list ? *P.cand_count : P.num_cand;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    __syncthreads();  // the previous candidate's LDS is dead
    shadow_image_any_body(P, S, P.cand_list ? P.cand_list[q] : q, scratch);
  }
}

// ---------------------------------------------------------------------------
// grasp_image_kernel: normals (3) and depth (1) channels per projection of one candidate
// (createNormalsImage / createDepthImage, image_strategy.cpp:124-190).
// ---------------------------------------------------------------------------
template <bool BIG>
__device__ __forceinline__ void grasp_image_body(const ImgParams &P, SmemPts<BIG> &S, const int cand) {
  constexpr int CAP = BIG ? PT_CAP_BIG : PT_CAP;
  unsigned long long t_last = __builtin_readcyclecounter();
  const ImgConsts &K = c_img;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int slot_s = P.meta[4 * cand + 0];
  const int N = P.meta[4 * cand + 1];
  const float *nn = P.nn + (size_t)slot_s * 6 * P.cap;
  uint8_t *out = P.images + (size_t)cand * kPix * K.C;
  // the point arrays: LDS, or this workgroup's row of the global scratch
  double *gt = nullptr;
  float4 *gan = nullptr;
  uint32_t *gidx = nullptr;
  if constexpr (BIG) {
    char *row = P.pts_scratch + (size_t)blockIdx.x * PTS_SCRATCH_BYTES;
    gt = reinterpret_cast<double *>(row);
    gan = reinterpret_cast<float4 *>(row + (size_t)CAP * 3 * sizeof(double));
    gidx = reinterpret_cast<uint32_t *>(row + (size_t)CAP * (3 * sizeof(double) + sizeof(float4)));
  }
  auto IDX = [&](int e) -> int {
    if constexpr (BIG) return (int)gidx[e];
    else return (int)S.place[e];
  };
  auto T = [&](int a, int e) -> double & {
    if constexpr (BIG) return gt[(size_t)a * CAP + e];
    else return S.p.t[a][e];
  };
  auto AN = [&](int e) -> float4 & {
    if constexpr (BIG) return gan[e];
    else return S.p.an[e];
  };
  Box B;
  load_box(P.hands[P.cand_hand[cand]], B);
  for (int i = tid; i < 3 * (kImg + 1); i += IMG_THREADS) (&S.thr[0][0])[i] = (&K.thr[0][0])[i];
  for (int i = tid; i < 128; i += IMG_THREADS) S.recip[i] = i ? 1.0 / (double)i : 0.0;
  if (tid == 0) {
    S.flag = 0;
    S.counter = 0;
    S.nzv[0] = make_float4(0.f, 0.f, 0.f, 0.f);  // the empty pixel of the index raster (the walks write slots >= 1)
  }
  __syncthreads();
  // The in-box points are numbered IN NEIGHBOUR ORDER (an ordered compaction: per 512 neighbours one ballot per wave
  // and a prefix over the eight wave counts), so that an entry's number is its rank among the in-box points: the
  // walks order a pixel's segment by entry number alone — no rank bits to carry, whatever the neighbourhood size.
  // Two passes over blocks of 16 rounds (8192 neighbours): first every thread tests its 16 points — straight-line
  // loads, all in flight together, no barrier — and the waves leave their 16 x 8 ballot counts in LDS; ONE barrier; then
  // every thread derives the entry numbers of its in-box points from those counts and writes them.  (One round per
  // barrier pair exposed the latency of its global loads six times per candidate: 28 of the kernel's 130 kcycles.)
  int n_before = 0;  // in-box points of the earlier blocks (the same in every thread)
  int *cnt = reinterpret_cast<int *>(S.cells);  // [16 + 3][IMG_WAVES] ballot counts; the cell counters are not in use yet
  constexpr int RB = 16;
  const int wave = tid >> 6;
  for (int b0 = 0; b0 < N; b0 += RB * IMG_THREADS) {
    unsigned inmask = 0;
    const int left = N - b0, nr = left >= RB * IMG_THREADS ? RB : (left + IMG_THREADS - 1) / IMG_THREADS;  // rounds of this block
    for (int r0 = 0; r0 < nr; r0 += 4) {  // four rounds at a time: twelve loads in flight before the first use
      float px[4], py[4], pz[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = b0 + (r0 + q) * IMG_THREADS + tid;
        const int ic = i < N ? i : N - 1;
        px[q] = nn[0 * P.cap + ic];
        py[q] = nn[1 * P.cap + ic];
        pz[q] = nn[2 * P.cap + ic];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + q;
        const int i = b0 + r * IMG_THREADS + tid;
        double t[3];
        to_hand(B, (double)px[q], (double)py[q], (double)pz[q], t);
        const bool in = i < N && in_box(B, t);
        const unsigned long long ballot = __ballot(in);
        if (lane == 0) cnt[r * IMG_WAVES + wave] = __popcll(ballot);  // (rounds past nr: zeros, never read)
        inmask |= (unsigned)in << r;
      }
    }
    __syncthreads();
    int run = n_before;
    for (int r = 0; r < nr; r++) {  // entry numbers of the thread's in-box points: their neighbour indices go to the list
      int base = run;
#pragma unroll
      for (int w = 0; w < IMG_WAVES; w++) {
        const int c = cnt[r * IMG_WAVES + w];
        if (w < wave) base += c;
        run += c;
      }
      const bool in = (inmask >> r) & 1u;
      const unsigned long long ballot = __ballot(in);
      if (in) {
        const int e = base + __popcll(ballot & ((1ull << lane) - 1ull));
        if (e < CAP) {
          if constexpr (BIG) gidx[e] = (uint32_t)(b0 + r * IMG_THREADS + tid);
          else S.place[e] = (uint16_t)(b0 + r * IMG_THREADS + tid);
        }
      }
    }
    n_before = run;
    __syncthreads();  // the counts are rewritten by the next block / the cell counters start here
  }
  const int n_box_all = n_before;
  if (n_box_all > CAP || (!BIG && N > 65536)) {
    // more in-box points than this instantiation holds — or a neighbourhood of more than 65536 points, whose neighbour
    // indices do not fit the 16-bit segment table the small instantiation lists them in (the reference has no limit,
    // hand_search.cpp:178; the large one keeps them 32 bits wide): queue the candidate for the large one
    // (nothing has been written yet), or report it when this already is the large one
    if (tid == 0) {
      if (!BIG && P.pts_overflow_list)
        P.pts_overflow_list[atomicAdd(P.pts_overflow_count, 1)] = cand;
      else
        atomicOr(P.status, 2);
    }
    return;
  }
  // The entries themselves, densely: one in-box point per lane (hand-frame coordinates, |normal| in the hand frame, the
  // three cell coordinates: ~150 instructions) — inside the rounds above a wave paid them per round for the one lane
  // in seven that was in the box (a sixth of the kernel's instructions; it is instruction bound, profiles/pmc_mix.sh).
  for (int e0 = 0; e0 < n_box_all; e0 += 2 * IMG_THREADS) {
    float v[2][6];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int e = e0 + q * IMG_THREADS + tid;
      const int i = e < n_box_all ? IDX(e) : 0;
#pragma unroll
      for (int a = 0; a < 6; a++) v[q][a] = nn[a * P.cap + i];
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int e = e0 + q * IMG_THREADS + tid;
      if (e < n_box_all) {
        double t[3];
        to_hand(B, (double)v[q][0], (double)v[q][1], (double)v[q][2], t);
        const double n0 = (double)v[q][3], n1 = (double)v[q][4], n2 = (double)v[q][5];
        T(0, e) = t[0];
        T(1, e) = t[1];
        T(2, e) = t[2];
        AN(e) = make_float4((float)fabs(B.F[0] * n0 + B.F[3] * n1 + B.F[6] * n2),
                            (float)fabs(B.F[1] * n0 + B.F[4] * n1 + B.F[7] * n2),
                            (float)fabs(B.F[2] * n0 + B.F[5] * n1 + B.F[8] * n2), __uint_as_float(cells_of(S, B, t)));
      }
    }
  }
  // (the index list lives in the segment table; the first projection rewrites it two barriers from here)
  if constexpr (BIG) {
    __threadfence_block();  // the point arrays are global memory here: written above, read by other lanes below
    __syncthreads();
  }
  const int nb = n_box_all;
  if (P.dbg && tid == 0) {
    atomicMax(&P.dbg[30], (unsigned long long)n_box_all);
    atomicAdd(&P.dbg[31], (unsigned long long)n_box_all);
  }
  TICK(5);
  for (int c = tid; c < kPix / 4; c += IMG_THREADS) reinterpret_cast<uint4 *>(S.cells)[c] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  for (int pr = 0; pr < K.nproj; pr++) {
    // (the cell counters are zero: cleared above, then by the copy-out of the previous projection)
    for (int e = tid; e < nb; e += IMG_THREADS) atomicAdd(&S.cells[cell_of_key(__float_as_uint(AN(e).w), pr)], 1u);
    __syncthreads();
    scan_cells(S);
    for (int e = tid; e < nb; e += IMG_THREADS) {
      const uint32_t key = __float_as_uint(AN(e).w);
      const uint32_t old = atomicAdd(&S.cells[cell_of_key(key, pr)], 1u);
      S.place[(old >> 16) + (old & 0xffffu)] = (uint16_t)e;
    }
    __syncthreads();
    TICK(6);
    // the pixel owner walks its segment in neighbour order
    const int da = depth_axis(pr);
    const double offd = da == 0 ? B.off[0] : (da == 1 ? B.off[1] : B.off[2]);
    uint16_t *nz = S.nzlist;
    const int n_nz = list_nonempty_cells(S, nz);
    TICK(11);
    for (int qn = tid; qn < n_nz; qn += IMG_THREADS) {
      const int c = nz[qn];
      const uint32_t w = S.cells[c];
      const int cn = (int)(w & 0xffffu), start = (int)(w >> 16);
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      float avg = 0.f, fc = 0.f;
      auto visit = [&](int e) {  // one in-box point, in neighbour order
        const float4 an = AN(e);
        const float a0 = an.x, a1 = an.y, a2 = an.z;
        if (v0 == 0.f && v1 == 0.f && v2 == 0.f) {
          v0 = a0;
          v1 = a1;
          v2 = a2;
        } else {
          const float sq = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
          const double inv = 1.0 / (double)sq;
          const float d0 = a0 - v0, d1 = a1 - v1, d2 = a2 - v2;
          v0 = v0 + (float)((double)d0 * inv);
          v1 = v1 + (float)((double)d1 * inv);
          v2 = v2 + (float)((double)d2 * inv);
        }
        const double d = div_len(T(da, e) - offd, da);
        fc = (float)((double)fc + 1.0);
        avg = (float)((double)avg + (d - (double)avg) * recip_count<128>(S.recip, fc));
      };
      sort_u16_lean(&S.place[start], cn);  // neighbour order = entry order
      for (int q = 0; q < cn; q++) visit((int)S.place[start + q]);
      S.nzv[qn + 1] = make_float4(v0, v1, v2, (float)(1.0 - (double)avg));
      S.cells[c] = 0x80000000u | (uint32_t)(qn + 1);  // the pixel's own word, read above: from here on the index raster
    }
    TICK(13);
    __syncthreads();
    TICK(7);
    // ---- the four planes of the projection at once.  `cells` is now an index raster: bit 31 set <-> the pixel holds
    //      points, low bits = its slot in nzv (float4: the three normal values and the depth value); every other word has
    //      bit 31 clear (segment starts are < 2^15) and stands for the empty pixel, slot 0 = zeros.  A group of four pixels
    //      is dilated for all four planes from one set of index reads + float4 gathers (a dense float raster per plane
    //      cost four scatter / dilate / barrier rounds); empty pixels take part with the value 0 exactly as in the
    //      reference's zero-initialised cv::Mat (the running normal "average" can go negative, so 0 matters).
    //      Five groups in six have no point in their 3 x 6 window: they are background in all four planes — the value 0
    //      enters the min / max and the normalised byte of 0 is stored, nothing to gather, dilate or round.  The others
    //      are LISTED first (one ballot + one LDS atomic per wave) and dealt out over the whole workgroup, one or two per
    //      thread.  (Measured: image stage 1.254 -> 1.242 ms; skipping the background groups in place, without the
    //      list, gave the same.)
    const bool with_normals = K.C != 1, with_depth = K.C == 1 || K.C >= 12;
    uint16_t *alist = S.place;  // the segment table is dead after the walks
    if (tid == 0) S.counter = 0;
    __syncthreads();
    bool live[GPT];  // this thread's own groups tid + k * IMG_THREADS: does the window hold points?
    auto window = [&](int g, uint32_t(&ix)[3][6]) {
      const int r = g / 15, c0 = (g - r * 15) * 4;
      const int rows[3] = {r > 0 ? r - 1 : 0, r, r < kImg - 1 ? r + 1 : kImg - 1};
      const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + 4 < kImg ? c0 + 4 : kImg - 1;
      uint32_t any = 0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const uint4 m = *reinterpret_cast<const uint4 *>(&S.cells[rows[q] * kImg + c0]);
        ix[q][0] = S.cells[rows[q] * kImg + cl];
        ix[q][1] = m.x;
        ix[q][2] = m.y;
        ix[q][3] = m.z;
        ix[q][4] = m.w;
        ix[q][5] = S.cells[rows[q] * kImg + cr];
#pragma unroll
        for (int e = 0; e < 6; e++) any |= ix[q][e];
      }
      return (any >> 31) != 0u;
    };
    TICK(12);
#pragma unroll
    for (int k = 0; k < GPT; k++) {
      const int g = tid + k * IMG_THREADS;
      uint32_t ix[3][6];
      live[k] = g < 900 && window(g, ix);
      const unsigned long long ballot = __ballot(live[k]);
      if (ballot) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&S.counter, __popcll(ballot));
        base = __builtin_amdgcn_readfirstlane(base);
        if (live[k]) alist[base + __popcll(ballot & ((1ull << lane) - 1ull))] = (uint16_t)g;
      }
    }
    __syncthreads();
    const int n_live = S.counter;
    float d[GPT][4][4];  // the thread's live groups alist[tid + k * IMG_THREADS]: [plane][pixel]
    float mn0 = FLT_MAX, mx0 = -FLT_MAX, mn1 = FLT_MAX, mx1 = -FLT_MAX;
    if (n_live < 900) {  // some group is background
      mn0 = mn1 = 0.f;
      mx0 = mx1 = 0.f;
    }
#pragma unroll
    for (int k = 0; k < GPT; k++) {
      const int a = tid + k * IMG_THREADS;
      if (a < n_live) {
        uint32_t ix[3][6];
        window((int)alist[a], ix);
        float col[4][6];  // per plane: column maxima over the three rows
#pragma unroll
        for (int e = 0; e < 6; e++) {
          const float4 va = S.nzv[(ix[0][e] >> 31) ? (ix[0][e] & 0xffffu) : 0u];
          const float4 vb = S.nzv[(ix[1][e] >> 31) ? (ix[1][e] & 0xffffu) : 0u];
          const float4 vc = S.nzv[(ix[2][e] >> 31) ? (ix[2][e] & 0xffffu) : 0u];
          col[0][e] = fmaxf(fmaxf(va.x, vb.x), vc.x);
          col[1][e] = fmaxf(fmaxf(va.y, vb.y), vc.y);
          col[2][e] = fmaxf(fmaxf(va.z, vb.z), vc.z);
          col[3][e] = fmaxf(fmaxf(va.w, vb.w), vc.w);
        }
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
#pragma unroll
          for (int j = 0; j < 4; j++) d[k][pl][j] = fmaxf(fmaxf(col[pl][j], col[pl][j + 1]), col[pl][j + 2]);
          const float lo = fminf(fminf(d[k][pl][0], d[k][pl][1]), fminf(d[k][pl][2], d[k][pl][3]));
          const float hi = fmaxf(fmaxf(d[k][pl][0], d[k][pl][1]), fmaxf(d[k][pl][2], d[k][pl][3]));
          if (pl < 3) {
            mn0 = fminf(mn0, lo);
            mx0 = fmaxf(mx0, hi);
          } else {
            mn1 = fminf(mn1, lo);
            mx1 = fmaxf(mx1, hi);
          }
        }
      }
    }
    // createNormalsImage: the three planes are normalised as ONE 3-channel image; createDepthImage on its own
    // (image_strategy.cpp:144-153, 178-187; Image1ChannelsStrategy is the depth plane alone)
    if (64 * (tid >> 6) < n_live) {  // (a wave without a live group holds the initial values in every lane)
      mn0 = wave_min_f32(mn0);
      mx0 = wave_max_f32(mx0);
      mn1 = wave_min_f32(mn1);
      mx1 = wave_max_f32(mx1);
    }
    if (lane == 0) {
      S.red_f[4 * (tid >> 6) + 0] = mn0;
      S.red_f[4 * (tid >> 6) + 1] = mx0;
      S.red_f[4 * (tid >> 6) + 2] = mn1;
      S.red_f[4 * (tid >> 6) + 3] = mx1;
    }
    __syncthreads();
    float fs[2], fb[2];
    uint32_t bg[2];
    auto to_byte = [](float x, float s, float b) {  // cv::normalize + convertTo(CV_8U, 255.0)
      const float v = x * s + b;
      const float u = v * 255.0f + 0.0f;
      float t = rintf(u);
      t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
      return (uint32_t)(int)t;
    };
#pragma unroll
    for (int q = 0; q < 2; q++) {
      float a = S.red_f[2 * q], b = S.red_f[2 * q + 1];
#pragma unroll
      for (int w = 1; w < IMG_WAVES; w++) {
        a = fminf(a, S.red_f[4 * w + 2 * q]);
        b = fmaxf(b, S.red_f[4 * w + 2 * q + 1]);
      }
      const double smin = (double)a, smax = (double)b;
      const double scale = 1.0 * ((smax - smin) > DBL_EPSILON ? 1.0 / (smax - smin) : 0.0);
      const double shift = 0.0 - smin * scale;
      fs[q] = (float)scale;
      fb[q] = (float)shift;
      bg[q] = to_byte(0.f, fs[q], fb[q]) * 0x01010101u;  // four background pixels: the value 0 through the same arithmetic
    }
    // The bytes are staged in LDS — the index raster is dead after the barrier above: 4 planes x 900 dwords in MEMORY order
    // (image row = 59 - cell row, image_strategy.cpp:128-129) — and leave as 16-byte stores, 15 per wave instead of 56
    // four-byte ones (the store phase was issue bound: 36 of the 158 us a projection costs, profiles/img_phases.sh).
    uint32_t *stage = S.cells;
    uint4 *stage4 = reinterpret_cast<uint4 *>(S.cells);
    for (int c = tid; c < 900; c += IMG_THREADS) {  // every pixel background ...
      const uint32_t w = bg[c < 3 * 225 ? 0 : 1];
      stage4[c] = make_uint4(w, w, w, w);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GPT; k++) {  // ... the live groups over it
      const int a = tid + k * IMG_THREADS;
      if (a < n_live) {
        const int g = (int)alist[a];
        const int r = g / 15, cg = g - r * 15;
        const int m = (kImg - 1 - r) * 15 + cg;
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
          const int q = pl < 3 ? 0 : 1;
          uint32_t packed = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) packed |= to_byte(d[k][pl][j], fs[q], fb[q]) << (8 * j);
          stage[pl * 900 + m] = packed;
        }
      }
    }
    __syncthreads();
    for (int a = tid; a < 900; a += IMG_THREADS) {
      const int pl = a / 225, j = a - pl * 225;
      const uint4 v = stage4[a];
      stage4[a] = make_uint4(0u, 0u, 0u, 0u);  // the cell counters of the next projection start from zero
      if (pl < 3 ? !with_normals : !with_depth) continue;
      const int ch = K.C == 1 ? 0 : pr * K.per + pl;
      *reinterpret_cast<uint4 *>(out + (size_t)ch * kPix + 16 * j) = v;
    }
    __syncthreads();
    TICK(8);
  }
  if (tid == 0 && S.flag) atomicOr(P.status, S.flag);
}

template <bool BIG>
__global__ __launch_bounds__(IMG_THREADS, BIG ? 2 : 4) void grasp_image_kernel(ImgParams P) {
  __shared__ SmemPts<BIG> S;
  if constexpr (BIG) {
    const int count = *P.cand_count;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
      __syncthreads();  // the previous candidate's LDS is dead
      grasp_image_body<true>(P, S, P.cand_list[q]);
    }
  } else {
    const int cand = xcd_candidate(P.num_cand);
    if (cand < 0) return;
    grasp_image_body<false>(P, S, cand);
  }
}

// ---------------------------------------------------------------------------
// shadow_set_kernel: HandSet::calculateShadow / calculateShadowForCamera (hand_set.cpp:118-233)
// for one camera, once per hand set.  The 33 * N_i LCG draws of the set (stream offset given by
// the host, hand_set.cpp:263-266) are voxelised into a bitset over the cube of +-43 voxels around
// the sample, which contains every image box of the set; the candidates then cut their boxes
// out of it.
// ---------------------------------------------------------------------------
struct SetParams {
  const float *nn;
  int cap;
  const double *centers;
  const double *frames;     // [S][12], sample first
  const int32_t *set_meta;  // [bitsets][8]: sample slot, N_images, lcg offset lo, hi, camera, -
  uint32_t *set_bits;       // [sets][Vox::SETWORDS]
  const gpd_hand *hands;    // [S][slots] records of the search ...
  const int32_t *hand_cand; // ... and which of them are candidates (>= 0): the default mode writes only the part of a set's
  int slots;                //     region its candidates' boxes can touch
  unsigned long long lcg_base;  // draws consumed before this call's first hand set (gpd_hip_detect_sharded; else 0)
  double view_point[3 * kMaxCams];  // of the cloud (a kernel argument, not a device constant: clouds of a batch
                                    // with different cameras run side by side)
};

// MODE 0: the default region (86^3 bits, built in LDS, written out once); 1: WIDE (128^3 bits = 256 KB: straight into the
// set's global row, which the host has cleared); 2: a region of run-time size (c_img.set_sd), also straight into global memory —
// image volumes of any size the general shadow kernel below takes
template <int MODE>
__global__ __launch_bounds__(SET_THREADS) void shadow_set_kernel(SetParams P) {
  constexpr bool WIDE = MODE != 0;
  const ImgConsts &K = c_img;
  const int SD = MODE == 2 ? K.set_sd : Vox<MODE == 1>::SD, SR = MODE == 2 ? K.set_sr : Vox<MODE == 1>::SR;
  const int SETWORDS = MODE == 2 ? (int)(((long long)SD * SD * SD + 31) / 32) : Vox<MODE == 1>::SETWORDS;
  __shared__ uint32_t lds_bits[WIDE ? 1 : Vox<false>::SETWORDS];
  const int set = blockIdx.x;
  const int tid = threadIdx.x;
  const int slot_s = P.set_meta[8 * set + 0];
  const int N = P.set_meta[8 * set + 1];
  // position in the cloud's ONE stream of shadow draws (hand_set.cpp:268-283): the plan's offset among this call's hand sets
  // + the draws of the sample ranges before it, when the cloud's samples are sharded over several contexts
  const unsigned long long off =
      P.lcg_base + (((unsigned long long)(uint32_t)P.set_meta[8 * set + 3] << 32) | (uint32_t)P.set_meta[8 * set + 2]);
  const int cam = P.set_meta[8 * set + 4];
  const float *nn = P.nn + (size_t)slot_s * 6 * P.cap;
  uint32_t *out = P.set_bits + (size_t)set * SETWORDS;
  uint32_t *bits = WIDE ? out : lds_bits;
  if (!WIDE) {
    for (int w = tid; w < SETWORDS; w += SET_THREADS) lds_bits[w] = 0u;
    __syncthreads();
  }
  const double *smp = P.frames + 12 * (size_t)slot_s;
  int ox = (int)floor(smp[0] * K.voxel_mult) - SR, oy = (int)floor(smp[1] * K.voxel_mult) - SR,
      oz = (int)floor(smp[2] * K.voxel_mult) - SR;
  // opaque to the optimiser: it re-associated v - (floor - SR) into (v - floor) + SR, two operations per coordinate and draw
  asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz));
  // shadow_vec = shadow_length * (center - view_point) / norm (hand_set.cpp:147-150)
  const double *cen = P.centers + 3 * (size_t)slot_s;
  double vec[3];
  for (int r = 0; r < 3; r++) vec[r] = cen[r] - P.view_point[3 * cam + r];
  const double nrm = sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
  for (int r = 0; r < 3; r++) vec[r] = K.shadow_length * vec[r] / nrm;
  uint32_t state = lcg_jump(0u, off + (unsigned long long)tid * (unsigned)K.num_shadow);
  for (int i = tid; i < N; i += SET_THREADS) {
    const double p0 = (double)nn[0 * P.cap + i], p1 = (double)nn[1 * P.cap + i], p2 = (double)nn[2 * P.cap + i];
    uint32_t st = state;
    for (int k = 0; k < K.num_shadow; k++) {
      const double t = (double)(int)lcg_step(st) * K.rand_inv;
      const int vx = (int)((p0 + t * vec[0]) * K.voxel_mult) - ox;
      const int vy = (int)((p1 + t * vec[1]) * K.voxel_mult) - oy;
      const int vz = (int)((p2 + t * vec[2]) * K.voxel_mult) - oz;
      if ((unsigned)vx < (unsigned)SD && (unsigned)vy < (unsigned)SD && (unsigned)vz < (unsigned)SD) {
        // 24-bit multiply-adds, spelled out: the compiler picks v_mad_u64_u32 for a 32-bit multiply-add
        unsigned bit;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(bit) : "v"(vx), "s"(SD), "v"(vy));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(bit) : "v"(bit), "s"(SD), "v"(vz));
        atomicOr(&bits[bit >> 5], 1u << (bit & 31));
      }
    }
    state = K.stride_a * state + K.stride_c;  // advance by SET_THREADS * num_shadow draws
  }
  if (!WIDE) {
    // Written out: only the rows (lines along z) inside the voxel bounding box of the set's candidate boxes, one voxel of margin
    // around each — shadow_image_kernel reads the rows of a candidate's window and keeps the bits its box test passes, and no bit
    // outside a box passes it, so whatever an earlier launch left in the rest of the row is never seen (129 MB of bitsets per
    // 5000 candidates were written in full before: half of the image stage's excess HBM traffic).
    __shared__ int s_bb[4];  // x0, x1, y0, y1 in region voxels
    if (tid == 0) {
      s_bb[0] = s_bb[2] = SD;
      s_bb[1] = s_bb[3] = -1;
    }
    __syncthreads();
    if (tid < P.slots && P.hand_cand[(size_t)slot_s * P.slots + tid] >= 0) {
      const gpd_hand &H = P.hands[(size_t)slot_s * P.slots + tid];
      const double lo[3] = {H.bottom, H.center - K.half_od, -1.0 * K.vol_height};
      const double hi[3] = {H.bottom + K.vol_depth, H.center + K.half_od, K.vol_height};
      double wmin[2] = {DBL_MAX, DBL_MAX}, wmax[2] = {-DBL_MAX, -DBL_MAX};
      for (int k = 0; k < 8; k++) {
        const double bx = (k & 1) ? hi[0] : lo[0], by = (k & 2) ? hi[1] : lo[1], bz = (k & 4) ? hi[2] : lo[2];
        for (int a = 0; a < 2; a++) {
          const double w = H.sample[a] + H.frame[3 * a] * bx + H.frame[3 * a + 1] * by + H.frame[3 * a + 2] * bz;
          wmin[a] = fmin(wmin[a], w);
          wmax[a] = fmax(wmax[a], w);
        }
      }
      // the window shadow_image_kernel opens for this box: floor(min) - 1 .. floor(max) + 1, here relative to the region
      atomicMin(&s_bb[0], (int)floor(wmin[0] * K.voxel_mult) - 1 - ox);
      atomicMax(&s_bb[1], (int)floor(wmax[0] * K.voxel_mult) + 1 - ox);
      atomicMin(&s_bb[2], (int)floor(wmin[1] * K.voxel_mult) - 1 - oy);
      atomicMax(&s_bb[3], (int)floor(wmax[1] * K.voxel_mult) + 1 - oy);
    }
    __syncthreads();
    const int x0 = max(s_bb[0], 0), x1 = min(s_bb[1], SD - 1), y0 = max(s_bb[2], 0), y1 = min(s_bb[3], SD - 1);
    if (x1 >= x0 && y1 >= y0) {
      const int wpr = (((y1 - y0 + 1) * SD + 31) >> 5) + 1;  // words one x-slab's rows can span
      for (int idx = tid; idx < (x1 - x0 + 1) * wpr; idx += SET_THREADS) {
        const int x = x0 + idx / wpr, k = idx - (idx / wpr) * wpr;
        const int first = (x * SD + y0) * SD, last = (x * SD + y1 + 1) * SD - 1;  // the slab's bits
        const int w = (first >> 5) + k;
        if (w <= (last >> 5)) out[w] = lds_bits[w];
      }
    }
  }
}

// planar [n][C][3600] <-> HWC [n][3600][C] (cv::Mat CV_8UC(C), the reference's image layout)
__global__ void planar_to_hwc_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int C, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t per = (size_t)kPix * C;
  const size_t img = i / per, r = i - img * per;
  const int pix = (int)(r / C), c = (int)(r - (size_t)pix * C);
  dst[i] = src[img * per + (size_t)c * kPix + pix];
}
__global__ void hwc_to_planar_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int C, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t per = (size_t)kPix * C;
  const size_t img = i / per, r = i - img * per;
  const int c = (int)(r / kPix), pix = (int)(r - (size_t)c * kPix);
  dst[i] = src[img * per + (size_t)pix * C + c];
}
hipError_t planar_to_hwc(const uint8_t *src, uint8_t *dst, int n, int C, hipStream_t stream) {
  const size_t total = (size_t)n * kPix * C;
  if (!total) return hipSuccess;
  planar_to_hwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(src, dst, C, total);
  return hipGetLastError();
}
hipError_t hwc_to_planar(const uint8_t *src, uint8_t *dst, int n, int C, hipStream_t stream) {
  const size_t total = (size_t)n * kPix * C;
  if (!total) return hipSuccess;
  hwc_to_planar_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(src, dst, C, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
// smallest positive double x with floor((x/len)/(1.0/60)) >= k — the exact image of the
// reference's cell formula (image_strategy.cpp:92-102 applied to transformPointsToUnitImage's
// quotient, :72-90), found by bisection on the bit pattern with the same IEEE operations.
static double cell_threshold(double len, int k) {
  const double cellsize = 1.0 / (double)kImg;
  auto f = [&](double x) { return std::floor((x / len) / cellsize) >= (double)k; };
  uint64_t lo = 0, hi;  // f(bits lo) false, f(bits hi) true
  double top = len * 2.0;
  std::memcpy(&hi, &top, sizeof(hi));
  while (hi - lo > 1) {
    const uint64_t mid = lo + (hi - lo) / 2;
    double x;
    std::memcpy(&x, &mid, sizeof(x));
    if (f(x))
      hi = mid;
    else
      lo = mid;
  }
  double x;
  std::memcpy(&x, &hi, sizeof(x));
  return x;
}

void image_cell_thresholds(double len, double *out) {
  out[0] = 0.0;
  for (int k = 1; k < kImg; k++) out[k] = cell_threshold(len, k);
  out[kImg] = DBL_MAX;
}

void images_free(ImageState &im) {
  void *ptrs[] = {im.d_images, im.d_images_hwc, im.d_status, im.d_set_bits, im.d_overflow, im.d_overflow2, im.d_pts_overflow, im.d_pts_scratch, im.d_huge_scratch};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (im.ev_fork) (void)hipEventDestroy(im.ev_fork);
  if (im.ev_join) (void)hipEventDestroy(im.ev_join);
  if (im.aux) (void)hipStreamDestroy(im.aux);
  im = ImageState();
}

// Image buffers for up to n candidates and `shadow_sets` (live hand set, camera) voxel bitsets of the geometry in p
// (grow with 25 % slack, never shrink; a growth stalls the device).
int images_reserve(const gpd_params &p, ImageState &im, int n, int shadow_sets) {
  const int C = p.image_num_channels;
  if (!im.d_status) {
    HIP_RET(hipMalloc(&im.d_status, 4 * sizeof(int32_t)));  // [0] flags, [1..3] the counters of the three overflow lists: ONE memset per launch
    HIP_RET(hipMemset(im.d_status, 0, 4 * sizeof(int32_t)));
    HIP_RET(hipMalloc(&im.d_pts_scratch, (size_t)LGRID * PTS_SCRATCH_BYTES));
  }
  if (n > im.capacity) {
    note_alloc(__func__);
    void *ptrs[] = {im.d_images, im.d_images_hwc, im.d_overflow, im.d_overflow2, im.d_pts_overflow};
    for (void *q : ptrs)
      if (q) (void)hipFree(q);
    im.d_overflow = nullptr;
    im.d_overflow2 = nullptr;
    im.d_pts_overflow = nullptr;
    im.d_images = nullptr;
    im.d_images_hwc = nullptr;
    im.capacity = 0;
    const int cap = n + n / 4;  // slack: the clouds of a batch differ a little
    HIP_RET(hipMalloc(&im.d_images, (size_t)cap * kPix * C));
    HIP_RET(hipMalloc(&im.d_overflow, (size_t)cap * sizeof(int32_t)));  // list (its counter: d_status[1])
    HIP_RET(hipMalloc(&im.d_overflow2, (size_t)cap * sizeof(int32_t)));  // ... of the large instantiation, for the general kernel
    HIP_RET(hipMalloc(&im.d_pts_overflow, (size_t)cap * sizeof(int32_t)));
    im.capacity = cap;
  }
  {
    // which voxel windows the shadow kernels need (Vox<WIDE>): the window of a candidate must hold the diagonal of the
    // image box (+ 3 voxels of margins), the region of a set every box of the set.  A box spans x in [bottom, bottom + depth]
    // with init_bite - hand_depth <= bottom <= 0, y in center -+ width / 2 with |center| <= (outer_diameter - finger_width) / 2,
    // |z| <= height (finger_hand.cpp:155-168, image_strategy.cpp:53-70).
    const double vox = 0.003;
    const double diag = std::sqrt(p.volume_depth * p.volume_depth + p.volume_width * p.volume_width + 4.0 * p.volume_height * p.volume_height);
    const double rx = std::fmax(p.hand_depth - p.init_bite, p.volume_depth);
    const double ry = 0.5 * (p.hand_outer_diameter - p.finger_width) + 0.5 * p.volume_width;
    const double reach = std::sqrt(rx * rx + ry * ry + p.volume_height * p.volume_height);
    const double need_win = std::ceil(diag / vox) + 3.0, need_reach = std::ceil(reach / vox) + 1.0;
    im.wide = need_win > (double)Vox<false>::VD || need_reach > (double)(Vox<false>::SR - 1);
    // beyond the wide windows as well: the general shadow kernel with a set region of the size this geometry needs
    // (the reference has no limit, hand_set.cpp:138; what is left of ours: 256 voxels = 0.77 m of window edge, 65535 voxels in a box)
    im.huge = need_win > (double)Vox<true>::VD || need_reach > (double)(Vox<true>::SR - 1);
    if (im.huge) {
      if (need_reach + 2.0 > 640.0) {  // 1280^3 bits = 262 MB per hand set and camera: an image volume of metres is a mistake, not a gripper
        set_error("images: an image volume with %.2f m of reach is beyond the shadow region this library allocates", reach);
        return GPD_ERR_CAPACITY;
      }
      im.set_sr = (int)need_reach + 2;
      im.set_sd = 2 * im.set_sr;
    } else {
      im.set_sr = im.wide ? Vox<true>::SR : Vox<false>::SR;
      im.set_sd = im.wide ? Vox<true>::SD : Vox<false>::SD;
    }
  }
  const size_t setwords = (size_t)(((long long)im.set_sd * im.set_sd * im.set_sd + 31) / 32);
  if (C == 15 && (shadow_sets > im.cap_shadow_sets || setwords > im.cap_setwords)) {
    note_alloc(__func__);
    if (im.d_set_bits) (void)hipFree(im.d_set_bits);
    im.d_set_bits = nullptr;
    im.cap_shadow_sets = 0;
    const int want = shadow_sets > im.cap_shadow_sets ? shadow_sets : im.cap_shadow_sets;
    const int cap = want + want / 4;
    const size_t words = setwords > im.cap_setwords ? setwords : im.cap_setwords;
    HIP_RET(hipMalloc(&im.d_set_bits, (size_t)cap * words * sizeof(uint32_t)));
    // defined contents from the first launch on: shadow_set_kernel<0> writes only the rows its candidates' windows cover, and
    // the words around them keep what is there (zero, or an earlier launch's voxels — see the reader's note in shadow_image_kernel)
    HIP_RET(hipMemset(im.d_set_bits, 0, (size_t)cap * words * sizeof(uint32_t)));
    im.cap_shadow_sets = cap;
    im.cap_setwords = words;  // (a larger row also serves the smaller regions)
  }
  if (C == 15 && (im.wide || im.huge) && !im.d_huge_scratch) {
    // the general shadow kernel's list rows, one per workgroup of its persistent grid; the default geometry cannot reach it
    // (its box holds fewer voxel cells than the large instantiation lists)
    note_alloc(__func__);
    HIP_RET(hipMalloc(&im.d_huge_scratch, (size_t)LGRID * HUGE_SCRATCH_BYTES));
  }
  return GPD_OK;
}

// The candidate list itself (which hands, in which order, where each hand set starts in the LCG stream) is
// built on the device by plan_kernel; its sizes are in pl.h_summary by now.  This sizes the image buffers,
// prepares the constant block of the image geometry and launches the kernels.
int images_run(const gpd_params &p, const Cloud &c, const SearchState &s, const Plan &pl, ImageState &im, hipStream_t stream) {
  const int C = p.image_num_channels;
  const PlanSummary &sm = *pl.h_summary;
  const int n = sm.num_candidates;
  im.num_candidates = n;
  im.channels = C;
  im.slots = p.num_hand_axes * p.num_orientations;
  im.stat_sets = sm.live_sets;
  im.stat_sum_set_ni = sm.sum_set_ni;
  im.stat_sum_cand_ni = sm.sum_cand_ni;
  im.num_shadow_sets = sm.num_shadow_sets;
  std::memcpy(im.view_points, c.view_points, sizeof(im.view_points));
  {
    const int rc = images_reserve(p, im, n, im.num_shadow_sets);
    if (rc) return rc;
  }
  HIP_RET(hipMemsetAsync(im.d_status, 0, 4 * sizeof(int32_t), stream));  // the flags and the overflow counters of this launch
  if (n == 0) return GPD_OK;
  ImgConsts k;
  std::memset(&k, 0, sizeof(k));
  k.vol_depth = p.volume_depth;
  k.vol_width = p.volume_width;
  k.vol_height = p.volume_height;
  k.half_od = p.volume_width / 2.0;
  k.dbl_h = 2.0 * p.volume_height;
  k.C = C;
  k.nproj = (C <= 3) ? 1 : 3;
  k.per = (C == 15) ? 5 : (C == 12 ? 4 : C);
  // shadow_length_ = max(volume_depth, volume_height/2, volume_width) (image_15_channels_strategy.h:70-75)
  k.shadow_length = std::fmax(std::fmax(p.volume_depth, p.volume_height / 2.0), p.volume_width);
  k.voxel = 0.003;
  k.voxel_mult = 1.0 / 0.003;
  k.rand_inv = 1.0 / 32767.0;
  k.num_shadow = (int)std::floor(k.shadow_length / k.voxel);  // plan_kernel places the sets in the LCG stream with the same count
  k.len[0] = k.vol_depth;
  k.len[1] = k.vol_width;
  k.len[2] = k.dbl_h;
  k.true_div = 0;
  k.set_sd = im.set_sd;
  k.set_sr = im.set_sr;
  {
    // thresholds and the div_len() self-check depend on the box extents only: computed once per
    // geometry (the self-check alone is ~2 ms of host time)
    struct AxisCache {
      double len = -1.0, thr[kImg + 1], inv_len = 0.0;
      int true_div = 0;
    };
    static AxisCache cache[3];
    static std::mutex cache_mutex;
    std::lock_guard<std::mutex> lock(cache_mutex);
    for (int a = 0; a < 3; a++) {
      AxisCache &ac = cache[a];
      if (ac.len != k.len[a]) {
        ac.len = k.len[a];
        image_cell_thresholds(k.len[a], ac.thr);
        ac.inv_len = 1.0 / k.len[a];
        ac.true_div = 0;
        // self-check of div_len(): the FMA sequence must reproduce the IEEE quotient
        uint64_t rs = 88172645463325252ull;
        for (int i = 0; i < 200000 && !ac.true_div; i++) {
          rs ^= rs << 13;
          rs ^= rs >> 7;
          rs ^= rs << 17;
          double x = (double)(rs >> 11) * (1.0 / 9007199254740992.0) * k.len[a];
          if (i & 1) x = (double)(float)x;
          if (i < kImg) x = ac.thr[i];
          const double q = x * ac.inv_len;
          const double r = std::fma(-k.len[a], q, x);
          if (std::fma(r, ac.inv_len, q) != x / k.len[a]) ac.true_div = 1;
        }
      }
      k.inv_cell[a] = (double)kImg / k.len[a];
      std::memcpy(k.thr[a], ac.thr, sizeof(ac.thr));
      k.inv_len[a] = ac.inv_len;
      if (ac.true_div) k.true_div = 1;
    }
  }
  {  // affine map of SET_THREADS * num_shadow LCG steps
    uint32_t a = 214013u, cc = 2531011u, A = 1u, Cc = 0u;
    unsigned long long nsteps = (unsigned long long)SET_THREADS * (unsigned)k.num_shadow;
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
  const unsigned char *kb = reinterpret_cast<const unsigned char *>(&k);
  if (im.consts.size() != sizeof(k) || std::memcmp(im.consts.data(), kb, sizeof(k)) != 0) im.consts.assign(kb, kb + sizeof(k));
  return images_launch(s, pl, im, stream, /*counters_clean=*/true);
}

// c_img is ONE block per device, shared by every context of the process on that device.  Under the
// lock: a context whose block (geometry, view points of its cloud) differs from the loaded one waits
// for the device — kernels of another context, on another stream, may still be reading it — and
// loads its own; the caller keeps the lock until its kernels are enqueued.  Same-stream order covers
// the context's own earlier kernels.
static std::mutex g_img_mutex;
static std::vector<unsigned char> g_img_loaded[64];

static int load_img_consts(const std::vector<unsigned char> &want, hipStream_t stream) {
  if (want.size() != sizeof(ImgConsts)) return GPD_ERR_STATE;
  int dev = 0;
  HIP_RET(hipGetDevice(&dev));
  std::vector<unsigned char> &loaded = g_img_loaded[dev & 63];
  if (loaded == want) return GPD_OK;
  if (!loaded.empty()) HIP_RET(hipDeviceSynchronize());
  HIP_RET(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_img), want.data(), sizeof(ImgConsts), 0, hipMemcpyHostToDevice, stream));
  // the block counts as loaded only once the copy has landed: a context with the same values skips
  // the copy and launches on ITS stream, which is not ordered after this one
  HIP_RET(hipStreamSynchronize(stream));
  loaded = want;
  return GPD_OK;
}

// Launches the image kernels over the candidate list resident on the device.  Nothing here waits for the
// device: the large instantiations read the length of their queues on the device, capacity flags
// accumulate in im.d_status (read by the caller with its results).
int images_launch(const SearchState &s, const Plan &pl, ImageState &im, hipStream_t stream, bool counters_clean) {
  const int n = im.num_candidates;
  if (n <= 0) return GPD_OK;
  // the three overflow counters of the launch (the flags in d_status[0] accumulate over re-launches)
  if (!counters_clean) HIP_RET(hipMemsetAsync(im.d_status + 1, 0, 3 * sizeof(int32_t), stream));
  std::lock_guard<std::mutex> consts_lock(g_img_mutex);
  {
    const int rc = load_img_consts(im.consts, stream);
    if (rc) return rc;
  }
  ImgParams ip;
  std::memset(&ip, 0, sizeof(ip));
  ip.nn = s.d_nn;
  ip.cap = s.nn_cap;
  ip.centers = s.d_centers;
  ip.hands = s.d_hands;
  ip.cand_hand = pl.d_cand_hand;
  ip.meta = pl.d_cand_meta;
  ip.images = im.d_images;
  ip.status = im.d_status;
  static unsigned long long *d_dbg = nullptr;
  ip.dbg = nullptr;
  ip.exit_after = prof_env("GPD_IMG_EXIT") ? atoi(prof_env("GPD_IMG_EXIT")) : 0;
  if (prof_env("GPD_IMG_TIMING")) {
    if (!d_dbg) HIP_RET(hipMalloc(&d_dbg, 32 * sizeof(unsigned long long)));
    HIP_RET(hipMemsetAsync(d_dbg, 0, 32 * sizeof(unsigned long long), stream));
    ip.dbg = d_dbg;
  }
  ip.set_bits = im.d_set_bits;
  ip.cand_list = nullptr;
  ip.cand_count = nullptr;
  ip.num_cand = n;
  ip.overflow_list = nullptr;
  ip.overflow_count = nullptr;
  // normals + depth on the side stream (15 channels: beside the shadow kernels).  Nearly every box holds fewer than
  // PT_CAP points; the others are queued and redone by the instantiation that keeps its point arrays in a global
  // scratch row.
  if (!im.aux) {
    HIP_RET(hipStreamCreate(&im.aux));
    HIP_RET(hipEventCreateWithFlags(&im.ev_fork, hipEventDisableTiming));
    HIP_RET(hipEventCreateWithFlags(&im.ev_join, hipEventDisableTiming));
  }
  static const bool serial = prof_env("GPD_IMG_SERIAL") != nullptr;  // profiling aid: each image kernel alone on the chip
  // profiling aid: unused dynamic LDS per workgroup of the two per-candidate kernels (8192: one workgroup per CU instead of two)
  static const size_t lds_pad = prof_env("GPD_IMG_LDS_PAD") ? (size_t)atoi(prof_env("GPD_IMG_LDS_PAD")) : 0;
  hipStream_t pts_stream = (im.channels == 15 && !ip.dbg && im.side_stream && !serial) ? im.aux : stream;
  if (pts_stream != stream) {
    HIP_RET(hipEventRecord(im.ev_fork, stream));
    HIP_RET(hipStreamWaitEvent(pts_stream, im.ev_fork, 0));
  }
  ip.pts_overflow_list = im.d_pts_overflow;
  ip.pts_overflow_count = im.d_status + 3;
  ip.pts_scratch = nullptr;
  grasp_image_kernel<false><<<8 * ((n + 7) / 8), IMG_THREADS, lds_pad, pts_stream>>>(ip);
  HIP_RET(hipGetLastError());
  {
    ImgParams ib = ip;
    ib.cand_list = im.d_pts_overflow;
    ib.cand_count = im.d_status + 3;
    ib.pts_overflow_list = nullptr;
    ib.pts_overflow_count = nullptr;
    ib.pts_scratch = im.d_pts_scratch;
    grasp_image_kernel<true><<<LGRID, IMG_THREADS, 0, pts_stream>>>(ib);
    HIP_RET(hipGetLastError());
  }
  ip.pts_overflow_list = nullptr;
  ip.pts_overflow_count = nullptr;
  if (im.num_shadow_sets > 0) {
    SetParams sp;
    sp.nn = s.d_nn;
    sp.cap = s.nn_cap;
    sp.centers = s.d_centers;
    sp.frames = s.d_frames;
    sp.set_meta = pl.d_set_meta;
    sp.set_bits = im.d_set_bits;
    sp.hands = s.d_hands;
    sp.hand_cand = pl.d_hand_cand;
    sp.slots = im.slots;
    sp.lcg_base = im.lcg_base;
    std::memcpy(sp.view_point, im.view_points, sizeof(sp.view_point));
    if (im.huge) {
      const size_t setwords = (size_t)(((long long)im.set_sd * im.set_sd * im.set_sd + 31) / 32);
      HIP_RET(hipMemsetAsync(im.d_set_bits, 0, (size_t)im.num_shadow_sets * setwords * sizeof(uint32_t), stream));
      shadow_set_kernel<2><<<im.num_shadow_sets, SET_THREADS, 0, stream>>>(sp);
    } else if (im.wide) {
      HIP_RET(hipMemsetAsync(im.d_set_bits, 0, (size_t)im.num_shadow_sets * Vox<true>::SETWORDS * sizeof(uint32_t), stream));
      shadow_set_kernel<1><<<im.num_shadow_sets, SET_THREADS, 0, stream>>>(sp);
    } else {
      shadow_set_kernel<0><<<im.num_shadow_sets, SET_THREADS, 0, stream>>>(sp);
    }
    HIP_RET(hipGetLastError());
  }
  ip.cand_list = nullptr;
  ip.cand_count = nullptr;
  ip.num_cand = n;
  ip.overflow_list = im.d_overflow;
  ip.overflow_count = im.d_status + 1;
  if (im.channels == 15) {
    // most boxes fit the two-per-CU instantiation; the few that do not are queued by it and
    // redone by the large one
    ImgParams ib = ip;
    ib.cand_list = im.d_overflow;
    ib.cand_count = im.d_status + 1;
    ib.overflow_list = nullptr;
    ib.overflow_count = nullptr;
    if (im.huge) {
      // an image volume beyond the windows of the tuned kernels: every candidate through the general one
      ImgParams ia = ip;
      ia.overflow_list = nullptr;
      ia.overflow_count = nullptr;
      ia.huge_scratch = im.d_huge_scratch;
      shadow_image_any_kernel<<<LGRID, IMG_THREADS, 0, stream>>>(ia);
    } else if (im.wide) {
      // a wide box can hold more voxels than the large instantiation lists: it queues those for the general kernel
      ib.overflow_list = im.d_overflow2;
      ib.overflow_count = im.d_status + 2;
      shadow_image_kernel<SH_CAP, true><<<8 * ((n + 7) / 8), IMG_THREADS, 0, stream>>>(ip);
      HIP_RET(hipGetLastError());
      shadow_image_kernel<SH_CAP_BIG, true><<<LGRID, IMG_THREADS, 0, stream>>>(ib);
      HIP_RET(hipGetLastError());
      ImgParams ia = ip;
      ia.cand_list = im.d_overflow2;
      ia.cand_count = im.d_status + 2;
      ia.overflow_list = nullptr;
      ia.overflow_count = nullptr;
      ia.huge_scratch = im.d_huge_scratch;
      shadow_image_any_kernel<<<LGRID, IMG_THREADS, 0, stream>>>(ia);
    } else {
      shadow_image_kernel<SH_CAP, false><<<8 * ((n + 7) / 8), IMG_THREADS, lds_pad, stream>>>(ip);
      HIP_RET(hipGetLastError());
      shadow_image_kernel<SH_CAP_BIG, false><<<LGRID, IMG_THREADS, 0, stream>>>(ib);
    }
    HIP_RET(hipGetLastError());
  }
  HIP_RET(hipEventRecord(im.ev_join, pts_stream));
  if (pts_stream != stream) HIP_RET(hipStreamWaitEvent(stream, im.ev_join, 0));
  if (ip.dbg) {
    unsigned long long h[32];
    HIP_RET(hipMemcpyAsync(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipStreamSynchronize(stream));
    static const char *names[14] = {"sh_extract",  "sh_list",         "sh_count_place", "sh_walk_tail", "sh_final",
                                    "pts_collect", "pts_count_place", "pts_walk_tail",  "pts_final",    "sh_nzlist",
                                    "sh_walk_loop", "pts_nzlist",     "pts_zero",       "pts_walk_loop"};
    unsigned long long tot = 0;
    for (int i = 0; i < 14; i++) tot += h[i];
    for (int i = 0; i < 14; i++)
      fprintf(stderr, "[img-timing] %-16s %10.1f kcycles/cand  %5.1f%%\n", names[i], (double)h[i] / n / 1e3,
              100.0 * h[i] / (double)tot);
    fprintf(stderr, "[img-timing] shadow voxels in box: max %llu mean %.0f; in-box points: max %llu mean %.0f\n", h[28],
            (double)h[29] / n, h[30], (double)h[31] / n);
  }
  return GPD_OK;
}

// text of the capacity flags an image kernel left in d_status
void images_status_text(int status, char *buf, size_t len) {
  snprintf(buf, len, "images: kernel capacity exceeded (flags %d: 1 image box beyond %d voxels of window edge, 2 in-box points > %d, "
           "4 shadow voxels in a box > %d)", status, HUGE_WIN, PT_CAP_BIG, HUGE_CAP);
}

}  // namespace gpd
