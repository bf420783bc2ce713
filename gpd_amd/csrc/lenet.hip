// LeNet forward for batches of 60x60xC u8 grasp images on gfx950.
//
// Replaces EigenClassifier::forward (net/eigen_classifier.cpp:81-128), ConvLayer
// (net/conv_layer.cpp:26-98: im2col + GEMM, batch 1) and DenseLayer
// (net/dense_layer.cpp:6-15: GEMV re-streaming 14.4 MB of FC1 weights per image).
//
// Numerics contract (matches oracle/gpd_oracle.cpp): every dot product is an f32
// fmaf chain in ascending k (k = c*25 + kh*5 + kw for the convolutions, k = flatten
// index for the dense layers) starting from 0, bias added last.  The f32-input MFMA
// used for FC1 is bitwise such a chain (cdna_hip_programming.md §3).
//
// Kernels:
//   conv1_mfma   planar u8 images + padded weight rows in LDS; pixel-stationary: every lane keeps
//                the converted 6x6 patch of its pooled pixel in registers and k = 1 MFMAs
//                (16x16x1 + 4x4x1 = 20 filters) stream the weights past it; channels whose
//                patches are all zero are skipped (exact); 2x2 max-pool fused
//                                                               -> pool1 [n][20][28][28]
//   conv2_mfma   persistent; pool1 planes + 96 KB of weights in LDS, implicit GEMM + pool
//                (+ 2 filters riding in the same waves on the VALU), output in the reference's
//                flatten order j = pixel*50 + channel            -> flat  [n][7200]
//   fc1_mfma     [500 x 7200] x [7200 x n] on v_mfma_f32_16x16x4_f32, 128 x 16..128 tiles picked per launch, + bias, ReLU
//                                                               -> fc1t  [500][n]
//   fc2_score    2 x 500 chains per image, score = y1 - y0       -> scores[n]
#include <cstdlib>

#include "gpd_internal.h"
#include <type_traits>

namespace gpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// conv1 + pool1, pixel-stationary form.  Two images per workgroup; a task is a chunk of 64 pooled
// pixels, lane <-> pooled pixel.  Per live channel a lane converts ITS 6x6 input patch once (36
// converts for 200 MFMAs) and keeps it in registers as the B operands; the weights stream from LDS
// as the A operands of k = 1 MFMAs:
//   v_mfma_f32_16x16x1 (4 blocks): D_b[f][j] += w[f][k] * x_b[j]   filters 0..15 x 64 pixels
//   v_mfma_f32_4x4x1  (16 blocks): D_b[f][j] += w[16+f][k] * x_b[j] filters 16..19 x 64 pixels
// once per tap and per conv pixel of the 2x2 pool window.  One k per instruction: every output is
// the k-ascending fmaf chain of the oracle, and 16 + 4 filters fill the two tile shapes exactly.
// Compared with the implicit GEMM on 16x16x4 (one byte read + one convert per MFMA, 54 % issue
// overhead measured) the per-MFMA operand traffic drops to 0.3 LDS reads and 0.18 converts.
// max(a_i + b) == max(a_i) + b (rounding is monotone), so the bias is added after the pool.
// Zero skipping: grasp images are ~70 % zeros.  A channel in which the patches of all 64 lanes are
// zero adds exact zeros to every chain (fmaf(w, 0, acc) == acc for finite w; acc is never -0) and
// is skipped — decided from the patch bytes themselves (one ballot), which are fetched a channel
// ahead anyway.  Pooled pixels are numbered strip-major (strips of 8, 8, 8 and 4 columns), so a chunk is
// a compact 8 x 8 block (strips of 7 columns, ~9 x 7 blocks, measured the same skip rate): 28 % of the (chunk, channel) pairs drop out (14 % with row-major chunks;
// profiles/r01m_conv1_skip_stats.txt).
// Persistent workgroups (one per CU: 142 KB of LDS) over a two-slot image ring.  Workgroup b of G takes the images
// b, b + G, b + 2 G, ... (neighbouring candidates have similar images: dealing them out keeps the workgroups' loads
// alike, contiguous ranges differed by 12 %), numbers their pooled pixels through (784 per image) and cuts them into
// chunks of 64 — 245 chunks for 20 images, whatever the image boundaries.  The s-th image of the workgroup sits in
// slot s & 1; the wave that completes the last chunk touching image s loads image s + 2 into its slot while the other
// waves work on image s + 1 (a chunk takes ~45 us of wave time, the load ~10), so the task stream never runs dry
// inside a workgroup.  The two-images-per-workgroup version this replaces spent, per workgroup of 350 kcycles: 17 in the
// prologue (images + weights), 34 with the average wave waiting for the last one at the end of the 25-task queue, and
// another 10 % of the kernel between workgroups (dispatch, 9.77 rounds on 256 CUs): 1.69 ms for 1.30 ms of tasks.
constexpr int C1_WAVES = 8, C1_THREADS = 64 * C1_WAVES, C1_PIX = 784;
// pool1 in HBM: [n][20][784] floats with the 784 pooled pixels of a plane IN CHUNK ORDER (the strip-major numbering
// below, not row-major): the 64 pixels of a chunk are then 256 consecutive bytes per filter plane and every store of a
// wave is whole cache lines.  (Row-major planes made a chunk's stores runs of 28-32 bytes: 672 MB of write traffic for
// 314 MB of pool1, profiles/r02_traffic.json; rows padded to whole sectors still 614 MB.)  conv2 undoes the
// permutation when it stages a plane into LDS (c1_pixel_of).
constexpr int P1_PLANE = 784, P1_IMG = 20 * P1_PLANE;
// pooled pixel (row, column) of chunk-order position pc: strips of 8, 8, 8 and 4 columns, 28 rows each
__host__ __device__ inline void c1_pixel_of(int pc, int &py, int &px) {
  const int strip = pc / (28 * 8), within = pc - strip * (28 * 8);
  const int sw = strip < 3 ? 8 : 4;
  py = within / sw;
  px = strip * 8 + (within - py * sw);
}

// chunks of the workgroup that touch its j-th image (the number of releases its slot waits for): the pooled pixels of its
// images are numbered through and cut into chunks of 64; the chunk that holds an image's last pixels always exists
__device__ inline int c1_chunks_of_image(int j) { return (C1_PIX * j + C1_PIX - 1) / 64 - (C1_PIX * j) / 64 + 1; }

template <int C>
__global__ __launch_bounds__(C1_THREADS) void conv1_mfma_kernel(const uint8_t *__restrict__ images, const float *__restrict__ wp,
                                                               const float *__restrict__ bias, float *__restrict__ out, int n,
                                                               unsigned long long *__restrict__ stats, int fault, int *__restrict__ queue) {
  __shared__ __attribute__((aligned(16))) uint8_t s_img[2][C * kPix];
  __shared__ __attribute__((aligned(16))) float s_wa[16 * C * 28];  // filters 0..15: [f][c][25 taps + 3 pad]
  __shared__ __attribute__((aligned(16))) float s_wb[4 * C * 28];   // filters 16..19
  __shared__ int s_next;
  __shared__ int s_slot_img[2];  // local sequence number of the image held by the slot (published after its bytes), -1: none
  __shared__ int s_gid[2];       // ... and which image of the launch that is
  __shared__ int s_refs[2];      // chunks of this workgroup that still have to release the slot's image
  __shared__ int s_abort;        // a wave gave up waiting for a slot: the others of the workgroup leave too
  __shared__ int s_last;         // the workgroup's last local sequence number; INT_MAX until the launch's queue ran dry
  __shared__ int s_drawn;        // local sequence numbers that have drawn their image (the draws happen in this order)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // The workgroup's images: its own number first, then whatever the launch's counter (`queue`, zero at launch) hands out —
  // drawn one slot refill ahead, strictly in the order of the local sequence numbers, so that "the queue is dry" at number q
  // means it is dry for every later one (two refills racing for the counter could otherwise leave a hole: a drawn image
  // behind an empty number).  Round 2 dealt the images out (b, b + G, ...): neighbouring candidates have similar images, and
  // with the zero skipping a fixed share made the slowest workgroup 13 % longer than the average; the first counter version
  // drew at release time without the ordering and was withdrawn.  Inside gpd_hip_detect_batch a fixed share also made the
  // kernel as slow as its last-started workgroup.
  const int G = gridDim.x;
  if ((int)blockIdx.x >= n) return;
  if (tid == 0) {
    const int g1 = G + atomicAdd(queue, 1);
    s_gid[0] = blockIdx.x;
    s_gid[1] = g1;
    s_last = g1 < n ? INT_MAX : 0;
    s_drawn = 2;
  }
  __syncthreads();
  for (int q = 0; q < 2; q++) {  // the first two images, by everybody
    if (q > *(volatile int *)&s_last) break;
    const uint4 *src = reinterpret_cast<const uint4 *>(images + (size_t)s_gid[q] * kPix * C);
    uint4 *dst = reinterpret_cast<uint4 *>(s_img[q & 1]);
    for (int i = tid; i < kPix * C / 16; i += C1_THREADS) dst[i] = src[i];
  }
  {  // padded weight table [20][C][28] (built once on the host side of the C-ABI): filters 0..15, then 16..19
    const uint4 *src = reinterpret_cast<const uint4 *>(wp);
    uint4 *da = reinterpret_cast<uint4 *>(s_wa), *db = reinterpret_cast<uint4 *>(s_wb);
    for (int i = tid; i < 16 * C * 7; i += C1_THREADS) da[i] = src[i];
    for (int i = tid; i < 4 * C * 7; i += C1_THREADS) db[i] = src[16 * C * 7 + i];
  }
  if (tid == 0) {
    s_next = 0;
    s_abort = 0;
    s_slot_img[0] = s_slot_img[1] = -1;
    for (int q = 0; q < 2; q++)
      if (q <= s_last) {
        s_slot_img[q] = q;
        s_refs[q] = c1_chunks_of_image(q);
      }
  }
  __syncthreads();
  const float *wa_row = s_wa + (lane & 15) * C * 28;
  const float *wb_row = s_wb + (lane & 3) * C * 28;
  unsigned n_live = 0, n_tasks = 0;  // (chunk, channel) pairs this wave executed / chunks it took: the executed-FLOP count
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(&s_next, 1);
    task = __builtin_amdgcn_readfirstlane(task);
    // the images this chunk reads (the second one only when the chunk straddles an image boundary).  Whether they exist is
    // known at the latest when their slots would have been refilled: s_last turns from INT_MAX into the last number then.
    const int ia = (64 * task) / C1_PIX, ib0 = (64 * task + 63) / C1_PIX;
    if (ia > *(volatile int *)&s_last) break;  // (chunks are taken in ascending order: every later one is past the end too)
    n_tasks++;
    // (a slot is refilled within ~10 us of its release; a wait of ~0.5 s can only be a broken protocol.  The wave then
    //  raises the launch's error word — stats[2], which the host turns into GPD_ERR_HIP at its next synchronisation — and
    //  the workgroup's abort flag, and every wave of the workgroup leaves: the kernel ends, the context stays usable,
    //  the scores of this launch are not handed out.  It used to be a __builtin_trap(), which killed the HIP context.)
    bool give_up = false, past_end = false;
    int ib = ib0;
    for (int spins = 0;; spins++) {
      const int last = *(volatile int *)&s_last;
      if (ia > last) {
        past_end = true;
        break;
      }
      ib = ib0 < last ? ib0 : last;
      if (*(volatile int *)&s_slot_img[ia & 1] == ia && *(volatile int *)&s_slot_img[ib & 1] == ib) break;
      __builtin_amdgcn_s_sleep(4);
      if (*(volatile int *)&s_abort || spins > (1 << 22)) {
        give_up = true;
        break;
      }
    }
    if (past_end) {
      n_tasks--;
      break;
    }
    if (give_up) {
      if (lane == 0) {
        *(volatile int *)&s_abort = 1;
        if (stats) atomicMax(&stats[2], 1ull);
      }
      break;
    }
    __threadfence_block();  // the image bytes are read after the slot numbers
    // pooled pixel of the lane: image, then strip-major pixel number -> (row, column).  Lanes past the last pixel of
    // the workgroup redo it and store nothing.
    const int g = task * 64 + lane;
    const int npix = (ib + 1) * C1_PIX;  // pixels up to the end of the chunk's last image (ib < ib0: the workgroup's last one)
    const int gc = g < npix ? g : npix - 1;
    const int q = gc >= C1_PIX * (ia + 1) ? ia + 1 : ia, pc = gc - q * C1_PIX;
    int py, px;
    c1_pixel_of(pc, py, px);
    // where the lane's pixel goes in pool1; -1: nothing to store
    const int ooff = g < npix ? s_gid[q & 1] * P1_IMG + pc : -1;  // chunk order: see P1_PLANE
    const uint8_t *base = s_img[q & 1] + (2 * py) * kImg + 2 * px;
    f32x16 acc16[4];
    f32x4 acc4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
      for (int r = 0; r < 16; r++) acc16[e][r] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; r++) acc4[e][r] = 0.f;
    }
    uint32_t raw[6][3];  // the patch bytes of the channel about to be looked at: 6 rows x 3 byte pairs
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int e = 0; e < 3; e++) raw[r][e] = *reinterpret_cast<const uint16_t *>(base + r * kImg + 2 * e);
    // Per channel three phases that are kept apart (sched_barrier): LDS requests, converts, then 200
    // MFMAs back to back.  A VALU instruction between two MFMAs of one wave costs ~9 cycles of the
    // matrix pipe, the same instruction issued by ANOTHER wave of the SIMD while this one streams
    // MFMAs costs nothing (measured) — so each wave keeps its MFMA run pure and the waves of a SIMD
    // fall out of phase.
    for (int c = 0;;) {
      // next live channel (the accumulators are not touched in this search loop)
      bool live = false;
      for (;;) {
        uint32_t any = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int e = 0; e < 3; e++) any |= raw[r][e];
        live = __builtin_amdgcn_ballot_w64(any != 0u) != 0ull;
        if (live || ++c >= C) break;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int e = 0; e < 3; e++) raw[r][e] = *reinterpret_cast<const uint16_t *>(base + c * kPix + r * kImg + 2 * e);
      }
      if (!live) break;
      n_live++;
      const uint8_t *nb = base + (c + 1 < C ? c + 1 : c) * kPix;
      float4 ta[7], tb[7];
#pragma unroll
      for (int g = 0; g < 7; g++) {
        ta[g] = *reinterpret_cast<const float4 *>(wa_row + c * 28 + 4 * g);
        tb[g] = *reinterpret_cast<const float4 *>(wb_row + c * 28 + 4 * g);
      }
      float patch[6][6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
          patch[r][2 * e] = (float)(raw[r][e] & 0xff);
          patch[r][2 * e + 1] = (float)(raw[r][e] >> 8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int e = 0; e < 3; e++) raw[r][e] = *reinterpret_cast<const uint16_t *>(nb + r * kImg + 2 * e);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 5; kh++) {
#pragma unroll
        for (int kw = 0; kw < 5; kw++) {
          const int t = kh * 5 + kw;
          const float ka = t % 4 == 0 ? ta[t / 4].x : t % 4 == 1 ? ta[t / 4].y : t % 4 == 2 ? ta[t / 4].z : ta[t / 4].w;
          const float kb = t % 4 == 0 ? tb[t / 4].x : t % 4 == 1 ? tb[t / 4].y : t % 4 == 2 ? tb[t / 4].z : tb[t / 4].w;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float x = patch[kh + (e >> 1)][kw + (e & 1)];
            acc16[e] = __builtin_amdgcn_mfma_f32_16x16x1f32(ka, x, acc16[e], 0, 0, 0);
            acc4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(kb, x, acc4[e], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (++c >= C) break;
    }
    {
      // D of the 4-block 16x16x1: lane l, register 4 b + r  <->  the pixel of lane 16 b + (l & 15), filter 4 (l >> 4) + r
      float *o = out;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int po = __shfl(ooff, 16 * b + (lane & 15));
        if (po >= 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int reg = 4 * b + r, f = 4 * (lane >> 4) + r;
            o[f * P1_PLANE + po] = fmaxf(fmaxf(acc16[0][reg], acc16[1][reg]), fmaxf(acc16[2][reg], acc16[3][reg])) + bias[f];
          }
        }
      }
      if (ooff >= 0) {
#pragma unroll
        for (int f = 0; f < 4; f++)
          o[(16 + f) * P1_PLANE + ooff] = fmaxf(fmaxf(acc4[0][f], acc4[1][f]), fmaxf(acc4[2][f], acc4[3][f])) + bias[16 + f];
      }
    }
    // release the chunk's image(s); the wave that releases an image last refills its slot with the image two ahead
    int refill = 0;  // bit 0: slot of ia, bit 1: slot of ib
    if (lane == 0) {
      if (atomicSub(&s_refs[ia & 1], 1) == 1) refill |= 1;
      if (ib != ia && atomicSub(&s_refs[ib & 1], 1) == 1) refill |= 2;
    }
    refill = __builtin_amdgcn_readfirstlane(refill);
    for (int r = 0; r < 2; r++) {
      if (!(refill >> r & 1)) continue;
      const int nxt = (r ? ib : ia) + 2;
      // the image for local number nxt: drawn when the numbers before it have drawn theirs
      int gid = -1;
      if (lane == 0) {
        for (int spins = 0;; spins++) {
          if (nxt > *(volatile int *)&s_last) break;  // the queue ran dry at an earlier number
          if (*(volatile int *)&s_drawn == nxt) {
            // (s_last is written before s_drawn: seeing the ticket, look at s_last once more — the number before may have found
            //  the queue dry between the two reads above, and its s_last must not be overwritten by a later number's)
            if (nxt > *(volatile int *)&s_last) break;
            gid = G + atomicAdd(queue, 1);
            if (gid >= n) {
              gid = -1;
              atomicMin(&s_last, nxt - 1);  // dry: this workgroup ends with nxt - 1 (waiting chunks see it and leave)
            }
            __threadfence_block();
            *(volatile int *)&s_drawn = nxt + 1;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          if (*(volatile int *)&s_abort || spins > (1 << 22)) {  // the same watchdog as for the slots
            *(volatile int *)&s_abort = 1;
            if (stats) atomicMax(&stats[2], 1ull);
            break;
          }
        }
      }
      gid = __builtin_amdgcn_readfirstlane(gid);
      if (gid < 0) continue;
      const uint4 *src = reinterpret_cast<const uint4 *>(images + (size_t)gid * kPix * C);
      uint4 *dst = reinterpret_cast<uint4 *>(s_img[nxt & 1]);
      constexpr int NV = kPix * C / 16;
      for (int i0 = 0; i0 < NV; i0 += 64 * 8) {  // eight loads in flight per lane
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = i0 + 64 * u + lane;
          v[u] = src[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = i0 + 64 * u + lane;
          if (i < NV) dst[i] = v[u];
        }
      }
      if (lane == 0) s_gid[nxt & 1] = gid;
      if (lane == 0) s_refs[nxt & 1] = c1_chunks_of_image(nxt);
      __threadfence_block();  // the bytes and the release count are in LDS before the slot is published
      // (fault: the test hook GPD_C1_FAULT=1 makes workgroup 0 "forget" to publish — the protocol bug the watchdog is for)
      if (lane == 0 && !(fault && blockIdx.x == 0)) *(volatile int *)&s_slot_img[nxt & 1] = nxt;
    }
  }
  if (stats && lane == 0) {  // two atomics per wave and launch: what bench.py's roofline divides by
    atomicAdd(&stats[0], (unsigned long long)n_live);
    atomicAdd(&stats[1], (unsigned long long)n_tasks * C);
  }
}

// ---------------------------------------------------------------------------
// conv2 is an implicit GEMM D[f][pix] = sum_k W[f][k] X[k][pix] on v_mfma_f32_16x16x4_f32,
// k = c*25 + kh*5 + kw ascending — the MFMA accumulates k..k+3 in order, so each output is
// still the oracle's fmaf chain (cdna_hip_programming.md §3, "bit-for-bit a k-ordered chain").
//   A operand (lane l): W[f = l&15][k0 + (l>>4)]   (k-major weight copy in LDS)
//   B operand (lane l): X[k0 + (l>>4)][pix = l&15] (pool1 planes in LDS)
//   D (lane l, reg r):  pixel l&15, filter 4*(l>>4) + r
// A pixel tile is 2 rows x 8 columns of conv outputs, so the 2x2 max-pool is two lane
// exchanges (xor 1, xor 8).  max(a_i + b) == max(a_i) + b (rounding is monotone).
// The 2 filters that do not fill a 16-row tile run as a direct convolution on the VALU.
// ---------------------------------------------------------------------------
// conv2 + pool2.  Persistent workgroups (one per CU) loop over images: the k-major weights of
// filters 0..47 (96 KB) stay in LDS next to one image's pool1 planes (62.7 KB).
//   waves 0-11  filters 0..47 on v_mfma_f32_16x16x4_f32: a wave owns one band of two output rows
//               (3 pixel tiles of 2x8) x 3 filter tiles = 9 independent accumulators; per k-step
//               3 weight reads + 3 input reads feed 9 MFMAs.  48 = 3 x 16 filters fill the MFMA
//               tiles exactly (50 would pad 64-row tiles to 78 %).
//               filters 48, 49 ride along in the same waves as a direct convolution on the VALU
//               (lane <-> conv pixel of the band, scalar weights): 8 fmas per k-step in the
//               shadow of the other waves' MFMAs.  A separate VALU wave for them was the
//               critical path of the workgroup (1.53 ms vs 1.16 ms without it).
// k-ascending fmaf chains as before; output in the reference's flatten order pixel*50 + filter.
constexpr int C2_MFMA_WAVES = 12, C2_THREADS = 64 * C2_MFMA_WAVES;

// tap offset inside a 4-channel period: k in [0, 100) -> c * 784 + kh * 28 + kw
__host__ __device__ constexpr int c2_tap_off(int k) { return (k / 25) * 784 + ((k % 25) / 5) * 28 + (k % 5); }

__global__ __launch_bounds__(C2_THREADS) void conv2_mfma_kernel(const float *__restrict__ pool1, const float *__restrict__ wt,
                                                                const float *__restrict__ w, const float *__restrict__ bias,
                                                                float *__restrict__ flat, int n, int *__restrict__ queue) {
  // one array: reads one step past the image / the weights (operand prefetch of the last step)
  // land in the next region, never outside the allocation
  __shared__ __attribute__((aligned(16))) float s_all[20 * 784 + 500 * 48 + 4 * 48];
  float *s_in = s_all, *s_w = s_all + 20 * 784;
  const int tid = threadIdx.x;
  for (int i = tid; i < 504 * 48; i += C2_THREADS) {
    const int k = i / 48, f = i - k * 48;
    s_w[i] = k < 500 ? wt[k * 50 + f] : 0.f;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int kq = lane >> 4, j = lane & 15;
  // The k axis is walked in periods of 100 taps = 4 channels = 25 MFMA steps: inside a period
  // every tap offset is a compile-time constant, except the B operand's, whose tap 4s + kq
  // depends on the lane group: those 25 offsets live in registers.
  int offp[25];
#pragma unroll
  for (int sI = 0; sI < 25; sI++) {
    const int k = 4 * sI + kq;
    const int c = k / 25, tap = k - 25 * c;
    offp[sI] = c * 784 + (tap / 5) * 28 + tap % 5;
  }
  // the next image's planes travel HBM -> registers while the current image is convolved (six named
  // 16-byte registers per lane: an array would live in scratch memory)
  constexpr int IN_V = P1_IMG / 4;  // float4 of one image in HBM (planes in conv1's chunk order)
  static_assert(IN_V <= 6 * C2_THREADS, "prefetch registers");
  float4 p0, p1, p2, p3, p4, p5;
  p0 = p1 = p2 = p3 = p4 = p5 = make_float4(0.f, 0.f, 0.f, 0.f);
#define C2_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define C2_FETCH1(v) \
  if (tid + v * C2_THREADS < IN_V) p##v = fsrc[tid + v * C2_THREADS];
#define C2_STAGE1(v)                                                                                         \
  {                                                                                                          \
    const int iv = tid + v * C2_THREADS;                                                                     \
    if (iv < IN_V) {                                                                                         \
      const int pl_ = iv / (P1_PLANE / 4), pc_ = 4 * (iv - pl_ * (P1_PLANE / 4));                            \
      int py_, px_;                                                                                          \
      c1_pixel_of(pc_, py_, px_); /* four consecutive positions = four consecutive columns of one row */     \
      *reinterpret_cast<float4 *>(s_in + pl_ * 784 + py_ * 28 + px_) = p##v;                                 \
    }                                                                                                        \
  }
#define C2_FETCH(IMG)                                                                              \
  {                                                                                                \
    const float4 *fsrc = reinterpret_cast<const float4 *>(pool1 + (size_t)(IMG) * P1_IMG);         \
    C2_EACH(C2_FETCH1)                                                                             \
  }
  // The images are drawn from a counter of the launch (`queue`, zero at launch: the first gridDim.x images are the
  // workgroups' own numbers, the counter hands out the rest), one ahead of the image being convolved.  A fixed deal
  // (image b, b + G, ...) made the kernel as slow as its LAST-STARTED workgroup: in gpd_hip_detect_batch the other lane's
  // kernels hold CUs when this one is launched, a 156 KB workgroup starts on such a CU only once it is empty, and with a
  // fixed share it then finished that much later (conv2 1.51 -> 1.73 ms per 6077 images inside the batch, r04_batch_timeline.txt).
  __shared__ int s_nxt;
  int img = blockIdx.x;
  if (img < n) C2_FETCH(img)
  for (; img < n;) {
    if (tid == 0) s_nxt = (int)gridDim.x + atomicAdd(queue, 1);
    __syncthreads();  // previous image fully consumed (and the weights are in place); s_nxt is visible
    C2_EACH(C2_STAGE1)
    const int nxt = s_nxt;
    __syncthreads();
    if (nxt < n) C2_FETCH(nxt)
    {
      const int rp = wave;  // output rows 2rp, 2rp+1
      f32x4 acc[3][3];      // [pixel tile][filter tile]
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int ft = 0; ft < 3; ft++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[t][ft][r] = 0.f;
      // per-lane bases, advanced by one period per outer iteration
      const float *xin = s_in + (2 * rp + (j >> 3)) * 28 + (j & 7);  // MFMA B operand: pixel j of tile 0
      const float *aw = s_w + kq * 48 + j;                            // MFMA A operand: W[k0 + kq][f = j]
      // filters 48, 49 on the VALU: lane = r * 32 + col is conv pixel (2rp + r, col), col < 24
      const float *xv = s_in + (2 * rp + (lane >> 5)) * 28 + min(lane & 31, 23);
      const float *__restrict__ wf = w + (size_t)48 * 500;
      float t48 = 0.f, t49 = 0.f;
      // operands of step s+1 are requested before the nine MFMAs of step s
      float a_cur[3], b_cur[3], a_nxt[3], b_nxt[3], xc[4], xn[4];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        a_cur[q] = aw[16 * q];
        b_cur[q] = xin[offp[0] + 8 * q];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) xc[i] = xv[c2_tap_off(i)];
      for (int it = 0; it < 5; it++) {
#pragma unroll
        for (int sI = 0; sI < 25; sI++) {
          // ---- requests for step s+1 (step 0 of the next period when s = 24)
          const int s1 = sI + 1 < 25 ? sI + 1 : 0;
          const int bump = sI + 1 < 25 ? 0 : 4 * 784;
          const float *x1 = xin + offp[s1] + bump;
#pragma unroll
          for (int q = 0; q < 3; q++) {
            a_nxt[q] = aw[(sI + 1) * 4 * 48 + 16 * q];
            b_nxt[q] = x1[8 * q];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) xn[i] = xv[c2_tap_off(4 * s1 + i) + bump];
          // ---- filters 48, 49: four taps of this step, k ascending
          const float4 wa = *reinterpret_cast<const float4 *>(wf + 4 * sI);
          const float4 wb = *reinterpret_cast<const float4 *>(wf + 500 + 4 * sI);
          t48 = __builtin_fmaf(wa.x, xc[0], t48);
          t49 = __builtin_fmaf(wb.x, xc[0], t49);
          t48 = __builtin_fmaf(wa.y, xc[1], t48);
          t49 = __builtin_fmaf(wb.y, xc[1], t49);
          t48 = __builtin_fmaf(wa.z, xc[2], t48);
          t49 = __builtin_fmaf(wb.z, xc[2], t49);
          t48 = __builtin_fmaf(wa.w, xc[3], t48);
          t49 = __builtin_fmaf(wb.w, xc[3], t49);
          __builtin_amdgcn_sched_barrier(0);
          // the nine MFMAs at raised wave priority: the SIMD's other two waves are in their VALU tail / operand requests at
          // any time, and the arbiter otherwise lets those instructions in between this wave's MFMAs (conv2 1.21 -> 1.19 ms;
          // the same around conv1's and ip1's MFMA runs changes nothing: profiles/NOTES.md E)
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int t = 0; t < 3; t++)
#pragma unroll
            for (int ft = 0; ft < 3; ft++)
              acc[t][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[ft], b_cur[t], acc[t][ft], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 3; q++) {
            a_cur[q] = a_nxt[q];
            b_cur[q] = b_nxt[q];
          }
#pragma unroll
          for (int i = 0; i < 4; i++) xc[i] = xn[i];
        }
        xin += 4 * 784;
        xv += 4 * 784;
        aw += 100 * 48;
        wf += 100;
      }
      {
        // pool 2x2: column partner lane ^ 1, row partner lane ^ 32; bias after the max
        float m48 = fmaxf(t48, __shfl_xor(t48, 1)), m49 = fmaxf(t49, __shfl_xor(t49, 1));
        m48 = fmaxf(m48, __shfl_xor(m48, 32));
        m49 = fmaxf(m49, __shfl_xor(m49, 32));
        const int col = lane & 31;
        if (lane < 32 && col < 24 && !(col & 1)) {
          float *o = flat + (size_t)img * kFc1In + (rp * 12 + (col >> 1)) * 50 + 48;
          *reinterpret_cast<float2 *>(o) = make_float2(m48 + bias[48], m49 + bias[49]);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int ft = 0; ft < 3; ft++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float x = acc[t][ft][r];
            x = fmaxf(x, __shfl_xor(x, 1));
            x = fmaxf(x, __shfl_xor(x, 8));
            acc[t][ft][r] = x;
          }
        }
        if (!(j & 1) && !(j & 8)) {
          // flatten: index = pixel*50 + channel (eigen_classifier.cpp:103-107)
          float *o = flat + (size_t)img * kFc1In + (rp * 12 + 4 * t + ((j & 7) >> 1)) * 50;
#pragma unroll
          for (int ft = 0; ft < 3; ft++) {
            const int f0 = 16 * ft + 4 * kq;
            float4 v = make_float4(acc[t][ft][0] + bias[f0], acc[t][ft][1] + bias[f0 + 1], acc[t][ft][2] + bias[f0 + 2],
                                   acc[t][ft][3] + bias[f0 + 3]);
            *reinterpret_cast<float2 *>(o + f0) = make_float2(v.x, v.y);
            *reinterpret_cast<float2 *>(o + f0 + 2) = make_float2(v.z, v.w);
          }
        }
      }
    }
    img = nxt;
  }
}

#undef C2_FETCH
#undef C2_FETCH1
#undef C2_STAGE1
#undef C2_EACH

// ---------------------------------------------------------------------------
// FC1 on f32 MFMA:  D[u][m] = sum_k W[k][u] * X[m][k]   (W = ip1 weights, column-major 500x7200 ==
// row-major [7200][500], dense_layer.cpp:7; X = flat, the pixel-major flatten of eigen_classifier.cpp:103-107).
//
// Workgroup tile 128 (u) x 16*NT (m), eight waves, each 16 (u) x 16*NT (m) on v_mfma_f32_16x16x4_f32:
//   A operand (lane l): W[k0 + (l >> 4)][u = l & 15]   B operand: X[m = l & 15][k0 + (l >> 4)]
//   D (lane l, reg r):  u = 4 * (l >> 4) + r, m = l & 15
// K is walked in order, 4 per MFMA, so every output is still the k-ascending fmaf chain of the oracle.
//
// The GEMM is small (36 GFLOP at n = 5000) against 1024 SIMDs, so the tile shape decides the balance:
// 4 u-tiles x ceil(n / (16 NT)) m-tiles, NT picked per launch so that the m-tiles fill the 64 workgroup
// columns of the chip in whole rounds (n = 5000: NT = 5, 4 x 63 = 252 workgroups on 256 CUs, one round —
// the 64 x 64 tiles of round 1 left a 3-tile makespan on 2.47 tiles per SIMD).  128-wide u-tiles also halve
// the operand traffic per flop from L2 (2.3 GB per launch at 64 x 64).  The four u-tiles of an m-tile run
// on ONE XCD (workgroup L runs on XCD L % 8), so the image rows are fetched from HBM once.
// LDS: two buffers of K = 64 (W rows padded to 144 floats, X k-major and swizzled: both the transposing writes and
// the MFMA operand reads are bank-conflict free, see fc_sx); global loads run two steps ahead through two register sets.
// ---------------------------------------------------------------------------
constexpr int FC_BU = 128, FC_BK = 64, FC_THREADS = 512, FC_SW = FC_BU + 16;
constexpr int FC_STEPS = (kFc1In + FC_BK - 1) / FC_BK, FC_TAIL_KS = (kFc1In - (FC_STEPS - 1) * FC_BK) / 4;  // 113 steps, the last one 8 k-quads
static_assert(kFc1In % 4 == 0 && FC_TAIL_KS >= 1 && FC_TAIL_KS <= FC_BK / 4, "K steps");
// X tile in LDS: k-major rows of stride = 16 (mod 32) with the column of element (k, m) swizzled to
// m ^ (((k >> 2) & 7) << 2).  The MFMA operand read (lanes: two k of one k-quad x 16 consecutive m) then touches
// 32 different banks — row k + 1 sits 16 banks further, and the swizzle maps an aligned 16-block of m onto an aligned
// 16-block — and so does the transposing write of the loader (lanes: 4 rows m x 8 k-quads, one k of each quad: the
// swizzle spreads the 8 quads over the banks the 4 rows leave free).  (Row stride = 17 (mod 32) without the swizzle cost
// one 2-way conflict per read: 40 % of fc1's LDS cycles, profiles/r02_pmc_sq.txt of the first collection.)
__host__ __device__ constexpr int fc_sx(int bm) { return (bm + 31) / 32 * 32 + 16; }

template <int NT>
__global__ __launch_bounds__(FC_THREADS) void fc1_mfma_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                              const float *__restrict__ X, float *__restrict__ out_t, int n, int ld_out) {
  constexpr int BM = 16 * NT, SX = fc_sx(BM);
  constexpr int STAGE = FC_BK * FC_SW + FC_BK * SX;  // floats per stage: W tile, then X tile
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  static_assert(2 * STAGE * 4 <= 160 * 1024, "LDS");
  const int tid = threadIdx.x;
  const int L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3;
  const int u0 = (slot & 3) * FC_BU;
  const int m0 = ((slot >> 2) * 8 + xcd) * BM;
  if (m0 >= n) return;
  const int wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, j = lane & 15;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) acc[t][r] = 0.f;
  // loader roles.  W tile: 64 rows x 32 float4 -> four per thread.  X tile: BM rows x 16 float4 (sixteen lanes
  // per image row: 256-byte segments), ceil(BM * 16 / 512) per thread.
  constexpr int W_PT = FC_BK * 32 / FC_THREADS, XV = BM * (FC_BK / 4), X_PT = (XV + FC_THREADS - 1) / FC_THREADS;
  const int wk = tid >> 5, wuq = tid & 31;  // + 16 rows per further float4
  // every load is unconditional (out-of-range lanes and the k past 7200 of the last step read a clamped address and
  // their value is dropped or never multiplied): straight-line loads let the compiler count them, so the wait
  // before a stage's LDS stores covers only ITS loads, not the ones issued for the stage after it
  const bool w_ok = u0 + 4 * wuq < kFc1Out;
  const float *wsrc = W + min(u0 + 4 * wuq, kFc1Out - 4);
  const float *xsrc[X_PT];
  int xm[X_PT], xkq[X_PT];
#pragma unroll
  for (int i = 0; i < X_PT; i++) {
    const int v = min(tid + i * FC_THREADS, XV - 1);
    xm[i] = v / (FC_BK / 4);
    xkq[i] = v % (FC_BK / 4);
    xsrc[i] = X + (size_t)min(m0 + xm[i], n - 1) * kFc1In;
  }
  float4 rw[2][W_PT], rx[2][X_PT];
  auto fetch = [&](int step, float4(&w2)[W_PT], float4(&x2)[X_PT]) {
    const int k0 = step * FC_BK;
#pragma unroll
    for (int i = 0; i < W_PT; i++) w2[i] = *reinterpret_cast<const float4 *>(wsrc + (size_t)min(k0 + wk + 16 * i, kFc1In - 1) * kFc1Out);
#pragma unroll
    for (int i = 0; i < X_PT; i++) x2[i] = *reinterpret_cast<const float4 *>(xsrc[i] + min(k0 + 4 * xkq[i], kFc1In - 4));
  };
  auto stage = [&](int buf, const float4(&w2)[W_PT], const float4(&x2)[X_PT]) {
    float *sw = smem + buf * STAGE, *sx = sw + FC_BK * FC_SW;
#pragma unroll
    for (int i = 0; i < W_PT; i++)
      *reinterpret_cast<float4 *>(sw + (wk + 16 * i) * FC_SW + 4 * wuq) = w_ok ? w2[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < X_PT; i++)
      if (tid + i * FC_THREADS < XV) {
        float *d = sx + (4 * xkq[i]) * SX + (xm[i] ^ ((xkq[i] & 7) << 2));
        d[0] = x2[i].x;
        d[SX] = x2[i].y;
        d[2 * SX] = x2[i].z;
        d[3 * SX] = x2[i].w;
      }
  };
  // (Requesting the operands of four k-quads in one run of LDS reads ahead of 4 NT back-to-back MFMAs — pure MFMA
  // runs as in conv1 — measured 0.362 ms against 0.336 ms for this per-quad interleave; storing the next tile in the
  // middle of the MFMAs instead of at the end 0.339.)
  auto compute = [&](int buf, auto nks_tag) {
    constexpr int NKS = decltype(nks_tag)::value;
    const float *sw = smem + buf * STAGE + g * FC_SW + wave * 16 + j;
    const float *sx = smem + buf * STAGE + FC_BK * FC_SW + g * SX;
    float a_cur = sw[0], a_nxt = 0.f, b_cur[NT], b_nxt[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) b_cur[t] = sx[16 * t + j];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      if (ks + 1 < NKS) {  // operands of the next four k are requested before this step's MFMAs ...
        a_nxt = sw[(4 * (ks + 1)) * FC_SW];
#pragma unroll
        for (int t = 0; t < NT; t++) b_nxt[t] = sx[(4 * (ks + 1)) * SX + ((16 * t + j) ^ (((ks + 1) & 7) << 2))];
      }
      __builtin_amdgcn_sched_barrier(0);  // ... and arrive while they run (the scheduler would sink the reads to their uses)
#pragma unroll
      for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
#pragma unroll
      for (int t = 0; t < NT; t++) b_cur[t] = b_nxt[t];
    }
  };
  using FullStep = std::integral_constant<int, FC_BK / 4>;
  using TailStep = std::integral_constant<int, FC_TAIL_KS>;
  // prologue: stage 0 in LDS, stage 1 in register set 1
  fetch(0, rw[0], rx[0]);
  stage(0, rw[0], rx[0]);
  fetch(1, rw[1], rx[1]);
  __syncthreads();
  // step t: loads of step t + 2 go out first (register set t % 2), the MFMAs of buffer t % 2 run (K = 64: one
  // barrier per 16 k-quads — with K = 32 stages the eight waves met twice as often and MfmaUtil stood at 64 %),
  // the loads of step t + 1 (issued a whole step ago) are written to the other buffer, whose readers finished
  // before the last barrier.
  // No conditions on the step number: the tail refetches the last step and stages a tile nobody reads, and in
  // exchange the compiler counts the outstanding loads exactly (with `if (t + 2 < FC_STEPS)` around the fetch the two
  // paths merged to vmcnt(0) at the loop head and before the LDS stores).
  auto step = [&](int t, auto nks_tag, float4(&w_far)[W_PT], float4(&x_far)[X_PT], const float4(&w_near)[W_PT], const float4(&x_near)[X_PT]) {
    fetch(min(t + 2, FC_STEPS - 1), w_far, x_far);
    compute(t & 1, nks_tag);
    stage((t + 1) & 1, w_near, x_near);
    __syncthreads();
  };
  static_assert(FC_STEPS % 2 == 1, "the loop below ends on an even step");
  int t = 0;
  for (; t + 1 < FC_STEPS; t += 2) {
    step(t, FullStep{}, rw[0], rx[0], rw[1], rx[1]);
    step(t + 1, FullStep{}, rw[1], rx[1], rw[0], rx[0]);
  }
  compute(t & 1, TailStep{});  // k = 7168 .. 7199
  // bias, ReLU (eigen_classifier.cpp:113), transposed store for ip2
#pragma unroll
  for (int tt = 0; tt < NT; tt++) {
    const int m = m0 + 16 * tt + j;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int u = u0 + wave * 16 + 4 * g + r;
      if (u < kFc1Out && m < n) out_t[(size_t)u * ld_out + m] = fmaxf(acc[tt][r] + bias[u], 0.f);
    }
  }
}

template <int NT>
static void fc1_launch(const float *W, const float *bias, const float *X, float *out_t, int n, int ld_out, hipStream_t stream) {
  const int m_tiles = (n + 16 * NT - 1) / (16 * NT);
  const int groups = (m_tiles + 7) / 8;
  fc1_mfma_kernel<NT><<<groups * 4 * 8, FC_THREADS, 0, stream>>>(W, bias, X, out_t, n, ld_out);
}
// m-tile width: the smallest multiple of 16 (at most 128) whose tiles fill the chip's 64 workgroup columns
// (256 CUs / 4 u-tiles) in r whole rounds, r as small as possible
static int fc1_pick_nt(int n) {
  for (int r = 1;; r++) {
    const int nt = (n + 64 * r * 16 - 1) / (64 * r * 16);
    if (nt <= 8) return nt < 1 ? 1 : nt;
  }
}

// FC2 + score (dense_layer.cpp:6-15 with 2 units; eigen_classifier.cpp:74: score = y1 - y0).  Each logit is ONE fmaf
// chain over the 500 ip1 units in ascending order — the definition the oracle and the reference pins share — so it
// cannot be folded into ip1's epilogue as partial sums over ip1's u-tiles (that changes the rounding).  What can be
// spread is everything else.  32 images per workgroup; phase 1, all 512 threads: the images' 500 ip1 outputs into the LDS,
// many loads in flight per thread (the one-wave version of rounds 1-5 walked its chain straight off global memory, ten
// loads at a time: 29 us per 5000 images for 10 MFLOP) — on the split path this is also where ip1's four K quarters meet:
// ((q0 + q1) + q2) + q3 + bias, ReLU (eigen_classifier.cpp:113; a kernel of its own until round 6, 9 us), read as 16-byte
// pieces from ONE contiguous 256 KB (fc1p_index: ip1 writes its partial sums blocked by 32 images); phase 2, one wave: the
// two logits of an image are a lane pair, the chain reads the LDS.
constexpr int FC2_THREADS = 512, FC2_IMGS = 32;
template <bool COMBINE>
__global__ __launch_bounds__(FC2_THREADS) void fc2_score_kernel(const float *__restrict__ fc1p, const float *__restrict__ b1,
                                                               const float *__restrict__ fc1t, const float *__restrict__ w,
                                                               const float *__restrict__ b, float *__restrict__ scores, int n, int ld) {
  __shared__ __attribute__((aligned(16))) float hs[kFc1Out][FC2_IMGS];
  __shared__ float ws[2 * kFc1Out];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * kFc1Out; i += FC2_THREADS) ws[i] = w[i];
  if constexpr (COMBINE) {
    const float4 *q = reinterpret_cast<const float4 *>(fc1p + fc1p_index(blockIdx.x * FC2_IMGS, 0, 0));
    constexpr int PLANE4 = kFc1Out * FC2_IMGS / 4;  // float4s of one quarter
#pragma unroll 4
    for (int i = tid; i < PLANE4; i += FC2_THREADS) {
      const float4 a = q[i], c = q[PLANE4 + i], d = q[2 * PLANE4 + i], e = q[3 * PLANE4 + i];
      const float bu = b1[i >> 3];
      reinterpret_cast<float4 *>(&hs[0][0])[i] = make_float4(fmaxf(((a.x + c.x) + d.x) + e.x + bu, 0.f), fmaxf(((a.y + c.y) + d.y) + e.y + bu, 0.f),
                                                             fmaxf(((a.z + c.z) + d.z) + e.z + bu, 0.f), fmaxf(((a.w + c.w) + d.w) + e.w + bu, 0.f));
    }
  } else {
    const int mi = tid & (FC2_IMGS - 1);
    const int m = blockIdx.x * FC2_IMGS + mi, mc = m < n ? m : n - 1;
#pragma unroll 8
    for (int j = tid / FC2_IMGS; j < kFc1Out; j += FC2_THREADS / FC2_IMGS) hs[j][mi] = fc1t[(size_t)j * ld + mc];
  }
  __syncthreads();
  if (tid >= 64) return;
  const int which = tid >> 5;  // logit 0 in lanes 0..31, logit 1 in lanes 32..63
  const int mi = tid & 31, m = blockIdx.x * FC2_IMGS + mi;
  float y = 0.f;
#pragma unroll 20
  for (int j = 0; j < kFc1Out; j++) y = __builtin_fmaf(ws[2 * j + which], hs[j][mi], y);
  y += b[which];
  const float other = __shfl_xor(y, 32);
  if (which == 1 && m < n) scores[m] = y - other;  // y1 - y0
}

// ---------------------------------------------------------------------------
hipError_t lenet_scratch_reserve(LeNetScratch &s, int n) {
  if (n <= s.capacity) return hipSuccess;
  const int num_cus = s.num_cus;
  note_alloc(__func__);
  lenet_scratch_free(s);
  s.num_cus = num_cus;
  n = (n + n / 4 + 3) & ~3;  // slack: the clouds of a batch differ a little, every growth stalls the device; a multiple of 4: the row length of the transposed ip1 buffers (16-byte accesses)
  hipError_t e;
  if ((e = hipMalloc(&s.pool1, (size_t)n * P1_IMG * sizeof(float))) != hipSuccess) return e;
  if ((e = hipMalloc(&s.flat, (size_t)n * kFc1In * sizeof(float))) != hipSuccess) return e;
  constexpr size_t kXld = kLenetXld;  // 7200 + 96 zeros, which no kernel ever writes
  const size_t xs_rows = ((size_t)n + 15) & ~(size_t)15;  // whole 16-image blocks (lenet_fast.hip f3_blocked)
  if ((e = hipMalloc(&s.xs, 3 * xs_rows * kXld * sizeof(unsigned short))) != hipSuccess) return e;
  if ((e = hipMemset(s.xs, 0, 3 * xs_rows * kXld * sizeof(unsigned short))) != hipSuccess) return e;
  if ((e = hipMalloc(&s.fc1t, (size_t)n * kFc1Out * sizeof(float))) != hipSuccess) return e;
  if ((e = hipMalloc(&s.fc1p, (((size_t)n + 31) & ~(size_t)31) * 4 * kFc1Out * sizeof(float))) != hipSuccess) return e;  // whole 32-image blocks (fc1p_index)
  if ((e = hipMalloc(&s.c1_stats, 4 * sizeof(unsigned long long))) != hipSuccess) return e;
  if ((e = hipMemset(s.c1_stats, 0, 4 * sizeof(unsigned long long))) != hipSuccess) return e;
  s.capacity = n;
  return hipSuccess;
}

int lenet_check(LeNetScratch &s) {
  if (!s.c1_stats) return GPD_OK;
  unsigned long long flag = 0;
  if (hipMemcpy(&flag, s.c1_stats + 2, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess) {
    set_error("lenet: reading the launch error word failed");
    return GPD_ERR_HIP;
  }
  if (!flag) return GPD_OK;
  (void)hipMemset(s.c1_stats + 2, 0, sizeof(flag));
  set_error("lenet: conv1 gave up waiting for an image slot (slot protocol timeout); the scores of that launch are invalid");
  return GPD_ERR_HIP;
}

void lenet_scratch_free(LeNetScratch &s) {
  if (s.pool1) (void)hipFree(s.pool1);
  if (s.flat) (void)hipFree(s.flat);
  if (s.xs) (void)hipFree(s.xs);
  if (s.fc1t) (void)hipFree(s.fc1t);
  if (s.fc1p) (void)hipFree(s.fc1p);
  if (s.c1_stats) (void)hipFree(s.c1_stats);
  s = LeNetScratch();
}

// kernel_events (may be null): three events recorded after conv1, conv2 and fc1 of the first chunk,
// so that a caller bracketing the call with its own events gets the four kernel durations.
hipError_t lenet_forward(const LeNetWeights &w, LeNetScratch &s, const uint8_t *d_images, int n, float *d_scores,
                         hipStream_t stream, hipEvent_t *kernel_events) {
  if (n <= 0) return hipSuccess;
  const int kChunk = 65536;  // images per pass: 6.1 GB of scratch (pool1 + flat + fc1) at the full chunk — sized for 288 GB of HBM
  if (!s.num_cus) {  // persistent conv2 workgroups: one per CU (256 on MI355X)
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) s.num_cus = prop.multiProcessorCount;
    if (s.num_cus <= 0) s.num_cus = 256;
  }
  const int num_cus = s.num_cus;
  hipError_t e = lenet_scratch_reserve(s, n < kChunk ? n : kChunk);
  if (e != hipSuccess) return e;
  for (int off = 0; off < n; off += kChunk) {
    const int m = (n - off < kChunk) ? (n - off) : kChunk;
    const uint8_t *img = d_images + (size_t)off * kPix * w.channels;
    int *queue = reinterpret_cast<int *>(s.c1_stats + 3);  // the image counters of the two conv launches (conv1's, conv2's): one memset for both
    if (hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream) != hipSuccess) return hipGetLastError();
    if (w.mode == GPD_LENET_SPLIT) {
      // the default: int8 / bf16 MFMA kernels on exactly split operands (lenet_fast.hip)
      if ((e = lenet_forward_fast(w, s, img, m, d_scores + off, stream, kernel_events && off == 0 ? kernel_events : nullptr, queue)) != hipSuccess)
        return e;
    } else {
    // persistent conv1: one workgroup per CU, at least two images each
    const int fault = prof_env("GPD_C1_FAULT") != nullptr;  // test hook of the slot watchdog (tests/test_gpu_lenet_stress.py)
    const int c1_grid = m / 2 < 1 ? 1 : (m / 2 < num_cus ? m / 2 : num_cus);
    switch (w.channels) {
      case 15: conv1_mfma_kernel<15><<<c1_grid, C1_THREADS, 0, stream>>>(img, w.c1wp, w.c1b, s.pool1, m, s.c1_stats, fault, queue); break;
      case 12: conv1_mfma_kernel<12><<<c1_grid, C1_THREADS, 0, stream>>>(img, w.c1wp, w.c1b, s.pool1, m, s.c1_stats, fault, queue); break;
      case 3: conv1_mfma_kernel<3><<<c1_grid, C1_THREADS, 0, stream>>>(img, w.c1wp, w.c1b, s.pool1, m, s.c1_stats, fault, queue); break;
      case 1: conv1_mfma_kernel<1><<<c1_grid, C1_THREADS, 0, stream>>>(img, w.c1wp, w.c1b, s.pool1, m, s.c1_stats, fault, queue); break;
      default: return hipErrorInvalidValue;
    }
    if (kernel_events && off == 0) (void)hipEventRecord(kernel_events[0], stream);
    // conv2 draws its images from the second counter (zero since the memset above)
    conv2_mfma_kernel<<<(m < num_cus ? m : num_cus), C2_THREADS, 0, stream>>>(s.pool1, w.c2wt, w.c2w, w.c2b, s.flat, m, queue + 1);
    if (kernel_events && off == 0) (void)hipEventRecord(kernel_events[1], stream);
    switch (fc1_pick_nt(m)) {
      case 1: fc1_launch<1>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 2: fc1_launch<2>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 3: fc1_launch<3>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 4: fc1_launch<4>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 5: fc1_launch<5>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 6: fc1_launch<6>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      case 7: fc1_launch<7>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
      default: fc1_launch<8>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity, stream); break;
    }
    if (kernel_events && off == 0) (void)hipEventRecord(kernel_events[2], stream);
    }
    if (w.mode == GPD_LENET_SPLIT)
      fc2_score_kernel<true><<<(m + FC2_IMGS - 1) / FC2_IMGS, FC2_THREADS, 0, stream>>>(s.fc1p, w.f1b, nullptr, w.f2w, w.f2b, d_scores + off, m, s.capacity);
    else
      fc2_score_kernel<false><<<(m + FC2_IMGS - 1) / FC2_IMGS, FC2_THREADS, 0, stream>>>(nullptr, nullptr, s.fc1t, w.f2w, w.f2b, d_scores + off, m, s.capacity);
  }
  return hipGetLastError();
}

}  // namespace gpd
