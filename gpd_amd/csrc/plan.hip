// The host glue of GraspDetector::detectGrasps between its stages, as device tables.
//
// Replaces, for the fused entry points (gpd_hip_detect, gpd_hip_detect_batch) and for
// gpd_hip_images:
//   * filterGraspsWorkspace dropping hand sets without a valid hand (grasp_detector.cpp:238, 334-398;
//     the per-hand test itself runs at the end of hand_eval_kernel),
//   * the compaction of the frames (frame_estimator.cpp:24-29: samples without a 0.01 m neighbour are
//     dropped before the sets are numbered),
//   * createImageList's set-major, slot-minor gather of the valid hands (image_generator.cpp:91-98),
//   * the position of every hand set in the single LCG stream of the shadow draws
//     (hand_set.cpp:118-185, 263-266: every camera that sees a neighbourhood point consumes
//     N * num_shadow_points draws, in set order),
//   * the score write-back hands[i]->setScore(scores[i]) (grasp_detector.cpp:269-273),
//   * selectGrasps (grasp_detector.cpp:405-420) over the device score array.
//
// plan_kernel: the tables are prefix sums over the samples in order.  One workgroup per 256 samples; the workgroups
// number themselves by a ticket (arrival order), publish the sums of their samples and add up the published sums of the
// workgroups before them (decoupled look-back: nobody waits for a workgroup that has not started), so the table stores
// — one scattered cache line per lane, what a single workgroup spent 40 of its 43 us on — spread over the chip.
#include <cstddef>
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

constexpr int PLAN_THREADS = 256;

struct PlanParams {
  const int32_t *counts;   // [S][8]: N_hands, N_images, k_frames, found, mask of cameras that see the image neighbourhood
  const uint8_t *fvalid;   // [S][slots]
  const uint8_t *set_flags;   // non-null: use these instead, [num_sets_given][slots] in SET order (sets beyond: invalid)
  const double *set_samples;  // ... whose samples [num_sets_given][3] must equal the search's
  const double *frames;       // [S][12], sample first
  int num_sets_given;
  int S, slots, num_cams;
  int shadow;              // 15 channels: the shadow draws are consumed
  int num_shadow;          // draws per neighbourhood point and camera (hand_set.cpp:127: floor(shadow_length / 0.003))
  int32_t *sample_of_set, *hand_cand, *cand_hand, *cand_out, *cand_meta, *set_meta;
  PlanSummary *summary;
  PlanPart *parts;        // [workgroups] published sums (gpd_internal.h)
  unsigned *ticket;       // arrival counter: 0 at launch, put back to 0 by the workgroup that draws the last ticket
  unsigned epoch;         // stamp of this launch in PlanPart::ready_*
};

// exclusive scan of (a, b, c, d) over the workgroup; totals returned through the references
struct Scan4 {
  int a, b, c;
  unsigned long long d;
};
__device__ inline Scan4 block_scan4(Scan4 v, Scan4 &total, Scan4 *s_part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Scan4 incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int xa = __shfl_up(incl.a, o), xb = __shfl_up(incl.b, o), xc = __shfl_up(incl.c, o);
    const unsigned lo = __shfl_up((unsigned)incl.d, o), hi = __shfl_up((unsigned)(incl.d >> 32), o);
    if (lane >= o) {
      incl.a += xa;
      incl.b += xb;
      incl.c += xc;
      incl.d += ((unsigned long long)hi << 32) | lo;
    }
  }
  __syncthreads();
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  Scan4 base = {0, 0, 0, 0ull};
  total = base;
  for (int w = 0; w < PLAN_THREADS / 64; w++) {
    const Scan4 x = s_part[w];
    if (w < wave) {
      base.a += x.a;
      base.b += x.b;
      base.c += x.c;
      base.d += x.d;
    }
    total.a += x.a;
    total.b += x.b;
    total.c += x.c;
    total.d += x.d;
  }
  Scan4 excl = {base.a + incl.a - v.a, base.b + incl.b - v.b, base.c + incl.c - v.c, base.d + incl.d - v.d};
  return excl;
}

// valid flags of one row ([slots] bytes) as a bit mask.  Every load is requested before the first one is looked at (a
// loop over a run-time slot count made them dependent round trips: 8 x ~1.5 us per chunk of samples, most of the
// kernel's 43 us); rows of whole words — 8 orientations x 1..3 axes — come as aligned words.
__device__ inline unsigned valid_mask(const uint8_t *row, int slots) {
  unsigned m = 0;
  static_assert(GPD_MAX_SLOTS % 4 == 0 && GPD_MAX_SLOTS <= 32, "slot mask");
  if (!(slots & 3)) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(row);
    uint32_t x[GPD_MAX_SLOTS / 4];
#pragma unroll
    for (int q = 0; q < GPD_MAX_SLOTS / 4; q++) x[q] = 4 * q < slots ? w[q] : 0u;
#pragma unroll
    for (int q = 0; q < GPD_MAX_SLOTS / 4; q++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        if ((x[q] >> (8 * b)) & 0xffu) m |= 1u << (4 * q + b);
  } else {
    uint8_t x[GPD_MAX_SLOTS];
#pragma unroll
    for (int j = 0; j < GPD_MAX_SLOTS; j++) x[j] = j < slots ? row[j] : (uint8_t)0;
#pragma unroll
    for (int j = 0; j < GPD_MAX_SLOTS; j++)
      if (x[j]) m |= 1u << j;
  }
  return m;
}

// sum of v over the workgroup, in every thread (s_red: PLAN_THREADS / 64 slots; barriers inside)
__device__ inline long long block_sum_i64(long long v, long long *s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += (long long)(((unsigned long long)(unsigned)__shfl_xor((int)((unsigned long long)v >> 32), o) << 32) |
                     (unsigned)__shfl_xor((int)(unsigned)(unsigned long long)v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  long long t = 0;
#pragma unroll
  for (int w = 0; w < PLAN_THREADS / 64; w++) t += s_red[w];
  return t;
}
__device__ inline int block_max_i32(int v, long long *s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = (int)s_red[0];
#pragma unroll
  for (int w = 1; w < PLAN_THREADS / 64; w++) t = max(t, (int)s_red[w]);
  return t;
}
// a published word of another workgroup: spin until it carries this launch's stamp (the writer fences before it stamps)
__device__ inline void wait_stamp(const unsigned *flag, unsigned epoch) {
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(PLAN_THREADS) void plan_kernel(PlanParams P) {
  __shared__ Scan4 s_part[PLAN_THREADS / 64];
  __shared__ long long s_red[PLAN_THREADS / 64];
  __shared__ int s_mismatch, s_blk;
  const int tid = threadIdx.x;
  const unsigned long long tk0 = wall_clock64();
  if (tid == 0) {
    s_mismatch = 0x7fffffff;
    // the workgroups are numbered in the order they start.  The counter looks after itself: whoever draws the last
    // ticket of the launch knows that all are taken and puts it back to 0 for the next launch on this stream — no host
    // mirror that a failed / mis-reported launch could leave out of step
    const unsigned t = atomicAdd(P.ticket, 1u);
    if (t + 1u >= gridDim.x) __hip_atomic_store(P.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_blk = (int)t;
  }
  __syncthreads();
  const int blk = s_blk, nblk = (int)gridDim.x;
  if (blk >= nblk) return;  // cannot happen with a counter that starts at 0; never index parts[] beyond the launch
  const int s = blk * PLAN_THREADS + tid;
  const bool on = s < P.S;
  PlanPart *mine_part = P.parts + blk;
  const unsigned cam_mask = P.num_cams >= 32 ? 0xffffffffu : ((1u << P.num_cams) - 1u);

  // ---- inputs of this thread's sample
  int4 cn = make_int4(0, 0, 0, 0);  // N_hands, N_images, k_frames, found
  int cams = 0;
  if (on) {
    cn = *reinterpret_cast<const int4 *>(P.counts + 8 * (size_t)s);
    cams = P.counts[8 * (size_t)s + 4];
  }
  const int has_set = on && cn.z > 0, Ni = cn.y;
  const unsigned seen = (unsigned)cams & cam_mask;
  unsigned vmask = 0;  // valid slots (slots <= GPD_MAX_SLOTS = 24)
  bool mismatch = false;
  int sets_before = 0;  // (caller flags only, below)
  if (!P.set_flags) {
    if (has_set) vmask = valid_mask(P.fvalid + (size_t)s * P.slots, P.slots);
  } else {
    // the caller's flags are numbered by hand set: the set number of the sample first — its own scan and look-back
    Scan4 v0 = {has_set, 0, 0, 0ull}, t0;
    const int ex0 = block_scan4(v0, t0, s_part).a;
    if (tid == 0) {
      mine_part->sets = t0.a;
      __threadfence();
      __hip_atomic_store(&mine_part->ready_sets, P.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long before = 0;
    for (int b = tid; b < blk; b += PLAN_THREADS) {
      wait_stamp(&P.parts[b].ready_sets, P.epoch);
      before += P.parts[b].sets;
    }
    sets_before = (int)block_sum_i64(before, s_red);
    const int set_no = sets_before + ex0;
    if (has_set && set_no < P.num_sets_given) {
      vmask = valid_mask(P.set_flags + (size_t)set_no * P.slots, P.slots);
      double a[3], b[3];  // checked for the sets that contribute a candidate (the caller may pass husks for the others)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        a[r] = P.set_samples[3 * (size_t)set_no + r];
        b[r] = P.frames[12 * (size_t)s + r];
      }
      mismatch = vmask && !(a[0] == b[0] && a[1] == b[1] && a[2] == b[2]);
      if (mismatch) atomicMin(&s_mismatch, set_no);
    }
  }
  const int nv = __popc(vmask);
  int ncam = 0, n_bits = 0;
  if (nv && P.shadow && Ni > 0) {
    // HandSet::calculateShadow (hand_set.cpp:118-185): every camera that sees a neighbourhood point casts
    // Ni * num_shadow draws, in camera order.  One camera: its voxel set (empty if it sees nothing).  Several:
    // camera 0's set (empty if camera 0 sees nothing) intersected with the sets of the other seeing cameras —
    // the draws are consumed even when the result is discarded.
    ncam = __popc(seen);
    n_bits = (seen & 1u) ? ncam : 0;
  }
  const unsigned long long draws = (unsigned long long)Ni * (unsigned long long)P.num_shadow * (unsigned long long)ncam;

  // ---- this workgroup's sums, published before anything is waited for
  Scan4 v = {has_set, nv, n_bits, draws}, total;
  const Scan4 ex = block_scan4(v, total, s_part);
  const int g_worst = block_max_i32(on ? cn.w : 0, s_red);
  const long long g_live = block_sum_i64(nv ? 1 : 0, s_red);
  const long long g_set_ni = block_sum_i64(nv ? Ni : 0, s_red);
  const long long g_cand_ni = block_sum_i64((long long)(nv ? Ni : 0) * nv, s_red);  // (the barriers order s_mismatch too)
  if (tid == 0) {
    mine_part->sum[0] = total.a;
    mine_part->sum[1] = total.b;
    mine_part->sum[2] = total.c;
    mine_part->worst = g_worst;
    mine_part->draws = total.d;
    mine_part->live = (int)g_live;
    mine_part->mismatch = s_mismatch;
    mine_part->sum_set_ni = g_set_ni;
    mine_part->sum_cand_ni = g_cand_ni;
    __threadfence();
    __hip_atomic_store(&mine_part->ready, P.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned long long tk1 = wall_clock64();
  // ---- sums of the workgroups before this one (the last one also gathers the statistics of the summary)
  const bool last = blk == nblk - 1;
  long long pa = 0, pb = 0, pc = 0, pd = 0, p_live = 0, p_set_ni = 0, p_cand_ni = 0;
  int p_worst = 0, p_mis = 0x7fffffff;
  for (int b = tid; b < blk; b += PLAN_THREADS) {
    const PlanPart *q = P.parts + b;
    wait_stamp(&q->ready, P.epoch);
    pa += q->sum[0];
    pb += q->sum[1];
    pc += q->sum[2];
    pd += (long long)q->draws;
    if (last) {
      p_live += q->live;
      p_set_ni += q->sum_set_ni;
      p_cand_ni += q->sum_cand_ni;
      p_worst = max(p_worst, q->worst);
      p_mis = min(p_mis, q->mismatch);
    }
  }
  Scan4 carry;  // sets, candidates, shadow bitsets, LCG draws before this workgroup's samples
  if (blk > 0) {
    carry.a = (int)block_sum_i64(pa, s_red);
    carry.b = (int)block_sum_i64(pb, s_red);
    carry.c = (int)block_sum_i64(pc, s_red);
    carry.d = (unsigned long long)block_sum_i64(pd, s_red);
  } else {
    carry = Scan4{0, 0, 0, 0ull};
  }
  const unsigned long long tk2 = wall_clock64();

  // ---- the tables of this thread's sample
  if (on) {
    const int ns = carry.a + ex.a;
    if (has_set) P.sample_of_set[ns] = s;
    int cand = carry.b + ex.b;
    const int first_bits = n_bits ? carry.c + ex.c : -1;
    for (int j = 0; j < P.slots; j++) {
      int hc = -1;
      if (vmask >> j & 1u) {
        hc = cand++;
        P.cand_hand[hc] = s * P.slots + j;
        P.cand_out[hc] = ns * P.slots + j;
        *reinterpret_cast<int4 *>(P.cand_meta + 4 * (size_t)hc) = make_int4(s, Ni, first_bits, n_bits);
      }
      P.hand_cand[(size_t)s * P.slots + j] = hc;
    }
    if (ncam) {
      unsigned long long lcg = carry.d + ex.d;
      int row = carry.c + ex.c;
      for (int cam = 0; cam < P.num_cams; cam++) {
        if (!(seen >> cam & 1u)) continue;
        if (n_bits) {
          int4 *m = reinterpret_cast<int4 *>(P.set_meta + 8 * (size_t)row++);
          m[0] = make_int4(s, Ni, (int32_t)(uint32_t)(lcg & 0xffffffffull), (int32_t)(uint32_t)(lcg >> 32));
          m[1] = make_int4(cam, 0, 0, 0);
        }
        lcg += (unsigned long long)Ni * (unsigned long long)P.num_shadow;
      }
    }
  }
  if (!last) return;
  // ---- the summary, by the workgroup with the last ticket
  const long long t_live = block_sum_i64(p_live, s_red) + g_live;
  const long long t_set_ni = block_sum_i64(p_set_ni, s_red) + g_set_ni;
  const long long t_cand_ni = block_sum_i64(p_cand_ni, s_red) + g_cand_ni;
  const int t_worst = max(block_max_i32(p_worst, s_red), g_worst);
  const int t_mis = min(-block_max_i32(-p_mis, s_red), s_mismatch);
  if (tid == 0) {
    PlanSummary sm;
    sm.num_sets = carry.a + total.a;
    sm.num_candidates = carry.b + total.b;
    sm.num_shadow_sets = carry.c + total.c;
    sm.total_draws = carry.d + total.d;
    sm.worst_found = t_worst;
    sm.live_sets = (int)t_live;
    sm.mismatch_set = t_mis != 0x7fffffff ? t_mis : -1;
    if (P.set_flags && P.num_sets_given > sm.num_sets && sm.mismatch_set < 0) sm.mismatch_set = sm.num_sets;  // more sets than the search has
    // profiling aid (GPD_PLAN_TIMING=1 prints them): phase times of the last workgroup's thread 0 in 10 ns ticks, 16 bits each
    const unsigned long long tk3 = wall_clock64();
    auto tick = [](unsigned long long a, unsigned long long b) { return (int32_t)((b - a) > 0xffffull ? 0xffffull : (b - a)); };
    sm.pad_[0] = tick(tk0, tk1) | (tick(tk1, tk2) << 16);
    sm.pad_[1] = tick(tk2, tk3);
    sm.sum_set_ni = t_set_ni;
    sm.sum_cand_ni = t_cand_ni;
    *P.summary = sm;
  }
}

void plan_free(Plan &pl) {
  void *ptrs[] = {pl.d_sample_of_set, pl.d_hand_cand, pl.d_cand_hand, pl.d_cand_out, pl.d_cand_meta, pl.d_set_meta, pl.d_summary,
                  pl.d_set_flags, pl.d_set_samples, pl.d_parts, pl.d_ticket};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (pl.h_summary) (void)hipHostFree(pl.h_summary);
  pl = Plan();
}

// Tables for up to capS samples x slots hands and `cams` cameras (grow only; a growth stalls the device).
int plan_reserve(Plan &pl, int want_samples, int slots, int cams, hipStream_t stream) {
  if (want_samples <= pl.cap_samples && slots <= pl.cap_slots && cams <= pl.cap_cams) return GPD_OK;
  note_alloc(__func__);
  const int capS = want_samples > pl.cap_samples ? want_samples : pl.cap_samples;
  const int capC = cams > pl.cap_cams ? cams : pl.cap_cams;
  const int capL = slots > pl.cap_slots ? slots : pl.cap_slots;
  plan_free(pl);
  const size_t H = (size_t)capS * capL;
  HIP_RET(hipMalloc(&pl.d_sample_of_set, (size_t)capS * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_hand_cand, H * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_cand_hand, H * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_cand_out, H * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_cand_meta, H * 4 * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_set_meta, (size_t)capS * capC * 8 * sizeof(int32_t)));
  HIP_RET(hipMalloc(&pl.d_summary, sizeof(PlanSummary)));
  // look-back state of plan_kernel: zeroed once (stamps are compared with a launch number that starts at 1)
  const size_t nparts = (size_t)(capS + PLAN_THREADS - 1) / PLAN_THREADS + 1;
  HIP_RET(hipMalloc(&pl.d_parts, nparts * sizeof(PlanPart)));
  HIP_RET(hipMalloc(&pl.d_ticket, sizeof(unsigned)));
  HIP_RET(hipMemsetAsync(pl.d_parts, 0, nparts * sizeof(PlanPart), stream));
  HIP_RET(hipMemsetAsync(pl.d_ticket, 0, sizeof(unsigned), stream));
  pl.epoch = 0;
  HIP_RET(hipHostMalloc(reinterpret_cast<void **>(&pl.h_summary), sizeof(PlanSummary), 0));
  pl.cap_samples = capS;
  pl.cap_slots = capL;
  pl.cap_cams = capC;
  return GPD_OK;
}

int plan_build(const gpd_params &p, const Cloud &c, const SearchState &s, Plan &pl, hipStream_t stream, const uint8_t *set_flags,
               const double *set_samples, int num_sets_given) {
  const int slots = p.num_hand_axes * p.num_orientations;
  const int S = s.num_samples;
  {
    const int rc = plan_reserve(pl, s.capacity_samples, slots, c.num_cams, stream);
    if (rc) return rc;
  }
  PlanParams pp;
  pp.set_flags = nullptr;
  pp.set_samples = nullptr;
  pp.num_sets_given = 0;
  if (set_flags) {
    if (num_sets_given > pl.cap_set_flags || !pl.d_set_flags) {
      if (pl.d_set_flags) (void)hipFree(pl.d_set_flags);
      if (pl.d_set_samples) (void)hipFree(pl.d_set_samples);
      pl.d_set_flags = nullptr;
      pl.d_set_samples = nullptr;
      pl.cap_set_flags = 0;
      const int cap = num_sets_given > pl.cap_samples ? num_sets_given : pl.cap_samples;
      HIP_RET(hipMalloc(&pl.d_set_flags, (size_t)cap * pl.cap_slots + 16));
      HIP_RET(hipMalloc(&pl.d_set_samples, ((size_t)cap * 3 + 1) * sizeof(double)));
      pl.cap_set_flags = cap;
    }
    if (num_sets_given > 0) {
      HIP_RET(hipMemcpyAsync(pl.d_set_flags, set_flags, (size_t)num_sets_given * slots, hipMemcpyHostToDevice, stream));
      HIP_RET(hipMemcpyAsync(pl.d_set_samples, set_samples, (size_t)num_sets_given * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    pp.set_flags = pl.d_set_flags;
    pp.set_samples = pl.d_set_samples;
    pp.num_sets_given = num_sets_given;
  }
  pp.frames = s.d_frames;
  pp.counts = s.d_counts;
  pp.fvalid = s.d_fvalid;
  pp.S = S;
  pp.slots = slots;
  pp.num_cams = c.num_cams;
  pp.shadow = p.image_num_channels == 15 ? 1 : 0;
  // shadow_length_ = max(volume_depth, volume_height/2, volume_width) (image_15_channels_strategy.h:70-75);
  // num_shadow_points = floor(shadow_length / voxel_grid_size), voxel_grid_size = 0.003 (hand_set.cpp:125-127)
  const double shadow_length = std::fmax(std::fmax(p.volume_depth, p.volume_height / 2.0), p.volume_width);
  pp.num_shadow = (int)std::floor(shadow_length / 0.003);
  pp.sample_of_set = pl.d_sample_of_set;
  pp.hand_cand = pl.d_hand_cand;
  pp.cand_hand = pl.d_cand_hand;
  pp.cand_out = pl.d_cand_out;
  pp.cand_meta = pl.d_cand_meta;
  pp.set_meta = pl.d_set_meta;
  pp.summary = pl.d_summary;
  const int workgroups = S > 0 ? (S + PLAN_THREADS - 1) / PLAN_THREADS : 1;
  if (++pl.epoch == 0u) {
    // the launch number has wrapped (2^32 launches): a stamp of the first lap could pass for this one — clear them
    const size_t nparts = (size_t)(pl.cap_samples + PLAN_THREADS - 1) / PLAN_THREADS + 1;
    HIP_RET(hipMemsetAsync(pl.d_parts, 0, nparts * sizeof(PlanPart), stream));
    pl.epoch = 1u;
  }
  pp.parts = pl.d_parts;
  pp.ticket = pl.d_ticket;
  pp.epoch = pl.epoch;
  plan_kernel<<<workgroups, PLAN_THREADS, 0, stream>>>(pp);
  HIP_RET(hipGetLastError());
  HIP_RET(hipMemcpyAsync(pl.h_summary, pl.d_summary, sizeof(PlanSummary), hipMemcpyDeviceToHost, stream));
  return GPD_OK;
}

// ---------------------------------------------------------------------------
// Hand records for the caller
// ---------------------------------------------------------------------------
struct EmitParams {
  const gpd_hand *hands;  // [S][slots] as the search wrote them
  const uint8_t *fvalid;
  const int32_t *sample_of_set, *hand_cand, *cand_hand, *cand_out;
  const float *scores;    // per candidate (may be null: scores stay 0)
  const int32_t *sel;     // gather mode: candidate ordinals
  gpd_hand *out;
  int n, slots;
};
// 11 x 16 bytes per record, one lane per 16-byte piece: coalesced in and out
__device__ inline void copy_hand(const gpd_hand *src, gpd_hand *dst, int set_index, int valid, float score, int piece) {
  uint4 v = reinterpret_cast<const uint4 *>(src)[piece];
  if (piece == 9) {         // bytes 144..159: grasp_width (8), score (4), finger_placement_index (4)
    v.z = __float_as_uint(score);
  } else if (piece == 10) { // bytes 160..175: set_index, slot, valid | half | full | pad, pad
    v.x = (uint32_t)set_index;
    v.z = (v.z & 0xffffff00u) | (uint32_t)(valid & 0xff);
  }
  reinterpret_cast<uint4 *>(dst)[piece] = v;
}
static_assert(sizeof(gpd_hand) == 176, "copy_hand assumes the 176-byte record");
static_assert(offsetof(gpd_hand, score) == 152 && offsetof(gpd_hand, set_index) == 160 && offsetof(gpd_hand, valid) == 168,
              "copy_hand field offsets");

// all sets: out[ns][slot]
__global__ __launch_bounds__(256) void emit_sets_kernel(EmitParams P) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int rec = g >> 4, piece = g & 15;
  if (rec >= P.n || piece >= 11) return;
  const int ns = rec / P.slots, j = rec - ns * P.slots;
  const int h = P.sample_of_set[ns] * P.slots + j;
  const int c = P.hand_cand[h];
  const float score = (c >= 0 && P.scores) ? P.scores[c] : P.hands[h].score;
  copy_hand(P.hands + h, P.out + rec, ns, P.fvalid[h], score, piece);
}
// candidates only (sel == nullptr: all, in candidate order; else the listed ones)
__global__ __launch_bounds__(256) void emit_cands_kernel(EmitParams P) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int rec = g >> 4, piece = g & 15;
  if (rec >= P.n || piece >= 11) return;
  const int c = P.sel ? P.sel[rec] : rec;
  const int h = P.cand_hand[c];
  copy_hand(P.hands + h, P.out + rec, P.cand_out[c] / P.slots, 1, P.scores ? P.scores[c] : 0.f, piece);
}

int plan_emit_hands(const gpd_params &p, const SearchState &s, const Plan &pl, const float *d_scores, gpd_hand *d_out,
                    bool candidates_only, hipStream_t stream) {
  EmitParams ep;
  ep.hands = s.d_hands;
  ep.fvalid = s.d_fvalid;
  ep.sample_of_set = pl.d_sample_of_set;
  ep.hand_cand = pl.d_hand_cand;
  ep.cand_hand = pl.d_cand_hand;
  ep.cand_out = pl.d_cand_out;
  ep.scores = d_scores;
  ep.sel = nullptr;
  ep.out = d_out;
  ep.slots = p.num_hand_axes * p.num_orientations;
  ep.n = candidates_only ? pl.h_summary->num_candidates : pl.h_summary->num_sets * ep.slots;
  if (ep.n <= 0) return GPD_OK;
  const unsigned grid = (unsigned)(((size_t)ep.n * 16 + 255) / 256);
  if (candidates_only)
    emit_cands_kernel<<<grid, 256, 0, stream>>>(ep);
  else
    emit_sets_kernel<<<grid, 256, 0, stream>>>(ep);
  HIP_RET(hipGetLastError());
  return GPD_OK;
}

int gather_hands(const gpd_params &p, const SearchState &s, const Plan &pl, const float *d_scores, const int32_t *d_sel, int k,
                 gpd_hand *d_out, hipStream_t stream) {
  if (k <= 0) return GPD_OK;
  EmitParams ep;
  ep.hands = s.d_hands;
  ep.fvalid = s.d_fvalid;
  ep.sample_of_set = pl.d_sample_of_set;
  ep.hand_cand = pl.d_hand_cand;
  ep.cand_hand = pl.d_cand_hand;
  ep.cand_out = pl.d_cand_out;
  ep.scores = d_scores;
  ep.sel = d_sel;
  ep.out = d_out;
  ep.slots = p.num_hand_axes * p.num_orientations;
  ep.n = k;
  emit_cands_kernel<<<(unsigned)(((size_t)k * 16 + 255) / 256), 256, 0, stream>>>(ep);
  HIP_RET(hipGetLastError());
  return GPD_OK;
}

// the selected records out of a flat candidate list that was emitted earlier (the fused entries with num_selected > 0
// keep such a list per lane, so that a selection can be redone after the lane's search / plan buffers have moved on to
// the next cloud of a batch)
__global__ __launch_bounds__(256) void gather_records_kernel(const gpd_hand *__restrict__ all, const int32_t *__restrict__ sel, int k,
                                                             gpd_hand *__restrict__ out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int rec = g / 11, piece = g % 11;  // 176 bytes = 11 x 16
  if (rec >= k) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(all + sel[rec]);
  reinterpret_cast<uint4 *>(out + rec)[piece] = src[piece];
}

int gather_records(const gpd_hand *d_all, const int32_t *d_sel, int k, gpd_hand *d_out, hipStream_t stream) {
  if (k <= 0) return GPD_OK;
  static_assert(sizeof(gpd_hand) == 176, "gather_records_kernel copies 11 x 16 bytes per record");
  gather_records_kernel<<<(unsigned)(((size_t)k * 11 + 255) / 256), 256, 0, stream>>>(d_all, d_sel, k, d_out);
  HIP_RET(hipGetLastError());
  return GPD_OK;
}

// ---------------------------------------------------------------------------
// selectGrasps (grasp_detector.cpp:405-420): std::partial_sort of the hands by score, descending, first
// num_selected kept.  One workgroup: radix select of the k-th largest score (four 8-bit passes over the
// order-preserving integer image of the floats), then the k winners are sorted in LDS by
// (score descending, candidate ordinal ascending).  Equal scores make the reference's result depend on
// libstdc++'s heap-select history; they are reported through *tie and the caller reruns the selection with
// std::partial_sort itself on the downloaded scores (4 bytes per candidate) — the records still stay on
// the device until the winners are gathered.
// ---------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAX_K = 8192;  // keys sorted in LDS (64 KB)

__device__ inline uint32_t score_key(float f) {  // larger float <-> larger key; -0 is mapped onto +0 (they compare equal under
                                                 // isScoreGreater, so they must share a key for the tie test at the cut)
  const uint32_t u = __float_as_uint(f + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(SEL_THREADS) void select_topk_kernel(const float *__restrict__ scores, int n, int k, int32_t *sel,
                                                                  int32_t *tie) {
  __shared__ unsigned long long s_keys[SEL_MAX_K];
  __shared__ int s_hist[256];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining, s_count_gt, s_count_eq, s_tie;
  __shared__ int s_wcount[SEL_THREADS / 64];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_prefix = 0u;
    s_remaining = k;
    s_count_gt = 0;
    s_count_eq = 0;
    s_tie = 0;
  }
  __syncthreads();
  // the k-th largest key, most significant byte first
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < n; i += SEL_THREADS) {
      const uint32_t key = score_key(scores[i]);
      if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 0xffu], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining;
      int b = 255;
      for (; b > 0; b--) {
        if (s_hist[b] >= rem) break;
        rem -= s_hist[b];
      }
      s_prefix = prefix | ((uint32_t)b << shift);
      s_remaining = rem;
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;  // key of the k-th largest score
  const int need_eq = s_remaining;  // how many entries equal to it belong to the selection
  // entries above the threshold, then the first need_eq equal ones by candidate ordinal
  for (int base = 0; base < n; base += SEL_THREADS) {
    const int i = base + tid;
    const uint32_t key = i < n ? score_key(scores[i]) : 0u;
    const bool gt = i < n && key > kth;
    if (gt) {
      const int pos = atomicAdd(&s_count_gt, 1);
      s_keys[pos] = ((unsigned long long)(~key) << 32) | (unsigned)i;  // ascending sort of ~key = descending score
    }
  }
  __syncthreads();
  const int n_gt = s_count_gt;
  // equal entries in ordinal order: chunked ballot scan
  for (int base = 0; base < n; base += SEL_THREADS) {
    const int i = base + tid;
    const bool eq = i < n && score_key(scores[i]) == kth;
    const unsigned long long ballot = __ballot(eq);
    if ((tid & 63) == 0) s_wcount[tid >> 6] = __popcll(ballot);
    __syncthreads();
    int off = s_count_eq;
    for (int w = 0; w < (tid >> 6); w++) off += s_wcount[w];
    if (eq) {
      const int r = off + __popcll(ballot & ((1ull << (tid & 63)) - 1ull));
      if (r < need_eq) s_keys[n_gt + r] = ((unsigned long long)(~kth) << 32) | (unsigned)i;
    }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < SEL_THREADS / 64; w++) t += s_wcount[w];
      s_count_eq += t;
    }
    __syncthreads();
  }
  if (tid == 0 && s_count_eq > need_eq) s_tie = 1;  // a tie at the cut
  // bitonic sort of the k keys
  int m = 1;
  while (m < k) m <<= 1;
  for (int i = k + tid; i < m; i += SEL_THREADS) s_keys[i] = ~0ull;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (m >> 1); t += SEL_THREADS) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & size) == 0;
        const unsigned long long a = s_keys[lo], b = s_keys[hi];
        if ((a > b) == up) {
          s_keys[lo] = b;
          s_keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += SEL_THREADS) {
    const int idx = (int)(unsigned)(s_keys[i] & 0xffffffffull);
    sel[i] = idx;
    // equal scores next to each other (float comparison: -0 == +0)
    if (i + 1 < k && scores[idx] == scores[(int)(unsigned)(s_keys[i + 1] & 0xffffffffull)]) s_tie = 1;
  }
  __syncthreads();
  if (tid == 0) *tie = s_tie;
}

int select_topk(const float *d_scores, int n, int k, int32_t *d_sel, int32_t *d_tie, hipStream_t stream) {
  if (k <= 0 || n <= 0) return GPD_OK;
  if (k > n) k = n;
  if (k > SEL_MAX_K) {  // the callers route such selections through std::partial_sort on the host (select_topk_capacity)
    set_error("select_topk: k = %d exceeds the device selection capacity %d", k, SEL_MAX_K);
    return GPD_ERR_CAPACITY;
  }
  static_assert(sizeof(unsigned long long) * SEL_MAX_K == 64 * 1024, "s_keys alone is 64 KB of LDS: a gfx950-class (160 KB) workgroup");
  select_topk_kernel<<<1, SEL_THREADS, 0, stream>>>(d_scores, n, k, d_sel, d_tie);
  HIP_RET(hipGetLastError());
  return GPD_OK;
}

int select_topk_capacity() { return SEL_MAX_K; }

}  // namespace gpd
