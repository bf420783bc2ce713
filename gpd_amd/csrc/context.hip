// C-ABI of libgpd_hip.so (include/gpd_hip.h): context, uploads, stage launches.
//
// A context owns two LANES — each a HIP stream with its own cloud, search buffers, candidate plan,
// image buffers and LeNet scratch.  Every single-cloud entry point runs on lane 0.
// gpd_hip_detect_batch alternates the lanes: while the image + LeNet kernels of cloud i run on one
// lane, the upload + grid + search of cloud i+1 is already enqueued on the other, so host hops and
// the host-device copies of one cloud hide behind the kernels of its neighbour, and the
// tail of one cloud's kernel is filled by the other's.  (Measured against ONE stream carrying
// search(i+1) ahead of images+LeNet(i), i.e. the same pipelining with strictly sequential kernels: two streams
// 796 k candidates/s, one stream 735 k, same box, same 48 clouds.)
//
// A fused detect is three steps per cloud:
//   begin   enqueue sample upload, neighbourhood / centre / hand_eval kernels (incl. the workspace filter)
//           and plan_kernel (candidate list, shadow LCG offsets) + the 48-byte summary copy — no waiting
//   middle  wait for the summary (the only mid-pipeline wait: the launch sizes), enqueue image kernels,
//           LeNet, the record gather (all sets / candidates / the num_selected best) and ONE device-to-host
//           copy into pinned memory
//   end     wait, hand the records to the caller
// Between the stages nothing crosses PCIe but that summary.
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "gpd_internal.h"

namespace gpd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local int g_allocs = 0;
void note_alloc(const char *where) {
  g_allocs++;
  if (prof_env("GPD_ALLOC_TRACE")) fprintf(stderr, "[alloc] %s\n", where);  // (profiling build only)
}

}  // namespace gpd

using namespace gpd;

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GPD_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

namespace {

constexpr int kLanes = 2;

struct HostFlags {  // pinned; written by the last copies of a job
  int32_t status;   // capacity flags of the image kernels
  int32_t tie;      // select_topk: equal scores among the winners or at the cut
  int32_t lenet;    // != 0: a conv1 launch of this job gave up on its slot protocol (lenet.hip)
  int32_t pad_;
};

struct Lane {
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // search start / end, images end, LeNet end, images start
  hipEvent_t ev_plan = nullptr, ev_done = nullptr;  // the plan summary / the results of the job in flight are on the host
  hipEvent_t ev_chunk[4] = {nullptr, nullptr, nullptr, nullptr};  // a large record list leaves in four copies: the host copies one on while the next travels
  float stage_ms[3] = {0.f, 0.f, 0.f};
  Cloud cloud;
  PreState pre;      // raw scans of gpd_hip_detect_batch: workspace cut + voxeliser of the cloud this lane works on
  SearchState search;
  Plan plan;
  ImageState images;
  LeNetScratch lenet_scratch;
  float *d_scores = nullptr;
  int d_scores_cap = 0;
  gpd_hand *d_out = nullptr;  // hand records gathered for the caller
  size_t d_out_cap = 0;       // records
  char *h_out = nullptr;      // pinned: records, then scores
  size_t h_out_bytes = 0;
  HostFlags *h_flags = nullptr;  // pinned
  int32_t *d_sel = nullptr;      // [SEL capacity] candidate ordinals of the selection, then the tie flag
  int d_sel_cap = 0;
  gpd_hand *d_all = nullptr;     // selections (num_selected > 0): every candidate record of the job, scored — what a selection
  size_t d_all_cap = 0;          // gathers from, also after the lane's search / plan buffers belong to the next cloud
  // staging for gpd_hip_score with host images
  uint8_t *d_img_in = nullptr;      // HWC images handed to gpd_hip_score
  uint8_t *d_img_planar = nullptr;  // their planar copy
  size_t d_img_in_bytes = 0;
};

// one fused detect in flight on a lane
struct Job {
  const int32_t *sample_idx = nullptr;
  const double *sample_xyz = nullptr;
  int S = 0;
  int mode = 0;          // 0: all hand sets [num_sets][slots]; 1: candidates only (num_selected > 0: the best ones)
  int num_selected = 0;
  gpd_hand *hands = nullptr;
  long long capacity = 0;  // records `hands` can take
  int num_sets = 0, num_candidates = 0, num_hands = 0;
  bool live = false;       // device work enqueued, end() still has to collect it
  int out_records = 0;
  double t_plan_ms = 0.0;  // host clock when the plan summary had arrived (job_middle past its wait)
  double copy_ms = 0.0;    // job_end: handing the records over (after the wait)
  int chunks = 0;          // > 0: the records leave the device in this many copies, an event behind each
  unsigned long long lcg_base = 0;   // in: shadow draws of the cloud's sample ranges before this one (gpd_hip_detect_sharded)
  unsigned long long lcg_draws = 0;  // out: shadow draws of this job's hand sets
};

}  // namespace

struct gpd_hip_ctx {
  int device = 0;
  bool in_batch = false;  // gpd_hip_detect_batch is driving the lanes
  gpd_params params;
  LeNetWeights lenet;
  Lane lane[kLanes];
  PreState pre;
  ClusterState cluster;
  std::vector<hipEvent_t> replay_events;  // 6 per gpd_hip_replay call: start, images done, conv1, conv2, fc1, end
  float replay_kernel_ms[4] = {0, 0, 0, 0};  // conv1, conv2, fc1, fc2 sums of the replays of the last gpd_hip_replay_times
  size_t replay_used = 0;
  // GPD_REPLAY_PIPE=1 (experiment): the image stage of replay k + 1 beside the LeNet stage of replay k —
  // images on lane 0's stream into one of two image buffers, LeNet on `pipe_stream` behind the buffer's event
  hipStream_t pipe_stream = nullptr;
  uint8_t *pipe_images[2] = {nullptr, nullptr};  // [0] is lane 0's own buffer while the mode is on
  size_t pipe_bytes = 0;
  hipEvent_t pipe_filled[2] = {nullptr, nullptr}, pipe_read[2] = {nullptr, nullptr};
  bool pipe_read_valid[2] = {false, false};
  unsigned pipe_k = 0;
};

static int lane_init(Lane &L, hipStream_t shared = nullptr) {
  if (L.stream) return GPD_OK;
  if (shared) {
    L.stream = shared;
  } else {
    HIP_TRY(hipStreamCreate(&L.stream));
    L.owns_stream = true;
  }
  for (auto &e : L.ev) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipEventCreate(&L.ev_plan));
  HIP_TRY(hipEventCreate(&L.ev_done));
  for (auto &e : L.ev_chunk) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&L.h_flags), sizeof(HostFlags), 0));
  std::memset(L.h_flags, 0, sizeof(HostFlags));
  return GPD_OK;
}

static void lane_free(Lane &L) {
  if (L.stream) (void)hipStreamSynchronize(L.stream);
  lenet_scratch_free(L.lenet_scratch);
  preprocess_free(L.pre);
  for (auto &e : L.pre.ev)
    if (e) (void)hipEventDestroy(e);
  if (L.pre.ev_keys) (void)hipEventDestroy(L.pre.ev_keys);
  cloud_free(L.cloud);
  search_free(L.search);
  plan_free(L.plan);
  images_free(L.images);
  void *dev[] = {L.d_scores, L.d_out, L.d_sel, L.d_all, L.d_img_in, L.d_img_planar};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  if (L.h_out) (void)hipHostFree(L.h_out);
  if (L.h_flags) (void)hipHostFree(L.h_flags);
  for (auto &e : L.ev)
    if (e) (void)hipEventDestroy(e);
  if (L.ev_plan) (void)hipEventDestroy(L.ev_plan);
  if (L.ev_done) (void)hipEventDestroy(L.ev_done);
  for (auto &e : L.ev_chunk)
    if (e) (void)hipEventDestroy(e);
  if (L.stream && L.owns_stream) (void)hipStreamDestroy(L.stream);
  L = Lane();
}

static int reserve_scores(Lane &L, int n) {
  if (n <= L.d_scores_cap) return GPD_OK;
  note_alloc(__func__);
  if (L.d_scores) (void)hipFree(L.d_scores);
  L.d_scores = nullptr;
  L.d_scores_cap = 0;
  const int cap = n + n / 4;
  HIP_TRY(hipMalloc(&L.d_scores, (size_t)cap * sizeof(float)));
  L.d_scores_cap = cap;
  return GPD_OK;
}

static int reserve_out(Lane &L, size_t records, size_t extra_bytes) {
  if (records > L.d_out_cap) {
    note_alloc(__func__);
    if (L.d_out) (void)hipFree(L.d_out);
    L.d_out = nullptr;
    L.d_out_cap = 0;
    const size_t cap = records + records / 4;
    HIP_TRY(hipMalloc(&L.d_out, cap * sizeof(gpd_hand)));
    L.d_out_cap = cap;
  }
  const size_t bytes = L.d_out_cap * sizeof(gpd_hand) + extra_bytes;
  if (bytes > L.h_out_bytes) {
    note_alloc(__func__);
    if (L.h_out) (void)hipHostFree(L.h_out);
    L.h_out = nullptr;
    L.h_out_bytes = 0;
    const size_t cap = bytes + bytes / 8;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&L.h_out), cap, 0));
    L.h_out_bytes = cap;
  }
  return GPD_OK;
}

// selections (num_selected > 0): the winners' ordinals + tie flag, and the job's own list of every scored candidate
static int reserve_selection(Lane &L, int k, int n) {
  if (k + 1 > L.d_sel_cap) {
    note_alloc(__func__);
    if (L.d_sel) (void)hipFree(L.d_sel);
    L.d_sel = nullptr;
    L.d_sel_cap = 0;
    const int cap = k + 1 + k / 4;
    HIP_TRY(hipMalloc(&L.d_sel, (size_t)cap * sizeof(int32_t)));
    L.d_sel_cap = cap;
  }
  if ((size_t)n > L.d_all_cap) {
    note_alloc(__func__);
    if (L.d_all) (void)hipFree(L.d_all);
    L.d_all = nullptr;
    L.d_all_cap = 0;
    const size_t cap = (size_t)n + (size_t)n / 4;
    HIP_TRY(hipMalloc(&L.d_all, cap * sizeof(gpd_hand)));
    L.d_all_cap = cap;
  }
  return GPD_OK;
}

// Every buffer of a lane for clouds of up to `points` points / `cams` cameras, `samples` samples and `candidates`
// scored hands (selections of up to `selected` winners; -1: none), so that no call within those sizes allocates:
// growing a buffer is hipFree + hipMalloc, which waits for the whole device — in a batch that is a hole in BOTH
// lanes' queues.  gpd_hip_reserve and gpd_hip_detect_batch call this ahead of the first cloud.
static int lane_reserve(gpd_hip_ctx *ctx, Lane &L, int points, int cams, int samples, int candidates, int selected);

constexpr int kLeNetChunk = 65536;                // lenet_forward's images per pass (lenet.hip)
constexpr size_t kReserveBudget = 16ull << 30;   // candidate-sized buffers of a lane when the caller names no bound

// the most candidates `samples` samples can give (every slot a valid hand), cut to what `budget` bytes hold: per candidate
// its image, the LeNet scratch of both scoring modes (pool1, the f32 and the three-plane bf16 flatten, ip1 and its four K
// quarters), score and records; per (hand set, camera) a shadow voxel bitset of the 86^3-bit default window (wider image
// volumes take more: images_reserve grows them on demand)
static int candidate_bound(const gpd_params &p, int samples, int cams = 1, size_t budget = kReserveBudget) {
  const long long upper = (long long)samples * p.num_hand_axes * p.num_orientations;
  const size_t per = (size_t)kPix * p.image_num_channels + (20 * 784 + kFc1In + 5 * kFc1Out + 1) * sizeof(float) + 3 * (size_t)kLenetXld * 2 +
                     2 * sizeof(gpd_hand);
  const size_t bitsets = p.image_num_channels == 15 ? (size_t)std::max(samples, 1) * std::max(cams, 1) * ((86 * 86 * 86 + 31) / 32 * 4) : 0;
  const long long fit = budget > bitsets ? (long long)((budget - bitsets) / per) : 0;
  return (int)std::max(1ll, std::min(upper, fit));
}

static int lane_reserve(gpd_hip_ctx *ctx, Lane &L, int points, int cams, int samples, int candidates, int selected) {
  const gpd_params &p = ctx->params;
  const int slots = p.num_hand_axes * p.num_orientations;
  int rc = cloud_reserve(L.cloud, points, cams);
  if (!rc) rc = cloud_reserve_grid(L.cloud, 1 << 20);  // 2 cm cells of a scene up to ~8 m^3 (8 MB); a larger one grows the tables
  if (!rc && samples > 0) rc = search_reserve_samples(L.search, samples, slots);
  if (!rc && samples > 0) rc = plan_reserve(L.plan, L.search.capacity_samples, slots, cams, L.stream);
  if (rc || candidates <= 0) return rc;
  const long long sets = std::min((long long)samples, (long long)candidates) * cams;  // (live hand set, camera) voxel bitsets
  rc = images_reserve(p, L.images, candidates, p.image_num_channels == 15 ? (int)sets : 0);
  if (rc) return rc;
  HIP_TRY(lenet_scratch_reserve(L.lenet_scratch, std::min(candidates, kLeNetChunk)));
  rc = reserve_scores(L, candidates);
  if (rc) return rc;
  if (selected >= 0) {
    const int k = selected > 0 ? std::min(selected, candidates) : candidates;
    rc = reserve_out(L, (size_t)k, selected > 0 ? (size_t)candidates * sizeof(float) : 0);
    if (!rc && selected > 0) rc = reserve_selection(L, k, candidates);
  } else {
    rc = reserve_out(L, (size_t)samples * slots, 0);  // gpd_hip_detect: every slot of every set
  }
  return rc;
}

// ---- the three steps of a fused detect -------------------------------------------------------
static int job_begin(gpd_hip_ctx *ctx, Lane &L, Job &J) {
  J.live = false;
  J.num_sets = J.num_candidates = J.num_hands = 0;
  if (J.S == 0) return GPD_OK;
  HIP_TRY(hipEventRecord(L.ev[0], L.stream));
  int rc;
  {
    StageRange r("gpd:search (neighbourhoods, frames, hand evaluation, workspace filter)");
    rc = search_run(ctx->params, L.cloud, L.search, J.sample_idx, J.sample_xyz, J.S, L.stream, /*sync_counts=*/false);
  }
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev[1], L.stream));
  {
    StageRange r("gpd:plan (hand sets, candidate list, shadow LCG offsets)");
    rc = plan_build(ctx->params, L.cloud, L.search, L.plan, L.stream);
  }
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev_plan, L.stream));
  J.live = true;
  return GPD_OK;
}

// the middle step in two halves: wait for the plan summary (the only mid-pipeline wait; a list-capacity retry happens here), then
// enqueue images + LeNet + gather.  gpd_hip_detect_sharded puts the host-side scan of the shards' draw totals between the two.
static int job_wait_plan(gpd_hip_ctx *ctx, Lane &L, Job &J) {
  if (!J.live) return GPD_OK;
  J.live = false;  // set again once everything is enqueued
  HIP_TRY(hipEventSynchronize(L.ev_plan));  // not the stream: in a batch the next cloud's search is already queued behind
  J.t_plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (L.plan.h_summary->worst_found > L.search.nn_cap) {
    // a neighbourhood overflowed the list capacity of the search kernel: once more with the large lists
    const int cap = search_next_capacity(L.search, L.plan.h_summary->worst_found);
    if (!cap) {
      set_error("search: a neighbourhood holds %d points, more than the list capacity %d", L.plan.h_summary->worst_found, kNnCapMax);
      return GPD_ERR_CAPACITY;
    }
    int rc = search_force_capacity(L.search, cap);
    if (rc) return rc;
    rc = job_begin(ctx, L, J);
    if (rc) return rc;
    J.live = false;
    HIP_TRY(hipStreamSynchronize(L.stream));
  }
  J.lcg_draws = L.plan.h_summary->total_draws;
  J.live = true;
  return GPD_OK;
}

static int job_enqueue(gpd_hip_ctx *ctx, Lane &L, Job &J) {
  if (!J.live) return GPD_OK;
  J.live = false;
  const PlanSummary sm = *L.plan.h_summary;
  static const bool plan_timing = prof_env("GPD_PLAN_TIMING") != nullptr;
  if (plan_timing)
    fprintf(stderr, "[plan-timing] own sums %.2f us, look-back %.2f, tables + summary %.2f (last workgroup's thread 0, 100 MHz clock)\n",
            (sm.pad_[0] & 0xffff) * 0.01, ((unsigned)sm.pad_[0] >> 16) * 0.01, (sm.pad_[1] & 0xffff) * 0.01);
  const int slots = ctx->params.num_hand_axes * ctx->params.num_orientations;
  J.num_sets = sm.num_sets;
  J.num_candidates = sm.num_candidates;
  const int n = sm.num_candidates;
  int k = 0;
  if (J.mode == 0)
    J.out_records = sm.num_sets * slots;
  else if (J.num_selected > 0)
    J.out_records = k = std::min(J.num_selected, n);
  else
    J.out_records = n;
  J.num_hands = J.out_records;
  if ((long long)J.out_records > J.capacity) {
    set_error("detect: %d hand records to return, the caller's buffer holds %lld", J.out_records, J.capacity);
    return GPD_ERR_INVALID;
  }
  (void)hipEventElapsedTime(&L.stage_ms[0], L.ev[0], L.ev[1]);  // here: the next job on this lane records them again
  HIP_TRY(hipEventRecord(L.ev[4], L.stream));
  L.images.side_stream = !ctx->in_batch;
  L.images.lcg_base = J.lcg_base;
  int rc;
  {
    StageRange r("gpd:images (shadow sets, shadow channels, normals + depth channels)");
    rc = images_run(ctx->params, L.cloud, L.search, L.plan, L.images, L.stream);
  }
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev[2], L.stream));
  if (n > 0) {
    rc = reserve_scores(L, n);
    if (rc) return rc;
    {
      StageRange r("gpd:lenet (conv1, conv2, ip1, ip2)");
      HIP_TRY(lenet_forward(ctx->lenet, L.lenet_scratch, L.images.d_images, n, L.d_scores, L.stream));
    }
    HIP_TRY(hipMemcpyAsync(&L.h_flags->lenet, L.lenet_scratch.c1_stats + 2, sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
  } else {
    L.h_flags->lenet = 0;
  }
  HIP_TRY(hipEventRecord(L.ev[3], L.stream));
  rc = reserve_out(L, (size_t)J.out_records, k ? (size_t)n * sizeof(float) : 0);
  if (rc) return rc;
  L.h_flags->tie = 0;
  if (J.mode == 0) {
    rc = plan_emit_hands(ctx->params, L.search, L.plan, n > 0 ? L.d_scores : nullptr, L.d_out, false, L.stream);
  } else if (k > 0) {
    rc = reserve_selection(L, k, n);
    if (rc) return rc;
    // every candidate record, scored, in a list of this job's own: the selection gathers from it, and so does the
    // std::partial_sort rerun of job_end — by then, in a batch, the lane's search / plan buffers already hold the
    // cloud after next (begin(i + 1) is enqueued before end(i - 1))
    rc = plan_emit_hands(ctx->params, L.search, L.plan, L.d_scores, L.d_all, true, L.stream);
    if (rc) return rc;
    if (k <= select_topk_capacity()) {
      rc = select_topk(L.d_scores, n, k, L.d_sel, L.d_sel + k, L.stream);
      if (rc) return rc;
      rc = gather_records(L.d_all, L.d_sel, k, L.d_out, L.stream);
      if (rc) return rc;
      HIP_TRY(hipMemcpyAsync(&L.h_flags->tie, L.d_sel + k, sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
    } else {
      L.h_flags->tie = 2;  // more winners than the device selection sorts: std::partial_sort on the host (job_end), no limit
    }
    // the scores (4 bytes per candidate) ride along: equal scores are settled with std::partial_sort on the host
    HIP_TRY(hipMemcpyAsync(L.h_out + L.d_out_cap * sizeof(gpd_hand), L.d_scores, (size_t)n * sizeof(float), hipMemcpyDeviceToHost,
                           L.stream));
  } else {
    rc = plan_emit_hands(ctx->params, L.search, L.plan, L.d_scores, L.d_out, true, L.stream);
  }
  if (rc) return rc;
  // A megabyte or more of records (all hand sets of a cloud: 3.6 MB) leaves in four copies with an event behind each, so that
  // job_end hands chunk c to the caller while chunk c + 1 is still on the bus: the pinned-to-caller memcpy (0.2 ms for 3.6 MB)
  // used to start only after the last byte had arrived.  Not for selections: their records may be gathered again (ties).
  J.chunks = ((size_t)J.out_records * sizeof(gpd_hand) >= (1u << 20) && !(J.mode == 1 && J.num_selected > 0)) ? 4 : 0;
  if (J.chunks) {
    const size_t per = ((size_t)J.out_records + J.chunks - 1) / J.chunks;
    for (int c = 0; c < J.chunks; c++) {
      const size_t r0 = std::min((size_t)c * per, (size_t)J.out_records), r1 = std::min(r0 + per, (size_t)J.out_records);
      if (r1 > r0)
        HIP_TRY(hipMemcpyAsync(L.h_out + r0 * sizeof(gpd_hand), L.d_out + r0, (r1 - r0) * sizeof(gpd_hand), hipMemcpyDeviceToHost, L.stream));
      HIP_TRY(hipEventRecord(L.ev_chunk[c], L.stream));
    }
  } else if (J.out_records > 0) {
    HIP_TRY(hipMemcpyAsync(L.h_out, L.d_out, (size_t)J.out_records * sizeof(gpd_hand), hipMemcpyDeviceToHost, L.stream));
  }
  HIP_TRY(hipMemcpyAsync(&L.h_flags->status, L.images.d_status, sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
  HIP_TRY(hipEventRecord(L.ev_done, L.stream));
  J.live = true;
  return GPD_OK;
}

static int job_middle(gpd_hip_ctx *ctx, Lane &L, Job &J) {
  const int rc = job_wait_plan(ctx, L, J);
  return rc ? rc : job_enqueue(ctx, L, J);
}

static bool score_greater(const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) { return a.first > b.first; }

static int job_end(gpd_hip_ctx *ctx, Lane &L, Job &J) {
  if (!J.live) return GPD_OK;
  J.live = false;
  double early_copy_ms = 0.0;
  if (J.chunks) {
    // (should a flag below turn out set, the caller's buffer holds records of a failed call: its content is unspecified then)
    const size_t per = ((size_t)J.out_records + J.chunks - 1) / J.chunks;
    for (int c = 0; c < J.chunks; c++) {
      HIP_TRY(hipEventSynchronize(L.ev_chunk[c]));
      const auto t0 = std::chrono::steady_clock::now();
      const size_t r0 = std::min((size_t)c * per, (size_t)J.out_records), r1 = std::min(r0 + per, (size_t)J.out_records);
      if (r1 > r0) std::memcpy(J.hands + r0, L.h_out + r0 * sizeof(gpd_hand), (r1 - r0) * sizeof(gpd_hand));
      early_copy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  HIP_TRY(hipEventSynchronize(L.ev_done));
  const auto t_done = std::chrono::steady_clock::now();
  (void)hipEventElapsedTime(&L.stage_ms[1], L.ev[4], L.ev[2]);
  (void)hipEventElapsedTime(&L.stage_ms[2], L.ev[2], L.ev[3]);
  if (L.h_flags->status) {
    images_status_text(L.h_flags->status, g_err, sizeof(g_err));
    return GPD_ERR_CAPACITY;
  }
  if (L.h_flags->lenet) {
    const int rc = lenet_check(L.lenet_scratch);  // clears the device word, sets the error text
    return rc ? rc : GPD_ERR_HIP;
  }
  const int n = J.num_candidates;
  if (J.mode == 1 && J.num_selected > 0 && J.out_records > 0 && L.h_flags->tie) {
    // equal scores among the winners: the reference's result is whatever std::partial_sort leaves
    // (grasp_detector.cpp:409), which depends on the history of its heap — so run exactly that, on
    // (score, candidate) pairs in candidate order, and gather the winners again
    const float *sc = reinterpret_cast<const float *>(L.h_out + L.d_out_cap * sizeof(gpd_hand));
    std::vector<std::pair<float, int32_t>> v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = {sc[i], i};
    const int k = J.out_records;
    std::partial_sort(v.begin(), v.begin() + k, v.end(), score_greater);
    std::vector<int32_t> sel((size_t)k);
    for (int i = 0; i < k; i++) sel[i] = v[i].second;
    HIP_TRY(hipMemcpyAsync(L.d_sel, sel.data(), (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, L.stream));
    int rc = gather_records(L.d_all, L.d_sel, k, L.d_out, L.stream);  // not from L.search / L.plan: see job_middle
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(L.h_out, L.d_out, (size_t)k * sizeof(gpd_hand), hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
  }
  if (J.out_records > 0 && !J.chunks) std::memcpy(J.hands, L.h_out, (size_t)J.out_records * sizeof(gpd_hand));
  J.copy_ms = early_copy_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_done).count();
  return GPD_OK;
}

static int check_samples(gpd_hip_ctx *ctx, const Lane &L, const char *who, const int32_t *sample_indices, const double *sample_xyz,
                         int num_samples, int num_points) {
  (void)ctx;
  (void)L;
  if (sample_indices) {
    for (int i = 0; i < num_samples; i++)
      if (sample_indices[i] < 0 || sample_indices[i] >= num_points) {
        set_error("%s: sample index %d out of range", who, sample_indices[i]);
        return GPD_ERR_INVALID;
      }
  } else {
    for (int i = 0; i < 3 * num_samples; i++)
      if (!std::isfinite(sample_xyz[i])) {
        set_error("%s: sample %d is not finite", who, i / 3);
        return GPD_ERR_INVALID;
      }
  }
  return GPD_OK;
}

extern "C" {

void gpd_hip_default_params(gpd_params *p) {
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
  p->workspace_grasps[0] = -1;
  p->workspace_grasps[1] = 1;
  p->workspace_grasps[2] = -1;
  p->workspace_grasps[3] = 1;
  p->workspace_grasps[4] = -1;
  p->workspace_grasps[5] = 1;
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

const char *gpd_hip_last_error(void) { return g_err; }

int gpd_hip_create(int device, const gpd_params *params, gpd_hip_ctx **out) {
  if (!params || !out) {
    set_error("gpd_hip_create: null argument");
    return GPD_ERR_INVALID;
  }
  *out = nullptr;
  const int C = params->image_num_channels;
  if (params->image_size != kImg || (C != 1 && C != 3 && C != 12 && C != 15)) {
    set_error("gpd_hip_create: image_size must be 60 and image_num_channels one of 1/3/12/15");
    return GPD_ERR_INVALID;
  }
  const int slots = params->num_hand_axes * params->num_orientations;
  if (params->num_hand_axes < 1 || params->num_hand_axes > 3 || params->num_orientations < 1 || slots < 1 || slots > GPD_MAX_SLOTS ||
      params->num_finger_placements < 1 || params->num_finger_placements > 16) {
    set_error("gpd_hip_create: unsupported num_hand_axes/num_orientations/num_finger_placements");
    return GPD_ERR_INVALID;
  }
  for (int a = 0; a < params->num_hand_axes; a++)
    if (params->hand_axes[a] < 0 || params->hand_axes[a] > 2) {  // index into the unit axes (hand_set.cpp:52-53)
      set_error("gpd_hip_create: hand_axes[%d] = %d is not one of 0, 1, 2", a, params->hand_axes[a]);
      return GPD_ERR_INVALID;
    }
  {
    // lengths that end up as divisors, box extents and radii
    const double pos[] = {params->finger_width,  params->hand_outer_diameter, params->hand_depth,       params->hand_height,
                          params->init_bite,     params->volume_width,        params->volume_depth,     params->volume_height,
                          params->nn_radius_frames};
    static const char *names[] = {"finger_width", "hand_outer_diameter", "hand_depth",   "hand_height",     "init_bite",
                                  "volume_width", "volume_depth",        "volume_height", "nn_radius_frames"};
    for (size_t i = 0; i < sizeof(pos) / sizeof(pos[0]); i++)
      if (!(pos[i] > 0.0) || !std::isfinite(pos[i])) {
        set_error("gpd_hip_create: %s must be positive and finite", names[i]);
        return GPD_ERR_INVALID;
      }
    const double fin[] = {params->friction_coeff,      params->min_aperture,        params->max_aperture,
                          params->workspace_grasps[0], params->workspace_grasps[1], params->workspace_grasps[2],
                          params->workspace_grasps[3], params->workspace_grasps[4], params->workspace_grasps[5]};
    for (double v : fin)
      if (std::isnan(v)) {
        set_error("gpd_hip_create: NaN in friction_coeff / apertures / workspace_grasps");
        return GPD_ERR_INVALID;
      }
    if (params->filter_approach_direction)
      for (double v : {params->direction[0], params->direction[1], params->direction[2], params->thresh_rad})
        if (!std::isfinite(v)) {
          set_error("gpd_hip_create: direction / thresh_rad must be finite when filter_approach_direction is set");
          return GPD_ERR_INVALID;
        }
    if (!(params->hand_outer_diameter > params->finger_width)) {
      set_error("gpd_hip_create: hand_outer_diameter must exceed finger_width");
      return GPD_ERR_INVALID;
    }
    // deepenHand's steps (finger_hand.cpp:116-121) come from a 128-entry table (fingers up to init_bite + 0.64 m)
    int steps = 0;
    for (double d = params->init_bite + 0.005; d <= params->hand_depth && steps <= 128; d += 0.005) steps++;
    if (params->deepen_hand && steps > 128) {
      set_error("gpd_hip_create: hand_depth %.3f needs more than 128 deepening steps of 5 mm from init_bite %.3f", params->hand_depth,
                params->init_bite);
      return GPD_ERR_CAPACITY;
    }
  }
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) {
    set_error("gpd_hip_create: device %d out of range (%d devices)", device, count);
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(device));
  gpd_hip_ctx *ctx = new gpd_hip_ctx();
  ctx->device = device;
  ctx->params = *params;
  const int rc = lane_init(ctx->lane[0]);
  if (rc) {
    gpd_hip_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return GPD_OK;
}

void gpd_hip_destroy(gpd_hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (int l = kLanes - 1; l >= 0; l--) lane_free(ctx->lane[l]);
  preprocess_free(ctx->pre);
  cluster_free(ctx->cluster);
  for (auto &e : ctx->pre.ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->pre.ev_keys) (void)hipEventDestroy(ctx->pre.ev_keys);
  float **ws[] = {&ctx->lenet.c1w, &ctx->lenet.c1b, &ctx->lenet.c2w, &ctx->lenet.c2b,
                  &ctx->lenet.f1w, &ctx->lenet.f1b, &ctx->lenet.f2w, &ctx->lenet.f2b, &ctx->lenet.c1wp, &ctx->lenet.c2wt};
  for (float **p : ws)
    if (*p) (void)hipFree(*p);
  lenet_fast_free(ctx->lenet.fast);
  for (auto &e : ctx->replay_events) (void)hipEventDestroy(e);
  if (ctx->pipe_stream) {
    (void)hipStreamSynchronize(ctx->pipe_stream);
    (void)hipStreamDestroy(ctx->pipe_stream);
    if (ctx->pipe_images[1]) (void)hipFree(ctx->pipe_images[1]);
    for (int b = 0; b < 2; b++) {
      (void)hipEventDestroy(ctx->pipe_filled[b]);
      (void)hipEventDestroy(ctx->pipe_read[b]);
    }
  }
  delete ctx;
}

int gpd_hip_reserve(gpd_hip_ctx *ctx, int max_points, int max_cams, int max_samples, int max_candidates, int max_selected) {
  if (!ctx || max_points < 1 || max_cams < 1 || max_cams > kMaxCams || max_samples < 0 || max_candidates < 0 || max_selected < 0) {
    set_error("gpd_hip_reserve: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const int cand = max_candidates > 0 ? max_candidates : candidate_bound(ctx->params, max_samples, max_cams);
  for (int l = 0; l < kLanes; l++) {
    int rc = lane_init(ctx->lane[l]);
    if (rc) return rc;
    Lane &L = ctx->lane[l];
    HIP_TRY(hipStreamSynchronize(L.stream));
    // lane 0 also serves the single-cloud entries (all slots of all sets back); both serve the batch (candidates / winners)
    if (l == 0) {
      rc = lane_reserve(ctx, L, max_points, max_cams, max_samples, cand, -1);
      if (rc) return rc;
    }
    rc = lane_reserve(ctx, L, max_points, max_cams, max_samples, cand, 0);
    if (!rc && max_selected > 0) rc = lane_reserve(ctx, L, max_points, max_cams, max_samples, cand, max_selected);
    if (rc) return rc;
  }
  return GPD_OK;
}

int gpd_hip_set_lenet_weights(gpd_hip_ctx *ctx, int channels, const float *conv1_w, const float *conv1_b, const float *conv2_w,
                              const float *conv2_b, const float *ip1_w, const float *ip1_b, const float *ip2_w,
                              const float *ip2_b) {
  if (!ctx || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !ip1_w || !ip1_b || !ip2_w || !ip2_b) {
    set_error("gpd_hip_set_lenet_weights: null argument");
    return GPD_ERR_INVALID;
  }
  if (channels != ctx->params.image_num_channels) {
    set_error("gpd_hip_set_lenet_weights: channels %d != image_num_channels %d", channels, ctx->params.image_num_channels);
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  struct Item {
    float **dst;
    const float *src;
    size_t n;
  } items[] = {{&ctx->lenet.c1w, conv1_w, (size_t)20 * channels * 25}, {&ctx->lenet.c1b, conv1_b, 20},
               {&ctx->lenet.c2w, conv2_w, (size_t)50 * 500},          {&ctx->lenet.c2b, conv2_b, 50},
               {&ctx->lenet.f1w, ip1_w, (size_t)kFc1In * kFc1Out},    {&ctx->lenet.f1b, ip1_b, kFc1Out},
               {&ctx->lenet.f2w, ip2_w, (size_t)2 * kFc1Out},         {&ctx->lenet.f2b, ip2_b, 2}};
  // the f32 chain's conv1 drops input windows that are entirely zero; that is exact only for finite weights
  // (inf * 0 would be NaN in the reference's dense GEMM); the split path cuts conv1 / conv2 / ip1 weights into
  // fixed-point digits / bf16 pieces, which are defined for finite numbers
  {
    struct {
      const char *name;
      const float *w;
      size_t n;
    } fin[] = {{"conv1", conv1_w, (size_t)20 * channels * 25}, {"conv2", conv2_w, (size_t)50 * 500}, {"ip1", ip1_w, (size_t)kFc1In * kFc1Out}};
    for (auto &it : fin)
      for (size_t i = 0; i < it.n; i++)
        if (!std::isfinite(it.w[i])) {
          set_error("gpd_hip_set_lenet_weights: %s weight %zu is not finite", it.name, i);
          return GPD_ERR_INVALID;
        }
  }
  for (auto &L : ctx->lane)
    if (L.stream) HIP_TRY(hipStreamSynchronize(L.stream));  // no kernel still reads the old weights
  // from here until every upload has succeeded the context holds NO weights: an update that fails half way (out of memory in
  // lenet_fast_prepare, say) leaves scoring calls refused with "LeNet weights not set" instead of launching on freed tables
  ctx->lenet.channels = 0;
  for (auto &it : items) {
    if (*it.dst) (void)hipFree(*it.dst);
    *it.dst = nullptr;
    HIP_TRY(hipMalloc(it.dst, it.n * sizeof(float)));
    HIP_TRY(hipMemcpy(*it.dst, it.src, it.n * sizeof(float), hipMemcpyHostToDevice));
  }
  // device-side layouts of the conv weights (file layout is [filter][k], conv_layer.cpp:35-36):
  // conv1 rows padded from 25 to 28 taps per channel, conv2 k-major
  {
    const int K1 = channels * 25;
    std::vector<float> t1((size_t)20 * channels * 28, 0.f), t2((size_t)500 * 50);
    for (int f = 0; f < 20; f++)
      for (int c = 0; c < channels; c++)
        for (int t = 0; t < 25; t++) t1[((size_t)f * channels + c) * 28 + t] = conv1_w[(size_t)f * K1 + c * 25 + t];
    for (int f = 0; f < 50; f++)
      for (int k = 0; k < 500; k++) t2[(size_t)k * 50 + f] = conv2_w[(size_t)f * 500 + k];
    struct {
      float **dst;
      std::vector<float> *src;
    } tr[] = {{&ctx->lenet.c1wp, &t1}, {&ctx->lenet.c2wt, &t2}};
    for (auto &it : tr) {
      if (*it.dst) (void)hipFree(*it.dst);
      *it.dst = nullptr;
      HIP_TRY(hipMalloc(it.dst, it.src->size() * sizeof(float)));
      HIP_TRY(hipMemcpy(*it.dst, it.src->data(), it.src->size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(lenet_fast_prepare(ctx->lenet.fast, channels, conv1_w, conv2_w, ip1_w));
  ctx->lenet.channels = channels;
  return GPD_OK;
}

int gpd_hip_set_lenet_mode(gpd_hip_ctx *ctx, int mode) {
  if (!ctx || (mode != GPD_LENET_SPLIT && mode != GPD_LENET_F32_CHAIN)) {
    set_error("gpd_hip_set_lenet_mode: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  for (auto &L : ctx->lane)
    if (L.stream) HIP_TRY(hipStreamSynchronize(L.stream));  // the two modes lay pool1 out differently: no launch in flight
  ctx->lenet.mode = mode;
  return GPD_OK;
}

// test hook: the intermediate tensors of lane 0's last LeNet pass (which = 0: pool1 as f32 [n][15680], 1: the three bf16
// planes of the flattened pool2 [3][n][7200] (split path), 2: fc1 transposed f32 [500][n])
int gpd_hip_lenet_debug(gpd_hip_ctx *ctx, int which, int n, void *out) {
  if (!ctx || !out || n < 1) {
    set_error("gpd_hip_lenet_debug: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  LeNetScratch &s = L.lenet_scratch;
  if (n > s.capacity) {
    set_error("gpd_hip_lenet_debug: n = %d exceeds the scratch capacity %d", n, s.capacity);
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipStreamSynchronize(L.stream));
  if (which == 0) {
    HIP_TRY(hipMemcpy(out, s.pool1, (size_t)n * 15680 * sizeof(float), hipMemcpyDeviceToHost));
  } else if (which == 1) {
    const size_t rows = ((size_t)n + 15) & ~(size_t)15;
    std::vector<unsigned short> blocked(3 * rows * kLenetXld);
    HIP_TRY(hipMemcpy(blocked.data(), s.xs, blocked.size() * sizeof(unsigned short), hipMemcpyDeviceToHost));
    lenet_fast_unblock_x(blocked.data(), n, static_cast<unsigned short *>(out));
  } else if (which == 2 && ctx->lenet.mode == GPD_LENET_SPLIT) {
    // the split path keeps ip1's output as four partial sums (they meet inside ip2's kernel): added here as there, in order
    const size_t rows = ((size_t)n + 31) & ~(size_t)31;
    std::vector<float> part(rows * 4 * kFc1Out), b1(kFc1Out);
    HIP_TRY(hipMemcpy(part.data(), s.fc1p, part.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b1.data(), ctx->lenet.f1b, b1.size() * sizeof(float), hipMemcpyDeviceToHost));
    float *o = static_cast<float *>(out);
    for (int u = 0; u < kFc1Out; u++)
      for (int m = 0; m < n; m++) {
        const float v = ((part[fc1p_index(m, 0, u)] + part[fc1p_index(m, 1, u)]) + part[fc1p_index(m, 2, u)]) + part[fc1p_index(m, 3, u)] + b1[u];
        o[(size_t)u * n + m] = v > 0.f ? v : 0.f;
      }
  } else if (which == 2) {
    HIP_TRY(hipMemcpy2D(out, (size_t)n * sizeof(float), s.fc1t, (size_t)s.capacity * sizeof(float), (size_t)n * sizeof(float), kFc1Out,
                        hipMemcpyDeviceToHost));
  } else {
    set_error("gpd_hip_lenet_debug: which = %d", which);
    return GPD_ERR_INVALID;
  }
  return GPD_OK;
}

int gpd_hip_score(gpd_hip_ctx *ctx, const uint8_t *images, int n, float *scores) {
  StageRange range_("gpd:score (classifyImages)");
  if (!ctx || !scores || n < 0) {
    set_error("gpd_hip_score: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->lenet.channels) {
    set_error("gpd_hip_score: LeNet weights not set");
    return GPD_ERR_STATE;
  }
  if (n == 0) return GPD_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  const size_t bytes = (size_t)n * kPix * ctx->lenet.channels;
  const uint8_t *d_img = nullptr;
  if (images) {
    if (bytes > L.d_img_in_bytes) {
      if (L.d_img_in) (void)hipFree(L.d_img_in);
      if (L.d_img_planar) (void)hipFree(L.d_img_planar);
      L.d_img_in = nullptr;
      L.d_img_planar = nullptr;
      L.d_img_in_bytes = 0;
      HIP_TRY(hipMalloc(&L.d_img_in, bytes));
      HIP_TRY(hipMalloc(&L.d_img_planar, bytes));
      L.d_img_in_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(L.d_img_in, images, bytes, hipMemcpyHostToDevice, L.stream));
    HIP_TRY(hwc_to_planar(L.d_img_in, L.d_img_planar, n, ctx->lenet.channels, L.stream));
    d_img = L.d_img_planar;
  } else {
    if (n != L.images.num_candidates || !L.images.d_images) {
      set_error("gpd_hip_score: no device images for n=%d (gpd_hip_images produced %d)", n, L.images.num_candidates);
      return GPD_ERR_STATE;
    }
    d_img = L.images.d_images;
  }
  int rc = reserve_scores(L, n);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev[2], L.stream));
  HIP_TRY(lenet_forward(ctx->lenet, L.lenet_scratch, d_img, n, L.d_scores, L.stream));
  HIP_TRY(hipEventRecord(L.ev[3], L.stream));
  HIP_TRY(hipMemcpyAsync(scores, L.d_scores, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, L.stream));
  HIP_TRY(hipStreamSynchronize(L.stream));
  HIP_TRY(hipEventElapsedTime(&L.stage_ms[2], L.ev[2], L.ev[3]));
  return lenet_check(L.lenet_scratch);
}

int gpd_hip_upload_cloud(gpd_hip_ctx *ctx, const float *xyz, const float *normals, int num_points, const int32_t *cam_source,
                         int num_cams, const double *view_points) {
  StageRange range_("gpd:upload_cloud (+ uniform grid)");
  if (!ctx || !xyz || !normals || num_points <= 0 || !cam_source || num_cams < 1 || !view_points) {
    set_error("gpd_hip_upload_cloud: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  return cloud_upload(L.cloud, xyz, normals, num_points, cam_source, num_cams, view_points, L.stream, /*sync=*/true);
}

int gpd_hip_find_clusters(gpd_hip_ctx *ctx, const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers,
                          gpd_hand *out, double *out_scores, int32_t *out_src, int *num_out) {
  StageRange range_("gpd:find_clusters");
  if (!ctx || !num_out || n < 0 || (n > 0 && (!hands || !scores || !out || !out_scores || !out_src))) {
    set_error("gpd_hip_find_clusters: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return cluster_run(ctx->cluster, hands, scores, n, min_inliers, remove_inliers, out, out_scores, out_src, num_out, ctx->lane[0].stream);
}

int gpd_hip_preprocess_cloud(gpd_hip_ctx *ctx, const float *xyz, const int32_t *cam_source, int num_points, int num_cams,
                             const double *workspace, float voxel_size, float *xyz_out, int32_t *cam_out, int32_t *src_out,
                             int *num_out, float *kernel_ms) {
  StageRange range_("gpd:preprocess_cloud (workspace cut, voxeliser)");
  if (!ctx || !num_out || num_points < 0 || num_cams < 0 || (num_points > 0 && (!xyz || !xyz_out)) ||
      (num_points > 0 && num_cams > 0 && (!cam_source || !cam_out)) || !(voxel_size == voxel_size)) {
    set_error("gpd_hip_preprocess_cloud: bad argument");
    return GPD_ERR_INVALID;
  }
  if (workspace)
    for (int a = 0; a < 6; a++)
      if (workspace[a] != workspace[a]) {
        set_error("gpd_hip_preprocess_cloud: the workspace holds a NaN");
        return GPD_ERR_INVALID;
      }
  HIP_TRY(hipSetDevice(ctx->device));
  return preprocess_run(ctx->pre, xyz, cam_source, num_points, num_cams, workspace, voxel_size, xyz_out, cam_out, src_out, num_out,
                        kernel_ms, ctx->lane[0].stream);
}

int gpd_hip_estimate_normals(gpd_hip_ctx *ctx, double radius, float *normals) {
  StageRange range_("gpd:estimate_normals");
  if (!ctx || !normals || !(radius > 0.0)) {
    set_error("gpd_hip_estimate_normals: bad argument");
    return GPD_ERR_INVALID;
  }
  Lane &L = ctx->lane[0];
  if (!L.cloud.num_points) {
    set_error("gpd_hip_estimate_normals: no cloud uploaded");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return normals_run(L.cloud, radius, normals, L.stream);
}

// samples by index (sample_xyz == nullptr) or by coordinates (sample_indices == nullptr)
static int search_any(gpd_hip_ctx *ctx, const char *who, const int32_t *sample_indices, const double *sample_xyz, int num_samples,
                      gpd_hand *hands, int *num_sets) {
  if (!ctx || (!sample_indices && !sample_xyz) || num_samples < 0 || !hands || !num_sets) {
    set_error("%s: bad argument", who);
    return GPD_ERR_INVALID;
  }
  Lane &L = ctx->lane[0];
  if (!L.cloud.num_points) {
    set_error("%s: no cloud uploaded", who);
    return GPD_ERR_STATE;
  }
  *num_sets = 0;
  if (num_samples == 0) return GPD_OK;
  int rc = check_samples(ctx, L, who, sample_indices, sample_xyz, num_samples, L.cloud.num_points);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord(L.ev[0], L.stream));
  rc = search_run(ctx->params, L.cloud, L.search, sample_indices, sample_xyz, num_samples, L.stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev[1], L.stream));
  rc = search_download(ctx->params, L.search, hands, num_sets, L.stream);
  if (rc) return rc;
  HIP_TRY(hipEventElapsedTime(&L.stage_ms[0], L.ev[0], L.ev[1]));
  return GPD_OK;
}

int gpd_hip_search(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, gpd_hand *hands, int *num_sets) {
  StageRange range_("gpd:search (unfused entry)");
  if (!sample_indices) {
    set_error("gpd_hip_search: bad argument");
    return GPD_ERR_INVALID;
  }
  return search_any(ctx, "gpd_hip_search", sample_indices, nullptr, num_samples, hands, num_sets);
}

int gpd_hip_search_samples(gpd_hip_ctx *ctx, const double *samples_xyz, int num_samples, gpd_hand *hands, int *num_sets) {
  if (!samples_xyz) {
    set_error("gpd_hip_search_samples: bad argument");
    return GPD_ERR_INVALID;
  }
  return search_any(ctx, "gpd_hip_search_samples", nullptr, samples_xyz, num_samples, hands, num_sets);
}

int gpd_hip_reevaluate(gpd_hip_ctx *ctx, gpd_hand *hands, int num_hands, int32_t *labels) {
  StageRange range_("gpd:reevaluate");
  if (!ctx || num_hands < 0 || (num_hands > 0 && (!hands || !labels))) {
    set_error("gpd_hip_reevaluate: bad argument");
    return GPD_ERR_INVALID;
  }
  Lane &L = ctx->lane[0];
  if (!L.cloud.num_points) {
    set_error("gpd_hip_reevaluate: no cloud uploaded");
    return GPD_ERR_STATE;
  }
  if (num_hands == 0) return GPD_OK;
  for (int i = 0; i < num_hands; i++)
    for (int r = 0; r < 3; r++)
      if (!std::isfinite(hands[i].sample[r])) {
        set_error("gpd_hip_reevaluate: hand %d has a non-finite sample", i);
        return GPD_ERR_INVALID;
      }
  HIP_TRY(hipSetDevice(ctx->device));
  L.images.num_candidates = 0;  // the search buffers the resident candidate list points into are reused
  return reevaluate_run(ctx->params, L.cloud, L.search, hands, num_hands, labels, L.stream);
}

int gpd_hip_images(gpd_hip_ctx *ctx, const gpd_hand *hands, int num_sets, uint8_t *images, int32_t *cand_index,
                   int *num_candidates) {
  StageRange range_("gpd:images (unfused entry)");
  if (!ctx || !hands || num_sets < 0 || !num_candidates) {
    set_error("gpd_hip_images: bad argument");
    return GPD_ERR_INVALID;
  }
  Lane &L = ctx->lane[0];
  if (!L.cloud.num_points) {
    set_error("gpd_hip_images: no cloud uploaded");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  SearchState &s = L.search;
  const int slots = ctx->params.num_hand_axes * ctx->params.num_orientations;
  *num_candidates = 0;
  if (s.cloud_generation != L.cloud.generation || s.num_samples == 0) {
    set_error("images: hands must come from gpd_hip_search / gpd_hip_detect on this context and cloud");
    return GPD_ERR_STATE;
  }
  if (num_sets > s.num_samples) {
    set_error("images: %d sets passed, the search had %d samples", num_sets, s.num_samples);
    return GPD_ERR_INVALID;
  }
  // the caller's validity flags (the host filters between the stages clear them: grasp_detector.cpp:238-255)
  // replace the ones the search left on the device; everything else about the hands is already there.  The
  // sets' samples travel along: plan_kernel checks them against the search's (a moved hand set is refused).
  std::vector<uint8_t> fv((size_t)num_sets * slots + 1, 0);
  std::vector<double> smp((size_t)num_sets * 3 + 1, 0.0);
  for (int si = 0; si < num_sets; si++) {
    for (int j = 0; j < slots; j++) fv[(size_t)si * slots + j] = hands[(size_t)si * slots + j].valid ? 1 : 0;
    for (int r = 0; r < 3; r++) smp[3 * (size_t)si + r] = hands[(size_t)si * slots].sample[r];
  }
  HIP_TRY(hipEventRecord(L.ev[0], L.stream));
  int rc = plan_build(ctx->params, L.cloud, s, L.plan, L.stream, fv.data(), smp.data(), num_sets);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(L.stream));  // fv / smp are pageable: their copies have left them by now as well
  if (L.plan.h_summary->mismatch_set >= 0) {
    set_error("images: set %d does not match the last search (sample moved, or more sets than the search produced)",
              L.plan.h_summary->mismatch_set);
    return GPD_ERR_STATE;
  }
  L.images.lcg_base = 0;  // gpd_hip_images: the cloud's stream of shadow draws from its start
  rc = images_run(ctx->params, L.cloud, s, L.plan, L.images, L.stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(L.ev[1], L.stream));
  const int n = L.images.num_candidates;
  *num_candidates = n;
  if (cand_index && n > 0)
    HIP_TRY(hipMemcpyAsync(cand_index, L.plan.d_cand_out, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
  if (images && n > 0) {
    // the caller wants cv::Mat-layout pixels: planar -> HWC on the device, then one copy
    const size_t bytes = (size_t)L.images.capacity * kPix * ctx->params.image_num_channels;
    if (!L.images.d_images_hwc) HIP_TRY(hipMalloc(&L.images.d_images_hwc, bytes));
    HIP_TRY(planar_to_hwc(L.images.d_images, L.images.d_images_hwc, n, ctx->params.image_num_channels, L.stream));
    HIP_TRY(hipMemcpyAsync(images, L.images.d_images_hwc, (size_t)n * kPix * ctx->params.image_num_channels, hipMemcpyDeviceToHost,
                           L.stream));
  }
  HIP_TRY(hipMemcpyAsync(&L.h_flags->status, L.images.d_status, sizeof(int32_t), hipMemcpyDeviceToHost, L.stream));
  HIP_TRY(hipStreamSynchronize(L.stream));
  HIP_TRY(hipEventElapsedTime(&L.stage_ms[1], L.ev[0], L.ev[1]));
  if (L.h_flags->status) {
    images_status_text(L.h_flags->status, g_err, sizeof(g_err));
    return GPD_ERR_CAPACITY;
  }
  return GPD_OK;
}

static int detect_any(gpd_hip_ctx *ctx, const char *who, const int32_t *sample_indices, const double *sample_xyz, int num_samples,
                      int mode, int num_selected, gpd_hand *hands, long long capacity, int *num_sets, int *num_candidates,
                      int *num_hands) {
  if (!ctx || !hands || !num_sets || !num_candidates || (!sample_indices && !sample_xyz) || num_samples < 0 || num_selected < 0) {
    set_error("%s: bad argument", who);
    return GPD_ERR_INVALID;
  }
  if (!ctx->lenet.channels) {
    set_error("%s: LeNet weights not set", who);
    return GPD_ERR_STATE;
  }
  Lane &L = ctx->lane[0];
  if (!L.cloud.num_points) {
    set_error("%s: no cloud uploaded", who);
    return GPD_ERR_STATE;
  }
  *num_sets = 0;
  *num_candidates = 0;
  if (num_hands) *num_hands = 0;
  int rc = check_samples(ctx, L, who, sample_indices, sample_xyz, num_samples, L.cloud.num_points);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  // GPD_DETECT_TIMING=1: wall time of the three steps, to stderr
  const bool timing = prof_env("GPD_DETECT_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  Job J;
  J.sample_idx = sample_indices;
  J.sample_xyz = sample_xyz;
  J.S = num_samples;
  J.mode = mode;
  J.num_selected = num_selected;
  J.hands = hands;
  J.capacity = capacity;
  rc = job_begin(ctx, L, J);
  if (rc) return rc;
  const double t1 = now();
  rc = job_middle(ctx, L, J);
  if (rc) return rc;
  const double t2 = now();
  rc = job_end(ctx, L, J);
  if (rc) return rc;
  *num_sets = J.num_sets;
  *num_candidates = J.num_candidates;
  if (num_hands) *num_hands = J.num_hands;
  if (timing)
    fprintf(stderr, "[detect-timing] enqueue search+plan %.3f ms, wait+enqueue images/LeNet/gather %.3f, wait+copy out %.3f; kernels: search %.3f images %.3f LeNet %.3f\n",
            t1 - t0, t2 - t1, now() - t2, L.stage_ms[0], L.stage_ms[1], L.stage_ms[2]);
  return GPD_OK;
}

int gpd_hip_detect(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, gpd_hand *hands, int *num_sets,
                   int *num_candidates) {
  if (!sample_indices) {
    set_error("gpd_hip_detect: bad argument");
    return GPD_ERR_INVALID;
  }
  const long long cap = ctx ? (long long)num_samples * ctx->params.num_hand_axes * ctx->params.num_orientations : 0;
  return detect_any(ctx, "gpd_hip_detect", sample_indices, nullptr, num_samples, 0, 0, hands, cap, num_sets, num_candidates, nullptr);
}

int gpd_hip_detect_samples(gpd_hip_ctx *ctx, const double *samples_xyz, int num_samples, gpd_hand *hands, int *num_sets,
                           int *num_candidates) {
  if (!samples_xyz) {
    set_error("gpd_hip_detect_samples: bad argument");
    return GPD_ERR_INVALID;
  }
  const long long cap = ctx ? (long long)num_samples * ctx->params.num_hand_axes * ctx->params.num_orientations : 0;
  return detect_any(ctx, "gpd_hip_detect_samples", nullptr, samples_xyz, num_samples, 0, 0, hands, cap, num_sets, num_candidates,
                    nullptr);
}

int gpd_hip_detect_select(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, int num_selected, gpd_hand *hands,
                          int hands_capacity, int *num_sets, int *num_candidates, int *num_hands) {
  if (!sample_indices || !num_hands || hands_capacity < 0) {
    set_error("gpd_hip_detect_select: bad argument");
    return GPD_ERR_INVALID;
  }
  return detect_any(ctx, "gpd_hip_detect_select", sample_indices, nullptr, num_samples, 1, num_selected, hands, hands_capacity, num_sets,
                    num_candidates, num_hands);
}

int gpd_hip_detect_batch(gpd_hip_ctx *ctx, gpd_detect_job *jobs, int num_jobs) {
  StageRange range_("gpd:detect_batch");
  if (!ctx || num_jobs < 0 || (num_jobs > 0 && !jobs)) {
    set_error("gpd_hip_detect_batch: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->lenet.channels) {
    set_error("gpd_hip_detect_batch: LeNet weights not set");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  for (int i = 0; i < num_jobs; i++) {
    gpd_detect_job &j = jobs[i];
    j.status = GPD_OK;
    j.num_sets = j.num_candidates = j.num_hands = 0;
    j.stage_ms[0] = j.stage_ms[1] = j.stage_ms[2] = 0.f;
    j.allocs = 0;
    for (float &t : j.host_ms) t = 0.f;
    const bool samples_ok = j.raw ? (j.sample_xyz != nullptr || j.num_samples == 0) : j.sample_indices != nullptr;
    if (!j.xyz || (!j.raw && !j.normals) || j.num_points <= 0 || !j.cam_source || j.num_cams < 1 || !j.view_points || !samples_ok ||
        j.num_samples < 0 || !j.hands || j.hands_capacity < 0 || j.num_selected < 0 ||
        (j.raw && !(j.normals_radius > 0.0 && std::isfinite(j.normals_radius) && std::isfinite(j.voxel_size)))) {
      set_error("gpd_hip_detect_batch: job %d has a bad argument", i);
      return GPD_ERR_INVALID;
    }
  }
  for (int l = 1; l < kLanes; l++) {
    const int rc = lane_init(ctx->lane[l]);
    if (rc) return rc;
  }
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_entry = now_ms();
  {
    // Both lanes sized once for the largest cloud of the batch before anything is enqueued: a buffer that grows in the
    // middle of the batch is a hipFree + hipMalloc, which waits for the whole device (both lanes).  Points, cameras and
    // samples are known; the candidate-sized buffers (images, LeNet scratch, records) take the upper bound
    // samples x slots, cut to kReserveBudget per lane — a cloud beyond that still grows its lane, and says so in `allocs`.
    int maxP = 0, maxC = 0, maxS = 0, maxSel = 0;
    bool all_selected = num_jobs > 0;
    for (int i = 0; i < num_jobs; i++) {
      maxP = std::max(maxP, jobs[i].num_points);
      maxC = std::max(maxC, jobs[i].num_cams);
      maxS = std::max(maxS, jobs[i].num_samples);
      maxSel = std::max(maxSel, jobs[i].num_selected);
      all_selected = all_selected && jobs[i].num_selected > 0;
    }
    if (maxC > kMaxCams) {
      set_error("gpd_hip_detect_batch: at most %d cameras are supported", kMaxCams);
      return GPD_ERR_INVALID;
    }
    const int before = g_allocs;
    // Best effort: the bound samples x slots is an UPPER bound — sized against what the device has free right now
    // (half of it over the lanes in use, the set bitsets counted), and when the allocation still fails (several contexts
    // on one GPU, a fragmented heap) the batch goes on: the buffers then grow on demand to the real candidate counts, as
    // before round 4 — a failed pre-size is a slower first pass, not a failed batch.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      free_b = 2 * kReserveBudget;
    }
    const int lanes_used = std::min(kLanes, num_jobs);
    const size_t budget = std::min(kReserveBudget, free_b / 2 / (size_t)std::max(lanes_used, 1));
    for (int l = 0; l < lanes_used; l++) {
      Lane &L = ctx->lane[l];
      if (L.search.nn_cap > 16384) {  // lists beyond the LDS sizes, left by a dense cloud of an earlier call: not pre-sized for a whole batch
        const int rcf = search_force_capacity(L.search, 8192);
        if (rcf) return rcf;
      }
      const int cand = candidate_bound(ctx->params, maxS, maxC, budget);
      int rc = GPD_OK;
      if (!all_selected) rc = lane_reserve(ctx, L, maxP, maxC, maxS, cand, 0);
      if (!rc && maxSel > 0) rc = lane_reserve(ctx, L, maxP, maxC, maxS, cand, maxSel);
      if (rc == GPD_ERR_HIP) {
        (void)hipGetLastError();  // out of memory: clear it, the jobs size their buffers themselves
        break;
      }
      if (rc) return rc;
    }
    if (num_jobs > 0) jobs[0].allocs = g_allocs - before;  // the pre-sizing is booked on the first cloud
  }
  std::vector<Job> J((size_t)num_jobs);
  ctx->in_batch = true;
  int first_error = GPD_OK;
  char first_text[sizeof(g_err)] = "";
  auto fail = [&](int i, int rc) {
    jobs[i].status = rc;
    J[i].live = false;
    // a job that fails after its uploads were enqueued leaves async copies out of the lane's pinned staging in flight;
    // the next job of the lane rewrites (or frees) that staging, so they have to land first
    (void)hipStreamSynchronize(ctx->lane[i % kLanes].stream);
    if (!first_error) {
      first_error = rc;
      std::memcpy(first_text, g_err, sizeof(g_err));
    }
  };
  std::vector<char> raw_pending((size_t)num_jobs, 0);
  // a raw scan's samples after Cloud::filterWorkspace (cloud.cpp:225-237): the reference cuts `samples_` with the cloud
  std::vector<std::vector<double>> raw_samples((size_t)num_jobs);
  // a RAW scan's first half: upload, workspace cut, voxel keys, the keys on their way to the host — nothing waits
  auto begin_raw = [&](int i) {
    gpd_detect_job &j = jobs[i];
    Lane &L = ctx->lane[i % kLanes];
    const int allocs0 = g_allocs;
    j.num_points_processed = 0;
    j.num_samples_processed = 0;
    int rc = check_samples(ctx, L, "gpd_hip_detect_batch", nullptr, j.sample_xyz, j.num_samples, j.num_points);
    if (!rc && j.workspace) {
      // strict double comparisons on both sides, sample by sample, order kept (cloud.cpp:229-233): a sample outside the cut would
      // still find neighbours, produce hand sets and shift every later set's shadow LCG offset away from the reference's
      const double *w = j.workspace;
      std::vector<double> &keep = raw_samples[(size_t)i];
      keep.reserve((size_t)j.num_samples * 3);
      for (int k = 0; k < j.num_samples; k++) {
        const double *q = j.sample_xyz + 3 * (size_t)k;
        if (q[0] > w[0] && q[0] < w[1] && q[1] > w[2] && q[1] < w[3] && q[2] > w[4] && q[2] < w[5]) keep.insert(keep.end(), q, q + 3);
      }
    }
    // Cloud::removeNans (candidates_generator.cpp:17) is part of preprocessPointCloud: non-finite points are dropped here
    if (!rc) rc = preprocess_begin(L.pre, j.xyz, j.cam_source, j.num_points, j.num_cams, j.workspace, j.voxel_size, L.stream, /*drop_nonfinite=*/true);
    j.allocs += g_allocs - allocs0;
    if (rc) return fail(i, rc);
    raw_pending[(size_t)i] = 1;
  };
  // ... and its second half, called once the previous cloud's image / LeNet kernels are in the queue: the voxeliser's sequential
  // chain on this host core (beside those kernels), the gather, the cloud built from the device arrays, the normals, the search
  auto begin_raw_finish = [&](int i) {
    if (!raw_pending[(size_t)i]) return;
    raw_pending[(size_t)i] = 0;
    gpd_detect_job &j = jobs[i];
    Lane &L = ctx->lane[i % kLanes];
    const int allocs0 = g_allocs;
    int rc = preprocess_finish(L.pre, L.stream);
    if (!rc && L.pre.M < 1) {
      set_error("gpd_hip_detect_batch: cloud %d: no point is left after the workspace cut", i);
      rc = GPD_ERR_INVALID;
    }
    if (!rc) rc = cloud_from_device(L.cloud, L.pre.d_out_xyz, L.pre.d_out_cam, L.pre.M, j.num_cams, j.view_points, L.stream);
    if (!rc) rc = normals_run(L.cloud, j.normals_radius, nullptr, L.stream);
    if (rc) return fail(i, rc);
    j.num_points_processed = L.pre.M;
    J[i].sample_idx = nullptr;
    J[i].sample_xyz = j.workspace ? raw_samples[(size_t)i].data() : j.sample_xyz;
    J[i].S = j.workspace ? (int)(raw_samples[(size_t)i].size() / 3) : j.num_samples;
    j.num_samples_processed = J[i].S;
    J[i].mode = 1;
    J[i].num_selected = j.num_selected;
    J[i].hands = j.hands;
    J[i].capacity = j.hands_capacity;
    J[i].lcg_base = j.lcg_base;
    rc = job_begin(ctx, L, J[i]);
    j.allocs += g_allocs - allocs0;
    j.host_ms[0] = (float)(now_ms() - t_entry);
    if (rc) fail(i, rc);
  };
  auto begin = [&](int i) {
    gpd_detect_job &j = jobs[i];
    if (j.raw) return begin_raw(i);
    Lane &L = ctx->lane[i % kLanes];
    const int allocs0 = g_allocs;
    int rc = check_samples(ctx, L, "gpd_hip_detect_batch", j.sample_indices, nullptr, j.num_samples, j.num_points);
    if (!rc) rc = cloud_upload(L.cloud, j.xyz, j.normals, j.num_points, j.cam_source, j.num_cams, j.view_points, L.stream, /*sync=*/false);
    if (rc) return fail(i, rc);
    J[i].sample_idx = j.sample_indices;
    J[i].S = j.num_samples;
    J[i].mode = 1;
    J[i].num_selected = j.num_selected;
    J[i].hands = j.hands;
    J[i].capacity = j.hands_capacity;
    J[i].lcg_base = j.lcg_base;
    rc = job_begin(ctx, L, J[i]);
    j.allocs += g_allocs - allocs0;
    j.host_ms[0] = (float)(now_ms() - t_entry);
    if (rc) fail(i, rc);
  };
  auto end = [&](int i) {
    Lane &L = ctx->lane[i % kLanes];
    const int rc = job_end(ctx, L, J[i]);
    jobs[i].host_ms[4] = (float)(now_ms() - t_entry);
    jobs[i].host_ms[3] = jobs[i].host_ms[4] - (float)J[i].copy_ms;
    if (rc) return fail(i, rc);
    jobs[i].num_sets = J[i].num_sets;
    jobs[i].num_candidates = J[i].num_candidates;
    jobs[i].num_hands = J[i].num_hands;
    jobs[i].lcg_draws = J[i].lcg_draws;
    for (int k = 0; k < 3; k++) jobs[i].stage_ms[k] = L.stage_ms[k];
  };
  // cloud i+1's upload + search are enqueued (other lane's buffers) before the host waits for cloud i's plan;
  // cloud i-1's results are collected after cloud i's image / LeNet kernels are in the queue
  // (stream order keeps cloud i+1's search behind the image / LeNet kernels of cloud i-1, whose buffers it reuses; of
  //  the pinned host buffers, begin touches the cloud / summary staging only, which job i-1 is done with since its
  //  own middle step)
  if (num_jobs > 0) {
    begin(0);
    begin_raw_finish(0);
  }
  for (int i = 0; i < num_jobs; i++) {
    if (i + 1 < num_jobs) begin(i + 1);
    if (jobs[i].status == GPD_OK) {
      const int allocs0 = g_allocs;
      const int rc = job_middle(ctx, ctx->lane[i % kLanes], J[i]);
      jobs[i].allocs += g_allocs - allocs0;
      jobs[i].host_ms[1] = (float)(J[i].t_plan_ms - t_entry);
      jobs[i].host_ms[2] = (float)(now_ms() - t_entry);
      if (rc) fail(i, rc);
    }
    if (i + 1 < num_jobs) begin_raw_finish(i + 1);  // (a raw scan: its host-side chain runs beside cloud i's image / LeNet kernels)
    if (i >= 1) end(i - 1);
  }
  if (num_jobs > 0) end(num_jobs - 1);
  ctx->in_batch = false;
  for (int l = 0; l < kLanes; l++) ctx->lane[l].images.side_stream = true;
  if (first_error) std::memcpy(g_err, first_text, sizeof(g_err));
  return first_error;
}

// The CPUs of the NUMA node a device hangs off (sysfs: the PCI function's numa_node, the node's cpulist).  Eight
// processes / threads feeding eight GPUs from a two-socket host is SURVEY 8e's expected limiter: a feeding thread that
// runs on the far socket pays the inter-socket hop on every staging copy and every doorbell.
static int device_numa_cpus(int device, cpu_set_t *set) {
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) return -1;
  for (char *c = bdf; *c; c++) *c = (char)std::tolower((unsigned char)*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  if (node < 0) return -1;  // a single-node host (or a VM that hides the topology): nothing to bind to
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return -1;
  CPU_ZERO(set);
  int a = 0, b = 0, n = 0;
  for (;;) {  // "0-63,128-191"
    if (fscanf(f, "%d", &a) != 1) break;
    b = a;
    int c = fgetc(f);
    if (c == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      c = fgetc(f);
    }
    for (int k = a; k <= b && k < CPU_SETSIZE; k++) {
      CPU_SET(k, set);
      n++;
    }
    if (c != ',') break;
  }
  fclose(f);
  return n > 0 ? node : -1;
}

int gpd_hip_bind_host_thread(int device, int *num_cpus) {
  if (num_cpus) *num_cpus = 0;
  cpu_set_t set;
  const int node = device_numa_cpus(device, &set);
  if (node < 0) return -1;
  // only within what the process is allowed to use (a container's cpuset)
  cpu_set_t allowed, both;
  if (pthread_getaffinity_np(pthread_self(), sizeof(allowed), &allowed) != 0) return -1;
  CPU_AND(&both, &set, &allowed);
  if (CPU_COUNT(&both) == 0) return -1;
  if (pthread_setaffinity_np(pthread_self(), sizeof(both), &both) != 0) return -1;
  if (num_cpus) *num_cpus = CPU_COUNT(&both);
  return node;
}

// One host thread per context (one context per GPU; more than one on a device is allowed), job i -> context i mod
// num_ctx: the in-process form of "independent clouds shard over the GPUs of a node" (the reference's unit
// of work is one detect_grasps run per cloud, src/detect_grasps.cpp:20-86).  No device talks to another.
int gpd_hip_detect_batch_multi(gpd_hip_ctx *const *ctxs, int num_ctx, gpd_detect_job *jobs, int num_jobs) {
  if (!ctxs || num_ctx < 1 || num_jobs < 0 || (num_jobs > 0 && !jobs)) {
    set_error("gpd_hip_detect_batch_multi: bad argument");
    return GPD_ERR_INVALID;
  }
  for (int c = 0; c < num_ctx; c++) {
    if (!ctxs[c]) {
      set_error("gpd_hip_detect_batch_multi: context %d is null", c);
      return GPD_ERR_INVALID;
    }
    for (int d = 0; d < c; d++)
      if (ctxs[d] == ctxs[c]) {
        set_error("gpd_hip_detect_batch_multi: context %d is listed twice (a context serves one thread)", c);
        return GPD_ERR_INVALID;
      }
  }
  std::vector<std::vector<gpd_detect_job>> mine((size_t)num_ctx);
  for (int i = 0; i < num_jobs; i++) mine[(size_t)(i % num_ctx)].push_back(jobs[i]);
  std::vector<int> rcs((size_t)num_ctx, GPD_OK);
  std::vector<std::string> texts((size_t)num_ctx);
  std::vector<std::thread> threads;
  for (int c = 0; c < num_ctx; c++)
    threads.emplace_back([&, c]() {
      (void)gpd_hip_bind_host_thread(ctxs[c]->device, nullptr);  // this worker feeds ONE device: keep it on that device's socket
      rcs[(size_t)c] = gpd_hip_detect_batch(ctxs[c], mine[(size_t)c].data(), (int)mine[(size_t)c].size());
      if (rcs[(size_t)c]) texts[(size_t)c] = g_err;  // the error text is per thread
    });
  for (auto &t : threads) t.join();
  for (int i = 0; i < num_jobs; i++) jobs[i] = mine[(size_t)(i % num_ctx)][(size_t)(i / num_ctx)];
  for (int c = 0; c < num_ctx; c++)
    if (rcs[(size_t)c]) {
      set_error("context %d: %s", c, texts[(size_t)c].c_str());
      return rcs[(size_t)c];
    }
  return GPD_OK;
}

// ONE cloud, its samples cut into contiguous ranges, one range per context (sample-range sharding with the cloud
// replicated; BASELINE configs[3] across GPUs).  The reference draws every shadow point of a cloud from ONE LCG stream
// (hand_set.cpp:268-283), hand set after hand set, so range g has to start where the ranges before it stopped:
//   phase 1  every context uploads the cloud and searches + plans its range; the plan summary carries the range's draw total
//   host     exclusive scan of the G totals (G numbers: still no collective)
//   phase 2  images + LeNet + records with lcg_base = the draws before the range
// The concatenated results are byte for byte those of one gpd_hip_detect_select over all samples (tests/test_gpu_resident.py).
int gpd_hip_detect_sharded(gpd_hip_ctx *const *ctxs, int num_ctx, gpd_detect_job *shards) {
  if (!ctxs || num_ctx < 1 || !shards) {
    set_error("gpd_hip_detect_sharded: bad argument");
    return GPD_ERR_INVALID;
  }
  for (int c = 0; c < num_ctx; c++) {
    if (!ctxs[c]) {
      set_error("gpd_hip_detect_sharded: context %d is null", c);
      return GPD_ERR_INVALID;
    }
    for (int d = 0; d < c; d++)
      if (ctxs[d] == ctxs[c]) {
        set_error("gpd_hip_detect_sharded: context %d is listed twice (a context serves one thread)", c);
        return GPD_ERR_INVALID;
      }
    const gpd_detect_job &j = shards[c];
    if (!j.xyz || !j.normals || !j.cam_source || !j.view_points || (j.num_samples > 0 && (!j.sample_indices || !j.hands)) || j.num_points < 1 ||
        j.num_cams < 1 || j.num_samples < 0 || j.hands_capacity < 0 || j.num_selected != 0) {
      set_error("gpd_hip_detect_sharded: shard %d: bad argument (selectGrasps over a sharded cloud is the caller's: num_selected must be 0)", c);
      return GPD_ERR_INVALID;
    }
    if (!ctxs[c]->lenet.channels) {
      set_error("gpd_hip_detect_sharded: context %d: LeNet weights not set", c);
      return GPD_ERR_STATE;
    }
  }
  std::vector<Job> J((size_t)num_ctx);
  std::vector<int> rcs((size_t)num_ctx, GPD_OK);
  std::vector<std::string> texts((size_t)num_ctx);
  auto run = [&](auto &&body) {
    std::vector<std::thread> threads;
    for (int c = 0; c < num_ctx; c++)
      threads.emplace_back([&, c]() {
        if (rcs[(size_t)c]) return;
        (void)gpd_hip_bind_host_thread(ctxs[c]->device, nullptr);
        int rc = hipSetDevice(ctxs[c]->device) == hipSuccess ? GPD_OK : GPD_ERR_HIP;
        if (!rc) rc = body(c);
        if (rc) {
          rcs[(size_t)c] = rc;
          texts[(size_t)c] = g_err;  // the error text is per thread
          (void)hipStreamSynchronize(ctxs[c]->lane[0].stream);
        }
      });
    for (auto &t : threads) t.join();
  };
  run([&](int c) -> int {
    gpd_hip_ctx *ctx = ctxs[c];
    gpd_detect_job &j = shards[c];
    j.status = GPD_OK;
    j.num_sets = j.num_candidates = j.num_hands = 0;
    int rc = lane_init(ctx->lane[0]);
    if (rc) return rc;
    Lane &L = ctx->lane[0];
    rc = check_samples(ctx, L, "gpd_hip_detect_sharded", j.sample_indices, nullptr, j.num_samples, j.num_points);
    if (!rc) rc = cloud_upload(L.cloud, j.xyz, j.normals, j.num_points, j.cam_source, j.num_cams, j.view_points, L.stream, /*sync=*/false);
    if (rc) return rc;
    Job &jb = J[(size_t)c];
    jb.sample_idx = j.sample_indices;
    jb.S = j.num_samples;
    jb.mode = 1;
    jb.num_selected = 0;
    jb.hands = j.hands;
    jb.capacity = j.hands_capacity;
    rc = job_begin(ctx, L, jb);
    if (!rc) rc = job_wait_plan(ctx, L, jb);
    return rc;
  });
  unsigned long long base = 0;
  for (int c = 0; c < num_ctx; c++) {
    J[(size_t)c].lcg_base = shards[c].lcg_base = base;
    shards[c].lcg_draws = J[(size_t)c].lcg_draws;
    base += J[(size_t)c].lcg_draws;
  }
  run([&](int c) -> int {
    gpd_hip_ctx *ctx = ctxs[c];
    Lane &L = ctx->lane[0];
    Job &jb = J[(size_t)c];
    int rc = job_enqueue(ctx, L, jb);
    if (!rc) rc = job_end(ctx, L, jb);
    if (rc) return rc;
    shards[c].num_sets = jb.num_sets;
    shards[c].num_candidates = jb.num_candidates;
    shards[c].num_hands = jb.num_hands;
    for (int k = 0; k < 3; k++) shards[c].stage_ms[k] = L.stage_ms[k];
    return GPD_OK;
  });
  for (int c = 0; c < num_ctx; c++)
    if (rcs[(size_t)c]) {
      shards[c].status = rcs[(size_t)c];
      set_error("context %d: %s", c, texts[(size_t)c].c_str());
      return rcs[(size_t)c];
    }
  // a record's set_index counts the hand sets of the whole cloud: the sets of the ranges before are added
  int sets_before = 0;
  for (int c = 0; c < num_ctx; c++) {
    if (sets_before)
      for (int i = 0; i < shards[c].num_hands; i++) shards[c].hands[i].set_index += sets_before;
    sets_before += shards[c].num_sets;
  }
  return GPD_OK;
}

int gpd_hip_replay(gpd_hip_ctx *ctx, int stages) {
  StageRange range_("gpd:replay (images + lenet on the resident list)");
  if (!ctx || !(stages & 3) || (stages & ~3)) {
    set_error("gpd_hip_replay: bad argument");
    return GPD_ERR_INVALID;
  }
  Lane &L = ctx->lane[0];
  if (L.images.num_candidates <= 0 || !L.images.d_images || L.search.num_samples == 0) {
    set_error("gpd_hip_replay: no candidate list on the device (call gpd_hip_images / gpd_hip_detect first)");
    return GPD_ERR_STATE;
  }
  if ((stages & 2) && !ctx->lenet.channels) {
    set_error("gpd_hip_replay: LeNet weights not set");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const int n = L.images.num_candidates;
  int rc = reserve_scores(L, n);
  if (rc) return rc;
  while (ctx->replay_events.size() < ctx->replay_used + 6) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    ctx->replay_events.push_back(e);
  }
  hipEvent_t *ev = &ctx->replay_events[ctx->replay_used];
  ctx->replay_used += 6;
  static const bool pipe = prof_env("GPD_REPLAY_PIPE") && atoi(prof_env("GPD_REPLAY_PIPE")) > 0;
  if (pipe && stages == 3) {
    const size_t bytes = (size_t)L.images.capacity * L.images.channels * 3600;
    if (!ctx->pipe_stream) {
      HIP_TRY(hipStreamCreate(&ctx->pipe_stream));
      for (int b = 0; b < 2; b++) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->pipe_filled[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->pipe_read[b], hipEventDisableTiming));
      }
    }
    if (ctx->pipe_bytes != bytes || ctx->pipe_images[0] != L.images.d_images) {
      // (re)start: the list was rebuilt since; lane 0's buffer is [0], a second one of the same size is [1]
      HIP_TRY(hipStreamSynchronize(ctx->pipe_stream));
      if (ctx->pipe_images[1]) HIP_TRY(hipFree(ctx->pipe_images[1]));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->pipe_images[1]), bytes));
      ctx->pipe_images[0] = L.images.d_images;
      ctx->pipe_bytes = bytes;
      ctx->pipe_read_valid[0] = ctx->pipe_read_valid[1] = false;
      ctx->pipe_k = 0;
    }
    const int b = (int)(ctx->pipe_k++ & 1);
    if (ctx->pipe_read_valid[b]) HIP_TRY(hipStreamWaitEvent(L.stream, ctx->pipe_read[b], 0));  // LeNet of replay k - 2 has read it
    HIP_TRY(hipEventRecord(ev[0], L.stream));
    uint8_t *own = L.images.d_images;
    L.images.d_images = ctx->pipe_images[b];
    rc = images_launch(L.search, L.plan, L.images, L.stream);
    L.images.d_images = own;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev[1], L.stream));
    HIP_TRY(hipEventRecord(ctx->pipe_filled[b], L.stream));
    HIP_TRY(hipStreamWaitEvent(ctx->pipe_stream, ctx->pipe_filled[b], 0));
    HIP_TRY(lenet_forward(ctx->lenet, L.lenet_scratch, ctx->pipe_images[b], n, L.d_scores, ctx->pipe_stream, ev + 2));
    HIP_TRY(hipEventRecord(ev[5], ctx->pipe_stream));
    HIP_TRY(hipEventRecord(ctx->pipe_read[b], ctx->pipe_stream));
    ctx->pipe_read_valid[b] = true;
    return GPD_OK;
  }
  HIP_TRY(hipEventRecord(ev[0], L.stream));
  if (stages & 1) {
    rc = images_launch(L.search, L.plan, L.images, L.stream);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(ev[1], L.stream));
  if (stages & 2) {
    HIP_TRY(lenet_forward(ctx->lenet, L.lenet_scratch, L.images.d_images, n, L.d_scores, L.stream, ev + 2));
  } else {
    for (int i = 2; i < 5; i++) HIP_TRY(hipEventRecord(ev[i], L.stream));
  }
  HIP_TRY(hipEventRecord(ev[5], L.stream));
  return GPD_OK;
}

int gpd_hip_replay_times(gpd_hip_ctx *ctx, float ms[2], int *launches, float *scores) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  HIP_TRY(hipStreamSynchronize(L.stream));
  if (ctx->pipe_stream) HIP_TRY(hipStreamSynchronize(ctx->pipe_stream));
  ms[0] = ms[1] = 0.f;
  for (int k = 0; k < 4; k++) ctx->replay_kernel_ms[k] = 0.f;
  for (size_t i = 0; i + 5 < ctx->replay_used; i += 6) {
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, ctx->replay_events[i], ctx->replay_events[i + 1]));
    HIP_TRY(hipEventElapsedTime(&b, ctx->replay_events[i + 1], ctx->replay_events[i + 5]));
    ms[0] += a;
    ms[1] += b;
    for (int k = 0; k < 4; k++) {
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, ctx->replay_events[i + 1 + k], ctx->replay_events[i + 2 + k]));
      ctx->replay_kernel_ms[k] += t;
    }
  }
  if (launches) *launches = (int)(ctx->replay_used / 6);
  ctx->replay_used = 0;
  if (scores && L.images.num_candidates > 0 && L.d_scores)
    HIP_TRY(hipMemcpy(scores, L.d_scores, (size_t)L.images.num_candidates * sizeof(float), hipMemcpyDeviceToHost));
  int32_t status = 0;
  if (L.images.d_status) HIP_TRY(hipMemcpy(&status, L.images.d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (status) {
    images_status_text(status, g_err, sizeof(g_err));
    return GPD_ERR_CAPACITY;
  }
  return lenet_check(L.lenet_scratch);
}

int gpd_hip_conv1_stats(gpd_hip_ctx *ctx, unsigned long long pairs[2], int reset) {
  if (!ctx || !pairs) {
    set_error("gpd_hip_conv1_stats: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  LeNetScratch &s = ctx->lane[0].lenet_scratch;
  pairs[0] = pairs[1] = 0;
  if (!s.c1_stats) return GPD_OK;
  HIP_TRY(hipStreamSynchronize(ctx->lane[0].stream));
  HIP_TRY(hipMemcpy(pairs, s.c1_stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) HIP_TRY(hipMemset(s.c1_stats, 0, 2 * sizeof(unsigned long long)));
  return GPD_OK;
}

int gpd_hip_replay_kernel_ms(gpd_hip_ctx *ctx, float ms[4]) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  for (int k = 0; k < 4; k++) ms[k] = ctx->replay_kernel_ms[k];
  return GPD_OK;
}

int gpd_hip_last_images_stats(gpd_hip_ctx *ctx, long long out[4]) {
  if (!ctx || !out) return GPD_ERR_INVALID;
  const Lane &L = ctx->lane[0];
  out[0] = L.images.num_candidates;
  out[1] = L.images.stat_sets;
  out[2] = L.images.stat_sum_set_ni;
  out[3] = L.images.stat_sum_cand_ni;
  return GPD_OK;
}

int gpd_hip_last_fallbacks(gpd_hip_ctx *ctx, long long out[4]) {
  if (!ctx || !out) return GPD_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  HIP_TRY(hipStreamSynchronize(L.stream));
  out[0] = L.search.nn_cap;
  out[1] = out[2] = 0;
  const ImageState &im = L.images;
  int32_t v = 0;
  if (im.d_overflow && im.channels == 15 && im.num_candidates > 0) {
    HIP_TRY(hipMemcpy(&v, im.d_status + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
    out[1] = v;
  }
  if (im.d_pts_overflow && im.num_candidates > 0) {
    HIP_TRY(hipMemcpy(&v, im.d_status + 3, sizeof(int32_t), hipMemcpyDeviceToHost));
    out[2] = v;
  }
  out[3] = im.num_candidates > 0 ? (im.num_candidates + 65535) / 65536 : 0;
  return GPD_OK;
}

int gpd_hip_last_centre_chains(gpd_hip_ctx *ctx, long long *out) {
  if (!ctx || !out) return GPD_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  Lane &L = ctx->lane[0];
  HIP_TRY(hipStreamSynchronize(L.stream));
  *out = 0;
  const int S = L.search.num_samples;
  if (S <= 0 || !L.search.d_counts) return GPD_OK;
  std::vector<int32_t> h((size_t)S * 8);
  HIP_TRY(hipMemcpy(h.data(), L.search.d_counts, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  long long n = 0;
  for (int i = 0; i < S; i++) n += __builtin_popcount((unsigned)h[(size_t)8 * i + 5] & 7u);
  *out = n;
  return GPD_OK;
}

int gpd_hip_last_stage_ms(gpd_hip_ctx *ctx, float ms[3]) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  for (int i = 0; i < 3; i++) ms[i] = ctx->lane[0].stage_ms[i];
  return GPD_OK;
}

}  // extern "C"
