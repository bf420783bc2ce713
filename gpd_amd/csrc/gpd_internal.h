// Internal declarations shared by the HIP translation units of libgpd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include <vector>

#ifdef GPD_PROFILING
#include <cstdlib>
#include <rocprofiler-sdk-roctx/roctx.h>
#endif

#include "../../include/gpd_hip.h"

namespace gpd {

constexpr int kImg = 60;            // image_size (eigen_classifier.cpp:12)
constexpr int kPix = kImg * kImg;   // 3600
constexpr int kFc1In = 7200;        // 50 * 12 * 12
constexpr int kFc1Out = 500;
// split path: where ip1 leaves the partial sum of (image m, K quarter kq, unit u) for ip2's kernel (floats)
__host__ __device__ inline size_t fc1p_index(int m, int kq, int u) { return (((size_t)(m >> 5) * 4 + kq) * kFc1Out + u) * 32 + (m & 31); }
constexpr int kLenetXld = 7296;     // row length of the split path's flat bf16 planes: 7200 + 96 zeros = 4 K quarters x 57 steps of 32 (lenet_fast.hip)

// ---- LeNet (lenet.hip) ----------------------------------------------------
// operand tables of the split path (lenet_fast.hip): conv1 as four int8 digit planes of 32-bit fixed-point weights, conv2 / ip1
// as three bf16 pieces per weight, all in the kernels' MFMA fragment layouts
struct LeNetFast {
  uint4 *c1a = nullptr;            // conv1 A fragments [7 k-steps][5 row tiles][64 lanes] x 16 int8
  double *c1corr = nullptr;        // [20] 128 * sum of the filter's fixed-point weights (the x - 128 shift of the inputs)
  int *c1shift = nullptr;          // [20] fixed-point position s of the filter: value = integer * 2^-s
  uint4 *c2b = nullptr;            // conv2 fragments [4 slots][3 pieces][16 k-steps][64 lanes] x 8 bf16 (slot 3: the A fragments of filters 48, 49)
  unsigned short *f1wt = nullptr;  // ip1's weights as bf16 pieces, blocked [32 unit blocks][228 k steps][3 pieces][16 units][32 k] (lenet_fast.hip f3_blocked)
};
void lenet_fast_free(LeNetFast &f);
void lenet_fast_unblock_x(const unsigned short *blocked, int n, unsigned short *planes);  // test hook: [3][n][7200] from the blocked X
hipError_t lenet_fast_prepare(LeNetFast &f, int channels, const float *c1w, const float *c2w, const float *f1w);

struct LeNetWeights {
  int channels = 0;
  int mode = GPD_LENET_SPLIT;  // gpd_hip_set_lenet_mode
  LeNetFast fast;
  float *c1w = nullptr, *c1b = nullptr, *c2w = nullptr, *c2b = nullptr;
  float *c1wp = nullptr;  // conv1 weights padded to [20][C][28] (25 taps + 3 zeros): 16-byte rows for the LDS table
  float *c2wt = nullptr;  // conv2 weights k-major [K][F] for the implicit-GEMM A operand
  float *f1w = nullptr, *f1b = nullptr, *f2w = nullptr, *f2b = nullptr;
};

struct LeNetScratch {
  int capacity = 0;        // images per chunk
  float *pool1 = nullptr;  // f32 chain: [cap][20][784], planes in conv1's chunk order (whole-line stores, lenet.hip P1_PLANE);
                           // split path: [cap][784][20], pixel-major
  float *flat = nullptr;   // [cap][7200]  (pixel-major, channel-minor)
  unsigned short *xs = nullptr;  // split path: flat as three bf16 pieces, blocked like ip1's weights [cap / 16][228][3][16][32] (k >= 7200: zeros, written once at allocation)
  float *fc1t = nullptr;   // [500][cap]   (transposed, ReLU applied)
  float *fc1p = nullptr;   // split path: ip1's partial sums over the four K quarters, blocked [cap / 32][4][500][32] (fc1p_index)
  int num_cus = 0;         // compute units of the context's device (grid of the persistent conv2)
  unsigned long long *c1_stats = nullptr;  // device: [0] (chunk, channel) pairs conv1 executed, [1] pairs it looked at — summed
                                           // over its launches since the last gpd_hip_conv1_stats(reset); [2] != 0: a launch gave up
                                           // waiting for an image slot (its scores are invalid: lenet_check)
};

// Scores n images (device pointer, planar u8 [n][C][3600]) into d_scores (device). Async on stream.
hipError_t lenet_forward(const LeNetWeights &w, LeNetScratch &s, const uint8_t *d_images, int n, float *d_scores,
                         hipStream_t stream, hipEvent_t *kernel_events = nullptr);
// the split path's three matrix kernels (lenet_fast.hip); `queue`: two zeroed image counters (conv1's, conv2's)
hipError_t lenet_forward_fast(const LeNetWeights &w, LeNetScratch &s, const uint8_t *img, int m, float *d_scores, hipStream_t stream,
                              hipEvent_t *kernel_events, int *queue);
hipError_t lenet_scratch_reserve(LeNetScratch &s, int n);
// after the stream was synchronised: GPD_ERR_HIP (and the error word cleared) when a conv1 launch gave up on its slot protocol
int lenet_check(LeNetScratch &s);
void lenet_scratch_free(LeNetScratch &s);

void set_error(const char *fmt, ...);
// every growth of a device / pinned buffer (hipFree + hipMalloc: a device stall) is counted per host thread; the batch
// entry reports the count per cloud (gpd_detect_job::allocs) so that a pass that was supposed to run on pre-sized lanes
// can be seen to have done so
void note_alloc(const char *where);

// roctx range around a stage of the path (host side: what the stage enqueues); `rocprofv3 --marker-trace --kernel-trace`
// shows them over the kernel timeline (the reference times its stages with omp_get_wtime, grasp_detector.cpp:223-273)
// — in the profiling build only (make prof -> libgpd_hip_prof.so, -DGPD_PROFILING).  The release library links no
// profiler and reads no environment variable: the measurement switches (GPD_*_TIMING, GPD_IMG_SERIAL, GPD_IMG_EXIT,
// GPD_IMG_LDS_PAD, GPD_REPLAY_PIPE, GPD_DETECT_TIMING, GPD_PLAN_TIMING) and the watchdog's fault injector
// (GPD_C1_FAULT) go through prof_env(), which is a constant nullptr there.
#ifdef GPD_PROFILING
struct StageRange {
  explicit StageRange(const char *name) { roctxRangePushA(name); }
  ~StageRange() { roctxRangePop(); }
  StageRange(const StageRange &) = delete;
  StageRange &operator=(const StageRange &) = delete;
};
inline const char *prof_env(const char *name) { return getenv(name); }
#else
struct StageRange {
  explicit StageRange(const char *) {}
  StageRange(const StageRange &) = delete;
  StageRange &operator=(const StageRange &) = delete;
};
inline const char *prof_env(const char *) { return nullptr; }
#endif

// ---- point-cloud preparation (preprocess.hip): workspace cut + the reference's voxeliser ------------------------
struct PreState {
  int capacity = 0, cap_cams = 0;
  float *d_xyz = nullptr, *d_block_lo = nullptr, *d_out_xyz = nullptr;
  int32_t *d_cam = nullptr, *d_block_count = nullptr, *d_block_off = nullptr, *d_src = nullptr, *d_rank = nullptr, *d_out_cam = nullptr,
          *d_out_src = nullptr;
  int4 *d_keys = nullptr;
  void *d_meta = nullptr;
  char *h_pin = nullptr;        // pinned host side of the voxeliser's chain: the voxel keys of every point (16 B), what the walk
                                // decided (rank among the kept points, -1: dropped; 4 B), the counters
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipEvent_t ev_keys = nullptr; // the keys and counters have arrived in h_pin
  int n = 0, num_cams = 0, M = 0;  // the call in flight between preprocess_begin and preprocess_finish; M: points it left on the device
  float cell = 0.f;
};
void preprocess_free(PreState &s);
// the two halves of preprocess_run (gpd_hip_detect_batch on raw scans puts other clouds' work between them): begin enqueues the
// upload, the workspace cut, the voxel keys and their way back to pinned memory; finish waits for them, walks the voxeliser's
// chain on the host and gathers the kept points: s.M of them in s.d_out_xyz [M][3] / s.d_out_cam [cams][M] on the device.
// drop_nonfinite: Cloud::removeNans first (cloud.cpp:154-164: a NaN / Inf point is dropped); without it such a cloud is refused
int preprocess_begin(PreState &s, const float *xyz, const int32_t *cam_source, int n, int num_cams, const double *workspace, float cell,
                     hipStream_t stream, bool drop_nonfinite);
int preprocess_finish(PreState &s, hipStream_t stream);
// workspace: 6 doubles or nullptr; cell <= 0: no voxeliser.  src_out (may be nullptr): input index of every output point.
// ms (may be nullptr): device time of the kernels.
int preprocess_run(PreState &s, const float *xyz, const int32_t *cam_source, int n, int num_cams, const double *workspace, float cell,
                   float *xyz_out, int32_t *cam_out, int32_t *src_out, int *num_out, float *ms, hipStream_t stream);

// ---- Clustering::findClusters on the device (cluster.hip) ---------------------------------------------------------
struct ClusterState {
  int capacity = 0;
  gpd_hand *d_hands = nullptr, *d_out = nullptr;
  double *d_scores = nullptr, *d_res = nullptr, *d_out_scores = nullptr;
  uint8_t *d_used = nullptr;
  int32_t *d_keep = nullptr, *d_out_src = nullptr, *d_num = nullptr;
};
void cluster_free(ClusterState &s);
int cluster_run(ClusterState &s, const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers, gpd_hand *out,
                double *out_scores, int32_t *out_src, int *num_out, hipStream_t stream);

// ---- Cloud (search.hip) -----------------------------------------------------
// Device copy of what the path reads from util::Cloud, as SoA for coalesced streaming.
// cameras of a cloud: the kernels carry "which cameras see this neighbourhood" as a 32-bit mask and the view points
// as a kernel argument (768 bytes)
constexpr int kMaxCams = 32;
// largest neighbourhood the search lists per sample (the reference has no limit, hand_search.cpp:178; ours is a matter of
// memory: 44 bytes per list entry and sample).  Beyond 65535 entries hand_eval_kernel walks the full list instead of its
// LDS table, whose neighbour ranks are 16 bits wide.
constexpr int kNnCapMax = (1 << 20) - 1;
// scratch of Cloud::calculateNormals on the device (search.hip normals_run): per-point neighbour lists in one array
struct NormalsScratch {
  int cap_points = 0;
  long long arena_cap = 0;         // 8-byte units
  int32_t *d_count = nullptr, *d_big = nullptr;
  float4 *d_lists = nullptr;        // [P / 64][1024][64] transposed neighbour lists
  long long *d_big_off = nullptr;
  unsigned long long *d_arena = nullptr, *d_ctl = nullptr;
  float *d_out = nullptr;
  double *d_cent = nullptr;         // [P][3] centroids certified by the list kernel (NaN: walk the chain)
  int last_queued = 0;             // of the last run: points that went through the large-neighbourhood kernel
};
void normals_free(NormalsScratch &s);
struct Cloud {
  int num_points = 0, num_cams = 0, capacity = 0, cap_cams = 0;
  char *h_pin = nullptr;              // pinned staging of the caller's arrays (xyz, normals, cam_source)
  uint64_t generation = 0;            // bumped by every upload
  float *px = nullptr, *py = nullptr, *pz = nullptr;
  float *nx = nullptr, *ny = nullptr, *nz = nullptr;
  int32_t *cam_source = nullptr;      // [num_cams][num_points]
  float *staging = nullptr;           // AoS upload buffer, 6 floats per point
  double view_points[3 * kMaxCams] = {0};
  // uniform grid over the cloud (cells of 2 cm, z fastest): the radius searches visit the cells
  // that overlap the query sphere instead of streaming all P points (replaces the k-d tree of
  // hand_search.cpp:29-31 / image_generator.cpp:37-38; only a candidate filter — the distance
  // test and the (d2, index) order are unchanged)
  float g_lo[3] = {0, 0, 0};
  float g_cell = 0.02f;
  int g_dim[3] = {1, 1, 1};
  int g_cells_cap = 0;
  int32_t *g_start = nullptr;         // [cells + 1]
  int32_t *g_cursor = nullptr;        // [cells] scatter cursors
  float4 *g_p = nullptr;              // [P] the points in cell order: x, y, z and, as bits in w, the original index (one
                                      // 16-byte load per visited point instead of four dword loads)
  float4 *pxyz = nullptr, *pnrm = nullptr;  // [P] AoS copies (x, y, z, 0) / (nx, ny, nz, 0): one load per random access
  NormalsScratch normals;
};
struct GridView {
  float lo[3];
  float cell;
  int dim[3];
  const int32_t *start;
  const float4 *p;  // x, y, z, index bits
};
inline GridView grid_view(const Cloud &c) {
  GridView g;
  for (int i = 0; i < 3; i++) {
    g.lo[i] = c.g_lo[i];
    g.dim[i] = c.g_dim[i];
  }
  g.cell = c.g_cell;
  g.start = c.g_start;
  g.p = c.g_p;
  return g;
}
int cloud_upload(Cloud &c, const float *xyz, const float *normals, int n, const int32_t *cam_source, int num_cams,
                 const double *view_points, hipStream_t stream, bool sync);
int cloud_reserve(Cloud &c, int n, int num_cams);
// the cloud from device arrays (the preprocessing kernels' output): d_xyz [n][3], d_cam [cams][n]; normals zero until normals_run
int cloud_from_device(Cloud &c, const float *d_xyz, const int32_t *d_cam, int n, int num_cams, const double *view_points, hipStream_t stream);
int cloud_reserve_grid(Cloud &c, int cells);
void cloud_free(Cloud &c);
int normals_run(Cloud &c, double radius, float *normals_out, hipStream_t stream);

// ---- Candidate search (search.hip) -----------------------------------------------
struct SearchState {
  int num_samples = 0, capacity_samples = 0;
  int min_samples = 0;                // sample capacity to keep across a change of the list capacity (search_force_capacity)
  uint64_t seen_generation = 0;       // Cloud::generation of the last run: lists beyond the LDS capacities shrink back on a new cloud
  int nn_cap = 0;                     // entries per neighbourhood list: 8192 / 16384 (LDS sorts) or up to kNnCapMax (global-memory sort)
  uint64_t cloud_generation = 0;
  int32_t *d_sample_idx = nullptr;    // [S]
  double *d_sample_xyz = nullptr;     // [S][3] samples given by coordinates (used instead of the indices)
  int32_t *d_counts = nullptr;        // [S][8]: N_hands, N_images, k_frames, total found, seen by camera 0
  int32_t *d_nn_idx = nullptr;        // [S][nn_cap] scratch of the global-memory lists (neighbourhoods beyond the LDS capacities)
  float *d_nn = nullptr;              // [S][6][nn_cap] gathered px,py,pz,nx,ny,nz in that order
  double *d_frames = nullptr;         // [S][12] sample, normal, binormal, curvature
  double *d_centers = nullptr;        // [S][3] mean of the image neighbourhood
  gpd_hand *d_hands = nullptr;        // [S][slots]
  float4 *d_hl = nullptr;             // [S][cap] points that can be in-height for some orientation: x, y, z, rank bits (height_list_kernel)
  uint8_t *d_fvalid = nullptr;        // [S][slots] is_valid after filterGraspsWorkspace (hand_eval_kernel writes it; the
                                      // unfused gpd_hip_images overwrites it with the caller's flags)
  // centre_kernel (serial fp64 chains: a hundred-odd waves, latency-bound) runs on a side stream beside
  // hand_eval_kernel and is joined before anything reads d_centers
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<int32_t> h_counts;      // host copy of d_counts (unfused entry points only)
  std::vector<int32_t> h_set_sample;  // set -> sample slot
  std::vector<double> h_samples;      // [S][3] sample coordinates (to validate hands passed to images)
};
// samples by index (sample_xyz == nullptr) or by coordinates (sample_idx == nullptr).  sync_counts: download the
// neighbourhood sizes and grow the list capacity when one overflows (the unfused entry points); without it nothing
// waits for the device and the caller checks `worst found` in the plan summary (search_retry_needed).
int search_run(const gpd_params &p, const Cloud &c, SearchState &s, const int32_t *sample_idx, const double *sample_xyz, int S,
               hipStream_t stream, bool sync_counts = true);
// list capacity the next search_run would use after a neighbourhood of `worst` entries; 0: beyond every capacity
int search_reserve_samples(SearchState &s, int S, int slots);  // buffers for S samples at the current list capacity
int search_next_capacity(const SearchState &s, int worst);
int search_force_capacity(SearchState &s, int cap);
// HandSearch::reevaluateHypotheses on the uploaded cloud; invalidates the search state
int reevaluate_run(const gpd_params &p, const Cloud &c, SearchState &s, gpd_hand *hands, int n, int32_t *labels, hipStream_t stream);
int search_download(const gpd_params &p, SearchState &s, gpd_hand *hands, int *num_sets, hipStream_t stream);
void search_free(SearchState &s);

// GraspDetector::filterGraspsWorkspace (grasp_detector.cpp:334-398) [+ filterGraspsDirection, :423-456] for one valid
// hand: aperture and the workspace box around the hand's outline [, then the approach direction].  The reference computes right_top from left_bottom
// (:360-363); kept.  Evaluated with the same unfused fp64 operations on the host and on the device.
struct FilterConsts {
  double min_aperture, max_aperture, half_width, hand_depth;
  double workspace[6];
  // filterGraspsDirection (grasp_detector.cpp:423-456): a hand is dropped when acos(direction . approach) > thresh_rad.
  // The device has no glibc acos, and a one-ulp difference between two acos implementations would be a different
  // candidate list; so the HOST turns the test into one on the dot product itself: dot_drop_max = the largest double x
  // in [-1, 1] with acos(x) > thresh_rad by the host's libm (bisection over the doubles; acos is monotone), and a hand is
  // dropped iff -1 <= dot <= dot_drop_max.  A dot product outside [-1, 1] (acos = NaN) keeps the hand, as the reference does.
  int dir_on;
  double dir[3];
  double dot_drop_max;
};
FilterConsts filter_consts(const gpd_params &p);  // host_math.cpp
__host__ __device__ inline bool workspace_ok(const FilterConsts &f, const gpd_hand &h) {
  bool ok = h.grasp_width >= f.min_aperture && h.grasp_width <= f.max_aperture;
  for (int r = 0; r < 3 && ok; r++) {
    const double bin = h.frame[3 * r + 1], app = h.frame[3 * r + 0];
    const double lb = h.position[r] + f.half_width * bin;
    const double rb = h.position[r] - f.half_width * bin;
    const double lt = lb + f.hand_depth * app;
    const double rt = lb + f.hand_depth * app;
    const double ap = h.position[r] - 0.05 * app;
    const double mn = fmin(fmin(fmin(lb, rb), fmin(lt, rt)), ap);
    const double mx = fmax(fmax(fmax(lb, rb), fmax(lt, rt)), ap);
    ok = mn >= f.workspace[2 * r] && mx <= f.workspace[2 * r + 1];
  }
  if (ok && f.dir_on) {
    const double dot = f.dir[0] * h.frame[0] + f.dir[1] * h.frame[3] + f.dir[2] * h.frame[6];  // direction^T * approach, unfused
    ok = !(dot >= -1.0 && dot <= f.dot_drop_max);
  }
  return ok;
}
void filter_workspace_host(const gpd_params &p, gpd_hand *hands, int num_sets);

// ---- Candidate plan (plan.hip) -----------------------------------------------------
// What GraspDetector::detectGrasps does between its stages on the host — filterGraspsWorkspace
// dropping hand sets (grasp_detector.cpp:238, 334-398), createImageList's set-major / slot-minor
// gather of the valid hands (image_generator.cpp:91-98), the per-set shadow LCG offsets
// (hand_set.cpp:263-266 is one global stream) and the score write-back (:269-273) — as device
// tables, built by one plan_kernel launch from the search results.
struct PlanSummary {  // copied to the host once per call (pinned)
  int32_t num_sets;          // samples with a frame neighbourhood (frame_estimator.cpp:24-29)
  int32_t num_candidates;    // valid hands after the filter
  int32_t num_shadow_sets;   // (live set, camera) voxel bitsets
  int32_t worst_found;       // largest neighbourhood found by the search (list capacity check)
  int32_t live_sets;         // sets with at least one candidate
  int32_t mismatch_set;      // gpd_hip_images: first set whose sample differs from the search's (or that the search does
                             // not have), -1: none
  int32_t pad_[2];
  long long sum_set_ni, sum_cand_ni;  // for the algorithmic byte count of SURVEY §8d
  unsigned long long total_draws;     // shadow LCG draws of all hand sets of the call (the next sample range's lcg_base)
};
struct PlanPart {  // what one workgroup of plan_kernel publishes for the workgroups after it (64 bytes)
  int32_t sum[3];   // hand sets, candidates, shadow bitsets of its samples
  int32_t worst;
  unsigned long long draws;  // LCG draws of its samples
  long long sum_set_ni, sum_cand_ni;
  int32_t live, mismatch;
  int32_t sets;          // caller flags: hand sets of its samples, published first
  unsigned ready_sets;   // stamps: the launch number once `sets` / the rest is complete
  unsigned ready;
  unsigned pad_;
};
static_assert(sizeof(PlanPart) == 64, "PlanPart");
struct Plan {
  int cap_samples = 0, cap_slots = 0, cap_cams = 0;
  PlanPart *d_parts = nullptr;   // [cap_samples / 256 + 1]
  unsigned *d_ticket = nullptr;  // arrival counter of plan_kernel's workgroups (0 between launches: the kernel resets it)
  unsigned epoch = 0;
  int32_t *d_sample_of_set = nullptr;  // [S]
  int32_t *d_hand_cand = nullptr;      // [S][slots] candidate ordinal of a hand, -1: none
  int32_t *d_cand_hand = nullptr;      // [S*slots] hand (sample slot * slots + slot) of a candidate
  int32_t *d_cand_out = nullptr;       // [S*slots] index of the candidate in the set-major hand array handed to the caller
  int32_t *d_cand_meta = nullptr;      // [S*slots][4]: sample slot, N_images, first shadow bitset (< 0: none), number of bitsets
  int32_t *d_set_meta = nullptr;       // [S*cams][8]: sample slot, N_images, lcg offset lo, hi, camera
  PlanSummary *d_summary = nullptr;
  PlanSummary *h_summary = nullptr;    // pinned
  uint8_t *d_set_flags = nullptr;      // gpd_hip_images: the caller's is_valid flags, [sets][slots] ...
  double *d_set_samples = nullptr;     // ... and the samples of its hand sets, [sets][3]
  int cap_set_flags = 0;
};
void plan_free(Plan &pl);
int plan_reserve(Plan &pl, int want_samples, int slots, int cams, hipStream_t stream);
// Enqueues plan_kernel + the summary copy on `stream`; the caller waits for the stream before reading pl.h_summary.
// set_flags == nullptr: the validity flags are the ones hand_eval_kernel left (search + workspace filter).
// Otherwise (gpd_hip_images) the caller's flags, set-major [num_sets_given][slots], and the samples of its sets
// [num_sets_given][3] (host pointers, copied asynchronously: keep them alive until the stream has been waited for).
int plan_build(const gpd_params &p, const Cloud &c, const SearchState &s, Plan &pl, hipStream_t stream,
               const uint8_t *set_flags = nullptr, const double *set_samples = nullptr, int num_sets_given = 0);
// set-major hand records of the last search with `valid` = the filtered flag and the scores written back
// (out: [num_sets][slots], device); candidates_only: the scored valid hands in candidate order instead.
int plan_emit_hands(const gpd_params &p, const SearchState &s, const Plan &pl, const float *d_scores, gpd_hand *d_out,
                    bool candidates_only, hipStream_t stream);
// GraspDetector::selectGrasps (grasp_detector.cpp:405-420) over the device score array: the k best candidates,
// score descending; *tie: equal scores among them or at the cut (the caller then falls back to std::partial_sort)
int select_topk(const float *d_scores, int n, int k, int32_t *d_sel /* [k] candidate ordinals */, int32_t *d_tie,
                hipStream_t stream);
int gather_hands(const gpd_params &p, const SearchState &s, const Plan &pl, const float *d_scores, const int32_t *d_sel, int k,
                 gpd_hand *d_out, hipStream_t stream);
// records sel[0..k) of a flat candidate list emitted earlier (plan_emit_hands, candidates only)
int gather_records(const gpd_hand *d_all, const int32_t *d_sel, int k, gpd_hand *d_out, hipStream_t stream);
// the largest k select_topk takes; beyond it the fused entries select with std::partial_sort on the downloaded scores
int select_topk_capacity();

// ---- Grasp images (images.hip) ---------------------------------------------------
struct ImageState {
  unsigned long long lcg_base = 0;    // shadow draws consumed before this list's first hand set (a sample range of a sharded cloud)
  int num_candidates = 0;
  int capacity = 0;                   // images
  uint8_t *d_images = nullptr;        // planar [n][C][60][60]
  uint8_t *d_images_hwc = nullptr;    // [n][60][60][C], only when the caller downloads pixels
  uint32_t *d_set_bits = nullptr;     // [sets][SETWORDS] shadow voxel bitsets
  int num_shadow_sets = 0, cap_shadow_sets = 0;
  bool wide = false;                  // the shadow kernels' wide voxel windows (images.hip Vox<WIDE>), picked from the image volume
  bool huge = false;                  // ... or beyond those: the general shadow kernel, a set region of run-time size
  int set_sd = 0, set_sr = 0;         // edge / radius (voxels) of the region shadow_set_kernel fills around a sample
  size_t cap_setwords = 0;            // words per row of d_set_bits
  int32_t *d_overflow2 = nullptr;     // [capacity]: candidates the large shadow instantiation could not list (count: d_status[2])
  char *d_huge_scratch = nullptr;     // list rows of the general shadow kernel
  int channels = 0;
  int slots = 0;                      // hand slots per set of the list (num_hand_axes * num_orientations)
  int32_t *d_overflow = nullptr;      // [capacity]: candidates for the large shadow instantiation (count: d_status[1])
  int32_t *d_pts_overflow = nullptr;  // [capacity]: candidates for the large points instantiation (count: d_status[3])
  char *d_pts_scratch = nullptr;      // point arrays of the large instantiation, one row per workgroup of its grid
  int32_t *d_status = nullptr;        // [4]: error flags from the kernels, then the counters of the three overflow lists
  long long stat_sets = 0, stat_sum_set_ni = 0, stat_sum_cand_ni = 0;  // for the algorithmic byte count
  std::vector<unsigned char> consts;  // the constant block (image geometry) the candidate list was built with
  double view_points[3 * kMaxCams] = {0};  // of the cloud the list was built on (shadow_set_kernel argument)
  // the normals + depth kernel does not depend on the shadow kernels: it runs on a side stream beside them (both kinds
  // of workgroup fit a CU together) and is joined at the end of the stage
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_stream = true;  // off inside gpd_hip_detect_batch: the other cloud's kernels already fill the gaps (measured)
};
int images_reserve(const gpd_params &p, ImageState &im, int n, int shadow_sets);
// Sizes the image buffers for the plan's candidate list (summary already on the host) and launches the image kernels.
int images_run(const gpd_params &p, const Cloud &c, const SearchState &s, const Plan &pl, ImageState &im, hipStream_t stream);
// Re-launches them on the resident list.  Nothing waits for the device: capacity flags accumulate in im.d_status.
int images_launch(const SearchState &s, const Plan &pl, ImageState &im, hipStream_t stream, bool counters_clean = false);
void images_status_text(int status, char *buf, size_t len);
hipError_t planar_to_hwc(const uint8_t *src, uint8_t *dst, int n, int C, hipStream_t stream);
hipError_t hwc_to_planar(const uint8_t *src, uint8_t *dst, int n, int C, hipStream_t stream);
void image_cell_thresholds(double len, double *out);  // 61 doubles
void images_free(ImageState &im);

// Host-side constants of the path (host_math.cpp), computed with libm exactly as the
// reference computes them on the host.
struct HostConsts {
  double rot[GPD_MAX_SLOTS][9];  // Ry(pi) * R_axis(angle) factors: see host_math.cpp
  double rot_binormal[9];        // AngleAxisd(pi, UnitY)
  double finger_spacing[32];     // 2 * num_finger_placements
  double deepen_depths[128];
  int num_deepen;
  double cos_friction;
  double nn_radius_hands, nn_radius_images;
};
void host_consts(const gpd_params &p, HostConsts &h);

}  // namespace gpd
