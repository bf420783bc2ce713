// Internal declarations shared by the HIP translation units of libgpd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/gpd_hip.h"

namespace gpd {

constexpr int kImg = 60;            // image_size (eigen_classifier.cpp:12)
constexpr int kPix = kImg * kImg;   // 3600
constexpr int kFc1In = 7200;        // 50 * 12 * 12
constexpr int kFc1Out = 500;

// ---- LeNet (lenet.hip) ----------------------------------------------------
struct LeNetWeights {
  int channels = 0;
  float *c1w = nullptr, *c1b = nullptr, *c2w = nullptr, *c2b = nullptr;
  float *c1wp = nullptr;  // conv1 weights padded to [20][C][28] (25 taps + 3 zeros): 16-byte rows for the LDS table
  float *c2wt = nullptr;  // conv2 weights k-major [K][F] for the implicit-GEMM A operand
  float *f1w = nullptr, *f1b = nullptr, *f2w = nullptr, *f2b = nullptr;
};

struct LeNetScratch {
  int capacity = 0;        // images per chunk
  float *pool1 = nullptr;  // [cap][20][28][28]
  float *flat = nullptr;   // [cap][7200]  (pixel-major, channel-minor)
  float *fc1t = nullptr;   // [500][cap]   (transposed, ReLU applied)
  int num_cus = 0;         // compute units of the context's device (grid of the persistent conv2)
};

// Scores n images (device pointer, planar u8 [n][C][3600]) into d_scores (device). Async on stream.
hipError_t lenet_forward(const LeNetWeights &w, LeNetScratch &s, const uint8_t *d_images, int n, float *d_scores,
                         hipStream_t stream, hipEvent_t *kernel_events = nullptr);
hipError_t lenet_scratch_reserve(LeNetScratch &s, int n);
void lenet_scratch_free(LeNetScratch &s);

void set_error(const char *fmt, ...);

// ---- Cloud (search.hip) -----------------------------------------------------
// Device copy of what the path reads from util::Cloud, as SoA for coalesced streaming.
constexpr int kMaxCams = 8;
struct Cloud {
  int num_points = 0, num_cams = 0, capacity = 0;
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
  int32_t *g_idx = nullptr;           // [P] original index of the sorted points
  float *g_x = nullptr, *g_y = nullptr, *g_z = nullptr;  // [P] coordinates in cell order
};
struct GridView {
  float lo[3];
  float cell;
  int dim[3];
  const int32_t *start;
  const int32_t *idx;
  const float *x, *y, *z;
};
inline GridView grid_view(const Cloud &c) {
  GridView g;
  for (int i = 0; i < 3; i++) {
    g.lo[i] = c.g_lo[i];
    g.dim[i] = c.g_dim[i];
  }
  g.cell = c.g_cell;
  g.start = c.g_start;
  g.idx = c.g_idx;
  g.x = c.g_x;
  g.y = c.g_y;
  g.z = c.g_z;
  return g;
}
hipError_t cloud_upload(Cloud &c, const float *xyz, const float *normals, int n, const int32_t *cam_source, int num_cams,
                        const double *view_points, hipStream_t stream);
void cloud_free(Cloud &c);
int normals_run(Cloud &c, double radius, float *normals_out, hipStream_t stream);

// ---- Candidate search (search.hip) -----------------------------------------------
struct SearchState {
  int num_samples = 0, capacity_samples = 0;
  int nn_cap = 0;                     // entries per neighbourhood list (8192 or 16384)
  uint64_t cloud_generation = 0;
  int32_t *d_sample_idx = nullptr;    // [S]
  double *d_sample_xyz = nullptr;     // [S][3] samples given by coordinates (used instead of the indices)
  int32_t *d_counts = nullptr;        // [S][8]: N_hands, N_images, k_frames, total found, seen by camera 0
  int32_t *d_nn_idx = nullptr;        // [S][nn_cap] sorted by (d2, index)
  float *d_nn = nullptr;              // [S][6][nn_cap] gathered px,py,pz,nx,ny,nz in that order
  double *d_frames = nullptr;         // [S][12] sample, normal, binormal, curvature
  double *d_centers = nullptr;        // [S][3] mean of the image neighbourhood
  gpd_hand *d_hands = nullptr;        // [S][slots]
  std::vector<int32_t> h_counts;      // host copy of d_counts
  std::vector<int32_t> h_set_sample;  // set -> sample slot
  std::vector<double> h_samples;      // [S][3] sample coordinates (to validate hands passed to images)
};
// samples by index (sample_xyz == nullptr) or by coordinates (sample_idx == nullptr)
int search_run(const gpd_params &p, const Cloud &c, SearchState &s, const int32_t *sample_idx, const double *sample_xyz, int S,
               hipStream_t stream);
// HandSearch::reevaluateHypotheses on the uploaded cloud; invalidates the search state
int reevaluate_run(const gpd_params &p, const Cloud &c, SearchState &s, gpd_hand *hands, int n, int32_t *labels, hipStream_t stream);
int search_download(const gpd_params &p, SearchState &s, gpd_hand *hands, int *num_sets, hipStream_t stream);
void search_free(SearchState &s);

// GraspDetector::filterGraspsWorkspace (grasp_detector.cpp:334-398) on the host.
void filter_workspace_host(const gpd_params &p, gpd_hand *hands, int num_sets);

// ---- Grasp images (images.hip) ---------------------------------------------------
struct ImageState {
  int num_candidates = 0;
  int capacity = 0;                   // images
  uint8_t *d_images = nullptr;        // planar [n][C][60][60]
  uint8_t *d_images_hwc = nullptr;    // [n][60][60][C], only when the caller downloads pixels
  gpd_hand *d_hands = nullptr;        // candidate hand records
  int32_t *d_cand_meta = nullptr;     // [n][4]: sample slot, N_images, shadow-set ordinal, -
  int32_t *d_set_meta = nullptr;      // [bitsets][8]: sample slot, N_images, lcg offset lo, hi, camera
  uint32_t *d_set_bits = nullptr;     // [sets][SETWORDS] shadow voxel bitsets
  int num_shadow_sets = 0, cap_shadow_sets = 0;
  int channels = 0;
  int32_t *d_overflow = nullptr;      // [capacity + 1]: candidates for the large shadow instantiation, then their count
  int num_overflow = 0;               // as found by the last checked launch (replays reuse it)
  int32_t *d_pts_overflow = nullptr;  // [capacity + 1]: candidates for the large points instantiation, then their count
  int num_pts_overflow = 0;
  char *d_pts_scratch = nullptr;      // point arrays of the large instantiation, one row per listed candidate
  int cap_pts_scratch = 0;            // rows
  int32_t *d_status = nullptr;        // error flags from the kernel
  int cap_hands = 0;
  long long stat_sets = 0, stat_sum_set_ni = 0, stat_sum_cand_ni = 0;  // for the algorithmic byte count
  std::vector<unsigned char> consts;  // the constant block (geometry, view points) the candidate list was built with
};
int images_run(const gpd_params &p, const Cloud &c, const SearchState &s, ImageState &im, const gpd_hand *hands, int num_sets,
               int32_t *cand_index, hipStream_t stream);
int images_launch(const SearchState &s, ImageState &im, hipStream_t stream, bool check);
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
  double deepen_depths[32];
  int num_deepen;
  double cos_friction;
  double nn_radius_hands, nn_radius_images;
};
void host_consts(const gpd_params &p, HostConsts &h);

}  // namespace gpd
