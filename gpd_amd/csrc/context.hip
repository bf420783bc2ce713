// C-ABI of libgpd_hip.so (include/gpd_hip.h): context, uploads, stage launches.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include <chrono>
#include <cmath>

#include "gpd_internal.h"

namespace gpd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
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

struct gpd_hip_ctx {
  int device = 0;
  gpd_params params;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // EXPERIMENT
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float stage_ms[3] = {0.f, 0.f, 0.f};
  LeNetWeights lenet;
  LeNetScratch lenet_scratch;
  Cloud cloud;
  SearchState search;
  ImageState images;
  // staging for gpd_hip_score with host images
  uint8_t *d_img_in = nullptr;      // HWC images handed to gpd_hip_score
  uint8_t *d_img_planar = nullptr;  // their planar copy
  size_t d_img_in_bytes = 0;
  float *d_scores = nullptr;
  int d_scores_cap = 0;
  std::vector<hipEvent_t> replay_events;  // 6 per gpd_hip_replay call: start, images done, conv1, conv2, fc1, end
  float replay_kernel_ms[4] = {0, 0, 0, 0};  // conv1, conv2, fc1, fc2 sums of the replays of the last gpd_hip_replay_times
  size_t replay_used = 0;
};

static int reserve_scores(gpd_hip_ctx *ctx, int n) {
  if (n <= ctx->d_scores_cap) return GPD_OK;
  if (ctx->d_scores) (void)hipFree(ctx->d_scores);
  ctx->d_scores = nullptr;
  ctx->d_scores_cap = 0;
  HIP_TRY(hipMalloc(&ctx->d_scores, (size_t)n * sizeof(float)));
  ctx->d_scores_cap = n;
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
}

const char *gpd_hip_last_error(void) { return g_err; }

int gpd_hip_create(int device, const gpd_params *params, gpd_hip_ctx **out) {
  if (!params || !out) {
    set_error("gpd_hip_create: null argument");
    return GPD_ERR_INVALID;
  }
  const int C = params->image_num_channels;
  if (params->image_size != kImg || (C != 1 && C != 3 && C != 12 && C != 15)) {
    set_error("gpd_hip_create: image_size must be 60 and image_num_channels one of 1/3/12/15");
    return GPD_ERR_INVALID;
  }
  const int slots = params->num_hand_axes * params->num_orientations;
  if (params->num_hand_axes < 1 || params->num_hand_axes > 3 || slots < 1 || slots > GPD_MAX_SLOTS ||
      params->num_finger_placements < 1 || params->num_finger_placements > 16) {
    set_error("gpd_hip_create: unsupported num_hand_axes/num_orientations/num_finger_placements");
    return GPD_ERR_INVALID;
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
  HIP_TRY(hipStreamCreate(&ctx->stream));
  for (auto &e : ctx->ev) HIP_TRY(hipEventCreate(&e));
  *out = ctx;
  return GPD_OK;
}

void gpd_hip_destroy(gpd_hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  float **ws[] = {&ctx->lenet.c1w, &ctx->lenet.c1b, &ctx->lenet.c2w, &ctx->lenet.c2b,
                  &ctx->lenet.f1w, &ctx->lenet.f1b, &ctx->lenet.f2w, &ctx->lenet.f2b, &ctx->lenet.c1wp, &ctx->lenet.c2wt};
  for (float **p : ws)
    if (*p) (void)hipFree(*p);
  lenet_scratch_free(ctx->lenet_scratch);
  cloud_free(ctx->cloud);
  search_free(ctx->search);
  images_free(ctx->images);
  if (ctx->d_img_in) (void)hipFree(ctx->d_img_in);
  if (ctx->d_img_planar) (void)hipFree(ctx->d_img_planar);
  if (ctx->d_scores) (void)hipFree(ctx->d_scores);
  for (auto &e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->replay_events) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
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
  // conv1 drops input windows that are entirely zero; that is exact only for finite weights
  // (inf * 0 would be NaN in the reference's dense GEMM)
  for (size_t i = 0; i < (size_t)20 * channels * 25; i++)
    if (!std::isfinite(conv1_w[i])) {
      set_error("gpd_hip_set_lenet_weights: conv1 weight %zu is not finite", i);
      return GPD_ERR_INVALID;
    }
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
  ctx->lenet.channels = channels;
  return GPD_OK;
}

int gpd_hip_score(gpd_hip_ctx *ctx, const uint8_t *images, int n, float *scores) {
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
  const size_t bytes = (size_t)n * kPix * ctx->lenet.channels;
  const uint8_t *d_img = nullptr;
  if (images) {
    if (bytes > ctx->d_img_in_bytes) {
      if (ctx->d_img_in) (void)hipFree(ctx->d_img_in);
      if (ctx->d_img_planar) (void)hipFree(ctx->d_img_planar);
      ctx->d_img_in = nullptr;
      ctx->d_img_planar = nullptr;
      ctx->d_img_in_bytes = 0;
      HIP_TRY(hipMalloc(&ctx->d_img_in, bytes));
      HIP_TRY(hipMalloc(&ctx->d_img_planar, bytes));
      ctx->d_img_in_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_img_in, images, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hwc_to_planar(ctx->d_img_in, ctx->d_img_planar, n, ctx->lenet.channels, ctx->stream));
    d_img = ctx->d_img_planar;
  } else {
    if (n != ctx->images.num_candidates || !ctx->images.d_images) {
      set_error("gpd_hip_score: no device images for n=%d (gpd_hip_images produced %d)", n, ctx->images.num_candidates);
      return GPD_ERR_STATE;
    }
    d_img = ctx->images.d_images;
  }
  int rc = reserve_scores(ctx, n);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
  HIP_TRY(lenet_forward(ctx->lenet, ctx->lenet_scratch, d_img, n, ctx->d_scores, ctx->stream));
  HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
  HIP_TRY(hipMemcpyAsync(scores, ctx->d_scores, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipEventElapsedTime(&ctx->stage_ms[2], ctx->ev[2], ctx->ev[3]));
  return GPD_OK;
}

int gpd_hip_upload_cloud(gpd_hip_ctx *ctx, const float *xyz, const float *normals, int num_points, const int32_t *cam_source,
                         int num_cams, const double *view_points) {
  if (!ctx || !xyz || !normals || num_points <= 0 || !cam_source || num_cams < 1 || !view_points) {
    set_error("gpd_hip_upload_cloud: bad argument");
    return GPD_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(cloud_upload(ctx->cloud, xyz, normals, num_points, cam_source, num_cams, view_points, ctx->stream));
  return GPD_OK;
}

int gpd_hip_estimate_normals(gpd_hip_ctx *ctx, double radius, float *normals) {
  if (!ctx || !normals || !(radius > 0.0)) {
    set_error("gpd_hip_estimate_normals: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->cloud.num_points) {
    set_error("gpd_hip_estimate_normals: no cloud uploaded");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return normals_run(ctx->cloud, radius, normals, ctx->stream);
}

// samples by index (sample_xyz == nullptr) or by coordinates (sample_indices == nullptr)
static int search_any(gpd_hip_ctx *ctx, const char *who, const int32_t *sample_indices, const double *sample_xyz, int num_samples,
                      gpd_hand *hands, int *num_sets) {
  if (!ctx || (!sample_indices && !sample_xyz) || num_samples < 0 || !hands || !num_sets) {
    set_error("%s: bad argument", who);
    return GPD_ERR_INVALID;
  }
  if (!ctx->cloud.num_points) {
    set_error("%s: no cloud uploaded", who);
    return GPD_ERR_STATE;
  }
  *num_sets = 0;
  if (num_samples == 0) return GPD_OK;
  if (sample_indices) {
    for (int i = 0; i < num_samples; i++)
      if (sample_indices[i] < 0 || sample_indices[i] >= ctx->cloud.num_points) {
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
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
  int rc = search_run(ctx->params, ctx->cloud, ctx->search, sample_indices, sample_xyz, num_samples, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
  rc = search_download(ctx->params, ctx->search, hands, num_sets, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipEventElapsedTime(&ctx->stage_ms[0], ctx->ev[0], ctx->ev[1]));
  return GPD_OK;
}

int gpd_hip_search(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, gpd_hand *hands, int *num_sets) {
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
  if (!ctx || num_hands < 0 || (num_hands > 0 && (!hands || !labels))) {
    set_error("gpd_hip_reevaluate: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->cloud.num_points) {
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
  return reevaluate_run(ctx->params, ctx->cloud, ctx->search, hands, num_hands, labels, ctx->stream);
}

int gpd_hip_images(gpd_hip_ctx *ctx, const gpd_hand *hands, int num_sets, uint8_t *images, int32_t *cand_index,
                   int *num_candidates) {
  if (!ctx || !hands || num_sets < 0 || !num_candidates) {
    set_error("gpd_hip_images: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->cloud.num_points) {
    set_error("gpd_hip_images: no cloud uploaded");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
  int rc = images_run(ctx->params, ctx->cloud, ctx->search, ctx->images, hands, num_sets, cand_index, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
  *num_candidates = ctx->images.num_candidates;
  if (images && ctx->images.num_candidates > 0) {
    // the caller wants cv::Mat-layout pixels: planar -> HWC on the device, then one copy
    const size_t bytes = (size_t)ctx->images.capacity * kPix * ctx->params.image_num_channels;
    if (!ctx->images.d_images_hwc) HIP_TRY(hipMalloc(&ctx->images.d_images_hwc, bytes));
    HIP_TRY(planar_to_hwc(ctx->images.d_images, ctx->images.d_images_hwc, ctx->images.num_candidates,
                          ctx->params.image_num_channels, ctx->stream));
    HIP_TRY(hipMemcpyAsync(images, ctx->images.d_images_hwc,
                           (size_t)ctx->images.num_candidates * kPix * ctx->params.image_num_channels, hipMemcpyDeviceToHost,
                           ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipEventElapsedTime(&ctx->stage_ms[1], ctx->ev[0], ctx->ev[1]));
  return GPD_OK;
}

static int detect_any(gpd_hip_ctx *ctx, const int32_t *sample_indices, const double *sample_xyz, int num_samples, gpd_hand *hands,
                      int *num_sets, int *num_candidates) {
  if (!ctx || !hands || !num_sets || !num_candidates || (!sample_indices && !sample_xyz)) {
    set_error("gpd_hip_detect: bad argument");
    return GPD_ERR_INVALID;
  }
  if (!ctx->lenet.channels) {
    set_error("gpd_hip_detect: LeNet weights not set");
    return GPD_ERR_STATE;
  }
  *num_candidates = 0;
  // GPD_DETECT_TIMING=1: wall time of the stages incl. their host hops, to stderr
  const bool timing = getenv("GPD_DETECT_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  int rc = search_any(ctx, "gpd_hip_detect", sample_indices, sample_xyz, num_samples, hands, num_sets);
  if (rc) return rc;
  if (*num_sets == 0) return GPD_OK;
  const double t1 = now();
  filter_workspace_host(ctx->params, hands, *num_sets);
  const double t2 = now();
  std::vector<int32_t> cand((size_t)(*num_sets) * ctx->params.num_hand_axes * ctx->params.num_orientations);
  rc = gpd_hip_images(ctx, hands, *num_sets, nullptr, cand.data(), num_candidates);
  if (rc) return rc;
  if (*num_candidates == 0) return GPD_OK;
  const double t3 = now();
  std::vector<float> scores(*num_candidates);
  rc = gpd_hip_score(ctx, nullptr, *num_candidates, scores.data());
  if (rc) return rc;
  const double t4 = now();
  for (int i = 0; i < *num_candidates; i++) hands[cand[i]].score = scores[i];
  if (timing)
    fprintf(stderr, "[detect-timing] search %.2f ms (kernels %.2f)  filter %.2f  images %.2f (kernels %.2f)  score %.2f (kernels %.2f)  scatter %.2f\n",
            t1 - t0, ctx->stage_ms[0], t2 - t1, t3 - t2, ctx->stage_ms[1], t4 - t3, ctx->stage_ms[2], now() - t4);
  return GPD_OK;
}

int gpd_hip_detect(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, gpd_hand *hands, int *num_sets,
                   int *num_candidates) {
  if (!sample_indices) {
    set_error("gpd_hip_detect: bad argument");
    return GPD_ERR_INVALID;
  }
  return detect_any(ctx, sample_indices, nullptr, num_samples, hands, num_sets, num_candidates);
}

int gpd_hip_detect_samples(gpd_hip_ctx *ctx, const double *samples_xyz, int num_samples, gpd_hand *hands, int *num_sets,
                           int *num_candidates) {
  if (!samples_xyz) {
    set_error("gpd_hip_detect_samples: bad argument");
    return GPD_ERR_INVALID;
  }
  return detect_any(ctx, nullptr, samples_xyz, num_samples, hands, num_sets, num_candidates);
}

int gpd_hip_replay(gpd_hip_ctx *ctx, int stages) {
  if (!ctx || !(stages & 7)) {
    set_error("gpd_hip_replay: bad argument");
    return GPD_ERR_INVALID;
  }
  if (ctx->images.num_candidates <= 0 || !ctx->images.d_images) {
    set_error("gpd_hip_replay: no candidate list on the device (call gpd_hip_images first)");
    return GPD_ERR_STATE;
  }
  if ((stages & 2) && !ctx->lenet.channels) {
    set_error("gpd_hip_replay: LeNet weights not set");
    return GPD_ERR_STATE;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const int n = ctx->images.num_candidates;
  int rc = reserve_scores(ctx, n);
  if (rc) return rc;
  while (ctx->replay_events.size() < ctx->replay_used + 6) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    ctx->replay_events.push_back(e);
  }
  hipEvent_t *ev = &ctx->replay_events[ctx->replay_used];
  ctx->replay_used += 6;
  HIP_TRY(hipEventRecord(ev[0], ctx->stream));
  if (stages & 4) {  // EXPERIMENT: images on a side stream, concurrently with the LeNet pass
    if (!ctx->stream2) HIP_TRY(hipStreamCreate(&ctx->stream2));
    HIP_TRY(hipStreamWaitEvent(ctx->stream2, ev[0], 0));
    rc = images_launch(ctx->search, ctx->images, ctx->stream2, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev[1], ctx->stream2));
    HIP_TRY(lenet_forward(ctx->lenet, ctx->lenet_scratch, ctx->images.d_images, n, ctx->d_scores, ctx->stream, ev + 2));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ev[1], 0));
    HIP_TRY(hipEventRecord(ev[5], ctx->stream));
    return GPD_OK;
  }
  if (stages & 1) {
    rc = images_launch(ctx->search, ctx->images, ctx->stream, false);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(ev[1], ctx->stream));
  if (stages & 2) {
    HIP_TRY(lenet_forward(ctx->lenet, ctx->lenet_scratch, ctx->images.d_images, n, ctx->d_scores, ctx->stream, ev + 2));
  } else {
    for (int i = 2; i < 5; i++) HIP_TRY(hipEventRecord(ev[i], ctx->stream));
  }
  HIP_TRY(hipEventRecord(ev[5], ctx->stream));
  return GPD_OK;
}

int gpd_hip_replay_times(gpd_hip_ctx *ctx, float ms[2], int *launches, float *scores) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
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
  if (scores && ctx->images.num_candidates > 0 && ctx->d_scores)
    HIP_TRY(hipMemcpy(scores, ctx->d_scores, (size_t)ctx->images.num_candidates * sizeof(float), hipMemcpyDeviceToHost));
  int32_t status = 0;
  if (ctx->images.d_status) HIP_TRY(hipMemcpy(&status, ctx->images.d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (status) {
    set_error("gpd_hip_replay_times: image kernel reported capacity flags %d", status);
    return GPD_ERR_CAPACITY;
  }
  return GPD_OK;
}

int gpd_hip_replay_kernel_ms(gpd_hip_ctx *ctx, float ms[4]) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  for (int k = 0; k < 4; k++) ms[k] = ctx->replay_kernel_ms[k];
  return GPD_OK;
}

int gpd_hip_last_images_stats(gpd_hip_ctx *ctx, long long out[4]) {
  if (!ctx || !out) return GPD_ERR_INVALID;
  out[0] = ctx->images.num_candidates;
  out[1] = ctx->images.stat_sets;
  out[2] = ctx->images.stat_sum_set_ni;
  out[3] = ctx->images.stat_sum_cand_ni;
  return GPD_OK;
}

int gpd_hip_last_stage_ms(gpd_hip_ctx *ctx, float ms[3]) {
  if (!ctx || !ms) return GPD_ERR_INVALID;
  for (int i = 0; i < 3; i++) ms[i] = ctx->stage_ms[i];
  return GPD_OK;
}

}  // extern "C"
