/*
 * gpd_hip.h — C-ABI of libgpd_hip.so, the MI355X (gfx950) implementation of the
 * GPD hot path: candidate search -> grasp image -> LeNet score.
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  Host buffers are
 * owned by the caller, device memory is owned by the context.  A context is not
 * thread-safe (the reference's GraspDetector is not either: grasp_detector.cpp:192-328
 * is called from one thread); different contexts — on different devices or on the
 * same one — may be driven from different threads.  The hand / image geometry and
 * the view points live in one constant block per device: contexts on a device
 * that agree on them overlap freely, a context with different values waits for
 * the device before it loads its own (correct, but serialising).
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference tree).  Return value: 0 on success, <0 on error
 * (gpd_hip_last_error() gives the text).  Nothing throws across the boundary.
 */
#ifndef GPD_HIP_H_
#define GPD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPD_MAX_SLOTS 24 /* num_hand_axes * num_orientations upper bound */

/* Error codes (negative). */
#define GPD_OK 0
#define GPD_ERR_INVALID (-1)   /* bad argument                                   */
#define GPD_ERR_HIP (-2)       /* a HIP runtime call failed                      */
#define GPD_ERR_CAPACITY (-3)  /* a neighbourhood exceeded the LDS list capacity */
#define GPD_ERR_STATE (-4)     /* call order violated (no cloud / no weights)    */

/* How Classifier::classifyImages' dot products are summed (gpd_hip_set_lenet_mode):
 *  GPD_LENET_SPLIT      (default) conv1 on the int8 matrix pipe — exact integer dot products of the u8 inputs with 32-bit
 *                       fixed-point weights, one rounding — conv2 / ip1 on the bf16 matrix pipe with every f32 operand cut
 *                       into three bf16 pieces (six exact piece products per term, f32 accumulation); ip2 as f32 chains.
 *                       The reference's Eigen GEMM fixes no summation order (conv_layer.cpp:54, dense_layer.cpp:11); the
 *                       scores are within 1e-4 of its plain-float path and closer to float64 than an f32 chain's.
 *  GPD_LENET_F32_CHAIN  every dot product as ONE k-ascending f32 fmaf chain on the f32-input MFMA: bit-identical to
 *                       oracle/gpd_oracle.cpp, 1/16 of the matrix rate.  The checker mode. */
#define GPD_LENET_SPLIT 0
#define GPD_LENET_F32_CHAIN 1

/*
 * Parameters of the path.  Field names follow the reference's cfg keys:
 * hand geometry  — cfg/hand_geometry.cfg:8-12, candidate/hand_geometry.cpp:25-30
 * image geometry — cfg/image_geometry_15channels.cfg:8-12, descriptor/image_geometry.cpp:24-28
 * search         — grasp_detector.cpp:67-86 (HandSearch::Parameters)
 * filter         — grasp_detector.cpp:158-174
 */
typedef struct gpd_params {
  double finger_width;        /* 0.01 */
  double hand_outer_diameter; /* 0.12 */
  double hand_depth;          /* 0.06 */
  double hand_height;         /* 0.02 */
  double init_bite;           /* 0.01 */
  double volume_width;        /* 0.10 */
  double volume_depth;        /* 0.06 */
  double volume_height;       /* 0.02 */
  double nn_radius_frames;    /* nn_radius, 0.01 */
  double friction_coeff;      /* 20 */
  double min_aperture;        /* 0.0 */
  double max_aperture;        /* 0.085 */
  double workspace_grasps[6]; /* -1 1 -1 1 -1 1 */
  int32_t image_size;         /* 60 (only 60 is supported, as eigen_classifier.cpp:12) */
  int32_t image_num_channels; /* 1, 3, 12 or 15 (image_strategy.cpp:15-29) */
  int32_t num_orientations;   /* 8 */
  int32_t num_finger_placements; /* 10 */
  int32_t num_hand_axes;      /* 1 */
  int32_t hand_axes[3];       /* {2} */
  int32_t deepen_hand;        /* 1 */
  int32_t min_viable;         /* 6 */
  /* GraspDetector::filterGraspsDirection (grasp_detector.cpp:247-250, 423-456; cfg keys filter_approach_direction,
   * direction, thresh_rad): a valid hand whose approach axis makes an angle acos(direction . approach) > thresh_rad
   * with `direction` is dropped — in the fused entries, on the device, right after the workspace filter. */
  int32_t filter_approach_direction; /* 0 */
  int32_t reserved_;
  double direction[3];        /* 1 0 0 */
  double thresh_rad;          /* 2.0 */
} gpd_params;

/*
 * One grasp candidate = candidate::Hand (include/gpd/candidate/hand.h:80-277,
 * candidate/hand.cpp:24-45) flattened to POD.  `frame` is row-major; its columns
 * are approach | binormal | axis (hand.h getApproach/getBinormal/getAxis).
 */
typedef struct gpd_hand {
  double sample[3];
  double frame[9];
  double position[3];
  double top, bottom, center; /* closing box, hand.h BoundingBox */
  double grasp_width;
  float score;
  int32_t finger_placement_index; /* -1 when no feasible placement exists */
  int32_t set_index;              /* hand set = sample that produced it     */
  int32_t slot;                   /* axis_i * num_orientations + angle_i    */
  uint8_t valid;                  /* HandSet::is_valid_                     */
  uint8_t half_antipodal, full_antipodal;
  uint8_t pad_[5];
} gpd_hand;

typedef struct gpd_hip_ctx gpd_hip_ctx;

/* Fill `p` with the defaults of cfg/eigen_params.cfg + cfg/hand_geometry.cfg +
 * cfg/image_geometry_15channels.cfg. */
void gpd_hip_default_params(gpd_params *p);

/* Create a context on HIP device `device`.  Replaces the constructor work of
 * GraspDetector (grasp_detector.cpp:5-190) that sizes the path. */
int gpd_hip_create(int device, const gpd_params *params, gpd_hip_ctx **out);
void gpd_hip_destroy(gpd_hip_ctx *ctx);
const char *gpd_hip_last_error(void);

/* LeNet parameters, raw float32 exactly as the files read by
 * EigenClassifier::EigenClassifier (net/eigen_classifier.cpp:28-50):
 * conv1 [20][C*25] row-major, conv2 [50][500] row-major, ip1 column-major
 * 500 x 7200 over the pixel-major flatten (eigen_classifier.cpp:103-107,
 * dense_layer.cpp:7), ip2 column-major 2 x 500.  Copied to the device once.
 * All conv1 / conv2 / ip1 weights must be finite (GPD_ERR_INVALID otherwise): the f32 chain's conv1 skips input
 * patches that are all zero, which is exact for finite weights only (0 * inf = NaN), and the split
 * path's fixed-point / bf16 pieces are defined for finite numbers. */
int gpd_hip_set_lenet_weights(gpd_hip_ctx *ctx, int channels,
                              const float *conv1_w, const float *conv1_b,
                              const float *conv2_w, const float *conv2_b,
                              const float *ip1_w, const float *ip1_b,
                              const float *ip2_w, const float *ip2_b);

/* GPD_LENET_SPLIT or GPD_LENET_F32_CHAIN (above) for every later scoring call of the context. */
int gpd_hip_set_lenet_mode(gpd_hip_ctx *ctx, int mode);

/* Replaces Classifier::classifyImages (net/classifier.h:70-71,
 * eigen_classifier.cpp:59-79).  images: n contiguous 60x60xC u8 HWC images
 * (cv::Mat CV_8UC(C) layout); scores[i] = logit1 - logit0.
 * images == NULL scores the images left on the device by gpd_hip_images. */
int gpd_hip_score(gpd_hip_ctx *ctx, const uint8_t *images, int n, float *scores);

/* Upload the processed cloud: what HandSearch::searchHands and
 * ImageGenerator::createImages read from util::Cloud (hand_search.cpp:28-31,
 * 160-165; image_generator.cpp:24-29): float32 xyz (AoS), float32 normals
 * (AoS), camera source n_cams x P (row per camera, 0/1), view points 3 doubles
 * per camera.  1 <= num_cams <= 32 (GPD_ERR_INVALID beyond: the kernels carry the
 * cameras that see a neighbourhood as a 32-bit mask). */
int gpd_hip_upload_cloud(gpd_hip_ctx *ctx, const float *xyz, const float *normals,
                         int num_points, const int32_t *cam_source, int num_cams,
                         const double *view_points);

/* SURVEY §8f rank 3, the step after selectGrasps: replaces Clustering::findClusters (clustering.cpp:5-105).
 * hands: n records (axis = third column of `frame`, `position`); scores: their scores as doubles (Hand keeps a double,
 * hand.h:253; the record's float is not read).  A seed hand with at least min_inliers inliers (axis within 12 degrees,
 * position within 0.05 m and within 0.005 m of the seed's axis line, clustering.cpp:9-13) yields a cluster: the seed's
 * record with position = mean inlier position, out_scores = lower bound of the 99 % confidence interval of the inlier
 * scores (the record's float score is its rounding), out_src = index of the seed; clusters come in seed order.
 * remove_inliers != 0: a hand that was an inlier of an earlier seed is skipped by the later ones (:36, :70-72).
 * out / out_scores / out_src must hold n entries. */
int gpd_hip_find_clusters(gpd_hip_ctx *ctx, const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers,
                          gpd_hand *out, double *out_scores, int32_t *out_src, int *num_out);

/* SURVEY §8f rank 2 (the step before the normals): replaces the point-cloud part of Cloud::filterWorkspace
 * (util/cloud.cpp:243-266) followed by Cloud::voxelizeCloud (util/cloud.cpp:286-348) as
 * CandidatesGenerator::preprocessPointCloud runs them on a cloud without normals (candidates_generator.cpp:19-26).
 * xyz: num_points x 3 float32 without NaN/Inf (pcl::removeNaNFromPointCloud ran at load time, cloud.cpp:154-164);
 * cam_source: num_cams rows of num_points (may be NULL with num_cams = 0).
 * workspace: 6 doubles (min/max x, y, z; strict comparisons) or NULL = no cut.  voxel_size <= 0: no voxeliser, the
 * points inside the workspace come back in input order with their camera-source columns.  voxel_size > 0: the points
 * the reference's std::set keeps under its "differs" comparator (cloud.h:105-122) — not one per voxel, see
 * gpd_amd/csrc/preprocess.hip — replaced by their voxel corner min + voxel_size * index, in the set's iteration order,
 * camera source reduced to (== 1 ? 1 : 0) as cloud.cpp:325-327.
 * xyz_out / cam_out (num_cams rows of *num_out) / src_out (input index per output point, may be NULL) must hold
 * num_points entries.  kernel_ms (may be NULL) receives the device time of the kernels.  The context's uploaded cloud
 * is not touched: gpd_hip_upload_cloud + gpd_hip_estimate_normals follow with the result. */
int gpd_hip_preprocess_cloud(gpd_hip_ctx *ctx, const float *xyz, const int32_t *cam_source, int num_points, int num_cams,
                             const double *workspace, float voxel_size, float *xyz_out, int32_t *cam_out, int32_t *src_out,
                             int *num_out, float *kernel_ms);

/* SURVEY §8f rank 1 (the step that feeds the path its normals): replaces Cloud::calculateNormals
 * (util/cloud.cpp:451-476) = calculateNormalsOMP (:497-535, radius search, PCA, flip towards the
 * view point) + reverseNormals (:573-604), on the cloud uploaded last (its normals argument may
 * be zeros).  normals receives num_points*3 floats and also replaces the device copy. */
int gpd_hip_estimate_normals(gpd_hip_ctx *ctx, double radius, float *normals);

/* Replaces CandidatesGenerator::generateGraspCandidateSets ->
 * HandSearch::searchHands (candidates_generator.cpp:62-69, hand_search.cpp:24-64)
 * for samples given by index (Cloud::getSampleIndices).  Writes
 * num_sets * num_slots hands (set-major, slot-minor; num_slots = num_hand_axes *
 * num_orientations); samples without a frame neighbourhood are dropped before
 * sets are numbered (frame_estimator.cpp:24-29).  hands must hold
 * num_samples * num_slots records. */
int gpd_hip_search(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples,
                   gpd_hand *hands, int *num_sets);

/* The same for samples given by coordinates (Cloud::getSamples; hand_search.cpp:37-39,
 * FrameEstimator::calculateLocalFrames(cloud, samples, ...) frame_estimator.cpp:38-65):
 * samples_xyz holds 3 doubles per sample.  As in the reference the kd-tree queries use the float
 * cast of the sample (eigenVectorToPcl) while the hand frame keeps the double. */
int gpd_hip_search_samples(gpd_hip_ctx *ctx, const double *samples_xyz, int num_samples,
                           gpd_hand *hands, int *num_sets);

/* replaces HandSearch::reevaluateHypotheses (hand_search.cpp:66-134, 190-228;
 * GraspDetector::evalGroundTruth, grasp_detector.cpp:522-526) on the cloud uploaded last (the
 * ground-truth cloud): each hand is checked again with its own frame, `top` and
 * finger_placement_index; labels[i] = 1 for a full antipodal grasp, half_antipodal /
 * full_antipodal of the records are rewritten.  Reuses the search buffers: hands of an earlier
 * gpd_hip_search can no longer be passed to gpd_hip_images afterwards. */
int gpd_hip_reevaluate(gpd_hip_ctx *ctx, gpd_hand *hands, int num_hands, int32_t *labels);

/* Replaces ImageGenerator::createImages (image_generator.cpp:17-99) for hand
 * sets produced by the last gpd_hip_search / gpd_hip_detect on this context (optionally after
 * the host filters, grasp_detector.cpp:334-398 / :422-453, which clear `valid`: only the `valid`
 * flags and the sets' samples are read from `hands`, the records themselves are on the device).  One image per valid hand
 * in set-major, slot-minor order; sets without a valid hand are skipped like
 * filterGraspsWorkspace drops them.  images (may be NULL: keep on device) holds
 * n_cand * 60*60*C bytes HWC.  cand_index (may be NULL) receives for each image
 * the index into `hands`. */
int gpd_hip_images(gpd_hip_ctx *ctx, const gpd_hand *hands, int num_sets,
                   uint8_t *images, int32_t *cand_index, int *num_candidates);

/* Fused path used by GraspDetector::detectGrasps steps 1-4
 * (grasp_detector.cpp:222-273): search, workspace/aperture filter
 * (filterGraspsWorkspace, :238, :334-398), images, scores, score write-back (:269-273).
 * Everything stays on the device between the stages: the filter runs at the end of the
 * hand kernel, one small kernel builds the candidate list (set-major, slot-minor, as
 * image_generator.cpp:91-98) and places every hand set in the shadow LCG stream, and the
 * only host hop in the middle is a 48-byte summary that sizes the launches.  One copy in
 * (the samples), one copy out (the records).  hands receives num_sets*num_slots records
 * (room for num_samples*num_slots is required): `valid` is the flag after the filter,
 * `score` is set on the valid ones. */
int gpd_hip_detect(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples,
                   gpd_hand *hands, int *num_sets, int *num_candidates);

/* The same, returning what detectGrasps keeps after step 4 / step 5 instead of every slot:
 * num_selected == 0: the scored candidates — the `hands` list ImageGenerator::createImages moves
 *   out of the hand sets (image_generator.cpp:91-98), in that order;
 * num_selected  > 0: GraspDetector::selectGrasps (grasp_detector.cpp:405-420): the
 *   min(num_selected, candidates) best, score descending, picked on the device so that only those
 *   records cross PCIe (equal scores: the arrangement std::partial_sort leaves, as the reference).
 * hands holds hands_capacity records; *num_hands receives how many were written. */
int gpd_hip_detect_select(gpd_hip_ctx *ctx, const int32_t *sample_indices, int num_samples, int num_selected,
                          gpd_hand *hands, int hands_capacity, int *num_sets, int *num_candidates,
                          int *num_hands);

/* Sizes every buffer of the context — both lanes of the batch entry — for clouds of up to max_points points,
 * max_cams cameras and max_samples samples, so that no later call within those sizes allocates (growing a buffer is
 * a hipFree + hipMalloc, which waits for the whole device).  max_candidates: scored hands per cloud, 0 = the upper
 * bound max_samples x slots (cut to 16 GB of candidate-sized buffers per lane); max_selected: the largest
 * num_selected that will be asked for (0: none).  Optional: gpd_hip_detect_batch sizes its lanes itself from the
 * jobs it is given; the single-cloud entries grow their buffers on demand (25 % slack, never shrinking).  No
 * reference counterpart: the reference allocates per call. */
int gpd_hip_reserve(gpd_hip_ctx *ctx, int max_points, int max_cams, int max_samples, int max_candidates, int max_selected);

/* One independent cloud of a batch: the arguments of gpd_hip_upload_cloud + gpd_hip_detect_select.
 * ZERO the struct (memset / `gpd_detect_job j = {0}`) before filling it: fields added in later rounds (lcg_base, raw, voxel_size,
 * workspace, normals_radius, sample_xyz) are INPUTS, and an uninitialised `raw` or `lcg_base` selects the raw-scan route or
 * offsets the cloud's shadow stream — wrong images, not an error. */
typedef struct gpd_detect_job {
  const float *xyz;            /* in */
  const float *normals;
  const int32_t *cam_source;
  const double *view_points;
  const int32_t *sample_indices;
  gpd_hand *hands;             /* out: hands_capacity records */
  int32_t num_points, num_cams, num_samples;
  int32_t num_selected;        /* 0: all candidates; > 0: selectGrasps */
  int32_t hands_capacity;
  int32_t num_sets, num_candidates, num_hands; /* out */
  int32_t status;              /* out: GPD_OK or the error of this cloud */
  float stage_ms[3];           /* out: search, images, LeNet kernel time of this cloud */
  /* out: where the HOST was, in ms since the entry of gpd_hip_detect_batch: [0] upload + search + plan enqueued,
   * [1] plan summary arrived (the only mid-pipeline wait), [2] images + LeNet + gather enqueued, [3] results on the
   * host, [4] records handed over.  A host-side limiter shows here, not in stage_ms. */
  float host_ms[5];
  int32_t allocs;              /* out: buffer growths (hipFree + hipMalloc = a device stall) booked on this cloud;
                                  0 everywhere but the first cloud of a batch whose lanes were not yet sized */
  int32_t reserved_;
  /* The cloud's ONE stream of shadow draws (HandSet::fastrand, hand_set.cpp:263-283) when its samples are cut into ranges:
   * lcg_base (in) = draws of the sample ranges before this job's (0: the job starts the cloud's stream — every ordinary
   * call), lcg_draws (out) = draws of this job's hand sets.  gpd_hip_detect_sharded fills lcg_base itself. */
  uint64_t lcg_base;
  uint64_t lcg_draws;
  /* RAW scans (round 5): raw != 0 -> xyz / cam_source are the cloud as detect_grasps reads it from the sensor or the PCD, and
   * the job runs CandidatesGenerator::preprocessPointCloud (candidates_generator.cpp:14-37) on the device first:
   * Cloud::filterWorkspace with `workspace` (6 doubles xmin xmax ymin ymax zmin zmax; NULL: none), Cloud::voxelizeCloud
   * (voxel_size; <= 0: none), Cloud::calculateNormals(normals_radius, towards the cameras that see each point).  `normals` and
   * `sample_indices` are ignored (indices into the preprocessed cloud do not exist yet): the search runs at `sample_xyz`
   * (num_samples x 3 doubles, the `samples` route of the reference: cloud.h setSamples, hand_search.cpp:160-165).  The voxelised
   * cloud never leaves the device; the voxeliser's sequential keep / drop chain runs on the calling host thread while the
   * previous cloud's image / LeNet kernels run.  num_points_processed (out): points after preprocessing.
   * As in the reference, the preprocessing starts with Cloud::removeNans (a point with a NaN / Inf coordinate is dropped — the rows
   * an organised sensor scan carries for missing depth) and Cloud::filterWorkspace cuts the SAMPLES too (cloud.cpp:225-237: strict
   * double comparisons, order kept): the search runs at the samples inside `workspace` only, num_samples_processed (out) says how
   * many those were, and the job's records are those of that many samples. */
  int32_t raw;
  float voxel_size;
  const double *workspace;
  double normals_radius;
  const double *sample_xyz;
  int32_t num_points_processed;
  int32_t num_samples_processed;
} gpd_detect_job;

/* detect_grasps over a batch of independent clouds (src/detect_grasps.cpp:20-86 called once per
 * cloud; BASELINE configs[4]).  The context keeps two clouds in flight on two streams: upload, grid
 * and candidate search of cloud i+1 are enqueued while the image and LeNet kernels of cloud i run,
 * and the records of cloud i-1 are handed over meanwhile — the host hops of one cloud hide behind
 * the kernels of its neighbour.  Results are those of num_jobs separate
 * gpd_hip_upload_cloud + gpd_hip_detect_select calls (the shadow LCG restarts per cloud).  Returns
 * the first error; each job carries its own status.  The cloud uploaded with gpd_hip_upload_cloud
 * is replaced. */
int gpd_hip_detect_batch(gpd_hip_ctx *ctx, gpd_detect_job *jobs, int num_jobs);

/* The same over several contexts — one per GPU of a node, one host thread per context, job i -> context
 * i mod num_ctx, no device talks to another ("one host thread + one gpd_hip_ctx + 2-3 streams per
 * GPU"; the reference's unit of work is one detect_grasps process per cloud, src/detect_grasps.cpp:20-86).
 * Every context must carry the LeNet weights.  Results are those of gpd_hip_detect_batch on any one context
 * (the shadow LCG restarts per cloud, so they do not depend on the sharding).  Returns the first error. */
int gpd_hip_detect_batch_multi(gpd_hip_ctx *const *ctxs, int num_ctx, gpd_detect_job *jobs, int num_jobs);

/* ONE cloud over several contexts ("sample-range sharding with the cloud replicated"; BASELINE configs[3] across
 * GPUs): shards[g] is the job of ctxs[g] — the same cloud arrays, a CONTIGUOUS range of the cloud's samples (range g before
 * range g + 1 in the caller's sample order), its own output buffer, num_selected = 0.  The reference draws all shadow points of a
 * cloud from one LCG stream, hand set after hand set (hand_set.cpp:268-283): phase 1 searches every range and takes its draw
 * total, the totals are scanned on the host (num_ctx numbers — still no collective), phase 2 generates and scores the images
 * with lcg_base = the draws of the ranges before.  The concatenated records are byte for byte those of ONE
 * gpd_hip_detect_select(num_selected = 0) over all samples, whatever the split.  Every context must carry the LeNet weights. */
int gpd_hip_detect_sharded(gpd_hip_ctx *const *ctxs, int num_ctx, gpd_detect_job *shards);

/* Binds the CALLING host thread to the CPUs of the NUMA node `device` hangs off (sysfs numa_node / cpulist, within the
 * process's allowed set): the thread that feeds a GPU — staging copies, launches, result copies — should run on that
 * GPU's socket (8 feeding processes on a two-socket host).  Returns the node, or -1 when the host exposes no
 * topology (nothing changed); *num_cpus (may be NULL) receives the size of the new mask.  gpd_hip_detect_batch_multi does
 * this for its worker threads; one-process-per-GPU launchers (bench.py) call it before gpd_hip_create.  Pinned staging
 * memory is placed by hipHostMalloc near the current device already.  No reference counterpart. */
int gpd_hip_bind_host_thread(int device, int *num_cpus);

/* gpd_hip_detect for samples given by coordinates (see gpd_hip_search_samples). */
int gpd_hip_detect_samples(gpd_hip_ctx *ctx, const double *samples_xyz, int num_samples,
                           gpd_hand *hands, int *num_sets, int *num_candidates);

/* Stage times of the last call in milliseconds (HIP events on the context's
 * stream): [0] search, [1] images (incl. shadow), [2] score.  The counterpart
 * of the RUNTIMES printout, grasp_detector.cpp:313-320. */
int gpd_hip_last_stage_ms(gpd_hip_ctx *ctx, float ms[3]);

/* Sizes of the last gpd_hip_images call, for the algorithmic byte count of SURVEY §8d:
 * out[0] candidates, out[1] live hand sets, out[2] sum over live sets of the image
 * neighbourhood size N_i, out[3] sum over candidates of N_i. */
int gpd_hip_last_images_stats(gpd_hip_ctx *ctx, long long out[4]);

/* Which slow paths the last search / image stage took (none of them changes a result):
 * out[0] entries per neighbourhood list the search ran with (8192: bucket sort in LDS; 16384: bitonic sort in LDS;
 * more: global-memory lists), out[1] candidates whose box held more shadow voxels than the two-per-CU shadow
 * kernel lists (redone by the large instantiation), out[2] candidates with more in-box points than the two-per-CU
 * normals/depth kernel holds (redone with global scratch), out[3] LeNet passes of the last scoring (65536 images
 * each).  Waits for the context's stream. */
int gpd_hip_last_fallbacks(gpd_hip_ctx *ctx, long long out[4]);

/* Of the last search's 3 x num_samples centre coordinates (the mean of a sample's image neighbourhood, hand_set.cpp:131-133):
 * how many took the serial fp64 chain in neighbour order because the order-free sum taken inside the neighbourhood kernel could
 * not be certified exact (a point within micrometres of a coordinate plane among points decimetres away).  Wherever the
 * certificate holds the sum is the same in EVERY order — the oracle's sequential one and Eigen's packet reduction alike.
 * Measurement / tests only.  Waits for the context's stream. */
int gpd_hip_last_centre_chains(gpd_hip_ctx *ctx, long long *out);

/* Re-run stage 3 (stages == 1: grasp images), stage 4 (2: LeNet) or both (3) on the
 * candidate list that the last gpd_hip_images / gpd_hip_detect left resident on the
 * device — what calling ImageGenerator::createImages + Classifier::classifyImages
 * again on the same hand sets does (grasp_detector.cpp:261-273), without host hops.
 * Asynchronous on the context's stream; each call is bracketed by HIP events. */
int gpd_hip_replay(gpd_hip_ctx *ctx, int stages);

/* Synchronise; ms[0] / ms[1] = summed HIP-event time of the image / LeNet stage over the
 * gpd_hip_replay calls since the last query, *launches = their number; scores (may be
 * NULL) receives the scores of the last replay. */
int gpd_hip_replay_times(gpd_hip_ctx *ctx, float ms[2], int *launches, float *scores);

/* HIP-event durations of the four LeNet kernels (conv1+pool1, conv2+pool2, ip1, ip2+score; first
 * 65536-image chunk) summed over the replays covered by the last gpd_hip_replay_times call —
 * the per-kernel roofline input of bench.py. */
int gpd_hip_replay_kernel_ms(gpd_hip_ctx *ctx, float ms[4]);

/* conv1's zero skipping, counted by the kernel itself: pairs[0] = (64-pixel chunk, channel) pairs it executed,
 * pairs[1] = pairs it looked at, summed over the launches on the context's first lane since the last reset.
 * executed / looked-at x the dense FLOP count = the FLOPs the matrix pipe really ran (rounds 1-4: bench.py's roofline.frac).
 * GPD_LENET_F32_CHAIN only — the split conv1 executes every tile and counts nothing (both numbers stay 0).
 * Measurement only: no reference counterpart. */
int gpd_hip_conv1_stats(gpd_hip_ctx *ctx, unsigned long long pairs[2], int reset);
/* test hook: intermediate tensors of the last gpd_hip_score pass (n images) — which = 0: pool1 f32 [n][15680] (layout of the
 * mode), 1: the three bf16 planes of the flattened pool2 [3][n][7200] (GPD_LENET_SPLIT; un-blocked on the host), 2: ip1 after ReLU, transposed f32 [500][n]
 * (GPD_LENET_SPLIT: the four K-quarter partial sums added on the host as ip2's kernel adds them) */
int gpd_hip_lenet_debug(gpd_hip_ctx *ctx, int which, int n, void *out);
/* test hook, host only: the operand tables of the split path's conv kernels as uploaded — atab: conv1's int8 digit
 * fragments [7][5][64][16], corr / shift [20], btab: conv2's bf16 fragments [4 slots][3][16][64][8] (slots 0-2: filters 16 slot + lane % 16; slot 3, k-steps 0-3: the
 * (filter, kernel column) rows of filters 48 and 49) */
int gpd_hip_lenet_fast_tables(int channels, const float *conv1_w, const float *conv2_w, uint8_t *atab, double *corr, int *shift,
                              unsigned short *btab);

#ifdef __cplusplus
}
#endif
#endif /* GPD_HIP_H_ */
