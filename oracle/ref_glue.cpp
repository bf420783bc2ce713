// oracle/ref_glue.cpp — C entry points over the REFERENCE's own classes (atenpas/gpd), for oracle/_ref/libgpd_ref.so.
// TEST INFRASTRUCTURE ONLY: built by oracle/build_ref.sh together with the reference's translation units, which are
// compiled UNMODIFIED from /root/reference through the test-only headers of oracle/shim/ (Eigen / PCL / OpenCV / Boost
// subsets: read their headers for what is restated from memory).  Only tests/ and the fixture generator
// tests/golden/make_ref_pins.py load the library.  Nothing here restates the reference: every function below builds the
// reference's objects (util::Cloud, GraspDetector, CandidatesGenerator, descriptor::ImageGenerator, net::Classifier,
// Clustering) through their public interfaces and copies results into the POD records of include/gpd_hip.h.
//
// The reference's OpenMP pragmas are NOT enabled in this build (no -fopenmp): its parallel loops write shared state
// (EigenClassifier's member buffers, HandSet::seed_) and the single-thread order is the definition both
// the oracle and the product follow.  hand_set.cpp is compiled behind `-include oracle/shim/gpd_ref_no_jitter.h`, which renames
// `normal_distribution` to a stand-in returning 0, so that the shadow points carry no Gaussian jitter (the reference
// seeds it from std::random_device: irreproducible, hand_set.cpp:191-199); that is the one deviation from "unmodified".
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <gpd/candidate/candidates_generator.h>
#include <gpd/candidate/finger_hand.h>
#include <gpd/candidate/hand.h>
#include <gpd/candidate/hand_set.h>
#include <gpd/clustering.h>
#include <gpd/descriptor/image_generator.h>
#include <gpd/grasp_detector.h>
#include <gpd/net/classifier.h>
#include <gpd/net/conv_layer.h>
#include <gpd/util/cloud.h>

#include "../include/gpd_hip.h"

using namespace gpd;

namespace {

struct Ref {
  std::unique_ptr<GraspDetector> det;
  std::unique_ptr<descriptor::ImageGenerator> imgen;
  std::shared_ptr<net::Classifier> classifier;
  std::vector<std::unique_ptr<candidate::HandSet>> sets;  // the current hand set list (after generate / the filters)
  std::vector<int> origin;                                // for every current set: its number in generate's list
  std::vector<std::unique_ptr<candidate::Hand>> hands;    // hands moved out by createImages, in image order
  int n_slots = 0;
};

void fillRecord(const candidate::Hand &h, int set, int slot, bool valid, gpd_hand &r) {
  std::memset(&r, 0, sizeof(r));
  for (int k = 0; k < 3; k++) {
    r.sample[k] = h.getSample()(k);
    r.position[k] = h.getPosition()(k);
    for (int c = 0; c < 3; c++) r.frame[3 * k + c] = h.getFrame()(k, c);
  }
  r.top = h.getTop();
  r.bottom = h.getBottom();
  r.center = h.getCenter();
  r.grasp_width = h.getGraspWidth();
  r.score = (float)h.getScore();
  r.finger_placement_index = valid ? h.getFingerPlacementIndex() : 0;  // uninitialised in the reference when no placement exists
  r.set_index = set;
  r.slot = slot;
  r.valid = valid ? 1 : 0;
  r.half_antipodal = h.isHalfAntipodal() ? 1 : 0;
  r.full_antipodal = h.isFullAntipodal() ? 1 : 0;
}

// A candidate::Hand with the geometry of a record, through the public interface only: a FingerHand whose placement
// `finger_placement_index` is feasible (one far-away point keeps every gap free), then the closing box, then Hand's own
// constructor; the position is overwritten with the record's (it equals what construct() computes up to the record's bits).
std::unique_ptr<candidate::Hand> handFromRecord(const gpd_hand &r, const candidate::HandGeometry &g, int num_placements) {
  candidate::FingerHand fh(g.finger_width_, g.outer_diameter_, g.depth_, num_placements);
  fh.setForwardAxis(0);
  fh.setLateralAxis(1);
  Eigen::Matrix3Xd far_point(3, 1);
  far_point << 0.0, 1.0e6, 0.0;
  const int idx = r.finger_placement_index;
  fh.evaluateFingers(far_point, r.top, idx);
  fh.evaluateHand(idx);
  fh.setTop(r.top);
  fh.setBottom(r.bottom);
  fh.setCenter(r.center);
  Eigen::Vector3d sample(r.sample[0], r.sample[1], r.sample[2]);
  Eigen::Matrix3d frame;
  for (int k = 0; k < 3; k++)
    for (int c = 0; c < 3; c++) frame(k, c) = r.frame[3 * k + c];
  std::unique_ptr<candidate::Hand> h = std::make_unique<candidate::Hand>(sample, frame, fh, r.grasp_width);
  h->setPosition(Eigen::Vector3d(r.position[0], r.position[1], r.position[2]));
  h->setScore((double)r.score);
  h->setHalfAntipodal(r.half_antipodal != 0);
  h->setFullAntipodal(r.full_antipodal != 0);
  return h;
}

void writeValid(Ref *r, int n_sets_orig, uint8_t *valid_out) {
  std::memset(valid_out, 0, (size_t)n_sets_orig * r->n_slots);
  for (size_t s = 0; s < r->sets.size(); s++) {
    const auto &valid = r->sets[s]->getIsValid();
    for (int j = 0; j < r->n_slots && j < valid.size(); j++) valid_out[(size_t)r->origin[s] * r->n_slots + j] = valid(j) ? 1 : 0;
  }
}
template <class F>
void refilter(Ref *r, F f) {
  std::map<const candidate::HandSet *, int> where;
  for (size_t s = 0; s < r->sets.size(); s++) where[r->sets[s].get()] = r->origin[s];
  std::vector<std::unique_ptr<candidate::HandSet>> out = f(r->sets);
  std::vector<int> origin(out.size());
  for (size_t s = 0; s < out.size(); s++) origin[s] = where[out[s].get()];
  r->sets = std::move(out);
  r->origin = origin;
}
}  // namespace

extern "C" {

int gpd_ref_abi() { return (int)sizeof(gpd_hand); }

void gpd_ref_set_product_mode(int mode) { Eigen::shim::product_mode() = mode; }

// HandSet::seed_ (hand_set.cpp:14, 263-266) is a private static that keeps counting across detectGrasps calls of one
// process; the oracle and the product restart the shadow LCG at 0 for every cloud.  This file — and
// only this file — is compiled with -fno-access-control so that a test can put the reference into the same state.
void gpd_ref_reset_shadow_seed() { candidate::HandSet::seed_ = 0; }

// ---- detector (GraspDetector::GraspDetector, grasp_detector.cpp:5-190) ----------------------------------------
void *gpd_ref_create(const char *cfg_path) {
  Ref *r = new Ref;
  r->det = std::make_unique<GraspDetector>(std::string(cfg_path));
  const candidate::HandSearch::Parameters &hs = r->det->getHandSearchParameters();
  r->n_slots = (int)hs.hand_axes_.size() * hs.num_orientations_;
  // the detector keeps its ImageGenerator private; the same constructor call as grasp_detector.cpp:152-154
  r->imgen = std::make_unique<descriptor::ImageGenerator>(r->det->getImageGeometry(), hs.num_threads_, hs.num_orientations_, false, false);
  return r;
}
void gpd_ref_destroy(void *h) { delete static_cast<Ref *>(h); }
int gpd_ref_num_slots(void *h) { return static_cast<Ref *>(h)->n_slots; }

// ---- cloud (util::Cloud) -----------------------------------------------------------------------------------------
void *gpd_ref_cloud_create(const float *xyz, int P, const float *normals, const int32_t *cam_source, int n_cams, const double *view_points) {
  util::PointCloudRGB::Ptr pc(new util::PointCloudRGB);
  pc->points.resize((size_t)P);
  pc->width = (uint32_t)P;
  pc->height = 1;
  for (int i = 0; i < P; i++) {
    pc->points[(size_t)i].x = xyz[3 * i];
    pc->points[(size_t)i].y = xyz[3 * i + 1];
    pc->points[(size_t)i].z = xyz[3 * i + 2];
  }
  Eigen::MatrixXi cams(n_cams, P);
  for (int c = 0; c < n_cams; c++)
    for (int i = 0; i < P; i++) cams(c, i) = cam_source[(size_t)c * P + i];
  Eigen::Matrix3Xd vp(3, n_cams);
  for (int c = 0; c < n_cams; c++)
    for (int k = 0; k < 3; k++) vp(k, c) = view_points[3 * c + k];
  util::Cloud *cl = new util::Cloud(pc, cams, vp);
  if (normals) {
    Eigen::Matrix3Xd n(3, P);
    for (int i = 0; i < P; i++)
      for (int k = 0; k < 3; k++) n(k, i) = (double)normals[3 * i + k];
    cl->setNormals(n);
  }
  return cl;
}
void *gpd_ref_cloud_load(const char *pcd_path) {
  Eigen::Matrix3Xd vp(3, 1);
  vp.setZero();
  return new util::Cloud(std::string(pcd_path), vp);
}
void gpd_ref_cloud_destroy(void *c) { delete static_cast<util::Cloud *>(c); }
void gpd_ref_cloud_set_sample_indices(void *c, const int32_t *idx, int S) {
  static_cast<util::Cloud *>(c)->setSampleIndices(std::vector<int>(idx, idx + S));
}
void gpd_ref_cloud_set_samples(void *c, const double *xyz, int S) {
  Eigen::Matrix3Xd s(3, S);
  for (int i = 0; i < S; i++)
    for (int k = 0; k < 3; k++) s(k, i) = xyz[3 * i + k];
  static_cast<util::Cloud *>(c)->setSamples(s);
}
int gpd_ref_cloud_size(void *c) { return (int)static_cast<util::Cloud *>(c)->getCloudProcessed()->size(); }
int gpd_ref_cloud_num_normals(void *c) { return (int)static_cast<util::Cloud *>(c)->getNormals().cols(); }
void gpd_ref_cloud_get(void *c, float *xyz, double *normals, int32_t *cam_source) {
  util::Cloud *cl = static_cast<util::Cloud *>(c);
  const int P = (int)cl->getCloudProcessed()->size();
  for (int i = 0; i < P && xyz; i++) {
    xyz[3 * i] = cl->getCloudProcessed()->points[(size_t)i].x;
    xyz[3 * i + 1] = cl->getCloudProcessed()->points[(size_t)i].y;
    xyz[3 * i + 2] = cl->getCloudProcessed()->points[(size_t)i].z;
  }
  if (normals)
    for (int i = 0; i < (int)cl->getNormals().cols(); i++)
      for (int k = 0; k < 3; k++) normals[3 * i + k] = cl->getNormals()(k, i);
  if (cam_source)
    for (int cam = 0; cam < (int)cl->getCameraSource().rows(); cam++)
      for (int i = 0; i < P; i++) cam_source[(size_t)cam * P + i] = cl->getCameraSource()(cam, i);
}
int gpd_ref_cloud_sample_indices(void *c, int32_t *out, int cap) {
  const std::vector<int> &s = static_cast<util::Cloud *>(c)->getSampleIndices();
  for (int i = 0; i < (int)s.size() && i < cap; i++) out[i] = s[(size_t)i];
  return (int)s.size();
}
// Cloud::filterWorkspace (cloud.cpp:206-267), Cloud::voxelizeCloud (:286-348), Cloud::calculateNormals (:451-476)
void gpd_ref_cloud_filter_workspace(void *c, const double *ws) { static_cast<util::Cloud *>(c)->filterWorkspace(std::vector<double>(ws, ws + 6)); }
void gpd_ref_cloud_voxelize(void *c, float cell) { static_cast<util::Cloud *>(c)->voxelizeCloud(cell); }
void gpd_ref_cloud_calculate_normals(void *c, double radius) { static_cast<util::Cloud *>(c)->calculateNormals(1, radius); }
// GraspDetector::preprocessPointCloud (grasp_detector.cpp:330-332 -> candidates_generator.cpp:15-40)
void gpd_ref_preprocess(void *h, void *c) { static_cast<Ref *>(h)->det->preprocessPointCloud(*static_cast<util::Cloud *>(c)); }

// ---- candidate search (GraspDetector::generateGraspCandidates -> HandSearch::searchHands) ---------------------
int gpd_ref_generate(void *h, void *c, gpd_hand *hands, int cap_sets, int *n_sets) {
  Ref *r = static_cast<Ref *>(h);
  r->sets = r->det->generateGraspCandidates(*static_cast<util::Cloud *>(c));
  r->hands.clear();
  r->origin.resize(r->sets.size());
  *n_sets = (int)r->sets.size();
  for (int s = 0; s < (int)r->sets.size(); s++) {
    r->origin[(size_t)s] = s;
    if (s >= cap_sets) continue;
    const auto &hs = r->sets[(size_t)s]->getHands();
    const auto &valid = r->sets[(size_t)s]->getIsValid();
    for (int j = 0; j < r->n_slots; j++) {
      gpd_hand &rec = hands[(size_t)s * r->n_slots + j];
      if (j < (int)hs.size() && hs[(size_t)j])
        fillRecord(*hs[(size_t)j], s, j, j < valid.size() && valid(j), rec);
      else
        std::memset(&rec, 0, sizeof(rec));
    }
  }
  return 0;
}

// GraspDetector::filterGraspsWorkspace (grasp_detector.cpp:334-398) on the current list; valid_out [n_sets_orig][n_slots]
int gpd_ref_filter_workspace(void *h, const double *ws, int n_sets_orig, uint8_t *valid_out) {
  Ref *r = static_cast<Ref *>(h);
  const std::vector<double> w(ws, ws + 6);
  refilter(r, [&](std::vector<std::unique_ptr<candidate::HandSet>> &l) { return r->det->filterGraspsWorkspace(l, w); });
  writeValid(r, n_sets_orig, valid_out);
  return (int)r->sets.size();
}
// GraspDetector::filterGraspsDirection (grasp_detector.cpp:422-453)
int gpd_ref_filter_direction(void *h, const double *dir, double thresh_rad, int n_sets_orig, uint8_t *valid_out) {
  Ref *r = static_cast<Ref *>(h);
  const Eigen::Vector3d d(dir[0], dir[1], dir[2]);
  refilter(r, [&](std::vector<std::unique_ptr<candidate::HandSet>> &l) { return r->det->filterGraspsDirection(l, d, thresh_rad); });
  writeValid(r, n_sets_orig, valid_out);
  return (int)r->sets.size();
}

// ---- images (ImageGenerator::createImages, image_generator.cpp:17-99) on the current list ----------------------
// images: [cap][size][size][channels] u8 (cv::Mat HWC); cand: set (generate's numbering) * n_slots + slot per image
int gpd_ref_images(void *h, void *c, uint8_t *images, int32_t *cand, int cap, int *n_out) {
  Ref *r = static_cast<Ref *>(h);
  std::vector<int32_t> order;
  for (size_t s = 0; s < r->sets.size(); s++) {
    const auto &valid = r->sets[s]->getIsValid();
    for (int j = 0; j < (int)r->sets[s]->getHands().size(); j++)
      if (valid(j)) order.push_back(r->origin[s] * r->n_slots + j);
  }
  std::vector<std::unique_ptr<cv::Mat>> imgs;
  r->hands.clear();
  if (!r->sets.empty()) r->imgen->createImages(*static_cast<util::Cloud *>(c), r->sets, imgs, r->hands);
  *n_out = (int)imgs.size();
  if (imgs.size() != order.size()) return -1;
  const descriptor::ImageGeometry &g = r->det->getImageGeometry();
  const size_t bytes = (size_t)g.size_ * g.size_ * g.num_channels_;
  for (int i = 0; i < (int)imgs.size() && i < cap; i++) {
    const cv::Mat &m = *imgs[(size_t)i];
    if (!m.isContinuous() || m.rows != g.size_ || m.cols != g.size_ || m.channels() != g.num_channels_ || m.depth() != CV_8U) return -2;
    std::memcpy(images + (size_t)i * bytes, m.data, bytes);
    cand[i] = order[(size_t)i];
  }
  return 0;
}

// ---- classifier (Classifier::create + EigenClassifier::classifyImages, eigen_classifier.cpp:6-79) ----------------
int gpd_ref_classifier_load(void *h, const char *weights_dir) {
  Ref *r = static_cast<Ref *>(h);
  r->classifier = net::Classifier::create("", std::string(weights_dir), net::Classifier::Device::eCPU, 1);
  return r->classifier ? 0 : -1;
}
int gpd_ref_classify(void *h, const uint8_t *images, int n, int size, int channels, float *scores) {
  Ref *r = static_cast<Ref *>(h);
  if (!r->classifier) return -1;
  std::vector<std::unique_ptr<cv::Mat>> list;
  const size_t bytes = (size_t)size * size * channels;
  for (int i = 0; i < n; i++) {
    list.push_back(std::make_unique<cv::Mat>(size, size, CV_8UC(channels)));
    std::memcpy(list.back()->data, images + (size_t)i * bytes, bytes);
  }
  const std::vector<float> s = r->classifier->classifyImages(list);
  for (int i = 0; i < n; i++) scores[i] = s[(size_t)i];
  return 0;
}
// net::ConvLayer::forward on its own (conv_layer.cpp:26-98): x CHW float, w [F][C*K*K], out [F][(H-K+1)*(W-K+1)]
void gpd_ref_conv_forward(const float *x, int C, int H, int W, const float *w, const float *b, int F, int K, float *out) {
  net::ConvLayer layer(W, H, C, F, K, 1, 0);
  layer.setWeightsAndBiases(std::vector<float>(w, w + (size_t)F * C * K * K), std::vector<float>(b, b + F));
  const Eigen::MatrixXf y = layer.forward(std::vector<float>(x, x + (size_t)C * H * W));
  for (int f = 0; f < F; f++)
    for (int p = 0; p < (int)y.cols(); p++) out[(size_t)f * y.cols() + p] = y(f, p);
}

// ---- selection, clustering, re-evaluation over records --------------------------------------------------------------
// GraspDetector::selectGrasps (grasp_detector.cpp:405-420) with the cfg's num_selected: indices of the kept hands
int gpd_ref_select(void *h, const float *scores, int n, int32_t *out_idx) {
  Ref *r = static_cast<Ref *>(h);
  std::vector<std::unique_ptr<candidate::Hand>> hands;
  for (int i = 0; i < n; i++) {
    hands.push_back(std::make_unique<candidate::Hand>());
    hands.back()->setScore((double)scores[i]);
    hands.back()->setGraspWidth((double)i);  // the tag that survives the moves
  }
  std::vector<std::unique_ptr<candidate::Hand>> kept = r->det->selectGrasps(hands);
  for (size_t i = 0; i < kept.size(); i++) out_idx[i] = (int32_t)kept[i]->getGraspWidth();
  return (int)kept.size();
}
// Clustering::findClusters (clustering.cpp:5-105); scores are doubles as Hand keeps them
int gpd_ref_find_clusters(void *h, const gpd_hand *recs, const double *scores, int n, int min_inliers, int remove_inliers, gpd_hand *out,
                          double *out_scores) {
  Ref *r = static_cast<Ref *>(h);
  const candidate::HandSearch::Parameters &hs = r->det->getHandSearchParameters();
  std::vector<std::unique_ptr<candidate::Hand>> hands;
  for (int i = 0; i < n; i++) {
    hands.push_back(handFromRecord(recs[i], hs.hand_geometry_, hs.num_finger_placements_));
    hands.back()->setScore(scores[i]);
  }
  Clustering clustering(min_inliers);
  std::vector<std::unique_ptr<candidate::Hand>> cl = clustering.findClusters(hands, remove_inliers != 0);
  for (size_t i = 0; i < cl.size(); i++) {
    fillRecord(*cl[i], 0, 0, true, out[i]);
    out_scores[i] = cl[i]->getScore();
  }
  return (int)cl.size();
}
// GraspDetector::evalGroundTruth -> HandSearch::reevaluateHypotheses (hand_search.cpp:66-134)
int gpd_ref_reevaluate(void *h, void *c, gpd_hand *recs, int n, int32_t *labels) {
  Ref *r = static_cast<Ref *>(h);
  const candidate::HandSearch::Parameters &hs = r->det->getHandSearchParameters();
  std::vector<std::unique_ptr<candidate::Hand>> hands;
  for (int i = 0; i < n; i++) hands.push_back(handFromRecord(recs[i], hs.hand_geometry_, hs.num_finger_placements_));
  const std::vector<int> lab = r->det->evalGroundTruth(*static_cast<util::Cloud *>(c), hands);
  for (int i = 0; i < n; i++) {
    labels[i] = lab[(size_t)i];
    recs[i].half_antipodal = hands[(size_t)i]->isHalfAntipodal() ? 1 : 0;
    recs[i].full_antipodal = hands[(size_t)i]->isFullAntipodal() ? 1 : 0;
  }
  return 0;
}

// ---- the whole thing: GraspDetector::detectGrasps (grasp_detector.cpp:192-328) -----------------------------------
int gpd_ref_detect(void *h, void *c, gpd_hand *out, int cap) {
  Ref *r = static_cast<Ref *>(h);
  std::vector<std::unique_ptr<candidate::Hand>> hands = r->det->detectGrasps(*static_cast<util::Cloud *>(c));
  for (int i = 0; i < (int)hands.size() && i < cap; i++) {
    fillRecord(*hands[(size_t)i], 0, 0, true, out[i]);
    out[i].score = (float)hands[(size_t)i]->getScore();
  }
  return (int)hands.size();
}

}  // extern "C"
