// ref_fixture — fixture generator that runs the REFERENCE's own classes (atenpas/gpd: candidate/, descriptor/,
// net/, util/) on a given cloud and dumps what the oracle is compared with.  TEST INFRASTRUCTURE: built only by
// oracle/build_ref.sh, from the reference sources where they lie, into oracle/_ref/ — and only on a machine that has
// the reference's dependencies (Eigen 3, PCL >= 1.9, OpenCV >= 3.4).  This container has none of them, so the
// program has never been compiled here; it follows the call sequence of the reference's own
// src/tests/test_grasp_image.cpp:19-171 and src/gpd/grasp_detector.cpp:222-273.
//
// usage: ref_fixture CLOUD.pcd NORMALS.f32 SAMPLES.i32 PARAMS_DIR/ CHANNELS OUT.bin [num_orientations]
//   NORMALS.f32  N x 3 float32 (one per cloud point, in PCD order); SAMPLES.i32 int32 sample indices;
//   PARAMS_DIR/  the LeNet parameter files of models/lenet/15channels/params (with an ip1_weights.bin).
// OUT.bin (little endian): "GPDREF1\0", int32 n_sets, n_slots, n_images, channels; n_sets * n_slots records of
//   176 bytes laid out as include/gpd_hip.h `gpd_hand`; n_images int32 hand indices (set * n_slots + slot);
//   n_images * 3600 * channels image bytes (cv::Mat HWC); n_images float32 scores (15 channels only: the
//   reference's EigenClassifier is hard-wired to 15, eigen_classifier.cpp:12-13).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include <gpd/candidate/candidates_generator.h>
#include <gpd/candidate/hand.h>
#include <gpd/candidate/hand_geometry.h>
#include <gpd/descriptor/image_generator.h>
#include <gpd/net/classifier.h>
#include <gpd/util/cloud.h>

#include "../include/gpd_hip.h"

template <class T>
static std::vector<T> read_all(const char *path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<T> out(raw.size() / sizeof(T));
  std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
  return out;
}

int main(int argc, char *argv[]) {
  if (argc < 7) {
    std::cout << "usage: ref_fixture CLOUD.pcd NORMALS.f32 SAMPLES.i32 PARAMS_DIR/ CHANNELS OUT.bin [num_orientations]\n";
    return 2;
  }
  using namespace gpd;
  Eigen::Matrix3Xd view_points(3, 1);
  view_points.setZero();
  util::Cloud cloud(argv[1], view_points);
  const int P = (int)cloud.getCloudOriginal()->size();
  std::vector<float> nrm = read_all<float>(argv[2]);
  std::vector<int32_t> samples = read_all<int32_t>(argv[3]);
  if (P == 0 || (int)nrm.size() != 3 * P || samples.empty()) {
    std::cout << "ERROR: cloud / normals / samples do not fit together\n";
    return 1;
  }
  Eigen::Matrix3Xd normals(3, P);
  for (int i = 0; i < P; i++)
    for (int r = 0; r < 3; r++) normals(r, i) = (double)nrm[3 * (size_t)i + r];
  cloud.setNormals(normals);
  cloud.setSampleIndices(std::vector<int>(samples.begin(), samples.end()));
  const int channels = std::stoi(argv[5]);
  const int num_orientations = argc > 7 ? std::stoi(argv[7]) : 8;

  // cfg/eigen_params.cfg + cfg/hand_geometry.cfg + cfg/image_geometry_15channels.cfg
  candidate::CandidatesGenerator::Parameters generator_params;
  candidate::HandSearch::Parameters hand_search_params;
  candidate::HandGeometry hand_geom;
  hand_geom.finger_width_ = 0.01;
  hand_geom.outer_diameter_ = 0.12;
  hand_geom.depth_ = 0.06;
  hand_geom.height_ = 0.02;
  hand_geom.init_bite_ = 0.01;
  hand_search_params.hand_geometry_ = hand_geom;
  descriptor::ImageGeometry image_geom;
  image_geom.outer_diameter_ = 0.10;
  image_geom.depth_ = 0.06;
  image_geom.height_ = 0.02;
  image_geom.size_ = 60;
  image_geom.num_channels_ = channels;
  hand_search_params.num_samples_ = (int)samples.size();
  hand_search_params.num_threads_ = 1;  // single-thread semantics are the oracle's
  hand_search_params.nn_radius_frames_ = 0.01;
  hand_search_params.num_orientations_ = num_orientations;
  hand_search_params.num_finger_placements_ = 10;
  hand_search_params.deepen_hand_ = true;
  hand_search_params.friction_coeff_ = 20.0;
  hand_search_params.min_viable_ = 6;
  hand_search_params.hand_axes_ = {2};
  generator_params.num_samples_ = hand_search_params.num_samples_;
  generator_params.num_threads_ = 1;
  generator_params.remove_statistical_outliers_ = false;
  generator_params.voxelize_ = false;
  generator_params.workspace_ = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};

  candidate::CandidatesGenerator candidates_generator(generator_params, hand_search_params);
  std::vector<std::unique_ptr<candidate::HandSet>> hand_set_list = candidates_generator.generateGraspCandidateSets(cloud);
  const int n_sets = (int)hand_set_list.size();
  const int n_slots = num_orientations * (int)hand_search_params.hand_axes_.size();
  std::vector<gpd_hand> recs((size_t)n_sets * n_slots);
  std::memset(recs.data(), 0, recs.size() * sizeof(gpd_hand));
  for (int s = 0; s < n_sets; s++) {
    const auto &hands = hand_set_list[s]->getHands();
    const auto &valid = hand_set_list[s]->getIsValid();
    for (int j = 0; j < n_slots && j < (int)hands.size(); j++) {
      const candidate::Hand &h = *hands[j];
      gpd_hand &r = recs[(size_t)s * n_slots + j];
      for (int k = 0; k < 3; k++) {
        r.sample[k] = h.getSample()(k);
        r.position[k] = h.getPosition()(k);
        for (int c = 0; c < 3; c++) r.frame[3 * k + c] = h.getFrame()(k, c);
      }
      r.top = h.getTop();
      r.bottom = h.getBottom();
      r.center = h.getCenter();
      r.grasp_width = h.getGraspWidth();
      r.finger_placement_index = h.getFingerPlacementIndex();
      r.set_index = s;
      r.slot = j;
      r.valid = valid(j) ? 1 : 0;
      r.half_antipodal = h.isHalfAntipodal() ? 1 : 0;
      r.full_antipodal = h.isFullAntipodal() ? 1 : 0;
    }
  }
  // the images of every valid hand, in createImageList's order (image_generator.cpp:91-98): no workspace filter
  // here, the test applies the oracle's filter to both sides when it needs one
  std::vector<int32_t> cand;
  for (int s = 0; s < n_sets; s++)
    for (int j = 0; j < n_slots; j++)
      if (recs[(size_t)s * n_slots + j].valid) cand.push_back(s * n_slots + j);
  descriptor::ImageGenerator image_generator(image_geom, 1, num_orientations, false, false);
  std::vector<std::unique_ptr<cv::Mat>> images;
  std::vector<std::unique_ptr<candidate::Hand>> hands_out;
  image_generator.createImages(cloud, hand_set_list, images, hands_out);
  if (images.size() != cand.size()) {
    std::cout << "ERROR: " << images.size() << " images for " << cand.size() << " valid hands\n";
    return 1;
  }
  std::vector<float> scores(images.size(), 0.f);
  if (channels == 15 && !images.empty()) {
    std::shared_ptr<net::Classifier> classifier = net::Classifier::create("", argv[4], net::Classifier::Device::eCPU, 1);
    scores = classifier->classifyImages(images);
  }
  FILE *f = fopen(argv[6], "wb");
  if (!f) return 1;
  const char magic[8] = {'G', 'P', 'D', 'R', 'E', 'F', '1', 0};
  const int32_t head[4] = {n_sets, n_slots, (int32_t)images.size(), channels};
  fwrite(magic, 1, 8, f);
  fwrite(head, sizeof(int32_t), 4, f);
  fwrite(recs.data(), sizeof(gpd_hand), recs.size(), f);
  fwrite(cand.data(), sizeof(int32_t), cand.size(), f);
  for (const auto &im : images) {
    if (!im->isContinuous() || im->rows != 60 || im->cols != 60 || im->channels() != channels) {
      std::cout << "ERROR: unexpected image layout\n";
      return 1;
    }
    fwrite(im->data, 1, (size_t)3600 * channels, f);
  }
  fwrite(scores.data(), sizeof(float), scores.size(), f);
  fclose(f);
  std::cout << "ref_fixture: " << n_sets << " hand sets, " << images.size() << " images -> " << argv[6] << "\n";
  return 0;
}
