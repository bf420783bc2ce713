// gpd::GraspDetector — the reference's public API (include/gpd/grasp_detector.h:66-186) over the
// HIP path.  Same constructor (a cfg file), same call sequence in detectGrasps
// (grasp_detector.cpp:192-328): candidates -> workspace/aperture filter -> images -> classify ->
// select top num_selected -> sort; clustering (out of scope) is skipped like `min_inliers = 0`.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "gpd/candidate/hand.h"
#include "gpd/net/classifier.h"
#include "gpd/util/cloud.h"
#include "gpd_hip.h"

namespace gpd {

class GraspDetector {
 public:
  explicit GraspDetector(const std::string &config_filename);
  ~GraspDetector();
  std::vector<std::unique_ptr<candidate::Hand>> detectGrasps(const util::Cloud &cloud);
  // CandidatesGenerator::preprocessPointCloud (candidates_generator.cpp:14-37): voxelise (cfg
  // voxelize/voxel_size), normals on the GPU when the cloud has none (normals_radius), subsample.
  void preprocessPointCloud(util::Cloud &cloud);
  std::vector<std::unique_ptr<candidate::HandSet>> generateGraspCandidates(const util::Cloud &cloud);
  std::vector<std::unique_ptr<candidate::HandSet>> filterGraspsWorkspace(
      std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, const std::vector<double> &workspace) const;
  bool createGraspImages(util::Cloud &cloud, std::vector<std::unique_ptr<candidate::Hand>> &hands_out,
                         std::vector<std::unique_ptr<net::Image>> &images_out);
  std::vector<std::unique_ptr<candidate::Hand>> selectGrasps(std::vector<std::unique_ptr<candidate::Hand>> &hands) const;
  static bool isScoreGreater(const std::unique_ptr<candidate::Hand> &a, const std::unique_ptr<candidate::Hand> &b) {
    return a->getScore() > b->getScore();
  }
  const gpd_params &getParams() const { return params_; }
  bool ok() const { return ctx_ != nullptr; }
  // stage runtimes of the last detectGrasps, seconds: candidates, images, classification, total
  const double *lastRuntimes() const { return runtimes_; }

 private:
  bool upload(const util::Cloud &cloud);
  gpd_params params_;
  gpd_hip_ctx *ctx_ = nullptr;
  bool has_classifier_ = false;
  int num_samples_ = 1000;
  bool voxelize_ = true;
  double voxel_size_ = 0.003, normals_radius_ = 0.03;
  int num_selected_ = 100;
  std::vector<double> workspace_grasps_;
  double runtimes_[4] = {0, 0, 0, 0};
};

}  // namespace gpd
