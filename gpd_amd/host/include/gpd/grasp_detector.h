// gpd::GraspDetector — the reference's public API (include/gpd/grasp_detector.h:66-186) over the
// HIP path.  Same constructor (a cfg file), same call sequence in detectGrasps
// (grasp_detector.cpp:192-328): candidates -> workspace/aperture filter [-> approach-direction
// filter] -> images -> classify -> select top num_selected -> cluster (min_inliers > 0) -> sort.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "gpd/candidate/hand.h"
#include "gpd/clustering.h"
#include "gpd/net/classifier.h"
#include "gpd/util/cloud.h"
#include "gpd_hip.h"

namespace gpd {

namespace candidate {
// HandGeometry (candidate/hand_geometry.h:50-72) and HandSearch::Parameters (candidate/hand_search.h:76-94)
struct HandGeometry {
  double finger_width_, outer_diameter_, depth_, height_, init_bite_;
};
struct HandSearch {
  struct Parameters {
    double nn_radius_frames_;
    int num_threads_, num_samples_, num_orientations_, num_finger_placements_;
    std::vector<int> hand_axes_;
    bool deepen_hand_;
    double friction_coeff_;
    int min_viable_;
    HandGeometry hand_geometry_;
  };
};
}  // namespace candidate
namespace descriptor {
// ImageGeometry (descriptor/image_geometry.h:50-78)
struct ImageGeometry {
  double outer_diameter_, depth_, height_;
  int size_, num_channels_;
};
}  // namespace descriptor

class GraspDetector {
 public:
  explicit GraspDetector(const std::string &config_filename);
  // From the parameter objects the reference hands to CandidatesGenerator / HandSearch and ImageGenerator
  // directly (src/tests/test_grasp_image.cpp:50-108): no classifier, no voxelisation, workspaces +-1 m.
  GraspDetector(const candidate::HandSearch::Parameters &hand_search_params, const descriptor::ImageGeometry &image_geom,
                int hip_device = 0);
  ~GraspDetector();
  std::vector<std::unique_ptr<candidate::Hand>> detectGrasps(const util::Cloud &cloud);
  // CandidatesGenerator::preprocessPointCloud (candidates_generator.cpp:14-37): workspace cut (cfg
  // workspace), voxelise (voxelize/voxel_size), normals recomputed on the GPU (normals_radius), subsample;
  // cfg use_file_normals = 1 keeps the normals a PCD came with (and then does not voxelise).
  void preprocessPointCloud(util::Cloud &cloud);
  // Cloud::calculateNormals (cloud.cpp:451-476) on the device: radius PCA, flipped to the view points,
  // reverseNormals; replaces the cloud's normals.  False when the device call fails.
  bool calculateNormals(util::Cloud &cloud, double radius);
  std::vector<std::unique_ptr<candidate::HandSet>> generateGraspCandidates(const util::Cloud &cloud);
  std::vector<std::unique_ptr<candidate::HandSet>> filterGraspsWorkspace(
      std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, const std::vector<double> &workspace) const;
  // grasp_detector.cpp:422-453: clears hands whose approach axis is more than thresh_rad away from
  // `direction`; sets left without a valid hand are dropped.
  std::vector<std::unique_ptr<candidate::HandSet>> filterGraspsDirection(
      std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, const std::array<double, 3> &direction, double thresh_rad);
  // grasp_detector.cpp:528-551: images + scores for the given sets, keeps the hands with
  // score > min_score.  Sets of the last generateGraspCandidates use the neighbourhoods still on
  // the device; any other list (sets collected over several searches, as the importance sampler
  // does) has its neighbourhoods searched again from the sets' samples.
  std::vector<std::unique_ptr<candidate::Hand>> pruneGraspCandidates(
      const util::Cloud &cloud, const std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, double min_score);
  // grasp_detector.cpp:522-526 -> HandSearch::reevaluateHypotheses (hand_search.cpp:66-134): labels
  // (1 = full antipodal on cloud_gt) and rewritten half/full flags of `hands`.  Uploads cloud_gt.
  std::vector<int> evalGroundTruth(const util::Cloud &cloud_gt, std::vector<std::unique_ptr<candidate::Hand>> &hands);
  // ImageGenerator::createImages (descriptor/image_generator.cpp:17-99): one image per valid hand of the
  // given sets, set-major / slot-minor, no filtering.  The sets must come from the last
  // generateGraspCandidates on this cloud (their neighbourhoods are still on the device).
  bool createImages(const util::Cloud &cloud, const std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list,
                    std::vector<std::unique_ptr<net::Image>> &images_out, std::vector<std::unique_ptr<candidate::Hand>> &hands_out);
  bool createGraspImages(util::Cloud &cloud, std::vector<std::unique_ptr<candidate::Hand>> &hands_out,
                         std::vector<std::unique_ptr<net::Image>> &images_out);
  std::vector<std::unique_ptr<candidate::Hand>> selectGrasps(std::vector<std::unique_ptr<candidate::Hand>> &hands) const;
  static bool isScoreGreater(const std::unique_ptr<candidate::Hand> &a, const std::unique_ptr<candidate::Hand> &b) {
    return a->getScore() > b->getScore();
  }
  const gpd_params &getParams() const { return params_; }
  // grasp_detector.h:176-186
  candidate::HandSearch::Parameters getHandSearchParameters() const;
  descriptor::ImageGeometry getImageGeometry() const {
    return {params_.volume_width, params_.volume_depth, params_.volume_height, params_.image_size, params_.image_num_channels};
  }
  const std::vector<double> &getWorkspaceGrasps() const { return workspace_grasps_; }
  bool ok() const { return ctx_ != nullptr; }
  gpd_hip_ctx *context() const { return ctx_; }  // for the objects that run on the same device context (Clustering)
  // stage runtimes of the last detectGrasps, seconds: candidates, images, classification, total
  const double *lastRuntimes() const { return runtimes_; }

 private:
  bool upload(const util::Cloud &cloud);
  // device search for the cloud's samples (coordinates if set, else indices); recs sized here
  bool searchDevice(const util::Cloud &cloud, bool fused, std::vector<gpd_hand> &recs, int &n_sets, int &n_cand);
  std::vector<gpd_hand> flatten(const std::vector<std::unique_ptr<candidate::HandSet>> &sets) const;
  gpd_params params_;
  gpd_hip_ctx *ctx_ = nullptr;
  std::shared_ptr<net::Classifier> classifier_;  // grasp_detector.h: classifier_, made by net::Classifier::create
  bool has_classifier_ = false;
  bool fused_classifier_ = false;  // the classifier is the HIP back-end and its parameters are loaded in ctx_
  bool plugin_route_ = false;      // cfg classifier_plugin_route: score through Classifier::classifyImages
  int num_samples_ = 1000;
  bool voxelize_ = true;
  bool use_file_normals_ = false;  // cfg use_file_normals: keep normal_x/y/z of the PCD instead of recomputing them
  double voxel_size_ = 0.003, normals_radius_ = 0.03;
  int num_selected_ = 100;
  std::vector<double> workspace_grasps_;
  std::vector<double> workspace_;  // cfg `workspace`: the cloud is cut to it in preprocessPointCloud
  bool filter_approach_direction_ = false;
  std::array<double, 3> direction_ = {1, 0, 0};
  double thresh_rad_ = 2.3;
  std::unique_ptr<Clustering> clustering_;
  bool cluster_grasps_ = false;
  int last_num_sets_ = 0;  // hand sets of the last device search
  std::vector<double> last_samples_;  // their samples, [set][3]
  double runtimes_[4] = {0, 0, 0, 0};
};

}  // namespace gpd
