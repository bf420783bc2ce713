// gpd::Clustering — the reference's grasp clustering (include/gpd/clustering.h:49-78,
// src/gpd/clustering.cpp:5-105), step 6 of GraspDetector::detectGrasps (grasp_detector.cpp:283-303;
// SURVEY §8f rank 3): for every hand, the other hands whose axis is within 12 degrees, whose
// position is within 5 cm and within 5 mm of this hand's axis line are its inliers; a hand with
// >= min_inliers of them is emitted at the inliers' mean position with the lower 99 % confidence
// bound of their scores.  The pair tests and the running sums run on the device
// (gpd_hip_find_clusters, gpd_amd/csrc/cluster.hip); this class marshals the hands.
#pragma once
#include <memory>
#include <vector>

#include "gpd/candidate/hand.h"

namespace gpd {

class Clustering {
 public:
  explicit Clustering(int min_inliers) : min_inliers_(min_inliers) {}
  ~Clustering();
  Clustering(const Clustering &) = delete;
  Clustering &operator=(const Clustering &) = delete;
  // the device context to run on (GraspDetector hands over its own); without one the first call creates a context
  // on device 0 and keeps it
  void setContext(gpd_hip_ctx *ctx) { ctx_ = ctx; }
  std::vector<std::unique_ptr<candidate::Hand>> findClusters(const std::vector<std::unique_ptr<candidate::Hand>> &hand_list,
                                                              bool remove_inliers = false);
  int getMinInliers() const { return min_inliers_; }
  void setMinInliers(int m) { min_inliers_ = m; }
  // true when the last findClusters call could not run on the device (no context, HIP error): its empty result is then
  // a failure, not "no clusters" — callers must not mistake it for the reference's "fewer than 4 clusters" case
  bool failed() const { return failed_; }

 private:
  int min_inliers_;
  gpd_hip_ctx *ctx_ = nullptr, *own_ctx_ = nullptr;
  bool failed_ = false;
};

}  // namespace gpd
