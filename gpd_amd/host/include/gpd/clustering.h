// gpd::Clustering — the reference's grasp clustering (include/gpd/clustering.h:49-78,
// src/gpd/clustering.cpp:5-105), step 6 of GraspDetector::detectGrasps (grasp_detector.cpp:283-303;
// SURVEY §8f rank 3).  O(k^2) over the <= num_selected hands that survive selectGrasps, so it
// stays on the host: for every hand, the other hands whose axis is within 12 degrees, whose
// position is within 5 cm and within 5 mm of this hand's axis line are its inliers; a hand with
// >= min_inliers of them is emitted at the inliers' mean position with the lower 99 % confidence
// bound of their scores.
#pragma once
#include <memory>
#include <vector>

#include "gpd/candidate/hand.h"

namespace gpd {

class Clustering {
 public:
  explicit Clustering(int min_inliers) : min_inliers_(min_inliers) {}
  std::vector<std::unique_ptr<candidate::Hand>> findClusters(const std::vector<std::unique_ptr<candidate::Hand>> &hand_list,
                                                              bool remove_inliers = false);
  int getMinInliers() const { return min_inliers_; }
  void setMinInliers(int m) { min_inliers_ = m; }

 private:
  int min_inliers_;
};

}  // namespace gpd
