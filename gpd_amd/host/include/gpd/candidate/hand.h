// candidate::Hand / HandSet — result records with the reference's accessors
// (candidate/hand.h:80-277, candidate/hand_set.h).  Storage is the POD gpd_hand of the C-ABI.
#pragma once
#include <array>
#include <memory>
#include <vector>

#include "gpd_hip.h"

namespace gpd {
namespace candidate {

class Hand {
 public:
  Hand() { rec_ = gpd_hand(); }
  explicit Hand(const gpd_hand &r) : rec_(r), score_(r.score) {}
  std::array<double, 3> getSample() const { return {rec_.sample[0], rec_.sample[1], rec_.sample[2]}; }
  std::array<double, 3> getPosition() const { return {rec_.position[0], rec_.position[1], rec_.position[2]}; }
  std::array<double, 9> getFrame() const {  // row-major, columns approach | binormal | axis
    std::array<double, 9> f;
    for (int i = 0; i < 9; i++) f[i] = rec_.frame[i];
    return f;
  }
  std::array<double, 9> getOrientation() const { return getFrame(); }
  std::array<double, 3> getApproach() const { return {rec_.frame[0], rec_.frame[3], rec_.frame[6]}; }
  std::array<double, 3> getBinormal() const { return {rec_.frame[1], rec_.frame[4], rec_.frame[7]}; }
  std::array<double, 3> getAxis() const { return {rec_.frame[2], rec_.frame[5], rec_.frame[8]}; }
  double getGraspWidth() const { return rec_.grasp_width; }
  // the reference keeps the score as a double (hand.h:150-156); the POD record holds the LeNet
  // float, the double survives Clustering's confidence bound (clustering.cpp:95)
  double getScore() const { return score_; }
  void setScore(double s) {
    score_ = s;
    rec_.score = (float)s;
  }
  void setPosition(const std::array<double, 3> &p) {
    for (int i = 0; i < 3; i++) rec_.position[i] = p[i];
  }
  void setFullAntipodal(bool b) { rec_.full_antipodal = b ? 1 : 0; }
  void setHalfAntipodal(bool b) { rec_.half_antipodal = b ? 1 : 0; }
  bool isFullAntipodal() const { return rec_.full_antipodal; }
  bool isHalfAntipodal() const { return rec_.half_antipodal; }
  double getTop() const { return rec_.top; }
  double getBottom() const { return rec_.bottom; }
  double getCenter() const { return rec_.center; }
  int getFingerPlacementIndex() const { return rec_.finger_placement_index; }
  const gpd_hand &record() const { return rec_; }
  void print() const;

 private:
  gpd_hand rec_;
  double score_ = 0.0;
};

// HandSet (candidate/hand_set.h): the hands of one sample + their validity flags.
class HandSet {
 public:
  const std::vector<std::unique_ptr<Hand>> &getHands() const { return hands_; }
  std::vector<std::unique_ptr<Hand>> &getHands() { return hands_; }
  const std::vector<bool> &getIsValid() const { return is_valid_; }
  void setIsValid(const std::vector<bool> &v) { is_valid_ = v; }
  std::array<double, 3> getSample() const { return sample_; }
  std::vector<std::unique_ptr<Hand>> hands_;
  std::vector<bool> is_valid_;
  std::array<double, 3> sample_ = {0, 0, 0};
};

}  // namespace candidate
}  // namespace gpd
