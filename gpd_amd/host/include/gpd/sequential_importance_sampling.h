// gpd::SequentialImportanceSampling — the reference's cross-entropy / importance-sampling driver
// (include/gpd/sequential_importance_sampling.h:59-141, src/gpd/sequential_importance_sampling.cpp:11-272;
// SURVEY §8f rank 4) over the HIP path: initial candidates from subsampled cloud points, then
// rounds of candidates at samples drawn around the known hand sets (sum or max of Gaussians) plus
// uniform cloud samples, one classification of everything at the end, clustering.
// The reference draws with rand() and a std::random_device-seeded mt19937; here one seeded xorshift
// generator (Box-Muller normals) so that runs are reproducible.  GPD_SIS_DUMP=<file> writes the
// samples of every round (tests replay them through the oracle).
#pragma once
#include <array>
#include <memory>
#include <string>
#include <vector>

#include "gpd/grasp_detector.h"

namespace gpd {

class SequentialImportanceSampling {
 public:
  explicit SequentialImportanceSampling(const std::string &config_filename);
  std::vector<std::unique_ptr<candidate::Hand>> detectGrasps(util::Cloud &cloud);
  bool ok() const { return grasp_detector_ && grasp_detector_->ok(); }
  GraspDetector &detector() { return *grasp_detector_; }

 private:
  void drawSamplesFromSumOfGaussians(const std::vector<std::unique_ptr<candidate::HandSet>> &hand_sets, double sigma, int num_gauss_samples,
                                     std::vector<double> &samples_out);
  void drawSamplesFromMaxOfGaussians(const std::vector<std::unique_ptr<candidate::HandSet>> &hand_sets, double sigma, int num_gauss_samples,
                                     std::vector<double> &samples_out, double term);
  void drawUniformSamples(const util::Cloud &cloud, int num_samples, int start_idx, std::vector<double> &samples);
  unsigned long long nextRandom();  // xorshift64
  int randInt(int n) { return (int)(nextRandom() % (unsigned long long)n); }
  double randNormal(double sigma);

  std::unique_ptr<GraspDetector> grasp_detector_;
  std::unique_ptr<Clustering> clustering_;
  int num_iterations_, num_samples_, num_init_samples_;
  double prob_rand_samples_, radius_;
  int sampling_method_;
  double min_score_;
  std::vector<double> workspace_, workspace_grasps_;
  bool filter_approach_direction_;
  std::array<double, 3> direction_;
  double thresh_rad_;
  unsigned long long rng_state_;
};

}  // namespace gpd
