// host_selftest CONFIG PCD — exercises the members of gpd::GraspDetector that the detect_grasps CLI
// does not reach (run by tests/test_host_cli.py on the GPU):
//   1. samples by coordinates (Cloud::setSamples) give the same grasps as the same samples by index;
//   2. generateGraspCandidates -> filterGraspsWorkspace -> pruneGraspCandidates equals detectGrasps'
//      scores; 3. evalGroundTruth on the same cloud reproduces the search's full-antipodal flags.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "gpd/grasp_detector.h"
#include "gpd/util/config_file.h"

static int fail(const char *what) {
  printf("SELFTEST FAILED: %s\n", what);
  return 1;
}

int main(int argc, char *argv[]) {
  if (argc < 3) return fail("usage: host_selftest CONFIG PCD");
  gpd::util::ConfigFile config_file(argv[1]);
  if (!config_file.ExtractKeys()) return fail("config");
  gpd::util::Cloud cloud(argv[2], {0.0, 0.0, 0.0});
  gpd::GraspDetector detector(argv[1]);
  if (!detector.ok() || cloud.size() == 0) return fail("setup");
  detector.preprocessPointCloud(cloud);
  const std::vector<int> idx = cloud.getSampleIndices();
  auto by_index = detector.detectGrasps(cloud);
  // 1. the same samples as doubles
  gpd::util::Cloud cloud2 = cloud;
  std::vector<double> samples;
  for (int i : idx)
    for (int r = 0; r < 3; r++) samples.push_back((double)cloud.getCloudProcessed()[3 * (size_t)i + r]);
  cloud2.setSamples(samples);
  cloud2.setSampleIndices({});
  auto by_xyz = detector.detectGrasps(cloud2);
  if (by_index.empty() || by_index.size() != by_xyz.size()) return fail("sample count");
  for (size_t i = 0; i < by_index.size(); i++)
    if (std::memcmp(&by_index[i]->record(), &by_xyz[i]->record(), sizeof(gpd_hand)) != 0) return fail("samples by coordinates differ");
  // 2. the unfused route
  auto sets = detector.generateGraspCandidates(cloud);
  const std::vector<double> ws = {-1, 1, -1, 1, -1, 1};
  auto filtered = detector.filterGraspsWorkspace(sets, ws);
  auto pruned = detector.pruneGraspCandidates(cloud, filtered, -1e30);
  // 3. ground truth = the same cloud
  std::vector<bool> full;
  for (auto &h : pruned) full.push_back(h->isFullAntipodal());
  std::vector<int> labels = detector.evalGroundTruth(cloud, pruned);
  int n_full = 0;
  for (size_t i = 0; i < pruned.size(); i++) {
    if ((labels[i] != 0) != full[i] || pruned[i]->isFullAntipodal() != full[i]) return fail("evalGroundTruth labels differ");
    n_full += labels[i];
  }
  auto best = detector.selectGrasps(pruned);
  if (best.size() != by_index.size()) return fail("prune count");
  for (size_t i = 0; i < best.size(); i++)
    if (best[i]->getScore() != by_index[i]->getScore()) return fail("prune scores differ");
  printf("SELFTEST OK: %zu grasps, %zu candidates re-evaluated, %d full antipodal\n", by_index.size(), labels.size(), n_full);
  return 0;
}
