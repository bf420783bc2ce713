// host_selftest CONFIG PCD — exercises the members of gpd::GraspDetector that the detect_grasps CLI
// does not reach (run by tests/test_host_cli.py on the GPU):
//   1. samples by coordinates (Cloud::setSamples) give the same grasps as the same samples by index;
//   2. generateGraspCandidates -> filterGraspsWorkspace -> pruneGraspCandidates equals detectGrasps'
//      scores; 3. evalGroundTruth on the same cloud reproduces the search's full-antipodal flags;
//   4. the plugin surface (net/classifier.h:52-81): Classifier::create with the reference's default device ->
//      classifyImages on the images of createGraspImages equals the device-resident scores; an image that is
//      not continuous is skipped and scores 0 (eigen_classifier.cpp:68); a detector forced onto the plugin
//      route (cfg classifier_plugin_route = 1) returns the same grasps as the fused one.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
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
  // 4. the Classifier plugin
  {
    std::string weights = config_file.getValueOfKeyAsString("weights_file", "");
    std::string dir = argv[1];
    dir = dir.find_last_of('/') == std::string::npos ? std::string("") : dir.substr(0, dir.find_last_of('/') + 1);
    if (!weights.empty() && weights[0] != '/') weights = dir + weights;
    auto classifier = gpd::net::Classifier::create("", weights);  // default device, as classifier.h:63-66
    if (!classifier) return fail("Classifier::create");
    if (classifier->getBatchSize() != 1) return fail("getBatchSize");
    std::vector<std::unique_ptr<gpd::candidate::Hand>> hands;
    std::vector<std::unique_ptr<gpd::net::Image>> images;
    gpd::util::Cloud cloud3 = cloud;
    if (!detector.createGraspImages(cloud3, hands, images) || images.empty()) return fail("createGraspImages");
    std::vector<float> scores = classifier->classifyImages(images);
    if (scores.size() != images.size() || scores.size() != pruned.size()) return fail("classifyImages count");
    // pruned was reordered by selectGrasps above (partial_sort of the whole list): compare as sorted lists
    std::vector<float> a(scores), b;
    for (auto &h : best) b.push_back((float)h->getScore());
    std::sort(a.begin(), a.end(), std::greater<float>());
    if (a != b) return fail("classifyImages scores differ from the device-resident path");
    if (images.size() > 1) {
      images[1]->continuous = false;
      std::vector<float> s2 = classifier->classifyImages(images);
      if (s2[1] != 0.f) return fail("a non-continuous image must score 0");
      for (size_t i = 0; i < s2.size(); i++)
        if (i != 1 && s2[i] != scores[i]) return fail("skipping one image changed another score");
    }
    // the same cfg with the plugin route forced
    const std::string cfg2 = std::string(argv[1]) + ".plugin";
    {
      FILE *in = fopen(argv[1], "r"), *out = fopen(cfg2.c_str(), "w");
      if (!in || !out) return fail("cfg copy");
      char line[4096];
      while (fgets(line, sizeof(line), in)) fputs(line, out);
      fputs("\nclassifier_plugin_route = 1\n", out);
      fclose(in);
      fclose(out);
    }
    gpd::GraspDetector plugin_detector(cfg2);
    auto via_plugin = plugin_detector.detectGrasps(cloud);
    if (via_plugin.size() != by_index.size()) return fail("plugin route: grasp count");
    for (size_t i = 0; i < via_plugin.size(); i++)
      if (std::memcmp(&via_plugin[i]->record(), &by_index[i]->record(), sizeof(gpd_hand)) != 0) return fail("plugin route: grasps differ");
  }
  printf("SELFTEST OK: %zu grasps, %zu candidates re-evaluated, %d full antipodal\n", by_index.size(), labels.size(), n_full);
  return 0;
}
