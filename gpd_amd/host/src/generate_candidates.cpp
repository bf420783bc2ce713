// generate_candidates CONFIG PCD [NORMALS] — the reference's candidate-generation CLI
// (src/generate_candidates.cpp: CandidatesGenerator::preprocessPointCloud + generateGraspCandidates,
// then a plot) on the HIP path.  Instead of the PCL viewer it prints one line per valid hand:
// "CANDIDATE set slot finger_idx px py pz width half full".
#include <cstdio>
#include <iostream>

#include "gpd/grasp_detector.h"
#include "gpd/util/config_file.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    std::cout << "Error: Not enough input arguments!\n\n";
    std::cout << "Usage: generate_candidates CONFIG_FILE PCD_FILE [NORMALS_FILE]\n\n";
    std::cout << "Generate grasp candidates for a point cloud, PCD_FILE (*.pcd), using parameters from CONFIG_FILE (*.cfg).\n\n";
    std::cout << "[NORMALS_FILE] (optional) contains a surface normal for each point in the cloud (*.csv).\n";
    return -1;
  }
  gpd::util::ConfigFile config_file(argv[1]);
  if (!config_file.ExtractKeys()) return -1;
  std::vector<double> camera_position = config_file.getValueOfKeyAsStdVectorDouble("camera_position", "0.0 0.0 0.0");
  camera_position.resize(3, 0.0);
  gpd::util::Cloud cloud(argv[2], camera_position);
  if (cloud.size() == 0) {
    std::cout << "Input point cloud is empty or does not exist!\n";
    return -1;
  }
  if (argc > 3) {
    cloud.setNormalsFromFile(argv[3]);
    std::cout << "Loaded surface normals from file.\n";
  }
  gpd::GraspDetector detector(argv[1]);
  if (!detector.ok()) return -1;
  detector.preprocessPointCloud(cloud);
  std::vector<std::unique_ptr<gpd::candidate::HandSet>> sets = detector.generateGraspCandidates(cloud);
  int num_hands = 0;
  for (size_t s = 0; s < sets.size(); s++) {
    const auto &hands = sets[s]->getHands();
    for (size_t j = 0; j < hands.size(); j++) {
      if (!sets[s]->getIsValid()[j]) continue;
      num_hands++;
      const auto p = hands[j]->getPosition();
      printf("CANDIDATE %zu %zu %d %.17g %.17g %.17g %.17g %d %d\n", s, j, hands[j]->getFingerPlacementIndex(), p[0], p[1], p[2],
             hands[j]->getGraspWidth(), (int)hands[j]->isHalfAntipodal(), (int)hands[j]->isFullAntipodal());
    }
  }
  std::cout << "Generated " << num_hands << " grasp candidates.\n";
  return 0;
}
