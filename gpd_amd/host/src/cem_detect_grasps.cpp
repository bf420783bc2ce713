// cem_detect_grasps CONFIG PCD — the reference's importance-sampling CLI (src/cem_detect_grasps.cpp)
// on the HIP path.  Prints one line per resulting grasp: "GRASP score px py pz width finger_idx".
#include <cstdio>
#include <iostream>

#include "gpd/sequential_importance_sampling.h"
#include "gpd/util/config_file.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    std::cout << "Error: Not enough input arguments!\n\n";
    std::cout << "Usage: cem_detect_grasps CONFIG_FILE PCD_FILE [NORMALS_FILE]\n\n";
    std::cout << "Detect grasp poses for a point cloud, PCD_FILE (*.pcd), using parameters from CONFIG_FILE (*.cfg).\n";
    return -1;
  }
  gpd::util::Cloud cloud(argv[2], {0.0, 0.0, 0.0});
  if (cloud.size() == 0) {
    std::cout << "Error: Input point cloud is empty or does not exist!\n";
    return -1;
  }
  if (argc > 3) {
    cloud.setNormalsFromFile(argv[3]);
    std::cout << "Loaded surface normals from file: " << argv[3] << "\n";
  }
  gpd::SequentialImportanceSampling sis(argv[1]);
  if (!sis.ok()) return -1;
  sis.detector().preprocessPointCloud(cloud);
  std::vector<std::unique_ptr<gpd::candidate::Hand>> grasps = sis.detectGrasps(cloud);
  for (const auto &g : grasps) {
    auto p = g->getPosition();
    printf("GRASP %.9g %.17g %.17g %.17g %.17g %d\n", g->getScore(), p[0], p[1], p[2], g->getGraspWidth(), g->getFingerPlacementIndex());
  }
  return 0;
}
