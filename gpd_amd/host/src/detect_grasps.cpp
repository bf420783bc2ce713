// detect_grasps CONFIG PCD — the reference's CLI (src/detect_grasps.cpp:20-86) on the HIP path.
// The PCD must carry normals (fields normal_x normal_y normal_z): preprocessing is out of scope.
// Prints one line per selected grasp: "GRASP score px py pz width finger_idx".
#include <cstdio>
#include <iostream>

#include "gpd/grasp_detector.h"
#include "gpd/util/config_file.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    std::cout << "Error: Not enough input arguments!\n\n";
    std::cout << "Usage: detect_grasps CONFIG_FILE PCD_FILE [NORMALS_FILE]\n\n";
    std::cout << "Detect grasp poses for a point cloud, PCD_FILE (*.pcd with normals), using parameters from CONFIG_FILE (*.cfg).\n";
    return -1;
  }
  gpd::util::ConfigFile config_file(argv[1]);
  if (!config_file.ExtractKeys()) return -1;
  std::vector<double> camera_position = config_file.getValueOfKeyAsStdVectorDouble("camera_position", "0.0 0.0 0.0");
  camera_position.resize(3, 0.0);
  gpd::util::Cloud cloud(argv[2], camera_position);
  if (cloud.size() == 0) {
    std::cout << "Error: Input point cloud is empty or does not exist!\n";
    return -1;
  }
  if (argc > 3) {  // [NORMALS_FILE]: a surface normal for each point of the cloud (*.csv)
    cloud.setNormalsFromFile(argv[3]);
    std::cout << "Loaded surface normals from file: " << argv[3] << "\n";
  }
  gpd::GraspDetector detector(argv[1]);
  if (!detector.ok()) return -1;
  detector.preprocessPointCloud(cloud);
  std::vector<std::unique_ptr<gpd::candidate::Hand>> grasps = detector.detectGrasps(cloud);
  for (const auto &g : grasps) {
    auto p = g->getPosition();
    printf("GRASP %.9g %.17g %.17g %.17g %.17g %d\n", g->getScore(), p[0], p[1], p[2], g->getGraspWidth(), g->getFingerPlacementIndex());
  }
  return 0;
}
