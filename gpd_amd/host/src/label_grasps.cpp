// label_grasps CONFIG PCD MESH — the reference's ground-truth labelling CLI (src/label_grasps.cpp:20-128)
// on the HIP path: candidates + images on the (partial) cloud, then every candidate is checked again
// against the complete "mesh" cloud (GraspDetector::evalGroundTruth -> reevaluateHypotheses).
// As the reference does (:84, :92), the normals of both clouds are negated after they are computed.
// Prints "(i) label: L" per candidate like the reference, and "LABELS n_full / n".
#include <cstdio>
#include <iostream>

#include "gpd/grasp_detector.h"
#include "gpd/util/config_file.h"

static void negate_normals(gpd::util::Cloud &cloud) {
  std::vector<float> n = cloud.getNormals();
  for (float &v : n) v = -v;
  cloud.setNormals(n);
}

int main(int argc, char *argv[]) {
  if (argc < 4) {
    std::cout << "Error: Not enough input arguments!\n\n";
    std::cout << "Usage: label_grasps CONFIG_FILE PCD_FILE MESH_FILE\n\n";
    std::cout << "Find grasp poses for a point cloud, PCD_FILE (*.pcd), using parameters from CONFIG_FILE (*.cfg), and check them "
                 "against a mesh, MESH_FILE (*.pcd).\n\n";
    return -1;
  }
  gpd::util::ConfigFile config_file(argv[1]);
  if (!config_file.ExtractKeys()) return -1;
  const double normals_radius = config_file.getValueOfKey<double>("normals_radius", 0.03);
  gpd::util::Cloud cloud(argv[2], {0.0, 0.0, 0.0});
  if (cloud.size() == 0) {
    std::cout << "Error: Input point cloud is empty or does not exist!\n";
    return -1;
  }
  gpd::util::Cloud mesh(argv[3], {0.0, 0.0, 0.0});
  if (mesh.size() == 0) {
    std::cout << "Error: Mesh point cloud is empty or does not exist!\n";
    return -1;
  }
  gpd::GraspDetector detector(argv[1]);
  if (!detector.ok()) return -1;
  // prepare the cloud (:81-89) and the mesh (:91-93)
  detector.preprocessPointCloud(cloud);
  negate_normals(cloud);
  if (!mesh.hasNormals() && !detector.calculateNormals(mesh, normals_radius)) return -1;
  negate_normals(mesh);
  std::vector<std::unique_ptr<gpd::candidate::Hand>> hands;
  std::vector<std::unique_ptr<gpd::net::Image>> images;
  if (!detector.createGraspImages(cloud, hands, images)) return -1;
  std::vector<int> labels = detector.evalGroundTruth(mesh, hands);
  printf("labels: %zu\n", labels.size());
  int n_full = 0;
  for (size_t i = 0; i < hands.size(); i++) {
    printf("(%zu) label: %d\n", i, labels[i]);
    n_full += hands[i]->isFullAntipodal() ? 1 : 0;
  }
  printf("LABELS %d / %zu\n", n_full, hands.size());
  return 0;
}
