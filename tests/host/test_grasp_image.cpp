// test_grasp_image INPUT_FILE SAMPLE_INDEX DRAW_GRASP_IMAGES [IMAGE_CHANNELS] [HAND_AXES] — the one test
// program the reference builds (src/tests/test_grasp_image.cpp:19-171; README usage:
// `test_grasp_image ../tutorials/krylon.pcd 3456 1 ../models/lenet/15channels/params/`) on the HIP path:
// ONE sample index, normals computed with radius 0.03 and then negated (:116-118), one orientation,
// candidates for that sample, one grasp image per valid hand, the antipodal flags.  The parameters
// are the constants of the reference program (:55-107).  Instead of the PCL / OpenCV windows it
// prints, per image, "IMAGE i set slot finger_idx half full fnv1a64" and, with
// GPD_DUMP_IMAGES=FILE in the environment, writes the raw HWC u8 images to FILE.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "gpd/grasp_detector.h"

namespace gpd {
namespace apps {
namespace test_grasp_image {

int DoMain(int argc, char *argv[]) {
  if (argc < 4) {
    std::cout << "ERROR: Not enough arguments given!\n";
    std::cout << "Usage: rosrun gpd test_grasp_image INPUT_FILE SAMPLE_INDEX DRAW_GRASP_IMAGES [IMAGE_CHANNELS] [HAND_AXES]\n";
    return -1;
  }
  // View point from which the camera sees the point cloud.
  util::Cloud cloud(argv[1], {0.0, 0.0, 0.0});
  if (cloud.size() == 0) {
    std::cout << "Error: Input point cloud is empty or does not exist!\n";
    return -1;
  }
  const int sample_idx = std::stoi(argv[2]);
  if (sample_idx < 0 || (size_t)sample_idx >= cloud.size()) {
    std::cout << "Error: Sample index is larger than the number of points in the cloud!\n";
    return -1;
  }
  cloud.setSampleIndices({sample_idx});

  candidate::HandSearch::Parameters hand_search_params;
  candidate::HandGeometry hand_geom;
  hand_geom.finger_width_ = 0.01;
  hand_geom.outer_diameter_ = 0.12;
  hand_geom.depth_ = 0.06;
  hand_geom.height_ = 0.02;
  hand_geom.init_bite_ = 0.01;
  hand_search_params.hand_geometry_ = hand_geom;

  descriptor::ImageGeometry image_geom;
  image_geom.outer_diameter_ = 0.10;
  image_geom.depth_ = 0.06;
  image_geom.height_ = 0.02;
  image_geom.size_ = 60;
  image_geom.num_channels_ = 15;
  if (argc >= 5) image_geom.num_channels_ = std::stoi(argv[4]);

  hand_search_params.num_samples_ = 1;
  hand_search_params.num_threads_ = 1;
  hand_search_params.nn_radius_frames_ = 0.01;
  hand_search_params.num_orientations_ = 1;
  hand_search_params.num_finger_placements_ = 10;
  hand_search_params.deepen_hand_ = true;
  hand_search_params.friction_coeff_ = 20.0;
  hand_search_params.min_viable_ = 6;
  std::vector<int> hand_axes;
  if (argc >= 6) {
    for (int i = 5; i < argc; i++) hand_axes.push_back(std::stoi(argv[i]));
  } else {
    hand_axes.push_back(2);
  }
  hand_search_params.hand_axes_ = hand_axes;
  std::cout << "hand_axes: ";
  for (size_t i = 0; i < hand_axes.size(); i++) std::cout << hand_axes[i] << " ";
  std::cout << "\n";

  GraspDetector detector(hand_search_params, image_geom);
  if (!detector.ok()) return -1;

  // Calculate surface normals, then flip them (:113-118).
  const double normals_radius = 0.03;
  if (!detector.calculateNormals(cloud, normals_radius)) return -1;
  std::vector<float> normals = cloud.getNormals();
  for (float &v : normals) v = -v;
  cloud.setNormals(normals);

  // Generate grasp candidates.
  std::vector<std::unique_ptr<candidate::HandSet>> hand_set_list = detector.generateGraspCandidates(cloud);
  if (hand_set_list.size() == 0) return -1;
  const candidate::Hand &hand = *hand_set_list[0]->getHands()[0];
  const auto s = hand.getSample(), p = hand.getPosition();
  const auto a = hand.getApproach(), b = hand.getBinormal(), x = hand.getAxis();
  printf("sample: %.17g %.17g %.17g\n", s[0], s[1], s[2]);
  printf("grasp orientation:\n");
  for (int r = 0; r < 3; r++) printf("%.17g %.17g %.17g\n", a[r], b[r], x[r]);
  printf("grasp position: %.17g %.17g %.17g\n", p[0], p[1], p[2]);

  // Create the images for these grasp candidates.
  const bool plot_images = std::stoi(argv[3]) == 1;
  printf("Creating grasp image ...\n");
  printf("plot images: %d\n", plot_images);
  std::vector<std::unique_ptr<net::Image>> images;
  std::vector<std::unique_ptr<candidate::Hand>> hands;
  if (!detector.createImages(cloud, hand_set_list, images, hands)) return -1;
  FILE *dump = getenv("GPD_DUMP_IMAGES") ? fopen(getenv("GPD_DUMP_IMAGES"), "wb") : nullptr;
  for (size_t i = 0; i < images.size(); i++) {
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char v : images[i]->data) h = (h ^ v) * 1099511628211ull;
    const gpd_hand &r = hands[i]->record();
    printf("IMAGE %zu %d %d %d %d %d %016llx\n", i, r.set_index, r.slot, hands[i]->getFingerPlacementIndex(),
           (int)hands[i]->isHalfAntipodal(), (int)hands[i]->isFullAntipodal(), h);
    if (dump) fwrite(images[i]->data.data(), 1, images[i]->data.size(), dump);
  }
  if (dump) fclose(dump);

  // Evaluate if the grasp candidates are antipodal.
  std::cout << "Antipodal: ";
  for (size_t i = 0; i < hands.size(); i++) std::cout << hands[i]->isFullAntipodal() << " ";
  std::cout << "\n";
  return 0;
}

}  // namespace test_grasp_image
}  // namespace apps
}  // namespace gpd

int main(int argc, char *argv[]) { return gpd::apps::test_grasp_image::DoMain(argc, argv); }
