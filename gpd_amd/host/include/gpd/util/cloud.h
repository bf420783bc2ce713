// util::Cloud — the part of the reference's PCL wrapper that the hot path reads
// (util/cloud.h: getCloudProcessed, getNormals, getCameraSource, getViewPoints,
// getSampleIndices/setSampleIndices, getSamples/setSamples, subsample, voxelizeCloud).  Normal
// estimation runs on the device (gpd_hip_estimate_normals) from GraspDetector::preprocessPointCloud.
#pragma once
#include <string>
#include <vector>

namespace gpd {
namespace util {

class Cloud {
 public:
  Cloud() {}
  // xyz / normals: 3 floats per point; camera_source: n_cams x P (0/1); view_points: 3 doubles per camera
  Cloud(const std::vector<float> &xyz, const std::vector<float> &normals, const std::vector<int> &camera_source,
        const std::vector<double> &view_points);
  // ASCII PCD with FIELDS x y z [normal_x normal_y normal_z ...] (pcl::io::loadPCDFile, cloud.cpp:643-660)
  Cloud(const std::string &filename, const std::vector<double> &view_points);

  size_t size() const { return xyz_.size() / 3; }
  bool hasNormals() const { return normals_.size() == xyz_.size() && !xyz_.empty(); }
  const std::vector<float> &getCloudProcessed() const { return xyz_; }
  const std::vector<float> &getNormals() const { return normals_; }
  const std::vector<int> &getCameraSource() const { return camera_source_; }
  const std::vector<double> &getViewPoints() const { return view_points_; }
  int numCameras() const { return (int)(view_points_.size() / 3); }
  const std::vector<int> &getSampleIndices() const { return sample_indices_; }
  void setSampleIndices(const std::vector<int> &idx) { sample_indices_ = idx; }
  // Cloud::getSamples / setSamples (cloud.h:248-262): samples by coordinates, 3 doubles each; when
  // present they take precedence over the indices (hand_search.cpp:37-44)
  const std::vector<double> &getSamples() const { return samples_; }
  void setSamples(const std::vector<double> &samples) { samples_ = samples; }
  void setNormals(const std::vector<float> &normals) { normals_ = normals; }
  // Cloud::setNormalsFromFile (cloud.cpp:622-641): a CSV of three rows (x, y, z components) with one
  // column per point.  The device boundary carries float32 normals (the reference keeps these doubles).
  void setNormalsFromFile(const std::string &filename);
  // Cloud::filterWorkspace (cloud.cpp:206-267): keeps the points (normals, camera columns) strictly inside
  // the box [x0 x1 y0 y1 z0 z1]; samples by coordinate likewise.  As in the reference the sample
  // INDICES are replaced by the positions of the surviving entries (:215) and not remapped to the
  // filtered cloud — call it before sampling, as preprocessPointCloud does.
  void filterWorkspace(const std::vector<double> &workspace);
  // its first two blocks alone (cloud.cpp:208-241): sample indices and samples by coordinate; the detector runs the
  // third one — the points — on the device (gpd_hip_preprocess_cloud) and hands the result back through setProcessed
  void filterWorkspaceSamples(const std::vector<double> &workspace);
  // replaces points and camera source (rows of xyz.size() / 3); normals may be empty; sample indices are kept
  void setProcessed(const std::vector<float> &xyz, const std::vector<int> &camera_source, const std::vector<float> &normals);
  // Cloud::voxelizeCloud (cloud.cpp:286-348), including what its std::set comparator (cloud.h:105-122,
  // not an ordering) keeps under libstdc++; drops normals like the reference's preprocessing order does.
  void voxelizeCloud(float cell_size);
  // Cloud::subsample (cloud.cpp:350-405) draws with pcl::RandomSample (time-seeded); here a
  // seeded Fisher-Yates permutation so runs are reproducible.
  void subsample(int num_samples, unsigned seed = 0);

 private:
  std::vector<float> xyz_, normals_;
  std::vector<int> camera_source_;
  std::vector<double> view_points_;
  std::vector<int> sample_indices_;
  std::vector<double> samples_;
};

}  // namespace util
}  // namespace gpd
