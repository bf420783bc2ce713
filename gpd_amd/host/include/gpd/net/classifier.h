// net::Classifier — the reference's plugin surface (net/classifier.h:52-81) with a HIP back-end.
// cv::Mat is absent here; `Image` carries what classifyImages reads from it (rows, cols,
// channels, data pointer, continuity; eigen_classifier.cpp:64-74, 130-149).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct gpd_hip_ctx;

namespace gpd {
namespace net {

struct Image {  // stand-in for cv::Mat CV_8UC(C), HWC interleaved
  int rows = 0, cols = 0, channels_ = 0;
  std::vector<uint8_t> data;
  bool continuous = true;
  Image() {}
  Image(int r, int c, int ch) : rows(r), cols(c), channels_(ch), data((size_t)r * c * ch, 0) {}
  bool isContinuous() const { return continuous; }
  int channels() const { return channels_; }
};

class Classifier {
 public:
  enum class Device : uint8_t { eCPU = 0, eGPU = 1, eVPU = 2, eFPGA = 3 };
  virtual ~Classifier() {}
  // The reference's factory (classifier.cpp:46-62) picks its back-end at COMPILE time (USE_OPENVINO / USE_CAFFE /
  // Eigen) and only hands `device` on to it; the Eigen back-end ignores it altogether
  // (eigen_classifier.cpp:6-9).  This build is compiled with exactly one back-end, HipClassifier, so — like the
  // reference's Eigen build — every `device` value yields that back-end: a reference cfg without a `device` key
  // (-> 0 = eCPU, grasp_detector.cpp:133) gets the HIP classifier too.  There is no CPU path behind the
  // interface; a value other than eGPU is reported once on stdout, nothing else changes.
  // weights_file: the parameter directory ending in '/', as for EigenClassifier (eigen_classifier.cpp:28-50);
  // model_file and batch_size are accepted and unused, as there.  nullptr when the parameters cannot be loaded.
  static std::shared_ptr<Classifier> create(const std::string &model_file, const std::string &weights_file,
                                            Device device = Device::eCPU, int batch_size = 1);
  virtual std::vector<float> classifyImages(const std::vector<std::unique_ptr<Image>> &image_list) = 0;
  virtual int getBatchSize() const = 0;
};

class HipClassifier : public Classifier {
 public:
  HipClassifier(const std::string &model_file, const std::string &weights_file, Classifier::Device device, int batch_size);
  ~HipClassifier() override;
  std::vector<float> classifyImages(const std::vector<std::unique_ptr<Image>> &image_list) override;
  int getBatchSize() const override { return batch_size_; }
  bool ok() const { return ctx_ != nullptr && loaded_; }
  // Arithmetic of the scoring kernels (gpd_hip_set_lenet_mode): 0 = operands split exactly over the int8 / bf16 matrix
  // pipes (default: within 1e-4 of EigenClassifier's plain-float result, closer to float64 than it), 1 = the f32 chain
  // in the reference's own operation order.  GraspDetector sets it from the cfg key `hip_lenet_mode`.
  bool setScoringMode(int mode);
  // raw float32 parameter file -> vector (EigenClassifier::readBinaryFileIntoVector, :185-204)
  static std::vector<float> readBinaryFileIntoVector(const std::string &location);
  // The parameters as read from the files (conv1 w, b, conv2 w, b, ip1 w, b, ip2 w, b): GraspDetector loads the
  // same arrays into the context of its fused search -> image -> score path.
  int channels() const { return channels_; }
  const std::vector<float> &parameter(int i) const { return params_[i]; }

 private:
  gpd_hip_ctx *ctx_ = nullptr;
  bool loaded_ = false;
  int channels_ = 0;
  int batch_size_ = 1;
  std::vector<float> params_[8];
};

}  // namespace net
}  // namespace gpd
