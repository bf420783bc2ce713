// Host side of the drop-in: util::ConfigFile, util::Cloud, net::HipClassifier,
// gpd::GraspDetector — the reference's classes re-hosted on libgpd_hip.so (include/gpd_hip.h).
// Error convention of the reference: print + empty result (grasp_detector.cpp:201-205, 227-229).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <array>
#include <cfloat>
#include <cmath>
#include <numeric>
#include <set>

#include "gpd/clustering.h"
#include "gpd/grasp_detector.h"
#include "gpd/sequential_importance_sampling.h"
#include "gpd/util/config_file.h"

namespace gpd {

// ------------------------------------------------------------------ util::ConfigFile
namespace util {

static std::string trim(const std::string &s) {
  const char *ws = " \t\r\n";
  size_t a = s.find_first_not_of(ws);
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(ws);
  return s.substr(a, b - a + 1);
}

bool ConfigFile::ExtractKeys() {  // config_file.cpp:76-104
  std::ifstream file(fName_.c_str());
  if (!file) {
    std::cout << "Config file " + fName_ + " could not be found!\n";
    return false;
  }
  std::string line;
  size_t lineNo = 0;
  while (std::getline(file, line)) {
    lineNo++;
    size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    if (trim(line).empty()) continue;
    size_t eq = line.find('=');
    if (eq == std::string::npos) {
      std::cout << "CFG: Bad format for line: " << lineNo << "\n";
      continue;
    }
    std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
    if (key.empty()) {
      std::cout << "CFG: Bad format for line: " << lineNo << "\n";
      continue;
    }
    if (!keyExists(key)) contents_[key] = val;  // first definition wins, as the reference's map insert
  }
  return true;
}

std::string ConfigFile::getValueOfKeyAsString(const std::string &key, const std::string &defaultValue) const {
  auto it = contents_.find(key);
  return it == contents_.end() ? defaultValue : it->second;
}
std::vector<double> ConfigFile::getValueOfKeyAsStdVectorDouble(const std::string &key, const std::string &defaultValue) const {
  std::stringstream ss(getValueOfKeyAsString(key, defaultValue));
  std::vector<double> v;
  double x;
  while (ss >> x) v.push_back(x);
  return v;
}
std::vector<int> ConfigFile::getValueOfKeyAsStdVectorInt(const std::string &key, const std::string &defaultValue) const {
  std::stringstream ss(getValueOfKeyAsString(key, defaultValue));
  std::vector<int> v;
  double x;  // the reference parses ints through a double as well (config_file.cpp:154-167)
  while (ss >> x) v.push_back((int)x);
  return v;
}

// ------------------------------------------------------------------ util::Cloud
Cloud::Cloud(const std::vector<float> &xyz, const std::vector<float> &normals, const std::vector<int> &camera_source,
             const std::vector<double> &view_points)
    : xyz_(xyz), normals_(normals), camera_source_(camera_source), view_points_(view_points) {
  if (camera_source_.empty()) camera_source_.assign(size() * std::max(1, numCameras()), 1);
}

namespace {
// LZF (liblzf's stream format, what pcl::lzfDecompress reads): a control byte below 32 starts a run of ctrl + 1 literal
// bytes; otherwise it is a back reference of length (ctrl >> 5) + 2 (length field 7: one more byte is added to it) at
// distance ((ctrl & 31) << 8 | next byte) + 1.  Returns false unless exactly out_len bytes come out of a well-formed stream.
bool lzfDecompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return false;
      std::memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= in_len) return false;
        len += in[ip++];
      }
      if (ip >= in_len) return false;
      const size_t dist = (((size_t)ctrl & 31) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return false;
      for (size_t k = 0; k < len; k++, op++) out[op] = out[op - dist];  // overlapping copies repeat the pattern
    }
  }
  return op == out_len;
}
}  // namespace

Cloud::Cloud(const std::string &filename, const std::vector<double> &view_points) : view_points_(view_points) {
  // pcl::io::loadPCDFile (cloud.cpp:643-660): the ASCII, the binary and the binary_compressed (LZF, fields-major) layout
  std::ifstream f(filename.c_str(), std::ios::binary);
  if (!f) {
    printf("Couldn't read .pcd file: %s\n", filename.c_str());
    return;
  }
  std::string line;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  std::string kind;
  size_t points = 0;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::stringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "FIELDS") {
      std::string x;
      while (ss >> x) fields.push_back(x);
    } else if (tag == "SIZE") {
      int x;
      while (ss >> x) sizes.push_back(x);
    } else if (tag == "TYPE") {
      std::string x;
      while (ss >> x) types.push_back(x);
    } else if (tag == "COUNT") {
      int x;
      while (ss >> x) counts.push_back(x);
    } else if (tag == "POINTS") {
      ss >> points;
    } else if (tag == "DATA") {
      ss >> kind;
      break;
    }
  }
  if (kind != "ascii" && kind != "binary" && kind != "binary_compressed") {
    printf("Unknown .pcd data layout (DATA %s): %s\n", kind.c_str(), filename.c_str());
    return;
  }
  // an untrusted header: field counts must agree, sizes are 1 / 2 / 4 / 8 bytes, counts at least 1 and small
  counts.resize(fields.size(), 1);
  bool header_ok = !fields.empty() && fields.size() <= 64 && (sizes.empty() || sizes.size() == fields.size()) &&
                   (types.empty() || types.size() == fields.size()) && points <= ((size_t)1 << 31);
  for (int v : sizes) header_ok = header_ok && (v == 1 || v == 2 || v == 4 || v == 8);
  for (int v : counts) header_ok = header_ok && v >= 1 && v <= 1024;
  if (!header_ok) {
    printf("PCD header is malformed (FIELDS / SIZE / TYPE / COUNT / POINTS): %s\n", filename.c_str());
    return;
  }
  int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
  for (int i = 0; i < (int)fields.size(); i++) {
    if (fields[i] == "x") ix = i;
    if (fields[i] == "y") iy = i;
    if (fields[i] == "z") iz = i;
    if (fields[i] == "normal_x") inx = i;
    if (fields[i] == "normal_y") iny = i;
    if (fields[i] == "normal_z") inz = i;
  }
  if (ix < 0 || iy < 0 || iz < 0) {
    printf("PCD file has no x y z fields: %s\n", filename.c_str());
    return;
  }
  const bool with_normals = inx >= 0 && iny >= 0 && inz >= 0;
  auto keep = [&](const double *row) {
    if (row[ix] != row[ix] || row[iy] != row[iy] || row[iz] != row[iz]) return;  // removeNans
    xyz_.push_back((float)row[ix]);
    xyz_.push_back((float)row[iy]);
    xyz_.push_back((float)row[iz]);
    if (with_normals) {
      normals_.push_back((float)row[inx]);
      normals_.push_back((float)row[iny]);
      normals_.push_back((float)row[inz]);
    }
  };
  std::vector<double> row(fields.size());
  if (kind == "ascii") {
    while (std::getline(f, line)) {
      std::stringstream ss(line);
      bool good = true;
      for (size_t i = 0; i < fields.size() && good; i++)
        for (int c = 0; c < counts[i] && good; c++) {  // a field of COUNT n takes n columns; its first value is kept
          double v;
          if (!(ss >> v)) good = false;
          if (c == 0) row[i] = v;
        }
      if (good) keep(row.data());
    }
  } else {
    // one record per point: the fields back to back, SIZE bytes each (x COUNT), little endian
    if (sizes.size() != fields.size() || types.size() != fields.size()) {
      printf("PCD header is incomplete (SIZE / TYPE): %s\n", filename.c_str());
      return;
    }
    std::vector<size_t> offset(fields.size());
    size_t stride = 0;
    for (size_t i = 0; i < fields.size(); i++) {
      offset[i] = stride;
      stride += (size_t)sizes[i] * (size_t)counts[i];
    }
    auto value = [&](size_t i, const char *p) {
      double v = 0.0;
      if (types[i] == "F" && sizes[i] == 4) {
        float x;
        std::memcpy(&x, p, 4);
        v = x;
      } else if (types[i] == "F" && sizes[i] == 8) {
        std::memcpy(&v, p, 8);
      } else if (sizes[i] == 4) {
        int32_t x;
        std::memcpy(&x, p, 4);
        v = types[i] == "U" ? (double)(uint32_t)x : (double)x;
      } else if (sizes[i] == 2) {
        int16_t x;
        std::memcpy(&x, p, 2);
        v = types[i] == "U" ? (double)(uint16_t)x : (double)x;
      } else if (sizes[i] == 1) {
        v = types[i] == "U" ? (double)(uint8_t)p[0] : (double)(int8_t)p[0];
      }
      return v;
    };
    if (kind == "binary") {
      std::vector<char> rec(stride);
      for (size_t n = 0; n < points && f.read(rec.data(), (std::streamsize)stride); n++) {
        for (size_t i = 0; i < fields.size(); i++) row[i] = value(i, rec.data() + offset[i]);
        keep(row.data());
      }
    } else {
      // binary_compressed (PCDWriter::writeBinaryCompressed): two little-endian uint32 (compressed, uncompressed byte
      // count), then one LZF stream of the cloud stored fields-major (all x, all y, ...; a field of COUNT n keeps its n
      // values of a point together)
      uint32_t head[2];
      if (!f.read(reinterpret_cast<char *>(head), 8)) {
        printf("PCD file ends inside the compression header: %s\n", filename.c_str());
        return;
      }
      const size_t comp = head[0], raw = head[1];
      if (raw != stride * points) {
        printf("PCD compressed block does not match the header (%zu bytes for %zu points of %zu): %s\n", raw, points, stride,
               filename.c_str());
        return;
      }
      // the 8-byte block header is not trusted: the compressed size cannot exceed what is left of the file (a crafted
      // header would otherwise ask for up to 4 GiB before a single byte is read)
      const std::streampos here = f.tellg();
      f.seekg(0, std::ios::end);
      const std::streampos end = f.tellg();
      f.seekg(here);
      if (here < 0 || end < here || comp > (size_t)(end - here)) {
        printf("PCD file ends inside the compressed block: %s\n", filename.c_str());
        return;
      }
      std::vector<unsigned char> in;
      std::vector<char> out;
      try {
        in.resize(comp);
        out.resize(raw);
      } catch (const std::bad_alloc &) {
        printf("PCD compressed block is corrupt (%zu / %zu bytes cannot be held): %s\n", comp, raw, filename.c_str());
        return;
      }
      if (comp && !f.read(reinterpret_cast<char *>(in.data()), (std::streamsize)comp)) {
        printf("PCD file ends inside the compressed block: %s\n", filename.c_str());
        return;
      }
      if (!lzfDecompress(in.data(), comp, reinterpret_cast<unsigned char *>(out.data()), raw)) {
        printf("PCD compressed block is corrupt: %s\n", filename.c_str());
        return;
      }
      std::vector<size_t> base(fields.size());
      for (size_t i = 0; i < fields.size(); i++) base[i] = offset[i] * points;
      for (size_t n = 0; n < points; n++) {
        for (size_t i = 0; i < fields.size(); i++) row[i] = value(i, out.data() + base[i] + n * (size_t)sizes[i] * (size_t)counts[i]);
        keep(row.data());
      }
    }
  }
  if (view_points_.empty()) view_points_.assign(3, 0.0);
  camera_source_.assign(size() * numCameras(), 1);
}

namespace {
struct VoxelDiffers {  // UniqueVector4First3Comparator (cloud.h:105-122)
  bool operator()(const std::array<int, 4> &a, const std::array<int, 4> &b) const {
    return a[0] != b[0] || a[1] != b[1] || a[2] != b[2];
  }
};
}  // namespace

void Cloud::setProcessed(const std::vector<float> &xyz, const std::vector<int> &camera_source, const std::vector<float> &normals) {
  xyz_ = xyz;
  camera_source_ = camera_source;
  normals_ = normals;
}

void Cloud::filterWorkspaceSamples(const std::vector<double> &ws) {
  if (ws.size() < 6) return;
  auto inside = [&](double x, double y, double z) { return x > ws[0] && x < ws[1] && y > ws[2] && y < ws[3] && z > ws[4] && z < ws[5]; };
  if (!sample_indices_.empty()) {
    std::vector<int> keep;
    for (int i = 0; i < (int)sample_indices_.size(); i++) {
      const float *p = &xyz_[3 * (size_t)sample_indices_[i]];
      if (inside(p[0], p[1], p[2])) keep.push_back(i);  // the position, as cloud.cpp:215 stores it
    }
    sample_indices_ = keep;
    std::cout << sample_indices_.size() << " sample indices left after workspace filtering \n";
  }
  if (samples_.size() >= 3) {
    std::vector<double> keep;
    for (size_t i = 0; i + 2 < samples_.size(); i += 3)
      if (inside(samples_[i], samples_[i + 1], samples_[i + 2])) keep.insert(keep.end(), samples_.begin() + i, samples_.begin() + i + 3);
    samples_ = keep;
    std::cout << samples_.size() / 3 << " samples left after workspace filtering \n";
  }
}

void Cloud::filterWorkspace(const std::vector<double> &ws) {
  if (ws.size() < 6) return;
  auto inside = [&](double x, double y, double z) { return x > ws[0] && x < ws[1] && y > ws[2] && y < ws[3] && z > ws[4] && z < ws[5]; };
  filterWorkspaceSamples(ws);
  const size_t n = size();
  const int cams = numCameras();
  const bool with_normals = hasNormals();
  std::vector<size_t> idx;
  for (size_t i = 0; i < n; i++)
    if (inside(xyz_[3 * i], xyz_[3 * i + 1], xyz_[3 * i + 2])) idx.push_back(i);
  std::vector<float> xyz(idx.size() * 3), normals(with_normals ? idx.size() * 3 : 0);
  std::vector<int> cam((size_t)cams * idx.size());
  for (size_t k = 0; k < idx.size(); k++) {
    for (int r = 0; r < 3; r++) {
      xyz[3 * k + r] = xyz_[3 * idx[k] + r];
      if (with_normals) normals[3 * k + r] = normals_[3 * idx[k] + r];
    }
    for (int c = 0; c < cams; c++) cam[(size_t)c * idx.size() + k] = camera_source_[(size_t)c * n + idx[k]];
  }
  xyz_ = xyz;
  normals_ = normals;
  camera_source_ = cam;
}

void Cloud::setNormalsFromFile(const std::string &filename) {
  std::ifstream in(filename.c_str());
  std::string line;
  std::vector<float> normals(xyz_.size(), 0.f);
  size_t i = 0;
  while (i < 3 && std::getline(in, line)) {
    std::stringstream ls(line);
    std::string cell;
    size_t j = 0;
    while (std::getline(ls, cell, ',')) {
      if (j < size()) {
        char *end = nullptr;
        const double v = std::strtod(cell.c_str(), &end);
        if (end == cell.c_str()) {  // not a number: the file is refused as a whole (std::stod would throw)
          printf("ERROR: cannot parse the normals file %s\n", filename.c_str());
          return;
        }
        normals[3 * j + i] = (float)v;
      }
      j++;
    }
    i++;
  }
  if (i == 3) normals_ = normals;
}

void Cloud::voxelizeCloud(float cell_size) {
  const int n = (int)size();
  if (n == 0) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) mn[c] = std::min(mn[c], xyz_[3 * i + c]);
  std::set<std::array<int, 4>, VoxelDiffers> bins;
  for (int i = 0; i < n; i++) {
    std::array<int, 4> v;
    for (int c = 0; c < 3; c++) v[c] = (int)std::floor((xyz_[3 * i + c] - mn[c]) / cell_size);
    v[3] = i;
    bins.insert(v);
  }
  const int cams = numCameras();
  std::vector<float> out;
  std::vector<int> cam((size_t)cams * bins.size());
  size_t k = 0;
  for (const auto &v : bins) {
    for (int c = 0; c < 3; c++) out.push_back(mn[c] + cell_size * (float)v[c]);
    for (int j = 0; j < cams; j++) cam[(size_t)j * bins.size() + k] = camera_source_[(size_t)j * n + v[3]] == 1 ? 1 : 0;
    k++;
  }
  xyz_ = out;
  camera_source_ = cam;
  normals_.clear();
  sample_indices_.clear();
  printf("Voxelized cloud: %zu\n", size());
}

// Cloud::subsample (cloud.cpp:350-405): samples given by coordinates are thinned to num_samples of them without
// repetition (subsampleSamples), sample indices are redrawn num_samples times WITH repetition
// (subsampleSampleIndices: sample_indices_[rand() % size]), otherwise num_samples points are drawn uniformly
// (subsampleUniformly).  The reference's generators are time-seeded; here one seeded xorshift.
void Cloud::subsample(int num_samples, unsigned seed) {
  if (num_samples <= 0) return;
  uint64_t s = 0x9E3779B97F4A7C15ull ^ seed;
  auto next = [&s]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  if (samples_.size() >= 3) {
    const int have = (int)(samples_.size() / 3);
    if (num_samples >= have) return;
    printf("Using %d out of %d available samples.\n", num_samples, have);
    std::vector<int> seq(have);
    std::iota(seq.begin(), seq.end(), 0);
    for (int i = 0; i < num_samples; i++) std::swap(seq[i], seq[i + (int)(next() % (uint64_t)(have - i))]);
    std::vector<double> sub((size_t)num_samples * 3);
    for (int i = 0; i < num_samples; i++)
      for (int r = 0; r < 3; r++) sub[3 * (size_t)i + r] = samples_[3 * (size_t)seq[i] + r];
    samples_ = sub;
    return;
  }
  if (!sample_indices_.empty()) {
    if (num_samples >= (int)sample_indices_.size()) return;
    std::vector<int> indices(num_samples);
    for (int i = 0; i < num_samples; i++) indices[i] = sample_indices_[next() % sample_indices_.size()];
    sample_indices_ = indices;
    return;
  }
  const int n = (int)size();
  if (n == 0) return;
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  const int m = std::min(num_samples, n);
  for (int i = 0; i < m; i++) std::swap(idx[i], idx[i + (int)(next() % (uint64_t)(n - i))]);
  idx.resize(m);
  sample_indices_ = idx;
}

}  // namespace util

// ------------------------------------------------------------------ candidate::Hand
namespace candidate {
void Hand::print() const {  // hand.cpp:66-79
  auto p = getPosition(), a = getApproach(), b = getBinormal(), x = getAxis();
  printf("position: %g %g %g\napproach: %g %g %g\nbinormal: %g %g %g\naxis: %g %g %g\nscore: %g\n", p[0], p[1], p[2], a[0], a[1],
         a[2], b[0], b[1], b[2], x[0], x[1], x[2], getScore());
  printf("full-antipodal: %d\nhalf-antipodal: %d\nclosing box:\n bottom: %g\n top: %g\n center: %g\n", isFullAntipodal(),
         isHalfAntipodal(), getBottom(), getTop(), getCenter());
}
}  // namespace candidate

// ------------------------------------------------------------------ net::Classifier
namespace net {

std::shared_ptr<Classifier> Classifier::create(const std::string &model_file, const std::string &weights_file, Device device,
                                               int batch_size) {
  // one back-end is compiled in, as in the reference's builds (classifier.cpp:46-62); see classifier.h
  if (device != Device::eGPU)
    printf("NOTE: classifier device = %d requested; this build's only back-end (HipClassifier) runs on the GPU.\n", (int)device);
  auto c = std::make_shared<HipClassifier>(model_file, weights_file, device, batch_size);
  if (!c->ok()) return nullptr;
  return c;
}

std::vector<float> HipClassifier::readBinaryFileIntoVector(const std::string &location) {
  std::vector<float> vals;
  std::ifstream file(location.c_str(), std::ios::binary | std::ios::in);
  if (!file.is_open()) {
    std::cout << "ERROR: Cannot open file: " << location << "!\n";
    return vals;
  }
  float x;
  while (file.read(reinterpret_cast<char *>(&x), sizeof(float))) vals.push_back(x);
  return vals;
}

HipClassifier::HipClassifier(const std::string &, const std::string &weights_file, Classifier::Device, int batch_size)
    : batch_size_(batch_size) {
  const std::string &dir = weights_file;
  static const char *files[8] = {"conv1_weights.bin", "conv1_biases.bin", "conv2_weights.bin", "conv2_biases.bin",
                                 "ip1_weights.bin",   "ip1_biases.bin",   "ip2_weights.bin",   "ip2_biases.bin"};
  for (int i = 0; i < 8; i++) params_[i] = readBinaryFileIntoVector(dir + files[i]);
  const auto &c1w = params_[0];
  if (c1w.size() % 500 != 0 || c1w.empty() || params_[1].size() != 20 || params_[2].size() != 25000 || params_[3].size() != 50 ||
      params_[4].size() != 3600000 || params_[5].size() != 500 || params_[6].size() != 1000 || params_[7].size() != 2) {
    printf("ERROR: LeNet parameter files in %s are missing or have unexpected sizes\n", dir.c_str());
    return;
  }
  channels_ = (int)(c1w.size() / 500);
  gpd_params p;
  gpd_hip_default_params(&p);
  p.image_num_channels = channels_;
  if (gpd_hip_create(0, &p, &ctx_) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    ctx_ = nullptr;
    return;
  }
  if (gpd_hip_set_lenet_weights(ctx_, channels_, params_[0].data(), params_[1].data(), params_[2].data(), params_[3].data(),
                                params_[4].data(), params_[5].data(), params_[6].data(), params_[7].data()) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return;
  }
  loaded_ = true;
}

HipClassifier::~HipClassifier() {
  if (ctx_) gpd_hip_destroy(ctx_);
}

bool HipClassifier::setScoringMode(int mode) {
  if (!ok()) return false;
  if (gpd_hip_set_lenet_mode(ctx_, mode) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  return true;
}

std::vector<float> HipClassifier::classifyImages(const std::vector<std::unique_ptr<Image>> &image_list) {
  std::vector<float> out(image_list.size(), 0.f);
  if (!ok()) return out;
  const size_t bytes = (size_t)60 * 60 * channels_;
  std::vector<uint8_t> batch;
  std::vector<size_t> where;
  batch.reserve(image_list.size() * bytes);
  for (size_t i = 0; i < image_list.size(); i++) {
    const Image &im = *image_list[i];
    // non-continuous images are skipped and keep score 0 (eigen_classifier.cpp:68)
    if (!im.isContinuous() || im.rows != 60 || im.cols != 60 || im.channels() != channels_ || im.data.size() != bytes) continue;
    batch.insert(batch.end(), im.data.begin(), im.data.end());
    where.push_back(i);
  }
  std::vector<float> s(where.size());
  if (!where.empty() && gpd_hip_score(ctx_, batch.data(), (int)where.size(), s.data()) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return out;
  }
  for (size_t k = 0; k < where.size(); k++) out[where[k]] = s[k];
  return out;
}

}  // namespace net

// ------------------------------------------------------------------ GraspDetector
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static std::string dir_of(const std::string &path) {
  size_t s = path.find_last_of('/');
  return s == std::string::npos ? std::string("./") : path.substr(0, s + 1);
}
// cfg paths are relative to the working directory in the reference; also try the cfg file's own directory
static std::string resolve(const std::string &p, const std::string &cfg_dir) {
  if (p.empty() || p[0] == '/') return p;
  std::ifstream a(p.c_str());
  if (a.good()) return p;
  return cfg_dir + p;
}

GraspDetector::GraspDetector(const std::string &config_filename) {
  util::ConfigFile config_file(config_filename);
  if (!config_file.ExtractKeys()) return;
  const std::string cfg_dir = dir_of(config_filename);
  gpd_hip_default_params(&params_);
  // hand geometry (grasp_detector.cpp:13-19, hand_geometry.cpp:22-30)
  std::string hg = config_file.getValueOfKeyAsString("hand_geometry_filename", "");
  util::ConfigFile hand_cfg(hg == "0" || hg.empty() ? config_filename : resolve(hg, cfg_dir));
  hand_cfg.ExtractKeys();
  params_.finger_width = hand_cfg.getValueOfKey<double>("finger_width", 0.01);
  params_.hand_outer_diameter = hand_cfg.getValueOfKey<double>("hand_outer_diameter", 0.12);
  params_.hand_depth = hand_cfg.getValueOfKey<double>("hand_depth", 0.06);
  params_.hand_height = hand_cfg.getValueOfKey<double>("hand_height", 0.02);
  params_.init_bite = hand_cfg.getValueOfKey<double>("init_bite", 0.01);
  // candidate generation (grasp_detector.cpp:47-88)
  num_samples_ = config_file.getValueOfKey<int>("num_samples", 1000);
  workspace_ = config_file.getValueOfKeyAsStdVectorDouble("workspace", "-1 1 -1 1 -1 1");
  workspace_.resize(6, 0.0);
  voxelize_ = config_file.getValueOfKey<bool>("voxelize", true);
  voxel_size_ = config_file.getValueOfKey<double>("voxel_size", 0.003);
  normals_radius_ = config_file.getValueOfKey<double>("normals_radius", 0.03);
  params_.nn_radius_frames = config_file.getValueOfKey<double>("nn_radius", 0.01);
  params_.num_orientations = config_file.getValueOfKey<int>("num_orientations", 8);
  params_.num_finger_placements = config_file.getValueOfKey<int>("num_finger_placements", 10);
  params_.deepen_hand = config_file.getValueOfKey<bool>("deepen_hand", true) ? 1 : 0;
  std::vector<int> axes = config_file.getValueOfKeyAsStdVectorInt("hand_axes", "2");
  params_.num_hand_axes = (int)std::min<size_t>(axes.size(), 3);
  for (int i = 0; i < params_.num_hand_axes; i++) params_.hand_axes[i] = axes[i];
  params_.friction_coeff = config_file.getValueOfKey<double>("friction_coeff", 20.0);
  params_.min_viable = config_file.getValueOfKey<int>("min_viable", 6);
  // image geometry (grasp_detector.cpp:120-127, image_geometry.cpp:21-29)
  std::string ig = config_file.getValueOfKeyAsString("image_geometry_filename", "");
  util::ConfigFile img_cfg(ig == "0" || ig.empty() ? config_filename : resolve(ig, cfg_dir));
  img_cfg.ExtractKeys();
  params_.volume_width = img_cfg.getValueOfKey<double>("volume_width", 0.10);
  params_.volume_depth = img_cfg.getValueOfKey<double>("volume_depth", 0.06);
  params_.volume_height = img_cfg.getValueOfKey<double>("volume_height", 0.02);
  params_.image_size = img_cfg.getValueOfKey<int>("image_size", 60);
  params_.image_num_channels = img_cfg.getValueOfKey<int>("image_num_channels", 15);
  // filtering and selection (grasp_detector.cpp:157-185)
  workspace_grasps_ = config_file.getValueOfKeyAsStdVectorDouble("workspace_grasps", "-1 1 -1 1 -1 1");
  workspace_grasps_.resize(6, 0.0);
  for (int i = 0; i < 6; i++) params_.workspace_grasps[i] = workspace_grasps_[i];
  params_.min_aperture = config_file.getValueOfKey<double>("min_aperture", 0.0);
  params_.max_aperture = config_file.getValueOfKey<double>("max_aperture", 0.085);
  // approach-direction filter, clustering, selection (grasp_detector.cpp:168-185)
  filter_approach_direction_ = config_file.getValueOfKey<bool>("filter_approach_direction", false);
  std::vector<double> approach = config_file.getValueOfKeyAsStdVectorDouble("direction", "1 0 0");
  approach.resize(3, 0.0);
  direction_ = {approach[0], approach[1], approach[2]};
  thresh_rad_ = config_file.getValueOfKey<double>("thresh_rad", 2.3);
  // the fused device entries apply the direction filter themselves, right after the workspace filter (gpd_params)
  params_.filter_approach_direction = filter_approach_direction_ ? 1 : 0;
  for (int i = 0; i < 3; i++) params_.direction[i] = direction_[i];
  params_.thresh_rad = thresh_rad_;
  const int min_inliers = config_file.getValueOfKey<int>("min_inliers", 1);
  clustering_ = std::make_unique<Clustering>(min_inliers);  // runs on ctx_ once that exists (below)
  cluster_grasps_ = min_inliers > 0;
  num_selected_ = config_file.getValueOfKey<int>("num_selected", 100);
  use_file_normals_ = config_file.getValueOfKey<int>("use_file_normals", 0) != 0;
  // Preprocessing steps of CandidatesGenerator::preprocessPointCloud that are PCL algorithms of their own and are not
  // restated here (candidates_generator.cpp:28-34, grasp_detector.cpp:52-63; all off in the shipped cfg files): a cloud
  // that needs them has to go through them before it gets here.  Refused, not skipped: a silently different cloud would
  // give silently different grasps.
  const struct {
    const char *key, *what;
  } unsupported[] = {{"remove_outliers", "pcl::StatisticalOutlierRemoval (cloud.cpp:166-174)"},
                     {"sample_above_plane", "a RANSAC plane fit, pcl::SACSegmentation (cloud.cpp:407-436)"},
                     {"refine_normals_k", "pcl::NormalRefinement (cloud.cpp:176-204)"},
                     {"remove_plane_before_image_calculation",
                      "ImageGenerator::removePlane, pcl::SACSegmentation (image_generator.cpp:32, 101-125): it changes the point list behind every grasp image"}};
  for (const auto &u : unsupported)
    if (config_file.getValueOfKey<int>(u.key, 0) != 0) {
      printf("ERROR: %s = %d asks for %s, which this build does not have; unset it or preprocess the cloud beforehand\n", u.key,
             config_file.getValueOfKey<int>(u.key, 0), u.what);
      return;  // ok() stays false
    }
  printf("============ CANDIDATE GENERATION ============\n");
  printf("num_samples: %d\nnn_radius: %3.2f\nnum_orientations: %d\nnum_finger_placements: %d\ndeepen_hand: %s\n", num_samples_,
         params_.nn_radius_frames, params_.num_orientations, params_.num_finger_placements, params_.deepen_hand ? "true" : "false");
  printf("==============================================\n");
  if (gpd_hip_create(config_file.getValueOfKey<int>("hip_device", 0), &params_, &ctx_) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    ctx_ = nullptr;
    return;
  }
  if (ctx_) clustering_->setContext(ctx_);
  // classifier (grasp_detector.cpp:129-145): created through the plugin factory, as the reference does.  For the
  // Eigen-layout parameters weights_file is the parameter directory.
  std::string model_file = config_file.getValueOfKeyAsString("model_file", "");
  std::string weights_file = config_file.getValueOfKeyAsString("weights_file", "");
  if (!model_file.empty() || !weights_file.empty()) {
    const int device = config_file.getValueOfKey<int>("device", 0);
    const int batch_size = config_file.getValueOfKey<int>("batch_size", 1);
    const std::string dir = resolve(weights_file, cfg_dir);
    classifier_ = net::Classifier::create(model_file, dir, static_cast<net::Classifier::Device>(device), batch_size);
    printf("============ CLASSIFIER ======================\nmodel_file: %s\nweights_file: %s\nbatch_size: %d\n"
           "==============================================\n",
           model_file.c_str(), dir.c_str(), batch_size);
    // detectGrasps normally runs search -> images -> scores fused on the device, in ctx_: the HIP back-end's
    // parameters are loaded there as well.  Any other Classifier (or cfg classifier_plugin_route = 1) is driven
    // through classifyImages, the reference's own sequence (grasp_detector.cpp:261-273).
    auto *hc = dynamic_cast<net::HipClassifier *>(classifier_.get());
    if (!classifier_) {
      printf("ERROR: could not create the classifier from %s\n", dir.c_str());
    } else if (!hc) {
      has_classifier_ = true;  // a foreign back-end: plugin route only
    } else if (hc->channels() != params_.image_num_channels) {
      printf("ERROR: the classifier's %d channels do not match image_num_channels = %d\n", hc->channels(), params_.image_num_channels);
      classifier_.reset();
    } else if (gpd_hip_set_lenet_weights(ctx_, hc->channels(), hc->parameter(0).data(), hc->parameter(1).data(),
                                         hc->parameter(2).data(), hc->parameter(3).data(), hc->parameter(4).data(),
                                         hc->parameter(5).data(), hc->parameter(6).data(), hc->parameter(7).data()) == GPD_OK) {
      has_classifier_ = true;
      fused_classifier_ = true;
      // hip_lenet_mode (not a reference key): 0 = split operands on the int8 / bf16 matrix pipes (default), 1 = the f32
      // chain in EigenClassifier's operation order; both the fused path's context and the plugin's own take it
      const int mode = config_file.getValueOfKey<int>("hip_lenet_mode", 0);
      if (mode != 0) {
        bool set = gpd_hip_set_lenet_mode(ctx_, mode) == GPD_OK;
        if (!set) printf("ERROR: hip_lenet_mode = %d: %s\n", mode, gpd_hip_last_error());
        set = set && hc->setScoringMode(mode);
        if (!set) {
          has_classifier_ = fused_classifier_ = false;
          classifier_.reset();
        } else {
          printf("hip_lenet_mode: %d\n", mode);
        }
      }
    } else {
      printf("ERROR: %s\n", gpd_hip_last_error());
      classifier_.reset();
    }
    plugin_route_ = config_file.getValueOfKey<int>("classifier_plugin_route", 0) != 0;
  }
}

GraspDetector::GraspDetector(const candidate::HandSearch::Parameters &hs, const descriptor::ImageGeometry &ig, int hip_device) {
  gpd_hip_default_params(&params_);
  params_.finger_width = hs.hand_geometry_.finger_width_;
  params_.hand_outer_diameter = hs.hand_geometry_.outer_diameter_;
  params_.hand_depth = hs.hand_geometry_.depth_;
  params_.hand_height = hs.hand_geometry_.height_;
  params_.init_bite = hs.hand_geometry_.init_bite_;
  params_.nn_radius_frames = hs.nn_radius_frames_;
  params_.num_orientations = hs.num_orientations_;
  params_.num_finger_placements = hs.num_finger_placements_;
  params_.deepen_hand = hs.deepen_hand_ ? 1 : 0;
  params_.num_hand_axes = (int)std::min<size_t>(hs.hand_axes_.size(), 3);
  for (int i = 0; i < params_.num_hand_axes; i++) params_.hand_axes[i] = hs.hand_axes_[i];
  params_.friction_coeff = hs.friction_coeff_;
  params_.min_viable = hs.min_viable_;
  params_.volume_width = ig.outer_diameter_;
  params_.volume_depth = ig.depth_;
  params_.volume_height = ig.height_;
  params_.image_size = ig.size_;
  params_.image_num_channels = ig.num_channels_;
  num_samples_ = hs.num_samples_;
  voxelize_ = false;
  workspace_ = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};
  workspace_grasps_ = workspace_;
  for (int i = 0; i < 6; i++) params_.workspace_grasps[i] = workspace_grasps_[i];
  clustering_ = std::make_unique<Clustering>(0);
  if (gpd_hip_create(hip_device, &params_, &ctx_) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    ctx_ = nullptr;
  }
  if (ctx_) clustering_->setContext(ctx_);
}

candidate::HandSearch::Parameters GraspDetector::getHandSearchParameters() const {
  candidate::HandSearch::Parameters p;
  p.nn_radius_frames_ = params_.nn_radius_frames;
  p.num_threads_ = 1;
  p.num_samples_ = num_samples_;
  p.num_orientations_ = params_.num_orientations;
  p.num_finger_placements_ = params_.num_finger_placements;
  p.hand_axes_.assign(params_.hand_axes, params_.hand_axes + params_.num_hand_axes);
  p.deepen_hand_ = params_.deepen_hand != 0;
  p.friction_coeff_ = params_.friction_coeff;
  p.min_viable_ = params_.min_viable;
  p.hand_geometry_ = {params_.finger_width, params_.hand_outer_diameter, params_.hand_depth, params_.hand_height, params_.init_bite};
  return p;
}

GraspDetector::~GraspDetector() {
  if (ctx_) gpd_hip_destroy(ctx_);
}

bool GraspDetector::calculateNormals(util::Cloud &cloud, double radius) {
  if (!ctx_ || cloud.size() == 0) return false;
  std::vector<float> zeros(cloud.size() * 3, 0.f), normals(cloud.size() * 3, 0.f);
  printf("Calculating surface normals ...\n");
  if (gpd_hip_upload_cloud(ctx_, cloud.getCloudProcessed().data(), zeros.data(), (int)cloud.size(), cloud.getCameraSource().data(),
                           cloud.numCameras(), cloud.getViewPoints().data()) != GPD_OK ||
      gpd_hip_estimate_normals(ctx_, radius, normals.data()) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  last_num_sets_ = 0;
  cloud.setNormals(normals);
  return true;
}

// CandidatesGenerator::preprocessPointCloud (candidates_generator.cpp:14-37): workspace cut, voxelise when enabled,
// normals ALWAYS recomputed, subsample.  The reference never reads normals from a PCD; normals that came with the
// file are used only when the cfg says so (use_file_normals = 1: no voxelisation then either, the normals belong to
// the points as loaded) — the synthetic benchmark clouds ship their analytic normals this way.
// The workspace cut and the voxeliser run on the device (gpd_hip_preprocess_cloud); util::Cloud's own
// filterWorkspace / voxelizeCloud are the host-side API mirror for callers that hold a Cloud and no detector.
void GraspDetector::preprocessPointCloud(util::Cloud &cloud) {
  printf("Processing cloud with %zu points.\n", cloud.size());
  const bool keep_normals = use_file_normals_ && cloud.hasNormals();
  if (!ctx_) {
    printf("ERROR: no device context\n");
    return;
  }
  cloud.filterWorkspaceSamples(workspace_);  // candidates_generator.cpp:19 (NaN rows are dropped at load time)
  const int n = (int)cloud.size(), cams = cloud.numCameras();
  if (n > 0) {
    std::vector<float> xyz((size_t)n * 3);
    std::vector<int> cam((size_t)n * cams), src(n);
    int m = 0;
    const bool voxelise = voxelize_ && !keep_normals;
    if (gpd_hip_preprocess_cloud(ctx_, cloud.getCloudProcessed().data(), cloud.getCameraSource().data(), n, cams,
                                 workspace_.size() >= 6 ? workspace_.data() : nullptr, voxelise ? (float)voxel_size_ : 0.f, xyz.data(),
                                 cam.data(), src.data(), &m, nullptr) != GPD_OK) {
      printf("ERROR: %s\n", gpd_hip_last_error());
      return;
    }
    xyz.resize((size_t)m * 3);
    cam.resize((size_t)m * cams);
    std::vector<float> normals;
    if (keep_normals) {
      normals.resize((size_t)m * 3);
      for (int k = 0; k < m; k++)
        for (int r = 0; r < 3; r++) normals[3 * (size_t)k + r] = cloud.getNormals()[3 * (size_t)src[k] + r];
    }
    cloud.setProcessed(xyz, cam, normals);
    if (voxelise) printf("Voxelized cloud: %zu\n", cloud.size());
  }
  if (!keep_normals && cloud.size() > 0 && !calculateNormals(cloud, normals_radius_)) return;
  cloud.subsample(num_samples_);
}

bool GraspDetector::upload(const util::Cloud &cloud) {
  if (!ctx_) return false;
  if (cloud.size() == 0) {
    printf("ERROR: Point cloud is empty!");
    return false;
  }
  if (!cloud.hasNormals()) {
    printf("ERROR: the cloud has no normals (call preprocessPointCloud first)\n");
    return false;
  }
  if (gpd_hip_upload_cloud(ctx_, cloud.getCloudProcessed().data(), cloud.getNormals().data(), (int)cloud.size(),
                           cloud.getCameraSource().data(), cloud.numCameras(), cloud.getViewPoints().data()) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  return true;
}

static std::vector<std::unique_ptr<candidate::HandSet>> to_sets(const std::vector<gpd_hand> &recs, int n_sets, int slots) {
  std::vector<std::unique_ptr<candidate::HandSet>> sets(n_sets);
  for (int s = 0; s < n_sets; s++) {
    sets[s] = std::make_unique<candidate::HandSet>();
    sets[s]->is_valid_.resize(slots);
    for (int j = 0; j < slots; j++) {
      const gpd_hand &r = recs[(size_t)s * slots + j];
      sets[s]->hands_.push_back(std::make_unique<candidate::Hand>(r));
      sets[s]->is_valid_[j] = r.valid != 0;
    }
    sets[s]->sample_ = {recs[(size_t)s * slots].sample[0], recs[(size_t)s * slots].sample[1], recs[(size_t)s * slots].sample[2]};
  }
  return sets;
}

bool GraspDetector::searchDevice(const util::Cloud &cloud, bool fused, std::vector<gpd_hand> &recs, int &n_sets, int &n_cand) {
  const std::vector<double> &samples = cloud.getSamples();
  const std::vector<int> &idx = cloud.getSampleIndices();
  const int slots = params_.num_hand_axes * params_.num_orientations;
  n_sets = n_cand = 0;
  int rc;
  if (samples.size() >= 3) {  // use samples (hand_search.cpp:37-39)
    const int S = (int)(samples.size() / 3);
    recs.assign((size_t)S * slots, gpd_hand());
    rc = fused ? gpd_hip_detect_samples(ctx_, samples.data(), S, recs.data(), &n_sets, &n_cand)
               : gpd_hip_search_samples(ctx_, samples.data(), S, recs.data(), &n_sets);
  } else if (!idx.empty()) {  // use indices (:40-43)
    recs.assign(idx.size() * slots, gpd_hand());
    rc = fused ? gpd_hip_detect(ctx_, idx.data(), (int)idx.size(), recs.data(), &n_sets, &n_cand)
               : gpd_hip_search(ctx_, idx.data(), (int)idx.size(), recs.data(), &n_sets);
  } else {
    std::cout << "Error: No samples or no indices!\n";  // hand_search.cpp:44-48
    return false;
  }
  if (rc != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  last_num_sets_ = n_sets;
  last_samples_.resize((size_t)n_sets * 3);
  for (int s = 0; s < n_sets; s++)
    for (int r = 0; r < 3; r++) last_samples_[3 * (size_t)s + r] = recs[(size_t)s * slots].sample[r];
  return true;
}

// grasp_detector.cpp:522-526
std::vector<int> GraspDetector::evalGroundTruth(const util::Cloud &cloud_gt, std::vector<std::unique_ptr<candidate::Hand>> &hands) {
  std::vector<int> labels(hands.size(), 0);
  if (hands.empty() || !upload(cloud_gt)) return labels;
  last_num_sets_ = 0;  // the device search buffers now belong to cloud_gt
  std::vector<gpd_hand> recs(hands.size());
  for (size_t i = 0; i < hands.size(); i++) recs[i] = hands[i]->record();
  std::vector<int32_t> l(hands.size(), 0);
  if (gpd_hip_reevaluate(ctx_, recs.data(), (int)recs.size(), l.data()) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return labels;
  }
  for (size_t i = 0; i < hands.size(); i++) {
    labels[i] = l[i];
    hands[i]->setHalfAntipodal(recs[i].half_antipodal != 0);
    hands[i]->setFullAntipodal(recs[i].full_antipodal != 0);
  }
  return labels;
}

std::vector<std::unique_ptr<candidate::HandSet>> GraspDetector::generateGraspCandidates(const util::Cloud &cloud) {
  std::vector<std::unique_ptr<candidate::HandSet>> none;
  if (!upload(cloud)) return none;
  const int slots = params_.num_hand_axes * params_.num_orientations;
  std::vector<gpd_hand> recs;
  int n_sets = 0, n_cand = 0;
  if (!searchDevice(cloud, false, recs, n_sets, n_cand)) return none;
  return to_sets(recs, n_sets, slots);
}

// grasp_detector.cpp:334-398 (the right_top typo at :360-363 is kept)
std::vector<std::unique_ptr<candidate::HandSet>> GraspDetector::filterGraspsWorkspace(
    std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, const std::vector<double> &workspace) const {
  int remaining = 0;
  std::vector<std::unique_ptr<candidate::HandSet>> out;
  printf("Filtering grasps outside of workspace ...\n");
  for (size_t i = 0; i < hand_set_list.size(); i++) {
    const auto &hands = hand_set_list[i]->getHands();
    std::vector<bool> is_valid = hand_set_list[i]->getIsValid();
    bool any = false;
    for (size_t j = 0; j < hands.size(); j++) {
      if (!is_valid[j]) continue;
      const double half_width = 0.5 * params_.hand_outer_diameter;
      auto pos = hands[j]->getPosition(), bin = hands[j]->getBinormal(), app = hands[j]->getApproach();
      bool ok = hands[j]->getGraspWidth() >= params_.min_aperture && hands[j]->getGraspWidth() <= params_.max_aperture;
      for (int r = 0; r < 3 && ok; r++) {
        const double lb = pos[r] + half_width * bin[r], rb = pos[r] - half_width * bin[r];
        const double lt = lb + params_.hand_depth * app[r], rt = lb + params_.hand_depth * app[r];
        const double ap = pos[r] - 0.05 * app[r];
        const double mn = std::min({lb, rb, lt, rt, ap}), mx = std::max({lb, rb, lt, rt, ap});
        ok = mn >= workspace[2 * r] && mx <= workspace[2 * r + 1];
      }
      is_valid[j] = ok;
      if (ok) {
        remaining++;
        any = true;
      }
    }
    if (any) {
      hand_set_list[i]->setIsValid(is_valid);
      out.push_back(std::move(hand_set_list[i]));
    }
  }
  printf("Number of grasp candidates within workspace and gripper width: %d\n", remaining);
  return out;
}

// grasp_detector.cpp:422-453
std::vector<std::unique_ptr<candidate::HandSet>> GraspDetector::filterGraspsDirection(
    std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, const std::array<double, 3> &direction, double thresh_rad) {
  std::vector<std::unique_ptr<candidate::HandSet>> out;
  int remaining = 0;
  for (size_t i = 0; i < hand_set_list.size(); i++) {
    const auto &hands = hand_set_list[i]->getHands();
    std::vector<bool> is_valid = hand_set_list[i]->getIsValid();
    bool any = false;
    for (size_t j = 0; j < hands.size(); j++) {
      if (!is_valid[j]) continue;
      const auto app = hands[j]->getApproach();
      const double angle = acos(direction[0] * app[0] + direction[1] * app[1] + direction[2] * app[2]);
      if (angle > thresh_rad) {
        is_valid[j] = false;
      } else {
        remaining++;
        any = true;
      }
    }
    if (any) {
      hand_set_list[i]->setIsValid(is_valid);
      out.push_back(std::move(hand_set_list[i]));
    }
  }
  printf("Number of grasp candidates with correct approach direction: %d\n", remaining);
  return out;
}

// the flat record array gpd_hip_images reads: one row of slots per set of the last search, rows of
// sets that are not in `sets` stay invalid
std::vector<gpd_hand> GraspDetector::flatten(const std::vector<std::unique_ptr<candidate::HandSet>> &sets) const {
  const int slots = params_.num_hand_axes * params_.num_orientations;
  std::vector<gpd_hand> recs((size_t)last_num_sets_ * slots, gpd_hand());
  for (const auto &hs : sets) {
    if (!hs || hs->getHands().empty()) continue;
    const int si = hs->getHands()[0]->record().set_index;
    if (si < 0 || si >= last_num_sets_) continue;
    for (int j = 0; j < slots && j < (int)hs->getHands().size(); j++) {
      recs[(size_t)si * slots + j] = hs->getHands()[j]->record();
      recs[(size_t)si * slots + j].valid = hs->getIsValid()[j] ? 1 : 0;
    }
  }
  return recs;
}

// grasp_detector.cpp:528-551
std::vector<std::unique_ptr<candidate::Hand>> GraspDetector::pruneGraspCandidates(
    const util::Cloud &cloud, const std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list, double min_score) {
  std::vector<std::unique_ptr<candidate::Hand>> hands_out;
  if (!ctx_ || !has_classifier_ || hand_set_list.empty()) return hands_out;
  const int slots = params_.num_hand_axes * params_.num_orientations;
  // do the sets still match the neighbourhoods on the device (same search, same numbering)?
  bool resident = last_num_sets_ > 0;
  for (const auto &hs : hand_set_list) {
    if (!resident) break;
    if (!hs || hs->getHands().empty()) continue;
    const gpd_hand &r0 = hs->getHands()[0]->record();
    resident = r0.set_index >= 0 && r0.set_index < last_num_sets_;
    for (int r = 0; r < 3 && resident; r++) resident = last_samples_[3 * (size_t)r0.set_index + r] == r0.sample[r];
  }
  std::vector<gpd_hand> recs;
  int n_sets = last_num_sets_;
  if (resident) {
    recs = flatten(hand_set_list);
  } else {
    // a list gathered over several searches: search the neighbourhoods of its samples again (the
    // samples are the same doubles, so the neighbourhoods are the same) and number the sets as listed
    if (!upload(cloud)) return hands_out;
    std::vector<double> samples;
    std::vector<const candidate::HandSet *> sets;
    for (const auto &hs : hand_set_list) {
      if (!hs || hs->getHands().empty()) continue;
      const gpd_hand &r0 = hs->getHands()[0]->record();
      for (int r = 0; r < 3; r++) samples.push_back(r0.sample[r]);
      sets.push_back(hs.get());
    }
    if (sets.empty()) return hands_out;
    std::vector<gpd_hand> fresh(sets.size() * slots);
    if (gpd_hip_search_samples(ctx_, samples.data(), (int)sets.size(), fresh.data(), &n_sets) != GPD_OK) {
      printf("ERROR: %s\n", gpd_hip_last_error());
      return hands_out;
    }
    if (n_sets != (int)sets.size()) {
      printf("ERROR: pruneGraspCandidates: %zu hand sets do not belong to this cloud\n", sets.size() - (size_t)n_sets);
      return hands_out;
    }
    last_num_sets_ = n_sets;
    last_samples_ = samples;
    recs.assign((size_t)n_sets * slots, gpd_hand());
    for (int s = 0; s < n_sets; s++)
      for (int j = 0; j < slots && j < (int)sets[s]->getHands().size(); j++) {
        gpd_hand r = sets[s]->getHands()[j]->record();
        r.set_index = s;
        r.valid = sets[s]->getIsValid()[j] ? 1 : 0;
        recs[(size_t)s * slots + j] = r;
      }
  }
  size_t nv = 0;
  for (const gpd_hand &r : recs) nv += r.valid;
  std::vector<int32_t> cand(nv);
  std::vector<float> scores(nv);
  int n_cand = 0;
  if (gpd_hip_images(ctx_, recs.data(), n_sets, nullptr, cand.data(), &n_cand) != GPD_OK ||
      (n_cand > 0 && gpd_hip_score(ctx_, nullptr, n_cand, scores.data()) != GPD_OK)) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return hands_out;
  }
  for (int k = 0; k < n_cand; k++) {
    if (scores[k] > min_score) {
      auto h = std::make_unique<candidate::Hand>(recs[cand[k]]);
      h->setScore(scores[k]);
      hands_out.push_back(std::move(h));
    }
  }
  return hands_out;
}

bool GraspDetector::createImages(const util::Cloud &cloud, const std::vector<std::unique_ptr<candidate::HandSet>> &hand_set_list,
                                 std::vector<std::unique_ptr<net::Image>> &images_out,
                                 std::vector<std::unique_ptr<candidate::Hand>> &hands_out) {
  (void)cloud;
  images_out.clear();
  hands_out.clear();
  if (!ctx_) return false;
  if (hand_set_list.empty()) return true;
  if ((int)hand_set_list.size() != last_num_sets_) {
    printf("ERROR: createImages: the hand sets are not those of the last generateGraspCandidates call\n");
    return false;
  }
  for (size_t s = 0; s < hand_set_list.size(); s++) {
    const auto &hs = hand_set_list[s];
    bool same = hs && !hs->getHands().empty() && hs->getHands()[0]->record().set_index == (int)s;
    for (int r = 0; r < 3 && same; r++) same = last_samples_[3 * s + r] == hs->getHands()[0]->record().sample[r];
    if (!same) {
      printf("ERROR: createImages: the hand sets are not those of the last generateGraspCandidates call\n");
      return false;
    }
  }
  std::vector<gpd_hand> recs = flatten(hand_set_list);
  const size_t bytes = (size_t)60 * 60 * params_.image_num_channels;
  size_t nv = 0;
  for (const gpd_hand &r : recs) nv += r.valid;
  std::vector<uint8_t> pix(nv * bytes);
  std::vector<int32_t> cand(nv);
  int n_cand = 0;
  if (gpd_hip_images(ctx_, recs.data(), last_num_sets_, pix.data(), cand.data(), &n_cand) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  for (int k = 0; k < n_cand; k++) {
    auto im = std::make_unique<net::Image>(60, 60, params_.image_num_channels);
    std::memcpy(im->data.data(), pix.data() + (size_t)k * bytes, bytes);
    images_out.push_back(std::move(im));
    hands_out.push_back(std::make_unique<candidate::Hand>(recs[cand[k]]));
  }
  return true;
}

bool GraspDetector::createGraspImages(util::Cloud &cloud, std::vector<std::unique_ptr<candidate::Hand>> &hands_out,
                                      std::vector<std::unique_ptr<net::Image>> &images_out) {
  hands_out.clear();
  images_out.clear();
  if (!upload(cloud)) return false;
  const int slots = params_.num_hand_axes * params_.num_orientations;
  std::vector<gpd_hand> recs;
  int n_sets = 0, n_cand = 0;
  if (!searchDevice(cloud, false, recs, n_sets, n_cand)) return false;
  printf("Generated %d hand sets.\n", n_sets);
  // workspace / aperture filter on the flat records (same predicate as filterGraspsWorkspace)
  {
    auto sets = to_sets(recs, n_sets, slots);
    std::vector<std::vector<bool>> keep(n_sets, std::vector<bool>(slots, false));
    for (int s = 0; s < n_sets; s++) sets[s]->sample_[0] = s;  // remember the set number through the move
    auto filtered = filterGraspsWorkspace(sets, workspace_grasps_);
    if (filter_approach_direction_)  // grasp_detector.cpp:505-512
      filtered = filterGraspsDirection(filtered, direction_, thresh_rad_);
    for (auto &hs : filtered) keep[(int)hs->sample_[0]] = hs->getIsValid();
    for (int s = 0; s < n_sets; s++)
      for (int j = 0; j < slots; j++) recs[(size_t)s * slots + j].valid = keep[s][j] ? 1 : 0;
  }
  const size_t bytes = (size_t)60 * 60 * params_.image_num_channels;
  size_t nv = 0;
  for (size_t i = 0; i < (size_t)n_sets * slots; i++) nv += recs[i].valid;
  std::vector<uint8_t> pix(nv * bytes);
  std::vector<int32_t> cand(nv);
  if (gpd_hip_images(ctx_, recs.data(), n_sets, pix.data(), cand.data(), &n_cand) != GPD_OK) {
    printf("ERROR: %s\n", gpd_hip_last_error());
    return false;
  }
  for (int k = 0; k < n_cand; k++) {  // order: set-major, slot-minor, valid only (image_generator.cpp:91-98)
    auto im = std::make_unique<net::Image>(60, 60, params_.image_num_channels);
    std::memcpy(im->data.data(), pix.data() + (size_t)k * bytes, bytes);
    images_out.push_back(std::move(im));
    hands_out.push_back(std::make_unique<candidate::Hand>(recs[cand[k]]));
  }
  return true;
}

std::vector<std::unique_ptr<candidate::Hand>> GraspDetector::selectGrasps(std::vector<std::unique_ptr<candidate::Hand>> &hands) const {
  printf("Selecting the %d highest scoring grasps ...\n", num_selected_);  // grasp_detector.cpp:405-420
  int middle = std::min((int)hands.size(), num_selected_);
  std::partial_sort(hands.begin(), hands.begin() + middle, hands.end(), isScoreGreater);
  std::vector<std::unique_ptr<candidate::Hand>> out;
  for (int i = 0; i < middle; i++) {
    out.push_back(std::move(hands[i]));
    printf(" grasp #%d, score: %3.4f\n", i, out[i]->getScore());
  }
  return out;
}

std::vector<std::unique_ptr<candidate::Hand>> GraspDetector::detectGrasps(const util::Cloud &cloud) {
  const double t0 = now_s();
  std::vector<std::unique_ptr<candidate::Hand>> hands_out;
  if (!upload(cloud)) return hands_out;
  if (!has_classifier_) {
    printf("ERROR: no classifier weights loaded (cfg key weights_file)\n");
    return hands_out;
  }
  const int slots = params_.num_hand_axes * params_.num_orientations;
  std::vector<gpd_hand> recs;
  int n_sets = 0, n_cand = 0;
  std::vector<std::unique_ptr<candidate::Hand>> hands;
  float ms[3] = {0, 0, 0};
  if (plugin_route_ || !fused_classifier_) {
    // the reference's own sequence through the plugin interface (grasp_detector.cpp:222-273): candidates ->
    // filterGraspsWorkspace [-> filterGraspsDirection] -> ImageGenerator::createImages -> classifier_->classifyImages
    // -> hands[i]->setScore
    util::Cloud work = cloud;
    std::vector<std::unique_ptr<net::Image>> images;
    if (!createGraspImages(work, hands, images)) return hands_out;
    float t[3];
    gpd_hip_last_stage_ms(ctx_, t);
    ms[0] = t[0];
    ms[1] = t[1];
    const double tc = now_s();
    const std::vector<float> scores = classifier_->classifyImages(images);
    ms[2] = (float)((now_s() - tc) * 1e3);
    for (size_t i = 0; i < hands.size(); i++) hands[i]->setScore(scores[i]);
  } else {
    // steps 1-4 fused on the device (grasp_detector.cpp:222-273), filterGraspsWorkspace and — when the cfg asks for it —
    // filterGraspsDirection (:247-250) included: both run at the end of the hand kernel, before the candidate list is built
    if (!searchDevice(cloud, true, recs, n_sets, n_cand)) return hands_out;
    printf("Generated %d hand sets.\n", n_sets);
    if (filter_approach_direction_) printf("Number of grasp candidates with correct approach direction: %d\n", n_cand);
    gpd_hip_last_stage_ms(ctx_, ms);
    for (size_t i = 0; i < (size_t)n_sets * slots; i++)
      if (recs[i].valid) hands.push_back(std::make_unique<candidate::Hand>(recs[i]));
  }
  // 5. select the highest scoring grasps
  hands = selectGrasps(hands);
  // 6. cluster the grasps (grasp_detector.cpp:283-303)
  std::vector<std::unique_ptr<candidate::Hand>> clusters;
  if (cluster_grasps_) {
    clusters = clustering_->findClusters(hands);
    if (clustering_->failed()) {  // a device error is not "fewer than four clusters": print-and-return-empty, as for every failure
      printf("ERROR: the clustering step failed; no grasps returned.\n");
      return std::vector<std::unique_ptr<candidate::Hand>>();
    }
    printf("Found %d clusters.\n", (int)clusters.size());
    if (clusters.size() <= 3) {
      printf("Not enough clusters found! Adding all grasps from previous step.");
      for (size_t i = 0; i < hands.size(); i++) clusters.push_back(std::move(hands[i]));
    }
  } else {
    clusters = std::move(hands);
  }
  // 7. sort by score
  std::sort(clusters.begin(), clusters.end(), isScoreGreater);
  printf("======== Selected grasps ========\n");
  for (size_t i = 0; i < clusters.size(); i++) std::cout << "Grasp " << i << ": " << clusters[i]->getScore() << "\n";
  printf("Selected the %d best grasps.\n", (int)clusters.size());
  runtimes_[0] = ms[0] / 1e3;
  runtimes_[1] = ms[1] / 1e3;
  runtimes_[2] = ms[2] / 1e3;
  runtimes_[3] = now_s() - t0;
  printf("======== RUNTIMES ========\n 1. Candidate generation: %3.4fs\n 2. Descriptor extraction: %3.4fs\n 3. Classification: %3.4fs\n"
         "==========\n TOTAL: %3.4fs\n",
         runtimes_[0], runtimes_[1], runtimes_[2], runtimes_[3]);
  return clusters;
}


// ---------------------------------------------------------------------------
// SequentialImportanceSampling — sequential_importance_sampling.cpp:11-272
// ---------------------------------------------------------------------------
static const int SUM_OF_GAUSSIANS = 0, MAX_OF_GAUSSIANS = 1;

SequentialImportanceSampling::SequentialImportanceSampling(const std::string &config_filename) {
  util::ConfigFile config_file(config_filename);
  config_file.ExtractKeys();
  num_init_samples_ = config_file.getValueOfKey<int>("num_init_samples", 50);
  num_iterations_ = config_file.getValueOfKey<int>("num_iterations", 5);
  num_samples_ = config_file.getValueOfKey<int>("num_samples_per_iteration", 50);
  prob_rand_samples_ = config_file.getValueOfKey<double>("prob_rand_samples", 0.3);
  radius_ = config_file.getValueOfKey<double>("standard_deviation", 0.02);
  sampling_method_ = config_file.getValueOfKey<int>("sampling_method", SUM_OF_GAUSSIANS);
  min_score_ = config_file.getValueOfKey<double>("min_score", 0);
  workspace_ = config_file.getValueOfKeyAsStdVectorDouble("workspace", "-1 1 -1 1 -1 1");
  workspace_.resize(6, 0.0);
  workspace_grasps_ = config_file.getValueOfKeyAsStdVectorDouble("workspace_grasps", "-1 1 -1 1 -1 1");
  workspace_grasps_.resize(6, 0.0);
  filter_approach_direction_ = config_file.getValueOfKey<bool>("filter_approach_direction", false);
  std::vector<double> approach = config_file.getValueOfKeyAsStdVectorDouble("direction", "1 0 0");
  approach.resize(3, 0.0);
  direction_ = {approach[0], approach[1], approach[2]};
  thresh_rad_ = config_file.getValueOfKey<double>("thresh_rad", 2.3);
  rng_state_ = 0x9E3779B97F4A7C15ull ^ (unsigned long long)config_file.getValueOfKey<int>("random_seed", 0);
  grasp_detector_ = std::make_unique<GraspDetector>(config_filename);
  clustering_ = std::make_unique<Clustering>(config_file.getValueOfKey<int>("min_inliers", 1));
  if (grasp_detector_->context()) clustering_->setContext(grasp_detector_->context());
}

unsigned long long SequentialImportanceSampling::nextRandom() {
  rng_state_ ^= rng_state_ << 13;
  rng_state_ ^= rng_state_ >> 7;
  rng_state_ ^= rng_state_ << 17;
  return rng_state_;
}

double SequentialImportanceSampling::randNormal(double sigma) {  // Box-Muller on two 53-bit uniforms
  const double u1 = ((double)(nextRandom() >> 11) + 1.0) * (1.0 / 9007199254740993.0);
  const double u2 = (double)(nextRandom() >> 11) * (1.0 / 9007199254740992.0);
  return sigma * sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

// :189-201
void SequentialImportanceSampling::drawSamplesFromSumOfGaussians(const std::vector<std::unique_ptr<candidate::HandSet>> &hand_sets,
                                                                 double sigma, int num_gauss_samples, std::vector<double> &samples_out) {
  for (int j = 0; j < num_gauss_samples; j++) {
    const int idx = randInt((int)hand_sets.size());
    const auto s = hand_sets[idx]->getSample();
    for (int r = 0; r < 3; r++) samples_out[3 * (size_t)j + r] = s[r] + randNormal(sigma);
  }
}

// :203-237, rejection sampling
void SequentialImportanceSampling::drawSamplesFromMaxOfGaussians(const std::vector<std::unique_ptr<candidate::HandSet>> &hand_sets,
                                                                 double sigma, int num_gauss_samples, std::vector<double> &samples_out,
                                                                 double term) {
  int j = 0;
  while (j < num_gauss_samples) {
    const int idx = randInt((int)hand_sets.size());
    const auto s = hand_sets[idx]->getSample();
    double x[3];
    for (int r = 0; r < 3; r++) x[r] = s[r] + randNormal(sigma);
    auto density = [&](const std::array<double, 3> &c) {
      double p = 0.0;
      for (int r = 0; r < 3; r++) p += (x[r] - c[r]) * (x[r] - c[r]);
      return term * exp((-1.0 / (2.0 * sigma)) * p);
    };
    double maxp = 0;
    for (size_t k = 0; k < hand_sets.size(); k++) {
      const double p = density(hand_sets[k]->getSample());
      if (p > maxp) maxp = p;
    }
    if (density(s) >= maxp) {
      for (int r = 0; r < 3; r++) samples_out[3 * (size_t)j + r] = x[r];
      j++;
    }
  }
}

// :239-270: uniform over the cloud's sample indices (or samples, or all points), inside the workspace
void SequentialImportanceSampling::drawUniformSamples(const util::Cloud &cloud, int num_samples, int start_idx, std::vector<double> &samples) {
  const std::vector<float> &xyz = cloud.getCloudProcessed();
  int i = 0, guard = 0;
  while (i < num_samples && guard++ < 1000000) {
    double sample[3];
    if (!cloud.getSampleIndices().empty()) {
      const int idx = cloud.getSampleIndices()[randInt((int)cloud.getSampleIndices().size())];
      for (int r = 0; r < 3; r++) sample[r] = (double)xyz[3 * (size_t)idx + r];
    } else if (cloud.getSamples().size() >= 3) {
      const int idx = randInt((int)(cloud.getSamples().size() / 3));
      for (int r = 0; r < 3; r++) sample[r] = cloud.getSamples()[3 * (size_t)idx + r];
    } else {
      const int idx = randInt((int)cloud.size());
      for (int r = 0; r < 3; r++) sample[r] = (double)xyz[3 * (size_t)idx + r];
    }
    if (sample[0] >= workspace_[0] && sample[0] <= workspace_[1] && sample[1] >= workspace_[2] && sample[1] <= workspace_[3] &&
        sample[2] >= workspace_[4] && sample[2] <= workspace_[5]) {
      for (int r = 0; r < 3; r++) samples[3 * (size_t)(start_idx + i) + r] = sample[r];
      i++;
    }
  }
}

// :54-187
std::vector<std::unique_ptr<candidate::Hand>> SequentialImportanceSampling::detectGrasps(util::Cloud &cloud) {
  std::vector<std::unique_ptr<candidate::Hand>> none;
  if (cloud.size() == 0) {
    printf("Error: Point cloud is empty!");
    return none;
  }
  const double t0 = now_s();
  FILE *dump = getenv("GPD_SIS_DUMP") ? fopen(getenv("GPD_SIS_DUMP"), "w") : nullptr;
  // 1. initial grasp hypotheses
  cloud.setSamples({});
  cloud.subsample(num_init_samples_);
  if (dump) {
    fprintf(dump, "INIT %zu\n", cloud.getSampleIndices().size());
    for (int i : cloud.getSampleIndices()) fprintf(dump, "%d\n", i);
  }
  std::vector<std::unique_ptr<candidate::HandSet>> hand_set_list = grasp_detector_->generateGraspCandidates(cloud);
  printf("Initially detected grasp candidates: %zu\n", hand_set_list.size());
  if (hand_set_list.empty()) {
    if (dump) fclose(dump);
    return none;
  }
  hand_set_list = grasp_detector_->filterGraspsWorkspace(hand_set_list, workspace_grasps_);
  printf("Grasps within workspace: %zu", hand_set_list.size());
  if (filter_approach_direction_) hand_set_list = grasp_detector_->filterGraspsDirection(hand_set_list, direction_, thresh_rad_);
  const int num_rand_samples = (int)(prob_rand_samples_ * num_samples_);
  const int num_gauss_samples = num_samples_ - num_rand_samples;
  const double sigma = radius_;
  const double term = 1.0 / sqrt(pow(2.0 * M_PI, 3.0) * pow(sigma, 3.0));
  std::vector<double> samples((size_t)3 * num_samples_, 0.0);
  // 2. importance sampling rounds
  for (int i = 0; i < num_iterations_ && !hand_set_list.empty(); i++) {
    std::cout << i << " " << num_gauss_samples << std::endl;
    if (sampling_method_ == SUM_OF_GAUSSIANS)
      drawSamplesFromSumOfGaussians(hand_set_list, sigma, num_gauss_samples, samples);
    else if (sampling_method_ == MAX_OF_GAUSSIANS)
      drawSamplesFromMaxOfGaussians(hand_set_list, sigma, num_gauss_samples, samples, term);
    drawUniformSamples(cloud, num_rand_samples, num_samples_ - num_rand_samples, samples);
    if (dump) {
      fprintf(dump, "ROUND %d %d\n", i, num_samples_);
      for (int k = 0; k < num_samples_; k++) fprintf(dump, "%.17g %.17g %.17g\n", samples[3 * k], samples[3 * k + 1], samples[3 * k + 2]);
    }
    cloud.setSamples(samples);
    std::vector<std::unique_ptr<candidate::HandSet>> hand_set_list_new = grasp_detector_->generateGraspCandidates(cloud);
    hand_set_list_new = grasp_detector_->filterGraspsWorkspace(hand_set_list_new, workspace_grasps_);
    if (filter_approach_direction_) hand_set_list_new = grasp_detector_->filterGraspsDirection(hand_set_list_new, direction_, thresh_rad_);
    const size_t added = hand_set_list_new.size();
    hand_set_list.insert(hand_set_list.end(), std::make_move_iterator(hand_set_list_new.begin()),
                         std::make_move_iterator(hand_set_list_new.end()));
    printf("Added %zu grasp candidates in round %d. Total: %zu.\n", added, i, hand_set_list.size());
  }
  if (dump) fclose(dump);
  // 3. classify everything at once
  std::vector<std::unique_ptr<candidate::Hand>> valid_grasps = grasp_detector_->pruneGraspCandidates(cloud, hand_set_list, min_score_);
  printf("Valid grasps: %zu\n", valid_grasps.size());
  // 4. cluster
  if (clustering_->getMinInliers() > 0) {
    valid_grasps = clustering_->findClusters(valid_grasps);
    if (clustering_->failed()) printf("ERROR: the clustering step failed; no grasps returned.\n");
  }
  printf("Final result: found %zu grasps.\n", valid_grasps.size());
  printf("Total runtime: %3.4fs\n.\n", now_s() - t0);
  return valid_grasps;
}

// ---------------------------------------------------------------------------
// Clustering::findClusters — clustering.cpp:5-105, on the device (gpd_hip_find_clusters); the per-cluster lines the
// reference prints (:88-91) are printed from the results.
// ---------------------------------------------------------------------------
Clustering::~Clustering() {
  if (own_ctx_) gpd_hip_destroy(own_ctx_);
}

std::vector<std::unique_ptr<candidate::Hand>> Clustering::findClusters(const std::vector<std::unique_ptr<candidate::Hand>> &hand_list,
                                                                        bool remove_inliers) {
  std::vector<std::unique_ptr<candidate::Hand>> hands_out;
  failed_ = false;
  const int n = (int)hand_list.size();
  if (n == 0) return hands_out;
  if (!ctx_) {
    gpd_params p;
    gpd_hip_default_params(&p);
    if (gpd_hip_create(0, &p, &own_ctx_) != GPD_OK) {
      printf("ERROR: Clustering::findClusters: %s\n", gpd_hip_last_error());
      failed_ = true;
      return hands_out;
    }
    ctx_ = own_ctx_;
  }
  std::vector<gpd_hand> in(n), out(n);
  std::vector<double> scores(n), out_scores(n);
  std::vector<int32_t> src(n);
  for (int i = 0; i < n; i++) {
    in[i] = hand_list[i]->record();
    scores[i] = hand_list[i]->getScore();
  }
  int k = 0;
  if (gpd_hip_find_clusters(ctx_, in.data(), scores.data(), n, min_inliers_, remove_inliers ? 1 : 0, out.data(), out_scores.data(), src.data(),
                            &k) != GPD_OK) {
    printf("ERROR: Clustering::findClusters: %s\n", gpd_hip_last_error());
    failed_ = true;
    return hands_out;
  }
  for (int c = 0; c < k; c++) {
    const gpd_hand &seed = in[src[c]];
    const double d[3] = {out[c].position[0] - seed.position[0], out[c].position[1] - seed.position[1], out[c].position[2] - seed.position[2]};
    printf("grasp %d, ||position_delta||: %3.4f, conf_lb: %3.4f\n", (int)src[c], sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), out_scores[c]);
    auto hand = std::make_unique<candidate::Hand>(out[c]);
    hand->setScore(out_scores[c]);
    hands_out.push_back(std::move(hand));
  }
  return hands_out;
}

}  // namespace gpd

// Flat entry for callers without the C++ classes (tests, ctypes): cluster n records with double
// scores; out / out_scores / out_src (the index of the seeding hand) hold up to n results.
extern "C" int gpd_host_find_clusters(const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers,
                                      gpd_hand *out, double *out_scores, int *out_src) {
  std::vector<std::unique_ptr<gpd::candidate::Hand>> list;
  for (int i = 0; i < n; i++) {
    gpd_hand r = hands[i];
    r.slot = i;  // carries the source index through the copy
    list.push_back(std::make_unique<gpd::candidate::Hand>(r));
    list.back()->setScore(scores[i]);
  }
  gpd::Clustering c(min_inliers);
  auto res = c.findClusters(list, remove_inliers != 0);
  for (size_t k = 0; k < res.size(); k++) {
    out[k] = res[k]->record();
    out_src[k] = out[k].slot;
    out[k].slot = hands[out_src[k]].slot;
    out_scores[k] = res[k]->getScore();
  }
  return (int)res.size();
}

// Flat entry for tests / ctypes: loads a PCD like util::Cloud does; returns the number of points
// (at most cap are copied), *has_normals tells whether normal_x/y/z fields were present.
extern "C" int gpd_host_load_pcd(const char *path, float *xyz, float *normals, int cap, int *has_normals) {
  gpd::util::Cloud cloud(path, {0.0, 0.0, 0.0});
  const int n = (int)cloud.size();
  if (has_normals) *has_normals = cloud.hasNormals() ? 1 : 0;
  const int m = n < cap ? n : cap;
  if (xyz && m > 0) std::memcpy(xyz, cloud.getCloudProcessed().data(), (size_t)m * 3 * sizeof(float));
  if (normals && cloud.hasNormals() && m > 0) std::memcpy(normals, cloud.getNormals().data(), (size_t)m * 3 * sizeof(float));
  return n;
}

extern "C" int gpd_host_load_normals_csv(const char *pcd_path, const char *csv_path, float *normals, int cap) {
  gpd::util::Cloud cloud(pcd_path, {0.0, 0.0, 0.0});
  cloud.setNormalsFromFile(csv_path);
  if (!cloud.hasNormals()) return -1;
  const int n = (int)cloud.size(), m = n < cap ? n : cap;
  std::memcpy(normals, cloud.getNormals().data(), (size_t)m * 3 * sizeof(float));
  return n;
}
