// oracle/shim/pcl/shim_all.h — the part of the PCL 1.9 interface the reference's hot path names, written for this
// repository.  TEST INFRASTRUCTURE ONLY (the rules of this directory are in oracle/shim/Eigen/Dense).
//
// Semantics restated FROM MEMORY of PCL 1.9 / FLANN 1.9 (none could be checked here):
//   * KdTreeFLANN::radiusSearch(p, r, idx, d2): the query and the points are float xyz; d2 = dx*dx + dy*dy + dz*dz
//     accumulated in float in that order (FLANN L2_Simple); a point is a neighbour iff d2 < (float)(r*r)... see
//     `radius2` below for the exact cast; results sorted by (d2, index); returns the count.  The shim searches by
//     brute force — the tree only changes the cost.  search::KdTree forwards to it (sorted results).
//   * NormalEstimationOMP::compute: per index a radius search, the covariance of the neighbours about their centroid
//     and the eigenvector of its smallest eigenvalue, flipped towards the view point (flipNormalTowardsViewpoint).
//     PCL accumulates in float with a one-pass formula and uses its own closed-form eigen33; the shim uses the
//     definition oracle/gpd_oracle.cpp states for itself (double, two passes, sequential sums in neighbour order,
//     Eigen's iterative 3x3 solver from the Eigen shim) because PCL's numerics are out of reach either way: this pins
//     the reference's logic AROUND the estimator (util/cloud.cpp:497-604), not the estimator.
//   * getMinMax3D, copyPointCloud, removeNaNFromPointCloud, ExtractIndices: exact by nature.
//   * io::loadPCDFile: ascii and binary PCD with float x y z fields (other fields skipped).
// Everything the path never reaches (statistical outlier removal, normal refinement, RANSAC plane fit, integral-image
// normals, random sampling, PLY, the visualiser) exists so that the reference's headers and util/cloud.cpp compile, and
// ABORTS with a message when called.
#ifndef GPD_REF_SHIM_PCL
#define GPD_REF_SHIM_PCL
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <boost/shim_all.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

namespace pcl {

namespace shim {
[[noreturn]] inline void fail(const char *what) {
  std::fprintf(stderr, "PCL shim: %s\n", what);
  std::abort();
}
}  // namespace shim

// point types: 16-byte aligned xyz + payload, 32 bytes as in PCL (getMatrixXfMap() is 8 x N for them)
struct alignas(16) PointXYZ {
  float x = 0, y = 0, z = 0, pad_ = 1.f;
  Eigen::View<float> getVector3fMap() { return Eigen::View<float>(&x, 3, 1, 1, 3); }
  Eigen::View<float> getVector3fMap() const { return Eigen::View<float>(const_cast<float *>(&x), 3, 1, 1, 3); }
};
struct alignas(16) PointXYZRGBA {
  float x = 0, y = 0, z = 0, pad_ = 1.f;
  std::uint32_t rgba = 0;
  float pad2_[3] = {0, 0, 0};
  Eigen::View<float> getVector3fMap() { return Eigen::View<float>(&x, 3, 1, 1, 3); }
  Eigen::View<float> getVector3fMap() const { return Eigen::View<float>(const_cast<float *>(&x), 3, 1, 1, 3); }
};
struct alignas(16) Normal {
  float normal_x = 0, normal_y = 0, normal_z = 0, pad_ = 0;
  float curvature = 0;
  float pad2_[3] = {0, 0, 0};
  Eigen::View<float> getNormalVector3fMap() { return Eigen::View<float>(&normal_x, 3, 1, 1, 3); }
  Eigen::View<float> getNormalVector3fMap() const { return Eigen::View<float>(const_cast<float *>(&normal_x), 3, 1, 1, 3); }
};
struct alignas(16) PointNormal {
  float x = 0, y = 0, z = 0, pad_ = 1.f;
  float normal_x = 0, normal_y = 0, normal_z = 0, pad2_ = 0;
  float curvature = 0;
  float pad3_[3] = {0, 0, 0};
  Eigen::View<float> getVector3fMap() { return Eigen::View<float>(&x, 3, 1, 1, 3); }
  Eigen::View<float> getVector3fMap() const { return Eigen::View<float>(const_cast<float *>(&x), 3, 1, 1, 3); }
  Eigen::View<float> getNormalVector3fMap() { return Eigen::View<float>(&normal_x, 3, 1, 1, 3); }
};
static_assert(sizeof(PointXYZRGBA) == 32 && sizeof(Normal) == 32 && sizeof(PointXYZ) == 16 && sizeof(PointNormal) == 48, "point layouts");

template <class PointT>
class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 1;
  bool is_dense = true;

  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(std::size_t n) {
    points.resize(n);
    width = (std::uint32_t)n;
    height = 1;
  }
  void clear() {
    points.clear();
    width = 0;
    height = 1;
  }
  void push_back(const PointT &p) {
    points.push_back(p);
    width = (std::uint32_t)points.size();
    height = 1;
  }
  PointT &at(std::size_t i) { return points.at(i); }
  const PointT &at(std::size_t i) const { return points.at(i); }
  PointT &operator[](std::size_t i) { return points[i]; }
  const PointT &operator[](std::size_t i) const { return points[i]; }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
  typename std::vector<PointT>::const_iterator end() const { return points.end(); }
  bool isOrganized() const { return height > 1; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
  // (sizeof(PointT) / 4) x N floats, one column per point
  Eigen::View<float> getMatrixXfMap() const {
    const Eigen::Index dim = (Eigen::Index)(sizeof(PointT) / sizeof(float));
    float *p = points.empty() ? nullptr : reinterpret_cast<float *>(const_cast<PointT *>(points.data()));
    return Eigen::View<float>(p, dim, (Eigen::Index)points.size(), 1, dim);
  }
  PointCloud &operator+=(const PointCloud &o) {
    points.insert(points.end(), o.points.begin(), o.points.end());
    width = (std::uint32_t)points.size();
    height = 1;
    return *this;
  }
  PointCloud operator+(const PointCloud &o) const {
    PointCloud r(*this);
    r += o;
    return r;
  }
};

struct PointIndices {
  typedef boost::shared_ptr<PointIndices> Ptr;
  typedef boost::shared_ptr<const PointIndices> ConstPtr;
  std::vector<int> indices;
};
struct ModelCoefficients {
  typedef boost::shared_ptr<ModelCoefficients> Ptr;
  std::vector<float> values;
};
typedef boost::shared_ptr<std::vector<int>> IndicesPtr;
typedef boost::shared_ptr<const std::vector<int>> IndicesConstPtr;
enum SacModel { SACMODEL_PLANE = 0 };
const static int SAC_RANSAC = 0;

namespace shim {
template <class A, class B>
inline void copyXYZ(const A &a, B &b) {
  b.x = a.x;
  b.y = a.y;
  b.z = a.z;
}
}  // namespace shim

template <class PA, class PB>
void copyPointCloud(const PointCloud<PA> &in, PointCloud<PB> &out) {
  out.points.resize(in.points.size());
  out.width = in.width;
  out.height = in.height;
  out.is_dense = in.is_dense;
  for (std::size_t i = 0; i < in.points.size(); i++) shim::copyXYZ(in.points[i], out.points[i]);
}

template <class PointT>
void removeNaNFromPointCloud(const PointCloud<PointT> &cloud, std::vector<int> &index) {
  index.clear();
  for (std::size_t i = 0; i < cloud.points.size(); i++) {
    const PointT &p = cloud.points[i];
    if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) index.push_back((int)i);
  }
}

template <class PointT>
void getMinMax3D(const PointCloud<PointT> &cloud, PointT &min_pt, PointT &max_pt) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const PointT &p : cloud.points) {
    if (!cloud.is_dense && !(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
    lo[0] = std::min(lo[0], p.x), lo[1] = std::min(lo[1], p.y), lo[2] = std::min(lo[2], p.z);
    hi[0] = std::max(hi[0], p.x), hi[1] = std::max(hi[1], p.y), hi[2] = std::max(hi[2], p.z);
  }
  min_pt.x = lo[0], min_pt.y = lo[1], min_pt.z = lo[2];
  max_pt.x = hi[0], max_pt.y = hi[1], max_pt.z = hi[2];
}

// ------------------------------------------------------------------------------------------------------------------
// radius search, FLANN semantics (see the header)
// ------------------------------------------------------------------------------------------------------------------
template <class PointT>
class KdTreeFLANN {
 public:
  typedef boost::shared_ptr<const PointCloud<PointT>> PointCloudConstPtr;
  typedef boost::shared_ptr<KdTreeFLANN<PointT>> Ptr;
  explicit KdTreeFLANN(bool sorted = true) : sorted_(sorted) {}
  void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr & = IndicesConstPtr()) { cloud_ = cloud; }
  PointCloudConstPtr getInputCloud() const { return cloud_; }
  int radiusSearch(const PointT &q, double radius, std::vector<int> &k_indices, std::vector<float> &k_sqr_distances,
                   unsigned int max_nn = 0) const {
    // pcl::KdTreeFLANN::radiusSearch hands FLANN `static_cast<float>(radius * radius)`
    const float radius2 = static_cast<float>(radius * radius);
    std::vector<std::pair<float, int>> hits;
    const std::vector<PointT> &pts = cloud_->points;
    for (std::size_t i = 0; i < pts.size(); i++) {
      const float dx = q.x - pts[i].x, dy = q.y - pts[i].y, dz = q.z - pts[i].z;
      float d2 = dx * dx;
      d2 = d2 + dy * dy;
      d2 = d2 + dz * dz;
      if (d2 < radius2) hits.push_back(std::make_pair(d2, (int)i));
    }
#if defined(GPD_SHIM_PERTURB) && GPD_SHIM_PERTURB == 1
    // sensitivity study (profiles/thirdparty_sensitivity.py): neighbours at EQUAL distance in the opposite order — what FLANN
    // returns for ties depends on its tree walk, which this subset does not restate
    std::sort(hits.begin(), hits.end(), [](const std::pair<float, int> &x, const std::pair<float, int> &y) {
      return x.first < y.first || (x.first == y.first && x.second > y.second);
    });
#else
    std::sort(hits.begin(), hits.end());
#endif
    if (max_nn > 0 && hits.size() > max_nn) hits.resize(max_nn);
    k_indices.resize(hits.size());
    k_sqr_distances.resize(hits.size());
    for (std::size_t i = 0; i < hits.size(); i++) {
      k_indices[i] = hits[i].second;
      k_sqr_distances[i] = hits[i].first;
    }
    return (int)hits.size();
  }
  int nearestKSearch(const PointT &, int, std::vector<int> &, std::vector<float> &) const { shim::fail("nearestKSearch is not on the path"); }

 private:
  PointCloudConstPtr cloud_;
  bool sorted_;
};
template <class PointT>
using KdTree = KdTreeFLANN<PointT>;

namespace search {
template <class PointT>
class KdTree {
 public:
  typedef boost::shared_ptr<KdTree<PointT>> Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT>> PointCloudConstPtr;
  explicit KdTree(bool sorted = true) : tree_(sorted) {}
  void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr & = IndicesConstPtr()) { tree_.setInputCloud(cloud); }
  int radiusSearch(const PointT &q, double radius, std::vector<int> &idx, std::vector<float> &d2, unsigned int max_nn = 0) const {
    return tree_.radiusSearch(q, radius, idx, d2, max_nn);
  }
  void nearestKSearch(const PointCloud<PointT> &, const std::vector<int> &, int, std::vector<std::vector<int>> &,
                      std::vector<std::vector<float>> &) const {
    shim::fail("search::KdTree::nearestKSearch (refineNormals) is not on the path");
  }

 private:
  KdTreeFLANN<PointT> tree_;
};
}  // namespace search

// ------------------------------------------------------------------------------------------------------------------
// filters
// ------------------------------------------------------------------------------------------------------------------
template <class PointT>
class ExtractIndices {
 public:
  explicit ExtractIndices(bool = false) {}
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &c) { cloud_ = c; }
  void setIndices(const boost::shared_ptr<PointIndices> &i) { idx_ = i->indices; }
  void setIndices(const IndicesConstPtr &i) { idx_ = *i; }
  void setNegative(bool n) { negative_ = n; }
  void filter(std::vector<int> &out) const {
    out.clear();
    if (!negative_) {
      out = idx_;
      return;
    }
    std::vector<char> in(cloud_->size(), 0);
    for (int i : idx_) in[(std::size_t)i] = 1;
    for (std::size_t i = 0; i < in.size(); i++)
      if (!in[i]) out.push_back((int)i);
  }
  void filter(PointCloud<PointT> &out) const {
    std::vector<int> keep;
    filter(keep);
    PointCloud<PointT> r;
    r.points.reserve(keep.size());
    for (int i : keep) r.points.push_back(cloud_->points[(std::size_t)i]);
    r.width = (std::uint32_t)r.points.size();
    r.height = 1;
    r.is_dense = true;
    out = r;
  }

 private:
  boost::shared_ptr<const PointCloud<PointT>> cloud_;
  std::vector<int> idx_;
  bool negative_ = false;
};

template <class PointT>
class StatisticalOutlierRemoval {
 public:
  explicit StatisticalOutlierRemoval(bool = false) {}
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &) {}
  void setMeanK(int) {}
  void setStddevMulThresh(double) {}
  void filter(PointCloud<PointT> &) { shim::fail("StatisticalOutlierRemoval is not on the path"); }
};

template <class PointT>
class RandomSample {
 public:
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &) {}
  void setSample(unsigned int) {}
  void setSeed(unsigned int) {}
  void filter(std::vector<int> &) { shim::fail("RandomSample (time-seeded in the reference) is not on the path: pass sample indices"); }
};

template <class NormalT>
class NormalRefinement {
 public:
  NormalRefinement(const std::vector<std::vector<int>> &, const std::vector<std::vector<float>> &) {}
  void setInputCloud(const boost::shared_ptr<const PointCloud<NormalT>> &) {}
  void filter(PointCloud<NormalT> &) { shim::fail("NormalRefinement is not on the path"); }
};

template <class PointT>
class SACSegmentation {
 public:
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &) {}
  void setOptimizeCoefficients(bool) {}
  void setModelType(int) {}
  void setMethodType(int) {}
  void setDistanceThreshold(double) {}
  void segment(PointIndices &, ModelCoefficients &) { shim::fail("SACSegmentation (plane removal) is not on the path"); }
};

// ------------------------------------------------------------------------------------------------------------------
// normals
// ------------------------------------------------------------------------------------------------------------------
template <class PointT, class NormalT>
class NormalEstimationOMP {
 public:
  explicit NormalEstimationOMP(unsigned int = 0) {}
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &c) { cloud_ = c; }
  void setSearchMethod(const typename search::KdTree<PointT>::Ptr &t) { tree_ = t; }
  void setRadiusSearch(double r) { radius_ = r; }
  void setIndices(const IndicesPtr &i) { indices_ = i; }
  void setViewPoint(float x, float y, float z) { vp_[0] = x, vp_[1] = y, vp_[2] = z; }
  void compute(PointCloud<NormalT> &out) {
    KdTreeFLANN<PointT> tree;
    tree.setInputCloud(cloud_);
    const std::vector<int> &idx = *indices_;
    out.points.assign(idx.size(), NormalT());
    out.width = (std::uint32_t)idx.size();
    out.height = 1;
    std::vector<int> nn;
    std::vector<float> d2;
    for (std::size_t q = 0; q < idx.size(); q++) {
      const PointT &p = cloud_->points[(std::size_t)idx[q]];
      NormalT &n = out.points[q];
      if (tree.radiusSearch(p, radius_, nn, d2) == 0) {
        n.normal_x = n.normal_y = n.normal_z = n.curvature = std::numeric_limits<float>::quiet_NaN();
        continue;
      }
      const int k = (int)nn.size();
      double c[3] = {0, 0, 0};
      for (int i = 0; i < k; i++) {
        const PointT &a = cloud_->points[(std::size_t)nn[i]];
        c[0] += (double)a.x, c[1] += (double)a.y, c[2] += (double)a.z;
      }
      for (int r = 0; r < 3; r++) c[r] /= (double)k;
      Eigen::Matrix3d M = Eigen::Matrix3d::Zero();
      for (int i = 0; i < k; i++) {
        const PointT &a = cloud_->points[(std::size_t)nn[i]];
        const double d0 = (double)a.x - c[0], d1 = (double)a.y - c[1], dd2 = (double)a.z - c[2];
        M(0, 0) += d0 * d0, M(1, 0) += d1 * d0, M(1, 1) += d1 * d1, M(2, 0) += dd2 * d0, M(2, 1) += dd2 * d1, M(2, 2) += dd2 * dd2;
      }
      M(0, 1) = M(1, 0), M(0, 2) = M(2, 0), M(1, 2) = M(2, 1);
      Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> es(M);
      int mn = 0;
      es.eigenvalues().minCoeff(&mn);
      double nx = es.eigenvectors()(0, mn), ny = es.eigenvectors()(1, mn), nz = es.eigenvectors()(2, mn);
      // flipNormalTowardsViewpoint
      const double dot = ((double)vp_[0] - (double)p.x) * nx + ((double)vp_[1] - (double)p.y) * ny + ((double)vp_[2] - (double)p.z) * nz;
      if (dot < 0) nx = -nx, ny = -ny, nz = -nz;
      n.normal_x = (float)nx, n.normal_y = (float)ny, n.normal_z = (float)nz;
    }
  }

 private:
  boost::shared_ptr<const PointCloud<PointT>> cloud_;
  typename search::KdTree<PointT>::Ptr tree_;
  IndicesPtr indices_;
  double radius_ = 0;
  float vp_[3] = {0, 0, 0};
};

template <class PointT, class NormalT>
class IntegralImageNormalEstimation {
 public:
  enum NormalEstimationMethod { COVARIANCE_MATRIX, AVERAGE_3D_GRADIENT, AVERAGE_DEPTH_CHANGE, SIMPLE_3D_GRADIENT };
  void setInputCloud(const boost::shared_ptr<const PointCloud<PointT>> &) {}
  void setViewPoint(float, float, float) {}
  void setNormalEstimationMethod(NormalEstimationMethod) {}
  void setNormalSmoothingSize(float) {}
  void compute(PointCloud<NormalT> &) { shim::fail("IntegralImageNormalEstimation (organized clouds) is not on the path"); }
};

// ------------------------------------------------------------------------------------------------------------------
// io
// ------------------------------------------------------------------------------------------------------------------
namespace io {
template <class PointT>
int loadPCDFile(const std::string &path, PointCloud<PointT> &cloud) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return -1;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  long points = -1, width = 0, height = 1;
  std::string data, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key, tok;
    ls >> key;
    if (key == "FIELDS") while (ls >> tok) fields.push_back(tok);
    else if (key == "SIZE") while (ls >> tok) sizes.push_back(std::atoi(tok.c_str()));
    else if (key == "TYPE") while (ls >> tok) types.push_back(tok);
    else if (key == "COUNT") while (ls >> tok) counts.push_back(std::atoi(tok.c_str()));
    else if (key == "WIDTH") ls >> width;
    else if (key == "HEIGHT") ls >> height;
    else if (key == "POINTS") ls >> points;
    else if (key == "DATA") {
      ls >> data;
      break;
    }
  }
  if (points < 0) points = width * height;
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() || counts.size() != fields.size()) return -1;
  int ix = -1, iy = -1, iz = -1;
  std::vector<int> offs(fields.size());
  int stride = 0;
  for (std::size_t i = 0; i < fields.size(); i++) {
    offs[i] = stride;
    stride += sizes[i] * counts[i];
    if (fields[i] == "x") ix = (int)i;
    if (fields[i] == "y") iy = (int)i;
    if (fields[i] == "z") iz = (int)i;
  }
  if (ix < 0 || iy < 0 || iz < 0 || types[ix] != "F" || sizes[ix] != 4) return -1;
  cloud.points.assign((std::size_t)points, PointT());
  cloud.width = (std::uint32_t)width;
  cloud.height = (std::uint32_t)height;
  cloud.is_dense = true;
  if (data == "ascii") {
    for (long p = 0; p < points; p++) {
      if (!std::getline(f, line)) return -1;
      std::istringstream ls(line);
      std::string tok;
      for (std::size_t i = 0; i < fields.size(); i++)
        for (int c = 0; c < counts[i]; c++) {
          if (!(ls >> tok)) return -1;
          const float v = (float)std::strtod(tok.c_str(), nullptr);  // "nan" parses to NaN
          if ((int)i == ix) cloud.points[(std::size_t)p].x = v;
          if ((int)i == iy) cloud.points[(std::size_t)p].y = v;
          if ((int)i == iz) cloud.points[(std::size_t)p].z = v;
        }
    }
  } else if (data == "binary") {
    std::vector<char> rec((std::size_t)stride);
    for (long p = 0; p < points; p++) {
      if (!f.read(rec.data(), stride)) return -1;
      std::memcpy(&cloud.points[(std::size_t)p].x, rec.data() + offs[ix], 4);
      std::memcpy(&cloud.points[(std::size_t)p].y, rec.data() + offs[iy], 4);
      std::memcpy(&cloud.points[(std::size_t)p].z, rec.data() + offs[iz], 4);
    }
  } else {
    return -1;
  }
  for (const PointT &p : cloud.points)
    if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) cloud.is_dense = false;
  return 0;
}
template <class PointT>
int loadPLYFile(const std::string &, PointCloud<PointT> &) {
  shim::fail("loadPLYFile is not on the path");
}
}  // namespace io

namespace visualization {
class PCLVisualizer {};
class KeyboardEvent {};
class CloudViewer {};
}  // namespace visualization

}  // namespace pcl
#endif
