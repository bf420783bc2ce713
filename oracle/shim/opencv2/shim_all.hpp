// oracle/shim/opencv2/shim_all.hpp — the part of the OpenCV interface the reference's hot path uses (descriptor/*.cpp,
// net/eigen_classifier.cpp, grasp_detector.cpp), written for this repository.  TEST INFRASTRUCTURE ONLY (the rules of
// this directory are in oracle/shim/Eigen/Dense).
//
// Every numeric routine is restated FROM MEMORY of OpenCV 3.4 / 4.x's documented scalar behaviour; none could be
// checked against OpenCV here, and SIMD builds of OpenCV may fuse `src*scale + shift`:
//   * Vec3f arithmetic (matx.hpp): a - b and a += b in float; Vec * double computes each a[i]*alpha in double and
//     rounds to float (saturate_cast<float>);
//   * dilate with a 3x3 MORPH_RECT element, default anchor / iterations / border: per channel maximum over the 3x3
//     window, pixels outside the image ignored;
//   * minMaxLoc(src, &min, &max, 0, 0, mask): extrema over the pixels whose mask is non-zero, both 0 when there is none;
//   * normalize(src, dst, a, b, NORM_MINMAX, dtype): smin / smax over ALL channels as doubles,
//     scale = (max(a,b) - min(a,b)) * (smax - smin > DBL_EPSILON ? 1/(smax - smin) : 0), shift = min(a,b) - smin*scale,
//     then convertTo(dst, dtype, scale, shift);
//   * convertTo 32F -> 32F: dst = src*(float)alpha + (float)beta in float, unfused;
//     convertTo 32F -> 8U: dst = saturate_cast<uchar>(src*(float)alpha + (float)beta), i.e. round half to even
//     (cvRound = lrint), clamped to 0..255;
//   * Mat(rows, cols, type, Scalar s): every pixel's channel c set to saturate_cast(s[c]) (c < 4; an all-zero scalar
//     clears any number of channels); Mat::setTo(s, mask): the same on the pixels whose mask is non-zero;
//   * a - b on CV_32F matrices: per-element float subtraction; merge / split: HWC interleave / de-interleave.
// Windows (namedWindow / imshow / waitKey) and cvtColor exist so that the strategies' showImage members link; they
// abort when called.
#ifndef GPD_REF_SHIM_OPENCV
#define GPD_REF_SHIM_OPENCV
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH(flags) ((flags) & 7)
#define CV_MAT_CN(flags) ((((flags) >> CV_CN_SHIFT) & 511) + 1)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC(n) CV_MAKETYPE(CV_8U, (n))
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC(n) CV_MAKETYPE(CV_32F, (n))
#define CV_VERSION_MAJOR 3
#define CV_VERSION_MINOR 4

typedef unsigned char uchar;  // OpenCV's interface.h declares it at global scope; the reference relies on that
typedef unsigned short ushort;

namespace cv {

using ::uchar;

namespace shim {
[[noreturn]] inline void fail(const char *what) {
  std::fprintf(stderr, "OpenCV shim: %s\n", what);
  std::abort();
}
inline void check(bool ok, const char *what) {
  if (!ok) fail(what);
}
inline int depthSize(int depth) {
  switch (depth) {
    case CV_8U: return 1;
    case CV_32F: return 4;
    case CV_64F: return 8;
    default: fail("unsupported depth");
  }
}
inline uchar saturate_u8(float v) {  // saturate_cast<uchar>(float): cvRound (round half to even) then clamp
#if defined(GPD_SHIM_PERTURB) && GPD_SHIM_PERTURB == 4
  long r = (long)std::floor((double)v + 0.5);  // sensitivity study: cvRound as some builds without SSE2 / lrint compute it
#else
  long r = std::lrintf(v);
#endif
  return (uchar)std::min(255L, std::max(0L, r));
}
}  // namespace shim

struct Scalar {
  double val[4];
  Scalar(double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0) {
    val[0] = v0;
    val[1] = v1;
    val[2] = v2;
    val[3] = v3;
  }
  double operator[](int i) const { return val[i]; }
};
struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};
struct Point {
  int x, y;
  Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {}
};
struct Rect {
  int x, y, width, height;
  Rect(int x_ = 0, int y_ = 0, int w = 0, int h = 0) : x(x_), y(y_), width(w), height(h) {}
};

template <class T, int N>
struct Vec {
  T val[N];
  Vec() {
    for (int i = 0; i < N; i++) val[i] = T(0);
  }
  Vec(T a, T b, T c) {
    static_assert(N == 3, "3-element constructor");
    val[0] = a;
    val[1] = b;
    val[2] = c;
  }
  T &operator()(int i) { return val[i]; }
  const T &operator()(int i) const { return val[i]; }
  T &operator[](int i) { return val[i]; }
  const T &operator[](int i) const { return val[i]; }
};
template <class T, int N>
inline Vec<T, N> operator-(const Vec<T, N> &a, const Vec<T, N> &b) {
  Vec<T, N> r;
  for (int i = 0; i < N; i++) r.val[i] = (T)(a.val[i] - b.val[i]);
  return r;
}
template <class T, int N>
inline Vec<T, N> operator+(const Vec<T, N> &a, const Vec<T, N> &b) {
  Vec<T, N> r;
  for (int i = 0; i < N; i++) r.val[i] = (T)(a.val[i] + b.val[i]);
  return r;
}
template <class T, int N>
inline Vec<T, N> &operator+=(Vec<T, N> &a, const Vec<T, N> &b) {
  for (int i = 0; i < N; i++) a.val[i] = (T)(a.val[i] + b.val[i]);
  return a;
}
template <class T, int N>
inline Vec<T, N> operator*(const Vec<T, N> &a, double alpha) {  // each product in double, then rounded to T
  Vec<T, N> r;
  for (int i = 0; i < N; i++) r.val[i] = (T)(a.val[i] * alpha);
  return r;
}
template <class T, int N>
inline Vec<T, N> operator*(const Vec<T, N> &a, float alpha) {
  Vec<T, N> r;
  for (int i = 0; i < N; i++) r.val[i] = (T)(a.val[i] * alpha);
  return r;
}
typedef Vec<float, 3> Vec3f;
typedef Vec<uchar, 3> Vec3b;

class Mat {
 public:
  int flags = 0, rows = 0, cols = 0;
  uchar *data = nullptr;
  size_t step = 0;  // bytes per row

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar &s) {
    create(r, c, type);
    setTo(s);
  }
  Mat(Size sz, int type) { create(sz.height, sz.width, type); }
  Mat(Size sz, int type, const Scalar &s) {
    create(sz.height, sz.width, type);
    setTo(s);
  }
  void create(int r, int c, int type) {
    flags = type;
    rows = r;
    cols = c;
    step = (size_t)c * elemSize();
    buf_ = std::make_shared<std::vector<uchar>>((size_t)r * step, (uchar)0);
    data = buf_->data();
  }
  int type() const { return flags; }
  int depth() const { return CV_MAT_DEPTH(flags); }
  int channels() const { return CV_MAT_CN(flags); }
  size_t elemSize() const { return (size_t)shim::depthSize(depth()) * channels(); }
  size_t elemSize1() const { return (size_t)shim::depthSize(depth()); }
  size_t total() const { return (size_t)rows * cols; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step == (size_t)cols * elemSize(); }
  Size size() const { return Size(cols, rows); }
  uchar *ptr(int r = 0) { return data + (size_t)r * step; }
  const uchar *ptr(int r = 0) const { return data + (size_t)r * step; }
  template <class T>
  T *ptr(int r = 0) {
    return reinterpret_cast<T *>(data + (size_t)r * step);
  }
  template <class T>
  const T *ptr(int r = 0) const {
    return reinterpret_cast<const T *>(data + (size_t)r * step);
  }
  template <class T>
  T &at(int r, int c) {
    shim::check(r >= 0 && r < rows && c >= 0 && c < cols && sizeof(T) == elemSize(), "Mat::at: index or element type out of range");
    return reinterpret_cast<T *>(data + (size_t)r * step)[c];
  }
  template <class T>
  const T &at(int r, int c) const {
    return const_cast<Mat *>(this)->at<T>(r, c);
  }
  Mat clone() const {
    Mat m;
    copyTo(m);
    return m;
  }
  void copyTo(Mat &dst) const {
    if (dst.rows != rows || dst.cols != cols || dst.flags != flags || !dst.data) dst.create(rows, cols, flags);
    for (int r = 0; r < rows; r++) std::memmove(dst.ptr(r), ptr(r), (size_t)cols * elemSize());
  }
  void copyTo(Mat &&dst) const {  // copyTo(image_out(Rect(...)))
    shim::check(dst.rows == rows && dst.cols == cols && dst.flags == flags, "copyTo into a window of another shape");
    for (int r = 0; r < rows; r++) std::memmove(dst.ptr(r), ptr(r), (size_t)cols * elemSize());
  }
  Mat operator()(const Rect &roi) const {
    shim::check(roi.x >= 0 && roi.y >= 0 && roi.x + roi.width <= cols && roi.y + roi.height <= rows, "window out of range");
    Mat m;
    m.flags = flags;
    m.rows = roi.height;
    m.cols = roi.width;
    m.step = step;
    m.buf_ = buf_;
    m.data = data + (size_t)roi.y * step + (size_t)roi.x * elemSize();
    return m;
  }
  Mat &setTo(const Scalar &s, const Mat &mask = Mat()) {
    const int cn = channels();
    const bool zero = s.val[0] == 0 && s.val[1] == 0 && s.val[2] == 0 && s.val[3] == 0;
    shim::check(zero || cn <= 4, "setTo: a non-zero scalar on more than four channels");
    shim::check(mask.empty() || (mask.rows == rows && mask.cols == cols && mask.type() == CV_8UC1), "setTo: mask shape");
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) {
        if (!mask.empty() && mask.at<uchar>(r, c) == 0) continue;
        for (int k = 0; k < cn; k++) {
          const double v = zero ? 0.0 : s.val[k];
          if (depth() == CV_8U)
            ptr(r)[(size_t)c * cn + k] = (uchar)std::min(255L, std::max(0L, std::lrint(v)));
          else if (depth() == CV_32F)
            ptr<float>(r)[(size_t)c * cn + k] = (float)v;
          else
            shim::fail("setTo: unsupported depth");
        }
      }
    return *this;
  }
  Mat &operator=(const Scalar &s) { return setTo(s); }
  // dst = saturate_cast<DT>(src * alpha + beta), see the header
  void convertTo(Mat &dst, int rtype, double alpha = 1.0, double beta = 0.0) const {
    const int ddepth = rtype < 0 ? depth() : CV_MAT_DEPTH(rtype);
    const int cn = channels();
    Mat out(rows, cols, CV_MAKETYPE(ddepth, cn));
    const float fa = (float)alpha, fb = (float)beta;
    for (int r = 0; r < rows; r++)
      for (int k = 0; k < cols * cn; k++) {
        if (depth() == CV_32F && ddepth == CV_32F) {
          out.ptr<float>(r)[k] = ptr<float>(r)[k] * fa + fb;
        } else if (depth() == CV_32F && ddepth == CV_8U) {
          out.ptr(r)[k] = shim::saturate_u8(ptr<float>(r)[k] * fa + fb);
        } else if (depth() == CV_8U && ddepth == CV_32F) {
          out.ptr<float>(r)[k] = (float)ptr(r)[k] * fa + fb;
        } else if (depth() == CV_8U && ddepth == CV_8U) {
          out.ptr(r)[k] = shim::saturate_u8((float)ptr(r)[k] * fa + fb);
        } else {
          shim::fail("convertTo: unsupported depth pair");
        }
      }
    dst = out;
  }

 private:
  std::shared_ptr<std::vector<uchar>> buf_;
};

inline Mat operator-(const Mat &a, const Mat &b) {
  shim::check(a.rows == b.rows && a.cols == b.cols && a.type() == b.type() && a.depth() == CV_32F, "Mat - Mat: CV_32F of one shape only");
  Mat out(a.rows, a.cols, a.type());
  const int n = a.cols * a.channels();
  for (int r = 0; r < a.rows; r++)
    for (int k = 0; k < n; k++) out.ptr<float>(r)[k] = a.ptr<float>(r)[k] - b.ptr<float>(r)[k];
  return out;
}

enum { MORPH_RECT = 0, MORPH_CROSS = 1, MORPH_ELLIPSE = 2 };
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_MINMAX = 32 };
enum { COLOR_RGB2BGR = 4, COLOR_GRAY2RGB = 8, COLOR_GRAY2BGR = 8 };
enum { WINDOW_NORMAL = 0, WINDOW_AUTOSIZE = 1 };

inline Mat getStructuringElement(int shape, Size ksize) {
  shim::check(shape == MORPH_RECT, "getStructuringElement: MORPH_RECT only");
  return Mat(ksize.height, ksize.width, CV_8UC1, Scalar(1));
}

inline void dilate(const Mat &src, Mat &dst, const Mat &kernel) {
  shim::check(kernel.rows == 3 && kernel.cols == 3, "dilate: 3x3 element only");
  const int cn = src.channels();
  Mat out(src.rows, src.cols, src.type());
  for (int r = 0; r < src.rows; r++)
    for (int c = 0; c < src.cols; c++)
      for (int k = 0; k < cn; k++) {
        bool have = false;
        float bestf = 0.f;
        uchar bestu = 0;
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            const int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= src.rows || cc < 0 || cc >= src.cols) continue;
            if (src.depth() == CV_32F) {
              const float v = src.ptr<float>(rr)[(size_t)cc * cn + k];
              if (!have || v > bestf) bestf = v;
            } else if (src.depth() == CV_8U) {
              const uchar v = src.ptr(rr)[(size_t)cc * cn + k];
              if (!have || v > bestu) bestu = v;
            } else {
              shim::fail("dilate: unsupported depth");
            }
            have = true;
          }
        if (src.depth() == CV_32F)
          out.ptr<float>(r)[(size_t)c * cn + k] = bestf;
        else
          out.ptr(r)[(size_t)c * cn + k] = bestu;
      }
  dst = out;
}

inline void minMaxLoc(const Mat &src, double *minVal, double *maxVal = nullptr, Point *minLoc = nullptr, Point *maxLoc = nullptr,
                      const Mat &mask = Mat()) {
  shim::check(src.channels() == 1, "minMaxLoc: single channel only");
  bool have = false;
  double mn = 0, mx = 0;
  Point pmn(-1, -1), pmx(-1, -1);
  for (int r = 0; r < src.rows; r++)
    for (int c = 0; c < src.cols; c++) {
      if (!mask.empty() && mask.at<uchar>(r, c) == 0) continue;
      const double v = src.depth() == CV_32F ? (double)src.at<float>(r, c) : (double)src.at<uchar>(r, c);
      if (!have || v < mn) {
        mn = v;
        pmn = Point(c, r);
      }
      if (!have || v > mx) {
        mx = v;
        pmx = Point(c, r);
      }
      have = true;
    }
  if (!have) mn = mx = 0;
  if (minVal) *minVal = mn;
  if (maxVal) *maxVal = mx;
  if (minLoc) *minLoc = pmn;
  if (maxLoc) *maxLoc = pmx;
}

inline void normalize(const Mat &src, Mat &dst, double a = 1, double b = 0, int norm_type = NORM_L2, int dtype = -1) {
  shim::check(norm_type == NORM_MINMAX, "normalize: NORM_MINMAX only");
  const int cn = src.channels();
  double smin = DBL_MAX, smax = -DBL_MAX;
  for (int r = 0; r < src.rows; r++)
    for (int k = 0; k < src.cols * cn; k++) {
      const double v = src.depth() == CV_32F ? (double)src.ptr<float>(r)[k] : (double)src.ptr(r)[k];
      smin = std::min(smin, v);
      smax = std::max(smax, v);
    }
  if (src.total() == 0) smin = smax = 0;
  const double dmin = std::min(a, b), dmax = std::max(a, b);
#if defined(GPD_SHIM_PERTURB) && GPD_SHIM_PERTURB == 5
  const double scale = smax - smin > DBL_EPSILON ? (dmax - dmin) / (smax - smin) : 0;  // sensitivity study: one division instead of reciprocal * range
#else
  const double scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
#endif
  const double shift = dmin - smin * scale;
  src.convertTo(dst, dtype < 0 ? src.depth() : CV_MAT_DEPTH(dtype), scale, shift);
}

inline void merge(const std::vector<Mat> &mv, Mat &dst) {
  shim::check(!mv.empty(), "merge: nothing to merge");
  int cn = 0;
  for (const Mat &m : mv) {
    shim::check(m.rows == mv[0].rows && m.cols == mv[0].cols && m.depth() == mv[0].depth(), "merge: planes differ");
    cn += m.channels();
  }
  Mat out(mv[0].rows, mv[0].cols, CV_MAKETYPE(mv[0].depth(), cn));
  const size_t es = mv[0].elemSize1();
  int k0 = 0;
  for (const Mat &m : mv) {
    const int mc = m.channels();
    for (int r = 0; r < out.rows; r++)
      for (int c = 0; c < out.cols; c++)
        std::memcpy(out.ptr(r) + ((size_t)c * cn + k0) * es, m.ptr(r) + (size_t)c * mc * es, (size_t)mc * es);
    k0 += mc;
  }
  dst = out;
}

inline void split(const Mat &src, std::vector<Mat> &mv) {
  const int cn = src.channels();
  const size_t es = src.elemSize1();
  mv.assign(cn, Mat());
  for (int k = 0; k < cn; k++) {
    mv[k] = Mat(src.rows, src.cols, CV_MAKETYPE(src.depth(), 1));
    for (int r = 0; r < src.rows; r++)
      for (int c = 0; c < src.cols; c++) std::memcpy(mv[k].ptr(r) + (size_t)c * es, src.ptr(r) + ((size_t)c * cn + k) * es, es);
  }
}

inline void cvtColor(const Mat &, Mat &, int) { shim::fail("cvtColor is only reachable from the plotting code"); }
inline void namedWindow(const std::string &, int = 0) { shim::fail("namedWindow: no windows in the shim"); }
inline void imshow(const std::string &, const Mat &) { shim::fail("imshow: no windows in the shim"); }
inline int waitKey(int = 0) { shim::fail("waitKey: no windows in the shim"); }

}  // namespace cv
#endif
