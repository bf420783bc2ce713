// oracle/ref_plot_stub.cpp — link-time stand-ins for the members of the reference's util::Plot (util/plot.cpp: 800 lines
// of PCL visualiser / VTK code, not compiled).  TEST INFRASTRUCTURE, part of oracle/_ref/libgpd_ref.so only.  The
// reference's detector calls them only when a plot_* key of its cfg is set; the cfg files the tests write clear them
// all, and every stand-in aborts if it is reached anyway.
#include <cstdio>
#include <cstdlib>

#include <gpd/util/plot.h>

namespace gpd {
namespace util {

[[noreturn]] static void noPlot(const char *what) {
  std::fprintf(stderr, "gpd_ref: util::Plot::%s reached — plotting is not part of the reference build under oracle/_ref\n", what);
  std::abort();
}

void Plot::plotFingers3D(const std::vector<std::unique_ptr<candidate::HandSet>> &, const PointCloudRGBA::Ptr &, std::string,
                         const candidate::HandGeometry &, bool, bool) {
  noPlot("plotFingers3D");
}
void Plot::plotFingers3D(const std::vector<std::unique_ptr<candidate::Hand>> &, const PointCloudRGBA::Ptr &, const std::string &,
                         const candidate::HandGeometry &, bool) {
  noPlot("plotFingers3D");
}
void Plot::plotSamples(const std::vector<int> &, const PointCloudRGBA::Ptr &) { noPlot("plotSamples"); }
void Plot::plotSamples(const Eigen::Matrix3Xd &, const PointCloudRGBA::Ptr &) { noPlot("plotSamples"); }
void Plot::plotNormals(const util::Cloud &, bool) { noPlot("plotNormals"); }
void Plot::plotLocalAxes(const std::vector<candidate::LocalFrame> &, const PointCloudRGBA::Ptr &) { noPlot("plotLocalAxes"); }

}  // namespace util
}  // namespace gpd
