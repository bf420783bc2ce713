// Host-side constants of the candidate search, evaluated in IEEE double with libm
// the way the reference's host code evaluates them.
#include <cmath>
#include <cstring>

#include "gpd_internal.h"

namespace gpd {

// Eigen::AngleAxisd(angle, axis).toRotationMatrix() — Rodrigues form as Eigen's
// Geometry/AngleAxis.h evaluates it (used at hand_set.cpp:52-53 and :68-69).
static void angle_axis(double angle, const double ax[3], double *R) {
  const double s = std::sin(angle), c = std::cos(angle);
  const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  const double c1[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double t = c1[0] * ax[1];
  R[1] = t - sa[2];
  R[3] = t + sa[2];
  t = c1[0] * ax[2];
  R[2] = t + sa[1];
  R[6] = t - sa[1];
  t = c1[1] * ax[2];
  R[5] = t - sa[0];
  R[7] = t + sa[0];
  R[0] = c1[0] * ax[0] + c;
  R[4] = c1[1] * ax[1] + c;
  R[8] = c1[2] * ax[2] + c;
}

void host_consts(const gpd_params &p, HostConsts &h) {
  std::memset(&h, 0, sizeof(h));
  // hand_search.cpp:151-155: LinSpaced(n+1, -pi/2, pi/2).head(n)
  const int no = p.num_orientations;
  const double lo = -1.0 * M_PI / 2.0, hi = M_PI / 2.0;
  const double step = (hi - lo) / (double)no;
  static const double AX[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int a = 0; a < p.num_hand_axes; a++)
    for (int i = 0; i < no; i++) angle_axis(lo + (double)i * step, AX[p.hand_axes[a]], h.rot[a * no + i]);
  const double uy[3] = {0, 1, 0};
  angle_axis(M_PI, uy, h.rot_binormal);
  // finger_hand.cpp:13-18
  const int n = p.num_finger_placements;
  const double top = p.hand_outer_diameter - p.finger_width;
  const double fstep = n > 1 ? (top - 0.0) / (double)(n - 1) : 0.0;
  for (int i = 0; i < n; i++) {
    const double half = (i == n - 1) ? top : 0.0 + (double)i * fstep;
    h.finger_spacing[i] = (half - p.hand_outer_diameter) + p.finger_width;
    h.finger_spacing[n + i] = half;
  }
  // finger_hand.cpp:116-121: for (d = min + 0.005; d <= max; d += 0.005)
  h.num_deepen = 0;
  for (double d = p.init_bite + 0.005; d <= p.hand_depth && h.num_deepen < 128; d += 0.005) h.deepen_depths[h.num_deepen++] = d;
  h.cos_friction = std::cos(p.friction_coeff * M_PI / 180.0);  // antipodal.cpp:30
  // hand_search.cpp:10-17, image_generator.cpp:43-46
  h.nn_radius_hands = std::fmax(std::fmax(p.hand_outer_diameter - p.finger_width, p.hand_depth), p.hand_height / 2.0);
  h.nn_radius_images = std::fmax(std::fmax(p.volume_depth, p.volume_height / 2.0), p.volume_width);
}

// doubles <-> integers in the same order (bisection over representable values)
static long long ord_of(double x) {
  long long b;
  std::memcpy(&b, &x, sizeof(b));
  return b < 0 ? -(b & 0x7fffffffffffffffll) : b;
}
static double of_ord(long long k) {
  long long b = k < 0 ? ((-k) | (long long)0x8000000000000000ull) : k;
  double x;
  std::memcpy(&x, &b, sizeof(x));
  return x;
}

FilterConsts filter_consts(const gpd_params &p) {
  FilterConsts f;
  std::memset(&f, 0, sizeof(f));
  f.min_aperture = p.min_aperture;
  f.max_aperture = p.max_aperture;
  f.half_width = 0.5 * p.hand_outer_diameter;
  f.hand_depth = p.hand_depth;
  for (int i = 0; i < 6; i++) f.workspace[i] = p.workspace_grasps[i];
  f.dir_on = p.filter_approach_direction ? 1 : 0;
  for (int i = 0; i < 3; i++) f.dir[i] = p.direction[i];
  f.dot_drop_max = -2.0;  // nothing dropped
  if (f.dir_on) {
    // largest x in [-1, 1] with acos(x) > thresh_rad, by THIS machine's libm — the one the reference's
    // filterGraspsDirection (grasp_detector.cpp:438) would call
    auto drop = [&](double x) { return std::acos(x) > p.thresh_rad; };
    if (drop(1.0)) {
      f.dot_drop_max = 1.0;
    } else if (drop(-1.0)) {
      long long lo = ord_of(-1.0), hi = ord_of(1.0);  // drop(lo), !drop(hi)
      while (hi - lo > 1) {
        const long long mid = lo + (hi - lo) / 2;
        if (drop(of_ord(mid)))
          lo = mid;
        else
          hi = mid;
      }
      f.dot_drop_max = of_ord(lo);
    }
  }
  return f;
}

// GraspDetector::filterGraspsWorkspace — grasp_detector.cpp:334-398 (workspace_ok in gpd_internal.h).
void filter_workspace_host(const gpd_params &p, gpd_hand *hands, int num_sets) {
  const int slots = p.num_hand_axes * p.num_orientations;
  const FilterConsts f = filter_consts(p);
  for (long i = 0; i < (long)num_sets * slots; i++) {
    gpd_hand &h = hands[i];
    if (h.valid) h.valid = workspace_ok(f, h) ? 1 : 0;
  }
}

}  // namespace gpd
