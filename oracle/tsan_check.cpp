// Race check of the OpenMP oracle (TEST INFRASTRUCTURE, not on the product path): every OpenMP loop of
// gpd_oracle.cpp restates a loop the reference runs in parallel (frame_estimator.cpp:13-15, hand_search.cpp:168-182,
// image_generator.cpp:83-89, eigen_classifier.cpp:64-66, cloud.cpp:497-535), two of which are racy THERE (SURVEY §9-Q9:
// the classifier's member buffers, HandSet::seed_).  The oracle's single-thread semantics are the definition, so its own
// loops must be race free: this driver runs them all on a small synthetic scene under ThreadSanitizer with the
// OpenMP-aware Archer tool, and then checks that 1 thread and 8 threads give the same bytes.
//   oracle/tsan.sh   (clang++ -fsanitize=thread -fopenmp of this file + gpd_oracle.cpp; report -> profiles/)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/gpd_hip.h"

extern "C" {
void gpd_oracle_default_params(gpd_params *p);
void gpd_oracle_set_num_threads(int n);
int gpd_oracle_num_threads();
void gpd_oracle_normals(const float *xyz, int P, const int32_t *cam_source, int n_cams, const double *view_points, double radius,
                        float *normals_out);
int gpd_oracle_detect(const gpd_params *P, const float *xyz, const float *normals, int np, const int32_t *cam_source, int n_cams,
                      const double *view_points, const int32_t *sample_idx, int S, const float *const *weights, gpd_hand *hands, int *num_sets,
                      int *num_cand, uint8_t *images_out, int max_cand, double *times);
void gpd_oracle_reevaluate(const gpd_params *P, const float *xyz, const float *normals, int np, gpd_hand *hands, int n, int32_t *labels);
}

namespace {
uint64_t g_state = 88172645463325252ull;
double uniform() {  // xorshift64, fixed seed: the scene is the same in every run
  g_state ^= g_state << 13;
  g_state ^= g_state >> 7;
  g_state ^= g_state << 17;
  return (double)(g_state >> 11) / 9007199254740992.0;
}

struct Run {
  std::vector<float> normals;
  std::vector<gpd_hand> hands;
  std::vector<uint8_t> images;
  std::vector<int32_t> labels;
  int num_sets = 0, num_cand = 0;
};
}  // namespace

int main(int argc, char **argv) {
  if (argc > 1 && !std::strcmp(argv[1], "--positive-control")) {
    // the detector must fire on a real race: an unsynchronised counter under the same OpenMP runtime
    long hits = 0;
#pragma omp parallel for
    for (int i = 0; i < 100000; i++) hits += i & 1;
    std::printf("positive control: racy counter = %ld (ThreadSanitizer must have reported it)\n", hits);
    return 0;
  }
  // a box (8 x 5 x 6 cm) standing on a table patch, on a 3 mm lattice, seen from above
  std::vector<float> xyz;
  const double step = 0.003;
  for (int i = -40; i <= 40; i++)
    for (int j = -40; j <= 40; j++) {
      const double x = i * step, y = j * step;
      const bool under = std::fabs(x) <= 0.04 && std::fabs(y) <= 0.025;
      xyz.insert(xyz.end(), {(float)x, (float)y, (float)(under ? 0.06 : 0.0)});
    }
  for (int k = 1; k < 20; k++)
    for (int i = -13; i <= 13; i++) {
      xyz.insert(xyz.end(), {(float)(i * step), 0.027f, (float)(k * step)});
      xyz.insert(xyz.end(), {(float)(i * step), -0.027f, (float)(k * step)});
    }
  for (int k = 1; k < 20; k++)
    for (int j = -8; j <= 8; j++) {
      xyz.insert(xyz.end(), {0.042f, (float)(j * step), (float)(k * step)});
      xyz.insert(xyz.end(), {-0.042f, (float)(j * step), (float)(k * step)});
    }
  const int np = (int)xyz.size() / 3;
  std::vector<int32_t> cam(np, 1);
  const double view[3] = {0.0, 0.0, 0.8};
  const int S = 96, C = 15;
  std::vector<int32_t> samples(S);
  for (int i = 0; i < S; i++) samples[i] = (int)(uniform() * np) % np;
  gpd_params P;
  gpd_oracle_default_params(&P);
  // LeNet weights of the shipped shape, small random values
  const size_t sizes[8] = {(size_t)20 * C * 25, 20, (size_t)50 * 500, 50, (size_t)500 * 7200, 500, 1000, 2};
  std::vector<std::vector<float>> w(8);
  const float *wp[8];
  for (int k = 0; k < 8; k++) {
    w[k].resize(sizes[k]);
    for (float &v : w[k]) v = (float)((uniform() - 0.5) * (k == 4 ? 0.002 : 0.1));
    wp[k] = w[k].data();
  }
  const int n_slots = P.num_hand_axes * P.num_orientations;
  auto run = [&](int threads) {
    gpd_oracle_set_num_threads(threads);
    Run r;
    r.normals.resize((size_t)np * 3);
    gpd_oracle_normals(xyz.data(), np, cam.data(), 1, view, 0.03, r.normals.data());
    r.hands.resize((size_t)S * n_slots);
    std::memset(r.hands.data(), 0, r.hands.size() * sizeof(gpd_hand));
    r.images.resize((size_t)S * n_slots * 3600 * C);
    double times[8] = {0};
    gpd_oracle_detect(&P, xyz.data(), r.normals.data(), np, cam.data(), 1, view, samples.data(), S, wp, r.hands.data(), &r.num_sets, &r.num_cand,
                      r.images.data(), S * n_slots, times);
    r.images.resize((size_t)r.num_cand * 3600 * C);
    r.labels.resize((size_t)r.num_sets * n_slots);
    std::vector<gpd_hand> again(r.hands.begin(), r.hands.begin() + (size_t)r.num_sets * n_slots);
    gpd_oracle_reevaluate(&P, xyz.data(), r.normals.data(), np, again.data(), (int)again.size(), r.labels.data());
    return r;
  };
  const Run a = run(8), b = run(1);
  const bool same = a.num_sets == b.num_sets && a.num_cand == b.num_cand && a.normals == b.normals && a.images == b.images &&
                    a.labels == b.labels && !std::memcmp(a.hands.data(), b.hands.data(), (size_t)a.num_sets * n_slots * sizeof(gpd_hand));
  std::printf("tsan_check: %d points, %d samples -> %d hand sets, %d candidates; 8 threads vs 1 thread: %s\n", np, S, a.num_sets, a.num_cand,
              same ? "identical bytes" : "DIFFERENT");
  return same && a.num_cand > 0 ? 0 : 1;
}
