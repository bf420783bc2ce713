// Clustering::findClusters (clustering.cpp:5-105) on the device — SURVEY §8f rank 3, the step after selectGrasps.
//
// For every seed hand i the hands j whose axis lies within 12 degrees of i's, whose position lies within 0.05 m and,
// projected onto the plane orthogonal to i's axis, within 0.005 m (:39-59) are its inliers; a seed with at least
// min_inliers of them becomes a cluster: position = mean inlier position, score = lower bound of the 99 % confidence
// interval of the inlier scores (running mean / variance in inlier order, :66-69, :83-92).
// One workgroup per seed: the lanes test the pairs (i, j) in chunks of 256 and compact the inliers of a chunk in j
// order (ballot + prefix), thread 0 feeds them to the running sums — the only sequential part, and it is as long as
// the inlier list, not as n.  With remove_inliers the seeds depend on each other (a hand that was an inlier once is
// skipped by the later seeds, :36, :70-72): the same workgroup code walks the seeds one after the other.
// A one-workgroup scan then writes the clusters in seed order.  fp64, unfused, sums in the reference's order.
#include "gpd_internal.h"
#include <cmath>

#define HIP_RET(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GPD_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

namespace gpd {

namespace {

constexpr int CL_THREADS = 256;

struct ClusterParams {
  const gpd_hand *hands;
  const double *scores;
  int n, min_inliers, remove_inliers;
  double cos_thresh, max_dist, proj_dist;  // cos(12 deg) as the host's libm gives it, 0.05, 0.005 (clustering.cpp:9-13)
  uint8_t *used;                            // [n] remove_inliers only
  int32_t *keep;                            // [n] 1: seed i is a cluster
  double *res;                              // [n][4] position + score of the cluster seeded by i
};

__device__ void seed(const ClusterParams &P, int i, int *s_list, int *s_cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const gpd_hand &hi = P.hands[i];
  const double ai[3] = {hi.frame[2], hi.frame[5], hi.frame[8]};  // Hand::getAxis = orientation.col(2)
  const double pi[3] = {hi.position[0], hi.position[1], hi.position[2]};
  double proj[3][3];  // I - a a^T (:32, :53)
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) proj[r][c] = (r == c ? 1.0 : 0.0) - ai[r] * ai[c];
  // thread 0's running sums
  int num_inliers = 0;
  double pd[3] = {0.0, 0.0, 0.0}, mean = 0.0, sd = 0.0;
  for (int j0 = 0; j0 < P.n; j0 += CL_THREADS) {
    const int j = j0 + tid;
    bool in = false;
    if (j < P.n && j != i && !(P.remove_inliers && P.used[j])) {
      const gpd_hand &hj = P.hands[j];
      const double axis_aligned = ai[0] * hj.frame[2] + ai[1] * hj.frame[5] + ai[2] * hj.frame[8];
      const double d[3] = {pi[0] - hj.position[0], pi[1] - hj.position[1], pi[2] - hj.position[2]};
      const double mag = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      double q[3];
#pragma unroll
      for (int r = 0; r < 3; r++) q[r] = proj[r][0] * d[0] + proj[r][1] * d[1] + proj[r][2] * d[2];
      const double pmag = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      in = fabs(axis_aligned) > P.cos_thresh && mag <= P.max_dist && pmag <= P.proj_dist;
    }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(in);
    if (lane == 0) s_cnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < CL_THREADS / 64; w++) {
      if (w < wave) before += s_cnt[w];
      total += s_cnt[w];
    }
    if (in) {
      s_list[before + __popcll(b & ((1ull << lane) - 1ull))] = j;
      if (P.remove_inliers) P.used[j] = 1;
    }
    __syncthreads();
    if (tid == 0)
      for (int q = 0; q < total; q++) {  // the chunk's inliers in j order (:62-69)
        const int jj = s_list[q];
        num_inliers++;
        for (int r = 0; r < 3; r++) pd[r] += P.hands[jj].position[r];
        const double sj = P.scores[jj], old_mean = mean;
        mean += (sj - mean) / static_cast<double>(num_inliers);
        sd += (sj - mean) * (sj - old_mean);
      }
    __syncthreads();
  }
  if (tid == 0) {
    int keep = 0;
    if (num_inliers >= P.min_inliers) {  // :76-97
      const double dn = static_cast<double>(num_inliers);
      for (int r = 0; r < 3; r++) pd[r] = pd[r] / dn - pi[r];
      sd /= dn;
      if (sd != 0) sd = sqrt(sd);
      const double conf_lb = mean - 2.576 * sd / sqrt((double)num_inliers);
      for (int r = 0; r < 3; r++) P.res[4 * (size_t)i + r] = pi[r] + pd[r];
      P.res[4 * (size_t)i + 3] = conf_lb;
      keep = 1;
    }
    P.keep[i] = keep;
  }
}

__global__ __launch_bounds__(CL_THREADS) void cluster_kernel(ClusterParams P) {
  __shared__ int s_list[CL_THREADS];
  __shared__ int s_cnt[CL_THREADS / 64];
  if (P.remove_inliers) {
    for (int i = 0; i < P.n; i++) {
      seed(P, i, s_list, s_cnt);
      __threadfence();  // the used[] flags of this seed are seen by the next one's tests
      __syncthreads();
    }
  } else {
    seed(P, blockIdx.x, s_list, s_cnt);
  }
}

// clusters in seed order: out[k] = the seed's record with the cluster's position, out_scores[k], out_src[k] = seed
__global__ __launch_bounds__(1024) void cluster_emit_kernel(ClusterParams P, gpd_hand *out, double *out_scores, int32_t *out_src, int32_t *num_out) {
  __shared__ int s_part[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < P.n; base += 1024) {
    const int i = base + tid;
    const int v = i < P.n ? P.keep[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; w++) before += s_part[w];
    if (v) {
      const int k = before + incl - 1;
      gpd_hand h = P.hands[i];
      for (int r = 0; r < 3; r++) h.position[r] = P.res[4 * (size_t)i + r];
      h.score = (float)P.res[4 * (size_t)i + 3];
      out[k] = h;
      out_scores[k] = P.res[4 * (size_t)i + 3];
      out_src[k] = i;
    }
    __syncthreads();
    if (tid == 1023) s_carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) *num_out = s_carry;
}

}  // namespace

void cluster_free(ClusterState &s) {
  void *dev[] = {s.d_hands, s.d_scores, s.d_used, s.d_keep, s.d_res, s.d_out, s.d_out_scores, s.d_out_src, s.d_num};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  s = ClusterState();
}

int cluster_run(ClusterState &s, const gpd_hand *hands, const double *scores, int n, int min_inliers, int remove_inliers, gpd_hand *out,
                double *out_scores, int32_t *out_src, int *num_out, hipStream_t stream) {
  *num_out = 0;
  if (n == 0) return GPD_OK;
  if (n > s.capacity) {
    const int cap = n + n / 4;
    cluster_free(s);
    HIP_RET(hipMalloc(&s.d_hands, (size_t)cap * sizeof(gpd_hand)));
    HIP_RET(hipMalloc(&s.d_scores, (size_t)cap * sizeof(double)));
    HIP_RET(hipMalloc(&s.d_used, (size_t)cap));
    HIP_RET(hipMalloc(&s.d_keep, (size_t)cap * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_res, (size_t)cap * 4 * sizeof(double)));
    HIP_RET(hipMalloc(&s.d_out, (size_t)cap * sizeof(gpd_hand)));
    HIP_RET(hipMalloc(&s.d_out_scores, (size_t)cap * sizeof(double)));
    HIP_RET(hipMalloc(&s.d_out_src, (size_t)cap * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_num, sizeof(int32_t)));
    s.capacity = cap;
  }
  HIP_RET(hipMemcpyAsync(s.d_hands, hands, (size_t)n * sizeof(gpd_hand), hipMemcpyHostToDevice, stream));
  HIP_RET(hipMemcpyAsync(s.d_scores, scores, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
  if (remove_inliers) HIP_RET(hipMemsetAsync(s.d_used, 0, (size_t)n, stream));
  ClusterParams P;
  P.hands = s.d_hands;
  P.scores = s.d_scores;
  P.n = n;
  P.min_inliers = min_inliers;
  P.remove_inliers = remove_inliers ? 1 : 0;
  P.cos_thresh = std::cos(12.0 * M_PI / 180.0);
  P.max_dist = 0.05;
  P.proj_dist = 0.005;
  P.used = s.d_used;
  P.keep = s.d_keep;
  P.res = s.d_res;
  cluster_kernel<<<remove_inliers ? 1 : n, CL_THREADS, 0, stream>>>(P);
  cluster_emit_kernel<<<1, 1024, 0, stream>>>(P, s.d_out, s.d_out_scores, s.d_out_src, s.d_num);
  HIP_RET(hipGetLastError());
  int32_t k = 0;
  HIP_RET(hipMemcpyAsync(&k, s.d_num, sizeof(k), hipMemcpyDeviceToHost, stream));
  HIP_RET(hipStreamSynchronize(stream));
  if (k > 0) {
    HIP_RET(hipMemcpyAsync(out, s.d_out, (size_t)k * sizeof(gpd_hand), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipMemcpyAsync(out_scores, s.d_out_scores, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipMemcpyAsync(out_src, s.d_out_src, (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP_RET(hipStreamSynchronize(stream));
  }
  *num_out = k;
  return GPD_OK;
}

}  // namespace gpd
