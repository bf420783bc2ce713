// Point-cloud preparation on the device: the point-cloud part of Cloud::filterWorkspace
// (util/cloud.cpp:243-266) followed by Cloud::voxelizeCloud (util/cloud.cpp:286-348), in the order
// CandidatesGenerator::preprocessPointCloud runs them (candidates_generator.cpp:19-26).
//
// The workspace cut is a predicate + an order-preserving compaction.  The voxeliser is not a de-duplication:
// the reference keys a std::set with UniqueVector4First3Comparator (cloud.h:105-122), comp(a, b) = "a and b differ",
// which is not an ordering.  Under libstdc++ the descent of _M_get_insert_unique_pos goes LEFT at every node whose
// voxel differs from the new point's and right at an equal one; so
//   * a point is dropped exactly when its voxel equals the voxel of a node on the tree's LEFT SPINE at that moment
//     (the descent turns right there and ends under that node, whose key then "equals" the new one);
//   * every point that is kept becomes the new leftmost node, so the set's iteration order is the reverse of the
//     order of insertion, and the red-black tree is the one that m leftmost insertions build: its left spine after
//     the rebalancing (_Rb_tree_insert_and_rebalance) is a function of (key, colour, colour of the right child) of
//     the spine nodes alone — recolouring walks up the spine, the only rotation is a right rotation at the
//     grandparent, which takes the grandparent off the spine and makes it the parent's (red) right child.
// (krylon.pcd: 4467 points in 2373 voxels -> 3366 points kept; checked against std::set itself in
// tests/test_gpu_preprocess.py through oracle/gpd_oracle.cpp.)
// That chain of decisions is strictly sequential — which spine a point meets depends on how many points before it were
// kept — a job for one scalar core: the keys go to the host (16 bytes per point inside the workspace), one core walks
// the <= 64-node spine with a table of what the m-th insertion does to it (spine_ops: data independent), the ranks come
// back.  (Rounds 1-2 ran the walk on ONE wavefront, the spine in its lanes: ~400 cycles per point, 20 ms per 120k points
// against 1.5 ms this way and 7.9 ms for the std::set itself — DESIGN §8.)  Everything around it (keys, minimum, gather
// of the kept voxels) is data parallel and stays on the device.
#include "gpd_internal.h"
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

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

constexpr int PP_THREADS = 256;

struct Workspace {
  double w[6];
  int active;
  int drop_nonfinite;  // raw scans (Cloud::removeNans, cloud.cpp:154-164): a point with a NaN / Inf coordinate is dropped, not refused
};
struct PreMeta {      // device-side results the host reads once
  float lo[3];        // minimum of the points inside the workspace (pcl::getMinMax3D, cloud.cpp:290)
  int32_t kept;       // points inside the workspace
  int32_t voxels;     // points the voxeliser keeps
  int32_t bad;        // bit 0: non-finite coordinate, bit 1: voxel index out of the int32 range, bit 2: spine overflow
};

__device__ inline bool inside_ws(const Workspace &W, float x, float y, float z) {
  // float coordinates against double bounds, strict on both sides (cloud.cpp:246-247)
  if (W.drop_nonfinite && !(isfinite(x) && isfinite(y) && isfinite(z))) return false;
  return !W.active || ((double)x > W.w[0] && (double)x < W.w[1] && (double)y > W.w[2] && (double)y < W.w[3] && (double)z > W.w[4] &&
                       (double)z < W.w[5]);
}

// pass 1: per block, the number of points inside the workspace and their minimum
__global__ __launch_bounds__(PP_THREADS) void ws_count_kernel(const float *__restrict__ xyz, int n, Workspace W, int32_t *__restrict__ block_count,
                                                              float *__restrict__ block_lo, PreMeta *meta) {
  __shared__ int s_cnt[PP_THREADS / 64];
  __shared__ float s_lo[PP_THREADS / 64][3];
  const int i = blockIdx.x * PP_THREADS + threadIdx.x;
  float x = FLT_MAX, y = FLT_MAX, z = FLT_MAX;
  bool keep = false;
  if (i < n) {
    const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    if (!W.drop_nonfinite && !(isfinite(px) && isfinite(py) && isfinite(pz))) atomicOr(&meta->bad, 1);
    keep = inside_ws(W, px, py, pz);
    if (keep) {
      x = px;
      y = py;
      z = pz;
    }
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    x = fminf(x, __shfl_xor(x, o));
    y = fminf(y, __shfl_xor(y, o));
    z = fminf(z, __shfl_xor(z, o));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_cnt[wave] = __popcll(b);
    s_lo[wave][0] = x;
    s_lo[wave][1] = y;
    s_lo[wave][2] = z;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int w = 0; w < PP_THREADS / 64; w++) {
      c += s_cnt[w];
      for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], s_lo[w][a]);
    }
    block_count[blockIdx.x] = c;
    for (int a = 0; a < 3; a++) block_lo[3 * blockIdx.x + a] = lo[a];
  }
}

// pass 2 (one workgroup): exclusive scan of the block counts, global minimum
__global__ __launch_bounds__(1024) void ws_scan_kernel(const int32_t *__restrict__ block_count, const float *__restrict__ block_lo, int nblocks,
                                                       int32_t *__restrict__ block_off, PreMeta *meta) {
  __shared__ int s_part[16];
  __shared__ int s_carry;
  __shared__ float s_lo[16][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + tid;
    const int v = i < nblocks ? block_count[i] : 0;
    if (i < nblocks)
      for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], block_lo[3 * i + a]);
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
    if (i < nblocks) block_off[i] = before + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = before + incl;
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
  if (lane == 0)
    for (int a = 0; a < 3; a++) s_lo[wave][a] = lo[a];
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; w++)
      for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], s_lo[w][a]);
    for (int a = 0; a < 3; a++) meta->lo[a] = lo[a];
    meta->kept = s_carry;
    meta->voxels = s_carry;  // without a voxeliser pass every point inside the workspace is kept
  }
}

// pass 3: the compaction.  src[pos] = index of the pos-th point inside the workspace; with a cell size its voxel
// (cloud.cpp:298-301: floor((pt - min_pt) / cell_size) in float) goes to keys[pos] = (ix, iy, iz, index).
__global__ __launch_bounds__(PP_THREADS) void ws_scatter_kernel(const float *__restrict__ xyz, int n, Workspace W, const int32_t *__restrict__ block_off,
                                                                float cell, PreMeta *meta, int32_t *__restrict__ src, int4 *__restrict__ keys) {
  __shared__ int s_cnt[PP_THREADS / 64];
  const int i = blockIdx.x * PP_THREADS + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  bool keep = false;
  if (i < n) {
    px = xyz[3 * (size_t)i];
    py = xyz[3 * (size_t)i + 1];
    pz = xyz[3 * (size_t)i + 2];
    keep = inside_ws(W, px, py, pz);
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wave] = __popcll(b);
  __syncthreads();
  if (!keep) return;
  int pos = block_off[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; w++) pos += s_cnt[w];
  src[pos] = i;
  if (cell > 0.f) {
    const float q[3] = {floorf((px - meta->lo[0]) / cell), floorf((py - meta->lo[1]) / cell), floorf((pz - meta->lo[2]) / cell)};
    if (!(q[0] < 2147483520.f && q[1] < 2147483520.f && q[2] < 2147483520.f)) atomicOr(&meta->bad, 2);
    keys[pos] = make_int4((int)q[0], (int)q[1], (int)q[2], i);
  }
}

// pass 5: the kept voxels in the set's iteration order (reverse order of insertion): voxel -> point
// (cloud.cpp:322: min_pt + cell_size * voxel), camera source of the point that opened the voxel (:325-327: == 1 ? 1 : 0)
__global__ __launch_bounds__(PP_THREADS) void voxel_emit_kernel(const int4 *__restrict__ keys, const int32_t *__restrict__ rank, const PreMeta *meta,
                                                                float cell, const int32_t *__restrict__ cam, int n, int num_cams,
                                                                float *__restrict__ out_xyz, int32_t *__restrict__ out_cam, int32_t *__restrict__ out_src) {
  const int pos = blockIdx.x * PP_THREADS + threadIdx.x;
  if (meta->bad || pos >= meta->kept) return;  // a refused cloud has no ranks (the host route never wrote them)
  const int r = rank[pos];
  const int M = meta->voxels;
  if (r < 0 || r >= M) return;
  const int o = M - 1 - r;
  const int4 k = keys[pos];
  out_xyz[3 * (size_t)o] = meta->lo[0] + cell * (float)k.x;
  out_xyz[3 * (size_t)o + 1] = meta->lo[1] + cell * (float)k.y;
  out_xyz[3 * (size_t)o + 2] = meta->lo[2] + cell * (float)k.z;
  out_src[o] = k.w;
  for (int c = 0; c < num_cams; c++) out_cam[(size_t)c * M + o] = cam[(size_t)c * n + k.w] == 1 ? 1 : 0;
}

// without a voxeliser: the points inside the workspace as they are, camera source columns copied (cloud.cpp:252-259)
__global__ __launch_bounds__(PP_THREADS) void ws_emit_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ src, const PreMeta *meta,
                                                             const int32_t *__restrict__ cam, int n, int num_cams, float *__restrict__ out_xyz,
                                                             int32_t *__restrict__ out_cam) {
  const int pos = blockIdx.x * PP_THREADS + threadIdx.x;
  const int K = meta->kept;
  if (pos >= K) return;
  const int i = src[pos];
  for (int a = 0; a < 3; a++) out_xyz[3 * (size_t)pos + a] = xyz[3 * (size_t)i + a];
  for (int c = 0; c < num_cams; c++) out_cam[(size_t)c * K + pos] = cam[(size_t)c * n + i];
}

}  // namespace

// The structural half of the voxeliser's tree walk, which is the same for every cloud: ops[m] = depth of the node that
// the m-th leftmost insertion rotates off the left spine (-1: recolouring only).  Colours as two bit masks over the
// spine depths (node red / its right child red); _Rb_tree_insert_and_rebalance restricted to left children: a red
// uncle recolours and moves two levels up, a black one ends with a right rotation at the grandparent.
static bool spine_ops(std::vector<int8_t> &ops, size_t n, unsigned long long &C, unsigned long long &R, int &L) {
  while (ops.size() < n) {
    if (L == 64) return false;
    C |= 1ull << L;
    R &= ~(1ull << L);
    int x = L++, gone = -1;
    while (x > 0 && (C >> (x - 1) & 1ull)) {
      const int g = x - 2;  // a red parent is not the root
      if (R >> g & 1ull) {
        C = (C & ~(1ull << (x - 1))) | (1ull << g);
        R &= ~(1ull << g);
        x = g;
      } else {
        C &= ~(1ull << (x - 1));
        R |= 1ull << (x - 1);
        const unsigned long long low = (1ull << g) - 1ull;
        C = (C & low) | ((C >> 1) & ~low);
        R = (R & low) | ((R >> 1) & ~low);
        L--;
        gone = g;
        break;
      }
    }
    C &= ~1ull;
    ops.push_back((int8_t)gone);
  }
  return true;
}

// The walk: a few dozen cycles per point (~150 for the std::set the reference itself uses, which allocates a node per
// kept point).  rank[pos] = how many points were kept before this one, -1 for a dropped point.
static int voxel_accept_host(const int4 *keys, int n, const int8_t *ops, int32_t *rank) {
  int sx[64], sy[64], sz[64];
  // A kept point — on a raw scan nearly every one — matches NO spine voxel, and proving that by comparing with all ~35 of
  // them was the walk's cost (8 ns per point).  A 1024-slot counting filter over the spine's voxels answers "certainly not on
  // the spine" in one load; only a point whose slot is occupied walks the spine (round 4: gpd_hip_preprocess_cloud on a raw 120k-point scan 1.42 -> 0.68 ms first to last device operation).
  unsigned short filter[1024];
  std::memset(filter, 0, sizeof(filter));
  auto slot = [](int x, int y, int z) {
    const unsigned h = (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u;
    return (h ^ (h >> 13)) & 1023u;
  };
  int L = 0, m = 0;
  for (int i = 0; i < n; i++) {
    const int qx = keys[i].x, qy = keys[i].y, qz = keys[i].z;
    const unsigned h = slot(qx, qy, qz);
    bool hit = false;
    if (filter[h])
      for (int d = L - 1; d >= 0; d--)  // the recent nodes sit at the bottom of the spine: most drops end here at once
        if (sx[d] == qx && sy[d] == qy && sz[d] == qz) {
          hit = true;
          break;
        }
    if (hit) {
      rank[i] = -1;
      continue;
    }
    rank[i] = m;
    sx[L] = qx;
    sy[L] = qy;
    sz[L] = qz;
    filter[h]++;
    const int g = ops[m];
    m++;
    L++;
    if (g >= 0) {  // the node at depth g leaves the spine, the ones below move up
      filter[slot(sx[g], sy[g], sz[g])]--;
      for (int d = g; d + 1 < L; d++) {
        sx[d] = sx[d + 1];
        sy[d] = sy[d + 1];
        sz[d] = sz[d + 1];
      }
      L--;
    }
  }
  return m;
}

void preprocess_free(PreState &s) {
  void *dev[] = {s.d_xyz, s.d_cam, s.d_block_count, s.d_block_off, s.d_block_lo, s.d_src, s.d_keys, s.d_rank, s.d_out_xyz, s.d_out_cam, s.d_out_src, s.d_meta};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  if (s.h_pin) (void)hipHostFree(s.h_pin);
  hipEvent_t e0 = s.ev[0], e1 = s.ev[1], e2 = s.ev_keys;  // the events outlive a re-allocation
  s = PreState();
  s.ev[0] = e0;
  s.ev[1] = e1;
  s.ev_keys = e2;
}

static int pre_check(const PreMeta &hm) {
  // a refused cloud (non-finite coordinate, voxel index beyond int32) stops: no ranks exist for it, and voxel_emit_kernel
  // must not run on the ranks an earlier call left in d_rank
  if (hm.bad & 1) {
    set_error("preprocess_cloud: the cloud holds non-finite coordinates (remove NaN/Inf points first)");
    return GPD_ERR_INVALID;
  }
  if (hm.bad & 6) {
    set_error("preprocess_cloud: %s", (hm.bad & 2) ? "a voxel index does not fit 32 bits (voxel size too small for the cloud's extent)"
                                                   : "more points than the voxeliser's tree walk supports");
    return GPD_ERR_CAPACITY;
  }
  return GPD_OK;
}

// the spine table up to n kept points (it only ever grows)
static std::vector<int8_t> g_ops;
static unsigned long long g_ops_C = 0ull, g_ops_R = 0ull;
static int g_ops_L = 0;
static std::mutex g_ops_mutex;

// Phase 1, nothing waits: the scan to the device, the workspace cut, the voxel keys, and the keys + the counters on their way
// back into pinned memory (all n slots: how many are inside the workspace is only known on the device).  gpd_hip_detect_batch
// enqueues this for cloud i + 1 BEFORE it waits for anything of cloud i.
int preprocess_begin(PreState &s, const float *xyz, const int32_t *cam_source, int n, int num_cams, const double *workspace, float cell,
                     hipStream_t stream, bool drop_nonfinite) {
  s.n = n;
  s.num_cams = num_cams;
  s.cell = cell;
  s.M = 0;
  if (n == 0) return GPD_OK;
  if (n > s.capacity || num_cams > s.cap_cams) {
    const int cap = n > s.capacity ? n + n / 4 : s.capacity, cams = num_cams > s.cap_cams ? num_cams : s.cap_cams;
    note_alloc(__func__);
    preprocess_free(s);
    s.n = n;
    s.num_cams = num_cams;
    s.cell = cell;
    const int blocks = (cap + PP_THREADS - 1) / PP_THREADS;
    HIP_RET(hipMalloc(&s.d_xyz, (size_t)cap * 3 * sizeof(float)));
    HIP_RET(hipMalloc(&s.d_cam, (size_t)cap * (cams > 0 ? cams : 1) * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_block_count, (size_t)blocks * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_block_off, (size_t)blocks * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_block_lo, (size_t)blocks * 3 * sizeof(float)));
    HIP_RET(hipMalloc(&s.d_src, (size_t)cap * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_keys, (size_t)cap * sizeof(int4)));
    HIP_RET(hipMalloc(&s.d_rank, (size_t)cap * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_out_xyz, (size_t)cap * 3 * sizeof(float)));
    HIP_RET(hipMalloc(&s.d_out_cam, (size_t)cap * (cams > 0 ? cams : 1) * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_out_src, (size_t)cap * sizeof(int32_t)));
    HIP_RET(hipMalloc(&s.d_meta, sizeof(PreMeta)));
    // pinned: [keys 16 B x cap][ranks 4 B x cap][PreMeta][bounds 6 floats]
    HIP_RET(hipHostMalloc(reinterpret_cast<void **>(&s.h_pin), (size_t)cap * 20 + 256, 0));
    s.capacity = cap;
    s.cap_cams = cams;
  }
  if (!s.ev[0])
    for (auto &e : s.ev) HIP_RET(hipEventCreate(&e));
  if (!s.ev_keys) HIP_RET(hipEventCreateWithFlags(&s.ev_keys, hipEventDisableTiming));
  Workspace W;
  W.active = workspace != nullptr;
  W.drop_nonfinite = drop_nonfinite ? 1 : 0;
  for (int a = 0; a < 6; a++) W.w[a] = workspace ? workspace[a] : 0.0;
  PreMeta *meta = static_cast<PreMeta *>(s.d_meta);
  const int blocks = (n + PP_THREADS - 1) / PP_THREADS;
  HIP_RET(hipMemcpyAsync(s.d_xyz, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, stream));
  if (num_cams > 0) HIP_RET(hipMemcpyAsync(s.d_cam, cam_source, (size_t)n * num_cams * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIP_RET(hipEventRecord(s.ev[0], stream));
  HIP_RET(hipMemsetAsync(s.d_meta, 0, sizeof(PreMeta), stream));
  ws_count_kernel<<<blocks, PP_THREADS, 0, stream>>>(s.d_xyz, n, W, s.d_block_count, s.d_block_lo, meta);
  ws_scan_kernel<<<1, 1024, 0, stream>>>(s.d_block_count, s.d_block_lo, blocks, s.d_block_off, meta);
  ws_scatter_kernel<<<blocks, PP_THREADS, 0, stream>>>(s.d_xyz, n, W, s.d_block_off, cell, meta, s.d_src, s.d_keys);
  HIP_RET(hipGetLastError());
  char *hp = s.h_pin;
  HIP_RET(hipMemcpyAsync(hp + (size_t)s.capacity * 20, s.d_meta, sizeof(PreMeta), hipMemcpyDeviceToHost, stream));
  if (cell > 0.f) {
    {
      std::lock_guard<std::mutex> lock(g_ops_mutex);
      if (!spine_ops(g_ops, (size_t)n, g_ops_C, g_ops_R, g_ops_L)) {
        set_error("preprocess_cloud: more points than the voxeliser's tree walk supports");
        return GPD_ERR_CAPACITY;
      }
    }
    HIP_RET(hipMemcpyAsync(hp, s.d_keys, (size_t)n * sizeof(int4), hipMemcpyDeviceToHost, stream));
  }
  HIP_RET(hipEventRecord(s.ev_keys, stream));
  return GPD_OK;
}

// Phase 2: waits for the keys, walks the spine on this host core (the strictly sequential chain of the voxeliser's keep / drop
// decisions), sends the ranks and gathers the kept voxels.  On return s.M points lie on the device: s.d_out_xyz [M][3],
// s.d_out_cam [cams][M], s.d_out_src [M] (input index; without a voxeliser: s.d_src), enqueued on `stream`.
int preprocess_finish(PreState &s, hipStream_t stream) {
  s.M = 0;
  const int n = s.n;
  if (n == 0) return GPD_OK;
  HIP_RET(hipEventSynchronize(s.ev_keys));
  char *hp = s.h_pin;
  const PreMeta hm = *reinterpret_cast<const PreMeta *>(hp + (size_t)s.capacity * 20);
  int rc = pre_check(hm);
  if (rc) return rc;
  PreMeta *meta = static_cast<PreMeta *>(s.d_meta);
  const int blocks = (n + PP_THREADS - 1) / PP_THREADS;
  const int kept = hm.kept;
  if (s.cell > 0.f) {
    int m = 0;
    if (kept > 0) {
      int32_t *rank = reinterpret_cast<int32_t *>(hp + (size_t)s.capacity * 16);
      {
        std::lock_guard<std::mutex> lock(g_ops_mutex);  // `g_ops` may grow under another context's call
        m = voxel_accept_host(reinterpret_cast<const int4 *>(hp), kept, g_ops.data(), rank);
      }
      int32_t *hm_voxels = reinterpret_cast<int32_t *>(hp + (size_t)s.capacity * 20 + 64);
      *hm_voxels = m;
      HIP_RET(hipMemcpyAsync(s.d_rank, rank, (size_t)kept * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      HIP_RET(hipMemcpyAsync(reinterpret_cast<char *>(s.d_meta) + offsetof(PreMeta, voxels), hm_voxels, sizeof(int32_t), hipMemcpyHostToDevice, stream));
      voxel_emit_kernel<<<blocks, PP_THREADS, 0, stream>>>(s.d_keys, s.d_rank, meta, s.cell, s.d_cam, n, s.num_cams, s.d_out_xyz, s.d_out_cam, s.d_out_src);
    }
    s.M = m;
  } else {
    if (kept > 0) ws_emit_kernel<<<blocks, PP_THREADS, 0, stream>>>(s.d_xyz, s.d_src, meta, s.d_cam, n, s.num_cams, s.d_out_xyz, s.d_out_cam);
    s.M = kept;
  }
  HIP_RET(hipGetLastError());
  HIP_RET(hipEventRecord(s.ev[1], stream));
  return GPD_OK;
}

int preprocess_run(PreState &s, const float *xyz, const int32_t *cam_source, int n, int num_cams, const double *workspace, float cell,
                   float *xyz_out, int32_t *cam_out, int32_t *src_out, int *num_out, float *ms, hipStream_t stream) {
  *num_out = 0;
  if (ms) *ms = 0.f;
  if (n == 0) return GPD_OK;
  int rc = preprocess_begin(s, xyz, cam_source, n, num_cams, workspace, cell, stream, /*drop_nonfinite=*/false);
  if (!rc) rc = preprocess_finish(s, stream);
  if (rc) {
    (void)hipStreamSynchronize(stream);
    return rc;
  }
  const int M = s.M;
  if (M > 0) {
    HIP_RET(hipMemcpyAsync(xyz_out, s.d_out_xyz, (size_t)M * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (num_cams > 0) HIP_RET(hipMemcpyAsync(cam_out, s.d_out_cam, (size_t)M * num_cams * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (src_out) HIP_RET(hipMemcpyAsync(src_out, cell > 0.f ? s.d_out_src : s.d_src, (size_t)M * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  }
  HIP_RET(hipStreamSynchronize(stream));
  if (ms) HIP_RET(hipEventElapsedTime(ms, s.ev[0], s.ev[1]));
  *num_out = M;
  return GPD_OK;
}

}  // namespace gpd
