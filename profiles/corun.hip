// Two questions VERDICT r3 left open, answered by measurement on one MI355X (gfx950):
//
//  A. What does the f32 matrix pipe sustain?  profiles/r01m_mfma_rates.txt saw 104-115 TFLOP/s chip-wide for 16x16x4 and
//     116-132 for 32x32x2, MI355X_MICROARCH.md records 155 measured for both.  Here: pure MFMA streams, operands from
//     registers only, 1 / 2 / 4 waves per SIMD, ~4 ms per launch (launch overhead < 1 %), the shader clock measured
//     INSIDE the kernel (s_memtime cycles against the 100 MHz s_memrealtime), cycles per MFMA per SIMD next to TFLOP/s.
//
//  B. Can MFMA work and VALU / LDS work share a CU?  Every kernel of the path owns its CUs through LDS today, so the
//     VALU-bound image stage (1.0 ms) and the MFMA-bound LeNet (3.05 ms) run one after the other.  Here: a 128-VGPR,
//     81 KB-LDS f32-MFMA stream (M) and a 128-VGPR, 78 KB-LDS VALU / LDS kernel (V: int32 address arithmetic, random LDS
//     atomics, 16-byte LDS reads, a little f64 — the instruction mix of shadow_image_kernel), 256 workgroups of 512
//     threads each, so that one M and one V workgroup fit a CU together and two of a kind do not.  Alone, then on two
//     streams at once; the HW_ID of every workgroup says on how many CUs both kinds actually sat together.
//     co-run time <= 0.8 x (sum of the alone times) would justify re-shaping conv1 / the image kernels to co-reside.
//
//   hipcc --offload-arch=gfx950 -O2 profiles/corun.hip -o /tmp/corun && /tmp/corun
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define CK(e)                                                                     \
  do {                                                                            \
    hipError_t e_ = (e);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

struct Probe {  // per workgroup
  unsigned hw_id, xcc_id;
  long long cycles, ticks;  // shader cycles / 100 MHz ticks of wave 0
};

__device__ __forceinline__ void probe(Probe *p, long long c0, long long r0) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p->hw_id = hw;
    p->xcc_id = xcc;
    p->cycles = __builtin_readcyclecounter() - c0;
    p->ticks = (long long)__builtin_amdgcn_s_memrealtime() - r0;
  }
}

// ---- M: a pure f32 MFMA stream.  shape 0: 16x16x4 (four independent accumulators), 1: 32x32x2 (two)
template <int SHAPE>
__global__ __launch_bounds__(1024) void mfma_kernel(int iters, float *sink, Probe *probes) {
  extern __shared__ float dyn_lds[];
  const long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  const float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
  float s = 0.f;
  if (SHAPE == 0) {
    f32x4 A0 = {0, 0, 0, 0}, A1 = A0, A2 = A0, A3 = A0;
    for (int it = 0; it < iters; it++) {
      REP16(A0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, A1, 0, 0, 0);
            A2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, A2, 0, 0, 0); A3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, A3, 0, 0, 0);)
    }
    s = A0[0] + A1[1] + A2[2] + A3[3];
  } else {
    f32x16 A0, A1;
    for (int i = 0; i < 16; i++) {
      A0[i] = 0;
      A1[i] = 0;
    }
    for (int it = 0; it < iters; it++) {
      REP16(A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A1, 0, 0, 0);
            A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A1, 0, 0, 0);)
    }
    for (int i = 0; i < 16; i++) s += A0[i] + A1[i];
  }
  if (s == 12345.678f) dyn_lds[threadIdx.x] = s;  // keeps the dynamic LDS allocated and the result alive
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  probe(probes + blockIdx.x, c0, r0);
}

// ---- V: VALU / LDS work in the mix of the image kernels.  MIX 0: VALU only (int32 + f32 + a little f64);
//      1: + random LDS atomics and 16-byte LDS reads (the image kernels); 2: LDS heavy
template <int MIX>
__global__ __launch_bounds__(512) void valu_kernel(int iters, float *sink, Probe *probes, int lds_words) {
  extern __shared__ unsigned dyn_u[];
  const long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  const int tid = threadIdx.x;
  for (int i = tid; i < lds_words; i += 512) dyn_u[i] = i * 2654435761u;
  __syncthreads();
  const unsigned mask = (unsigned)lds_words - 1u;  // lds_words is a power of two
  unsigned u = tid * 747796405u + 2891336453u, acc = 0;
  float f = (float)tid;
  double d = 1.0 + tid * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      // int32 address arithmetic (a third of the image kernels' vector instructions)
      u = u * 1664525u + 1013904223u;
      const unsigned cell = (u >> 9) & mask;
      const unsigned row = (cell >> 5) * 33u + (cell & 31u);
      acc += (row ^ (u >> 3)) + __popc(u);
      f = f * 1.0000001f + (float)(cell & 255u);
      if (MIX >= 1) {
        atomicAdd(&dyn_u[cell], 1u);  // counting sort by pixel: random addresses
        const uint4 v = *reinterpret_cast<const uint4 *>(&dyn_u[(row & mask) & ~3u]);
        acc += v.x + v.y + v.z + v.w;
      }
      if (MIX >= 2) {
        atomicMax(&dyn_u[(cell * 7u) & mask], u);
        const uint4 v = *reinterpret_cast<const uint4 *>(&dyn_u[((acc >> 4) & mask) & ~3u]);
        acc ^= v.x + v.w;
      }
    }
    d = d * 1.0000000001 + (double)f;  // unfused f64: one operation in eight
    if (MIX >= 1 && (it & 15) == 15) __syncthreads();  // barrier-separated phases
  }
  sink[blockIdx.x * 512 + tid] = f + (float)acc + (float)d;
  probe(probes + blockIdx.x, c0, r0);
}

static float median(std::vector<float> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

static unsigned cu_key(const Probe &p) {  // XCC | SE | SH | CU
  return ((p.xcc_id & 0xf) << 12) | (((p.hw_id >> 13) & 0x7) << 8) | (((p.hw_id >> 12) & 0x1) << 4) | ((p.hw_id >> 8) & 0xf);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, %zu KB LDS per workgroup max\n", prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock / 1024);
  float *sink;
  Probe *pm, *pv;
  CK(hipMalloc(&sink, 4096 * 1024 * sizeof(float)));
  CK(hipMalloc(&pm, 4096 * sizeof(Probe)));
  CK(hipMalloc(&pv, 4096 * sizeof(Probe)));
  std::vector<Probe> hm(4096), hv(4096);
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));

  // ---------------- A: what the f32 matrix pipe sustains
  printf("\n== A. pure f32 MFMA streams, 256 workgroups, operands in registers ==\n");
  printf("%-10s %-14s %9s %11s %10s %14s %12s\n", "shape", "waves/SIMD", "ms", "TFLOP/s", "clock GHz", "cyc/MFMA/SIMD", "TF @2.4 GHz");
  for (int shape = 0; shape < 2; shape++)
    for (int wps : {1, 2, 4}) {
      const int threads = 256 * wps;
      // ~4 ms: MFMAs per SIMD = 64 * iters * wps at 32 (16x16x4) / 64 (32x32x2) cycles each
      const int iters = (shape == 0 ? 4200 : 2100) / wps;
      std::vector<float> ms;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        if (shape == 0)
          hipLaunchKernelGGL(mfma_kernel<0>, 256, threads, 0, 0, iters, sink, pm);
        else
          hipLaunchKernelGGL(mfma_kernel<1>, 256, threads, 0, 0, iters, sink, pm);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
      }
      CK(hipMemcpy(hm.data(), pm, 256 * sizeof(Probe), hipMemcpyDeviceToHost));
      double cyc = 0, tick = 0;
      for (int i = 0; i < 256; i++) {
        cyc += (double)hm[i].cycles;
        tick += (double)hm[i].ticks;
      }
      const double ghz = cyc / (tick * 10.0) ;  // cycles per 10 ns -> GHz = cycles / (ticks * 10 ns) * 1e-9... (ticks of 10 ns)
      const double mfmas_per_wave = 64.0 * iters;
      const double flop_per = shape == 0 ? 16.0 * 16 * 4 * 2 : 32.0 * 32 * 2 * 2;
      const double total = 256.0 * (threads / 64) * mfmas_per_wave * flop_per;
      const float t = median(ms);
      const double cyc_per = (cyc / 256.0) / (mfmas_per_wave * wps);  // per SIMD: wps waves share it
      printf("%-10s %-14d %9.3f %11.1f %10.3f %14.2f %12.1f\n", shape == 0 ? "16x16x4" : "32x32x2", wps, t, total / t / 1e9, ghz, cyc_per,
             flop_per / cyc_per * 1024 * 2.4 / 1e3);
    }

  // ---------------- B: co-residency
  printf("\n== B. MFMA stream (M: 81 KB LDS) beside VALU / LDS work (V: 78 KB LDS), 256 workgroups x 512 threads each ==\n");
  const size_t lds_m = 81 * 1024, lds_v = 78 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  const int it_m = 1000;  // 64000 MFMAs per wave, two waves per SIMD: ~1.9 ms
  for (int mix = 0; mix < 3; mix++) {
    const void *vf = mix == 0 ? reinterpret_cast<const void *>(valu_kernel<0>)
                              : (mix == 1 ? reinterpret_cast<const void *>(valu_kernel<1>) : reinterpret_cast<const void *>(valu_kernel<2>));
    CK(hipFuncSetAttribute(vf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v));
    const int words = 16384;  // 64 KB of the 78 are addressed
    auto launch_v = [&](hipStream_t st, int iters) {
      if (mix == 0)
        hipLaunchKernelGGL(valu_kernel<0>, 256, 512, lds_v, st, iters, sink + 2048 * 1024, pv, words);
      else if (mix == 1)
        hipLaunchKernelGGL(valu_kernel<1>, 256, 512, lds_v, st, iters, sink + 2048 * 1024, pv, words);
      else
        hipLaunchKernelGGL(valu_kernel<2>, 256, 512, lds_v, st, iters, sink + 2048 * 1024, pv, words);
    };
    auto launch_m = [&](hipStream_t st) { hipLaunchKernelGGL(mfma_kernel<0>, 256, 512, lds_m, st, it_m, sink, pm); };
    auto time_one = [&](bool is_m, int iters, float *out) -> int {
      std::vector<float> ms;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, s1));
        if (is_m)
          launch_m(s1);
        else
          launch_v(s1, iters);
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
      }
      *out = median(ms);
      return 0;
    };
    float t_m = 0, t_v = 0;
    if (time_one(true, 0, &t_m)) return 1;
    // size V to the same duration
    int it_v = 400;
    if (time_one(false, it_v, &t_v)) return 1;
    it_v = std::max(16, (int)(it_v * t_m / t_v) / 16 * 16);
    if (time_one(false, it_v, &t_v)) return 1;
    std::vector<float> co;
    int both = 0;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      CK(hipStreamWaitEvent(s2, e0, 0));
      launch_v(s1, it_v);
      launch_m(s2);
      CK(hipEventRecord(e1, s1));
      CK(hipEventRecord(e2, s2));
      CK(hipEventSynchronize(e1));
      CK(hipEventSynchronize(e2));
      float a, b;
      CK(hipEventElapsedTime(&a, e0, e1));
      CK(hipEventElapsedTime(&b, e0, e2));
      if (rep) co.push_back(std::max(a, b));
    }
    CK(hipMemcpy(hm.data(), pm, 256 * sizeof(Probe), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hv.data(), pv, 256 * sizeof(Probe), hipMemcpyDeviceToHost));
    std::vector<unsigned> cm, cv;
    for (int i = 0; i < 256; i++) {
      cm.push_back(cu_key(hm[i]));
      cv.push_back(cu_key(hv[i]));
    }
    std::sort(cm.begin(), cm.end());
    std::sort(cv.begin(), cv.end());
    cm.erase(std::unique(cm.begin(), cm.end()), cm.end());
    cv.erase(std::unique(cv.begin(), cv.end()), cv.end());
    for (unsigned k : cm) both += std::binary_search(cv.begin(), cv.end(), k) ? 1 : 0;
    double mcyc = 0, mtick = 0, vcyc = 0, vtick = 0;
    for (int i = 0; i < 256; i++) {
      mcyc += (double)hm[i].cycles;
      mtick += (double)hm[i].ticks;
      vcyc += (double)hv[i].cycles;
      vtick += (double)hv[i].ticks;
    }
    const float t_co = median(co);
    static const char *names[3] = {"VALU only", "VALU + LDS atomics / 16-byte reads (image-kernel mix)", "LDS heavy"};
    printf("V = %s\n", names[mix]);
    printf("  alone: M %.3f ms, V %.3f ms (sum %.3f);  together on two streams: %.3f ms = %.2f x sum  (%.2f x the longer one)\n", t_m, t_v,
           t_m + t_v, t_co, t_co / (t_m + t_v), t_co / std::max(t_m, t_v));
    printf("  in the co-run: M workgroups on %zu distinct CUs, V on %zu, %d CUs hosted both; a M workgroup lasted %.3f ms (%.2f GHz), a V workgroup %.3f ms\n",
           cm.size(), cv.size(), both, mtick / 256 / 1e5, mcyc / (mtick * 10.0), vtick / 256 / 1e5);
  }
  return 0;
}
