// Instruction-rate microbenchmark behind DESIGN.md §5 "what the f32 matrix pipe sustains" (MI355X, gfx950).
//   hipcc --offload-arch=gfx950 -O2 profiles/mfma_rates.hip -o /tmp/mfma_rates && /tmp/mfma_rates
// Part 1 (one workgroup of 8 waves; waves 0 and 4 share SIMD 0): shader cycles of 4096 back-to-back
// instructions of one kind in wave 0, alone and beside a second stream in wave 4 — does VALU work
// hide under another wave's MFMAs, and what does a VALU instruction between two MFMAs of the same
// wave cost.
// Part 2 (whole chip, 2 waves per SIMD and more): sustained TFLOP/s of pure f32 MFMA streams per
// tile shape, with the shader clock measured inside the kernel (s_memtime against the 100 MHz
// s_memrealtime).
// modes:  1 v_cvt_f32_ubyteN   2 v_cvt_f32_u32   3 v_fma_f32   4 mfma 16x16x1 (4 blocks)
//         5 mfma 4x4x1 (16 blocks)   6 modes 4 and 5 alternating   7 integer VALU
//         8 4 x ds_read_u16 + wait   9 ds_read_b128 + wait   10 mfma 16x16x1 with one convert after each
//         11 mfma 16x16x4   12 mfma 32x32x2   99 clock calibration
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// 16 iterations x 64 repetitions x 4 instructions = 4096 instructions (modes 8, 9: 1024 groups)
__device__ __forceinline__ long long run(int mode, float *sink, uint32_t seed) {
  float a = seed * 1.0f, b = 2.0f;
  uint32_t u = seed | 0x01020304u;
  float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
  f32x16 A0, A1, A2, A3;
  f32x4 B0 = {0, 0, 0, 0}, B1 = B0, B2 = B0, B3 = B0;
  for (int i = 0; i < 16; i++) { A0[i] = 0; A1[i] = 0; A2[i] = 0; A3[i] = 0; }
  __builtin_amdgcn_s_barrier();
  if (mode == 99) {  // shader cycles per 1 ms of the 100 MHz real-time counter
    const long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    while (__builtin_amdgcn_s_memrealtime() - r0 < 100000) {}
    return __builtin_readcyclecounter() - c0;
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 16; it++) {
    if (mode == 1) { REP64(asm volatile("v_cvt_f32_ubyte0_e32 %0, %4\n v_cvt_f32_ubyte1_e32 %1, %4\n v_cvt_f32_ubyte2_e32 %2, %4\n v_cvt_f32_ubyte3_e32 %3, %4" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u));) }
    else if (mode == 2) { REP64(asm volatile("v_cvt_f32_u32_e32 %0, %4\n v_cvt_f32_u32_e32 %1, %4\n v_cvt_f32_u32_e32 %2, %4\n v_cvt_f32_u32_e32 %3, %4" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u));) }
    else if (mode == 3) { REP64(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));) }
    else if (mode == 4) { REP64(A0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A1, 0, 0, 0); A2 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A2, 0, 0, 0); A3 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A3, 0, 0, 0);) }
    else if (mode == 5) { REP64(B0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B0, 0, 0, 0); B1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B1, 0, 0, 0); B2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B2, 0, 0, 0); B3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B3, 0, 0, 0);) }
    else if (mode == 6) { REP64(A0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A0, 0, 0, 0); B0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A1, 0, 0, 0); B1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, B1, 0, 0, 0);) }
    else if (mode == 7) { REP64(asm volatile("v_and_b32 %0, 0xff, %4\n v_lshrrev_b32 %1, 8, %4\n v_or_b32 %2, %4, %4\n v_add_u32 %3, %4, %4" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u));) }
    else if (mode == 8) { REP64(asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:2\n ds_read_u16 %2, %4 offset:60\n ds_read_u16 %3, %4 offset:62\n s_waitcnt lgkmcnt(0)" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u & 0xffe));) }
    else if (mode == 9) { REP64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(B0) : "v"(u & 0xff0));) }
    else if (mode == 10) { REP64(A0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A0, 0, 0, 0); asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f0) : "v"(u)); A1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A1, 0, 0, 0); asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f1) : "v"(u)); A2 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A2, 0, 0, 0); asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f2) : "v"(u)); A3 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, A3, 0, 0, 0); asm volatile("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f3) : "v"(u));) }
    else if (mode == 11) { REP64(B0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, B0, 0, 0, 0); B1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, B1, 0, 0, 0); B2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, B2, 0, 0, 0); B3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, B3, 0, 0, 0);) }
    else if (mode == 12) { REP64(A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A1, 0, 0, 0); A2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A2, 0, 0, 0); A3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A3, 0, 0, 0);) }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = f0 + f1 + f2 + f3 + B0[0] + B1[1] + B2[2] + B3[3];
  for (int i = 0; i < 16; i++) s += A0[i] + A1[i] + A2[i] + A3[i];
  sink[threadIdx.x] = s;
  return t1 - t0;
}

// one workgroup: wave 0 runs m0, wave 4 (same SIMD) runs m4, the others nothing
__global__ void pair_kernel(int m0, int m4, float *sink, long long *out) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const int mode = wave == 0 ? m0 : (wave == 4 ? m4 : 0);
  const long long dt = run(mode, sink, threadIdx.x);
  if ((threadIdx.x & 63) == 0) out[wave] = dt;
}

// every wave of every workgroup runs the same stream
__global__ void chip_kernel(int mode, float *sink, long long *out) {
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long dt = run(mode, sink, threadIdx.x);
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = dt;
    out[1] = r1 - r0;
  }
}

int main() {
  float *sink;
  long long *out;
  hipMalloc(&sink, 4096);
  hipMalloc(&out, 64);
  long long h[8];
  hipLaunchKernelGGL(pair_kernel, 1, 512, 0, 0, 99, 0, sink, out);
  hipDeviceSynchronize();
  hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("shader cycles per ms (idle chip): %lld\n", h[0]);
  const int tests[][2] = {{1, 0}, {2, 0}, {3, 0}, {7, 0}, {8, 0}, {9, 0}, {4, 0}, {5, 0}, {6, 0}, {11, 0}, {12, 0}, {10, 0},
                          {4, 1}, {4, 3}, {4, 7}, {4, 8}, {4, 4}, {4, 5}, {5, 5}, {11, 11}, {3, 3}, {1, 1}};
  for (auto &t : tests) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(pair_kernel, 1, 512, 0, 0, t[0], t[1], sink, out);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    if (t[1])
      printf("wave 0 mode %2d beside wave 4 mode %2d : %8lld / %8lld cycles\n", t[0], t[1], h[0], h[4]);
    else
      printf("wave 0 mode %2d alone                  : %8lld cycles\n", t[0], h[0]);
  }
  for (int mode : {11, 12, 4, 5, 6})
    for (int blocks : {256, 1024, 4096}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(chip_kernel, blocks, 512, 0, 0, mode, sink, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(chip_kernel, blocks, 512, 0, 0, mode, sink, out);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double macs = mode == 4 || mode == 11 ? 1024.0 : mode == 5 ? 256.0 : mode == 12 ? 2048.0 : 640.0;
      const double fl = (double)blocks * 8 * 4096 * macs * 2;
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("chip mode %2d, %4d workgroups x 8 waves: %.3f ms  %6.1f TFLOP/s   (workgroup 0: %lld cycles in %.1f us = %.2f GHz)\n", mode,
             blocks, ms, fl / ms / 1e9, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3);
    }
  return 0;
}
