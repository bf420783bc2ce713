// Is v_mfma_f32_32x32x2_f32 the k-ordered fmaf chain the parity contract needs (as 16x16x4 is, cdna_hip_programming.md §3)?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off profiles/mfma_chain_32x32x2.hip -o /tmp/mfma_chain && /tmp/mfma_chain
// Random A (32 x 2), B (2 x 32), C (32 x 32) with wide exponent spread; D from one MFMA against
//   chain01 = fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], C))   and   chain10 (k = 1 first), bit for bit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, const float *C, float *D) {
  const int l = threadIdx.x;
  const float a = A[(l & 31) * 2 + (l >> 5)];  // A[i = l % 32][k = l / 32]
  const float b = B[(l >> 5) * 32 + (l & 31)];  // B[k][j = l % 32]
  f32x16 c;
  for (int r = 0; r < 16; r++) c[r] = C[((r >> 2) * 8 + (l >> 5) * 4 + (r & 3)) * 32 + (l & 31)];
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[((r >> 2) * 8 + (l >> 5) * 4 + (r & 3)) * 32 + (l & 31)] = c[r];
}
int main() {
  float hA[64], hB[64], hC[1024], hD[1024];
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
  srand(7);
  long same01 = 0, same10 = 0, total = 0, differ_orders = 0;
  for (int trial = 0; trial < 200; trial++) {
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX - 0.5) * std::pow(2.0, rand() % 24 - 12)); };
    for (float &v : hA) v = rnd();
    for (float &v : hB) v = rnd();
    for (float &v : hC) v = rnd();
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; i++)
      for (int j = 0; j < 32; j++) {
        const float c01 = fmaf(hA[i * 2 + 1], hB[32 + j], fmaf(hA[i * 2], hB[j], hC[i * 32 + j]));
        const float c10 = fmaf(hA[i * 2], hB[j], fmaf(hA[i * 2 + 1], hB[32 + j], hC[i * 32 + j]));
        const float d = hD[i * 32 + j];
        total++;
        same01 += std::memcmp(&d, &c01, 4) == 0;
        same10 += std::memcmp(&d, &c10, 4) == 0;
        differ_orders += std::memcmp(&c01, &c10, 4) != 0;
      }
  }
  printf("32x32x2 f32: %ld outputs, equal to the k-ascending chain %ld, to the k-descending chain %ld (the two chains differ in %ld)\n", total,
         same01, same10, differ_orders);
  return 0;
}
