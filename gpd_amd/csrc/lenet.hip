// LeNet forward for batches of 60x60xC u8 grasp images on gfx950.
//
// Replaces EigenClassifier::forward (net/eigen_classifier.cpp:81-128), ConvLayer
// (net/conv_layer.cpp:26-98: im2col + GEMM, batch 1) and DenseLayer
// (net/dense_layer.cpp:6-15: GEMV re-streaming 14.4 MB of FC1 weights per image).
//
// Numerics contract (matches oracle/gpd_oracle.cpp): every dot product is an f32
// fmaf chain in ascending k (k = c*25 + kh*5 + kw for the convolutions, k = flatten
// index for the dense layers) starting from 0, bias added last.  The f32-input MFMA
// used for FC1 is bitwise such a chain (cdna_hip_programming.md §3).
//
// Kernels:
//   conv1_pool   planar u8 image -> LDS -> direct 5x5 conv, weights as
//                wave-uniform scalars, 2x2 max-pool fused        -> pool1 [n][20][28][28]
//   conv2_pool   pool1 plane set in LDS (62.7 KB) -> direct conv + pool, output in the
//                reference's flatten order j = pixel*50 + channel -> flat  [n][7200]
//   fc1_mfma     [500 x 7200] x [7200 x n] on v_mfma_f32_32x32x2_f32, + bias, ReLU
//                                                               -> fc1t  [500][n]
//   fc2_score    2 x 500 chains per image, score = y1 - y0       -> scores[n]
#include "gpd_internal.h"

namespace gpd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// conv1 + pool1.  One workgroup per image, wave w <-> filters 5w..5w+4.
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void conv1_pool_kernel(const uint8_t *__restrict__ images, const float *__restrict__ w,
                                                         const float *__restrict__ b, float *__restrict__ out, int n) {
  __shared__ __attribute__((aligned(16))) uint8_t s_in[C * kPix];  // planar [c][y][x]
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  // planar u8 image [C][60][60] -> LDS, 16 bytes per lane
  const uint4 *src = reinterpret_cast<const uint4 *>(images + (size_t)img * kPix * C);
  uint4 *dst = reinterpret_cast<uint4 *>(s_in);
  for (int i = tid; i < kPix * C / 16; i += 256) dst[i] = src[i];
  __syncthreads();
  const int fg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const float *__restrict__ wf = w + (size_t)fg * 5 * C * 25;
  for (int chunk = 0; chunk < 13; chunk++) {
    const int p = chunk * 64 + lane;
    const bool act = p < 784;
    const int pp = act ? p : 0;
    const int py = pp / 28, px = pp - py * 28;
    float acc[5][4];
#pragma unroll
    for (int f = 0; f < 5; f++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[f][q] = 0.f;
    const uint8_t *base = s_in + (2 * py) * kImg + 2 * px;
    for (int c = 0; c < C; c++) {
      float patch[6][6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          uint16_t v = *reinterpret_cast<const uint16_t *>(base + c * kPix + r * kImg + 2 * q);
          patch[r][2 * q] = (float)(v & 0xff);
          patch[r][2 * q + 1] = (float)(v >> 8);
        }
      }
#pragma unroll
      for (int kh = 0; kh < 5; kh++) {
#pragma unroll
        for (int kw = 0; kw < 5; kw++) {
#pragma unroll
          for (int f = 0; f < 5; f++) {
            const float wv = wf[(f * C + c) * 25 + kh * 5 + kw];
            acc[f][0] = __builtin_fmaf(wv, patch[kh][kw], acc[f][0]);
            acc[f][1] = __builtin_fmaf(wv, patch[kh][kw + 1], acc[f][1]);
            acc[f][2] = __builtin_fmaf(wv, patch[kh + 1][kw], acc[f][2]);
            acc[f][3] = __builtin_fmaf(wv, patch[kh + 1][kw + 1], acc[f][3]);
          }
        }
      }
    }
    if (act) {
#pragma unroll
      for (int f = 0; f < 5; f++) {
        // max(a_i + b) == max(a_i) + b: rounding is monotone
        float m = fmaxf(fmaxf(acc[f][0], acc[f][1]), fmaxf(acc[f][2], acc[f][3])) + b[fg * 5 + f];
        out[((size_t)img * 20 + fg * 5 + f) * 784 + p] = m;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// conv2 + pool2.  One workgroup per image; a wave task = (filter group of 5,
// 64-pixel chunk of the 144 pooled pixels); 30 tasks over 4 waves.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv2_pool_kernel(const float *__restrict__ pool1, const float *__restrict__ w,
                                                         const float *__restrict__ b, float *__restrict__ flat, int n) {
  __shared__ __attribute__((aligned(16))) float s_in[20 * 784];
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const float4 *src = reinterpret_cast<const float4 *>(pool1 + (size_t)img * 20 * 784);
  float4 *dst = reinterpret_cast<float4 *>(s_in);
  for (int i = tid; i < 20 * 784 / 4; i += 256) dst[i] = src[i];
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  for (int task = wv; task < 30; task += 4) {
    const int fg = task / 3, chunk = task - fg * 3;
    const int p = chunk * 64 + lane;
    const bool act = p < 144;
    const int pp = act ? p : 0;
    const int py = pp / 12, px = pp - py * 12;
    const float *__restrict__ wf = w + (size_t)fg * 5 * 500;
    float acc[5][4];
#pragma unroll
    for (int f = 0; f < 5; f++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[f][q] = 0.f;
    const float *base = s_in + (2 * py) * 28 + 2 * px;
    for (int c = 0; c < 20; c++) {
      float patch[6][6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          float2 v = *reinterpret_cast<const float2 *>(base + c * 784 + r * 28 + 2 * q);
          patch[r][2 * q] = v.x;
          patch[r][2 * q + 1] = v.y;
        }
      }
#pragma unroll
      for (int kh = 0; kh < 5; kh++) {
#pragma unroll
        for (int kw = 0; kw < 5; kw++) {
#pragma unroll
          for (int f = 0; f < 5; f++) {
            const float wvv = wf[f * 500 + c * 25 + kh * 5 + kw];
            acc[f][0] = __builtin_fmaf(wvv, patch[kh][kw], acc[f][0]);
            acc[f][1] = __builtin_fmaf(wvv, patch[kh][kw + 1], acc[f][1]);
            acc[f][2] = __builtin_fmaf(wvv, patch[kh + 1][kw], acc[f][2]);
            acc[f][3] = __builtin_fmaf(wvv, patch[kh + 1][kw + 1], acc[f][3]);
          }
        }
      }
    }
    if (act) {
#pragma unroll
      for (int f = 0; f < 5; f++) {
        float m = fmaxf(fmaxf(acc[f][0], acc[f][1]), fmaxf(acc[f][2], acc[f][3])) + b[fg * 5 + f];
        // flatten: j = pixel*50 + channel (eigen_classifier.cpp:103-107)
        flat[(size_t)img * kFc1In + p * 50 + fg * 5 + f] = m;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// FC1 on f32 MFMA:  D[u][m] = sum_k W[k][u] * X[m][k]   (W = ip1 weights, column-
// major 500x7200 == row-major [7200][500]; X = flat).  Block tile 64(u) x 64(m),
// 4 waves each one 32x32 tile, K stepped by 16 through LDS.
// A operand (lane l): W[k0 + (l>>5)][u0 + (l&31)],  B operand: X[m0 + (l&31)][k0 + (l>>5)].
// D layout: col(m) = lane&31, row(u) = (r&3) + 8*(r>>2) + 4*(lane>>5).
// ---------------------------------------------------------------------------
constexpr int FC_BU = 64, FC_BM = 64, FC_BK = 16;

__global__ __launch_bounds__(256) void fc1_mfma_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                       const float *__restrict__ X, float *__restrict__ out_t, int n, int ld_out) {
  __shared__ __attribute__((aligned(16))) float s_w[FC_BK][FC_BU];
  __shared__ __attribute__((aligned(16))) float s_x[FC_BK][FC_BM + 1];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * FC_BU;
  const int m0 = blockIdx.y * FC_BM;
  const int wave = tid >> 6, lane = tid & 63;
  const int wu = (wave & 1) * 32, wm = (wave >> 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  // loader roles
  const int lw_k = tid >> 4, lw_u = (tid & 15) * 4;  // 16 rows x 16 float4
  const int lx_m = tid >> 2, lx_k = (tid & 3) * 4;   // 64 rows x 4 float4
  const int xm = min(m0 + lx_m, n - 1);
  for (int k0 = 0; k0 < kFc1In; k0 += FC_BK) {
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u0 + lw_u < kFc1Out) wv = *reinterpret_cast<const float4 *>(W + (size_t)(k0 + lw_k) * kFc1Out + u0 + lw_u);
    float4 xv = *reinterpret_cast<const float4 *>(X + (size_t)xm * kFc1In + k0 + lx_k);
    __syncthreads();
    *reinterpret_cast<float4 *>(&s_w[lw_k][lw_u]) = wv;
    s_x[lx_k + 0][lx_m] = xv.x;
    s_x[lx_k + 1][lx_m] = xv.y;
    s_x[lx_k + 2][lx_m] = xv.z;
    s_x[lx_k + 3][lx_m] = xv.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FC_BK; kk += 2) {
      const float a = s_w[kk + (lane >> 5)][wu + (lane & 31)];
      const float bb = s_x[kk + (lane >> 5)][wm + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
  }
  const int m = m0 + wm + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int u = u0 + wu + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (u < kFc1Out && m < n) out_t[(size_t)u * ld_out + m] = fmaxf(acc[r] + bias[u], 0.f);
  }
}

// FC2 + score.  One thread per image; reads of fc1t are coalesced over images.
__global__ __launch_bounds__(256) void fc2_score_kernel(const float *__restrict__ fc1t, const float *__restrict__ w,
                                                        const float *__restrict__ b, float *__restrict__ scores, int n, int ld) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  float y0 = 0.f, y1 = 0.f;
  for (int j = 0; j < kFc1Out; j++) {
    const float x = fc1t[(size_t)j * ld + m];
    y0 = __builtin_fmaf(w[2 * j], x, y0);
    y1 = __builtin_fmaf(w[2 * j + 1], x, y1);
  }
  y0 += b[0];
  y1 += b[1];
  scores[m] = y1 - y0;
}

// ---------------------------------------------------------------------------
hipError_t lenet_scratch_reserve(LeNetScratch &s, int n) {
  if (n <= s.capacity) return hipSuccess;
  lenet_scratch_free(s);
  hipError_t e;
  if ((e = hipMalloc(&s.pool1, (size_t)n * 20 * 784 * sizeof(float))) != hipSuccess) return e;
  if ((e = hipMalloc(&s.flat, (size_t)n * kFc1In * sizeof(float))) != hipSuccess) return e;
  if ((e = hipMalloc(&s.fc1t, (size_t)n * kFc1Out * sizeof(float))) != hipSuccess) return e;
  s.capacity = n;
  return hipSuccess;
}

void lenet_scratch_free(LeNetScratch &s) {
  if (s.pool1) (void)hipFree(s.pool1);
  if (s.flat) (void)hipFree(s.flat);
  if (s.fc1t) (void)hipFree(s.fc1t);
  s = LeNetScratch();
}

hipError_t lenet_forward(const LeNetWeights &w, LeNetScratch &s, const uint8_t *d_images, int n, float *d_scores,
                         hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int kChunk = 16384;
  hipError_t e = lenet_scratch_reserve(s, n < kChunk ? n : kChunk);
  if (e != hipSuccess) return e;
  for (int off = 0; off < n; off += kChunk) {
    const int m = (n - off < kChunk) ? (n - off) : kChunk;
    const uint8_t *img = d_images + (size_t)off * kPix * w.channels;
    switch (w.channels) {
      case 15: conv1_pool_kernel<15><<<m, 256, 0, stream>>>(img, w.c1w, w.c1b, s.pool1, m); break;
      case 12: conv1_pool_kernel<12><<<m, 256, 0, stream>>>(img, w.c1w, w.c1b, s.pool1, m); break;
      case 3: conv1_pool_kernel<3><<<m, 256, 0, stream>>>(img, w.c1w, w.c1b, s.pool1, m); break;
      default: return hipErrorInvalidValue;
    }
    conv2_pool_kernel<<<m, 256, 0, stream>>>(s.pool1, w.c2w, w.c2b, s.flat, m);
    dim3 g((kFc1Out + FC_BU - 1) / FC_BU, (m + FC_BM - 1) / FC_BM);
    fc1_mfma_kernel<<<g, 256, 0, stream>>>(w.f1w, w.f1b, s.flat, s.fc1t, m, s.capacity);
    fc2_score_kernel<<<(m + 255) / 256, 256, 0, stream>>>(s.fc1t, w.f2w, w.f2b, d_scores + off, m, s.capacity);
  }
  return hipGetLastError();
}

}  // namespace gpd
