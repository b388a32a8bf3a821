// On-device evaluation metrics: latitude-weighted squared-error / anomaly-product sums
// of reference scripts/evaluate.py:786-821 (`compute_metrics`: Eq. (2) / (A1) of arXiv:2002.00469) fused
// with the per-variable de-normalisation of evaluate.py:281-296 (x*std + mean: the means cancel in every
// difference, the stds become a per-variable scale).  Reduces a rollout [B, K, C, H, W] to [K, C] sums on
// the device, so a multi-GPU evaluation exchanges a few hundred bytes (all-reduce) instead of gathering
// trajectories, and nothing is copied to the host per step (evaluate.py:243 `outputs.append(output.cpu())`).
//
//   sums[0][k][c] = sum_{b,h,w} w_h * (s_c (out - tar))^2                          (RMSE numerator)
//   sums[1][k][c] = sum w_h * s_c^2 (out - clim)(tar - clim)                       (ACC numerator)
//   sums[2][k][c] = sum w_h * (s_c (out - clim))^2,  sums[3] = sum w_h (s_c (tar - clim))^2
// accumulated in fp64 (one double atomicAdd per workgroup and quantity).
#include "common.hpp"

namespace dlwp {
namespace metrics {

__global__ __launch_bounds__(256) void weighted_sums_kernel(const float* __restrict__ out, const float* __restrict__ tar,
                                                            const float* __restrict__ clim,  // [K][C][H][W] or null
                                                            const float* __restrict__ latw,  // [H]
                                                            const float* __restrict__ scale, // [C] or null
                                                            double* __restrict__ sums,       // [4][K][C]
                                                            int B, int K, int C, int H, int W, int chunks) {
  // grid: (chunks, K*C, B); each block reduces a slice of one (b, k, c) plane
  const int kc = blockIdx.y, b = blockIdx.z;
  const int c = kc % C;
  const long long HW = (long long)H * W;
  const long long plane = ((long long)b * K * C + kc) * HW;
  const long long per = (HW + chunks - 1) / chunks;
  const long long lo = blockIdx.x * per, hi = (lo + per < HW) ? lo + per : HW;
  const float s = scale ? scale[c] : 1.f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float w = latw[i / W];
    const float o = out[plane + i], t = tar[plane + i];
    const float d = s * (o - t);
    a0 = fmaf(w * d, d, a0);
    if (clim) {
      const float cl = clim[(long long)kc * HW + i];
      const float po = s * (o - cl), pt = s * (t - cl);
      a1 = fmaf(w * po, pt, a1);
      a2 = fmaf(w * po, po, a2);
      a3 = fmaf(w * pt, pt, a3);
    }
  }
  __shared__ float red[4][4];
  float v[4] = {a0, a1, a2, a3};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[q] += __shfl_xor(v[q], m);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    if (q == 0 || clim) {
      const double tot = (double)red[q][0] + (double)red[q][1] + (double)red[q][2] + (double)red[q][3];
      atomicAdd(&sums[(long long)q * K * C + kc], tot);
    }
  }
}

}  // namespace metrics
}  // namespace dlwp

using namespace dlwp;

static int32_t weighted_sums(const float* out, const float* target, const float* climatology, const float* lat_weights,
                             const float* scale, double* sums, int32_t batch, int32_t steps, int32_t channels, int32_t height,
                             int32_t width, void* stream, bool zero_first) {
  DLWP_REQUIRE(out && target && lat_weights && sums, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && steps > 0 && channels > 0 && height > 0 && width > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE((long long)steps * channels <= 65535 && batch <= 65535, DLWP_ERR_UNSUPPORTED, "grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (zero_first) DLWP_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * 4 * steps * channels, s));
  const long long HW = (long long)height * width;
  int chunks = (int)((HW + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(metrics::weighted_sums_kernel, dim3(chunks, steps * channels, batch), dim3(256), 0, s, out, target,
                     climatology, lat_weights, scale, sums, batch, steps, channels, height, width, chunks);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_weighted_error_sums_f32(const float* out, const float* target, const float* climatology,
                                                const float* lat_weights, const float* scale, double* sums,
                                                int32_t batch, int32_t steps, int32_t channels, int32_t height,
                                                int32_t width, void* stream) {
  return weighted_sums(out, target, climatology, lat_weights, scale, sums, batch, steps, channels, height, width, stream, true);
}

// the same sums ADDED to what sums_dev holds: the running sums of an evaluation (evaluate.py:786-821 accumulates squared errors over
// all batches before taking the root) without a zero-fill and an add kernel per batch
extern "C" int32_t dlwp_weighted_error_sums_acc_f32(const float* out, const float* target, const float* climatology,
                                                    const float* lat_weights, const float* scale, double* sums,
                                                    int32_t batch, int32_t steps, int32_t channels, int32_t height,
                                                    int32_t width, void* stream) {
  return weighted_sums(out, target, climatology, lat_weights, scale, sums, batch, steps, channels, height, width, stream, false);
}
