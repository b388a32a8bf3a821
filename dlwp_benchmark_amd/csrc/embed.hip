// Patch embedding of the token backbones for 1x1 patches (reference fourcastnet.py:530-543 `PatchEmbed`:
// Conv2d(kernel = stride = patch) -> flatten(2).transpose(1, 2); + pos_embed at :286-288):
//   tokens[b][hw][c] = bias[c] + pos[hw][c] + sum_ci W[c][ci] x[b][ci][hw]
// Through torch this is a MIOpen convolution that writes channels-first, a transposed VIEW of it, a broadcast add whose
// result inherits the permuted strides, and then a full-size .contiguous() in front of every kernel that wants
// token-major data: 1.3 ms per step at 32 x 128 x 256 tokens x 64 channels for 1 GFLOP of arithmetic.  Here: one
// pass, token-major 16-byte stores, the input planes read once (L1 serves the 64/C-fold reuse).
#include "common.hpp"

namespace dlwp {
namespace embed {

template <int CIN_MAX>
__global__ __launch_bounds__(256) void patch_embed_1x1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ pos, float* __restrict__ out,
                                                             long long B, int cin, long long HW, int C) {
  // A wave takes 64 CONSECUTIVE tokens at a time: their input values are read coalesced (one float per lane and input
  // channel) into the wave's LDS slice, then lpt = C / 4 lanes per token (4 output channels each) walk the 64 tokens
  // tpw at a time with broadcast LDS reads.  (The first version let every lane fetch its token's inputs from global memory:
  // one load instruction per input channel for 16 useful bytes -- 127 us per call at C4 for 300 MB of traffic.)
  __shared__ float s_x[4][CIN_MAX][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lpt = C >> 2, tpw = 64 / lpt;          // lanes per token, tokens per pass (64 % lpt lanes idle)
  const int tl = lane / lpt;
  const bool active = tl < tpw;
  const int sub = active ? lane % lpt : 0;
  float wr[CIN_MAX][4];
#pragma unroll
  for (int ci = 0; ci < CIN_MAX; ++ci)
#pragma unroll
    for (int k = 0; k < 4; ++k) wr[ci][k] = ci < cin ? w[(long long)(4 * sub + k) * cin + ci] : 0.f;
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * sub) : f32x4{0.f, 0.f, 0.f, 0.f};
  const long long total = B * HW;
  const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwave = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long t0 = wave_id * 64; t0 < total; t0 += nwave * 64) {
    {
      const long long tk = t0 + lane < total ? t0 + lane : total - 1;
      const long long b = tk / HW, hw = tk - b * HW;
#pragma unroll
      for (int ci = 0; ci < CIN_MAX; ++ci) s_x[wv][ci][lane] = ci < cin ? x[(b * cin + ci) * HW + hw] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes (no other wave touches this slice)
    __builtin_amdgcn_wave_barrier();
    for (int t1 = 0; t1 < 64; t1 += tpw) {
      const int ti = t1 + tl;
      const long long tok = t0 + ti;
      if (active && ti < 64 && tok < total) {
        const long long hw = tok % HW;
        f32x4 acc = pos ? *reinterpret_cast<const f32x4*>(pos + hw * C + 4 * sub) + bv : bv;
#pragma unroll
        for (int ci = 0; ci < CIN_MAX; ++ci) {
          const float xv = s_x[wv][ci][ti];
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaf(wr[ci][k], xv, acc[k]);
        }
        *reinterpret_cast<f32x4*>(out + tok * C + 4 * sub) = acc;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next chunk overwrites the slice
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace embed
}  // namespace dlwp

using namespace dlwp;

extern "C" int32_t dlwp_patch_embed_1x1_f32(const float* x_dev, const float* w_dev, const float* bias_dev,
                                            const float* pos_dev, float* out_dev, int32_t batch, int32_t in_channels,
                                            int64_t tokens, int32_t channels, void* stream) {
  DLWP_REQUIRE(x_dev && w_dev && out_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && in_channels > 0 && tokens > 0 && channels > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(in_channels <= 32, DLWP_ERR_UNSUPPORTED, "patch embed: in_channels %d > 32", in_channels);
  DLWP_REQUIRE(channels >= 4 && channels <= 256 && channels % 4 == 0, DLWP_ERR_UNSUPPORTED,
               "patch embed: channels %d (a multiple of 4 in [4, 256])", channels);
  const long long total = (long long)batch * tokens;
  long long blocks = (total + 255) / 256;          // 4 waves x 64 tokens per block pass
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DLWP_PE(N)                                                                                                   \
  hipLaunchKernelGGL((embed::patch_embed_1x1_kernel<N>), dim3((unsigned)blocks), dim3(256), 0, s, x_dev, w_dev, bias_dev, \
                     pos_dev, out_dev, (long long)batch, in_channels, (long long)tokens, channels)
  if (in_channels <= 8) DLWP_PE(8);
  else if (in_channels <= 16) DLWP_PE(16);
  else DLWP_PE(32);
#undef DLWP_PE
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
