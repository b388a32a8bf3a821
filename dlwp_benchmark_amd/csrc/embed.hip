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

// The other end of a 1x1-patch token backbone (reference fourcastnet.py:144 `head = nn.Linear(embed_dim, out_chans p1 p2, bias=False)`,
// applied at :296-303 and rearranged "b h w (p1 p2 c_out) -> b c_out (h p1) (w p2)"): with 1x1 patches that is
//   out[b][co][hw] = bias[co] + sum_c W[co][c] tokens[b][hw][c]
// -- a [tokens x C] x [C x 3] product whose result torch writes token-major (rocBLAS, 93 us at C4) and then permutes into a
// channels-first copy.  Here: a wave stages 64 consecutive token rows into its LDS slice with coalesced 16-byte loads, then lane =
// token walks its row against the weight rows (LDS broadcast reads) and stores channels-first, 256 contiguous bytes per wave and
// output channel.  fp32 FMAs in k order.
template <int COUT_MAX>
__global__ void patch_recover_1x1_kernel(const float* __restrict__ tok, const float* __restrict__ w,
                                         const float* __restrict__ bias, float* __restrict__ out, long long B, long long HW,
                                         int C, int cout) {
  extern __shared__ __align__(16) float smem_pr[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int ld = C + 4;                                     // row stride of the token tile (floats): 16-byte aligned rows
  float* s_w = smem_pr;                                     // [COUT_MAX][C]
  float* s_x = smem_pr + COUT_MAX * C + wv * 64 * ld;       // [64][ld] of this wave
  for (int i = threadIdx.x; i < COUT_MAX * C; i += blockDim.x) s_w[i] = i < cout * C ? w[i] : 0.f;
  __syncthreads();
  const long long total = B * HW;
  const long long wave_id = (long long)blockIdx.x * nwv + wv, nwave = (long long)gridDim.x * nwv;
  const int c4 = C >> 2;
  for (long long t0 = wave_id * 64; t0 < total; t0 += nwave * 64) {
    const long long left = total - t0;
    const int nt = left < 64 ? (int)left : 64;
    const f32x4* src = reinterpret_cast<const f32x4*>(tok + t0 * C);
    // eight 16-byte loads in flight per lane, then their LDS stores (a load-then-store loop pays one memory latency per 1 KB)
    const int nq = nt * c4;
    for (int i0 = lane; i0 < nq; i0 += 64 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        v[u] = i < nq ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        if (i < nq) {
          const int r = i / c4, q = i - r * c4;
          *reinterpret_cast<f32x4*>(s_x + r * ld + 4 * q) = v[u];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < nt) {
      float acc[COUT_MAX];
#pragma unroll
      for (int co = 0; co < COUT_MAX; ++co) acc[co] = (bias && co < cout) ? bias[co] : 0.f;
      const float* xr = s_x + lane * ld;
      for (int q = 0; q < c4; ++q) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + 4 * q);
#pragma unroll
        for (int co = 0; co < COUT_MAX; ++co) {
          const f32x4 wv4 = *reinterpret_cast<const f32x4*>(s_w + co * C + 4 * q);
          acc[co] = fmaf(wv4[0], xv[0], acc[co]);
          acc[co] = fmaf(wv4[1], xv[1], acc[co]);
          acc[co] = fmaf(wv4[2], xv[2], acc[co]);
          acc[co] = fmaf(wv4[3], xv[3], acc[co]);
        }
      }
      const long long tk = t0 + lane, b = tk / HW, hw = tk - b * HW;
#pragma unroll
      for (int co = 0; co < COUT_MAX; ++co)
        if (co < cout) out[(b * cout + co) * HW + hw] = acc[co];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next chunk overwrites the slice
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace embed
}  // namespace dlwp

using namespace dlwp;

extern "C" int32_t dlwp_patch_recover_1x1_f32(const float* tokens_dev, const float* w_dev, const float* bias_dev, float* out_dev,
                                              int32_t batch, int64_t tokens, int32_t channels, int32_t out_channels, void* stream) {
  DLWP_REQUIRE(tokens_dev && w_dev && out_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && tokens > 0 && channels > 0 && out_channels > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(channels % 4 == 0 && channels <= 256, DLWP_ERR_UNSUPPORTED, "patch recover: channels %d (a multiple of 4, <= 256)", channels);
  DLWP_REQUIRE(out_channels <= 16, DLWP_ERR_UNSUPPORTED, "patch recover: out_channels %d > 16", out_channels);
  DLWP_REQUIRE((reinterpret_cast<uintptr_t>(tokens_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_dev) & 15) == 0,
               DLWP_ERR_INVALID_ARGUMENT, "pointers must be 16-byte aligned");
  const int waves = channels <= 128 ? 4 : 2;
  const int cmax = out_channels <= 4 ? 4 : (out_channels <= 8 ? 8 : 16);
  const size_t lds = ((size_t)cmax * channels + (size_t)waves * 64 * (channels + 4)) * 4;
  const long long total = (long long)batch * tokens;
  long long blocks = (total + 64 * waves - 1) / (64 * waves);
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define DLWP_PR(N)                                                                                                              \
  do {                                                                                                                          \
    auto kern = embed::patch_recover_1x1_kernel<N>;                                                                             \
    if (lds > 48 * 1024)                                                                                                        \
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * waves), lds, s, tokens_dev, w_dev, bias_dev, out_dev,            \
                       (long long)batch, (long long)tokens, channels, out_channels);                                            \
  } while (0)
  if (cmax == 4) DLWP_PR(4);
  else if (cmax == 8) DLWP_PR(8);
  else DLWP_PR(16);
#undef DLWP_PR
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

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

// ---------------------------------------------------------------------------------------------------------------
// `_prepare_inputs` of the rollout loop (reference swin_transformer.py:679-692 and its copies): cat([constants[:, 0], prescribed window,
// prognostic window], dim 1) -- up to 8 channel segments, each a [B][Ci][HW] block with its own batch stride (views into the inputs and
// into the trajectory buffer), into one contiguous [B][sum Ci][HW] tensor.  torch.cat's generic kernel takes 17.6 us for the 2 MB of a
// Swin step (26.5 us for FourCastNet's 33 MB); this is a plain 16-byte copy.
// ---------------------------------------------------------------------------------------------------------------
namespace dlwp {
namespace embed {
struct ConcatSegs {
  const float* ptr[8];
  long long bstride[8];      // floats between samples of the segment
  int c0[9];                 // first destination channel of segment i; c0[n] = total channels
  int n;
};
__global__ __launch_bounds__(256) void concat_channels_kernel(const ConcatSegs S, float* __restrict__ dst, long long HW4, int B) {
  // grid.y = destination channel, grid.z = sample, grid.x strides over the plane in float4
  const int c = blockIdx.y, b = blockIdx.z;
  int seg = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < S.n && c >= S.c0[i]) seg = i;
  const f32x4* src = reinterpret_cast<const f32x4*>(S.ptr[seg] + (long long)b * S.bstride[seg]) + (long long)(c - S.c0[seg]) * HW4;
  f32x4* out = reinterpret_cast<f32x4*>(dst) + ((long long)b * S.c0[S.n] + c) * HW4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW4; i += (long long)gridDim.x * 256) out[i] = src[i];
}
}  // namespace embed
}  // namespace dlwp

extern "C" int32_t dlwp_concat_channels_f32(const float* const* seg_dev_ptrs, const int32_t* seg_channels, const int64_t* seg_batch_strides,
                                            int32_t n_segments, float* out_dev, int32_t batch, int64_t plane, void* stream) {
  DLWP_REQUIRE(seg_dev_ptrs && seg_channels && seg_batch_strides && out_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(n_segments >= 1 && n_segments <= 8, DLWP_ERR_UNSUPPORTED, "concat: %d segments (1..8)", n_segments);
  DLWP_REQUIRE(batch > 0 && batch <= 65535 && plane > 0 && plane % 4 == 0, DLWP_ERR_UNSUPPORTED,
               "concat: batch %d, plane %lld (a multiple of 4 floats)", batch, (long long)plane);
  embed::ConcatSegs S = {};
  S.n = n_segments;
  int tot = 0;
  for (int i = 0; i < n_segments; ++i) {
    DLWP_REQUIRE(seg_dev_ptrs[i] && seg_channels[i] > 0, DLWP_ERR_INVALID_ARGUMENT, "concat: segment %d", i);
    DLWP_REQUIRE((reinterpret_cast<uintptr_t>(seg_dev_ptrs[i]) & 15) == 0 && seg_batch_strides[i] % 4 == 0, DLWP_ERR_INVALID_ARGUMENT,
                 "concat: segment %d is not 16-byte aligned", i);
    S.ptr[i] = seg_dev_ptrs[i];
    S.bstride[i] = seg_batch_strides[i];
    S.c0[i] = tot;
    tot += seg_channels[i];
  }
  S.c0[n_segments] = tot;
  DLWP_REQUIRE(tot <= 65535 && (reinterpret_cast<uintptr_t>(out_dev) & 15) == 0, DLWP_ERR_UNSUPPORTED, "concat: %d channels / unaligned output", tot);
  const long long hw4 = plane / 4;
  int gx = (int)((hw4 + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(embed::concat_channels_kernel, dim3(gx, tot, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, out_dev,
                     hw4, batch);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
