// LayerNorm over the last (channel) dimension of token-major tensors, for the token backbones
// (FourCastNet: fourcastnet.py:180-193; Swin: swin_transformer.py:213,262; Pangu: panguweather.py:281,321).
// torch's generic kernel reaches ~0.5 TB/s on the short rows these models have (C = 64..384: 1.04 ms for
// the 268 MB of one FourCastNet LayerNorm at 128x256x32); rows here are register resident: 16 / 32 / 64
// lanes per row with one or more 16-byte vectors each, mean and variance by xor-shuffles inside the
// lane group, two-pass (mean, then sum of squared deviations) like torch's CPU kernel.
#include "common.hpp"

namespace dlwp {
namespace norm {

// OB16: y is bf16 [rows][C] (rounded to nearest even) -- for a consumer that rounds its input to bf16 anyway (dlwp_linear_bf16_io)
template <int LPR, int NV, bool OB16 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ pre,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;                        // position inside the row's lane group
  constexpr int RPW = 64 / LPR;                      // rows per wave
  const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwave = ((long long)gridDim.x * blockDim.x) >> 6;
  const int nvec = C >> 2;
  f32x4 gm[NV], bt[NV], pb[NV];   // pb: optional per-channel vector added to x BEFORE the statistics (deferred biases)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int iv = sub + v * LPR;
    const bool ok = iv < nvec;
    gm[v] = ok ? *reinterpret_cast<const f32x4*>(gamma + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
    bt[v] = ok ? *reinterpret_cast<const f32x4*>(beta + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
    pb[v] = (ok && pre) ? *reinterpret_cast<const f32x4*>(pre + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_c = 1.0f / (float)C;
  for (long long r0 = wave_id * RPW; r0 < rows; r0 += nwave * RPW) {
    const long long row = r0 + lane / LPR;
    const bool live = row < rows;
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int iv = sub + v * LPR;
      xv[v] = (live && iv < nvec) ? *reinterpret_cast<const f32x4*>(x + row * C + 4 * iv) + pb[v] : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
    }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int iv = sub + v * LPR;
      if (iv < nvec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dlt = xv[v][k] - mean;
          q = fmaf(dlt, dlt, q);
        }
      }
    }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    const float rstd = rsqrtf(q * inv_c + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int iv = sub + v * LPR;
      if (live && iv < nvec) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf((xv[v][k] - mean) * rstd, gm[v][k], bt[v][k]);
        if constexpr (OB16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + row * C + 4 * iv) = uint2{cvt_pk_bf16(o[0], o[1]), cvt_pk_bf16(o[2], o[3])};
        else
          *reinterpret_cast<f32x4*>(y + row * C + 4 * iv) = o;
      }
    }
  }
}

template <int LPR, int NV>
static int32_t launch(const float* x, const float* pre, const float* g, const float* b, float* y, long long rows, int C,
                      float eps, hipStream_t s, bool ob16 = false) {
  constexpr int RPW = 64 / LPR;
  long long waves = (rows + RPW - 1) / RPW;
  long long blocks = (waves + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (ob16) hipLaunchKernelGGL((layernorm_kernel<LPR, NV, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, pre, g, b, y, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<LPR, NV>), dim3((unsigned)blocks), dim3(256), 0, s, x, pre, g, b, y, rows, C, eps);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

}  // namespace norm
}  // namespace dlwp

using namespace dlwp;

static int32_t layernorm_prebias(const float* x, const float* pre, const float* gamma, const float* beta,
                                 float* y, int64_t rows, int32_t channels, float eps, void* stream, bool ob16) {
  DLWP_REQUIRE(x && gamma && beta && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(rows > 0 && channels > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(channels % 4 == 0 && channels <= 2048, DLWP_ERR_UNSUPPORTED,
               "channels %d: must be a multiple of 4 and <= 2048", channels);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nvec = channels / 4;
  if (nvec <= 16) return norm::launch<16, 1>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
  if (nvec <= 32) return norm::launch<32, 1>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
  const int nv = (nvec + 63) / 64;
  switch (nv) {
    case 1: return norm::launch<64, 1>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
    case 2: return norm::launch<64, 2>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
    case 3: return norm::launch<64, 3>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
    case 4: return norm::launch<64, 4>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
    case 5: case 6: return norm::launch<64, 6>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
    default: return norm::launch<64, 8>(x, pre, gamma, beta, y, rows, channels, eps, s, ob16);
  }
}

extern "C" int32_t dlwp_layernorm_prebias_f32(const float* x, const float* pre, const float* gamma, const float* beta,
                                              float* y, int64_t rows, int32_t channels, float eps, void* stream) {
  return layernorm_prebias(x, pre, gamma, beta, y, rows, channels, eps, stream, false);
}

// the same LayerNorm with a bf16 result (y_bf16 [rows][channels], round to nearest even): the input of a bf16-form Linear
extern "C" int32_t dlwp_layernorm_prebias_bf16out(const float* x, const float* pre, const float* gamma, const float* beta,
                                                  void* y_bf16, int64_t rows, int32_t channels, float eps, void* stream) {
  return layernorm_prebias(x, pre, gamma, beta, reinterpret_cast<float*>(y_bf16), rows, channels, eps, stream, true);
}

extern "C" int32_t dlwp_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                                      int32_t channels, float eps, void* stream) {
  return dlwp_layernorm_prebias_f32(x, nullptr, gamma, beta, y, rows, channels, eps, stream);
}

// ---------------------------------------------------------------------------------------------
// FourCastNet block glue (fourcastnet.py:180-193 around AFNO2D, :78-127), fused with the layout change
// the FFT needs: torch.fft on a channels-last tensor transposes it with a slow strided copy on both
// sides (1.4 ms per block at 128x256x32, more than the FFTs themselves).
//   ln_nhwc_to_nchw : y[b][c][t] = LayerNorm(x[b][t][:])[c]
//   afno_merge      : s[b][t][c] = f[b][c][t] + l[b][c][t] + x[b][t][c]    (irfft2 output + AFNO2D's
//                     "+ bias" (:127) + the block's first skip (:187)),  n = LayerNorm2(s)
// Tile = 64 tokens x C channels through LDS; token-major side as 16-byte vectors (16 lanes per token),
// channel-major side as 256-byte rows.
// ---------------------------------------------------------------------------------------------
namespace dlwp {
namespace norm {

constexpr int TT = 64;  // tokens per tile

template <int NV>  // 16-byte vectors per lane, 16 lanes per token: C <= 64 * NV
__global__ __launch_bounds__(256) void ln_nhwc_to_nchw_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             int B, long long T, int C, float eps) {
  extern __shared__ __align__(16) float smem[];   // [C][TT + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, tq = lane >> 4;
  const int nvec = C >> 2;
  const long long tiles_per_b = (T + TT - 1) / TT;
  const long long ntile = (long long)B * tiles_per_b;
  const float inv_c = 1.0f / (float)C;
  f32x4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int iv = sub + 16 * v;
    gm[v] = iv < nvec ? *reinterpret_cast<const f32x4*>(gamma + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
    bt[v] = iv < nvec ? *reinterpret_cast<const f32x4*>(beta + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int b = (int)(tile / tiles_per_b);
    const long long t0 = (tile % tiles_per_b) * TT;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {   // wave handles tokens wave*16 + pass*4 + tq
      const int tl = wave * 16 + pass * 4 + tq;
      const long long tok = t0 + tl;
      const bool live = tok < T;
      f32x4 xv[NV];
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int iv = sub + 16 * v;
        xv[v] = (live && iv < nvec) ? *reinterpret_cast<const f32x4*>(x + ((long long)b * T + tok) * C + 4 * iv)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
      const float mean = s * inv_c;
      float q = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (sub + 16 * v < nvec) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dlt = xv[v][k] - mean;
            q = fmaf(dlt, dlt, q);
          }
        }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) q += __shfl_xor(q, m);
      const float rstd = rsqrtf(q * inv_c + eps);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int iv = sub + 16 * v;
        if (iv < nvec) {
#pragma unroll
          for (int k = 0; k < 4; ++k) smem[(4 * iv + k) * (TT + 1) + tl] = fmaf((xv[v][k] - mean) * rstd, gm[v][k], bt[v][k]);
        }
      }
    }
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
      const long long tok = t0 + lane;
      if (tok < T) y[((long long)b * C + c) * T + tok] = smem[c * (TT + 1) + lane];
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void afno_merge_kernel(const float* __restrict__ f, const float* __restrict__ l,
                                                        const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ sum_bias,
                                                        float* __restrict__ s_out, float* __restrict__ n_out, int B,
                                                        long long T, int C, float eps) {
  extern __shared__ __align__(16) float smem[];   // [TT][C + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, tq = lane >> 4;
  const int nvec = C >> 2, ld = C + 4;
  const long long tiles_per_b = (T + TT - 1) / TT;
  const long long ntile = (long long)B * tiles_per_b;
  const float inv_c = 1.0f / (float)C;
  f32x4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int iv = sub + 16 * v;
    gm[v] = (n_out && iv < nvec) ? *reinterpret_cast<const f32x4*>(gamma + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
    bt[v] = (n_out && iv < nvec) ? *reinterpret_cast<const f32x4*>(beta + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 sb[NV];   // optional per-channel constant added to the STORED sum only (not to the LayerNorm input)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int iv = sub + 16 * v;
    sb[v] = (sum_bias && iv < nvec) ? *reinterpret_cast<const f32x4*>(sum_bias + 4 * iv) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int b = (int)(tile / tiles_per_b);
    const long long t0 = (tile % tiles_per_b) * TT;
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
      const long long tok = t0 + lane;
      if (tok < T) {
        const long long o = ((long long)b * C + c) * T + tok;
        smem[lane * ld + c] = f[o] + l[o];
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int tl = wave * 16 + pass * 4 + tq;
      const long long tok = t0 + tl;
      const bool live = tok < T;
      f32x4 sv[NV];
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int iv = sub + 16 * v;
        if (live && iv < nvec) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((long long)b * T + tok) * C + 4 * iv);
          sv[v] = *reinterpret_cast<const f32x4*>(smem + tl * ld + 4 * iv) + xv;
          *reinterpret_cast<f32x4*>(s_out + ((long long)b * T + tok) * C + 4 * iv) = sv[v] + sb[v];
        } else {
          sv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s += (sv[v][0] + sv[v][1]) + (sv[v][2] + sv[v][3]);
      }
      if (!n_out) continue;   // caller normalises downstream (dlwp_token_mlp_f32 with ln_eps >= 0)
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
      const float mean = s * inv_c;
      float q = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (sub + 16 * v < nvec) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float dlt = sv[v][k] - mean;
            q = fmaf(dlt, dlt, q);
          }
        }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) q += __shfl_xor(q, m);
      const float rstd = rsqrtf(q * inv_c + eps);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int iv = sub + 16 * v;
        if (live && iv < nvec) {
          f32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = fmaf((sv[v][k] - mean) * rstd, gm[v][k], bt[v][k]);
          *reinterpret_cast<f32x4*>(n_out + ((long long)b * T + tok) * C + 4 * iv) = o;
        }
      }
    }
  }
}

}  // namespace norm
}  // namespace dlwp

extern "C" int32_t dlwp_layernorm_nhwc_to_nchw_f32(const float* x, const float* gamma, const float* beta, float* y,
                                                   int32_t batch, int64_t tokens, int32_t channels, float eps,
                                                   void* stream) {
  DLWP_REQUIRE(x && gamma && beta && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && tokens > 0 && channels > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(channels % 4 == 0 && channels <= 256, DLWP_ERR_UNSUPPORTED, "channels %d: multiple of 4, <= 256", channels);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long long ntile = (long long)batch * ((tokens + norm::TT - 1) / norm::TT);
  const unsigned grid = (unsigned)(ntile < 256 * 16 ? ntile : 256 * 16);
  const size_t lds = (size_t)channels * (norm::TT + 1) * 4;
  const int nv = (channels / 4 + 15) / 16;
#define DLWP_L(NV_)                                                                                                  \
  do {                                                                                                               \
    if (lds > 48 * 1024)                                                                                             \
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(norm::ln_nhwc_to_nchw_kernel<NV_>),           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
    hipLaunchKernelGGL((norm::ln_nhwc_to_nchw_kernel<NV_>), dim3(grid), dim3(256), lds, s, x, gamma, beta, y, batch, \
                       (long long)tokens, channels, eps);                                                            \
  } while (0)
  switch (nv) { case 1: DLWP_L(1); break; case 2: DLWP_L(2); break; case 3: DLWP_L(3); break; default: DLWP_L(4); break; }
#undef DLWP_L
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_afno_merge_f32(const float* f_nchw, const float* l_nchw, const float* x_nhwc, const float* gamma,
                                       const float* beta, const float* sum_bias, float* sum_nhwc, float* norm_nhwc,
                                       int32_t batch, int64_t tokens, int32_t channels, float eps, void* stream) {
  DLWP_REQUIRE(f_nchw && l_nchw && x_nhwc && sum_nhwc && (!norm_nhwc || (gamma && beta)), DLWP_ERR_INVALID_ARGUMENT,
               "null argument");
  DLWP_REQUIRE(batch > 0 && tokens > 0 && channels > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(channels % 4 == 0 && channels <= 256, DLWP_ERR_UNSUPPORTED, "channels %d: multiple of 4, <= 256", channels);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long long ntile = (long long)batch * ((tokens + norm::TT - 1) / norm::TT);
  const unsigned grid = (unsigned)(ntile < 256 * 16 ? ntile : 256 * 16);
  const size_t lds = (size_t)norm::TT * (channels + 4) * 4;
  const int nv = (channels / 4 + 15) / 16;
#define DLWP_M(NV_)                                                                                                  \
  do {                                                                                                               \
    if (lds > 48 * 1024)                                                                                             \
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(norm::afno_merge_kernel<NV_>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
    hipLaunchKernelGGL((norm::afno_merge_kernel<NV_>), dim3(grid), dim3(256), lds, s, f_nchw, l_nchw, x_nhwc, gamma,  \
                       beta, sum_bias, sum_nhwc, norm_nhwc, batch, (long long)tokens, channels, eps);                          \
  } while (0)
  switch (nv) { case 1: DLWP_M(1); break; case 2: DLWP_M(2); break; case 3: DLWP_M(3); break; default: DLWP_M(4); break; }
#undef DLWP_M
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
