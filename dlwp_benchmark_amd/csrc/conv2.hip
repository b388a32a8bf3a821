// The rest of the U-Net / ModernUNet operator family (reference models/unet/unet.py): GroupNorm (+ activation), strided and
// 1x1 convolutions, transposed convolutions and 2x2 average pooling as hand-written kernels, so that a step of those
// backbones never leaves libdlwp_hip (round 1 ran them through nn.GroupNorm / MIOpen).
//   GroupNorm         unet.py:739 (final_norm, 8 groups), :887-888 (ResidualBlock, n_groups = 1) -- two-sweep statistics
//                     per (sample, group) in one workgroup, affine + activation fused into the third sweep
//   Conv2d k x k, s   unet.py:583 (3x3, stride 2, zero padding 1), :584 (1x1), :879-881 (1x1 shortcut), :450 / :533 (1x1 head)
//   ConvTranspose2d   unet.py:719 (4x4, stride 2, padding 1), :523 (2x2, stride 2)
//   AvgPool2d(2)      unet.py:450
// These layers work on maps of 2x2 ... 64x64 with 8 ... 1024 channels: each launch is a few microseconds and bound by
// latency, so the kernels are direct and simple -- one thread per output element, lanes along the map's fastest axis
// (coalesced), weights through the scalar / L1 path (they are wave-uniform per output channel).
#include "common.hpp"

namespace dlwp {
namespace conv2 {

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case 1: return gelu_erf(v);
    case 2: return tanhf(v);
    case 3: return fmaxf(v, 0.f);
    case 4: return v / (1.f + __expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float block_sum(float v, float* s_red, int tid) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}

// one workgroup per (sample, group): elements [n][g * cpg .. (g + 1) * cpg)[HW] are contiguous in NCHW
__global__ __launch_bounds__(256) void groupnorm_act_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int C,
                                                            int HW, int groups, float eps, int act) {
  __shared__ float s_red[8];
  const int tid = threadIdx.x;
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const int E = cpg * HW;
  float s = 0.f;
  for (int i = tid; i < E; i += 256) s += x[base + i];
  const float mean = block_sum(s, s_red, tid) / (float)E;
  float q = 0.f;
  for (int i = tid; i < E; i += 256) {
    const float dlt = x[base + i] - mean;
    q += dlt * dlt;
  }
  const float var = block_sum(q, s_red, tid) / (float)E;      // biased, like torch.nn.GroupNorm
  const float rstd = rsqrtf(var + eps);
  for (int i = tid; i < E; i += 256) {
    const int c = g * cpg + i / HW;
    float v = (x[base + i] - mean) * rstd;
    v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
    y[base + i] = apply_act(v, act);
  }
}

struct ConvP {
  const float* x;      // [N][Cin][H][W]
  const float* w;      // conv: [Cout][Cin][K][K]; transposed: [Cin][Cout][K][K]
  const float* bias;   // [Cout] or null
  const float* resid;  // [N][Cout][OH][OW] or null
  float* y;            // [N][Cout][OH][OW]
  int N, Cin, H, W, Cout, K, stride, pad, OH, OW, act, pre_act;
};

// zero-padded direct convolution, any K / stride: thread = (n, co, oh, ow), ow fastest
__global__ __launch_bounds__(256) void conv2d_kernel(const ConvP p) {
  const long long total = (long long)p.N * p.Cout * p.OH * p.OW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % p.OW);
    long long r = i / p.OW;
    const int oh = (int)(r % p.OH);
    r /= p.OH;
    const int co = (int)(r % p.Cout);
    const int n = (int)(r / p.Cout);
    float acc = p.bias ? p.bias[co] : 0.f;
    const float* wr = p.w + (long long)co * p.Cin * p.K * p.K;
    const float* xn = p.x + (long long)n * p.Cin * p.H * p.W;
    for (int ci = 0; ci < p.Cin; ++ci) {
      const float* xc = xn + (long long)ci * p.H * p.W;
      for (int kh = 0; kh < p.K; ++kh) {
        const int ih = oh * p.stride - p.pad + kh;
        if (ih < 0 || ih >= p.H) continue;
        for (int kw = 0; kw < p.K; ++kw) {
          const int iw = ow * p.stride - p.pad + kw;
          if (iw < 0 || iw >= p.W) continue;
          float v = xc[(long long)ih * p.W + iw];
          if (p.pre_act) v = apply_act(v, p.pre_act);
          acc = fmaf(v, wr[(ci * p.K + kh) * p.K + kw], acc);
        }
      }
    }
    if (p.resid) acc += p.resid[i];
    p.y[i] = apply_act(acc, p.act);
  }
}

// transposed convolution (torch.nn.ConvTranspose2d semantics, output_padding 0, dilation 1):
//   y[n][co][oh][ow] = bias[co] + sum_{ci, kh, kw : oh = ih * s - pad + kh, ow = iw * s - pad + kw} x[n][ci][ih][iw] w[ci][co][kh][kw]
__global__ __launch_bounds__(256) void conv_transpose2d_kernel(const ConvP p) {
  const long long total = (long long)p.N * p.Cout * p.OH * p.OW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % p.OW);
    long long r = i / p.OW;
    const int oh = (int)(r % p.OH);
    r /= p.OH;
    const int co = (int)(r % p.Cout);
    const int n = (int)(r / p.Cout);
    float acc = p.bias ? p.bias[co] : 0.f;
    const float* xn = p.x + (long long)n * p.Cin * p.H * p.W;
    for (int kh = 0; kh < p.K; ++kh) {
      const int th = oh + p.pad - kh;
      if (th < 0 || th % p.stride) continue;
      const int ih = th / p.stride;
      if (ih >= p.H) continue;
      for (int kw = 0; kw < p.K; ++kw) {
        const int tw = ow + p.pad - kw;
        if (tw < 0 || tw % p.stride) continue;
        const int iw = tw / p.stride;
        if (iw >= p.W) continue;
        const float* xp = xn + (long long)ih * p.W + iw;
        const float* wp = p.w + ((long long)co * p.K + kh) * p.K + kw;
        for (int ci = 0; ci < p.Cin; ++ci)
          acc = fmaf(xp[(long long)ci * p.H * p.W], wp[(long long)ci * p.Cout * p.K * p.K], acc);
      }
    }
    p.y[i] = apply_act(acc, p.act);
  }
}

__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes,
                                                         int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long long total = planes * OH * OW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % OW);
    const long long r = i / OW;
    const int oh = (int)(r % OH);
    const long long pl = r / OH;
    const float* xp = x + pl * H * W + (long long)(2 * oh) * W + 2 * ow;
    y[i] = 0.25f * ((xp[0] + xp[1]) + (xp[W] + xp[W + 1]));
  }
}

}  // namespace conv2
}  // namespace dlwp

using namespace dlwp;

static unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)(b > 0 ? b : 1);
}

extern "C" int32_t dlwp_groupnorm_act_f32(const float* x, const float* gamma, const float* beta, float* y, int32_t batch,
                                          int32_t channels, int32_t hw, int32_t groups, float eps, int32_t act, void* stream) {
  DLWP_REQUIRE(x && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && channels > 0 && hw > 0 && groups > 0 && channels % groups == 0, DLWP_ERR_INVALID_ARGUMENT,
               "bad shape: %d channels in %d groups", channels, groups);
  DLWP_REQUIRE(act >= 0 && act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation %d", act);
  DLWP_REQUIRE((long long)(channels / groups) * hw < (1ll << 31), DLWP_ERR_UNSUPPORTED, "group too large");
  hipLaunchKernelGGL(conv2::groupnorm_act_kernel, dim3((unsigned)(batch * groups)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, gamma, beta, y, channels, hw, groups, eps, act);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_conv2d_f32(const float* x, const float* weight, const float* bias, const float* resid, float* y,
                                   int32_t batch, int32_t cin, int32_t H, int32_t W, int32_t cout, int32_t k, int32_t stride,
                                   int32_t pad, int32_t pre_act, int32_t act, void* stream) {
  DLWP_REQUIRE(x && weight && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && cin > 0 && cout > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(act >= 0 && act <= 4 && pre_act >= 0 && pre_act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation");
  conv2::ConvP p;
  p.x = x; p.w = weight; p.bias = bias; p.resid = resid; p.y = y;
  p.N = batch; p.Cin = cin; p.H = H; p.W = W; p.Cout = cout; p.K = k; p.stride = stride; p.pad = pad;
  p.OH = (H + 2 * pad - k) / stride + 1; p.OW = (W + 2 * pad - k) / stride + 1; p.act = act; p.pre_act = pre_act;
  DLWP_REQUIRE(p.OH > 0 && p.OW > 0, DLWP_ERR_INVALID_ARGUMENT, "empty output");
  hipLaunchKernelGGL(conv2::conv2d_kernel, dim3(grid_for((long long)batch * cout * p.OH * p.OW)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_conv_transpose2d_f32(const float* x, const float* weight, const float* bias, float* y, int32_t batch,
                                             int32_t cin, int32_t H, int32_t W, int32_t cout, int32_t k, int32_t stride,
                                             int32_t pad, int32_t act, void* stream) {
  DLWP_REQUIRE(x && weight && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && cin > 0 && cout > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(act >= 0 && act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation %d", act);
  conv2::ConvP p;
  p.x = x; p.w = weight; p.bias = bias; p.resid = nullptr; p.y = y;
  p.N = batch; p.Cin = cin; p.H = H; p.W = W; p.Cout = cout; p.K = k; p.stride = stride; p.pad = pad;
  p.OH = (H - 1) * stride - 2 * pad + k; p.OW = (W - 1) * stride - 2 * pad + k; p.act = act; p.pre_act = 0;
  DLWP_REQUIRE(p.OH > 0 && p.OW > 0, DLWP_ERR_INVALID_ARGUMENT, "empty output");
  hipLaunchKernelGGL(conv2::conv_transpose2d_kernel, dim3(grid_for((long long)batch * cout * p.OH * p.OW)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_avgpool2x2_f32(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
  DLWP_REQUIRE(x && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(planes > 0 && H >= 2 && W >= 2, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  hipLaunchKernelGGL(conv2::avgpool2x2_kernel, dim3(grid_for(planes * (H / 2) * (W / 2))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long long)planes, H, W);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
