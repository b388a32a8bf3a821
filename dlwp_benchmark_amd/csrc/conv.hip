// Cylinder-padded 3x3 convolution stack pieces for the U-Net / ConvLSTM backbones (gfx950).
//
// Replaces, per layer, `CylinderPad(1)` + `Conv2d(k=3, padding=0)` + activation
// (reference utils/utils.py:11-26; models/unet/unet.py:456-470, :512-525; models/convlstm/convlstm.py:47-55,
// :148-157) -- and, for the skip connections and the ConvLSTM cell, the preceding `torch.cat`
// (unet.py:553; convlstm.py:94): the input may be given as TWO channel segments.
// Direct convolution: one workgroup = one output tile (8x32, 16x16 or 8x8 by map width) of one sample and one
// 16-channel output chunk; the (TH+2)x(TW+2) input halo
// tile of a chunk of input channels is staged in LDS with the cylinder rule applied at load time
// (longitude wraps, latitude pads with zeros), every thread owns one pixel and accumulates a chunk of
// output channels with weights broadcast from LDS; bias + activation fused in the epilogue.
#include "common.hpp"

namespace dlwp {
namespace conv {

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2, ACT_RELU = 3, ACT_SILU = 4 };

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return gelu_erf(v);
    case ACT_TANH: return tanhf(v);
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SILU: return v / (1.f + __expf(-v));
    default: return v;
  }
}

constexpr int CI_CHUNK = 8;

struct Params {
  const float* x0; int c0;   // first input segment [B][c0][H][W]
  const float* x1; int c1;   // second segment or null
  const float* w;            // [Cout][c0+c1][3][3]
  const float* bias;         // [Cout] or null
  float* y;                  // [B][Cout][H][W]
  int B, H, W, Cout, act;
  int pre_act;               // activation applied to the INPUT while it is staged (pre-activation residual blocks, unet.py:886)
  const float* resid;        // [B][Cout][H][W] or null: added after bias, before `act` (block shortcut, unet.py:901)
  // HEALPix topology (null = cylinder): [12][(H+2)*(W+2)] entries (a, b) for the halo ring of every face;
  // a, b = face*H*W + pixel inside the same sample, b < 0 -> single source, else the mean of both
  // (the synthesised corners of the equatorial faces, reference utils/healpix.py:316-368).
  const int2* hpx;
};

// TH x TW output tile per workgroup (TH*TW threads); blockIdx.z selects the 16-channel output chunk, so the deep,
// low-resolution layers of a U-Net (8x8 maps with 64 channels) still spread over enough workgroups.
// CO_CHUNK output channels per thread: 16 for big maps; 4 when the map is small, so that a layer still spreads over
// >= 512 workgroups (an 8x8 map with 64 channels took 117 us as 128 one-wave workgroups of 16 channels each).
template <int TH, int TW, int CO_CHUNK>
__global__ __launch_bounds__(TH * TW) void conv3x3_cyl_kernel(const Params p) {
  constexpr int NT = TH * TW;
  __shared__ float s_in[CI_CHUNK][TH + 2][TW + 2];
  __shared__ float s_w[CO_CHUNK][CI_CHUNK][9];
  const int tid = threadIdx.x;
  const int tx = tid % TW, ty = tid / TW;
  const int tiles_w = (p.W + TW - 1) / TW;
  const int w0 = (blockIdx.x % tiles_w) * TW, h0 = (blockIdx.x / tiles_w) * TH;
  const int b = blockIdx.y;
  const int cin = p.c0 + p.c1;
  const int ow = w0 + tx, oh = h0 + ty;
  const long long HW = (long long)p.H * p.W;
  {
    const int co0 = blockIdx.z * CO_CHUNK;
    float acc[CO_CHUNK];
#pragma unroll
    for (int k = 0; k < CO_CHUNK; ++k) acc[k] = 0.f;
    for (int ci0 = 0; ci0 < cin; ci0 += CI_CHUNK) {
      __syncthreads();
      for (int i = tid; i < CI_CHUNK * (TH + 2) * (TW + 2); i += NT) {
        const int ci = i / ((TH + 2) * (TW + 2));
        const int rem = i % ((TH + 2) * (TW + 2));
        const int r = rem / (TW + 2), cc = rem % (TW + 2);
        const int c = ci0 + ci;
        const int ih = h0 + r - 1;
        int iw = w0 + cc - 1;
        float v = 0.f;
        bool corner_done = false;
        if (p.hpx) {
          if (c < cin && ih >= -1 && ih <= p.H && iw >= -1 && iw <= p.W) {
            const bool seg0 = c < p.c0;
            const float* base = seg0 ? p.x0 : p.x1;
            const int cs = seg0 ? p.c0 : p.c1, cl = seg0 ? c : c - p.c0;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
              v = base[((long long)b * cs + cl) * HW + (long long)ih * p.W + iw];
            } else {
              const int face = b % 12, s0 = b - face;
              const int2 e = p.hpx[(long long)face * (p.H + 2) * (p.W + 2) + (ih + 1) * (p.W + 2) + (iw + 1)];
              const int fa = e.x / (int)HW;
              v = base[((long long)(s0 + fa) * cs + cl) * HW + (e.x - fa * (int)HW)];
              if (e.y >= 0) {
                // a synthesised corner is the mean of two cells of the ACTIVATED tensor (the reference pads after the
                // activation, unet.py:886-887): activate each source, then average; flag the value as done
                const int fb = e.y / (int)HW;
                const float v2 = base[((long long)(s0 + fb) * cs + cl) * HW + (e.y - fb * (int)HW)];
                if (p.pre_act) { v = 0.5f * apply_act(v, p.pre_act) + 0.5f * apply_act(v2, p.pre_act); corner_done = true; }
                else v = 0.5f * v + 0.5f * v2;
              }
            }
          }
        } else if (c < cin && ih >= 0 && ih < p.H && iw >= -1 && iw <= p.W) {
          iw = iw < 0 ? iw + p.W : (iw >= p.W ? iw - p.W : iw);   // circular longitude
          const float* src = c < p.c0 ? p.x0 + ((long long)b * p.c0 + c) * HW
                                      : p.x1 + ((long long)b * p.c1 + (c - p.c0)) * HW;
          v = src[(long long)ih * p.W + iw];
        }
        (&s_in[0][0][0])[i] = (p.pre_act && !corner_done) ? apply_act(v, p.pre_act) : v;   // padding zeros stay zero (act(0) = 0)
      }
      for (int i = tid; i < CO_CHUNK * CI_CHUNK * 9; i += NT) {
        const int k = i / (CI_CHUNK * 9), rem = i % (CI_CHUNK * 9);
        const int ci = rem / 9, t = rem % 9;
        const int co = co0 + k, c = ci0 + ci;
        (&s_w[0][0][0])[i] = (co < p.Cout && c < cin) ? p.w[((long long)co * cin + c) * 9 + t] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int ci = 0; ci < CI_CHUNK; ++ci) {
        float v[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) v[r * 3 + cc] = s_in[ci][ty + r][tx + cc];
#pragma unroll
        for (int k = 0; k < CO_CHUNK; ++k)
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[k] = fmaf(v[t], s_w[k][ci][t], acc[k]);
      }
    }
    if (ow < p.W && oh < p.H) {
#pragma unroll
      for (int k = 0; k < CO_CHUNK; ++k) {
        const int co = co0 + k;
        if (co < p.Cout) {
          float v = acc[k] + (p.bias ? p.bias[co] : 0.f);
          const long long o = ((long long)b * p.Cout + co) * HW + (long long)oh * p.W + ow;
          if (p.resid) v += p.resid[o];
          p.y[o] = apply_act(v, p.act);
        }
      }
    }
  }
}

// HEALPixPadding(p) as a table-driven gather (reference utils/healpix.py:165-368): y [(B*12)][C][H+2p][W+2p],
// table [12][(H+2p)*(W+2p)] of (a, b) sources as above (interior cells map to themselves).
__global__ __launch_bounds__(256) void healpix_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const int2* __restrict__ table, int C, int HW, int PHW,
                                                          long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cell = (int)(i % PHW);
    const long long fc = i / PHW;
    const int c = (int)(fc % C);
    const long long n = fc / C;
    const int face = (int)(n % 12);
    const long long s0 = n - face;
    const int2 e = table[(long long)face * PHW + cell];
    const int fa = e.x / HW;
    float v = x[((s0 + fa) * C + c) * HW + (e.x - fa * HW)];
    if (e.y >= 0) {
      const int fb = e.y / HW;
      v = 0.5f * v + 0.5f * x[((s0 + fb) * C + c) * HW + (e.y - fb * HW)];
    }
    y[i] = v;
  }
}

// ConvLSTM cell gate math (convlstm.py:96-109): gates [B][4*hid][H][W] = (netin, igate, fgate, ogate)
__global__ __launch_bounds__(256) void convlstm_gates_kernel(const float* __restrict__ gates,
                                                             const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                             float* __restrict__ c_out, int hid, long long HW,
                                                             long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i % HW;
    const long long bc = i / HW;
    const int ch = (int)(bc % hid);
    const long long b = bc / hid;
    const float* gb = gates + (b * 4 * hid) * HW + pix;
    const float netin = gb[(long long)ch * HW];
    const float ig = gb[(long long)(hid + ch) * HW];
    const float fg = gb[(long long)(2 * hid + ch) * HW];
    const float og = gb[(long long)(3 * hid + ch) * HW];
    const float sig_i = 1.f / (1.f + __expf(-ig)), sig_f = 1.f / (1.f + __expf(-fg)), sig_o = 1.f / (1.f + __expf(-og));
    const float c = sig_f * c_prev[i] + sig_i * tanhf(netin);
    c_out[i] = c;
    h_out[i] = sig_o * tanhf(c);
  }
}

}  // namespace conv
}  // namespace dlwp

using namespace dlwp;

template <int CO>
static void launch_conv3x3_co(const conv::Params& p, hipStream_t s) {
  const int zc = (p.Cout + CO - 1) / CO;
  auto tiles = [&](int th, int tw) { return ((p.W + tw - 1) / tw) * ((p.H + th - 1) / th); };
  if (p.W >= 32)
    hipLaunchKernelGGL((conv::conv3x3_cyl_kernel<8, 32, CO>), dim3(tiles(8, 32), p.B, zc), dim3(256), 0, s, p);
  else if (p.W >= 16)
    hipLaunchKernelGGL((conv::conv3x3_cyl_kernel<16, 16, CO>), dim3(tiles(16, 16), p.B, zc), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv::conv3x3_cyl_kernel<8, 8, CO>), dim3(tiles(8, 8), p.B, zc), dim3(64), 0, s, p);
}
static void launch_conv3x3(const conv::Params& p, hipStream_t s) {
  const int th = p.W >= 32 ? 8 : (p.W >= 16 ? 16 : 8), tw = p.W >= 32 ? 32 : (p.W >= 16 ? 16 : 8);
  const long long wgs16 = (long long)((p.W + tw - 1) / tw) * ((p.H + th - 1) / th) * p.B * ((p.Cout + 15) / 16);
  // waves in flight with 4 output channels per thread: below ~8 per CU a layer is a chain of exposed load latencies (the 8 x 8
  // and 16 x 16 levels of a U-Net at B = 32: 512 / 1024 waves on 256 CUs) -- one channel per thread quadruples the workgroups
  const long long waves4 = (long long)((p.W + tw - 1) / tw) * ((p.H + th - 1) / th) * p.B * ((p.Cout + 3) / 4) * (th * tw / 64);
  if (wgs16 >= 1024) launch_conv3x3_co<16>(p, s);
  else if (waves4 >= 2048) launch_conv3x3_co<4>(p, s);
  else launch_conv3x3_co<1>(p, s);
}

extern "C" int32_t dlwp_conv3x3_ex_f32(const float* x0, int32_t c0, const float* x1, int32_t c1, const float* weight,
                                       const float* bias, const float* resid, float* y, int32_t batch, int32_t H, int32_t W,
                                       int32_t cout, int32_t pre_act, int32_t act, const int32_t* ring_table, void* stream) {
  DLWP_REQUIRE(x0 && weight && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && H > 0 && W > 0 && c0 > 0 && cout > 0 && c1 >= 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(c1 == 0 || x1, DLWP_ERR_INVALID_ARGUMENT, "second segment pointer missing");
  DLWP_REQUIRE(act >= 0 && act <= 4 && pre_act >= 0 && pre_act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation");
  DLWP_REQUIRE(batch <= 65535, DLWP_ERR_UNSUPPORTED, "batch %d exceeds the grid's y dimension", batch);
  if (ring_table) {
    DLWP_REQUIRE(batch % 12 == 0, DLWP_ERR_INVALID_ARGUMENT, "n_faces=%d is not a multiple of 12", batch);
    DLWP_REQUIRE((long long)12 * H * W < (1ll << 31), DLWP_ERR_INVALID_ARGUMENT, "face too large for the 32-bit table");
  } else {
    DLWP_REQUIRE(W > 1, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  }
  conv::Params p;
  p.x0 = x0; p.c0 = c0; p.x1 = x1; p.c1 = c1; p.w = weight; p.bias = bias; p.y = y;
  p.B = batch; p.H = H; p.W = W; p.Cout = cout; p.act = act; p.pre_act = pre_act; p.resid = resid;
  p.hpx = reinterpret_cast<const int2*>(ring_table);
  launch_conv3x3(p, reinterpret_cast<hipStream_t>(stream));
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_conv3x3_cyl_f32(const float* x0, int32_t c0, const float* x1, int32_t c1, const float* weight,
                                        const float* bias, float* y, int32_t batch, int32_t H, int32_t W, int32_t cout,
                                        int32_t act, void* stream) {
  DLWP_REQUIRE(x0 && weight && y, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && H > 0 && W > 1 && c0 > 0 && cout > 0 && c1 >= 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE(c1 == 0 || x1, DLWP_ERR_INVALID_ARGUMENT, "second segment pointer missing");
  DLWP_REQUIRE(act >= 0 && act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation %d", act);
  conv::Params p;
  p.x0 = x0; p.c0 = c0; p.x1 = x1; p.c1 = c1; p.w = weight; p.bias = bias; p.y = y;
  p.B = batch; p.H = H; p.W = W; p.Cout = cout; p.act = act; p.hpx = nullptr; p.pre_act = 0; p.resid = nullptr;
  DLWP_REQUIRE(batch <= 65535, DLWP_ERR_UNSUPPORTED, "batch %d exceeds the grid's y dimension", batch);
  launch_conv3x3(p, reinterpret_cast<hipStream_t>(stream));
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_conv3x3_hpx_f32(const float* x0, int32_t c0, const float* x1, int32_t c1, const float* weight,
                                        const float* bias, float* y, int32_t n_faces, int32_t H, int32_t W, int32_t cout,
                                        int32_t act, const int32_t* ring_table, void* stream) {
  DLWP_REQUIRE(x0 && weight && y && ring_table, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(n_faces > 0 && n_faces % 12 == 0, DLWP_ERR_INVALID_ARGUMENT, "n_faces=%d is not a multiple of 12", n_faces);
  DLWP_REQUIRE(H > 0 && W > 0 && c0 > 0 && cout > 0 && c1 >= 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE((long long)12 * H * W < (1ll << 31), DLWP_ERR_INVALID_ARGUMENT, "face too large for the 32-bit table");
  DLWP_REQUIRE(c1 == 0 || x1, DLWP_ERR_INVALID_ARGUMENT, "second segment pointer missing");
  DLWP_REQUIRE(act >= 0 && act <= 4, DLWP_ERR_INVALID_ARGUMENT, "unknown activation %d", act);
  conv::Params p;
  p.x0 = x0; p.c0 = c0; p.x1 = x1; p.c1 = c1; p.w = weight; p.bias = bias; p.y = y;
  p.B = n_faces; p.H = H; p.W = W; p.Cout = cout; p.act = act; p.hpx = reinterpret_cast<const int2*>(ring_table);
  p.pre_act = 0; p.resid = nullptr;
  DLWP_REQUIRE(n_faces <= 65535, DLWP_ERR_UNSUPPORTED, "n_faces %d exceeds the grid's y dimension", n_faces);
  launch_conv3x3(p, reinterpret_cast<hipStream_t>(stream));
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_healpix_pad_f32(const float* x, float* y, const int32_t* table, int32_t n_faces, int32_t channels,
                                        int32_t H, int32_t W, int32_t pad, void* stream) {
  DLWP_REQUIRE(x && y && table, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(n_faces > 0 && n_faces % 12 == 0, DLWP_ERR_INVALID_ARGUMENT, "n_faces=%d is not a multiple of 12", n_faces);
  DLWP_REQUIRE(channels > 0 && H > 0 && W > 0 && pad > 0 && pad <= H && pad <= W, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  DLWP_REQUIRE((long long)12 * (H + 2 * pad) * (W + 2 * pad) < (1ll << 31), DLWP_ERR_INVALID_ARGUMENT, "face too large");
  const int PHW = (H + 2 * pad) * (W + 2 * pad);
  const long long total = (long long)n_faces * channels * PHW;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(conv::healpix_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, y, reinterpret_cast<const int2*>(table), channels, H * W, PHW, total);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_convlstm_gates_f32(const float* gates, const float* c_prev, float* h_out, float* c_out,
                                           int32_t batch, int32_t hidden, int32_t H, int32_t W, void* stream) {
  DLWP_REQUIRE(gates && c_prev && h_out && c_out, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && hidden > 0 && H > 0 && W > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  const long long HW = (long long)H * W, total = (long long)batch * hidden * HW;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(conv::convlstm_gates_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), gates, c_prev, h_out, c_out, hidden, HW, total);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
