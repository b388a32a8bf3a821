// AFNO2D frequency-domain mixing for FourCastNet on MI355X (gfx950).
//
// Replaces reference models/fourcastnet/fourcastnet.py:87-121 -- the four full-size zero buffers, the
// slice-assigns, eight einsums, ReLU, and softshrink of AFNO2D.forward -- with ONE pass over the
// spectrum: for every kept frequency point the complex block-diagonal 2-layer MLP
//     o1 = relu(x W1 + b1)   (real and imaginary parts rectified separately, :96-106)
//     o2 = o1 W2 + b2        (:108-118)
//     y  = softshrink(o2)    (on re / im separately, :120-121)
// is evaluated (kept = rows [tm-km, tm+km), cols [0, km), tm = H/2+1, km = int(tm*frac) -- km along W
// is derived from H, :93-96) and zeros are written everywhere else, so the output is the complete
// spectrum irfft2 expects.  The rfft2 / irfft2 around it go through torch.fft (rocFFT).
//
// Layout: CHANNELS-FIRST interleaved complex [B][C][H][Wf][2] -- that is the physical layout
// torch.fft.rfft2(x_nhwc, dim=(1,2)) produces (it transposes internally and returns a permuted view), so
// neither side needs a copy.  One thread = one kept frequency point, all C channels, one block of `BS`
// channels at a time in registers; lanes are consecutive along Wf (coalesced); the weights are
// wave-uniform (scalar loads).  The thread of kept column j also zero-fills the non-kept columns
// km + j, 2 km + j, ... of its row.
#include "common.hpp"

namespace dlwp {
namespace afno {

struct Params {
  const float2* x;   // [B][C][H][Wf]
  float2* y;         // [B][C][H][Wf]
  const float* w1;   // [2][nb][bs][bs]
  const float* b1;   // [2][nb][bs]
  const float* w2;   // [2][nb][bs][bs]
  const float* b2;   // [2][nb][bs]
  int B, H, Wf, C, nb;
  int row_lo, row_hi, km;   // kept rows [row_lo, row_hi), kept cols [0, km)
  float lambd;
};

__device__ __forceinline__ float softshrink(float v, float l) { return v > l ? v - l : (v < -l ? v + l : 0.f); }

template <int BS>
__global__ __launch_bounds__(256) void afno_mix_kernel(const Params p) {
  const long long plane = (long long)p.H * p.Wf;
  const long long npts = (long long)p.B * p.H * p.km;   // one thread per (b, h, kept column)
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npts;
       t += (long long)gridDim.x * blockDim.x) {
    const int jcol = (int)(t % p.km);
    const int h = (int)((t / p.km) % p.H);
    const int b = (int)(t / ((long long)p.km * p.H));
    const bool kept_row = h >= p.row_lo && h < p.row_hi;
    const long long base = (long long)b * p.C * plane + (long long)h * p.Wf;
    for (int blk = 0; blk < p.nb; ++blk) {
      float2 out[BS];
      if (kept_row) {
        float2 xin[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) xin[i] = p.x[base + (long long)(blk * BS + i) * plane + jcol];
        const float* w1r = p.w1 + (long long)blk * BS * BS;
        const float* w1i = w1r + (long long)p.nb * BS * BS;
        float2 h1[BS];
#pragma unroll
        for (int o = 0; o < BS; ++o) {
          float ar = p.b1[blk * BS + o], ai = p.b1[p.nb * BS + blk * BS + o];
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const float r = w1r[i * BS + o], im = w1i[i * BS + o];
            ar = fmaf(xin[i].x, r, fmaf(-xin[i].y, im, ar));
            ai = fmaf(xin[i].y, r, fmaf(xin[i].x, im, ai));
          }
          h1[o] = float2{fmaxf(ar, 0.f), fmaxf(ai, 0.f)};
        }
        const float* w2r = p.w2 + (long long)blk * BS * BS;
        const float* w2i = w2r + (long long)p.nb * BS * BS;
#pragma unroll
        for (int o = 0; o < BS; ++o) {
          float ar = p.b2[blk * BS + o], ai = p.b2[p.nb * BS + blk * BS + o];
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const float r = w2r[i * BS + o], im = w2i[i * BS + o];
            ar = fmaf(h1[i].x, r, fmaf(-h1[i].y, im, ar));
            ai = fmaf(h1[i].y, r, fmaf(h1[i].x, im, ai));
          }
          out[o] = float2{softshrink(ar, p.lambd), softshrink(ai, p.lambd)};
        }
      } else {
#pragma unroll
        for (int o = 0; o < BS; ++o) out[o] = float2{0.f, 0.f};
      }
#pragma unroll
      for (int o = 0; o < BS; ++o) {
        float2* yrow = p.y + base + (long long)(blk * BS + o) * plane;
        yrow[jcol] = out[o];
        for (int c2 = p.km + jcol; c2 < p.Wf; c2 += p.km) yrow[c2] = float2{0.f, 0.f};
      }
    }
  }
}

}  // namespace afno
}  // namespace dlwp

using namespace dlwp;

extern "C" int32_t dlwp_afno2d_mix_f32(const float* xf, float* yf, const float* w1, const float* b1, const float* w2,
                                       const float* b2, int32_t batch, int32_t H, int32_t Wf, int32_t C,
                                       int32_t num_blocks, float sparsity_threshold, float hard_thresholding_fraction,
                                       void* stream) {
  DLWP_REQUIRE(xf && yf && w1 && b1 && w2 && b2, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && H > 0 && Wf > 0 && C > 0 && num_blocks > 0 && C % num_blocks == 0, DLWP_ERR_INVALID_ARGUMENT,
               "bad shape");
  afno::Params p;
  p.x = reinterpret_cast<const float2*>(xf);
  p.y = reinterpret_cast<float2*>(yf);
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
  p.B = batch; p.H = H; p.Wf = Wf; p.C = C; p.nb = num_blocks;
  const int bs = C / num_blocks;
  const int total = H / 2 + 1;
  const int kept = (int)((double)total * (double)hard_thresholding_fraction);
  DLWP_REQUIRE(kept >= 1, DLWP_ERR_INVALID_ARGUMENT, "hard_thresholding_fraction keeps no mode");
  p.row_lo = total - kept < 0 ? 0 : total - kept;
  p.row_hi = total + kept > H ? H : total + kept;
  p.km = kept > Wf ? Wf : kept;
  p.lambd = sparsity_threshold;
  const long long npts = (long long)batch * H * p.km;
  long long blocks = (npts + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (bs) {
    case 4: hipLaunchKernelGGL(afno::afno_mix_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL(afno::afno_mix_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 16: hipLaunchKernelGGL(afno::afno_mix_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 32: hipLaunchKernelGGL(afno::afno_mix_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    default: return fail(DLWP_ERR_UNSUPPORTED, "AFNO block size %d not supported (4, 8, 16, 32)", bs);
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
