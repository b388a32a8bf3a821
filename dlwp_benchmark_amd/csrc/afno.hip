// AFNO2D frequency-domain mixing for FourCastNet on MI355X (gfx950).
//
// Replaces reference models/fourcastnet/fourcastnet.py:87-121 -- the four full-size zero buffers, the
// slice-assigns, eight einsums, ReLU, and softshrink of AFNO2D.forward -- with ONE pass over the
// spectrum: for every frequency point the complex block-diagonal 2-layer MLP
//     o1 = relu(x W1 + b1)   (real and imaginary parts rectified separately, :96-106)
//     o2 = o1 W2 + b2        (:108-118)
//     y  = softshrink(o2)    (on re / im separately, :120-121)
// is evaluated for kept modes (rows [tm-km, tm+km), cols [0, km), tm = H/2+1, km = int(tm*frac) --
// note km along W is derived from H, :93-96) and zeros are written elsewhere, so the output is the
// complete [B, H, W/2+1, C] spectrum irfft2 expects.  (The rfft2 / irfft2 around it currently go
// through torch.fft = rocFFT)
//
// Layout: spectrum as interleaved complex [B][H][Wf][C][2]; one thread per (point, channel); the
// C channels of a point are consecutive threads, x and o1 are shared through LDS.
#include "common.hpp"

namespace dlwp {
namespace afno {

struct Params {
  const float2* x;   // [P][C]
  float2* y;         // [P][C]
  const float* w1;   // [2][nb][bs][bs]
  const float* b1;   // [2][nb][bs]
  const float* w2;   // [2][nb][bs][bs]
  const float* b2;   // [2][nb][bs]
  long long npoint;  // B*H*Wf
  int H, Wf, C, nb, bs;
  int row_lo, row_hi, col_hi;  // kept region
  float lambd;
};

__device__ __forceinline__ float softshrink(float v, float l) { return v > l ? v - l : (v < -l ? v + l : 0.f); }

__global__ __launch_bounds__(256) void afno_mix_kernel(const Params p) {
  extern __shared__ __align__(16) float smem[];
  float2* s_x = reinterpret_cast<float2*>(smem);  // [ppb][C]
  float2* s_h = s_x + blockDim.x;                 // [ppb][C]
  const int tid = threadIdx.x;
  const int C = p.C, bs = p.bs;
  const int ppb = blockDim.x / C;                 // points per block
  const int pl = tid / C, c = tid % C;
  const int blk = c / bs, o = c % bs;
  for (long long base = (long long)blockIdx.x * ppb; base < p.npoint; base += (long long)gridDim.x * ppb) {
    const long long pt = base + pl;
    const bool live = pl < ppb && pt < p.npoint;
    bool kept = false;
    if (live) {
      const int wf = (int)(pt % p.Wf);
      const int h = (int)((pt / p.Wf) % p.H);
      kept = h >= p.row_lo && h < p.row_hi && wf < p.col_hi;
    }
    __syncthreads();
    if (live && kept) s_x[tid] = p.x[pt * C + c];
    __syncthreads();
    float2 h1 = {0.f, 0.f};
    if (live && kept) {
      const float* wr = p.w1 + ((long long)blk * bs) * bs + o;
      const float* wi = wr + (long long)p.nb * bs * bs;
      float ar = p.b1[blk * bs + o], ai = p.b1[p.nb * bs + blk * bs + o];
      const float2* xb = s_x + pl * C + blk * bs;
      for (int i = 0; i < bs; ++i) {
        const float2 xv = xb[i];
        const float r = wr[i * bs], im = wi[i * bs];
        ar = fmaf(xv.x, r, fmaf(-xv.y, im, ar));
        ai = fmaf(xv.y, r, fmaf(xv.x, im, ai));
      }
      h1 = float2{fmaxf(ar, 0.f), fmaxf(ai, 0.f)};
      s_h[tid] = h1;
    }
    __syncthreads();
    if (live) {
      float2 out = {0.f, 0.f};
      if (kept) {
        const float* wr = p.w2 + ((long long)blk * bs) * bs + o;
        const float* wi = wr + (long long)p.nb * bs * bs;
        float ar = p.b2[blk * bs + o], ai = p.b2[p.nb * bs + blk * bs + o];
        const float2* hb = s_h + pl * C + blk * bs;
        for (int i = 0; i < bs; ++i) {
          const float2 hv = hb[i];
          const float r = wr[i * bs], im = wi[i * bs];
          ar = fmaf(hv.x, r, fmaf(-hv.y, im, ar));
          ai = fmaf(hv.y, r, fmaf(hv.x, im, ai));
        }
        out = float2{softshrink(ar, p.lambd), softshrink(ai, p.lambd)};
      }
      p.y[pt * C + c] = out;
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
  DLWP_REQUIRE(C <= 256, DLWP_ERR_UNSUPPORTED, "hidden size %d > 256 not supported", C);
  afno::Params p;
  p.x = reinterpret_cast<const float2*>(xf);
  p.y = reinterpret_cast<float2*>(yf);
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
  p.npoint = (long long)batch * H * Wf;
  p.H = H; p.Wf = Wf; p.C = C; p.nb = num_blocks; p.bs = C / num_blocks;
  const int total = H / 2 + 1;
  const int kept = (int)((double)total * (double)hard_thresholding_fraction);
  p.row_lo = total - kept < 0 ? 0 : total - kept;
  p.row_hi = total + kept > H ? H : total + kept;
  p.col_hi = kept > Wf ? Wf : kept;
  p.lambd = sparsity_threshold;
  const int ppb = 256 / C;
  const int threads = ppb * C;
  long long blocks = (p.npoint + ppb - 1) / ppb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(afno::afno_mix_kernel, dim3((unsigned)blocks), dim3(threads), (size_t)threads * 16,
                     reinterpret_cast<hipStream_t>(stream), p);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
