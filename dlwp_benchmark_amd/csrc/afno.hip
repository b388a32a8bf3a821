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
  float in_scale, out_scale;   // y = out_scale * mix(in_scale * x): the "ortho" factors of unnormalised transforms
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
        for (int i = 0; i < BS; ++i) {
          const float2 v = p.x[base + (long long)(blk * BS + i) * plane + jcol];
          xin[i] = float2{v.x * p.in_scale, v.y * p.in_scale};
        }
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
          out[o] = float2{softshrink(ar, p.lambd) * p.out_scale, softshrink(ai, p.lambd) * p.out_scale};
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


// ---------------------------------------------------------------------------------------------
// MFMA form for block size 16 (FourCastNet: embed 64 / 4 blocks, embed 768 / ... -> bs 16 is the BASELINE case).
// A block's complex 16 -> 16 -> 16 MLP is a real 32 -> 32 -> 32 one with W_real = [[Wr, Wi], [-Wi, Wr]]; it is
// evaluated TRANSPOSED, D[n][point] = sum_k W_real^T[n][k] x^T[k][point], on v_mfma_f32_16x16x4_f32:
//   * one wave-iteration = 16 consecutive kept points of one spectrum row (the MFMA's columns, lane & 15);
//   * B operand of layer 1 = the spectrum itself: lane (j, g) loads float2 x[point j][channel 4s + g], s = 0..3
//     (coalesced 8-byte loads) and feeds .x in k-step s, .y in k-step s + 4  (k = ri * 16 + channel);
//   * the accumulator of layer 1 (rows n = 16 nt + 4 g + r of point j) IS the B operand of layer 2 -- k-step (nt, r)
//     supplies hidden index 16 nt + 4 g + r, the weights of layer 2 are gathered in that k order;
//   * layer 2's accumulator holds re (tile 0) and im (tile 1) of output channel 4 g + r of point j in the same lane:
//     one coalesced float2 store per channel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void afno_mix_mfma16_kernel(const Params p) {
  constexpr int BS = 16;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const long long plane = (long long)p.H * p.Wf;
  // every 16-column tile of every row is visited: kept points are mixed, the rest of the row (columns >= km) and the
  // rows outside [row_lo, row_hi) are written as zeros by the same pass -- whole rows leave the kernel contiguously
  // (the separate zero-fill launch + the half-row writes of the first version: 103 + 47 us per call at C4)
  const int kt = (int)((plane + 15) / 16);               // 16-point tiles per spectrum plane
  const long long ntiles = (long long)p.B * kt;
  // wave w of the workgroup owns channel block w (w + 4, ...): its weights are gathered into registers ONCE and
  // reused for every 16-point tile the workgroup visits
  const int wave = threadIdx.x >> 6;
  for (int blk = wave; blk < p.nb; blk += 4) {
    // ---- operands: weights of this block in MFMA A layout (row n = lane & 15 of tile nt, k = 4 s + g)
    float a1[2][8], a2[2][8];
    const float* w1r = p.w1 + (long long)blk * BS * BS;
    const float* w1i = w1r + (long long)p.nb * BS * BS;
    const float* w2r = p.w2 + (long long)blk * BS * BS;
    const float* w2i = w2r + (long long)p.nb * BS * BS;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)        // output tile: 0 = real parts, 1 = imaginary parts; output channel o = j
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        {   // layer 1: k = ri * 16 + ch, ri = s >> 2, ch = 4 (s & 3) + g
          const int ri = s >> 2, ch = 4 * (s & 3) + g;
          const float wr = w1r[ch * BS + j], wi = w1i[ch * BS + j];
          a1[nt][s] = nt == 0 ? (ri == 0 ? wr : -wi) : (ri == 0 ? wi : wr);
        }
        {   // layer 2: k-step s = (nt1, r1) supplies hidden index (ro1 = nt1, o1 = 4 g + r1)
          const int ri = s >> 2, ch = 4 * g + (s & 3);
          const float wr = w2r[ch * BS + j], wi = w2i[ch * BS + j];
          a2[nt][s] = nt == 0 ? (ri == 0 ? wr : -wi) : (ri == 0 ? wi : wr);
        }
      }
    f32x4 bias1[2], bias2[2];   // bias of row n = 16 nt + 4 g + r: b[ro = nt][blk][o = 4 g + r]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float* bb1 = p.b1 + (long long)nt * p.nb * BS + blk * BS + 4 * g;
      const float* bb2 = p.b2 + (long long)nt * p.nb * BS + blk * BS + 4 * g;
      bias1[nt] = f32x4{bb1[0], bb1[1], bb1[2], bb1[3]};
      bias2[nt] = f32x4{bb2[0], bb2[1], bb2[2], bb2[3]};
    }
    // tile geometry + the tile's inputs; the NEXT tile's inputs are requested before the current tile's 32 matrix
    // instructions (a wave otherwise alternates between waiting for 512 bytes and computing on them)
    struct Tile { long long base; bool kept, live, in_row; };
    auto geom = [&](long long t) {
      // a tile = 16 CONSECUTIVE points of the flattened [H][Wf] plane: 128 aligned bytes per channel whatever Wf is (tiles
      // cut along rows put every access of a 65-column spectrum across two cache lines)
      Tile q;
      const int tile = (int)(t % kt);
      const int b = (int)(t / kt);
      const int pt = tile * 16 + j;
      const int h = pt / p.Wf, col = pt - h * p.Wf;
      q.in_row = pt < (int)plane;
      q.live = q.in_row && h >= p.row_lo && h < p.row_hi && col < p.km;
      q.kept = __any(q.live);                                       // wave-uniform
      q.base = (long long)b * p.C * plane + (q.in_row ? pt : 0);
      return q;
    };
    auto fetch = [&](const Tile& q, float2 (&v)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = q.live ? p.x[q.base + (long long)(blk * BS + 4 * s + g) * plane] : float2{0.f, 0.f};
    };
    float2 xnext[4];
    Tile qn = geom(blockIdx.x < ntiles ? blockIdx.x : 0);
    if (blockIdx.x < ntiles) fetch(qn, xnext);
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const Tile q = qn;
      const long long base = q.base;
      const bool live = q.live, in_row = q.in_row;
      float2 xin[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) xin[s] = xnext[s];
      if (t + gridDim.x < ntiles) {
        qn = geom(t + gridDim.x);
        fetch(qn, xnext);
      }
      if (!q.kept) {
        if (in_row) {
#pragma unroll
          for (int r = 0; r < 4; ++r) p.y[base + (long long)(blk * BS + 4 * g + r) * plane] = float2{0.f, 0.f};
        }
        continue;
      }
      // ---- layer 1
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        xin[s].x *= p.in_scale;
        xin[s].y *= p.in_scale;
      }
      f32x4 d1[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        d1[nt] = bias1[nt];
#pragma unroll
        for (int s = 0; s < 8; ++s) d1[nt] = mfma16x16x4(a1[nt][s], s < 4 ? xin[s].x : xin[s - 4].y, d1[nt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[nt][r] = fmaxf(d1[nt][r], 0.f);
      }
      // ---- layer 2 (B operand = layer-1 accumulator as it stands)
      f32x4 d2[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        d2[nt] = bias2[nt];
#pragma unroll
        for (int s = 0; s < 8; ++s) d2[nt] = mfma16x16x4(a2[nt][s], d1[s >> 2][s & 3], d2[nt]);
      }
      if (in_row) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          p.y[base + (long long)(blk * BS + 4 * g + r) * plane] =
              live ? float2{softshrink(d2[0][r], p.lambd) * p.out_scale, softshrink(d2[1][r], p.lambd) * p.out_scale}
                   : float2{0.f, 0.f};
      }
    }
  }
}

}  // namespace afno
}  // namespace dlwp

using namespace dlwp;

// ---------------------------------------------------------------------------------------------------------------
// Backward of the mixing (training: loss.backward() of reference scripts/train.py:271 through fourcastnet.py:96-121).
// Forward per kept point, with UNNORMALISED transforms around it:  xin = in_scale X,  o1 = relu(xin W1 + b1),
// o2 = o1 W2 + b2,  Y = out_scale softshrink(o2),  y = C2R(Y).  Given G = R2C(grad_y) (the same forward transform):
//   gz  = out_scale c_k G            (adjoint of the unnormalised C2R: c_k = 1 for the self-conjugate columns, 2 otherwise)
//   d2  = gz  where |o2| > lambda    (softshrink)                 -> dW2 = sum_p conj(o1) (x) d2,  db2 = sum_p d2
//   d1  = (d2 W2^H) where pre-activation > 0 (relu, re / im apart) -> dW1 = sum_p conj(xin) (x) d1, db1 = sum_p d1
//   gX  = in_scale (d1 W1^H) / c_k   (adjoint of the unnormalised R2C is C2R of this)
// The kernel recomputes the forward per point and writes gX plus the four per-point factors (xin, o1, d1, d2) of the weight
// gradients, zeros outside the kept rows; the sums over the points are einsums on the caller's side (rocBLAS through torch,
// like the spectral convolution's weight gradient).  One thread = one kept point, one block of BS channels at a time.
// ---------------------------------------------------------------------------------------------------------------
namespace dlwp {
namespace afno {

struct BwdParams {
  const float2* x;    // [B][C][H][Wf]  R2C(x)
  const float2* g;    // [B][C][H][Wf]  R2C(grad_y)
  float2* gx;         // [B][C][H][Wf]  -> C2R gives grad_x
  float2* xin;        // factors of the weight gradients, same layout
  float2* o1;
  float2* d1;
  float2* d2;
  const float* w1; const float* b1; const float* w2; const float* b2;
  int B, H, Wf, C, nb;
  int row_lo, row_hi, km;
  int nyq;            // column index of the Nyquist column (W / 2 for even W), or -1
  float lambd, in_scale, out_scale;
};

template <int BS>
__global__ __launch_bounds__(256) void afno_mix_bwd_kernel(const BwdParams p) {
  const long long plane = (long long)p.H * p.Wf;
  const long long npts = (long long)p.B * p.H * p.Wf;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npts; t += (long long)gridDim.x * blockDim.x) {
    const int jcol = (int)(t % p.Wf);
    const int h = (int)((t / p.Wf) % p.H);
    const int b = (int)(t / ((long long)p.Wf * p.H));
    const bool kept = h >= p.row_lo && h < p.row_hi && jcol < p.km;
    const long long base = (long long)b * p.C * plane + (long long)h * p.Wf + jcol;
    const float ck = (jcol == 0 || jcol == p.nyq) ? 1.0f : 2.0f;
    for (int blk = 0; blk < p.nb; ++blk) {
      float2 xin[BS], h1[BS], e2[BS], e1[BS], gxo[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) { xin[i] = h1[i] = e2[i] = e1[i] = gxo[i] = float2{0.f, 0.f}; }
      if (kept) {
        const float* w1r = p.w1 + (long long)blk * BS * BS;
        const float* w1i = w1r + (long long)p.nb * BS * BS;
        const float* w2r = p.w2 + (long long)blk * BS * BS;
        const float* w2i = w2r + (long long)p.nb * BS * BS;
        bool m1r[BS], m1i[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          const float2 v = p.x[base + (long long)(blk * BS + i) * plane];
          xin[i] = float2{v.x * p.in_scale, v.y * p.in_scale};
        }
#pragma unroll
        for (int o = 0; o < BS; ++o) {
          float ar = p.b1[blk * BS + o], ai = p.b1[p.nb * BS + blk * BS + o];
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const float r = w1r[i * BS + o], im = w1i[i * BS + o];
            ar = fmaf(xin[i].x, r, fmaf(-xin[i].y, im, ar));
            ai = fmaf(xin[i].y, r, fmaf(xin[i].x, im, ai));
          }
          m1r[o] = ar > 0.f;
          m1i[o] = ai > 0.f;
          h1[o] = float2{fmaxf(ar, 0.f), fmaxf(ai, 0.f)};
        }
#pragma unroll
        for (int o = 0; o < BS; ++o) {
          float ar = p.b2[blk * BS + o], ai = p.b2[p.nb * BS + blk * BS + o];
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const float r = w2r[i * BS + o], im = w2i[i * BS + o];
            ar = fmaf(h1[i].x, r, fmaf(-h1[i].y, im, ar));
            ai = fmaf(h1[i].y, r, fmaf(h1[i].x, im, ai));
          }
          const float2 gv = p.g[base + (long long)(blk * BS + o) * plane];
          const float sc = p.out_scale * ck;
          e2[o] = float2{fabsf(ar) > p.lambd ? gv.x * sc : 0.f, fabsf(ai) > p.lambd ? gv.y * sc : 0.f};
        }
        // d1 = d2 W2^H: o2r = h1r w2r - h1i w2i, o2i = h1i w2r + h1r w2i  ->  dh1r = d2r w2r + d2i w2i, dh1i = -d2r w2i + d2i w2r
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          float ar = 0.f, ai = 0.f;
#pragma unroll
          for (int o = 0; o < BS; ++o) {
            const float r = w2r[i * BS + o], im = w2i[i * BS + o];
            ar = fmaf(e2[o].x, r, fmaf(e2[o].y, im, ar));
            ai = fmaf(e2[o].y, r, fmaf(-e2[o].x, im, ai));
          }
          e1[i] = float2{m1r[i] ? ar : 0.f, m1i[i] ? ai : 0.f};
        }
        const float sx = p.in_scale / ck;
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          float ar = 0.f, ai = 0.f;
#pragma unroll
          for (int o = 0; o < BS; ++o) {
            const float r = w1r[i * BS + o], im = w1i[i * BS + o];
            ar = fmaf(e1[o].x, r, fmaf(e1[o].y, im, ar));
            ai = fmaf(e1[o].y, r, fmaf(-e1[o].x, im, ai));
          }
          gxo[i] = float2{ar * sx, ai * sx};
        }
      }
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        const long long o = base + (long long)(blk * BS + i) * plane;
        p.gx[o] = gxo[i];
        p.xin[o] = xin[i];
        p.o1[o] = h1[i];
        p.d1[o] = e1[i];
        p.d2[o] = e2[i];
      }
    }
  }
}

}  // namespace afno
}  // namespace dlwp

extern "C" int32_t dlwp_afno2d_mix_bwd_f32(const float* xf, const float* gf, float* gxf, float* xin, float* o1, float* d1, float* d2,
                                           const float* w1, const float* b1, const float* w2, const float* b2, int32_t batch,
                                           int32_t H, int32_t Wf, int32_t C, int32_t num_blocks, int32_t width,
                                           float sparsity_threshold, float hard_thresholding_fraction, float in_scale,
                                           float out_scale, void* stream) {
  DLWP_REQUIRE(xf && gf && gxf && xin && o1 && d1 && d2 && w1 && b1 && w2 && b2, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && H > 0 && Wf > 0 && C > 0 && num_blocks > 0 && C % num_blocks == 0 && width > 0, DLWP_ERR_INVALID_ARGUMENT,
               "bad shape");
  afno::BwdParams p;
  p.x = reinterpret_cast<const float2*>(xf); p.g = reinterpret_cast<const float2*>(gf);
  p.gx = reinterpret_cast<float2*>(gxf); p.xin = reinterpret_cast<float2*>(xin); p.o1 = reinterpret_cast<float2*>(o1);
  p.d1 = reinterpret_cast<float2*>(d1); p.d2 = reinterpret_cast<float2*>(d2);
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
  p.B = batch; p.H = H; p.Wf = Wf; p.C = C; p.nb = num_blocks;
  const int bs = C / num_blocks;
  const int total = H / 2 + 1;
  const int kept = (int)((double)total * (double)hard_thresholding_fraction);
  DLWP_REQUIRE(kept >= 1, DLWP_ERR_INVALID_ARGUMENT, "hard_thresholding_fraction keeps no mode");
  p.row_lo = total - kept < 0 ? 0 : total - kept;
  p.row_hi = total + kept > H ? H : total + kept;
  p.km = kept > Wf ? Wf : kept;
  p.nyq = (width % 2 == 0) ? width / 2 : -1;
  p.lambd = sparsity_threshold; p.in_scale = in_scale; p.out_scale = out_scale;
  const long long npts = (long long)batch * H * Wf;
  long long blocks = (npts + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (bs) {
    case 4: hipLaunchKernelGGL(afno::afno_mix_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL(afno::afno_mix_bwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 16: hipLaunchKernelGGL(afno::afno_mix_bwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    default: return fail(DLWP_ERR_UNSUPPORTED, "AFNO backward: block size %d not supported (4, 8, 16)", bs);
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_afno2d_mix_scaled_f32(const float* xf, float* yf, const float* w1, const float* b1,
                                              const float* w2, const float* b2, int32_t batch, int32_t H, int32_t Wf,
                                              int32_t C, int32_t num_blocks, float sparsity_threshold,
                                              float hard_thresholding_fraction, float in_scale, float out_scale,
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
  p.in_scale = in_scale;
  p.out_scale = out_scale;
  const long long npts = (long long)batch * H * p.km;
  long long blocks = (npts + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (bs) {
    case 4: hipLaunchKernelGGL(afno::afno_mix_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL(afno::afno_mix_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    case 16: {
      // MFMA form: kept points on the matrix lanes, zeros elsewhere from the same pass
      const long long tiles = (long long)batch * (((long long)H * Wf + 15) / 16);
      long long wg = tiles;
      if (wg > 256 * 8) wg = 256 * 8;
      hipLaunchKernelGGL(afno::afno_mix_mfma16_kernel, dim3((unsigned)(wg > 0 ? wg : 1)), dim3(256), 0, s, p);
      break;
    }
    case 32: hipLaunchKernelGGL(afno::afno_mix_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
    default: return fail(DLWP_ERR_UNSUPPORTED, "AFNO block size %d not supported (4, 8, 16, 32)", bs);
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_afno2d_mix_f32(const float* xf, float* yf, const float* w1, const float* b1, const float* w2,
                                       const float* b2, int32_t batch, int32_t H, int32_t Wf, int32_t C,
                                       int32_t num_blocks, float sparsity_threshold, float hard_thresholding_fraction,
                                       void* stream) {
  return dlwp_afno2d_mix_scaled_f32(xf, yf, w1, b1, w2, b2, batch, H, Wf, C, num_blocks, sparsity_threshold,
                                    hard_thresholding_fraction, 1.0f, 1.0f, stream);
}
