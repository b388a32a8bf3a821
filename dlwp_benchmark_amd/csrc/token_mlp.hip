// Token MLP of the AFNO block (reference fourcastnet.py:41-57 `Mlp`, called at :191-192):
//   out[t][:] = resid[t][:] + b2 + W2 * gelu(W1 * n[t][:] + b1),   n / resid / out token-major [T][C], C = 64.
// torch runs this as GEMM -> GELU -> GEMM with the [T][4C] hidden activation making four trips through HBM
// (1 GB each at 128x256x32 tokens).  Here one wave owns 32 tokens at a time and the hidden activation never
// leaves its registers: both layers run on the bf16 matrix pipe as bf16x6 (every fp32 operand split exactly into
// three bf16 parts, six cross products accumulated in fp32 -- fp32-GEMM accuracy, see common.hpp), the GELU
// output of layer 1 IS the B operand of layer 2 in accumulator order (hidden tiles are taken in pairs so that
// the 8 registers a lane holds are the 8 k-slots it supplies, as in the FNO lifting kernel).
//
// Operand layout (v_mfma_f32_16x16x32_bf16: lane (j = l & 15, g = l >> 4) supplies k-slots 8g..8g+7 of row/column j
// and receives rows 4g..4g+3 of column j):
//   layer 1: A = W1 tile t (hidden 16t + j), k-step ks: channel 32 ks + 8 g + e;  B = token (q, j) = base + 16 q + j
//   layer 2: A = W2 tile ot (out channel 16 ot + j), pair u: k-slot (g, e) = hidden 16 (2u + e/4) + 4 g + e%4
// Weights: the h and m parts of both layers live in LDS (C = 64, hid = 256: 128 KB), the l parts (used by one of the
// six products) are read from global memory -- 192 KB of A operands do not fit the CU's 160 KB.
#include "common.hpp"

namespace dlwp {
namespace tmlp {

struct Params {
  const float* n;
  const float* resid;   // nullable
  const float* b2;      // nullable
  float* out;
  const u32x4* w1hm;    // [hid/16][KS][2][64]
  const u32x4* w1l;     // [hid/16][KS][64]
  const u32x4* w2hm;    // [hid/32][2][OT][64]
  const u32x4* w2l;     // [hid/32][OT][64]
  const float* b1;      // [hid]
  long long T;
  int hid;
  float ln_eps;         // LN variant: n is the un-normalised token, LayerNorm (affine folded into W1 / b1) happens here
  // NEXT variant: additionally emit LayerNorm(out) with its own affine parameters CHANNELS-FIRST, [T / HW][C][HW] --
  // what the next AFNO block's norm1 + layout change would compute from `out` in a separate pass
  float* next_cf;
  const float* next_gamma;
  const float* next_beta;
  float next_eps;
  long long HW;         // tokens per sample (multiple of 32)
  // MERGE variant (the whole tail of an AFNO block): the fc1 input / residual is sum = f_cf + l_cf + n, with f_cf, l_cf
  // CHANNELS-FIRST [T / HW][C][HW] (irfft2 output and the `+ bias` path) and n the block's input tokens
  const float* f_cf;
  const float* l_cf;
  unsigned long long* trace;   // diagnostics (DLWP_TMLP_TRACE): [wave of workgroup 0][256] s_memtime stamps, or null
};

// packed-buffer layout (u32x4 units): w1hm | w2hm | w1l | w2l
__host__ __device__ inline size_t n_w1hm(int hid, int KS) { return (size_t)(hid / 16) * KS * 2 * 64; }
__host__ __device__ inline size_t n_w2hm(int hid, int OT) { return (size_t)(hid / 32) * 2 * OT * 64; }
__host__ __device__ inline size_t n_w1l(int hid, int KS) { return (size_t)(hid / 16) * KS * 64; }
__host__ __device__ inline size_t n_w2l(int hid, int OT) { return (size_t)(hid / 32) * OT * 64; }

// gamma / beta (both or neither): the affine part of a LayerNorm in front of fc1 is folded into the operands,
//   W1 (gamma * xhat + beta) + b1 = (W1 diag(gamma)) xhat + (b1 + W1 beta);  the folded bias lands behind w2l.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ b1, unsigned* __restrict__ dst, int C,
                                                   int hid, int merged_layout, int f16) {
  const int KS = C / 32, OT = C / 16;
  unsigned* w1hm = dst;
  unsigned* w2hm = w1hm + n_w1hm(hid, KS) * 4;
  unsigned* w1l = w2hm + n_w2hm(hid, OT) * 4;
  unsigned* w2l = w1l + n_w1l(hid, KS) * 4;
  const int n1 = (hid / 16) * KS * 64 * 4, n2 = (hid / 32) * OT * 64 * 4;
  // f16: the "f16x3" images (common.hpp) -- h = f16(w 2^s), m = f16((w 2^s - h) * 2^11), no third part.  W1 (its input is a
  // LayerNorm's output: the unit-scale form, split_f16_pair_unit) keeps s = 0; W2 takes the s of its maximum, found by
  // absmax_bits_kernel into scale[0] (the start of the unused w1l region), and scale[1] receives 2^-(11+s)
  float ws2 = 1.f;
  if (f16) {
    float osc;
    ws2 = f16x3_weight_scale(reinterpret_cast<const unsigned*>(w1l)[0], osc);
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(w1l)[1] = osc;
  }
  auto split = [&](float a, float b, unsigned& h, unsigned& m, unsigned& lo) {
    if (f16) {
      const f16x2v hh = __builtin_convertvector(f32x2{a, b}, f16x2v);
      const f32x2 r = {(a - (float)hh[0]) * 2048.0f, (b - (float)hh[1]) * 2048.0f};
      h = __builtin_bit_cast(unsigned, hh);
      m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
      lo = 0u;
    } else {
      split3_pair(a, b, h, m, lo);
    }
  };
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
    unsigned h, m, lo;
    if (i < n1) {
      const int d = i & 3, l = (i >> 2) & 63, tk = i >> 8;   // tk = t * KS + ks
      const int t = tk / KS, ks = tk % KS;
      // which channel k-slot (ks, g = l >> 4, e = 2d) of layer 1 carries: 32 ks + 8 g + e for the plain kernels (8 consecutive
      // channels of the token row per lane), 16 (2 ks + e / 4) + 4 g + e % 4 for the MERGE kernel (the lane's B-operand
      // channels are then exactly the channels of its accumulator rows: one set of loads feeds LayerNorm and residual)
      const int e0 = 2 * d;
      const int ch = merged_layout ? 16 * (2 * ks + e0 / 4) + 4 * (l >> 4) + e0 % 4 : 32 * ks + 8 * (l >> 4) + e0;
      const float* src = w1 + (size_t)(16 * t + (l & 15)) * C + ch;
      split(gamma ? src[0] * gamma[ch] : src[0], gamma ? src[1] * gamma[ch + 1] : src[1], h, m, lo);
      w1hm[((size_t)(tk * 2 + 0) * 64 + l) * 4 + d] = h;
      w1hm[((size_t)(tk * 2 + 1) * 64 + l) * 4 + d] = m;
      if (!f16) w1l[((size_t)tk * 64 + l) * 4 + d] = lo;
    } else {
      const int k = i - n1;
      const int d = k & 3, l = (k >> 2) & 63, uo = k >> 8;   // uo = u * OT + ot
      const int u = uo / OT, ot = uo % OT, g = l >> 4, jj = 2 * d;
      const int ch = 16 * (2 * u + jj / 4) + 4 * g + jj % 4;
      const float* src = w2 + (size_t)(16 * ot + (l & 15)) * hid + ch;
      split(src[0] * ws2, src[1] * ws2, h, m, lo);
      w2hm[((size_t)((u * 2 + 0) * OT + ot) * 64 + l) * 4 + d] = h;
      w2hm[((size_t)((u * 2 + 1) * OT + ot) * 64 + l) * 4 + d] = m;
      w2l[((size_t)uo * 64 + l) * 4 + d] = lo;
    }
  }
  if (gamma) {
    float* b1f = reinterpret_cast<float*>(w2l + n_w2l(hid, OT) * 4);
    for (int hrow = blockIdx.x * blockDim.x + threadIdx.x; hrow < hid; hrow += gridDim.x * blockDim.x) {
      float a = b1 ? b1[hrow] : 0.f;
      for (int c = 0; c < C; ++c) a = fmaf(w1[(size_t)hrow * C + c], beta[c], a);
      b1f[hrow] = a;
    }
  }
}

// six bf16 products per accumulator, smallest terms first: (A part, B part) = (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
__device__ constexpr int kPA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int kPB[6] = {0, 2, 1, 0, 1, 0};

// F16: the f16x3 product form (common.hpp), A operands the two LDS-resident images (wh, wm'), three products per accumulator;
// the third weight image and its global loads do not exist.  Layer 1 (input = a LayerNorm's output, O(1) by construction) in
// the unit-scale form: B operands (xh, xh * 2^-11, xm), true-scale accumulator.  Layer 2 (input = GELU(fc1), as small as fc1's
// weights make it) in the scaled form: B operands (xh, xm'), whB = wh * 2^11 formed from the wh fragment, accumulator at
// 2^(11+s) times the true scale -- it starts from (residual + bias) * 2^(11+s) and is multiplied back once per pass.
__device__ constexpr int kPA16[3] = {1, 0, 0};
__device__ constexpr int kPB16[3] = {1, 2, 0};

template <int KS, int OT, bool RESID, bool LN, bool NEXT, bool MERGE = false, bool F16 = false>
__global__ __launch_bounds__(512) void token_mlp_kernel(const Params p) {
  constexpr int C = 32 * KS;
  static_assert(OT * 16 == C, "square MLP");
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int ntile = p.hid >> 4, npair = ntile >> 1;
  const unsigned long long rt_entry = p.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  u32x4* s_w1 = reinterpret_cast<u32x4*>(smem);        // [ntile][KS][2][64]
  u32x4* s_w2 = s_w1 + (size_t)ntile * KS * 2 * 64;    // [npair][2][OT][64]
  float* s_b1 = reinterpret_cast<float*>(s_w2 + (size_t)npair * 2 * OT * 64);
  int* s_next = reinterpret_cast<int*>(s_b1 + p.hid);   // the workgroup's pass counter
  float* s_ng = reinterpret_cast<float*>(s_next + 4);   // NEXT: [C] gamma, [C] beta of the emitted LayerNorm
  float* s_b2 = s_ng + 2 * C;                           // [C] fc2 bias (zeros without one)
  if (tid == 0) *s_next = 0;
  if (tid < C) s_b2[tid] = p.b2 ? p.b2[tid] : 0.f;
  if (NEXT && tid < 2 * C) s_ng[tid] = tid < C ? p.next_gamma[tid] : p.next_beta[tid - C];
  {
    // all loads of a round in flight before the first LDS write (a load -> wait -> write loop took ~90 us for the 128 KB)
    auto stage = [&](u32x4* dst, const u32x4* src, int n) {
      for (int base = tid; base < n; base += 8 * (int)blockDim.x) {
        u32x4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = base + k * (int)blockDim.x;
          r[k] = src[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = base + k * (int)blockDim.x;
          if (i < n) dst[i] = r[k];
        }
      }
    };
    stage(s_w1, p.w1hm, ntile * KS * 2 * 64);
    stage(s_w2, p.w2hm, npair * 2 * OT * 64);
    for (int i = tid; i < p.hid; i += blockDim.x) s_b1[i] = p.b1[i];
  }
  __syncthreads();
  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.trace && blockIdx.x == 0 && lane == 0 && n_stamp < 253) p.trace[wave * 256 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  if (p.trace && blockIdx.x == 0 && lane == 0) {
    p.trace[wave * 256 + 253] = rt_entry;
    p.trace[wave * 256 + 254] = __builtin_amdgcn_s_memrealtime();
  }
  // fc2 bias: read from LDS where it is needed (16 registers held across the whole kernel were the difference between
  // 256 VGPRs with spills and none in the MERGE + NEXT variant)
  // (g_op: an OPAQUE copy of g, refreshed every pass -- otherwise hipcc hoists these pass-invariant LDS reads out of the
  // pass loop and keeps their 16 + 32 registers live across the whole kernel)
  float up2 = 1.f, down2 = 1.f;     // f16x3, layer 2: 2^(11+s) of the packed W2 and its inverse
  if constexpr (F16) {
    down2 = reinterpret_cast<const float*>(p.w1l)[1];
    up2 = pow2_reciprocal(down2);
  }
  int g_op = g;
  auto b2v = [&](int ot) { return *reinterpret_cast<const f32x4*>(s_b2 + 16 * ot + 4 * g_op); };

  // l parts (global, L2 resident): ONE register set.  A unit re-requests its layer-1 set right after its layer-1 MFMAs have
  // issued (they have read it) and its layer-2 set right after its layer-2 MFMAs -- for the NEXT unit (the last unit of a
  // pass requests unit 0's again), so each load has most of a unit to land.
  u32x4 wl1[2][KS], wl2[OT];
  auto load_lo1 = [&](int u) {
    if constexpr (F16) return;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wl1[tt][ks] = p.w1l[((2 * u + tt) * KS + ks) * 64 + lane];
  };
  auto load_lo2 = [&](int u) {
    if constexpr (F16) return;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) wl2[ot] = p.w2l[(u * OT + ot) * 64 + lane];
  };
  load_lo1(0);
  load_lo2(0);

  // Passes are handed out dynamically inside the workgroup: the older wave of a SIMD wins the issue arbitration and
  // runs ~1.5x faster than its partner (DLWP_TMLP_TRACE: 3.9k vs 5.4k cycles per tile pair), and a wave left alone
  // on its SIMD cannot overlap its own MFMA bursts with its GELU -- with a static split the fast half idled through
  // the last 25 % of the kernel.
  const long long npass = (p.T + 31) >> 5;
  const long long per_wg = (npass + gridDim.x - 1) / gridDim.x;
  const long long pass_lo = (long long)blockIdx.x * per_wg;
  const long long pass_hi = pass_lo + per_wg < npass ? pass_lo + per_wg : npass;
  for (;;) {
    int mine = 0;
    if (lane == 0) mine = atomicAdd(s_next, 1);
    const long long pass = pass_lo + __builtin_amdgcn_readfirstlane(mine);
    if (pass >= pass_hi) break;
    stamp();
    asm volatile("" : "+v"(g_op));
    // channels-first views (MERGE inputs, NEXT output): the 32 tokens of a pass share their sample (HW % 32 == 0)
    long long cf_off = 0;
    if (MERGE || NEXT) {
      const unsigned hw = (unsigned)p.HW, t0 = (unsigned)(pass * 32);   // tokens < 2^31 (checked on the host)
      const unsigned bsample = t0 / hw;
      cf_off = (long long)bsample * C * p.HW + (t0 - bsample * hw);
    }
    long long tok[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const long long t = pass * 32 + 2 * j + q;   // the lane's two tokens are neighbours: 8-byte channels-first accesses
      live[q] = t < p.T;
      tok[q] = live[q] ? t : p.T - 1;
    }
    u32x4 bx[2][KS][3];
    f32x4 acc2[OT][2];
    if constexpr (MERGE) {
      // sum = f + l (channels-first: for a fixed channel the lane group's 32 tokens are 128 contiguous bytes) + n (the
      // block's input tokens); this lane's 16 channels 16 ot + 4 g + r of token (q, j) are its accumulator rows AND, with
      // the merged k-slot layout of the packed W1, its layer-1 B operand: LayerNorm, residual and operand from one load set
      const float* fb = p.f_cf + cf_off;
      const float* lb = p.l_cf + cf_off;
      f32x4 svq[2][OT];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        svq[0][ot] = *reinterpret_cast<const f32x4*>(p.n + tok[0] * C + 16 * ot + 4 * g);
        svq[1][ot] = *reinterpret_cast<const f32x4*>(p.n + tok[1] * C + 16 * ot + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long o = (long long)(16 * ot + 4 * g + r) * p.HW + 2 * j;
          const float2 fv = *reinterpret_cast<const float2*>(fb + o), lv = *reinterpret_cast<const float2*>(lb + o);
          svq[0][ot][r] += fv.x + lv.x;
          svq[1][ot][r] += fv.y + lv.y;
        }
        const f32x4 bias2 = b2v(ot);
        acc2[ot][0] = svq[0][ot] + bias2;
        acc2[ot][1] = svq[1][ot] + bias2;
        if constexpr (F16) {
          acc2[ot][0] *= up2;
          acc2[ot][1] *= up2;
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4(&sv)[OT] = svq[q];
        float sum = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) sum += (sv[ot][0] + sv[ot][1]) + (sv[ot][2] + sv[ot][3]);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[ot][r] -= mean;
            sq = fmaf(sv[ot][r], sv[ot][r], sq);
          }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / C) + p.ln_eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // k-slots 2i, 2i+1 of k-step ks = channels 16 (2 ks + i / 2) + 4 g + 2 (i % 2), + 1
            unsigned hh, mm, ll;
            if constexpr (F16) split_f16_pair_unit(sv[2 * ks + i / 2][2 * (i % 2)] * rstd, sv[2 * ks + i / 2][2 * (i % 2) + 1] * rstd, hh, mm, ll);
            else split_pair_x<false>(sv[2 * ks + i / 2][2 * (i % 2)] * rstd, sv[2 * ks + i / 2][2 * (i % 2) + 1] * rstd, hh, mm, ll);
            bx[q][ks][0][i] = hh;
            bx[q][ks][1][i] = mm;
            bx[q][ks][2][i] = ll;
          }
      }
    } else {
      // layer-1 B operands: the token's channels 32 ks + 8 g .. + 7 (LayerNorm'd here in the LN variant: the four g lanes
      // of a token hold its C channels between them; two-pass statistics like torch), split
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float* src = p.n + tok[q] * C + 32 * ks + 8 * g;
          v[ks][0] = *reinterpret_cast<const f32x4*>(src);
          v[ks][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
        if (LN) {
          float sum = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) sum += (v[ks][hh][0] + v[ks][hh][1]) + (v[ks][hh][2] + v[ks][hh][3]);
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          const float mean = sum * (1.0f / C);
          float sq = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[ks][hh][k] -= mean;
                sq = fmaf(v[ks][hh][k], v[ks][hh][k], sq);
              }
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          const float rstd = rsqrtf(sq * (1.0f / C) + p.ln_eps);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) v[ks][hh] *= rstd;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned hh, mm, ll;
            static_assert(!F16 || MERGE, "f16x3: layer 1 uses the unit-scale split, which needs the LayerNorm of the MERGE variant in front");
            split_pair_x<false>(v[ks][i / 2][2 * (i % 2)], v[ks][i / 2][2 * (i % 2) + 1], hh, mm, ll);
            bx[q][ks][0][i] = hh;
            bx[q][ks][1][i] = mm;
            bx[q][ks][2][i] = ll;
          }
      }
      // the residual is the accumulator's start value: acc2[ot][q][r] = out[token (q, j)][16 ot + 4 g + r]
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc2[ot][q] = b2v(ot);
          if (RESID) acc2[ot][q] += *reinterpret_cast<const f32x4*>(p.resid + tok[q] * C + 16 * ot + 4 * g);
        }

    }
    // one hidden-tile pair
    auto unit = [&](int u) {
      const int un = u + 1 < npair ? u + 1 : 0;
      // ---- layer 1 of hidden tiles 2u, 2u+1 (4 independent accumulators, term-major so that back-to-back
      //      MFMAs never depend on each other)
      f32x4 a1[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + 16 * (2 * u + tt) + 4 * g);
        a1[tt][0] = bb;
        a1[tt][1] = bb;
      }
      __builtin_amdgcn_s_setprio(1);     // matrix bursts first: the SIMD's other wave fills their gaps with its GELU / splits
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 wa[2][3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int tk = (2 * u + tt) * KS + ks;
          wa[tt][0] = s_w1[(tk * 2 + 0) * 64 + lane];
          wa[tt][1] = s_w1[(tk * 2 + 1) * 64 + lane];
          if constexpr (!F16) wa[tt][2] = wl1[tt][ks];
        }
        if constexpr (F16) {
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int q = 0; q < 2; ++q)
                a1[tt][q] = mfma16x16x32_f16(wa[tt][kPA16[term]], bx[q][ks][kPB16[term]], a1[tt][q]);
        } else {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              a1[tt][q] = mfma16x16x32_bf16(wa[tt][kPA[term]], bx[q][ks][kPB[term]], a1[tt][q]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      load_lo1(un);
      // ---- GELU, then the 3-way split straight into layer-2 B operands
      gelu_erf8_fma(a1[0][0], a1[0][1]);
      gelu_erf8_fma(a1[1][0], a1[1][1]);
      u32x4 bg[2][3];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // dword i: k-slots 2i, 2i+1 -> tile i/2, registers 2(i%2), 2(i%2)+1
          unsigned hh, mm, ll;
          split_pair_x<F16>(a1[i / 2][q][2 * (i % 2)], a1[i / 2][q][2 * (i % 2) + 1], hh, mm, ll);
          bg[q][0][i] = hh;
          bg[q][1][i] = mm;
          bg[q][2][i] = ll;
        }
      // ---- layer 2: out tiles two at a time
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int oh = 0; oh < OT; oh += 2) {
        u32x4 wb[2][3];
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) {
          wb[oo][0] = s_w2[((u * 2 + 0) * OT + oh + oo) * 64 + lane];
          wb[oo][1] = s_w2[((u * 2 + 1) * OT + oh + oo) * 64 + lane];
          if constexpr (!F16) wb[oo][2] = wl2[oh + oo];
        }
        if constexpr (F16) {   // (wm', xh) (wh, xm') (whB, xh), all at 2^(11+s)
#pragma unroll
          for (int oo = 0; oo < 2; ++oo) wb[oo][2] = f16x8_times_2048(wb[oo][0]);
          constexpr int PA2[3] = {1, 0, 2}, PB2[3] = {0, 1, 0};
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int oo = 0; oo < 2; ++oo)
#pragma unroll
              for (int q = 0; q < 2; ++q)
                acc2[oh + oo][q] = mfma16x16x32_f16(wb[oo][PA2[term]], bg[q][PB2[term]], acc2[oh + oo][q]);
        } else {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int oo = 0; oo < 2; ++oo)
#pragma unroll
            for (int q = 0; q < 2; ++q)
              acc2[oh + oo][q] = mfma16x16x32_bf16(wb[oo][kPA[term]], bg[q][kPB[term]], acc2[oh + oo][q]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      load_lo2(un);
    };
    stamp();
    for (int u = 0; u < npair; ++u) {
      unit(u);
      stamp();
    }
    if constexpr (F16) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc2[ot][q] *= down2;
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (live[q]) *reinterpret_cast<f32x4*>(p.out + tok[q] * C + 16 * ot + 4 * g) = acc2[ot][q];
    if (NEXT) {
      // LayerNorm of the finished tokens (two-pass statistics over the 4 g lanes of a token), written channels-first:
      // for a fixed channel the 32 tokens of a lane group (two per lane) are 128 contiguous bytes
      float* dst = p.next_cf + cf_off;
      float mean[2], rstd[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) sum += (acc2[ot][q][0] + acc2[ot][q][1]) + (acc2[ot][q][2] + acc2[ot][q][3]);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        mean[q] = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dlt = acc2[ot][q][r] - mean[q];
            sq = fmaf(dlt, dlt, sq);
          }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        rstd[q] = rsqrtf(sq * (1.0f / C) + p.next_eps);
      }
      // (the channels-first forms require tokens % 32 == 0 on the host: no partial pass, both tokens of a lane are live)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(s_ng + 16 * ot + 4 * g_op);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(s_ng + C + 16 * ot + 4 * g_op);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float2*>(dst + (long long)(16 * ot + 4 * g + r) * p.HW + 2 * j) =
              float2{fmaf((acc2[ot][0][r] - mean[0]) * rstd[0], gm[r], bt[r]), fmaf((acc2[ot][1][r] - mean[1]) * rstd[1], gm[r], bt[r])};
      }
    }
  }
  stamp();
  if (p.trace && blockIdx.x == 0 && lane == 0) p.trace[wave * 256 + 255] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace tmlp
}  // namespace dlwp

using namespace dlwp;

static bool token_mlp_shape_ok(int C, int hid, size_t* lds) {
  if (C != 64 || hid < 64 || hid % 64) return false;
  const size_t bytes = (tmlp::n_w1hm(hid, C / 32) + tmlp::n_w2hm(hid, C / 16)) * 16 + (size_t)hid * 4 + 16 + (size_t)3 * C * 4;
  if (lds) *lds = bytes;
  return bytes <= 160 * 1024;
}

extern "C" size_t dlwp_token_mlp_packed_bytes(int32_t channels, int32_t hidden) {
  if (!token_mlp_shape_ok(channels, hidden, nullptr)) return 0;
  const int KS = channels / 32, OT = channels / 16;
  return (tmlp::n_w1hm(hidden, KS) + tmlp::n_w2hm(hidden, OT) + tmlp::n_w1l(hidden, KS) + tmlp::n_w2l(hidden, OT)) * 16 +
         (size_t)hidden * 4;   // + the folded fc1 bias of the LayerNorm variant
}

static int32_t token_mlp_pack(const float* w1_dev, const float* w2_dev, const float* ln_gamma_dev,
                              const float* ln_beta_dev, const float* b1_dev, int32_t channels,
                              int32_t hidden, int32_t merged_layout, void* packed_dev, void* stream, int f16) {
  DLWP_REQUIRE(w1_dev && w2_dev && packed_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE((ln_gamma_dev == nullptr) == (ln_beta_dev == nullptr), DLWP_ERR_INVALID_ARGUMENT,
               "LayerNorm weight and bias must be given together");
  DLWP_REQUIRE(token_mlp_shape_ok(channels, hidden, nullptr), DLWP_ERR_UNSUPPORTED,
               "token MLP: channels %d (64 supported), hidden %d (multiple of 64, <= 256: weights must fit LDS)", channels, hidden);
  if (f16) {   // the shift of W2's f16 images: max |w2| -> scale[0] at the start of the (unused) w1l region
    const int KS = channels / 32, OT = channels / 16;
    unsigned* scale = reinterpret_cast<unsigned*>(packed_dev) + (tmlp::n_w1hm(hidden, KS) + tmlp::n_w2hm(hidden, OT)) * 4;
    DLWP_HIP_CHECK(hipMemsetAsync(scale, 0, 8, reinterpret_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w2_dev,
                       (long long)channels * hidden, scale, (const float*)nullptr, 1);
  }
  hipLaunchKernelGGL(tmlp::pack_kernel, dim3(64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w1_dev, w2_dev,
                     ln_gamma_dev, ln_beta_dev, b1_dev, reinterpret_cast<unsigned*>(packed_dev), channels, hidden, merged_layout ? 1 : 0,
                     f16);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_token_mlp_pack_f32(const float* w1_dev, const float* w2_dev, const float* ln_gamma_dev,
                                           const float* ln_beta_dev, const float* b1_dev, int32_t channels,
                                           int32_t hidden, int32_t merged_layout, void* packed_dev, void* stream) {
  return token_mlp_pack(w1_dev, w2_dev, ln_gamma_dev, ln_beta_dev, b1_dev, channels, hidden, merged_layout, packed_dev, stream, 0);
}

extern "C" int32_t dlwp_token_mlp_pack_f16x3(const float* w1_dev, const float* w2_dev, const float* ln_gamma_dev,
                                             const float* ln_beta_dev, const float* b1_dev, int32_t channels,
                                             int32_t hidden, int32_t merged_layout, void* packed_dev, void* stream) {
  return token_mlp_pack(w1_dev, w2_dev, ln_gamma_dev, ln_beta_dev, b1_dev, channels, hidden, merged_layout, packed_dev, stream, 1);
}

static int32_t token_mlp_impl(const float* n_dev, const float* resid_dev, const void* packed_dev, const float* b1_dev,
                              const float* b2_dev, float* out_dev, int64_t tokens, int32_t channels, int32_t hidden,
                              float ln_eps, const float* next_gamma_dev, const float* next_beta_dev, float next_eps,
                              float* next_cf_dev, int64_t tokens_per_sample, void* stream,
                              const float* f_cf_dev = nullptr, const float* l_cf_dev = nullptr, bool f16 = false) {
  const bool ln = ln_eps >= 0.f;
  const bool next = next_cf_dev != nullptr;
  const bool merge = f_cf_dev != nullptr;
  if (next) DLWP_REQUIRE(next_gamma_dev && next_beta_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  if (merge) DLWP_REQUIRE(l_cf_dev && ln, DLWP_ERR_INVALID_ARGUMENT, "the merged form needs both channels-first inputs and the fused LayerNorm");
  if (next || merge) {
    DLWP_REQUIRE(tokens < (1ll << 31), DLWP_ERR_UNSUPPORTED, "token MLP: too many tokens for the channels-first forms");
    DLWP_REQUIRE(tokens_per_sample > 0 && tokens_per_sample % 32 == 0 && tokens % tokens_per_sample == 0,
                 DLWP_ERR_UNSUPPORTED, "token MLP: tokens per sample %lld must be a multiple of 32 and divide %lld tokens",
                 (long long)tokens_per_sample, (long long)tokens);
  }
  DLWP_REQUIRE(n_dev && packed_dev && (b1_dev || ln) && out_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(tokens > 0, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  size_t lds = 0;
  DLWP_REQUIRE(token_mlp_shape_ok(channels, hidden, &lds), DLWP_ERR_UNSUPPORTED,
               "token MLP: channels %d (64 supported), hidden %d (multiple of 64, <= 256: weights must fit LDS)", channels, hidden);
  const int KS = channels / 32, OT = channels / 16;
  tmlp::Params p;
  p.n = n_dev;
  p.resid = resid_dev;
  p.b2 = b2_dev;
  p.out = out_dev;
  const u32x4* base = reinterpret_cast<const u32x4*>(packed_dev);
  p.w1hm = base;
  p.w2hm = p.w1hm + tmlp::n_w1hm(hidden, KS);
  p.w1l = p.w2hm + tmlp::n_w2hm(hidden, OT);
  p.w2l = p.w1l + tmlp::n_w1l(hidden, KS);
  p.b1 = ln ? reinterpret_cast<const float*>(p.w2l + tmlp::n_w2l(hidden, OT)) : b1_dev;
  p.ln_eps = ln_eps;
  p.next_cf = next_cf_dev;
  p.next_gamma = next_gamma_dev;
  p.next_beta = next_beta_dev;
  p.next_eps = next_eps;
  p.HW = tokens_per_sample;
  p.f_cf = f_cf_dev;
  p.l_cf = l_cf_dev;
  p.T = tokens;
  p.hid = hidden;
  p.trace = nullptr;
  static const char* trace_path = getenv("DLWP_TMLP_TRACE");
  static unsigned long long* trace_buf = nullptr;
  static int traced = 0;
  if (trace_path && traced < 2) {
    if (!trace_buf) DLWP_HIP_CHECK(hipMalloc(&trace_buf, (size_t)8 * 256 * 8));
    DLWP_HIP_CHECK(hipMemsetAsync(trace_buf, 0, (size_t)8 * 256 * 8, reinterpret_cast<hipStream_t>(stream)));
    p.trace = trace_buf;
  }
  const long long npass = (tokens + 31) / 32;
  const unsigned grid = (unsigned)(npass < 8 * 256 ? (npass + 7) / 8 : 256);
  void (*kern)(const tmlp::Params) = nullptr;
#define DLWP_TM(R_, L_, N_) kern = tmlp::token_mlp_kernel<2, 4, R_, L_, N_>
  const int sel = (resid_dev ? 4 : 0) | (ln ? 2 : 0) | (next ? 1 : 0);
  switch (sel) {
    case 0: DLWP_TM(false, false, false); break;
    case 1: DLWP_TM(false, false, true); break;
    case 2: DLWP_TM(false, true, false); break;
    case 3: DLWP_TM(false, true, true); break;
    case 4: DLWP_TM(true, false, false); break;
    case 5: DLWP_TM(true, false, true); break;
    case 6: DLWP_TM(true, true, false); break;
    default: DLWP_TM(true, true, true); break;
  }
#undef DLWP_TM
  if (merge) kern = next ? tmlp::token_mlp_kernel<2, 4, true, true, true, true> : tmlp::token_mlp_kernel<2, 4, true, true, false, true>;
  if (f16) {
    DLWP_REQUIRE(merge, DLWP_ERR_UNSUPPORTED, "token MLP: the f16x3 form exists for the merged block tail only");
    kern = next ? tmlp::token_mlp_kernel<2, 4, true, true, true, true, true> : tmlp::token_mlp_kernel<2, 4, true, true, false, true, true>;
  }
  DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p);
  DLWP_HIP_CHECK(hipGetLastError());
  if (p.trace) {
    std::vector<unsigned long long> h((size_t)8 * 256);
    DLWP_HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    DLWP_HIP_CHECK(hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, traced ? "a" : "w")) {
      for (int w = 0; w < 8; ++w) {
        fprintf(f, "%d %d", traced, w);
        for (int i = 0; i < 256; ++i) fprintf(f, " %llu", h[(size_t)w * 256 + i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    ++traced;
  }
  return DLWP_OK;
}

extern "C" int32_t dlwp_token_mlp_f32(const float* n_dev, const float* resid_dev, const void* packed_dev,
                                      const float* b1_dev, const float* b2_dev, float* out_dev, int64_t tokens,
                                      int32_t channels, int32_t hidden, float ln_eps, void* stream) {
  return token_mlp_impl(n_dev, resid_dev, packed_dev, b1_dev, b2_dev, out_dev, tokens, channels, hidden, ln_eps, nullptr,
                        nullptr, 0.f, nullptr, 0, stream);
}

extern "C" int32_t dlwp_token_mlp_emit_norm_f32(const float* n_dev, const float* resid_dev, const void* packed_dev,
                                                const float* b1_dev, const float* b2_dev, float* out_dev,
                                                int64_t tokens, int32_t channels, int32_t hidden, float ln_eps,
                                                const float* next_gamma_dev, const float* next_beta_dev, float next_eps,
                                                float* next_cf_dev, int64_t tokens_per_sample, void* stream) {
  DLWP_REQUIRE(next_cf_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  return token_mlp_impl(n_dev, resid_dev, packed_dev, b1_dev, b2_dev, out_dev, tokens, channels, hidden, ln_eps,
                        next_gamma_dev, next_beta_dev, next_eps, next_cf_dev, tokens_per_sample, stream);
}

extern "C" int32_t dlwp_afno_block_tail_f32(const float* f_cf_dev, const float* l_cf_dev, const float* x_nhwc_dev,
                                            const void* packed_dev, const float* b2_dev, float* out_nhwc_dev,
                                            int32_t batch, int64_t tokens_per_sample, int32_t channels, int32_t hidden,
                                            float ln_eps, const float* next_gamma_dev, const float* next_beta_dev,
                                            float next_eps, float* next_cf_dev, void* stream) {
  DLWP_REQUIRE(f_cf_dev && l_cf_dev && x_nhwc_dev && out_nhwc_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && ln_eps >= 0.f, DLWP_ERR_INVALID_ARGUMENT, "bad arguments");
  return token_mlp_impl(x_nhwc_dev, x_nhwc_dev, packed_dev, nullptr, b2_dev, out_nhwc_dev, (int64_t)batch * tokens_per_sample,
                        channels, hidden, ln_eps, next_gamma_dev, next_beta_dev, next_eps, next_cf_dev, tokens_per_sample,
                        stream, f_cf_dev, l_cf_dev);
}

extern "C" int32_t dlwp_afno_block_tail_f16x3(const float* f_cf_dev, const float* l_cf_dev, const float* x_nhwc_dev,
                                              const void* packed_dev, const float* b2_dev, float* out_nhwc_dev,
                                              int32_t batch, int64_t tokens_per_sample, int32_t channels, int32_t hidden,
                                              float ln_eps, const float* next_gamma_dev, const float* next_beta_dev,
                                              float next_eps, float* next_cf_dev, void* stream) {
  DLWP_REQUIRE(f_cf_dev && l_cf_dev && x_nhwc_dev && out_nhwc_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && ln_eps >= 0.f, DLWP_ERR_INVALID_ARGUMENT, "bad arguments");
  return token_mlp_impl(x_nhwc_dev, x_nhwc_dev, packed_dev, nullptr, b2_dev, out_nhwc_dev, (int64_t)batch * tokens_per_sample,
                        channels, hidden, ln_eps, next_gamma_dev, next_beta_dev, next_eps, next_cf_dev, tokens_per_sample,
                        stream, f_cf_dev, l_cf_dev, /*f16=*/true);
}
