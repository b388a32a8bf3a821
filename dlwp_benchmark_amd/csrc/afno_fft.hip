// Hand-written 2-D real FFTs for the AFNO filter (reference models/fourcastnet/fourcastnet.py:85 `rfft2(dim=(1,2))`
// and :124 `irfft2`), restricted to what the filter keeps: the mixing of :93-118 reads and writes only the first
// `kept` columns of the half spectrum (hard thresholding), so the forward transform produces [H][KC] coefficients per
// plane (KC = kept columns) instead of [H][W/2+1] and the inverse one takes exactly those, treating the rest as zero.
// At the C4 shape (128 x 256, KC = 65 of 129) the spectrum is half the size rocFFT had to write, read, re-write (zeros)
// and read again, and the column transforms run on half the columns.
//
// One workgroup per [H][W] plane (channels-first: plane = (batch, channel)).  Everything between the coalesced load of
// the plane and the coalesced store of its spectrum happens in LDS and registers:
//   rows    : a real length-W FFT as a complex length-W/2 FFT of (even, odd) pairs + the split post-pass
//             F[k] = (Z[k] + conj Z[N-k]) / 2 - i/2 w_W^k (Z[k] - conj Z[N-k]),  only k < KC is ever formed;
//   columns : KC complex length-H FFTs on the LDS-resident [KC][H] image;
//   every complex FFT of length N = A * B runs as TWO register passes with one LDS exchange: radix-A butterflies over
//   n2 of x[n1 + B n2] (in registers), twiddle w_N^(n1 k2), stored in place; then radix-B over n1 of the B contiguous
//   values at B k2 -> X[k2 + A k1].  Radices are 4, 8 or 16 with compile-time twiddles.
// Both transforms are UNNORMALISED; the two 1/sqrt(HW) factors of norm="ortho" ride in the mixing kernel
// (dlwp_afno2d_mix_scaled_f32), which updates the [H][KC] spectrum in place.
// The c2r direction ignores the imaginary part of the DC column, like torch.fft.irfft2 / hipFFT C2R.
#include <cmath>

#include "common.hpp"

namespace dlwp {
namespace afft {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return float2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return float2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ float2 cconj(float2 a) { return float2{a.x, -a.y}; }
// multiply by SIGN * i  (SIGN = -1: the forward kernel's w_4 = -i)
template <int SIGN>
__device__ __forceinline__ float2 mul_si(float2 a) { return SIGN < 0 ? float2{a.y, -a.x} : float2{-a.y, a.x}; }

// w_16^k = exp(SIGN 2 pi i k / 16), k = 0..7 (compile-time constants)
template <int SIGN>
__device__ __forceinline__ float2 w16(int k) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
  const float cs[8] = {1.f, c1, r2, s1, 0.f, -s1, -r2, -c1};
  const float sn[8] = {0.f, s1, r2, c1, 1.f, c1, r2, s1};
  return float2{cs[k], SIGN * sn[k]};
}

template <int SIGN>
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_si<SIGN>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}
template <int R, int SIGN>
struct Dft;
template <int SIGN>
struct Dft<4, SIGN> {
  static __device__ __forceinline__ void run(float2 (&v)[4]) { dft4<SIGN>(v[0], v[1], v[2], v[3]); }
};
template <int SIGN>
struct Dft<8, SIGN> {
  static __device__ __forceinline__ void run(float2 (&v)[8]) {
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    Dft<4, SIGN>::run(e);
    Dft<4, SIGN>::run(o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 t = cmul(w16<SIGN>(2 * k), o[k]);
      v[k] = cadd(e[k], t);
      v[k + 4] = csub(e[k], t);
    }
  }
};
template <int SIGN>
struct Dft<16, SIGN> {
  static __device__ __forceinline__ void run(float2 (&v)[16]) {
    float2 e[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    Dft<8, SIGN>::run(e);
    Dft<8, SIGN>::run(o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 t = cmul(w16<SIGN>(k), o[k]);
      v[k] = cadd(e[k], t);
      v[k + 8] = csub(e[k], t);
    }
  }
};

// `count` FFTs of length N = A * B living in LDS as buf[f * stride + n]; items are dealt f-fastest (bank-conflict free
// for stride = odd number of float2 ... see the callers); tw[j] = exp(-2 pi i j / N), conjugated for SIGN = +1.
// PASS 1 (in place).
template <int A, int B, int SIGN, int NT>
__device__ __forceinline__ void fft_pass1(float2* buf, int stride, int count, const float2* tw, int tid) {
  for (int it = tid; it < count * B; it += NT) {
    const int f = it % count, n1 = it / count;
    float2* p = buf + f * stride + n1;
    float2 v[A];
#pragma unroll
    for (int n2 = 0; n2 < A; ++n2) v[n2] = p[B * n2];
    Dft<A, SIGN>::run(v);
#pragma unroll
    for (int k2 = 0; k2 < A; ++k2) {
      float2 w = tw[n1 * k2];
      if (SIGN > 0) w.y = -w.y;
      p[B * k2] = k2 == 0 ? v[0] : cmul(v[k2], w);
    }
  }
}
// PASS 2: reads the B contiguous values at B k2, leaves X[k2 + A k1] (k1 = 0..B-1) in `v`.
template <int A, int B, int SIGN>
__device__ __forceinline__ void fft_pass2_regs(const float2* row, int k2, float2 (&v)[B]) {
#pragma unroll
  for (int n1 = 0; n1 < B; ++n1) v[n1] = row[B * k2 + n1];
  Dft<B, SIGN>::run(v);
}
// PASS 2 in place: X[k2 + A k1] is stored where its inputs were, at B k2 + k1 -- "scrambled" order; a reader finds
// X[k] at pos<A, B>(k).  No second buffer, no barrier between the reads and the writes of different items.
template <int A, int B>
__device__ __forceinline__ int pos(int k) { return B * (k % A) + k / A; }
template <int A, int B, int SIGN, int NT>
__device__ __forceinline__ void fft_pass2_inplace(float2* buf, int stride, int count, int tid) {
  for (int it = tid; it < count * A; it += NT) {
    float2* row = buf + (it % count) * stride;
    const int k2 = it / count;
    float2 v[B];
    fft_pass2_regs<A, B, SIGN>(row, k2, v);
#pragma unroll
    for (int k1 = 0; k1 < B; ++k1) row[B * k2 + k1] = v[k1];
  }
}

struct Tables {
  const float2* tw_row;   // [NR]  exp(-2 pi i j / NR)
  const float2* tw_col;   // [H]   exp(-2 pi i j / H)
  const float2* tw_w;     // [NR+1] exp(-2 pi i k / W)
};

// ---------------------------------------------------------------------------------------------------------------
// forward: x [planes][H][W] -> spec [planes][H][KC] complex (unnormalised)
// ---------------------------------------------------------------------------------------------------------------
template <int H, int W, int AR, int BR, int AC, int BC, int NT>
__global__ __launch_bounds__(NT) void afno_rfft2_kept_kernel(const float* __restrict__ x, float2* __restrict__ spec,
                                                            Tables T, int KC, int planes) {
  constexpr int NR = W / 2;
  static_assert(AR * BR == NR && AC * BC == H, "decomposition");
  constexpr int RCH = H < 64 ? H : 64;          // rows per chunk
  constexpr int RS = NR + 1, CS = H + 1;        // LDS strides (float2): odd -> items that differ in the FFT index hit different banks
  extern __shared__ __align__(16) float smem[];
  float2* cb = reinterpret_cast<float2*>(smem);                 // [KC][CS]   column image
  float2* rb = cb + (size_t)KC * CS;                            // [RCH][RS]  row work buffer
  float2* s_twr = rb + RCH * RS;                                // [NR]
  float2* s_twc = s_twr + NR;                                   // [H]
  float2* s_tww = s_twc + H;                                    // [NR + 1]
  const int tid = threadIdx.x;
  for (int i = tid; i < NR; i += NT) s_twr[i] = T.tw_row[i];
  for (int i = tid; i < H; i += NT) s_twc[i] = T.tw_col[i];
  for (int i = tid; i <= NR; i += NT) s_tww[i] = T.tw_w[i];
  constexpr int F4R = W / 4;                    // float4 per row = pairs of packed complex values
  constexpr int NCH = H / RCH;                  // row chunks per plane
  constexpr int PF = (RCH * F4R + NT - 1) / NT; // float4 per thread and chunk
  // Persistent over planes: the global loads of the NEXT row chunk (of this plane or the next one) are issued before the
  // current chunk's passes and land in registers while they run -- one workgroup per CU has nothing else to hide them.
  float4 pre[PF];
  auto prefetch = [&](long long pl, int chunk) {
    const float4* xp = reinterpret_cast<const float4*>(x + pl * H * W) + (long long)chunk * RCH * F4R;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = tid + q * NT;
      if (i < RCH * F4R) pre[q] = xp[i];
    }
  };
  long long plane = blockIdx.x;
  if (plane < planes) prefetch(plane, 0);
  for (; plane < planes; plane += gridDim.x) {
  for (int ch = 0; ch < NCH; ++ch) {
    const int r0 = ch * RCH;
    // (a) the chunk's rows, already in registers: a float4 is two complex values z[n] = x[2n] + i x[2n+1]
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = tid + q * NT;
      if (i < RCH * F4R) {
        const int r = i / F4R, m = i % F4R;
        rb[r * RS + 2 * m] = float2{pre[q].x, pre[q].y};
        rb[r * RS + 2 * m + 1] = float2{pre[q].z, pre[q].w};
      }
    }
    if (ch + 1 < NCH) prefetch(plane, ch + 1);
    else if (plane + gridDim.x < planes) prefetch(plane + gridDim.x, 0);
    __syncthreads();
    fft_pass1<AR, BR, -1, NT>(rb, RS, RCH, s_twr, tid);
    __syncthreads();
    fft_pass2_inplace<AR, BR, -1, NT>(rb, RS, RCH, tid);     // Z[k] now sits at pos<AR, BR>(k)
    __syncthreads();
    // (d) split post-pass: the kept bins of the real transform, into the column image cb[k][row]
    for (int it = tid; it < RCH * KC; it += NT) {
      const int k = it % KC, r = it / KC;
      const float2 zk = rb[r * RS + pos<AR, BR>(k == NR ? 0 : k)], zc = cconj(rb[r * RS + pos<AR, BR>((NR - k) % NR)]);
      const float2 s = cadd(zk, zc), dd = csub(zk, zc);
      const float2 t = cmul(s_tww[k], dd);           // w_W^k (Z[k] - conj Z[N-k])
      // F = s/2 - (i/2) t
      cb[k * CS + r0 + r] = float2{0.5f * (s.x + t.y), 0.5f * (s.y - t.x)};
    }
    __syncthreads();
  }
  // ---- columns: KC FFTs of length H in cb
  fft_pass1<AC, BC, -1, NT>(cb, CS, KC, s_twc, tid);
  __syncthreads();
  float2* sp = spec + plane * H * KC;
  for (int it = tid; it < KC * AC; it += NT) {
    const int kx = it % KC, k2 = it / KC;
    float2 v[BC];
    fft_pass2_regs<AC, BC, -1>(cb + kx * CS, k2, v);
#pragma unroll
    for (int k1 = 0; k1 < BC; ++k1) sp[(long long)(k2 + AC * k1) * KC + kx] = v[k1];   // lanes run along kx: coalesced
  }
  __syncthreads();   // cb is rewritten by the next plane's post-passes
  }
}

// ---------------------------------------------------------------------------------------------------------------
// inverse: spec [planes][H][KC] complex (columns >= KC are zero) -> y [planes][H][W]  (unnormalised)
// ---------------------------------------------------------------------------------------------------------------
template <int H, int W, int AR, int BR, int AC, int BC, int NT, int KCB>
__global__ __launch_bounds__(NT) void afno_irfft2_kept_kernel(const float2* __restrict__ spec, float* __restrict__ y,
                                                             Tables T, int KC, int planes) {
  constexpr int NR = W / 2;
  static_assert(AR * BR == NR && AC * BC == H, "decomposition");
  constexpr int RCH = H < 64 ? H : 64;
  constexpr int RS = NR + 1, CS = H + 1;
  extern __shared__ __align__(16) float smem[];
  float2* cb = reinterpret_cast<float2*>(smem);
  float2* rb = cb + (size_t)KC * CS;
  float2* s_twr = rb + RCH * RS;
  float2* s_twc = s_twr + NR;
  float2* s_tww = s_twc + H;
  const int tid = threadIdx.x;
  for (int i = tid; i < NR; i += NT) s_twr[i] = T.tw_row[i];
  for (int i = tid; i < H; i += NT) s_twc[i] = T.tw_col[i];
  for (int i = tid; i <= NR; i += NT) s_tww[i] = T.tw_w[i];
  // Persistent over planes; the NEXT plane's spectrum travels into registers while this plane's passes run.
  constexpr int PSMAX = (H * KCB + NT - 1) / NT;         // float2 per thread at most (KC <= KCB, checked on the host)
  const int per = (H * KC + NT - 1) / NT;
  float2 pre[PSMAX];
  auto prefetch = [&](long long pl) {
    const float2* sp = spec + pl * H * KC;
#pragma unroll
    for (int q = 0; q < PSMAX; ++q) {
      const int i = tid + q * NT;
      if (q < per && i < H * KC) pre[q] = sp[i];
    }
  };
  long long plane = blockIdx.x;
  if (plane < planes) prefetch(plane);
  for (; plane < planes; plane += gridDim.x) {
  // (a) spectrum -> cb[kx][ky] (the global reads ran coalesced along kx)
#pragma unroll
  for (int q = 0; q < PSMAX; ++q) {
    const int i = tid + q * NT;
    if (q < per && i < H * KC) cb[(i % KC) * CS + i / KC] = pre[q];
  }
  if (plane + gridDim.x < planes) prefetch(plane + gridDim.x);
  __syncthreads();
  // (b) columns, inverse
  fft_pass1<AC, BC, +1, NT>(cb, CS, KC, s_twc, tid);
  __syncthreads();
  fft_pass2_inplace<AC, BC, +1, NT>(cb, CS, KC, tid);       // column value of row h now sits at pos<AC, BC>(h)
  __syncthreads();
  // (c) rows: Z[k] = (G[k] + conj G[N-k]) + i w_W^-k (G[k] - conj G[N-k]),  G = 0 beyond KC, Im G[0] ignored
  float4* yp = reinterpret_cast<float4*>(y + plane * H * W);
  constexpr int F4R = W / 4;
  for (int r0 = 0; r0 < H; r0 += RCH) {
    for (int it = tid; it < RCH * NR; it += NT) {
      const int r = it % RCH, k = it / RCH;
      const int kc = NR - k;
      const int hp = pos<AC, BC>(r0 + r);
      float2 gk = k < KC ? cb[k * CS + hp] : float2{0.f, 0.f};
      float2 gc = kc < KC ? cconj(cb[kc * CS + hp]) : float2{0.f, 0.f};
      if (k == 0) gk.y = 0.f;                      // c2r semantics: imaginary part of the DC bin is ignored
      // (k = 0 pairs with the Nyquist bin N, which only exists when KC = N + 1; its imaginary part is ignored too)
      if (k == 0) gc.y = 0.f;
      const float2 s = cadd(gk, gc), dd = csub(gk, gc);
      const float2 t = cmul(cconj(s_tww[k]), dd);  // w_W^-k (G[k] - conj G[N-k])
      rb[r * RS + k] = float2{s.x - t.y, s.y + t.x};   // s + i t
    }
    __syncthreads();
    fft_pass1<AR, BR, +1, NT>(rb, RS, RCH, s_twr, tid);
    __syncthreads();
    fft_pass2_inplace<AR, BR, +1, NT>(rb, RS, RCH, tid);     // z[n] now sits at pos<AR, BR>(n)
    __syncthreads();
    // x[2n] = Re z[n], x[2n+1] = Im z[n]: the row is the float2 array itself; coalesced 16-byte stores
    for (int i = tid; i < RCH * F4R; i += NT) {
      const int r = i / F4R, m = i % F4R;
      const float2 a = rb[r * RS + pos<AR, BR>(2 * m)], b = rb[r * RS + pos<AR, BR>(2 * m + 1)];
      yp[(long long)(r0 + r) * F4R + m] = float4{a.x, a.y, b.x, b.y};
    }
    __syncthreads();
  }
  }
}

}  // namespace afft
}  // namespace dlwp

using namespace dlwp;

struct dlwp_afno_fft_plan {
  int H = 0, W = 0, KC = 0, cus = 256;
  DevBuf tables;   // tw_row [NR] | tw_col [H] | tw_w [NR + 1]
};

namespace {
template <int H, int W>
struct Shape {
  static constexpr bool ok = false;
};
#define DLWP_AFFT_SHAPE(H_, W_, AR_, BR_, AC_, BC_, NT_)                          \
  template <>                                                                       \
  struct Shape<H_, W_> {                                                            \
    static constexpr bool ok = true;                                                \
    static constexpr int AR = AR_, BR = BR_, AC = AC_, BC = BC_, NT = NT_;          \
    static constexpr int NTI = NT_ > 512 ? 512 : NT_;   /* inverse: 512 threads (256 VGPRs) -- 1024 spilled */ \
    /* inverse with at most half the columns kept (a smaller register prefetch): 768 threads, 12 waves to cover the barriers */ \
    static constexpr int NTIH = NT_ > 512 ? 768 : NT_; \
  };
// (H, W) -> row FFT of W/2 = AR * BR, column FFT of H = AC * BC, threads per workgroup
DLWP_AFFT_SHAPE(128, 256, 8, 16, 16, 8, 1024)
DLWP_AFFT_SHAPE(64, 128, 8, 8, 8, 8, 512)
DLWP_AFFT_SHAPE(32, 64, 4, 8, 8, 4, 256)
DLWP_AFFT_SHAPE(64, 64, 4, 8, 8, 8, 256)
DLWP_AFFT_SHAPE(32, 32, 4, 4, 8, 4, 256)
#undef DLWP_AFFT_SHAPE

template <int H, int W>
size_t lds_bytes(int KC) {
  constexpr int NR = W / 2, RCH = H < 64 ? H : 64;
  return ((size_t)KC * (H + 1) + (size_t)RCH * (NR + 1) + NR + H + NR + 1) * sizeof(float2);
}

template <int H, int W>
int32_t run_shape(const dlwp_afno_fft_plan* p, const float* x, float* spec, float* y, int planes, bool inverse, hipStream_t s) {
  using S = Shape<H, W>;
  constexpr int NR = W / 2;
  const size_t lds = lds_bytes<H, W>(p->KC);
  DLWP_REQUIRE(lds <= 160 * 1024, DLWP_ERR_UNSUPPORTED, "AFNO FFT %dx%d with %d kept columns needs %zu bytes of LDS", H, W, p->KC, lds);
  // persistent workgroups: as many as fit the chip at once (LDS-bound), each walks planes blockIdx.x, + grid, ...
  const int per_cu = (int)((160 * 1024) / (lds ? lds : 1)) > 0 ? (int)((160 * 1024) / lds) : 1;
  int grid = p->cus * (per_cu > 8 ? 8 : per_cu);
  if (grid > planes) grid = planes;
  afft::Tables T;
  T.tw_row = p->tables.as<float2>();
  T.tw_col = T.tw_row + NR;
  T.tw_w = T.tw_col + H;
  if (!inverse) {
    auto kern = afft::afno_rfft2_kept_kernel<H, W, S::AR, S::BR, S::AC, S::BC, S::NT>;
    if (lds > 48 * 1024)
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::NT), lds, s, x, reinterpret_cast<float2*>(spec), T, p->KC, planes);
  } else {
    // two instantiations by how many columns are kept: the next plane's spectrum is prefetched into registers
    constexpr int KHALF = NR / 2 + 1;
    const bool half = p->KC <= KHALF;
    auto kern = half ? afft::afno_irfft2_kept_kernel<H, W, S::AR, S::BR, S::AC, S::BC, S::NTIH, KHALF>
                     : afft::afno_irfft2_kept_kernel<H, W, S::AR, S::BR, S::AC, S::BC, S::NTI, NR + 1>;
    if (lds > 48 * 1024)
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(half ? S::NTIH : S::NTI), lds, s, reinterpret_cast<const float2*>(spec), y, T, p->KC, planes);
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

bool shape_supported(int H, int W) {
  return (H == 128 && W == 256) || (H == 64 && W == 128) || (H == 32 && W == 64) || (H == 64 && W == 64) || (H == 32 && W == 32);
}

int32_t dispatch(const dlwp_afno_fft_plan* p, const float* x, float* spec, float* y, int planes, bool inverse, hipStream_t s) {
  const int H = p->H, W = p->W;
  if (H == 128 && W == 256) return run_shape<128, 256>(p, x, spec, y, planes, inverse, s);
  if (H == 64 && W == 128) return run_shape<64, 128>(p, x, spec, y, planes, inverse, s);
  if (H == 32 && W == 64) return run_shape<32, 64>(p, x, spec, y, planes, inverse, s);
  if (H == 64 && W == 64) return run_shape<64, 64>(p, x, spec, y, planes, inverse, s);
  if (H == 32 && W == 32) return run_shape<32, 32>(p, x, spec, y, planes, inverse, s);
  return fail(DLWP_ERR_UNSUPPORTED, "AFNO FFT: grid %dx%d is not instantiated", H, W);
}
}  // namespace

extern "C" int32_t dlwp_afno_fft_supported(int32_t H, int32_t W, int32_t kept_cols) {
  return shape_supported(H, W) && kept_cols >= 1 && kept_cols <= W / 2 + 1 ? 1 : 0;
}

extern "C" int32_t dlwp_afno_fft_plan_create(dlwp_afno_fft_plan** out, int32_t H, int32_t W, int32_t kept_cols, void* stream) {
  DLWP_REQUIRE(out, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  DLWP_REQUIRE(dlwp_afno_fft_supported(H, W, kept_cols), DLWP_ERR_UNSUPPORTED,
               "AFNO FFT: grid %dx%d with %d kept columns is not supported (rocFFT path: dlwp_fft2_plan_create)", H, W, kept_cols);
  auto* p = new dlwp_afno_fft_plan;
  p->H = H; p->W = W; p->KC = kept_cols;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      p->cus = cus;
  }
  const int NR = W / 2;
  std::vector<float> tw((size_t)(NR + H + NR + 1) * 2);
  const double pi = 3.14159265358979323846;
  size_t o = 0;
  for (int j = 0; j < NR; ++j, ++o) { tw[2 * o] = (float)std::cos(2 * pi * j / NR); tw[2 * o + 1] = (float)-std::sin(2 * pi * j / NR); }
  for (int j = 0; j < H; ++j, ++o) { tw[2 * o] = (float)std::cos(2 * pi * j / H); tw[2 * o + 1] = (float)-std::sin(2 * pi * j / H); }
  for (int j = 0; j <= NR; ++j, ++o) { tw[2 * o] = (float)std::cos(2 * pi * j / W); tw[2 * o + 1] = (float)-std::sin(2 * pi * j / W); }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = p->tables.upload(tw.data(), tw.size() * 4, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { delete p; return fail(DLWP_ERR_HIP, "twiddle upload failed: %s", hipGetErrorString(e)); }
  *out = p;
  return DLWP_OK;
}

extern "C" int32_t dlwp_afno_fft_plan_destroy(dlwp_afno_fft_plan* plan) {
  delete plan;
  return DLWP_OK;
}

extern "C" int32_t dlwp_afno_rfft2_kept_f32(const dlwp_afno_fft_plan* plan, const float* x_dev, float* spec_dev, int32_t planes,
                                            void* stream) {
  DLWP_REQUIRE(plan && x_dev && spec_dev && planes > 0, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(spec_dev) & 7) == 0,
               DLWP_ERR_INVALID_ARGUMENT, "pointers must be 16-byte (spectrum 8-byte) aligned");
  return dispatch(plan, x_dev, spec_dev, nullptr, planes, false, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int32_t dlwp_afno_irfft2_kept_f32(const dlwp_afno_fft_plan* plan, const float* spec_dev, float* y_dev, int32_t planes,
                                             void* stream) {
  DLWP_REQUIRE(plan && spec_dev && y_dev && planes > 0, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE((reinterpret_cast<uintptr_t>(y_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(spec_dev) & 7) == 0,
               DLWP_ERR_INVALID_ARGUMENT, "pointers must be 16-byte (spectrum 8-byte) aligned");
  return dispatch(plan, nullptr, const_cast<float*>(spec_dev), y_dev, planes, true, reinterpret_cast<hipStream_t>(stream));
}
