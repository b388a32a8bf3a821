// fp32 Linear layers on the bf16 matrix pipe ("bf16x6" GEMM) with the block's pointwise work in the epilogue.
//
// Replaces the qkv / proj / fc1 / fc2 Linears of the Swin and Pangu blocks (reference
// models/swintransformer/swin_transformer.py:21-39 `Mlp`, :107-120 qkv / proj, :254-262 the two residual adds;
// models/panguweather/panguweather.py:176-211, :318-322) which round 1 ran as fp32 rocBLAS GEMMs (51 % of the Pangu step)
// plus separate GELU and add passes:
//   out[m][n] = act( sum_k x[m][k] W[n][k] + bias[n] ) + resid[m][n]          (act: none | exact-erf GELU)
// fp32 in, fp32 out, fp32-GEMM accuracy: every operand is split EXACTLY into three bf16 parts (x = h + m + l) and the six
// significant cross products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (common.hpp; measured as accurate as
// a plain fp32 GEMM).  The bf16 pipe is 16x the fp32 matrix rate, so six passes still leave 2.6x -- and the fp32 vector
// lanes stay free for the splits and the epilogue.
//
// Weights are split ONCE (dlwp_linear_pack_f32: three bf16 images [N][K]); activations are split while their tile is
// staged into LDS.  Tile: BM = 128 rows of x, BN = 128 or 64 rows of W, BK = 32; 256 threads = 2 x 2 waves, a wave owns
// 64 x BN/2 outputs = 4 x (BN/32) MFMA tiles.  The MFMA takes W as its A operand and x as its B operand, so a lane ends
// up with 4 CONSECUTIVE output columns n of one row m: bias / residual / store are 16-byte accesses.  One LDS buffer,
// next k-step's global loads in flight in registers during the current step's MFMAs.
#include "common.hpp"

namespace dlwp {
namespace lin {

constexpr int BM = 128, BK = 32;
constexpr int LDT = BK + 16;   // bf16 elements per LDS row (96 bytes: conflict-free for the b128 operand reads, see window_attn2.hip)

struct Params {
  const float* x;              // [M][K]
  const unsigned short* w;     // bf16 parts [3][N][K]
  const float* bias;           // [N] or null
  const float* resid;          // [M][N] or null (may alias out)
  float* out;                  // [M][N]
  long long M, wpart;          // wpart: elements between two parts of w
  int N, K, act;
};

__device__ __forceinline__ float apply_act(float v, int act) { return act == 1 ? gelu_erf(v) : v; }

template <int BN>
__global__ __launch_bounds__(256, 2) void linear_bf16x6_kernel(const Params p) {
  constexpr int TN = BN / 32;                        // 16-row W tiles per wave (wave tile: 64 m x BN/2 n)
  extern __shared__ __align__(16) unsigned short smem_u16[];
  typedef unsigned short (*XT)[BM][LDT];
  typedef unsigned short (*WT)[BN][LDT];
  XT s_x = reinterpret_cast<XT>(smem_u16);                              // [3][BM][LDT]
  WT s_w = reinterpret_cast<WT>(smem_u16 + 3 * BM * LDT);               // [3][BN][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;           // wave grid 2 (m) x 2 (n)
  const int tiles_n = (p.N + BN - 1) / BN;
  // the tiles_n workgroups that share one 128-row slab of x run on ONE XCD (blocks are dealt round-robin over the 8 XCDs
  // by linear id): the slab is fetched into that XCD's L2 once
  const int xcd = blockIdx.x & 7;
  const long long rloc = blockIdx.x >> 3;
  const long long tile_m = (rloc / tiles_n) * 8 + xcd;
  const int tile_n = (int)(rloc % tiles_n);
  const long long m0 = tile_m * BM;
  if (m0 >= p.M) return;
  const int n0 = tile_n * BN;

  f32x4 acc[4][TN];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: x tile 128 rows x 32 floats = 1024 float4 -> 4 per thread; W tiles 3 x BN rows x 64 bytes = 12 BN 16-byte chunks
  constexpr int WCH = 3 * BN * 4 / 256;              // 16-byte chunks per thread (BN = 128: 6, BN = 64: 3)
  float4 px[4];
  u32x4 pw[WCH];
  // rows past M / N are CLAMPED to the last valid row (no branches, no zero fill): their products land in accumulators
  // the epilogue never stores
  constexpr int WPP = BN * 4 / 256;                  // 16-byte chunks per thread per W part (BN = 128: 2, BN = 64: 1)
  const float* xsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid + q * 256, r = i >> 3, c4 = i & 7;
    const long long m = m0 + r < p.M ? m0 + r : p.M - 1;
    xsrc[q] = p.x + m * p.K + 4 * c4;
  }
  const unsigned short* wsrc[WPP];
#pragma unroll
  for (int q = 0; q < WPP; ++q) {
    const int rem = tid + q * 256, r = rem >> 2, c8 = rem & 3;
    const int n = n0 + r < p.N ? n0 + r : p.N - 1;
    wsrc[q] = p.w + (long long)n * p.K + 8 * c8;
  }
  const long long wpart = p.wpart;
  auto load = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) px[q] = *reinterpret_cast<const float4*>(xsrc[q] + k0);
#pragma unroll
    for (int q = 0; q < WCH; ++q) pw[q] = *reinterpret_cast<const u32x4*>(wsrc[q % WPP] + (q / WPP) * wpart + k0);
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + q * 256, r = i >> 3, c4 = i & 7;
      unsigned h0, m0_, l0, h1, m1, l1;
      split3_pair(px[q].x, px[q].y, h0, m0_, l0);
      split3_pair(px[q].z, px[q].w, h1, m1, l1);
      *reinterpret_cast<uint2*>(&s_x[0][r][4 * c4]) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(&s_x[1][r][4 * c4]) = uint2{m0_, m1};
      *reinterpret_cast<uint2*>(&s_x[2][r][4 * c4]) = uint2{l0, l1};
    }
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
      const int rem = tid + (q % WPP) * 256, r = rem >> 2, c8 = rem & 3;
      *reinterpret_cast<u32x4*>(&s_w[q / WPP][r][8 * c8]) = pw[q];
    }
  };

  const int nk = p.K / BK;
  load(0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();                 // every wave is done reading the previous step's tiles
    stage();
    if (ks + 1 < nk) load((ks + 1) * BK);
    __syncthreads();
    u32x4 xb[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int part = 0; part < 3; ++part)
        xb[a][part] = *reinterpret_cast<const u32x4*>(&s_x[part][wm * 64 + 16 * a + j][8 * g]);
    // six cross products, smallest first: (A part, B part) = (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); A = W, B = x
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      u32x4 wa[3];
#pragma unroll
      for (int part = 0; part < 3; ++part)
        wa[part] = *reinterpret_cast<const u32x4*>(&s_w[part][wn * (BN / 2) + 16 * b + j][8 * g]);
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_bf16(wa[PA[term]], xb[a][PB[term]], acc[a][b]);
    }
  }

  // epilogue: lane (j, g) holds out[m = m0 + wm 64 + 16 a + j][n = n0 + wn BN/2 + 16 b + 4 g + 0..3]
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long long m = m0 + wm * 64 + 16 * a + j;
    if (m >= p.M) continue;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn * (BN / 2) + 16 * b + 4 * g;
      if (n >= p.N) continue;            // N is a multiple of 4: a lane's four columns are all in or all out
      f32x4 v = acc[a][b];
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      }
      float* o = p.out + m * p.N + n;
      if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.N + n);
      *reinterpret_cast<f32x4*>(o) = v;
    }
  }
}

// weights [N][K] fp32 -> three bf16 images [N][K]
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ h,
                                                          unsigned short* __restrict__ m, unsigned short* __restrict__ l,
                                                          long long pairs) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long long)gridDim.x * 256) {
    const float2 v = reinterpret_cast<const float2*>(w)[i];
    unsigned hh, mm, ll;
    split3_pair(v.x, v.y, hh, mm, ll);
    reinterpret_cast<unsigned*>(h)[i] = hh;
    reinterpret_cast<unsigned*>(m)[i] = mm;
    reinterpret_cast<unsigned*>(l)[i] = ll;
  }
}

}  // namespace lin
}  // namespace dlwp

using namespace dlwp;

extern "C" size_t dlwp_linear_packed_bytes(int32_t out_features, int32_t in_features) {
  if (out_features <= 0 || in_features <= 0 || in_features % 32 || out_features % 4) return 0;
  return 3 * align_up((size_t)out_features * in_features * 2, 256);
}

extern "C" int32_t dlwp_linear_pack_f32(const float* weight_dev, int32_t out_features, int32_t in_features, void* packed_dev,
                                        void* stream) {
  DLWP_REQUIRE(weight_dev && packed_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  const size_t bytes = dlwp_linear_packed_bytes(out_features, in_features);
  DLWP_REQUIRE(bytes > 0, DLWP_ERR_UNSUPPORTED, "linear: in_features %d must be a multiple of 32, out_features %d of 4",
               in_features, out_features);
  const size_t part = bytes / 3;
  char* b = reinterpret_cast<char*>(packed_dev);
  const long long pairs = (long long)out_features * in_features / 2;
  long long blocks = (pairs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lin::linear_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     weight_dev, reinterpret_cast<unsigned short*>(b), reinterpret_cast<unsigned short*>(b + part),
                     reinterpret_cast<unsigned short*>(b + 2 * part), pairs);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

extern "C" int32_t dlwp_linear_f32(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                                   float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                                   void* stream) {
  DLWP_REQUIRE(x_dev && packed_dev && out_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(rows > 0, DLWP_ERR_INVALID_ARGUMENT, "rows must be positive");
  DLWP_REQUIRE(act == 0 || act == 1, DLWP_ERR_INVALID_ARGUMENT, "linear: act must be 0 (none) or 1 (GELU)");
  const size_t bytes = dlwp_linear_packed_bytes(out_features, in_features);
  DLWP_REQUIRE(bytes > 0, DLWP_ERR_UNSUPPORTED, "linear: in_features %d must be a multiple of 32, out_features %d of 4",
               in_features, out_features);
  DLWP_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_dev) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(packed_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias_dev) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(resid_dev) & 15) == 0,
               DLWP_ERR_INVALID_ARGUMENT, "pointers must be 16-byte aligned");
  lin::Params p;
  const size_t part = bytes / 3;
  const char* b = reinterpret_cast<const char*>(packed_dev);
  p.w = reinterpret_cast<const unsigned short*>(b);
  p.wpart = (long long)(part / 2);
  p.x = x_dev; p.bias = bias_dev; p.resid = resid_dev; p.out = out_dev;
  p.M = rows; p.N = out_features; p.K = in_features; p.act = act;
  const long long tiles_m = (rows + lin::BM - 1) / lin::BM;
  // 128-wide W tiles unless that wastes a quarter or more of the last one (N = 192, 576, ...)
  const bool wide = (out_features % 128) == 0 || (out_features % 128) > 96;
  const long long tiles_n = wide ? (out_features + 127) / 128 : (out_features + 63) / 64;
  DLWP_REQUIRE(tiles_m * tiles_n < (1ll << 31), DLWP_ERR_UNSUPPORTED, "linear: too many tiles");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;     // m-tile slabs in groups of 8 (one per XCD)
  DLWP_REQUIRE(grid < (1ll << 31), DLWP_ERR_UNSUPPORTED, "linear: too many tiles");
  if (wide) {
    constexpr size_t lds = (size_t)3 * (lin::BM + 128) * lin::LDT * 2;
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lin::linear_bf16x6_kernel<128>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lin::linear_bf16x6_kernel<128>, dim3((unsigned)grid), dim3(256), lds, s, p);
  } else {
    constexpr size_t lds = (size_t)3 * (lin::BM + 64) * lin::LDT * 2;
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lin::linear_bf16x6_kernel<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lin::linear_bf16x6_kernel<64>, dim3((unsigned)grid), dim3(256), lds, s, p);
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}
