// fp32 Linear layers on the bf16 matrix pipe ("bf16x6" GEMM) with the block's pointwise work in the epilogue.
//
// Replaces the qkv / proj / fc1 / fc2 Linears of the Swin and Pangu blocks (reference
// models/swintransformer/swin_transformer.py:21-39 `Mlp`, :107-120 qkv / proj, :254-262 the two residual adds;
// models/panguweather/panguweather.py:176-211, :318-322) which round 1 ran as fp32 rocBLAS GEMMs (51 % of the Pangu step)
// plus separate GELU and add passes:
//   out[m][n] = act( sum_k x[m][k] W[n][k] + bias[n] ) + resid[m][n]          (act: none | exact-erf GELU)
// fp32 in, fp32 out.  NP = 3 (dlwp_linear_f32): fp32-GEMM accuracy -- every operand is split EXACTLY into three bf16 parts
// (x = h + m + l) and the six significant cross products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (common.hpp;
// measured MORE accurate than rocBLAS' fp32 GEMM on the same inputs, tools/bench_linear.py).  The bf16 pipe is 16x the fp32
// matrix rate, so six passes still leave 2.6x -- and the fp32 vector lanes stay free for the splits and the epilogue.
// NP = 1 (dlwp_linear_bf16): bf16 operands (one product), fp32 accumulation -- what autocast(bfloat16) makes of nn.Linear.
//
// Weights are split ONCE (dlwp_linear_pack_f32: three bf16 images [N][K]); activations are split while their tile is
// staged into LDS.  Tile: BM = 128 rows of x, BN = 128 or 96 rows of W, BK = 32; 256 threads = 2 x 2 waves, a wave owns
// 64 x BN/2 outputs = 4 x (BN/32) MFMA tiles; 72 KB (60 KB) of LDS: two workgroups per CU.  The MFMA takes W as its A operand
// and x as its B operand, so a lane ends up with 4 CONSECUTIVE output columns n of one row m: bias / residual / store
// are 16-byte accesses.
//
// Structure (what each piece bought is in DESIGN.md section 7.5):
//   * persistent workgroups walk a list of output tiles as ONE flattened sequence of k-steps: the loads of the first
//     k-steps of the next tile are issued during the last k-steps of the current one, so short-K Linears (K = 96: three
//     k-steps) have no prologue bubble per tile.  XCD x (= blockIdx % 8) owns the 128-row slabs x, x + 8, ... of the
//     activation; its workgroups walk (slab, n-tile) pairs n-tile fastest, so a slab is fetched into that XCD's L2 once;
//   * W tiles (already bf16) go global -> LDS by LDS-DMA into two buffers, one k-step ahead; x tiles (fp32) come through
//     registers: x(t+1) is split at the start of step t behind a barrier (every wave holds its x fragments of step t in
//     registers first, so ONE x buffer suffices) and x(t+2) is issued into the same registers right behind the split;
//   * every load in the loop is inline asm with hand-counted s_waitcnt vmcnt(N), tied to the registers it releases:
//     hipcc cannot count loads whose use is an iteration away and drains the queue (vmcnt(0)) instead -- the same reason
//     the output stores are asm and every compiler-visible load of the epilogue is consumed on every path;
//   * LDS rows are 64 bytes (32 bf16), unpadded, with the 16-byte chunks XOR-swizzled by f(row) = (4 - (row & 15) / 4) & 3:
//     the ds_read_b128 operand reads (lane groups {0-3, 12-15, 20-27}, ...) are conflict free (SQ_LDS_BANK_CONFLICT = 0).
#include "common.hpp"
#include <atomic>
#include <type_traits>

namespace dlwp {
namespace lin {

constexpr int BM = 128, BK = 32;

#ifndef DLWP_RING_LDPOL
#define DLWP_RING_LDPOL ""        // cache policy of the operand DMAs (A/B: " nt", " sc0", " sc1")
#endif
#ifndef DLWP_RING_STPOL
#define DLWP_RING_STPOL " nt"     // cache policy of the output stores: streamed past the L2 -- the 200-600 MB an output tensor writes
#endif                            // evicted the x slab / W panel the next tiles re-read from it (C5 l2 qkv 124 -> 97 us, A/B r03m)
#ifndef DLWP_RING_PATCH
#define DLWP_RING_PATCH 1         // 1: whole-line stores through a per-wave LDS patch; 0: straight from the accumulator layout
#endif
#ifndef DLWP_LIN_STPOL
#define DLWP_LIN_STPOL " nt"      // the same for linear_kernel's fp32 stores (16 bytes per lane, 64-byte runs: f16x3 qkv 280 -> 253 us;
#endif                            // NOT its 8-byte bf16 stores: partial lines streamed past the L2 cost 2-3x, A/B r03n)
struct Params {
  const float* x;              // [M][K]
  const unsigned short* w;     // bf16 parts [3][N][K]
  const float* bias;           // [N] or null
  const float* resid;          // [M][N] or null (may alias out)
  float* out;                  // [M][N]
  long long M, wpart;          // wpart: elements between two parts of w
  int N, K, act;
  const float* wscale;         // f16x3 form: {bits of max |w|, 2^-(11+s)} written by the packer behind the two images; else null
};

// Output stores as inline asm: hipcc guards the reuse of a store's data registers with s_waitcnt vmcnt -- and knowing
// nothing of the asm prefetches queued behind the stores it makes that vmcnt(0), draining the pipeline once per tile.
// The hardware reads the data at issue; it only asks for wait states before a write to the registers of a wide store
// (the trailing s_nop 1, checked by tools/asm_hazard_check.py rule C).
__device__ __forceinline__ void store4(float* dst, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off" DLWP_LIN_STPOL "\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}

__device__ __forceinline__ void store2(unsigned short* dst, const uint2& v) {
  asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}

__device__ __forceinline__ int swz(int row) { return (4 - ((row & 15) >> 2)) & 3; }

struct Cursor {          // one (tile, k-step) position of this workgroup's flattened step sequence
  int slab, nt, ks, left;    // slab index inside the XCD, n-tile, k-step, tiles after this one
};

// F16 (dlwp_linear_f16x3, NP = 3): the "f16x3" form of common.hpp -- x parts (xh, xm' = (x - xh) * 2^11), W parts (wh, wm' = (w 2^s - wh) * 2^11;
// whB = wh * 2^11 is formed in registers), three f16 products per output instead of six bf16 ones, 5 instead of 11 split slots per
// pair; only TWO W images are staged.  The accumulator holds 2^(11+s) times the result: the epilogue's bias add is an FMA.
// XB16 / OB16 (bf16-operand form only): x is ALREADY bf16 [M][K] / the output is stored as bf16 [M][N].  The MLP of a block
// in the bf16 form hands its hidden activation from fc1 to fc2 this way: fc2 rounds its input to bf16 anyway, so the
// result is bit-identical and the widest tensor of the block crosses HBM at half the bytes in both directions.
#ifndef DLWP_LIN_WAVES_NP1
#define DLWP_LIN_WAVES_NP1 2     // resident workgroups per CU the bf16-operand form is compiled for (A/B: tools/ab_build2.sh)
#endif
template <int BN, int NP, bool F16 = false, bool XB16 = false, bool OB16 = false>
__global__ __launch_bounds__(256, NP == 1 ? DLWP_LIN_WAVES_NP1 : 2) void linear_kernel(const Params p) {
  static_assert(!(XB16 || OB16) || (NP == 1 && !F16), "bf16 in / out exists for the bf16-operand form");
  constexpr int XQ = XB16 ? 2 : 4;                   // 16-byte x loads per thread and k-step
  constexpr int TN = BN / 32;                        // 16-row W tiles per wave (wave tile: 64 m x BN/2 n)
  constexpr int NPW = F16 ? 2 : NP;                  // W parts staged per k-step
  constexpr int NPX = F16 ? 2 : NP;                  // x images in LDS (f16x3: xh and xm')
  constexpr int XBYTES = NPX * 8192;                 // x tile: NPX parts x 128 rows x 64 bytes
  constexpr int WPART = BN * 64, WSTAGE = NPW * WPART;
  constexpr int CH = NPW * BN * 4;                   // 16-byte chunks per W stage
  constexpr int PIECES = CH / 64;                    // 1 KiB LDS-DMA pieces (16 rows x 64 bytes) per W stage
  constexpr int G = (PIECES + 3) / 4;                // pieces per wave and k-step
  extern __shared__ __align__(1024) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  float osc = 1.f;
  if constexpr (F16) osc = p.wscale[1];

  // this workgroup's tiles: XCD x (= blockIdx % 8) owns the 128-row slabs x, x + 8, ...; its workgroups walk the
  // (slab, n-tile) pairs n-tile fastest, so the workgroups that run together on an XCD share slabs of x in its L2
  const int tiles_n = (p.N + BN - 1) / BN;
  const long long tiles_m = (p.M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int slabs = tiles_m > xcd ? (int)((tiles_m - xcd + 7) / 8) : 0;
  const int ntile = slabs * tiles_n;                 // (host: fits an int)
  if (ntile <= li) return;
  const int my_tiles = (ntile - li + per_xcd - 1) / per_xcd;
  const int nk = p.K / BK;
  const long long total = (long long)my_tiles * nk;
  const int dq = per_xcd / tiles_n, dr = per_xcd % tiles_n;      // one stride of the tile sequence in (slab, n-tile) steps

  auto advance = [&](Cursor& c) -> bool {            // true when the cursor moved to another tile
    if (++c.ks < nk) return false;
    c.ks = 0;
    if (c.left == 0) return false;                   // past the end: stay on the last tile (re-loaded, results unused)
    --c.left;
    c.slab += dq;
    c.nt += dr;
    if (c.nt >= tiles_n) { c.nt -= tiles_n; ++c.slab; }
    return true;
  };
  auto row0 = [&](const Cursor& c) -> long long { return ((long long)c.slab * 8 + xcd) * BM; };

  // ---- x stream: 128 rows x 32 floats per k-step, 4 float4 per thread.  Loads are `global_load_dwordx4 v, voffset, s[base]`:
  // a scalar tile base (advanced by the cursor) plus a 32-bit per-thread offset, half the address registers of flat pointers.
  unsigned xoff[4];
  auto x_rows = [&](const Cursor& c) {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const long long left = p.M - 1 - row0(c);                   // rows past M: clamped to the last row, never stored
      if constexpr (XB16) {                                       // a row of the k-step is 64 bytes = four 16-byte chunks of 8 bf16
        const int i = tid + q * 256, r = i >> 2, c8 = i & 3;
        const int rr = r < left ? r : (int)left;
        xoff[q] = ((unsigned)rr * (unsigned)p.K + 8u * c8) * 2u;
      } else {
        const int i = tid + q * 256, r = i >> 3, c4 = i & 7;
        const int rr = r < left ? r : (int)left;
        xoff[q] = ((unsigned)rr * (unsigned)p.K + 4u * c4) * 4u;
      }
    }
  };
  auto x_issue = [&](f32x4 (&dst)[4], const Cursor& c) {
    const float* base = XB16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.x) + row0(c) * p.K + c.ks * BK)
                             : p.x + row0(c) * p.K + c.ks * BK;
#pragma unroll
    for (int q = 0; q < XQ; ++q) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[q]) : "v"(xoff[q]), "s"(base));
  };
  // ---- W stream: NP x BN rows x 64 bytes per k-step = PIECES 1 KiB pieces (16 rows x 64 bytes) moved global -> LDS by
  // global_load_lds_dwordx4: no registers, no ds_write.  Wave w moves pieces w, w + 4, ... (a wave whose last index
  // falls past PIECES repeats its previous piece: every wave issues exactly G, so one vmcnt constant serves all).
  // M0 = LDS base of the piece; lane l lands at + 16 l = row l / 4, position l % 4, which must hold the k-chunk
  // (l % 4) ^ f(row): the swizzle is applied to the SOURCE address.
  // [Register staging of W (global_load_dwordx4 + ds_write_b128, 24 more VGPRs) was measured on the same structure: equal
  // within 4 %, slower on the K >= 384 shapes.  The pieces are slow to ISSUE -- 6 per wave took 570-1700 cycles, the
  // LDS-DMA path moves ~13 B/clk/CU -- but the other workgroup of the CU computes meanwhile.]
  unsigned woff[G], wdst[G];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto w_rows = [&](const Cursor& c) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      int pc = wave + 4 * i;
      if (pc >= PIECES) pc -= 4;
      const int part = pc / (BN / 16), rb = pc % (BN / 16), r = lane >> 2, pos = lane & 3;
      const int nn = c.nt * BN + 16 * rb + r;
      const int n = nn < p.N ? nn : p.N - 1;                       // rows past N: clamped, never stored
      woff[i] = ((unsigned)part * (unsigned)p.wpart + (unsigned)n * (unsigned)p.K + 8u * (pos ^ swz(r))) * 2u;
      wdst[i] = lds0 + XBYTES + pc * 1024;
    }
  };
  auto w_dma = [&](int buf, const Cursor& c) {
    const unsigned short* base = p.w + c.ks * BK;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(wdst[i] + buf * WSTAGE);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(woff[i]), "s"(base), "s"(dst) : "memory");
    }
  };
  // ---- split a register set into the x tile
  auto x_store = [&](const f32x4 (&src)[4]) {
    if constexpr (XB16) {   // already bf16: a plain (swizzled) copy
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int i = tid + q * 256, r = i >> 2, c8 = i & 3;
        *reinterpret_cast<f32x4*>(smem + r * 64 + ((c8 ^ swz(r)) << 4)) = src[q];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + q * 256, r = i >> 3, c4 = i & 7;
      unsigned char* d = smem + r * 64 + (((c4 >> 1) ^ swz(r)) << 4) + ((c4 & 1) << 3);
      if constexpr (NP == 3) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split_pair_x<F16>(src[q][0], src[q][1], h0, m0_, l0);
        split_pair_x<F16>(src[q][2], src[q][3], h1, m1, l1);
        *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
        if constexpr (F16) {   // split order (xh, xm', -)
          *reinterpret_cast<uint2*>(d + 8192) = uint2{m0_, m1};
        } else {
          *reinterpret_cast<uint2*>(d + 8192) = uint2{m0_, m1};
          *reinterpret_cast<uint2*>(d + 16384) = uint2{l0, l1};
        }
      } else {
        *reinterpret_cast<uint2*>(d) = uint2{cvt_pk_bf16(src[q][0], src[q][1]), cvt_pk_bf16(src[q][2], src[q][3])};
      }
    }
  };

  f32x4 acc[4][TN];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cursor cc, cw, cx;        // compute position, W load position (one step ahead), x load position (two steps ahead)
  cc.slab = li / tiles_n; cc.nt = li % tiles_n; cc.ks = 0; cc.left = my_tiles - 1;
  cw = cc; cx = cc;
  f32x4 px[4];
  // prologue: x(0) split into the x tile, W(0) into buffer 0, x(1) on its way
  x_rows(cx);
  w_rows(cw);
  x_issue(px, cx);
  w_dma(0, cw);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(px[0]), "+v"(px[1]), "+v"(px[2]), "+v"(px[3]));
  x_store(px);
  if (advance(cx)) x_rows(cx);
  x_issue(px, cx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const unsigned char* xfrag = smem + (wm * 64 + j) * 64 + ((g ^ swz(j)) << 4);
  const unsigned char* wfrag = smem + XBYTES + (wn * (BN / 2) + j) * 64 + ((g ^ swz(j)) << 4);

  // epilogue: lane (j, g) holds out[m = m0 + wm 64 + 16 a + j][n = n0 + wn BN/2 + 16 b + 4 g + 0..3].  Two copies (with /
  // without a residual operand): one body with `resid ? load : 0` made hipcc guard the zero-initialised registers of the
  // other path with s_waitcnt vmcnt(0) -- a full drain of the stores just issued, once per tile.
  auto epilogue = [&](auto has_resid) __attribute__((always_inline)) {
    constexpr bool RES = decltype(has_resid)::value;
    const long long mbase = row0(cc) + wm * 64 + j;
    const int nbase = cc.nt * BN + wn * (BN / 2) + 4 * g;
    // bias and ALL residual rows of the tile are requested together, from clamped addresses, and awaited once (one
    // dependent load per bounds branch made hipcc drain the whole load queue, prefetches included, at every one of them;
    // one batch per pair of row tiles paid the HBM latency twice per tile)
    f32x4 bv[TN], rs[4][TN];
    int ncl[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) ncl[b] = nbase + 16 * b < p.N ? nbase + 16 * b : p.N - 4;     // N % 4 == 0: a lane's 4 columns are all in or all out
#pragma unroll
    for (int b = 0; b < TN; ++b) bv[b] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ncl[b]) : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RES) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const long long m = mbase + 16 * a < p.M ? mbase + 16 * a : p.M - 1;
#pragma unroll
        for (int b = 0; b < TN; ++b) rs[a][b] = *reinterpret_cast<const f32x4*>(p.resid + m * p.N + ncl[b]);
      }
    }
    // every compiler-visible load is CONSUMED here, on every path: hipcc sinks `u += rs` into the bounds branch of the
    // store, and a load still pending on the skip path costs an s_waitcnt vmcnt(0) at the first reuse of its register --
    // the operand reads at the top of the next k-step, right behind the freshly issued prefetches
#pragma unroll
    for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(bv[b]));
    if constexpr (RES) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(rs[a][b]));
    }
#pragma unroll
    for (int a = 0; a < 4; a += 2) {
      const long long m_u = mbase + 16 * a, m_v = m_u + 16;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        f32x4 u, v;
        if constexpr (F16) {   // the accumulator holds 2^(11+s) times the product sum
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            u[e] = __builtin_fmaf(acc[a][b][e], osc, bv[b][e]);
            v[e] = __builtin_fmaf(acc[a + 1][b][e], osc, bv[b][e]);
          }
        } else {
          u = acc[a][b] + bv[b];
          v = acc[a + 1][b] + bv[b];
        }
        if (p.act) gelu_erf8_fma(u, v);
        if constexpr (RES) {
          u += rs[a][b];
          v += rs[a + 1][b];
        }
        const int n = nbase + 16 * b;
        if (n < p.N) {
          if constexpr (OB16) {
            unsigned short* ob = reinterpret_cast<unsigned short*>(p.out);
            if (m_u < p.M) store2(ob + m_u * p.N + n, uint2{cvt_pk_bf16(u[0], u[1]), cvt_pk_bf16(u[2], u[3])});
            if (m_v < p.M) store2(ob + m_v * p.N + n, uint2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])});
          } else {
            if (m_u < p.M) store4(p.out + m_u * p.N + n, u);
            if (m_v < p.M) store4(p.out + m_v * p.N + n, v);
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // One k-step t.  Load queue order: x(t+1) [issued in step t-1 behind its split], W(t+1), x(t+2).
  auto step = [&](const int cur) __attribute__((always_inline)) {
    if (advance(cw)) w_rows(cw);
    w_dma(cur ^ 1, cw);
    u32x4 xb[4][NP];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (F16) {
        xb[a][0] = *reinterpret_cast<const u32x4*>(xfrag + a * 1024);
        xb[a][1] = *reinterpret_cast<const u32x4*>(xfrag + 8192 + a * 1024);
      } else {
#pragma unroll
        for (int part = 0; part < NP; ++part) xb[a][part] = *reinterpret_cast<const u32x4*>(xfrag + part * 8192 + a * 1024);
      }
    }
    const unsigned char* wf = wfrag + cur * WSTAGE;
    u32x4 wa[NPW];         // the first W fragments too: the MFMAs can start right behind the barrier
#pragma unroll
    for (int part = 0; part < NPW; ++part) wa[part] = *reinterpret_cast<const u32x4*>(wf + part * WPART);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // every wave holds its x fragments: the x tile is free
    asm volatile("" ::: "memory");
    // x(t+1) has landed when at most this step's G loads of W are outstanding
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(px[0]), "+v"(px[1]), "+v"(px[2]), "+v"(px[3]) : "n"(G));
    x_store(px);
    if (advance(cx)) x_rows(cx);
    x_issue(px, cx);                         // x(t+2), into the registers just split
    __builtin_amdgcn_s_setprio(1);           // matrix phase first: the other resident workgroup's vector phases fill its gaps (2-7 % measured)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if (b > 0) {
#pragma unroll
        for (int part = 0; part < NPW; ++part) wa[part] = *reinterpret_cast<const u32x4*>(wf + part * WPART + b * 1024);
      }
      if constexpr (F16) {
        // three products at one scale, the residual ones first: (A part, B part) = (wm', xh) (wh, xm') (whB, xh); whB = wh * 2^11
        // is four packed multiplies per W fragment (it replaces the xs = xh * 2^-11 of round 2: the same count at TN = 4)
        const u32x4 whb = f16x8_times_2048(wa[0]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_f16(wa[1], xb[a][0], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_f16(wa[0], xb[a][1], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_f16(whb, xb[a][0], acc[a][b]);
      } else if constexpr (NP == 3) {
        // six cross products, smallest first: (A part, B part) = (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); A = W, B = x
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_bf16(wa[PA[term]], xb[a][PB[term]], acc[a][b]);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_bf16(wa[0], xb[a][0], acc[a][b]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // W(t+1) has landed when at most the 4 x loads issued behind it are outstanding
    if (cc.ks == nk - 1) {
      if (p.resid) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
    advance(cc);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XQ) : "memory");   // W(t+1) is in its buffer (queued behind it: the XQ loads of x(t+2))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the x tile and W buffer writes are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (long long t = 0; t < total; ++t) step((int)(t & 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prefetches issued past the end
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-operand form with a bf16 x tensor (dlwp_linear_bf16_io, x_is_bf16) -- "ring" kernel (round 3).
//
// What bounded linear_kernel at the Swin / Pangu block shapes in their bf16 form (M = 65 536 ... 262 144 rows, K = 96 ... 1 536;
// knock-out builds, profiles/r03_linear_knockouts.txt): NOT the matrix instructions (removing them changed nothing) and not the
// epilogue arithmetic, but the memory operations and the way one wave's vmcnt queue couples them -- vmcnt retires in order, so
// (a) the wait for the next operand stage also waited for the output stores of the tile just finished, (b) hipcc's wait for the
// epilogue's bias / residual loads drained every operand stage in flight, once per output tile, and (c) 128 x 128 tiles stage
// 196 KB of operands per 16 K outputs through a path (L2 -> LDS) that moves ~18 TB/s chip-wide at best.
//
// This kernel: 256 x BN tiles (8 waves = 4 x 2, a wave owns 64 x BN/2 as before: a quarter less staging per output), BOTH
// operands by LDS-DMA into a RING of S stages, S - 1 k-steps ahead, ONE barrier per k-step; and EVERY memory operation of a wave
// is inline asm with a statically known count, so that each wait names exactly what it needs:
//   * operand stage t:   all but the DMAs of stages t + 1 .. t + S - 2 AND the foreign operations (epilogue loads, stores) issued
//                        in the last S - 1 steps -- those are younger than the stage's DMA and may stay in flight;
//   * bias / residual:   requested DL <= 4 steps BEFORE the epilogue that uses them (asm loads into registers held until then),
//                        awaited with the count of the DMAs issued since;
//   * output stores:     buffer_store with the range check of the descriptor (rows past M and columns past N get an offset
//                        beyond num_records and are dropped): always issued, never branched around -- a static count.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BMR = 256;

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {      // s_waitcnt takes an immediate: one scalar jump over 64 cases
  switch (n < 63 ? n : 63) {
#define DLWP_W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define DLWP_W8(k) DLWP_W1(k) DLWP_W1(k + 1) DLWP_W1(k + 2) DLWP_W1(k + 3) DLWP_W1(k + 4) DLWP_W1(k + 5) DLWP_W1(k + 6) DLWP_W1(k + 7)
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
    case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
    case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
    case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
    case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
    case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
    case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    case 57: asm volatile("s_waitcnt vmcnt(57)" ::: "memory"); break;
    case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
    case 59: asm volatile("s_waitcnt vmcnt(59)" ::: "memory"); break;
    case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    case 61: asm volatile("s_waitcnt vmcnt(61)" ::: "memory"); break;
    case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
#undef DLWP_W8
#undef DLWP_W1
  }
}

template <int BN, bool OB16, int S, bool RES>
__global__ __launch_bounds__(512, 1) void linear_ring_kernel(const Params p) {
  // k-steps of 64: a tile row is 128 bytes = ONE request of the vector L1 for a whole line.  The L1's outstanding-request slots
  // times the L2 latency are what bound this kernel (profiles/r03_linear_pmc_mem.txt: the L1 stalled "pending" 61 % of the time,
  // 0.17 requests per clock and CU at a read latency of 430 cycles): the same bytes in half the requests.
  constexpr int BKR = 64, ROWB = 2 * BKR;            // bytes per tile row
  constexpr int TN = BN / 32;
  constexpr int XP = BMR / 8, WP = BN / 8;           // 1 KiB pieces (8 rows x 128 bytes) per stage
  constexpr int GX = XP / 8, GW = (WP + 7) / 8;      // pieces per wave and stage
  constexpr int GD = GX + GW;                        // DMAs a wave issues per step
  constexpr int XBYTES = BMR * ROWB, STAGE = XBYTES + BN * ROWB;
  constexpr int NFLY = GD * (S - 2);                 // DMAs that may stay in flight when stage t is needed
  constexpr int EL = TN + (RES ? 4 * TN : 0);        // epilogue loads of a wave and tile (bias, residual rows)
#if DLWP_RING_PATCH
  constexpr int ES = OB16 ? 8 : 8 * ((TN + 1) / 2);  // output stores of a wave and tile (whole-line stores through the LDS patch)
#else
  constexpr int ES = 4 * TN;                         // output stores of a wave and tile
#endif
  static_assert(S >= 3 && S <= 8 && NFLY + EL + ES <= 63, "ring depth / counted waits");
  extern __shared__ __align__(1024) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int wm = wave & 3, wn = wave >> 2;
  const unsigned short* xg = reinterpret_cast<const unsigned short*>(p.x);

  const int tiles_n = (p.N + BN - 1) / BN;
  const long long tiles_m = (p.M + BMR - 1) / BMR;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int slabs = tiles_m > xcd ? (int)((tiles_m - xcd + 7) / 8) : 0;
  const int ntile = slabs * tiles_n;
  if (ntile <= li) return;
  const int my_tiles = (ntile - li + per_xcd - 1) / per_xcd;
  const int nk = p.K / BKR;                                    // >= 3 (host)
  const long long total = (long long)my_tiles * nk;
  const int dq = per_xcd / tiles_n, dr = per_xcd % tiles_n;
  auto advance = [&](Cursor& c) -> bool {
    if (++c.ks < nk) return false;
    c.ks = 0;
    if (c.left == 0) return false;
    --c.left;
    c.slab += dq;
    c.nt += dr;
    if (c.nt >= tiles_n) { c.nt -= tiles_n; ++c.slab; }
    return true;
  };
  auto row0 = [&](const Cursor& c) -> long long { return ((long long)c.slab * 8 + xcd) * BMR; };

  // LDS image: rows of 128 bytes = 8 chunks of 16; chunk c of row r sits at position c ^ ((r >> 1) & 7): the ds_read_b128
  // operand reads (lane groups {0-3, 12-15, 20-27}, ... = rows j of one 16-row block at chunk 4 h + g) then touch 16 different
  // 16-byte slots of the 256-byte bank row.  LDS-DMA writes lane-linear pieces: the swizzle is applied to the SOURCE address.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned xoff[GX], xdst[GX], woff[GW], wdst[GW];
  auto rows_of = [&](const Cursor& c) {
    const int r = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      const int pc = wave + 8 * i;
      const long long left = p.M - 1 - row0(c);                    // rows past M: clamped to the last row, never stored
      const int rr = 8 * pc + r;
      const int row = rr < left ? rr : (int)left;
      xoff[i] = ((unsigned)row * (unsigned)p.K + 8u * (pos ^ ((rr >> 1) & 7))) * 2u;
      xdst[i] = lds0 + pc * 1024;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
      int pc = wave + 8 * i;
      if (pc >= WP) pc -= 8;                                       // a wave past the last piece repeats its previous one
      if (pc < 0 || pc >= WP) pc = wave % WP;
      const int rr = 8 * pc + r;
      const int nn = c.nt * BN + rr;
      const int n = nn < p.N ? nn : p.N - 1;
      woff[i] = ((unsigned)n * (unsigned)p.K + 8u * (pos ^ ((rr >> 1) & 7))) * 2u;
      wdst[i] = lds0 + XBYTES + pc * 1024;
    }
  };
  auto dma = [&](int slot, const Cursor& c) {
    const unsigned short* xb = xg + row0(c) * p.K + c.ks * BKR;
    const unsigned short* wb = p.w + c.ks * BKR;
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(xdst[i] + slot * STAGE);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" DLWP_RING_LDPOL "\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(xoff[i]), "s"(xb), "s"(dst) : "memory");
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(wdst[i] + slot * STAGE);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" DLWP_RING_LDPOL "\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(woff[i]), "s"(wb), "s"(dst) : "memory");
    }
  };

  f32x4 acc[4][TN];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cursor cc, cl;
  cc.slab = li / tiles_n; cc.nt = li % tiles_n; cc.ks = 0; cc.left = my_tiles - 1;
  cl = cc;
  rows_of(cl);
#pragma unroll
  for (int s0 = 0; s0 < S - 1; ++s0) {       // stages 0 .. S-2 (past the end: the last tile again, results unused)
    dma(s0, cl);
    if (advance(cl)) rows_of(cl);
  }
  // operand fragments: row j of a 16-row block, chunk 4 h + g (k-half h), at its swizzled position
  const int fsw = (j >> 1) & 7;
  const unsigned char* xfrag = smem + (wm * 64 + j) * ROWB;
  const unsigned char* wfrag = smem + XBYTES + (wn * (BN / 2) + j) * ROWB;
  const int ch0 = (g ^ fsw) << 4, ch1 = ((4 + g) ^ fsw) << 4;

  // ---- epilogue operands, requested DL steps ahead: clamped addresses, always EL loads per wave
  const int DL = nk - 1 < 2 ? nk - 1 : 2;
  f32x4 bv[TN], rs[RES ? 4 : 1][TN];
  const float* zero4 = p.bias ? p.bias : p.x;       // (a valid 16-byte aligned address for the loads of an absent bias: values unused)
  auto epi_request = [&]() {
    const long long mbase = row0(cc) + wm * 64 + j;
    const int nbase = cc.nt * BN + wn * (BN / 2) + 4 * g;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int ncl = nbase + 16 * b < p.N ? nbase + 16 * b : p.N - 4;
      const float* src = p.bias ? p.bias + ncl : zero4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[b]) : "v"(src) : "memory");
    }
    if constexpr (RES) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const long long m = mbase + 16 * a < p.M ? mbase + 16 * a : p.M - 1;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int ncl = nbase + 16 * b < p.N ? nbase + 16 * b : p.N - 4;
          const float* src = p.resid + m * p.N + ncl;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rs[a][b]) : "v"(src) : "memory");
        }
      }
    }
  };
  // output buffer descriptor: raw buffer, range-checked against the tensor's bytes (stores beyond are dropped by the hardware)
  const unsigned long long out_bytes = (unsigned long long)p.M * p.N * (OB16 ? 2 : 4);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(out_bytes < 0xFFFFFFF0ull ? out_bytes : 0xFFFFFFF0ull), 0x00020000);
  auto epilogue = [&]() __attribute__((always_inline)) {
    const long long mbase = row0(cc) + wm * 64 + j;
    const int nbase = cc.nt * BN + wn * (BN / 2) + 4 * g;
    // the operands requested DL steps ago: younger in the queue are the DMAs of the steps since (this step's included)
    wait_vmcnt_dyn(GD * DL);
#pragma unroll
    for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(bv[b]));
    if constexpr (RES) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(rs[a][b]));
    }
#if !DLWP_RING_PATCH
    {
      const long long mbase = row0(cc) + wm * 64 + j;
      const int nbase = cc.nt * BN + wn * (BN / 2) + 4 * g;
#pragma unroll
      for (int a = 0; a < 4; a += 2) {
        const long long m_u = mbase + 16 * a, m_v = m_u + 16;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          f32x4 u = acc[a][b], v = acc[a + 1][b];
          if (p.bias) { u += bv[b]; v += bv[b]; }
          if (p.act) gelu_erf8_fma(u, v);
          if constexpr (RES) {
            u += rs[a][b];
            v += rs[a + 1][b];
          }
          const int n = nbase + 16 * b;
          const bool okn = n < p.N;
          const unsigned off_u = (okn && m_u < p.M) ? (unsigned)((m_u * p.N + n) * (OB16 ? 2 : 4)) : 0xFFFFFFF0u;
          const unsigned off_v = (okn && m_v < p.M) ? (unsigned)((m_v * p.N + n) * (OB16 ? 2 : 4)) : 0xFFFFFFF0u;
          if constexpr (OB16) {
            const uint2 pu = uint2{cvt_pk_bf16(u[0], u[1]), cvt_pk_bf16(u[2], u[3])}, pv = uint2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
            asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(pu), "v"(off_u), "s"(orsrc) : "memory");
            asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(pv), "v"(off_v), "s"(orsrc) : "memory");
          } else {
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(u), "v"(off_u), "s"(orsrc) : "memory");
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(v), "v"(off_v), "s"(orsrc) : "memory");
          }
        }
      }
    }
#else
    // Results leave through a 2 KB per-wave LDS patch that turns the accumulator layout (a lane: 4 columns of one row; an
    // instruction: 16 rows x 64 or 32 bytes) into whole 128-byte lines (an instruction: 8 rows x 128 bytes): a quarter (bf16) /
    // half (fp32) of the write requests for the same bytes -- requests, not bytes, are what the vector L1 runs out of.
    unsigned char* patch = smem + S * STAGE + wave * 2048;
    const int rrow = lane >> 3, rchunk = lane & 7;             // read-back role: row (of 8), 16-byte chunk (of 8)

    const long long mrow = row0(cc) + wm * 64;                  // first row of the wave's tile
    const int ncol = cc.nt * BN + wn * (BN / 2);                // first column
#pragma unroll
    for (int a = 0; a < 4; a += 2) {
      f32x4 u[TN], v[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        u[b] = acc[a][b];
        v[b] = acc[a + 1][b];
        if (p.bias) { u[b] += bv[b]; v[b] += bv[b]; }
#ifndef DLWP_KO_GELU
        if (p.act) gelu_erf8_fma(u[b], v[b]);
#endif
        if constexpr (RES) {
          u[b] += rs[a][b];
          v[b] += rs[a + 1][b];
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {                   // row block a + half: rows 16 (a + half) .. + 15 of the wave's tile
        const f32x4 (&w)[TN] = half ? v : u;
        const long long m0 = mrow + 16 * (a + half);
        if constexpr (OB16) {
          // patch = [16 rows][128 bytes]: this lane's 4 bf16 of column block b at row j, byte 2 (16 b + 4 g)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            *reinterpret_cast<uint2*>(patch + j * 128 + 2 * (16 * b + 4 * g)) = uint2{cvt_pk_bf16(w[b][0], w[b][1]), cvt_pk_bf16(w[b][2], w[b][3])};
          asm volatile("" ::: "memory");       // other lanes' writes are read below: one wave's LDS operations complete in order
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const int row = 8 * rb + rrow;
            const u32x4 d = *reinterpret_cast<const u32x4*>(patch + row * 128 + 16 * rchunk);
            const int n = ncol + 8 * rchunk;
            const bool ok = (m0 + row < p.M) && (8 * rchunk < BN / 2) && (n < p.N);     // (N % 4 == 0; bf16 rows: N % 8 asked by the host)
            const unsigned off = ok ? (unsigned)(((m0 + row) * p.N + n) * 2) : 0xFFFFFFF0u;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(d), "v"(off), "s"(orsrc) : "memory");
          }
          asm volatile("" ::: "memory");
        } else {
          // column-block pairs: patch = [16 rows][32 floats]
#pragma unroll
          for (int bp = 0; bp < TN; bp += 2) {
            *reinterpret_cast<f32x4*>(patch + j * 128 + 4 * (4 * g)) = w[bp];
            if (bp + 1 < TN) *reinterpret_cast<f32x4*>(patch + j * 128 + 4 * (16 + 4 * g)) = w[bp + 1];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              const int row = 8 * rb + rrow;
              const f32x4 d = *reinterpret_cast<const f32x4*>(patch + row * 128 + 16 * rchunk);
              const int n = ncol + 16 * bp + 4 * rchunk;
              const bool ok = (m0 + row < p.M) && (16 * bp + 4 * rchunk < BN / 2) && (n < p.N);
              const unsigned off = ok ? (unsigned)(((m0 + row) * p.N + n) * 4) : 0xFFFFFFF0u;
              asm volatile("s_waitcnt lgkmcnt(0)\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen" DLWP_RING_STPOL "\n\ts_nop 1" : : "v"(d), "v"(off), "s"(orsrc) : "memory");
            }
            asm volatile("" ::: "memory");
          }
        }
      }
    }
#endif

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // foreign operations (epilogue loads, stores) issued in each of the last S - 1 steps: they are younger than the DMA of the
  // stage a step needs (issued S - 1 steps before it, ahead of that step's foreign operations) and stay out of its wait
  int fo[S - 1];
#pragma unroll
  for (int i = 0; i < S - 1; ++i) fo[i] = 0;
  int slot = 0;                                        // slot of stage t; stage t + S - 1 goes into the slot of stage t - 1
  for (long long t = 0; t < total; ++t) {
    int extra = 0;
#pragma unroll
    for (int i = 0; i < S - 1; ++i) extra += fo[i];
    wait_vmcnt_dyn(NFLY + extra);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* xf = xfrag + slot * STAGE;
    const unsigned char* wf = wfrag + slot * STAGE;
    u32x4 xb[2][4], wa[2][TN];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      xb[0][a] = *reinterpret_cast<const u32x4*>(xf + a * (16 * ROWB) + ch0);
      xb[1][a] = *reinterpret_cast<const u32x4*>(xf + a * (16 * ROWB) + ch1);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      wa[0][b] = *reinterpret_cast<const u32x4*>(wf + b * (16 * ROWB) + ch0);
      wa[1][b] = *reinterpret_cast<const u32x4*>(wf + b * (16 * ROWB) + ch1);
    }
    {
      const int free_slot = slot == 0 ? S - 1 : slot - 1;
      dma(free_slot, cl);
      if (advance(cl)) rows_of(cl);
    }
    int issued = 0;
    if (cc.ks == nk - 1 - DL) {          // this tile's bias / residual rows, DL steps before they are needed
      epi_request();
      issued += EL;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = mfma16x16x32_bf16(wa[h][b], xb[h][a], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
    if (cc.ks == nk - 1) {
      epilogue();
      issued += ES;
    }
    advance(cc);
#pragma unroll
    for (int i = S - 2; i > 0; --i) fo[i] = fo[i - 1];
    fo[0] = issued;
    slot = slot + 1 == S ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// weights [N][K] fp32 -> the two f16 images [N][K] of the f16x3 form (wh = f16(w 2^s), wm' = (w 2^s - wh) * 2^11); scale[0] holds the
// bits of max |w|, scale[1] receives 2^-(11+s)
__global__ __launch_bounds__(256) void linear_pack_f16_kernel(const float* __restrict__ w, unsigned short* __restrict__ h,
                                                              unsigned short* __restrict__ m, long long pairs, float* __restrict__ scale) {
  float osc;
  const float ws = f16x3_weight_scale(reinterpret_cast<const unsigned*>(scale)[0], osc);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale[1] = osc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long long)gridDim.x * 256) {
    const float2 v = reinterpret_cast<const float2*>(w)[i];
    const float a = v.x * ws, b = v.y * ws;                    // exact (a power of two)
    const f16x2v hh = __builtin_convertvector(f32x2{a, b}, f16x2v);
    const f32x2 r = {(a - (float)hh[0]) * 2048.0f, (b - (float)hh[1]) * 2048.0f};
    reinterpret_cast<unsigned*>(h)[i] = __builtin_bit_cast(unsigned, hh);
    reinterpret_cast<unsigned*>(m)[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
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


template <int BN, int NP, bool F16 = false, bool XB16 = false, bool OB16 = false>
static int32_t launch_v2(const Params& p, hipStream_t s) {
  // resident workgroups (2 per CU by LDS) of the CURRENT device, found once per device: hipFuncSetAttribute is per device too.
  // (atomic: concurrent first calls race benignly to the same value)
  static std::atomic<int> slots_of[64];
  constexpr size_t lds = (size_t)(F16 ? 2 : NP) * 8192 + 2 * (size_t)(F16 ? 2 : NP) * BN * 64;
  auto kern = linear_kernel<BN, NP, F16, XB16, OB16>;
  int dev = 0;
  DLWP_HIP_CHECK(hipGetDevice(&dev));
  DLWP_REQUIRE(dev >= 0 && dev < 64, DLWP_ERR_UNSUPPORTED, "linear: device ordinal %d", dev);
  int slots = slots_of[dev].load(std::memory_order_relaxed);
  if (!slots) {
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int cus = 0, per_cu = 0;
    DLWP_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    DLWP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
    DLWP_REQUIRE(per_cu > 0 && cus >= 8, DLWP_ERR_UNSUPPORTED, "linear: kernel does not fit on this device");
    slots = cus * per_cu;
    slots_of[dev].store(slots, std::memory_order_relaxed);
  }
  const long long tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  long long per_xcd = ((tiles_m + 7) / 8) * tiles_n;      // the busiest XCD's tile count
  DLWP_REQUIRE(per_xcd < (1ll << 30) && 3ll * p.wpart * 2 < (1ll << 31) && (long long)BM * p.K * 4 < (1ll << 31),
               DLWP_ERR_UNSUPPORTED, "linear: operand too large for 32-bit tile offsets");
  if (per_xcd > slots / 8) per_xcd = slots / 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * per_xcd)), dim3(256), lds, s, p);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

template <int BN, bool OB16, int S, bool RES>
static int32_t launch_ring(const Params& p, hipStream_t s) {
  static std::atomic<int> slots_of[64];
  constexpr size_t lds = (size_t)S * (BMR * 128 + BN * 128) + 8 * 2048;     // ring + the per-wave store patches
  auto kern = linear_ring_kernel<BN, OB16, S, RES>;
  int dev = 0;
  DLWP_HIP_CHECK(hipGetDevice(&dev));
  DLWP_REQUIRE(dev >= 0 && dev < 64, DLWP_ERR_UNSUPPORTED, "linear: device ordinal %d", dev);
  int slots = slots_of[dev].load(std::memory_order_relaxed);
  if (!slots) {
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int cus = 0, per_cu = 0;
    DLWP_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    DLWP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds));
    DLWP_REQUIRE(per_cu > 0 && cus >= 8, DLWP_ERR_UNSUPPORTED, "linear: kernel does not fit on this device");
    slots = cus * per_cu;
    slots_of[dev].store(slots, std::memory_order_relaxed);
  }
  const long long tiles_m = (p.M + BMR - 1) / BMR, tiles_n = (p.N + BN - 1) / BN;
  long long per_xcd = ((tiles_m + 7) / 8) * tiles_n;
  DLWP_REQUIRE(per_xcd < (1ll << 30) && 3ll * p.wpart * 2 < (1ll << 31) && (long long)BMR * p.K * 4 < (1ll << 31) &&
                   (long long)p.M * p.N * 4 < 0xFFFFFFF0ll,
               DLWP_ERR_UNSUPPORTED, "linear: operand too large for 32-bit tile offsets");
  if (per_xcd > slots / 8) per_xcd = slots / 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * per_xcd)), dim3(512), lds, s, p);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

#ifndef DLWP_LIN_RING_STAGES
#define DLWP_LIN_RING_STAGES 3
#endif

// form 3: fp32-accurate (six bf16 products), 2: fp32-grade f16x3 (three f16 products), 1: bf16 operands
static int32_t launch(const Params& p, int form, hipStream_t s) {
  static const bool ring = !(getenv("DLWP_LINEAR_RING") && atoi(getenv("DLWP_LINEAR_RING")) == 0);      // A/B: 0 = the register-staged kernel for bf16 x as well
  // the ring kernel where its 256-row tiles fill the chip several times over (one 512-thread workgroup per CU); smaller calls
  // (Swin stage 1: 16 384 rows) keep linear_kernel's 128-row tiles at two workgroups per CU
  static const long long ring_min = getenv("DLWP_LINEAR_RING_MIN_TILES") ? atoll(getenv("DLWP_LINEAR_RING_MIN_TILES")) : 1024;
  const long long ring_tiles = ((p.M + 255) / 256) * ((p.N + 127) / 128);
  // (between 512 and 1024 tiles -- two or three rounds of the chip -- the ring pays only where the k loop is long: Pangu layers 2 / 3,
  // 65 536 rows: fc2 (K = 1 536 -> 384) 147 -> 125 us, the output projection (K = 384 -> 384) 55 -> 58; profiles/r03_linear_ring_tiles.txt)
  const bool ring_fill = ring_tiles >= ring_min || (2 * ring_tiles >= ring_min && p.K >= 768);
  if (ring && (form == 4 || form == 6) && p.K % 64 == 0 && p.K >= 3 * 64 && (form == 4 || p.N % 8 == 0) && ring_fill) {
    const bool narrow_r = (p.N % 128) != 0 && (p.N % 128) <= 96 && (p.N % 96 == 0 || p.N < 128);
    constexpr int S = DLWP_LIN_RING_STAGES;
#define DLWP_RING(BN_, OB_) (p.resid ? launch_ring<BN_, OB_, S, true>(p, s) : launch_ring<BN_, OB_, S, false>(p, s))
    if (form == 4) return narrow_r ? DLWP_RING(96, false) : DLWP_RING(128, false);
    return narrow_r ? DLWP_RING(96, true) : DLWP_RING(128, true);
#undef DLWP_RING
  }
  // 96-wide W tiles where 128 would waste a quarter or more of the last tile (N = 96, 192, 288, 576, 1152)
  const bool narrow = (p.N % 128) != 0 && (p.N % 128) <= 96 && (p.N % 96 == 0 || p.N < 128);
  if (form == 3) return narrow ? launch_v2<96, 3>(p, s) : launch_v2<128, 3>(p, s);
  if (form == 2) return narrow ? launch_v2<96, 3, true>(p, s) : launch_v2<128, 3, true>(p, s);
  if (form == 4) return narrow ? launch_v2<96, 1, false, true, false>(p, s) : launch_v2<128, 1, false, true, false>(p, s);   // bf16 x
  if (form == 5) return narrow ? launch_v2<96, 1, false, false, true>(p, s) : launch_v2<128, 1, false, false, true>(p, s);   // bf16 out
  if (form == 6) return narrow ? launch_v2<96, 1, false, true, true>(p, s) : launch_v2<128, 1, false, true, true>(p, s);     // both
  return narrow ? launch_v2<96, 1>(p, s) : launch_v2<128, 1>(p, s);
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

extern "C" int32_t dlwp_linear_pack_f16x3(const float* weight_dev, int32_t out_features, int32_t in_features, void* packed_dev,
                                          void* stream) {
  DLWP_REQUIRE(weight_dev && packed_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  const size_t bytes = dlwp_linear_packed_bytes(out_features, in_features);
  DLWP_REQUIRE(bytes > 0, DLWP_ERR_UNSUPPORTED, "linear: in_features %d must be a multiple of 32, out_features %d of 4",
               in_features, out_features);
  const size_t part = bytes / 3;      // same buffer size as the bf16 images; the third part holds {bits of max |w|, 2^-(11+s)}
  char* b = reinterpret_cast<char*>(packed_dev);
  const long long pairs = (long long)out_features * in_features / 2;
  long long blocks = (pairs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* scale = reinterpret_cast<float*>(b + 2 * part);
  DLWP_HIP_CHECK(hipMemsetAsync(scale, 0, 8, st));
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weight_dev, 2 * pairs,
                     reinterpret_cast<unsigned*>(scale), (const float*)nullptr, 1);
  hipLaunchKernelGGL(lin::linear_pack_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                     weight_dev, reinterpret_cast<unsigned short*>(b), reinterpret_cast<unsigned short*>(b + part), pairs, scale);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

static int32_t linear_run(int form, const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                          float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act, void* stream) {
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
  p.w = reinterpret_cast<const unsigned short*>(packed_dev);
  p.wpart = (long long)(part / 2);
  p.x = x_dev; p.bias = bias_dev; p.resid = resid_dev; p.out = out_dev;
  p.M = rows; p.N = out_features; p.K = in_features; p.act = act;
  p.wscale = form == 2 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed_dev) + 2 * part) : nullptr;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return lin::launch(p, form, s);
}

extern "C" int32_t dlwp_linear_f32(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                                   float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                                   void* stream) {
  return linear_run(3, x_dev, packed_dev, bias_dev, resid_dev, out_dev, rows, in_features, out_features, act, stream);
}

extern "C" int32_t dlwp_linear_bf16(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                                    float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                                    void* stream) {
  return linear_run(1, x_dev, packed_dev, bias_dev, resid_dev, out_dev, rows, in_features, out_features, act, stream);
}

extern "C" int32_t dlwp_linear_f16x3(const float* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                                     float* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                                     void* stream) {
  return linear_run(2, x_dev, packed_dev, bias_dev, resid_dev, out_dev, rows, in_features, out_features, act, stream);
}

// bf16-operand Linear with a bf16 tensor on one side (the hidden activation of a block's MLP in the bf16 form):
//   x_is_bf16:   x_dev is bf16 [rows][in]  (what dlwp_linear_bf16 would round its fp32 input to -- bit-identical result);
//   out_is_bf16: out_dev is bf16 [rows][out], rounded to nearest even after bias / GELU (no residual operand).
// At least one of the two flags; everything else as dlwp_linear_bf16.
extern "C" int32_t dlwp_linear_bf16_io(const void* x_dev, const void* packed_dev, const float* bias_dev, const float* resid_dev,
                                       void* out_dev, int64_t rows, int32_t in_features, int32_t out_features, int32_t act,
                                       int32_t x_is_bf16, int32_t out_is_bf16, void* stream) {
  DLWP_REQUIRE(x_is_bf16 || out_is_bf16, DLWP_ERR_INVALID_ARGUMENT, "linear_bf16_io: neither side is bf16 (that is dlwp_linear_bf16)");
  DLWP_REQUIRE(!(out_is_bf16 && resid_dev), DLWP_ERR_UNSUPPORTED, "linear_bf16_io: a bf16 output takes no residual operand");
  return linear_run(x_is_bf16 ? (out_is_bf16 ? 6 : 4) : 5, reinterpret_cast<const float*>(x_dev), packed_dev, bias_dev, resid_dev,
                    reinterpret_cast<float*>(out_dev), rows, in_features, out_features, act, stream);
}
