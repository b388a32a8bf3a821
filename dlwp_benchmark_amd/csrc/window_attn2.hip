// Window attention, second generation: the 2-D (Swin) case on MI355X (gfx950), built around what bounded the first
// kernel (window_attn.hip): ~170 vector instructions per 32-key x 16-query tile for bias gather, mask, online softmax
// and operand staging against 4 matrix instructions (profiles/r01_f_swin_attention_sq_counters.txt).
//
// Same contract as dlwp_window_attn_{f32,bf16} (include/dlwp_hip.h) for descriptors with bias_mode 0, a 2-D window
// whose longitude extent is a multiple of 16 and no zero padding -- every Swin block of the reference
// (swin_transformer.py:122-154 + :217-251 + the mask build :383-401).  What changed, per score:
//
//  * relative-position bias: read from LDS STRAIGHT INTO THE ACCUMULATOR the QK^T matrix instruction starts from
//    (C-in = bias instead of 0).  With keys on the accumulator rows and 16 queries / 16 keys consecutive along
//    longitude, the bias tile is Toeplitz: lane (query j, key group g) needs 4 CONSECUTIVE entries of the reversed
//    table row, at an address that is (per-lane constant) + (wave-uniform term of the key block).  No vector
//    arithmetic per score;
//  * 0 / -100 shift mask: with the region boundaries on multiples of 16 a 16-key x 16-query tile is either fully masked
//    or not at all -- a wave-uniform decision.  A masked tile contributes exp(-100 + s - max) < 2^-144 * e^(s - max)
//    per key, below half an ulp of the row sum unless a masked logit exceeds the row's unmasked maximum by more than 83:
//    it is SKIPPED (matrix work, exponentials, staging).  Shifted blocks do a quarter of the work;
//  * softmax: no running maximum.  A reference offset m~ per query (a bf16-exact value near the maximum of the first
//    key tile the query sees) rides in a spare k-slot of the padded head dimension: K carries a column of ones, Q
//    carries -m~, so the matrix instruction delivers s + bias - m~ and the probability is ONE v_exp_f32.  Row sums come
//    from a column of ones in V (a spare output row of the padded head dimension).  Exponent range gives 2^+-100 of
//    slack around m~; the row sum is checked after the loop and a workgroup that ever leaves the slack redoes its tile
//    loop with the exact maximum (computed by a max-only pass) -- never silently wrong;
//  * operands: a prep kernel gathers the window token order once per call (roll as index arithmetic), scales q, and
//    writes Q, K, V as bf16 images (one image per bf16 part: 1 for the bf16 form, 3 for the fp32-accurate "bf16x6"
//    form) -- the tile loop stages with 16-byte copies, V is consumed through ds_read_b64_tr_b16 (no transposed copy).
//
// Precision forms: NP = 1 bf16 operands (dlwp_window_attn_bf16), NP = 3 exact three-way bf16 splits of Q, K, V, P with
// six cross products per contraction = fp32-GEMM accuracy (dlwp_window_attn_f32, form 1 / default).
#include "common.hpp"

namespace dlwp {
namespace wattn2 {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Geo {
  int lat, lon;            // token grid = padded grid (no zero padding on this path)
  int wlat, wlon;          // window
  int nlat, nlon;          // windows per dimension
  int sf[2], sb[2];        // forward / backward roll (lat, lon)
  int use_mask;
  int b1[2], b2[2];        // region id along a dim = (p >= b1) + (p >= b2), p in the shifted frame
  int heads, d, C, N, nwin;
  int TR, W2;              // table rows (2wlat-1)(2wlon-1), 2wlon-1
  float qscale;            // qk scale * log2 e
  int io16;                // bf16 form only: qkv and out are bfloat16 tensors (the block's qkv Linear writes bf16, proj reads bf16)
};

struct Images {           // device pointers into the caller's workspace
  unsigned short* q[3];    // [B][heads][nwin][N][DKP]
  unsigned short* k[3];    // [B][heads][nwin][N][DKP]   (column d of part 0 = 1.0)
  unsigned short* v[3];    // [B][heads][nwin][N][DVP]   (column d of part 0 = 1.0 when DVP > d)
  float* rev;              // [heads][TRP]  reversed bias column * log2 e
  int* dest;               // [nwin][N]     output token of window position n after the backward roll
  int* fallbacks;          // [1]           workgroups that left the exponent slack and redid their loop exactly (this call)
  float* kmax2;            // [rows * cq / 64]  per prep wave: max |k|^2 of its 64 / cq consecutive K rows
                           // (DIRECT: [groups], one maximum per (b, head, window), from wattn2_kmax_kernel)
  const unsigned short* qkv16;   // DIRECT: the bfloat16 qkv tensor [B][L][3 C] itself -- no operand images, no prep kernel
  const float* table;            // DIRECT: the relative-position bias table [TR][heads] as the caller holds it
};

__device__ __forceinline__ unsigned short bf16_bits(float x) { return (unsigned short)(cvt_pk_bf16(x, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// ---------------------------------------------------------------------------------------------------------------
// prep: window token order, bf16 operand images, reversed bias table, destination map
// ---------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void wattn2_prep_kernel(const Geo G, const float* __restrict__ qkv,
                                                          const float* __restrict__ table, Images I, int batch,
                                                          int DKP, int DVP, int TRP) {
  const int cq = DKP / 8, cv = DVP / 8, cper = 2 * cq + cv;   // 16-byte chunks per (token, head): Q | K | V
  const long long rows = (long long)batch * G.heads * G.nwin * G.N;
  const long long total = rows * cper;
  const long long L = (long long)G.lat * G.lon;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    // chunk-major inside a tensor so that consecutive threads write consecutive 16-byte chunks of one image
    int which;         // 0 q, 1 k, 2 v
    long long r;       // image row ((b * heads + head) * nwin + win) * N + n
    int c;
    {
      const long long nq = rows * cq;
      if (t < nq) { which = 0; r = t / cq; c = (int)(t % cq); }
      else if (t < 2 * nq) { which = 1; r = (t - nq) / cq; c = (int)((t - nq) % cq); }
      else { which = 2; r = (t - 2 * nq) / cv; c = (int)((t - 2 * nq) % cv); }
    }
    const int n = (int)(r % G.N);
    long long rr = r / G.N;
    const int win = (int)(rr % G.nwin);
    rr /= G.nwin;
    const int head = (int)(rr % G.heads);
    const int b = (int)(rr / G.heads);
    const int ilat = win / G.nlon, ilon = win % G.nlon;   // window_partition order: (lat window, lon window)
    const int zlat = n / G.wlon, zlon = n % G.wlon;
    const int A = ilat * G.wlat + zlat, O = ilon * G.wlon + zlon;         // shifted frame
    const int sa = (A + G.sf[0]) % G.lat, so = (O + G.sf[1]) % G.lon;     // shifted[p] = x[(p + sf) mod dim]
    const long long tok = (long long)sa * G.lon + so;
    const long long eoff = ((long long)b * L + tok) * 3 * G.C + (long long)which * G.C + head * G.d;
    const float* src = qkv + eoff;
    float x[8];
    const int e0 = 8 * c;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (e0 < G.d) {   // d is a multiple of 8: a chunk is entirely inside or outside the head dimension
      if (G.io16) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(qkv) + eoff + e0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[2 * i] = __uint_as_float(raw[i] << 16);
          x[2 * i + 1] = __uint_as_float(raw[i] & 0xFFFF0000u);
        }
      } else {
        const float4 lo = *reinterpret_cast<const float4*>(src + e0), hi = *reinterpret_cast<const float4*>(src + e0 + 4);
        x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
      }
      if (which == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] *= G.qscale;
      }
    } else if (e0 == G.d && which != 0) {
      x[0] = 1.0f;   // the column of ones: K -> carries -m~ of Q's spare slot; V -> row sums
    }
    if (G.use_mask && __builtin_amdgcn_readfirstlane(which) == 1) {
      // |k|^2 of the row (its cq chunks sit in cq adjacent lanes), then the wave's maximum, stored per wave: a wave covers
      // 64 / cq consecutive K rows of ONE (b, head, window) group (N is a multiple of 64; the K range starts on a wave
      // boundary because rows * cq is a multiple of 64).  The attention kernel reduces its group's N cq / 64 partial
      // maxima in its prologue: the bound that decides whether masked key blocks may be skipped.
      float n2 = 0.f;
      if (e0 < G.d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) n2 += x[i] * x[i];
      }
      for (int m = 1; m < cq; m <<= 1) n2 += __shfl_xor(n2, m);
      float wmax = n2;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, m));
      if ((threadIdx.x & 63) == 0) I.kmax2[(t - rows * cq) >> 6] = wmax;
    }
    u32x4 part[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NP == 3) {
        unsigned hh, mm, ll;
        split3_pair(x[2 * i], x[2 * i + 1], hh, mm, ll);
        part[0][i] = hh; part[1][i] = mm; part[2][i] = ll;
      } else {
        part[0][i] = cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
      }
    }
    const int ld = which == 2 ? DVP : DKP;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      unsigned short* img = which == 0 ? I.q[p] : (which == 1 ? I.k[p] : I.v[p]);
      *reinterpret_cast<u32x4*>(img + r * ld + 8 * c) = part[p];
    }
  }
  // reversed, log2e-scaled bias columns and the destination map (small; the first blocks do them)
  const float LOG2E = 1.4426950408889634f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *I.fallbacks = 0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)G.heads * TRP; t += (long long)gridDim.x * 256) {
    const int head = (int)(t / TRP), x = (int)(t % TRP);
    I.rev[t] = x < G.TR ? table[(long long)(G.TR - 1 - x) * G.heads + head] * LOG2E : 0.f;
  }
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)G.nwin * G.N; t += (long long)gridDim.x * 256) {
    const int win = (int)(t / G.N), n = (int)(t % G.N);
    const int ilat = win / G.nlon, ilon = win % G.nlon;
    const int A = ilat * G.wlat + n / G.wlon, O = ilon * G.wlon + n % G.wlon;
    const int da = (A + G.sb[0]) % G.lat, dq = (O + G.sb[1]) % G.lon;   // out[(p + sb) mod dim] = attn_shifted[p]
    I.dest[t] = da * G.lon + dq;
  }
}

// DIRECT form, shifted blocks only: max |k|^2 over the keys of every (b, head, window) group, straight from the bf16 qkv tensor
// (12 MB at Swin stage 0 against the 50 MB of images the prep kernel writes).  grid = (groups, N / 256), one key per thread with
// all of its loads in flight at once; block (group, y) leaves its partial maximum in kmax2[group * (N / 256) + y] (the attention
// kernel reduces the N / 256 partials of its group).  Block (0, 0) also resets the call's fallback counter.
__global__ __launch_bounds__(256) void wattn2_kmax_kernel(const Geo G, Images I) {
  __shared__ float s_m[4];
  const int group = blockIdx.x;
  if (group == 0 && blockIdx.y == 0 && threadIdx.x == 0) *I.fallbacks = 0;
  const int win = group % G.nwin, head = (group / G.nwin) % G.heads, b = group / (G.nwin * G.heads);
  const int ilat = win / G.nlon, ilon = win % G.nlon;
  const long long L = (long long)G.lat * G.lon;
  const int n = blockIdx.y * 256 + threadIdx.x;          // N is a multiple of 64 (make_plan), the grid covers it in 256s
  float n2 = 0.f;
  if (n < G.N) {
    int sa = ilat * G.wlat + n / G.wlon + G.sf[0], so = ilon * G.wlon + n % G.wlon + G.sf[1];
    sa -= sa >= G.lat ? G.lat : 0;
    so -= so >= G.lon ? G.lon : 0;
    const unsigned short* src = I.qkv16 + ((long long)b * L + (long long)sa * G.lon + so) * 3 * G.C + G.C + head * G.d;
    u32x4 raw[6];                                        // head_dim <= 48 (make_plan)
#pragma unroll
    for (int c = 0; c < 6; ++c) raw[c] = 8 * c < G.d ? *reinterpret_cast<const u32x4*>(src + 8 * c) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(raw[c][i] << 16), hi = __uint_as_float(raw[c][i] & 0xFFFF0000u);
        n2 += lo * lo + hi * hi;
      }
  }
  float m = n2;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) I.kmax2[(long long)group * gridDim.y + blockIdx.y] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// attention
//   DK  k-steps of 32 over the padded head dimension (DKP = 32 DK > d: the spare slot d carries -m~ / ones)
//   DB  16-row blocks of O^T (DVP = 16 DB >= d); ONES = DVP > d: row d of O^T is the row sum
//   SUB 16-query sub-tiles per wave; a workgroup = 4 waves = 64 SUB consecutive window positions
//   NP  bf16 parts per operand (1: bf16 form, 3: bf16x6 = fp32-accurate)
// ---------------------------------------------------------------------------------------------------------------
__device__ constexpr int kTA[6] = {2, 0, 1, 1, 0, 0};   // bf16x6 terms, smallest first: (A part, B part)
__device__ constexpr int kTB[6] = {0, 2, 1, 0, 1, 0};

//   DIRECT (NP = 1 with bfloat16 tensors, round 3): Q, K and V are gathered from the qkv tensor the block's qkv Linear wrote --
//   the window order (roll included) through an LDS token map, the spare-slot constants formed in registers, the bias slice
//   from the caller's table -- so that no prep kernel and no operand images exist (20 us per Swin stage-0 block, 50 MB)
template <int D8, int SUB, int NP, bool MASK, bool DIRECT = false>
__global__ __launch_bounds__(256, NP == 3 ? 2 : 4) void wattn2_kernel(const Geo G, const Images I, float* __restrict__ out,
                                                                       int groups, int nqb, int rev_floats) {
  static_assert(!DIRECT || NP == 1, "the direct form reads bf16 tensors: one part per operand");
  extern __shared__ __align__(16) float smem[];
  constexpr int KT = 32;
  constexpr int d = 8 * D8;                                         // head dimension (compile time)
  constexpr int DK = d / 32 + 1, DB = (d + 15) / 16;
  constexpr int DKP = 32 * DK, DVP = 16 * DB;
  // LDS row strides (bf16 elements), chosen bank-conflict free for the two operand reads: K rows are read with
  // ds_read_b128 by lanes (row j, 16-byte slot g) -> stride/8 dwords-of-4 must map the 16 lanes of each b128 lane group
  // to distinct slots: DKP + 16 (48, 80) does; V rows are read with ds_read_b64_tr_b16 by (row 4g + q, 8-byte slot p)
  // -> stride/2 dwords = 8 mod 16 puts the 8 rows of a 32-lane half on disjoint 8-dword ranges: 16, 48, 48, 80
  constexpr int LDK = DKP + 16, LDV = DVP == 16 ? 16 : (DVP <= 48 ? 48 : 80);
  constexpr int TILE_U16 = NP * KT * (LDK + LDV);                   // one K + V tile (all parts)
  constexpr int TILE_F = (((TILE_U16 + 1) / 2) + 3) & ~3;
  constexpr int LDO = DVP + 1;
  constexpr unsigned ALLSUB = (1u << SUB) - 1u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int N = G.N;
  constexpr bool ONES = DVP > d;

  // ---- workgroup -> (group = (b, head, window), query block): the query blocks of one group share K / V, so they
  // are dealt to ONE XCD (workgroups go round-robin over the 8 XCDs by linear id: id % 8 picks the XCD)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, rloc = bid >> 3;
  const int group = (rloc / nqb) * 8 + xcd, qblk = rloc % nqb;
  if (group >= groups) return;
  const int win = group % G.nwin;
  const int head = (group / G.nwin) % G.heads;
  const int ilat = win / G.nlon, ilon = win % G.nlon;
  const int ntile = N / KT;

  // the slice of the reversed bias column this workgroup can touch: its queries span nq_rows latitude rows, every key
  // row is wlat - 1 .. 0 rows away -> (wlat + nq_rows - 1) table rows, contiguous in the reversed column
  const int TRP = (G.TR + 3) & ~3;
  const int qrow0 = (qblk * (64 * SUB)) / G.wlon;
  const int nq_rows = (64 * SUB + G.wlon - 1) / G.wlon;
  const int rev_lo = (G.TR - (qrow0 + G.wlat + nq_rows - 1) * G.W2) & ~3;   // 16-byte aligned start (>= 0)
  float* s_rev = smem;                                               // [rev_floats] slice of the reversed bias column * log2 e
  float* s_tiles = s_rev + rev_floats;                               // 2 tile buffers
  int* s_flag = reinterpret_cast<int*>(s_tiles + 2 * TILE_F);        // [4]: redo vote, number of needed tiles
  int4* s_tinfo = reinterpret_cast<int4*>(s_flag + 4);               // [ntile] {bias term block 0, block 1, regions, tile}
  int* s_list = reinterpret_cast<int*>(s_tinfo + ntile);             // [ntile] needed tiles of this workgroup, in order
  int* s_src = s_list + ntile;                                       // DIRECT: [N] qkv row offset of window position n (forward roll applied)
  // DIRECT: [64 SUB] output token of the workgroup's queries, behind everything the epilogue's O tile overlays (launch(): lds)
  int* s_dst;
  {
    const int loop_f = rev_floats + 2 * TILE_F + 4 + 4 * ntile + ntile + N, epi_f = 64 * SUB * (16 * ((8 * D8 + 15) / 16) + 1);
    s_dst = reinterpret_cast<int*>(smem) + (loop_f > epi_f ? loop_f : epi_f);
  }
  if constexpr (DIRECT) {
    const float LOG2E = 1.4426950408889634f;
    // eight loads in flight per thread (a load-then-store loop would pay one L2 round trip per element: ~17 in a row at Swin stage 0)
    for (int i0 = tid; i0 < rev_floats; i0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = rev_lo + i0 + 256 * u;
        v[u] = (i0 + 256 * u < rev_floats && x < G.TR) ? I.table[(long long)(G.TR - 1 - x) * G.heads + head] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 256 * u < rev_floats) s_rev[i0 + 256 * u] = v[u] * LOG2E;
    }
    // token (times the row length 3 C of the qkv tensor) of window position n, forward roll applied; one division per thread,
    // then n += 256 as (zlat, zlon) += (256 / wlon, 256 % wlon) with a carry
    const int step_lat = 256 / G.wlon, step_lon = 256 % G.wlon;
    int zlat = tid / G.wlon, zlon = tid % G.wlon;
    for (int n = tid; n < N; n += 256) {
      int sa = ilat * G.wlat + zlat + G.sf[0], so = ilon * G.wlon + zlon + G.sf[1];
      sa -= sa >= G.lat ? G.lat : 0;
      so -= so >= G.lon ? G.lon : 0;
      s_src[n] = (sa * G.lon + so) * (3 * G.C);
      zlon += step_lon;
      zlat += step_lat + (zlon >= G.wlon ? 1 : 0);
      zlon -= zlon >= G.wlon ? G.wlon : 0;
    }
    if (tid < 64 * SUB) {   // output token of the workgroup's query positions: out[(p + sb) mod dim] = attn_shifted[p]
      const int qn = qblk * (64 * SUB) + tid;
      int da = ilat * G.wlat + qn / G.wlon + G.sb[0], dq = ilon * G.wlon + qn % G.wlon + G.sb[1];
      da -= da >= G.lat ? G.lat : 0;
      dq -= dq >= G.lon ? G.lon : 0;
      s_dst[tid] = da * G.lon + dq;
    }
  } else {
  // (entries past the table are ZEROED, not skipped: the bound of the masked-block skip takes the maximum over the whole slice, and
  // whatever an earlier kernel left in LDS there could flip a workgroup into its exact fallback -- correct, but not reproducible)
  for (int i = tid; i < rev_floats / 4; i += 256)
    reinterpret_cast<float4*>(s_rev)[i] = rev_lo + 4 * i < TRP
        ? reinterpret_cast<const float4*>(I.rev + (long long)head * TRP + rev_lo)[i] : float4{0.f, 0.f, 0.f, 0.f};
  }
  auto region_of = [&](int n0) {   // region id of the 16 consecutive window positions starting at n0 (one latitude row)
    const int A = ilat * G.wlat + n0 / G.wlon, O = ilon * G.wlon + n0 % G.wlon;
    return ((A >= G.b1[0]) + (A >= G.b2[0])) * 3 + (O >= G.b1[1]) + (O >= G.b2[1]);
  };
  for (int t = tid; t < ntile; t += 256) {
    int4 ti;
    const int k0 = t * KT, k1 = k0 + 16;
    ti.x = (k0 / G.wlon) * G.W2 + k0 % G.wlon;      // wave-uniform bias term of each 16-key block
    ti.y = (k1 / G.wlon) * G.W2 + k1 % G.wlon;
    ti.z = MASK ? (region_of(k0) | (region_of(k1) << 8)) : 0;
    ti.w = t;
    s_tinfo[t] = ti;
  }
  if (tid == 0) { s_flag[0] = 0; s_flag[2] = 0; s_flag[3] = 0; }
  __syncthreads();
  if (tid == 0) {
    // needed tiles: any of the workgroup's 4 SUB query sub-tiles shares a region with one of the tile's key blocks
    unsigned wgmask = 0;
    for (int c = 0; c < 4 * SUB; ++c) wgmask |= 1u << region_of(qblk * (64 * SUB) + 16 * c);
    int n = 0;
    for (int t = 0; t < ntile; ++t) {
      const int rz = s_tinfo[t].z;
      if (!MASK || ((wgmask >> (rz & 0xFF)) & 1u) || ((wgmask >> (rz >> 8)) & 1u)) s_list[n++] = t;
    }
    s_flag[1] = n;
  }

  const long long img_row0 = (long long)group * N;   // images are indexed [(b, head, win)][n]: group IS that index
  const unsigned short* qkv_b = DIRECT ? I.qkv16 + (long long)(group / (G.nwin * G.heads)) * G.lat * G.lon * (3 * G.C) : nullptr;
  // 16-query chunk c of the workgroup's block goes to (wave c % 4, sub c / 4): a wave's sub-tiles are 64 positions apart,
  // i.e. in the same longitude band on consecutive latitude rows when the window is 64 wide -> usually ONE region
  auto chunk_q0 = [&](int sub) { return qblk * (64 * SUB) + 16 * (4 * sub + wave); };

  // ---- Q operand (B of S^T = K Q^T): lane (j, g) holds head dims 8g .. 8g+7 of query j, per k-step and part
  u32x4 qb[SUB][DK][NP];
  int aq[SUB];        // element index into s_rev of this lane's (query, key group g) at bias term 0
  int qreg[SUB];      // region id of the sub-tile's queries (wave-uniform)
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub) {
    const int q0 = chunk_q0(sub), qn = q0 + j;
#pragma unroll
    for (int ks = 0; ks < DK; ++ks)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if constexpr (DIRECT) {
          // q * (scale log2 e), rounded to bf16 again: what the prep kernel writes for a bfloat16 qkv tensor
          u32x4 qv = {0u, 0u, 0u, 0u};
          if (32 * ks + 8 * g < d) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qkv_b + (unsigned)(s_src[qn] + head * d + 32 * ks + 8 * g));
#pragma unroll
            for (int i = 0; i < 4; ++i)
              qv[i] = cvt_pk_bf16(__uint_as_float(raw[i] << 16) * G.qscale, __uint_as_float(raw[i] & 0xFFFF0000u) * G.qscale);
          }
          qb[sub][ks][p] = qv;
        } else {
          qb[sub][ks][p] = *reinterpret_cast<const u32x4*>(I.q[p] + (img_row0 + qn) * DKP + 32 * ks + 8 * g);
        }
      }
    const int qlat = qn / G.wlon, qlon = qn % G.wlon;
    aq[sub] = G.TR - 1 - (qlat + G.wlat - 1) * G.W2 - (qlon + G.wlon - 1) + 4 * g - rev_lo;
    qreg[sub] = MASK ? region_of(q0) : 0;
  }
  // where -m~ sits in the Q operand: head-dim slot d -> k-step, lane group (dword 0, low half: d is a multiple of 8)
  constexpr int md_ks = d / 32, md_g = (d % 32) / 8;

  // ---- bound on what a SKIPPED (masked) key could contribute: |q| max|k| + max bias - 100 log2 e  (log2 units)
  float qbound[SUB];
  if (MASK) {
    float bm = -3.0e38f;
    for (int i = tid; i < rev_floats; i += 256) bm = fmaxf(bm, s_rev[i]);     // (s_rev is complete: barrier above)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) bm = fmaxf(bm, __shfl_xor(bm, m));
    if (lane == 0) atomicMax(s_flag + 2, __float_as_int(fmaxf(bm, 0.f)));      // non-negative floats order like ints
    // max |k|^2 over the window's keys: the prep kernel left one partial maximum per 64 / (DKP / 8) rows
    constexpr int RPW = 64 / (DKP / 8);
    float km = 0.f;
    if constexpr (DIRECT) {
      const int nb = (N + 255) / 256;
      for (int i = tid; i < nb; i += 256) km = fmaxf(km, I.kmax2[(long long)group * nb + i]);
    } else
    for (int i = tid; i < N / RPW; i += 256) km = fmaxf(km, I.kmax2[(long long)group * (N / RPW) + i]);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) km = fmaxf(km, __shfl_xor(km, m));
    if (lane == 0) atomicMax(s_flag + 3, __float_as_int(km));
  }
  f32x4 oacc[SUB][DB];
  float lsum[SUB];        // row sums on the vector unit when the head dimension leaves no spare output row
  unsigned inited = 0;    // bit sub: m~ of that sub-tile has been set (wave-uniform)
  auto reset_acc = [&]() {
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      lsum[sub] = 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db) oacc[sub][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto set_mref = [&](int sub, float m) {   // m bf16-exact; writes -m into the spare k-slot of Q (part 0)
    const unsigned nb = (unsigned)bf16_bits(-m);
    const unsigned v = qb[sub][md_ks][0][0];
    qb[sub][md_ks][0][0] = g == md_g ? ((v & 0xFFFF0000u) | nb) : v;
  };
  auto mref_of = [&](int sub) {   // m~ of this lane's query, read back from the spare k-slot (lane group md_g holds it)
    const float mine = -__uint_as_float(qb[sub][md_ks][0][0] << 16);
    return __shfl(mine, 16 * md_g + j);
  };
  reset_acc();

  // ---- staging: 16-byte chunks global -> registers (early) -> LDS (late)
  constexpr int CK = DKP / 8, CV = DVP / 8;
  constexpr int ITEMS = NP * KT * (CK + CV);
  constexpr int NI = (ITEMS + 255) / 256;
  u32x4 stg[NI];
  // DIRECT: what this thread stages, fixed for the kernel -- key row in the tile and element offset in the qkv row
  // (>= 0: 16 bytes of K / V; -1: the chunk with the column of ones; -2: zeros)
  int d_row[NI], d_off[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int i = tid + it * 256;
    const bool isk = i < KT * CK;
    const int iv = isk ? i : i - KT * CK;
    const int c = isk ? iv % CK : iv % CV;
    d_row[it] = isk ? iv / CK : iv / CV;
    d_off[it] = 8 * c < d ? (isk ? 1 : 2) * G.C + head * d + 8 * c : (8 * c == d ? -1 : -2);
  }
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      if (i < ITEMS) {
        const int p = i / (KT * (CK + CV)), ii = i % (KT * (CK + CV));
        if constexpr (DIRECT) {
          // chunk c of key kt KT + r: head dims 8c .. 8c+7 of K / V where they exist, the column of ones in slot d, zeros beyond
          u32x4 v = {0u, 0u, 0u, 0u};
          if (d_off[it] >= 0) v = *reinterpret_cast<const u32x4*>(qkv_b + (unsigned)(s_src[kt * KT + d_row[it]] + d_off[it]));
          else if (d_off[it] == -1) v[0] = 0x3F80u;   // bf16 1.0 in the low half
          stg[it] = v;
          (void)p; (void)ii;
          continue;
        }
        const long long row0 = img_row0 + (long long)kt * KT;
        if (ii < KT * CK) stg[it] = *reinterpret_cast<const u32x4*>(I.k[p] + (row0 + ii / CK) * DKP + 8 * (ii % CK));
        else {
          const int iv = ii - KT * CK;
          stg[it] = *reinterpret_cast<const u32x4*>(I.v[p] + (row0 + iv / CV) * DVP + 8 * (iv % CV));
        }
      }
    }
  };
  auto stage_write = [&](int bsel) {
    unsigned short* base = reinterpret_cast<unsigned short*>(s_tiles + bsel * TILE_F);
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      if (i < ITEMS) {
        const int p = i / (KT * (CK + CV)), ii = i % (KT * (CK + CV));
        if (ii < KT * CK) *reinterpret_cast<u32x4*>(base + (p * KT + ii / CK) * LDK + 8 * (ii % CK)) = stg[it];
        else {
          const int iv = ii - KT * CK;
          *reinterpret_cast<u32x4*>(base + NP * KT * LDK + (p * KT + iv / CV) * LDV + 8 * (iv % CV)) = stg[it];
        }
      }
    }
  };

  // ---- building blocks of a tile
  auto load_k = [&](const unsigned short* s_k, int kb, int ks, u32x4 (&ka)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      ka[p] = *reinterpret_cast<const u32x4*>(s_k + (p * KT + kb * 16 + j) * LDK + 32 * ks + 8 * g);
  };
  auto load_v = [&](const unsigned short* s_v, int db, u32x4 (&va)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // block of 4 keys x 16 dims: lane i of the 16-lane group addresses row i >> 2, columns 4 (i & 3) .. + 3 and gets
      // column i of the 4 rows (ds_read_b64_tr_b16): the A operand of O^T += V^T P^T without a transposed copy
      const unsigned short* v0 = s_v + (p * KT + 4 * g + (j >> 2)) * LDV + 16 * db + 4 * (j & 3);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v0 + 16 * LDV));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      va[p] = u32x4{l2.x, l2.y, h2.x, h2.y};
    }
  };
  auto qk = [&](const u32x4 (&ka)[NP], int sub, int ks, f32x4 c) {
    if constexpr (NP == 3) {
#pragma unroll
      for (int term = 0; term < 6; ++term) c = mfma16x16x32_bf16(ka[kTA[term]], qb[sub][ks][kTB[term]], c);
    } else {
      c = mfma16x16x32_bf16(ka[0], qb[sub][ks][0], c);
    }
    return c;
  };
  auto pack_p = [&](const float (&pe)[2][4], u32x4 (&pb)[NP]) {
    // k-slot jj of lane group g is key 4g + jj (jj < 4) or 16 + 4g + (jj - 4): accumulator rows of the two key blocks
    if constexpr (NP == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned hh, mm, ll;
        split3_pair(pe[i >> 1][2 * (i & 1)], pe[i >> 1][2 * (i & 1) + 1], hh, mm, ll);
        pb[0][i] = hh; pb[1][i] = mm; pb[2][i] = ll;
      }
    } else {
      pb[0] = u32x4{cvt_pk_bf16(pe[0][0], pe[0][1]), cvt_pk_bf16(pe[0][2], pe[0][3]),
                    cvt_pk_bf16(pe[1][0], pe[1][1]), cvt_pk_bf16(pe[1][2], pe[1][3])};
    }
  };
  auto pv = [&](const u32x4 (&va)[NP], const u32x4 (&pb)[NP], f32x4 c) {
    if constexpr (NP == 3) {
#pragma unroll
      for (int term = 0; term < 6; ++term) c = mfma16x16x32_bf16(va[kTA[term]], pb[kTB[term]], c);
    } else {
      c = mfma16x16x32_bf16(va[0], pb[0], c);
    }
    return c;
  };

  // FAST body: every sub-tile of the wave sees both key blocks and has its m~.  No branches: the SUB independent
  // chains (bias read -> QK^T -> exp -> pack -> PV) interleave; K and V operands are read from LDS once per tile.
  auto tile_fast = [&](const unsigned short* s_k, const unsigned short* s_v, int sk0, int sk1) {
    f32x4 sc[SUB][2];
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const float* b0 = s_rev + aq[sub] + sk0;
      const float* b1 = s_rev + aq[sub] + sk1;
      sc[sub][0] = f32x4{b0[0], b0[1], b0[2], b0[3]};     // C-in = bias tile (reversed table: 4 consecutive entries)
      sc[sub][1] = f32x4{b1[0], b1[1], b1[2], b1[3]};
    }
#pragma unroll
    for (int ks = 0; ks < DK; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4 ka[NP];
        load_k(s_k, kb, ks, ka);
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) sc[sub][kb] = qk(ka, sub, ks, sc[sub][kb]);
      }
    u32x4 pb[SUB][NP];
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      float pe[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pe[kb][r] = __builtin_amdgcn_exp2f(sc[sub][kb][r]);   // the offset came through the MFMA
      if constexpr (!ONES) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) lsum[sub] += pe[kb][r];
      }
      pack_p(pe, pb[sub]);
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      u32x4 va[NP];
      load_v(s_v, db, va);
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) oacc[sub][db] = pv(va, pb[sub], oacc[sub][db]);
    }
  };

  // GENERAL body (first tiles of a sub-tile, partly masked tiles, the max-only pass): wave-uniform branches per sub-tile
  float mx_run[SUB];
  bool exact_mask = false;   // workgroup-uniform: include masked key blocks (with -100) instead of skipping them
  auto tile_general = [&](const unsigned short* s_k, const unsigned short* s_v, int sk0, int sk1, unsigned use0, unsigned use1,
                          unsigned msk0, unsigned msk1, auto maxonly_c) {
    constexpr bool MAXONLY = decltype(maxonly_c)::value;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const bool u0 = (use0 >> sub) & 1u, u1 = (use1 >> sub) & 1u;
      if (!(u0 || u1)) continue;
      f32x4 sc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (!(kb == 0 ? u0 : u1)) continue;
        const float* br = s_rev + aq[sub] + (kb == 0 ? sk0 : sk1);
        sc[kb] = f32x4{br[0], br[1], br[2], br[3]};
        if (MASK && (((kb == 0 ? msk0 : msk1) >> sub) & 1u)) {
          // exact mode: a masked block is INCLUDED with the reference's additive -100 (swin_transformer.py:396-401)
          const float mv = -100.0f * 1.4426950408889634f;
          sc[kb] += f32x4{mv, mv, mv, mv};
        }
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
          u32x4 ka[NP];
          load_k(s_k, kb, ks, ka);
          sc[kb] = qk(ka, sub, ks, sc[kb]);
        }
      }
      if (MAXONLY || !((inited >> sub) & 1u)) {
        float mx = -3.0e38f;
        if (u0) mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
        if (u1) mx = fmaxf(mx, fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if constexpr (MAXONLY) {
          mx_run[sub] = fmaxf(mx_run[sub], mx);
          continue;
        } else {
          // first tile this sub-tile sees: m~ = that maximum rounded to bf16; this tile's scores get it on the vector unit
          const float m = bf16_to_f32(bf16_bits(mx));
          set_mref(sub, m);
          inited |= 1u << sub;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[kb][r] -= m;
        }
      }
      if constexpr (!MAXONLY) {
        float pe[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) pe[kb][r] = (kb == 0 ? u0 : u1) ? __builtin_amdgcn_exp2f(sc[kb][r]) : 0.f;
        if constexpr (!ONES) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) lsum[sub] += pe[kb][r];
        }
        u32x4 pb[NP];
        pack_p(pe, pb);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          u32x4 va[NP];
          load_v(s_v, db, va);
          oacc[sub][db] = pv(va, pb, oacc[sub][db]);
        }
      }
    }
  };

  // ---- one pass over the workgroup's needed key tiles (two LDS buffers, one barrier per tile; the global loads of
  // tile i + 2 are in flight during the compute of tile i + 1)
  auto run_pass = [&](auto maxonly_c) {
    constexpr bool MAXONLY = decltype(maxonly_c)::value;
    const int nl = s_flag[1];
    int li = 0, bsel = 0;
    if (nl > 0) {
      stage_load(s_list[0]);
      stage_write(0);
    }
    __syncthreads();
    if (nl > 1) stage_load(s_list[1]);
    for (; li < nl; ++li) {
      const unsigned short* s_k = reinterpret_cast<const unsigned short*>(s_tiles + bsel * TILE_F);
      const unsigned short* s_v = s_k + NP * KT * LDK;
      const int4 ti = s_tinfo[s_list[li]];
      const int sk0 = __builtin_amdgcn_readfirstlane(ti.x), sk1 = __builtin_amdgcn_readfirstlane(ti.y);
      unsigned use0 = ALLSUB, use1 = ALLSUB, msk0 = 0, msk1 = 0;
      if (MASK) {
        const int rz = __builtin_amdgcn_readfirstlane(ti.z);
        unsigned eq0 = 0, eq1 = 0;
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
          eq0 |= (unsigned)((rz & 0xFF) == qreg[sub]) << sub;
          eq1 |= (unsigned)((rz >> 8) == qreg[sub]) << sub;
        }
        if (exact_mask) { msk0 = ~eq0 & ALLSUB; msk1 = ~eq1 & ALLSUB; }   // every block, masked ones with -100
        else { use0 = eq0; use1 = eq1; }                                     // masked blocks skipped
      }
      if (!MAXONLY && inited == ALLSUB && use0 == ALLSUB && use1 == ALLSUB && (msk0 | msk1) == 0) tile_fast(s_k, s_v, sk0, sk1);
      else if ((use0 | use1) != 0) tile_general(s_k, s_v, sk0, sk1, use0, use1, msk0, msk1, maxonly_c);
      if (li + 1 < nl) {
        stage_write(bsel ^ 1);     // every wave finished reading that buffer before the previous barrier
        if (li + 2 < nl) stage_load(s_list[li + 2]);
      }
      bsel ^= 1;
      __syncthreads();
    }
  };

  __syncthreads();   // s_rev, tile table and tile list are ready
  if (MASK) {
    // log2-domain upper bound of any logit of this lane's query: |q| max|k| + max bias (Cauchy-Schwarz; q is pre-scaled)
    const float kmax2 = __int_as_float(s_flag[3]);
    const float bmax = __int_as_float(s_flag[2]);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      float n2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned u = qb[sub][ks][0][w];       // part 0 carries each value to 2^-9: inflate the bound by 1 %
          const float lo = __uint_as_float(u << 16), hi = __uint_as_float(u & 0xFFFF0000u);
          n2 += lo * lo + hi * hi;
        }
      n2 += __shfl_xor(n2, 16);
      n2 += __shfl_xor(n2, 32);
      qbound[sub] = 1.01f * __builtin_sqrtf(n2 * kmax2) + bmax;
    }
  }
  run_pass(std::false_type{});

  // ---- did every query stay inside the exponent slack?  row sum l = O^T row d (ONES) or the vector-unit sum
  auto row_sum = [&](int sub) {
    float l;
    if constexpr (ONES) {
      constexpr int db = d / 16, gg = (d % 16) / 4;     // row d of O^T: block db, lane group gg, register 0
      l = __shfl(oacc[sub][db][0], 16 * gg + j);
    } else {
      l = lsum[sub];
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
    }
    return l;
  };
  {
    bool bad = false;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const float l = row_sum(sub);
      if (!(l > 7.9e-31f && l < 1.2e30f)) bad = true;   // 2^-100 .. 2^100 (NaN fails both)
      if (MASK) {
        // Was skipping the masked blocks exact?  The row's unmasked maximum is >= m~ + log2(l / N); a skipped key
        // contributes at most 2^(qbound - 100 log2 e - that maximum) of the largest term.  Below 2^-30 (N keys: 2^-30 N
        // << 2^-24) it cannot change an fp32 result; otherwise redo with the masked blocks included.
        const float lower = mref_of(sub) + __builtin_amdgcn_logf(l) - __builtin_amdgcn_logf((float)N);
        if (!(qbound[sub] - 144.26950408889634f < lower - 30.0f)) bad = true;
      }
    }
    if (__any(bad)) s_flag[0] = 1;
  }
  __syncthreads();
  if (s_flag[0]) {
    // exact fallback (workgroup-uniform, counted): every key block (masked ones with -100), the exact maximum of every
    // query from a max-only pass, then the pass again
    if (tid == 0 && (!DIRECT || MASK)) atomicAdd(I.fallbacks, 1);
    if (MASK) {
      exact_mask = true;
      if (tid == 0) {
        for (int t = 0; t < ntile; ++t) s_list[t] = t;
        s_flag[1] = ntile;
      }
      __syncthreads();
    }
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      mx_run[sub] = -3.0e38f;
      set_mref(sub, 0.f);            // the max-only pass must see scores WITHOUT an offset
    }
    run_pass(std::true_type{});
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) set_mref(sub, bf16_to_f32(bf16_bits(mx_run[sub])));
    inited = ALLSUB;
    reset_acc();
    run_pass(std::false_type{});
  }

  // ---- epilogue: O^T / l through LDS (reusing the table + tile area), whole rows out through the destination map
  float* s_o = smem;
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub) {
    const float l = row_sum(sub);
    const float inv = 1.0f / l;
    const int ql = 16 * (4 * sub + wave) + j;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_o[ql * LDO + 16 * db + 4 * g + r] = oacc[sub][db][r] * inv;
  }
  __syncthreads();
  const int b = group / (G.nwin * G.heads);
  float* out_b = out + (long long)b * G.lat * G.lon * G.C + head * d;
  constexpr int d4 = d / 4;
  for (int i = tid; i < 64 * SUB * d4; i += 256) {
    const int ql = i / d4, e = 4 * (i % d4);
    const int qn = qblk * (64 * SUB) + ql;
    int dst;
    if constexpr (DIRECT) {
      dst = s_dst[ql];
    } else {
      dst = I.dest[win * N + qn];
    }
    const float* so = s_o + ql * LDO + e;
    if (G.io16) {
      unsigned short* o16 = reinterpret_cast<unsigned short*>(out) + ((long long)b * G.lat * G.lon + dst) * G.C + head * d + e;
      *reinterpret_cast<uint2*>(o16) = uint2{cvt_pk_bf16(so[0], so[1]), cvt_pk_bf16(so[2], so[3])};
    } else {
      *reinterpret_cast<float4*>(out_b + (long long)dst * G.C + e) = float4{so[0], so[1], so[2], so[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
struct Plan {
  bool ok = false;
  Geo G;
  int DK, DB, DKP, DVP, TRP, NP, SUB;
  size_t bytes = 0;
  size_t off_q[3], off_k[3], off_v[3], off_rev, off_dest, off_fb, off_kmax;
  int groups = 0;
};

static Plan make_plan(const dlwp_wattn_desc* u, int batch, int np) {
  Plan P;
  if (!u || batch <= 0) return P;
  if (u->bias_mode != 0 || u->grid[0] != 1 || u->window[0] != 1 || u->padded[0] != 1) return P;
  for (int i = 0; i < 3; ++i)
    if (u->padded[i] != u->grid[i] || u->pad_lead[i] != 0) return P;
  const int lat = u->grid[1], lon = u->grid[2], wlat = u->window[1], wlon = u->window[2];
  if (wlat <= 0 || wlon <= 0 || lat % wlat || lon % wlon || wlon % 16) return P;
  const int N = wlat * wlon, d = u->head_dim;
  if (N % 64 || N > 16384) return P;       // whole 64-query workgroup blocks; tile table + list fit LDS
  if (d != 8 && d != 16 && d != 24 && d != 48) return P;      // instantiated head dims (each leaves a spare k-slot)
  if (u->use_mask) {
    // a 16-key block lies in one latitude row, so latitude boundaries are free; longitude boundaries must not cut it
    for (int b : {u->mask_b1[2], u->mask_b2[2]})
      if (b < lon && b % 16) return P;
  }
  Geo& G = P.G;
  G.lat = lat; G.lon = lon; G.wlat = wlat; G.wlon = wlon; G.nlat = lat / wlat; G.nlon = lon / wlon;
  G.sf[0] = ((u->shift_fwd[1] % lat) + lat) % lat; G.sf[1] = ((u->shift_fwd[2] % lon) + lon) % lon;
  G.sb[0] = ((u->shift_back[1] % lat) + lat) % lat; G.sb[1] = ((u->shift_back[2] % lon) + lon) % lon;
  G.use_mask = u->use_mask;
  G.b1[0] = u->mask_b1[1]; G.b1[1] = u->mask_b1[2]; G.b2[0] = u->mask_b2[1]; G.b2[1] = u->mask_b2[2];
  G.heads = u->heads; G.d = d; G.C = u->heads * d; G.N = N; G.nwin = G.nlat * G.nlon;
  G.W2 = 2 * wlon - 1; G.TR = (2 * wlat - 1) * G.W2;
  G.qscale = u->scale * 1.4426950408889634f;
  G.io16 = 0;
  P.DK = d / 32 + 1; P.DKP = 32 * P.DK;
  P.DB = (d + 15) / 16; P.DVP = 16 * P.DB;
  P.TRP = (G.TR + 3) & ~3;
  P.NP = np;
  P.SUB = (N >= 2048 && N % 256 == 0) ? 4 : ((N >= 256 && N % 128 == 0) ? 2 : 1);   // N % (64 SUB) == 0
  if (d == 48 && P.SUB == 4) P.SUB = 2;     // 64 queries per wave at head_dim 48 do not fit the register file
  const size_t rows = (size_t)batch * G.heads * G.nwin * N;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += align_up(n, 256); return o; };
  for (int p = 0; p < np; ++p) { P.off_q[p] = take(rows * P.DKP * 2); P.off_k[p] = take(rows * P.DKP * 2); P.off_v[p] = take(rows * P.DVP * 2); }
  P.off_rev = take((size_t)G.heads * P.TRP * 4);
  P.off_dest = take((size_t)G.nwin * N * 4);
  P.off_fb = take(256);
  P.groups = batch * G.heads * G.nwin;
  P.off_kmax = take(rows * (P.DKP / 8) / 64 * 4 + 256);
  P.bytes = off;
  P.ok = true;
  return P;
}

template <int D8, int SUB, int NP, bool DIRECT = false>
static int32_t launch(const Plan& P, const Images& I, float* out, int batch, hipStream_t s) {
  constexpr int KT = 32, DK = (8 * D8) / 32 + 1, DB = (8 * D8 + 15) / 16, DKP = 32 * DK, DVP = 16 * DB, LDK = DKP + 16,
                LDV = DVP == 16 ? 16 : (DVP <= 48 ? 48 : 80);
  constexpr int TILE_F = (((NP * KT * (LDK + LDV) + 1) / 2) + 3) & ~3;
  const size_t ntile = (size_t)P.G.N / KT;
  // slice of the reversed bias column a workgroup keeps in LDS: (wlat + rows of its query block - 1) table rows (+ alignment)
  const int nq_rows = (64 * SUB + P.G.wlon - 1) / P.G.wlon;
  const int rev_floats = (((P.G.wlat + nq_rows - 1) * P.G.W2 + 3 + 3) & ~3) + 4;
  const size_t loop_f = (size_t)rev_floats + 2 * TILE_F + 4 + 4 * ntile + ntile + (DIRECT ? (size_t)P.G.N : 0);
  const size_t epi_f = (size_t)64 * SUB * (DVP + 1);
  const size_t lds = (loop_f > epi_f ? loop_f : epi_f) * 4 + (DIRECT ? 64 * SUB * 4 : 0);
  DLWP_REQUIRE(lds <= 160 * 1024, DLWP_ERR_UNSUPPORTED, "window attention needs %zu bytes of LDS (bias table too large)", lds);
  const int groups = batch * P.G.heads * P.G.nwin;
  const int nqb = P.G.N / (64 * SUB);
  const int grid = ((groups + 7) / 8) * 8 * nqb;
  auto kern = P.G.use_mask ? wattn2_kernel<D8, SUB, NP, true, DIRECT> : wattn2_kernel<D8, SUB, NP, false, DIRECT>;
  if (lds > 48 * 1024)
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, P.G, I, out, groups, nqb, rev_floats);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

template <int NP>
static int32_t run_np(const Plan& P_in, const float* qkv, const float* table, float* out, int batch, void* workspace,
                      hipStream_t s, int io16 = 0) {
  Plan P = P_in;
  P.G.io16 = io16;
  Images I = {};
  char* w = reinterpret_cast<char*>(workspace);
  for (int p = 0; p < NP; ++p) {
    I.q[p] = reinterpret_cast<unsigned short*>(w + P.off_q[p]);
    I.k[p] = reinterpret_cast<unsigned short*>(w + P.off_k[p]);
    I.v[p] = reinterpret_cast<unsigned short*>(w + P.off_v[p]);
  }
  I.rev = reinterpret_cast<float*>(w + P.off_rev);
  I.dest = reinterpret_cast<int*>(w + P.off_dest);
  I.fallbacks = reinterpret_cast<int*>(w + P.off_fb);
  I.kmax2 = reinterpret_cast<float*>(w + P.off_kmax);
  if constexpr (NP == 1) {
    // bfloat16 tensors: the attention kernel gathers its operands itself (DLWP_WATTN2_DIRECT=0, read once: the prep kernel + images, A/B)
    static const bool direct = !(getenv("DLWP_WATTN2_DIRECT") && atoi(getenv("DLWP_WATTN2_DIRECT")) == 0);
    if (io16 && direct) {
      I.qkv16 = reinterpret_cast<const unsigned short*>(qkv);
      I.table = table;
      const int groups = batch * P.G.heads * P.G.nwin;
      // shifted blocks: the key-norm pass (it also resets the fallback counter).  Plain blocks launch nothing in front of the attention
      // kernel: their counter is NOT maintained (dlwp_window_attn_fallbacks describes the fp32-tensor entry points; include/dlwp_hip.h)
      if (P.G.use_mask) {
        hipLaunchKernelGGL(wattn2_kmax_kernel, dim3((unsigned)groups, (unsigned)((P.G.N + 255) / 256)), dim3(256), 0, s, P.G, I);
        DLWP_HIP_CHECK(hipGetLastError());
      }
#define DLWP_W2D(D8_)                                                         \
  do {                                                                        \
    if (P.SUB == 4) return launch<D8_, 4, 1, true>(P, I, out, batch, s);      \
    if (P.SUB == 2) return launch<D8_, 2, 1, true>(P, I, out, batch, s);      \
    return launch<D8_, 1, 1, true>(P, I, out, batch, s);                      \
  } while (0)
      if (P.G.d == 8) DLWP_W2D(1);
      if (P.G.d == 16) DLWP_W2D(2);
      if (P.G.d == 24) DLWP_W2D(3);
      if (P.G.d == 48) DLWP_W2D(6);
#undef DLWP_W2D
      return fail(DLWP_ERR_UNSUPPORTED, "window attention fast path: head_dim %d not instantiated", P.G.d);
    }
  }
  const long long chunks = (long long)batch * P.G.heads * P.G.nwin * P.G.N * (2 * P.DKP / 8 + P.DVP / 8);
  long long blocks = (chunks + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(wattn2_prep_kernel<NP>, dim3((unsigned)blocks), dim3(256), 0, s, P.G, qkv, table, I, batch, P.DKP,
                     P.DVP, P.TRP);
  DLWP_HIP_CHECK(hipGetLastError());
#define DLWP_W2(D8_)                                                    \
  do {                                                                   \
    if (P.SUB == 4) return launch<D8_, 4, NP>(P, I, out, batch, s);      \
    if (P.SUB == 2) return launch<D8_, 2, NP>(P, I, out, batch, s);      \
    return launch<D8_, 1, NP>(P, I, out, batch, s);                      \
  } while (0)
  if (P.G.d == 8) DLWP_W2(1);
  if (P.G.d == 16) DLWP_W2(2);
  if (P.G.d == 24) DLWP_W2(3);
  if (P.G.d == 48) DLWP_W2(6);
#undef DLWP_W2
  return fail(DLWP_ERR_UNSUPPORTED, "window attention fast path: head_dim %d not instantiated", P.G.d);
}

}  // namespace wattn2
}  // namespace dlwp

// Entry points used by window_attn.hip's dispatcher
namespace dlwp {
size_t wattn2_workspace_bytes(const dlwp_wattn_desc* u, int batch, int np) {
  const wattn2::Plan P = wattn2::make_plan(u, batch, np);
  return P.ok ? P.bytes : 0;
}
// device address of the call's fallback counter inside the workspace (nullptr: descriptor not covered)
const int* wattn2_fallback_counter(const dlwp_wattn_desc* u, int batch, int np, const void* workspace) {
  const wattn2::Plan P = wattn2::make_plan(u, batch, np);
  if (!P.ok || !workspace) return nullptr;
  return reinterpret_cast<const int*>(reinterpret_cast<const char*>(workspace) + P.off_fb);
}
// returns DLWP_OK, an error, or 1 when this descriptor is not covered (the caller takes the generic kernel)
int32_t wattn2_run(const dlwp_wattn_desc* u, const float* qkv, const float* table, float* out, int batch, void* workspace,
                   size_t workspace_bytes, hipStream_t s, int np) {
  const int io16 = np == 16 ? 1 : 0;        // bf16 form with bfloat16 qkv / output tensors
  if (io16) np = 1;
  const wattn2::Plan P = wattn2::make_plan(u, batch, np);
  if (!P.ok || !workspace || workspace_bytes < P.bytes) return 1;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(qkv) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return 1;
  return np == 3 ? wattn2::run_np<3>(P, qkv, table, out, batch, workspace, s)
                 : wattn2::run_np<1>(P, qkv, table, out, batch, workspace, s, io16);
}
}  // namespace dlwp
