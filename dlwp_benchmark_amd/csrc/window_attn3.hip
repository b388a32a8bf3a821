// Window attention, third kernel: the small 3-D windows of Pangu-Weather's earth-specific attention on MI355X (gfx950).
//
// Same contract as dlwp_window_attn_{f32,bf16} (include/dlwp_hip.h) for descriptors with bias_mode 1 (earth-specific bias,
// panguweather.py:176-211 + utils/earth_position_index.py), head_dim 32, at most two pressure-level planes per window and
// at most 80 tokens per plane -- every block of the reference's Pangu (window (2, 6, 12): 144 tokens, of which one plane
// is zero padding: the model has ONE pressure level, panguweather.py:285-316 pads it to the window).  The generic kernel
// (window_attn.hip) streams 32-key tiles with an online softmax and decodes coordinates per score; here a window is small
// enough that ALL scores of 16 queries (<= 160 keys) sit in one wave's accumulators:
//
//  * one workgroup = (batch, window latitude row, head, query plane): the earth bias column of its (type, head) -- 13 KB,
//    the same for all longitude windows of the row -- is staged ONCE and the workgroup walks the row's windows; K / V / Q
//    of the next window are in flight (registers) while the current one computes;
//  * keys are re-ordered plane by plane, each plane padded to whole 16-key blocks with duplicates of its first key whose
//    V row and ones column are zero (they can neither raise the maximum nor add to a sum): a 16-key block never straddles
//    planes, so the pressure-level part of the 0 / -100 shift mask is one constant per block;
//  * the bias index is separable, idx = a(query) + b(key): the accumulator the QK^T matrix instruction starts from is
//    four LDS reads at a_q + b_k -- no coordinate decode per score;
//  * exact softmax without a running maximum: all (<= 10) score tiles of the 16 queries are in registers, the row maximum is
//    taken once, the probabilities are ONE v_exp_f32 each, the row sum comes out of the PV product (ones column in V);
//  * only queries whose destination is a real token are computed: one plane of the two (the other plane's outputs are
//    cropped away by the reference, panguweather.py:312-316), i.e. half the score matrix.
//
// NP = 1: bf16 operands (dlwp_window_attn_bf16); NP = 3: exact three-way bf16 splits of Q, K, V, P with six cross products
// per contraction = fp32-GEMM accuracy (dlwp_window_attn_f32).
#include "common.hpp"

namespace dlwp {
namespace wattn3 {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Geo {
  int pl, lat, lon;           // un-padded token grid
  int ppl, plat, plon;        // padded grid
  int pad_f, pad_t, pad_l;    // leading pads
  int wpl, wlat, wlon;        // window
  int npl, nlat, nlon;        // windows per dimension
  int sf[3], sb[3];           // forward / backward roll (window_attn.hip: Desc)
  int b1[3], b2[3];           // region id along a dim = (p >= b1) + (p >= b2), p in the shifted frame
  int heads, C;
  int TR, TRP, W2;            // bias rows, padded rows, 2 wlon - 1
  int nch, chunk;             // longitude windows are walked in nch chunks of `chunk` windows (one workgroup each)
  float qscale;               // qk scale * log2 e
  // the windows a latitude / longitude region boundary cuts (MODE 2 walks exactly these, one per workgroup)
  int ncr, cr[2], ncc, cc[2], ncut;
};

constexpr int D = 32;          // head dim
constexpr int LDK = 48;        // bf16 elements per K row in LDS (96 bytes: conflict free for the b128 operand reads)
constexpr int LDV = 48;        // V row: 32 dims + the ones column + 15 zeros (three 16-dim output blocks)
__device__ constexpr int kTA[6] = {2, 0, 1, 1, 0, 0};   // bf16x6 terms, smallest first: (A part, B part)
__device__ constexpr int kTB[6] = {0, 2, 1, 0, 1, 0};

// transposed + scaled bias table: [types][heads][TRP] = table[row][type][head] * log2 e
__global__ __launch_bounds__(256) void wattn3_table_kernel(const float* __restrict__ table, float* __restrict__ tabT, int rows,
                                                           int rows_p, int types, int heads) {
  const int th = blockIdx.x;                      // type * heads + head
  for (int i = threadIdx.x; i < rows_p; i += 256)
    tabT[(long long)th * rows_p + i] = i < rows ? table[(long long)i * types * heads + th] * 1.4426950408889634f : 0.f;
}

// RP real planes (staged per window) + PP (0 / 1) plane that is zero padding as a whole: its tokens all carry the qkv
// bias, so its K rows are ONE shared 16-row block and its V rows two (a full block and the partial last one), written
// once per workgroup.  Key blocks: the real planes' first, then the padding plane's.
//
// Shifted blocks run as TWO launches over complementary window sets: MODE 1 takes the windows that no latitude / longitude
// region boundary cuts (all but the last window row / column; every window of an unshifted block): they have ONE region
// per plane, so the 0 / -100 term is a constant per key block; MODE 2 takes the cut windows, one per workgroup, with a
// region compare per score.  One kernel doing both needed 156 VGPRs for the sake of 9 % of the windows.
// (<= 128 VGPRs, i.e. four waves per SIMD, matters even where LDS admits only two workgroups per CU: the NP = 3 kernel at 130
// VGPRs ran 35 % slower than at 126.)
// IO16 (bf16 form only, round 3): qkv, the qkv bias and the output are bfloat16 tensors -- the qkv Linear of a block in the bf16
// form writes bf16 (dlwp_linear_bf16_io) and proj reads bf16, so the widest tensor of the block (qkv: 604 MB per layer-1 call
// of C5 in fp32) crosses HBM at half the bytes in both directions; K and V chunks go to LDS as they are.
template <int RP, int PP, int PB, int NP, int MODE, bool IO16 = false>
__global__ __launch_bounds__(64 * PB) __attribute__((amdgpu_waves_per_eu(PP == 1 ? 4 : 3))) void wattn3_kernel(const Geo G, const float* __restrict__ qkv,
                                                          const float* __restrict__ qkv_bias, const float* __restrict__ tabT,
                                                          float* __restrict__ out, long long L) {
  static_assert(!IO16 || NP == 1, "bf16 tensors go with the bf16 form");
  using KV = std::conditional_t<IO16, uint2, float4>;          // one staged chunk: 4 head dims of a K or V row
  const unsigned short* qkv16 = reinterpret_cast<const unsigned short*>(qkv);
  const unsigned short* bias16 = reinterpret_cast<const unsigned short*>(qkv_bias);
  constexpr bool MASK = true;
  constexpr int WPL = RP + PP;
  constexpr int KB = WPL * PB;                    // 16-key blocks
  constexpr int NS = 16 * KB;                     // logical key slots (plane-major, each plane padded to 16 PB)
  constexpr int KR = RP * 16 * PB + PP * 16;      // K rows in LDS
  constexpr int VR = RP * 16 * PB + PP * 32;      // V rows in LDS
  constexpr int NT = 64 * PB;                     // threads
  constexpr int CPT = 2 * RP;                     // 16-byte fp32 chunks of K (and of V) a thread stages per window
  constexpr int KS2 = (KB + 1) / 2;               // 32-key k-steps of the PV product
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* s_tab = reinterpret_cast<float*>(smem_raw);                                   // [TRP]
  unsigned short* s_k = reinterpret_cast<unsigned short*>(s_tab + G.TRP);              // [NP][KR][LDK]
  unsigned short* s_v = s_k + NP * KR * LDK;                                           // [NP][VR][LDV]
  int* s_bkey = reinterpret_cast<int*>(s_v + NP * VR * LDV);                           // [NS] key part of the bias index
  int* s_rkey = s_bkey + NS;                                                           // [NS] shift-mask region of the slot

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // query block of the plane
  const int j = lane & 15, g = lane >> 4;
  const int head = blockIdx.x;
  int t = blockIdx.y, zq, ilat, ipl, b, lon0, lon1;
  if constexpr (MODE == 2) {
    // one cut window per workgroup: the cut rows' windows first, then the cut columns of the other rows
    int idx = t % G.ncut; t /= G.ncut;
    zq = t % WPL; t /= WPL;
    ipl = t % G.npl;
    b = t / G.npl;
    if (idx < G.ncr * G.nlon) {
      ilat = G.cr[idx / G.nlon];
      lon0 = idx % G.nlon;
    } else {
      idx -= G.ncr * G.nlon;
      lon0 = G.cc[idx % G.ncc];
      ilat = idx / G.ncc;                         // index among the rows that are not cut (cr ascending)
      if (G.ncr > 0 && ilat >= G.cr[0]) ++ilat;
      if (G.ncr > 1 && ilat >= G.cr[1]) ++ilat;
    }
    lon1 = lon0 + 1;
  } else {
    const int chunk = t % G.nch; t /= G.nch;
    zq = t % WPL; t /= WPL;
    ilat = t % G.nlat; t /= G.nlat;
    ipl = t % G.npl;
    b = t / G.npl;
    lon0 = chunk * G.chunk;
    lon1 = lon0 + G.chunk < G.nlon ? lon0 + G.chunk : G.nlon;
  }
  const int NPQ = G.wlat * G.wlon;                // tokens per plane
  const int C = G.C;

  // the query plane's outputs land on pressure level (P + sb) mod ppl - pad: nothing to do when that is padding
  const int dlev = (ipl * WPL + zq + G.sb[0]) % G.ppl - G.pad_f;
  if (dlev < 0 || dlev >= G.pl) return;

  // the windows of this launch's kind
  auto mine = [&](int ilon) -> bool {
    if constexpr (MODE != 1) return true;
    auto inside = [&](int bb, int lo, int n) { return bb > lo && bb < lo + n; };      // boundary strictly inside [lo, lo + n)
    const bool cut = inside(G.b1[1], ilat * G.wlat, G.wlat) || inside(G.b2[1], ilat * G.wlat, G.wlat) ||
                     inside(G.b1[2], ilon * G.wlon, G.wlon) || inside(G.b2[2], ilon * G.wlon, G.wlon);
    return !cut;
  };
  auto next_win = [&](int ilon) { while (ilon < lon1 && !mine(ilon)) ++ilon; return ilon; };

  int ilon = next_win(lon0);
  if (ilon >= lon1) return;                       // no window of this launch's kind in the chunk: before any staging

  // plane order of the key blocks: real planes first.  level_of(z) = source pressure level of plane z, or -1
  auto level_of = [&](int z) -> int {
    const int sp = (ipl * WPL + z + G.sf[0]) % G.ppl - G.pad_f;
    return sp >= 0 && sp < G.pl ? sp : -1;
  };
  int zplane[WPL];                                // plane of key-block group i
  if constexpr (PP == 0) {
#pragma unroll
    for (int i = 0; i < WPL; ++i) zplane[i] = i;
  } else {
    zplane[0] = level_of(0) >= 0 ? 0 : 1;         // (host: exactly one of the two planes is real)
    zplane[1] = 1 - zplane[0];
  }

  // ---- once per workgroup: bias column, key part of the bias index, the padding plane's shared K / V blocks
  {
    const int type = ipl * G.nlat + ilat;
    const float4* src = reinterpret_cast<const float4*>(tabT + ((long long)type * G.heads + head) * G.TRP);
    for (int i = tid; i < G.TRP / 4; i += NT) reinterpret_cast<float4*>(s_tab)[i] = src[i];
    for (int s = tid; s < NS; s += NT) {
      const int z = zplane[s / (16 * PB)], ps = s % (16 * PB), pp = ps < NPQ ? ps : 0;
      const int h = pp / G.wlon, w = pp % G.wlon;
      s_bkey[s] = (z * G.wpl * G.wlat * G.wlat + h * G.wlat) * G.W2 - w;
    }
    // V: zero everything once (pad columns 33..47, rows of duplicates, the partial block's tail)
    for (int i = tid; i < NP * VR * LDV / 2; i += NT) reinterpret_cast<unsigned*>(s_v)[i] = 0u;
  }
  __syncthreads();
  if constexpr (PP == 1) {
    // rows of the shared blocks: K 16 rows of bias_k; V 16 rows of bias_v ("full") + NPQ % 16 rows ("tail")
    const int tail = NPQ % 16;
    for (int i = tid; i < 48 * 8; i += NT) {
      const int row = i >> 3, c8 = i & 7;                     // rows 0..15 K, 16..31 V full, 32..47 V tail
      const bool isk = row < 16;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      uint2 v16 = {0u, 0u};
      if constexpr (IO16) v16 = *reinterpret_cast<const uint2*>(bias16 + (isk ? C : 2 * C) + head * D + 4 * c8);
      else v = *reinterpret_cast<const float4*>(qkv_bias + (isk ? C : 2 * C) + head * D + 4 * c8);
      if (!isk && row >= 32 && row - 32 >= tail) continue;
      unsigned short* d = isk ? s_k + (RP * 16 * PB + row) * LDK + 4 * c8 : s_v + (RP * 16 * PB + row - 16) * LDV + 4 * c8;
      const int prow = isk ? KR * LDK : VR * LDV;
      if constexpr (IO16) {
        *reinterpret_cast<uint2*>(d) = v16;
      } else if constexpr (NP == 3) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(d + prow) = uint2{m0, m1};
        *reinterpret_cast<uint2*>(d + 2 * prow) = uint2{l0, l1};
      } else {
        *reinterpret_cast<uint2*>(d) = uint2{cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
      }
      if (!isk && c8 == 0) s_v[(RP * 16 * PB + row - 16) * LDV + D] = (unsigned short)0x3F80;     // ones column
    }
  }

  // ---- per thread, fixed for the workgroup: the two in-plane slots it stages per real plane (row offsets without the
  // longitude term, or -1), the lane's query row, its destination row
  const int ch = tid & 7;                         // 16-byte chunk of the 32-float row
  int st_w[2], st_base[RP][2];
  bool st_dummy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ps = (tid >> 3) + 8 * PB * i;
    st_dummy[i] = ps >= NPQ;
    const int pp = st_dummy[i] ? 0 : ps;
    st_w[i] = pp % G.wlon;
    const int sa = (ilat * G.wlat + pp / G.wlon + G.sf[1]) % G.plat - G.pad_t;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const int sp = level_of(zplane[r]);
      st_base[r][i] = (sp >= 0 && sa >= 0 && sa < G.lat) ? (sp * G.lat + sa) * G.lon : -1;
    }
  }
  const int qrow = 16 * wave + j;
  const bool q_dummy = qrow >= NPQ;
  const int q_h = (q_dummy ? 0 : qrow) / G.wlon, q_w = (q_dummy ? 0 : qrow) % G.wlon;
  const int a_q = (zq * G.wlat * G.wlat + q_h) * G.W2 + q_w + G.wlon - 1;
  int q_base, d_base;
  {
    const int sp = level_of(zq);
    const int sa = (ilat * G.wlat + q_h + G.sf[1]) % G.plat - G.pad_t;
    q_base = (sp >= 0 && sa >= 0 && sa < G.lat) ? (sp * G.lat + sa) * G.lon : -1;
    const int da = (ilat * G.wlat + q_h + G.sb[1]) % G.plat - G.pad_t;
    d_base = (!q_dummy && da >= 0 && da < G.lat) ? (dlev * G.lat + da) * G.lon : -1;
  }
  // shift-mask regions: level and latitude parts are fixed, the longitude part moves with the window
  int rq_base = 0, rk_base = 0, rk_w = 0;
  if constexpr (MASK) {
    auto rid = [&](int p, int dim) { return (p >= G.b1[dim]) + (p >= G.b2[dim]); };
    rq_base = (rid(ipl * WPL + zq, 0) * 3 + rid(ilat * G.wlat + q_h, 1)) * 3;
    if (tid < NS) {
      const int z = zplane[tid / (16 * PB)], ps = tid % (16 * PB), pp = ps < NPQ ? ps : 0;
      rk_base = (rid(ipl * WPL + z, 0) * 3 + rid(ilat * G.wlat + pp / G.wlon, 1)) * 3;
      rk_w = pp % G.wlon;
    }
  }
  static_assert(NS <= NT || !MASK, "one region slot per thread");
  float plane_mask[WPL];                          // 0 / -100 log2 e of key plane group i against the query plane's level region
#pragma unroll
  for (int i = 0; i < WPL; ++i) {
    const int pk = ipl * WPL + zplane[i], pq = ipl * WPL + zq;
    const bool differ = ((pk >= G.b1[0]) + (pk >= G.b2[0])) != ((pq >= G.b1[0]) + (pq >= G.b2[0]));
    plane_mask[i] = (MASK && differ) ? -100.0f * LOG2E : 0.f;
  }

  const float* qkv_b = qkv + (long long)b * L * 3 * C;
  float* out_b = out + (long long)b * L * C;
  // longitude position of a window column in the source (forward roll) / destination (backward roll) frame
  auto src_lon = [&](int ilon, int w) {            // (both terms < plon: one conditional subtract, no division)
    const int x = ilon * G.wlon + w + G.sf[2];
    return (x >= G.plon ? x - G.plon : x) - G.pad_l;
  };

  KV kreg[CPT], vreg[CPT];
  float4 qreg[2];
  u32x4 qraw = {0u, 0u, 0u, 0u};
  const unsigned short* qkv_b16 = qkv16 + (long long)b * L * 3 * C;
  auto prefetch = [&](int ilon) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int so = src_lon(ilon, st_w[i]);
      const bool lon_ok = so >= 0 && so < G.lon;
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const bool real = lon_ok && st_base[r][i] >= 0;
        if constexpr (IO16) {
          const unsigned short* row = real ? qkv_b16 + (long long)(st_base[r][i] + so) * 3 * C : bias16;
          kreg[2 * r + i] = *reinterpret_cast<const uint2*>(row + C + head * D + 4 * ch);
          vreg[2 * r + i] = *reinterpret_cast<const uint2*>(row + 2 * C + head * D + 4 * ch);
        } else {
          const float* row = real ? qkv_b + (long long)(st_base[r][i] + so) * 3 * C : qkv_bias;
          kreg[2 * r + i] = *reinterpret_cast<const float4*>(row + C + head * D + 4 * ch);
          vreg[2 * r + i] = *reinterpret_cast<const float4*>(row + 2 * C + head * D + 4 * ch);
        }
      }
    }
    const int so = src_lon(ilon, q_w);
    const bool qreal = so >= 0 && so < G.lon && q_base >= 0;
    if constexpr (IO16) {
      const unsigned short* row = qreal ? qkv_b16 + (long long)(q_base + so) * 3 * C : bias16;
      qraw = *reinterpret_cast<const u32x4*>(row + head * D + 8 * g);
    } else {
      const float* row = qreal ? qkv_b + (long long)(q_base + so) * 3 * C : qkv_bias;
      qreg[0] = *reinterpret_cast<const float4*>(row + head * D + 8 * g);
      qreg[1] = *reinterpret_cast<const float4*>(row + head * D + 8 * g + 4);
    }
  };
  // K row (32 bf16 per part) and V row (+ ones column) of one staged chunk
  auto put = [&](unsigned short* base, int ld, int part_rows, int slot, const KV& v, bool zero) {
    unsigned short* d = base + slot * ld + 4 * ch;
    if constexpr (IO16) {
      *reinterpret_cast<uint2*>(d) = zero ? uint2{0u, 0u} : v;
    } else if constexpr (NP == 3) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3_pair(v.x, v.y, h0, m0, l0);
      split3_pair(v.z, v.w, h1, m1, l1);
      if (zero) h0 = m0 = l0 = h1 = m1 = l1 = 0u;
      *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(d + part_rows * ld) = uint2{m0, m1};
      *reinterpret_cast<uint2*>(d + 2 * part_rows * ld) = uint2{l0, l1};
    } else {
      uint2 r = uint2{cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
      if (zero) r = uint2{0u, 0u};
      *reinterpret_cast<uint2*>(d) = r;
    }
  };
  // LDS rows of key block kb
  auto krow = [&](int kb) { return kb < RP * PB ? 16 * kb : RP * 16 * PB; };
  auto vrow = [&](int kb) {
    return kb < RP * PB ? 16 * kb : RP * 16 * PB + ((kb == KB - 1 && NPQ % 16) ? 16 : 0);
  };

  if (ilon < lon1) prefetch(ilon);
  while (ilon < lon1) {
    const int inext = next_win(ilon + 1);
    // ---- stage the window: K / V rows of the real planes (the prefetched registers), regions, Q fragment
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int slot = (i >> 1) * 16 * PB + (tid >> 3) + 8 * PB * (i & 1);
      put(s_k, LDK, KR, slot, kreg[i], false);
      put(s_v, LDV, VR, slot, vreg[i], st_dummy[i & 1]);
      if (ch == 0) s_v[slot * LDV + D] = st_dummy[i & 1] ? (unsigned short)0 : (unsigned short)0x3F80;   // ones column, part 0
    }
    int rq = 0;
    if constexpr (MODE == 2) {
      auto rlon = [&](int w) { const int O = ilon * G.wlon + w; return (O >= G.b1[2]) + (O >= G.b2[2]); };
      if (tid < NS) s_rkey[tid] = rk_base + rlon(rk_w);
      rq = rq_base + rlon(q_w);
    }
    // Q fragment: B operand, lane (query j, k-slots 8g .. 8g+7), scaled into the log2 domain
    u32x4 qb[NP];
    {
      if constexpr (IO16) {   // bf16 q: unpack (the scale is applied in fp32, the product rounded to bf16 like the fp32 path's)
        qreg[0] = float4{__uint_as_float(qraw[0] << 16), __uint_as_float(qraw[0] & 0xFFFF0000u), __uint_as_float(qraw[1] << 16),
                         __uint_as_float(qraw[1] & 0xFFFF0000u)};
        qreg[1] = float4{__uint_as_float(qraw[2] << 16), __uint_as_float(qraw[2] & 0xFFFF0000u), __uint_as_float(qraw[3] << 16),
                         __uint_as_float(qraw[3] & 0xFFFF0000u)};
      }
      const float f[8] = {qreg[0].x * G.qscale, qreg[0].y * G.qscale, qreg[0].z * G.qscale, qreg[0].w * G.qscale,
                          qreg[1].x * G.qscale, qreg[1].y * G.qscale, qreg[1].z * G.qscale, qreg[1].w * G.qscale};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (NP == 3) {
          unsigned hh, mm, ll;
          split3_pair(f[2 * i], f[2 * i + 1], hh, mm, ll);
          qb[0][i] = hh; qb[1][i] = mm; qb[2][i] = ll;
        } else {
          qb[0][i] = cvt_pk_bf16(f[2 * i], f[2 * i + 1]);
        }
      }
    }
    // destination token of the lane's query (backward roll), -1: cropped away
    long long dest = -1;
    {
      const int dx = ilon * G.wlon + q_w + G.sb[2];
      const int dq = (dx >= G.plon ? dx - G.plon : dx) - G.pad_l;
      if (d_base >= 0 && dq >= 0 && dq < G.lon) dest = d_base + dq;
    }
    __syncthreads();                                   // the window's K / V / regions are in LDS
    if (inext < lon1) prefetch(inext);                 // next window's operands, in flight during the compute

    // ---- scores: S^T tile kb = K block (A) x Q (B), accumulator pre-loaded with the bias (+ mask).  All index reads first,
    // then all gathers, then the matrix instructions: two LDS latencies per window, not two per key block
    f32x4 sc[KB];
    constexpr int HB = (KB + 1) / 2;                   // in two halves: bounds the index registers in flight
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int k0 = half * HB, k1 = half ? KB : HB;
      int4 bk[HB];
#pragma unroll
      for (int kb = k0; kb < k1; ++kb) bk[kb - k0] = *reinterpret_cast<const int4*>(s_bkey + 16 * kb + 4 * g);
#pragma unroll
      for (int kb = k0; kb < k1; ++kb)
        sc[kb] = f32x4{s_tab[a_q + bk[kb - k0].x], s_tab[a_q + bk[kb - k0].y], s_tab[a_q + bk[kb - k0].z],
                       s_tab[a_q + bk[kb - k0].w]};
      asm volatile("" ::: "memory");                   // keep the halves apart
    }
    if constexpr (MODE == 2) {
      const float mv = -100.0f * LOG2E;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int4 rk = *reinterpret_cast<const int4*>(s_rkey + 16 * kb + 4 * g);
        sc[kb][0] += rk.x != rq ? mv : 0.f;
        sc[kb][1] += rk.y != rq ? mv : 0.f;
        sc[kb][2] += rk.z != rq ? mv : 0.f;
        sc[kb][3] += rk.w != rq ? mv : 0.f;
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const float pm = plane_mask[kb / PB];
        sc[kb] += f32x4{pm, pm, pm, pm};
      }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      u32x4 ka[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) ka[p] = *reinterpret_cast<const u32x4*>(s_k + (p * KR + krow(kb) + j) * LDK + 8 * g);
      if constexpr (NP == 3) {
#pragma unroll
        for (int term = 0; term < 6; ++term) sc[kb] = mfma16x16x32_bf16(ka[kTA[term]], qb[kTB[term]], sc[kb]);
      } else {
        sc[kb] = mfma16x16x32_bf16(ka[0], qb[0], sc[kb]);
      }
    }
    // ---- exact row maximum (the keys of query j: 4 per tile in this lane, x 4 lane groups)
    float mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
    for (int kb = 1; kb < KB; ++kb) mx = fmaxf(fmaxf(mx, sc[kb][0]), fmaxf(fmaxf(sc[kb][1], sc[kb][2]), sc[kb][3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // ---- probabilities -> P^T operand, out^T += V^T P^T (three 16-dim blocks: 32 dims + the ones column = row sums)
    f32x4 oacc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      constexpr int kb0 = 0;
      const int kbA = 2 * ks, kbB = 2 * ks + 1 < KB ? 2 * ks + 1 : kb0;
      float pe[2][4];
      if constexpr (NP == 3) {                     // (the packed form below costs the NP = 3 kernel registers it does not have)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pe[0][r] = __builtin_amdgcn_exp2f(sc[kbA][r] - mx);
          pe[1][r] = 2 * ks + 1 < KB ? __builtin_amdgcn_exp2f(sc[kbB][r] - mx) : 0.f;
        }
      } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 m2 = {mx, mx};                 // v_pk_add_f32: two subtractions per instruction
        const f32x2 a0 = f32x2{sc[kbA][0], sc[kbA][1]} - m2, a1 = f32x2{sc[kbA][2], sc[kbA][3]} - m2;
        const f32x2 b0 = f32x2{sc[kbB][0], sc[kbB][1]} - m2, b1 = f32x2{sc[kbB][2], sc[kbB][3]} - m2;
        pe[0][0] = __builtin_amdgcn_exp2f(a0.x); pe[0][1] = __builtin_amdgcn_exp2f(a0.y);
        pe[0][2] = __builtin_amdgcn_exp2f(a1.x); pe[0][3] = __builtin_amdgcn_exp2f(a1.y);
        const bool two = 2 * ks + 1 < KB;
        pe[1][0] = two ? __builtin_amdgcn_exp2f(b0.x) : 0.f; pe[1][1] = two ? __builtin_amdgcn_exp2f(b0.y) : 0.f;
        pe[1][2] = two ? __builtin_amdgcn_exp2f(b1.x) : 0.f; pe[1][3] = two ? __builtin_amdgcn_exp2f(b1.y) : 0.f;
      }
      // k-slot e of lane group g is key 4g + e (e < 4) of block 2ks or key 4g + e - 4 of block 2ks + 1
      u32x4 pb[NP];
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
      const int vA = vrow(kbA), vB = vrow(kbB);
#pragma unroll
      for (int db = 0; db < 3; ++db) {
        u32x4 va[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (db == 2 && p > 0) { va[p] = u32x4{0u, 0u, 0u, 0u}; continue; }     // the ones column lives in part 0 only
          // block of 4 keys x 16 dims: lane i of the 16-lane group addresses row i >> 2, columns 4 (i & 3) .. + 3 and gets
          // column i of the 4 rows (ds_read_b64_tr_b16): the A operand of out^T += V^T P^T without a transposed copy
          const int off = (4 * g + (j >> 2)) * LDV + 16 * db + 4 * (j & 3);
          const unsigned short* v0 = s_v + (p * VR + vA) * LDV + off;
          const unsigned short* v1 = s_v + (p * VR + vB) * LDV + off;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(v1));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          va[p] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
        if constexpr (NP == 3) {
          if (db == 2) {      // ones column: V parts m, l are zero -> the three products with V_h
            oacc[db] = mfma16x16x32_bf16(va[0], pb[2], oacc[db]);
            oacc[db] = mfma16x16x32_bf16(va[0], pb[1], oacc[db]);
            oacc[db] = mfma16x16x32_bf16(va[0], pb[0], oacc[db]);
          } else {
#pragma unroll
            for (int term = 0; term < 6; ++term) oacc[db] = mfma16x16x32_bf16(va[kTA[term]], pb[kTB[term]], oacc[db]);
          }
        } else {
          oacc[db] = mfma16x16x32_bf16(va[0], pb[0], oacc[db]);
        }
      }
    }
    // ---- normalise and store: lane (query j, group g) holds out[query][16 db + 4 g .. + 3]; the row sum is element 0 of
    // block 2 in the lanes of group 0
    const float lsum = __shfl(oacc[2][0], j);
    const float inv = 1.0f / lsum;
    if (dest >= 0) {
      if constexpr (IO16) {
        unsigned short* o = reinterpret_cast<unsigned short*>(out) + ((long long)b * L + dest) * C + head * D + 4 * g;
        const f32x4 u0 = oacc[0] * inv, u1 = oacc[1] * inv;
        *reinterpret_cast<uint2*>(o) = uint2{cvt_pk_bf16(u0[0], u0[1]), cvt_pk_bf16(u0[2], u0[3])};
        *reinterpret_cast<uint2*>(o + 16) = uint2{cvt_pk_bf16(u1[0], u1[1]), cvt_pk_bf16(u1[2], u1[3])};
      } else {
        float* o = out_b + dest * C + head * D + 4 * g;
        *reinterpret_cast<f32x4*>(o) = oacc[0] * inv;
        *reinterpret_cast<f32x4*>(o + 16) = oacc[1] * inv;
      }
    }
    __syncthreads();                                   // every wave is done with this window's LDS images
    ilon = inext;
  }
}

}  // namespace wattn3

using namespace wattn3;

struct Plan3 {
  Geo G;
  int rp, pp, pb;
  bool ok;
};

static Plan3 make_plan3(const dlwp_wattn_desc* u) {
  Plan3 P;
  P.ok = false;
  if (!u || u->bias_mode != 1 || u->head_dim != 32 || u->heads <= 0) return P;
  const int wpl = u->window[0], wlat = u->window[1], wlon = u->window[2];
  if (wpl < 1 || wpl > 2 || wlat < 1 || wlon < 1 || wlat * wlon > 80) return P;
  for (int i = 0; i < 3; ++i)
    if (u->grid[i] <= 0 || u->padded[i] <= 0 || u->padded[i] % u->window[i] || u->padded[i] < u->grid[i] + u->pad_lead[i] ||
        u->pad_lead[i] < 0)
      return P;
  Geo& G = P.G;
  G.pl = u->grid[0]; G.lat = u->grid[1]; G.lon = u->grid[2];
  G.ppl = u->padded[0]; G.plat = u->padded[1]; G.plon = u->padded[2];
  G.pad_f = u->pad_lead[0]; G.pad_t = u->pad_lead[1]; G.pad_l = u->pad_lead[2];
  G.wpl = wpl; G.wlat = wlat; G.wlon = wlon;
  G.npl = G.ppl / wpl; G.nlat = G.plat / wlat; G.nlon = G.plon / wlon;
  for (int i = 0; i < 3; ++i) {
    const int dim = u->padded[i];
    G.sf[i] = ((u->shift_fwd[i] % dim) + dim) % dim;
    G.sb[i] = ((u->shift_back[i] % dim) + dim) % dim;
    G.b1[i] = u->use_mask ? u->mask_b1[i] : (1 << 30);
    G.b2[i] = u->use_mask ? u->mask_b2[i] : (1 << 30);
  }
  G.heads = u->heads; G.C = u->heads * 32;
  G.W2 = 2 * wlon - 1;
  G.TR = wpl * wpl * wlat * wlat * G.W2;
  G.TRP = (G.TR + 3) & ~3;
  G.qscale = u->scale * 1.4426950408889634f;
  // planes: every plane of every window real (no level padding), or ONE window level of two planes of which one is real
  int real_min = wpl, real_max = 0;
  for (int ipl = 0; ipl < G.npl; ++ipl) {
    int real = 0;
    for (int z = 0; z < wpl; ++z) {
      const int sp = (ipl * wpl + z + G.sf[0]) % G.ppl - G.pad_f;
      real += sp >= 0 && sp < G.pl;
    }
    real_min = real < real_min ? real : real_min;
    real_max = real > real_max ? real : real_max;
  }
  if (real_min == wpl) { P.rp = wpl; P.pp = 0; }
  else if (wpl == 2 && G.npl == 1 && real_min == 1 && real_max == 1) { P.rp = 1; P.pp = 1; }
  else return P;
  const int npq = wlat * wlon;
  P.pb = npq <= 32 ? 2 : (npq <= 48 ? 3 : 5);
  // cut windows (shifted blocks): rows / columns with a region boundary strictly inside
  G.ncr = G.ncc = G.ncut = 0;
  G.cr[0] = G.cr[1] = G.cc[0] = G.cc[1] = 0;
  if (u->use_mask) {
    auto inside = [](int bb, int lo, int n) { return bb > lo && bb < lo + n; };
    for (int r = 0; r < G.nlat; ++r)
      if (inside(G.b1[1], r * wlat, wlat) || inside(G.b2[1], r * wlat, wlat)) {
        if (G.ncr == 2) return P;
        G.cr[G.ncr++] = r;
      }
    for (int c = 0; c < G.nlon; ++c)
      if (inside(G.b1[2], c * wlon, wlon) || inside(G.b2[2], c * wlon, wlon)) {
        if (G.ncc == 2) return P;
        G.cc[G.ncc++] = c;
      }
    G.ncut = G.ncr * G.nlon + G.ncc * (G.nlat - G.ncr);
  }
  // longitude windows per workgroup: ~6, so that a call is several thousand workgroups (the bias column is re-staged per
  // workgroup: 13 KB against ~20 KB of K / V per window)
  G.nch = (G.nlon + 5) / 6;
  G.chunk = (G.nlon + G.nch - 1) / G.nch;
  P.ok = true;
  return P;
}

template <int RP, int PP, int PB, int NP>
static size_t lds_bytes(const Geo& G) {
  constexpr int KB = (RP + PP) * PB, NS = 16 * KB, KR = RP * 16 * PB + PP * 16, VR = RP * 16 * PB + PP * 32;
  return (size_t)G.TRP * 4 + (size_t)NP * KR * LDK * 2 + (size_t)NP * VR * LDV * 2 + (size_t)2 * NS * 4;
}

template <int RP, int PP, int PB, int NP, bool IO16 = false>
static int32_t launch3(const Geo& G, bool mask, const float* qkv, const float* qkv_bias, const float* tabT, float* out, int batch,
                       long long L, hipStream_t s) {
  const size_t lds = lds_bytes<RP, PP, PB, NP>(G);
  if (lds > 160 * 1024) return 1;
  const long long gy = (long long)batch * G.npl * G.nlat * (RP + PP) * G.nch;
  const long long gy2 = (long long)batch * G.npl * (RP + PP) * G.ncut;
  if (gy >= 65536 || gy2 >= 65536) return 1;      // beyond the grid's y extent: the generic kernel takes the call
  const dim3 grid((unsigned)G.heads, (unsigned)gy), block(64 * PB);
  auto go = [&](auto kern, dim3 gr) -> int32_t {
    if (lds > 48 * 1024)
      DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, gr, block, lds, s, G, qkv, qkv_bias, tabT, out, L);
    return DLWP_OK;
  };
  // (unshifted blocks run the same kernel: boundaries at 2^30, nothing cut, plane constants 0.  A mask-free instantiation
  // was no faster at NP = 1 and needed 9 spilled registers at NP = 3.)
  int32_t rc = go(wattn3_kernel<RP, PP, PB, NP, 1, IO16>, grid);
  if (rc != DLWP_OK) return rc;
  if (mask && G.ncut > 0) {
    rc = go(wattn3_kernel<RP, PP, PB, NP, 2, IO16>, dim3((unsigned)G.heads, (unsigned)gy2));
    if (rc != DLWP_OK) return rc;
  }
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

size_t wattn3_workspace_bytes(const dlwp_wattn_desc* u, int /*batch*/, int /*np*/) {
  const Plan3 P = make_plan3(u);
  if (!P.ok) return 0;
  return (size_t)P.G.npl * P.G.nlat * P.G.heads * P.G.TRP * sizeof(float);
}

// 0 = done, 1 = not covered by this kernel (the caller falls through to the generic one), < 0 = error
// np: 3 bf16x6 (fp32-accurate), 1 bf16, 16 bf16 with bfloat16 qkv / qkv bias / output tensors (the pointers are reinterpreted)
int32_t wattn3_run(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias, const float* table, float* out, int batch,
                   void* workspace, size_t workspace_bytes, hipStream_t s, int np) {
  const Plan3 P = make_plan3(u);
  if (!P.ok) return 1;
  const Geo& G = P.G;
  const size_t need = wattn3_workspace_bytes(u, batch, np);
  if (!workspace || workspace_bytes < need) return 1;
  const bool padded = G.ppl != G.pl || G.plat != G.lat || G.plon != G.lon;
  if (padded && !qkv_bias) return 1;
  const long long L = (long long)G.pl * G.lat * G.lon;
  if (L * 3 * G.C >= (1ll << 31)) return 1;
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(qkv_bias) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return 1;
  float* tabT = reinterpret_cast<float*>(workspace);
  const int types = G.npl * G.nlat;
  hipLaunchKernelGGL(wattn3_table_kernel, dim3((unsigned)(types * G.heads)), dim3(256), 0, s, table, tabT, G.TR, G.TRP, types,
                     G.heads);
  DLWP_HIP_CHECK(hipGetLastError());
  const float* qb = qkv_bias ? qkv_bias : qkv;      // never dereferenced when nothing is padded
  const bool mask = u->use_mask != 0;
#define DLWP_W3(RP_, PP_, PB_)                                                                                  \
  if (P.rp == RP_ && P.pp == PP_ && P.pb == PB_)                                                                \
    return np == 3 ? launch3<RP_, PP_, PB_, 3>(G, mask, qkv, qb, tabT, out, batch, L, s)                        \
                   : (np == 16 ? launch3<RP_, PP_, PB_, 1, true>(G, mask, qkv, qb, tabT, out, batch, L, s)      \
                               : launch3<RP_, PP_, PB_, 1>(G, mask, qkv, qb, tabT, out, batch, L, s));
  DLWP_W3(1, 1, 5) DLWP_W3(1, 1, 3) DLWP_W3(1, 1, 2) DLWP_W3(2, 0, 5) DLWP_W3(2, 0, 3) DLWP_W3(2, 0, 2) DLWP_W3(1, 0, 5)
  DLWP_W3(1, 0, 3) DLWP_W3(1, 0, 2)
#undef DLWP_W3
  return 1;
}

}  // namespace dlwp
