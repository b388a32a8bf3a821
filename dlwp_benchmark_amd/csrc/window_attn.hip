// Fused (shifted-)window attention for the Swin and Pangu backbones on MI355X (gfx950).
//
// Replaces, per transformer block, everything between the qkv Linear and the proj Linear:
//   Swin : swin_transformer.py:217-251 (pad, roll, window_partition, WindowAttention.forward :122-154
//          minus its two Linears, window_reverse, roll back, crop) and the per-call mask build :383-401
//   Pangu: panguweather.py:285-316 (ZeroPad3d, roll, window_partition, EarthAttention3D.forward
//          :176-211 minus its two Linears, window_reverse, roll, crop3d), utils/shift_window_mask.py,
//          utils/earth_position_index.py
// Input is the qkv Linear output on the UN-padded, UN-shifted token sequence [B, L, 3, nH, d]; output
// is [B, L, C] in the same token order.  Padding, cyclic shift (forward and backward shifts may
// differ: Pangu's asymmetric roll, panguweather.py:291 vs :310), window partition / reverse, the
// relative-position (Swin) or earth-specific (Pangu) bias gather and the 0/-100 shift mask are all
// index arithmetic inside the kernel: none of those tensors is ever materialised.  Zero-padded
// tokens take part in the softmax with q = k = v = qkv bias, exactly as in the reference (the pad is
// applied before the Linear).
//
// Algorithm: flash-style streaming over 32-key tiles with online softmax, so the N x N score matrix
// (N = 2048 for the reference Swin, whose window is the whole map) never exists.  Products run on
// v_mfma_f32_16x16x4_f32 in the "swapped" orientation: S^T = K Q^T puts the key index on the
// accumulator ROWS, which is the contraction index of the following O^T += V^T P^T -- the P tile is
// consumed as an MFMA operand straight from the accumulator registers (no LDS round trip), and the
// softmax statistics of a query live in one lane column.
#include "common.hpp"
#include "window_attn_desc.hpp"

namespace dlwp {
namespace wattn {

// DT = head_dim / 4, DB = ceil(head_dim / 16) (16-row blocks of O^T), SUB = 16-query sub-tiles per wave.
// Block = 4 waves = 64*SUB queries of one (batch, window, head); keys stream through LDS in tiles of 32.
//
// PREC = 0: fp32 operands, v_mfma_f32_16x16x4_f32 (exact fp32 products; dlwp_window_attn_f32 for windows >= 512 tokens).
// PREC = 1: Q, K, V and P rounded to bf16, v_mfma_f32_16x16x32_bf16 with fp32 accumulation and fp32
//           softmax statistics; head_dim is one (<= 32) or two k-steps deep, so a 32-key x 16-query
//           tile costs 2-4 + DB matrix instructions instead of 2*DT + 8*DB.
// PREC = 2: "bf16x6" (dlwp_window_attn_f32 for smaller windows): Q, K, V and P split EXACTLY into three
//           bf16 parts each (common.hpp), six cross products per contraction accumulated in fp32 -- fp32-GEMM
//           accuracy on the bf16 matrix pipe: 6 x 16 cycles per 32-deep k-step instead of 8 x 32 for the same
//           contraction on fp32 MFMA.  Measured: the same time per call as PREC 0 (872 / 910 us vs 859 / 927 us at the
//           Swin C3 shape) -- 45 % more VALU instructions for the splits eat what the matrix pipe saves.
// LON4: the window's fastest axis is a multiple of 4, so the 4 keys a lane owns in a 16-key block
//       (rows 4g..4g+3 of the accumulator) are consecutive along longitude: ONE bias index per 4 scores.
// All score arithmetic is in the log2 domain (q scale, bias table and mask are pre-multiplied by
// log2 e) so the softmax exponentials are bare v_exp_f32.
__device__ constexpr int kPA[6] = {2, 0, 1, 1, 0, 0};   // bf16x6 terms, smallest first: (A part, B part) =
__device__ constexpr int kPB[6] = {0, 2, 1, 0, 1, 0};   // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)

template <int DT, int DB, int SUB, int PREC, bool LON4, bool MASK>
__global__ __launch_bounds__(256) void window_attn_kernel(const Desc D, const float* __restrict__ qkv,
                                                          const float* __restrict__ qkv_bias,
                                                          const float* __restrict__ table,
                                                          float* __restrict__ out, long long L) {
  extern __shared__ __align__(16) float smem[];
  constexpr bool BF16 = PREC != 0;
  constexpr int NP = PREC == 2 ? 3 : 1;        // bf16 parts per operand
  constexpr int KT = 32;                       // keys per tile
  constexpr int LDK = 4 * DT + 2;              // fp32 K tile row stride (floats)
  constexpr int LDV = 16 * DB + 4;             // fp32 V tile row stride
  constexpr int DK = (4 * DT + 31) / 32;       // bf16: k-steps of 32 over head_dim
  constexpr int LDKB = 32 * DK + 8;            // bf16 K tile row stride (bf16 elements), 16-byte aligned rows
  constexpr int LDVB = KT + 8;                 // bf16 V^T tile row stride (bf16 elements)
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int head = blockIdx.y;
  int wi = blockIdx.z;
  const int nwin = D.npl * D.nlat * D.nlon;
  const int b = wi / nwin;
  wi -= b * nwin;
  const int ilat = wi % D.nlat;
  const int t2 = wi / D.nlat;
  const int ipl = t2 % D.npl, ilon = t2 / D.npl;
  const int d = D.d, C = D.C, N = D.N;

  float* s_tab = smem;                                   // [table_rows] bias column * log2 e
  float* s_kv = s_tab + ((D.table_rows + 3) & ~3);       // TWO K / V tile buffers (layout depends on BF16)
  float* s_k;                                            // fp32: [KT][LDK]
  float* s_v;                                            // fp32: [KT][LDV]
  unsigned short* s_kb;                                  // bf16: [NP][KT][LDKB]
  unsigned short* s_vb;                                  // bf16: [NP][16*DB][LDVB]  (V transposed)
  int* s_info;                                           // [KT] packed key coords, -1 beyond N
  constexpr int KV_FLOATS = BF16 ? (NP * (KT * LDKB + 16 * DB * LDVB) + 1) / 2 : KT * (LDK + LDV);
  constexpr int EPI_FLOATS = 64 * SUB * (16 * DB + 1);
  constexpr int TILE_FLOATS = ((KV_FLOATS + 3) & ~3) + KT;
  auto set_buf = [&](int bsel) {   // point the tile views at buffer bsel
    float* base = s_kv + bsel * TILE_FLOATS;
    s_k = base;
    s_v = s_k + KT * LDK;
    s_kb = reinterpret_cast<unsigned short*>(base);
    s_vb = s_kb + NP * KT * LDKB;
    s_info = reinterpret_cast<int*>(base + ((KV_FLOATS + 3) & ~3));
  };
  set_buf(0);
  // window token map, computed once per workgroup: source token (or -1 = zero-padded) and packed
  // coordinates / region of every in-window position (token_coord is ~10 integer divisions)
  int* s_msrc = reinterpret_cast<int*>(s_kv + (((2 * TILE_FLOATS > EPI_FLOATS ? 2 * TILE_FLOATS : EPI_FLOATS) + 3) & ~3));  // [N]
  int* s_minfo = s_msrc + ((N + 3) & ~3);                                                                        // [N]

  {
    const int type = ipl * D.nlat + ilat;
    const long long stride = D.bias_mode ? (long long)D.types * D.heads : D.heads;
    const float* col = table + (D.bias_mode ? (long long)type * D.heads : 0) + head;
    for (int i = tid; i < D.table_rows; i += 256) s_tab[i] = col[i * stride] * LOG2E;
  }

  for (int n = tid; n < N; n += 256) {
    const Coord c = token_coord(D, ipl, ilat, ilon, n);
    s_msrc[n] = c.src >= 0 ? (int)c.src * 3 * C : -1;   // ELEMENT offset of the token's qkv row (fits 31 bits, checked on the host)
    s_minfo[n] = pack_info(c);
  }
  __syncthreads();

  const float* qkv_b = qkv + (long long)b * L * 3 * C;
  const int q_base = blockIdx.x * (64 * SUB) + wave * (16 * SUB);
  const float qscale = D.scale * LOG2E;
  float qreg[BF16 ? 1 : SUB][BF16 ? 1 : DT];      // fp32 path: B[k = g][col j] per k-step
  u32x4 qb[BF16 ? SUB : 1][BF16 ? DK : 1][NP];     // bf16 paths: 8 consecutive head dims per k-step (x parts)
  int qinfo[SUB];
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub) {
    const int qn = q_base + 16 * sub + j;
    const bool live = qn < N;
    const int qsrc = live ? s_msrc[qn] : -1;
    qinfo[sub] = live ? s_minfo[qn] : 0;
    const float* src = (live && qsrc >= 0) ? qkv_b + qsrc + head * d : qkv_bias + head * d;
    if constexpr (!BF16) {
#pragma unroll
      for (int s = 0; s < DT; ++s) qreg[sub][s] = live ? src[4 * s + g] * qscale : 0.f;
    } else {
#pragma unroll
      for (int ks = 0; ks < DK; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 32 * ks + 8 * g + 2 * i;
          const float v0 = (live && e < d) ? src[e] * qscale : 0.f;
          const float v1 = (live && e + 1 < d) ? src[e + 1] * qscale : 0.f;
          if constexpr (PREC == 2) {
            unsigned hh, mm, ll;
            split3_pair(v0, v1, hh, mm, ll);
            qb[sub][ks][0][i] = hh;
            qb[sub][ks][1][i] = mm;
            qb[sub][ks][2][i] = ll;
          } else {
            qb[sub][ks][0][i] = cvt_pk_bf16(v0, v1);
          }
        }
    }
  }

  f32x4 oacc[SUB][DB];
  float m_run[SUB], l_run[SUB];
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub) {
    m_run[sub] = -1e30f;
    l_run[sub] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) oacc[sub][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float mask_val = -100.0f * LOG2E;

  const int ntile = (N + KT - 1) / KT;
  // ---- K/V staging split in two (global -> registers early, registers -> LDS late) so the global
  // latency of tile kt+1 hides behind the MFMA / softmax work of tile kt
  constexpr int EP = 16 * DK;                                         // bf16: element pairs per key row
  constexpr int NI = BF16 ? (KT * EP + 255) / 256 : (KT * 4 * DT + 255) / 256;   // staging items per thread
  float pk0[NI], pk1[BF16 ? NI : 1], pv0[NI], pv1[BF16 ? NI : 1];
  int pinfo[NI];
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      if constexpr (!BF16) {
        const int key = i / (4 * DT), e = i % (4 * DT);
        const int kn = kt * KT + key;
        pk0[it] = 0.f; pv0[it] = 0.f; pinfo[it] = -1;
        if (i < KT * 4 * DT && kn < N) {
          const int ksrc = s_msrc[kn];
          pinfo[it] = s_minfo[kn];
          const float* src = ksrc >= 0 ? qkv_b + ksrc + head * d : qkv_bias + head * d;
          pk0[it] = src[C + e];
          pv0[it] = src[2 * C + e];
        }
      } else {
        // lanes 2m / 2m+1 hold the same element pair of keys 2kp / 2kp+1, so that the V^T tile can be written
        // as whole dwords (two keys each) after one lane exchange -- see stage_write
        const int key = 2 * (i / (2 * EP)) + (i & 1), e = 2 * ((i >> 1) % EP);
        const int kn = kt * KT + key;
        pk0[it] = 0.f; pk1[it] = 0.f; pv0[it] = 0.f; pv1[it] = 0.f; pinfo[it] = -1;
        if (i < KT * EP && kn < N) {
          const int ksrc = s_msrc[kn];
          pinfo[it] = s_minfo[kn];
          const float* src = ksrc >= 0 ? qkv_b + ksrc + head * d : qkv_bias + head * d;
          if (e < d) { pk0[it] = src[C + e]; pv0[it] = src[2 * C + e]; }
          if (e + 1 < d) { pk1[it] = src[C + e + 1]; pv1[it] = src[2 * C + e + 1]; }
        }
      }
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      if constexpr (!BF16) {
        if (i < KT * 4 * DT) {
          const int key = i / (4 * DT), e = i % (4 * DT);
          if (e == 0) s_info[key] = pinfo[it];
          s_k[key * LDK + e] = pk0[it];
          s_v[key * LDV + e] = pv0[it];
        }
      } else {
        // V^T[dim][key]: the first version wrote it with two 16-bit stores per thread (element pair e, e+1 of one
        // key): 4-way bank conflicts on sub-dword writes, 13 % of the CU's cycles (profiles/
        // r01_f_swin_attention_sq_counters.txt).  Now the lane pair (2m, 2m+1) = keys (2kp, 2kp+1) swaps halves and
        // each lane writes ONE dword: row e for the even lane, row e+1 for the odd one.
        unsigned vp[NP], kw[NP], vq[NP];
        if constexpr (PREC == 2) {
          split3_pair(pv0[it], pv1[it], vp[0], vp[1], vp[2]);
          split3_pair(pk0[it], pk1[it], kw[0], kw[1], kw[2]);
        } else {
          vp[0] = cvt_pk_bf16(pv0[it], pv1[it]);
          kw[0] = cvt_pk_bf16(pk0[it], pk1[it]);
        }
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) vq[pp] = (unsigned)__shfl_xor((int)vp[pp], 1);
        if (i < KT * EP) {
          const int kbit = i & 1, kp = i / (2 * EP), key = 2 * kp + kbit, e = 2 * ((i >> 1) % EP);
          if (e == 0) s_info[key] = pinfo[it];
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            *reinterpret_cast<unsigned*>(s_kb + (pp * KT + key) * LDKB + e) = kw[pp];
            if (e < 16 * DB) {
              const unsigned word = kbit ? ((vq[pp] >> 16) | (vp[pp] & 0xFFFF0000u)) : ((vp[pp] & 0xFFFFu) | (vq[pp] << 16));
              *reinterpret_cast<unsigned*>(s_vb + (pp * 16 * DB + e + kbit) * LDVB + 2 * kp) = word;
            }
          }
        }
      }
    }
  };
  if constexpr (!BF16) {
    if (16 * DB > 4 * DT) {   // pad columns of V (head_dim not a multiple of 16): zero once, in both buffers
      for (int i = tid; i < 2 * KT * (16 * DB - 4 * DT); i += 256) {
        const int bsel = i / (KT * (16 * DB - 4 * DT)), ii = i % (KT * (16 * DB - 4 * DT));
        const int key = ii / (16 * DB - 4 * DT), e = 4 * DT + ii % (16 * DB - 4 * DT);
        (s_kv + bsel * TILE_FLOATS + KT * LDK)[key * LDV + e] = 0.f;
      }
    }
  }
  // Two tile buffers, ONE barrier per tile: tile kt+1 is written (from the registers stage_load filled during tile kt-1's
  // compute) right after this wave's compute on tile kt, into the buffer every wave finished reading before the barrier
  // that ended iteration kt-1.  (The single-buffer form needed a second barrier per tile -- "everyone done reading" --
  // at which the 4 waves of the workgroup waited for the slowest one twice per 32 keys.)
  stage_load(0);
  stage_write();
  __syncthreads();
  if (ntile > 1) stage_load(1);
  for (int kt = 0; kt < ntile; ++kt) {
    const bool tail = (kt == ntile - 1) && (N % KT != 0);
    set_buf(kt & 1);

#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      // ---- S^T[key][query] for the two 16-key blocks of the tile
      f32x4 sc[2];
      sc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      sc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!BF16) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s = 0; s < DT; ++s) sc[kb] = mfma16x16x4(s_k[(kb * 16 + j) * LDK + 4 * s + g], qreg[sub][s], sc[kb]);
      } else {
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
          u32x4 ka[2][NP];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
              ka[kb][pp] = *reinterpret_cast<const u32x4*>(s_kb + (pp * KT + kb * 16 + j) * LDKB + 32 * ks + 8 * g);
          if constexpr (PREC == 2) {
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
              for (int kb = 0; kb < 2; ++kb)
                sc[kb] = mfma16x16x32_bf16(ka[kb][kPA[term]], qb[sub][ks][kPB[term]], sc[kb]);
          } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sc[kb] = mfma16x16x32_bf16(ka[kb][0], qb[sub][ks][0], sc[kb]);
          }
        }
      }
      // ---- bias + mask (log2 domain): this lane's query is column j, its keys are rows 4g + r
      const int qi = qinfo[sub];
      const int qpl = qi & 0xF, qlat = (qi >> 4) & 0xFF, qlon = (qi >> 12) & 0xFF, qreg_id = (qi >> 20) & 0x1F;
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int4 ki4 = *reinterpret_cast<const int4*>(s_info + kb * 16 + 4 * g);
        const int kia[4] = {ki4.x, ki4.y, ki4.z, ki4.w};
        int idx0 = 0;
        if (LON4) {
          const int ki = kia[0];
          const int kpl = ki & 0xF, klat = (ki >> 4) & 0xFF, klon = (ki >> 12) & 0xFF;
          idx0 = D.bias_mode == 0
                     ? (qlat - klat + D.wlat - 1) * (2 * D.wlon - 1) + (qlon - klon + D.wlon - 1)
                     : ((qpl + kpl * D.wpl) * D.wlat * D.wlat + (qlat + klat * D.wlat)) * (2 * D.wlon - 1) +
                           (qlon - klon + D.wlon - 1);
          if (ki < 0) idx0 = 3;  // keep the speculative reads below in bounds
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ki = kia[r];
          int idx;
          if (LON4) {
            idx = idx0 - r;
          } else {
            const int kpl = ki & 0xF, klat = (ki >> 4) & 0xFF, klon = (ki >> 12) & 0xFF;
            idx = D.bias_mode == 0
                      ? (qlat - klat + D.wlat - 1) * (2 * D.wlon - 1) + (qlon - klon + D.wlon - 1)
                      : ((qpl + kpl * D.wpl) * D.wlat * D.wlat + (qlat + klat * D.wlat)) * (2 * D.wlon - 1) +
                            (qlon - klon + D.wlon - 1);
            if (ki < 0) idx = 0;
          }
          float v = sc[kb][r] + s_tab[idx];
          if (MASK) v += (((ki >> 20) & 0x1F) != qreg_id) ? mask_val : 0.f;
          if (tail) v = ki < 0 ? -1e30f : v;   // only the last tile can hold keys beyond N
          sc[kb][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[sub], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_new);
      m_run[sub] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pexp = __builtin_amdgcn_exp2f(sc[kb][r] - m_new);
          sc[kb][r] = pexp;
          psum += pexp;
        }
      l_run[sub] = l_run[sub] * alpha + psum;
      // the accumulators live in AGPRs: rescaling them costs a read + multiply + write per register.  Once the running
      // maxima of the wave's 16 queries have settled (most tiles after the first few) alpha is exactly 1 everywhere.
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[sub][db] *= alpha;
      }
      // ---- O^T[dim][query] += V^T[dim][key] P^T[key][query]; the keys a lane holds (accumulator rows
      // 4g + r of the two blocks) are exactly the k-slots it supplies: no lane movement
      if constexpr (!BF16) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < DB; ++db)
              oacc[sub][db] = mfma16x16x4(s_v[(kb * 16 + 4 * g + r) * LDV + 16 * db + j], sc[kb][r], oacc[sub][db]);
      } else {
        // k-slot jj of lane group g: jj < 4 -> key 4g + jj, jj >= 4 -> key 16 + 4g + (jj - 4)
        u32x4 pb[NP];
        if constexpr (PREC == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned hh, mm, ll;
            split3_pair(sc[i >> 1][2 * (i & 1)], sc[i >> 1][2 * (i & 1) + 1], hh, mm, ll);
            pb[0][i] = hh;
            pb[1][i] = mm;
            pb[2][i] = ll;
          }
        } else {
          pb[0] = u32x4{cvt_pk_bf16(sc[0][0], sc[0][1]), cvt_pk_bf16(sc[0][2], sc[0][3]),
                        cvt_pk_bf16(sc[1][0], sc[1][1]), cvt_pk_bf16(sc[1][2], sc[1][3])};
        }
        u32x4 va[DB][NP];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const unsigned short* vr = s_vb + (pp * 16 * DB + 16 * db + j) * LDVB + 4 * g;
            const uint2 lo = *reinterpret_cast<const uint2*>(vr);
            const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
            va[db][pp] = u32x4{lo.x, lo.y, hi.x, hi.y};
          }
        if constexpr (PREC == 2) {
#pragma unroll
          for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int db = 0; db < DB; ++db)
              oacc[sub][db] = mfma16x16x32_bf16(va[db][kPA[term]], pb[kPB[term]], oacc[sub][db]);
        } else {
#pragma unroll
          for (int db = 0; db < DB; ++db) oacc[sub][db] = mfma16x16x32_bf16(va[db][0], pb[0], oacc[sub][db]);
        }
      }
    }
    if (kt + 1 < ntile) {
      set_buf((kt + 1) & 1);
      stage_write();
      if (kt + 2 < ntile) stage_load(kt + 2);
    }
    __syncthreads();
  }

  // ---- epilogue: normalise, transpose through LDS (reusing the K/V tile area), store whole rows
  float* s_o = s_kv;  // [64*SUB queries][16*DB + 1]
  constexpr int LDO = 16 * DB + 1;
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub) {
    float l = l_run[sub];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int ql = wave * (16 * SUB) + 16 * sub + j;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_o[ql * LDO + 16 * db + 4 * g + r] = oacc[sub][db][r] * inv;
  }
  __syncthreads();
  float* out_b = out + (long long)b * L * C;
  for (int i = tid; i < 64 * SUB * d; i += 256) {
    const int ql = i / d, e = i % d;
    const int qn = blockIdx.x * (64 * SUB) + ql;
    if (qn < N) {
      const long long dst = token_dest(D, ipl, ilat, ilon, qn);
      if (dst >= 0) out_b[dst * C + head * d + e] = s_o[ql * LDO + e];
    }
  }
}

}  // namespace wattn
}  // namespace dlwp

namespace dlwp {   // window_attn2.hip: the 2-D (Swin) fast path
size_t wattn2_workspace_bytes(const dlwp_wattn_desc* u, int batch, int np);
int32_t wattn2_run(const dlwp_wattn_desc* u, const float* qkv, const float* table, float* out, int batch, void* workspace,
                   size_t workspace_bytes, hipStream_t s, int np);
const int* wattn2_fallback_counter(const dlwp_wattn_desc* u, int batch, int np, const void* workspace);
// window_attn3.hip: the small 3-D windows of Pangu's earth-specific attention (bias_mode 1, head_dim 32)
size_t wattn3_workspace_bytes(const dlwp_wattn_desc* u, int batch, int np);
int32_t wattn3_run(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias, const float* table, float* out, int batch,
                   void* workspace, size_t workspace_bytes, hipStream_t s, int np);
}

using namespace dlwp;
using namespace dlwp::wattn;

template <int DT, int DB, int SUB, int PREC, bool LON4>
static int32_t launch_wattn_k(const Desc& D, const float* qkv, const float* qkv_bias, const float* table, float* out,
                              int batch, long long L, hipStream_t s) {
  const int nwin = D.npl * D.nlat * D.nlon;
  constexpr int KT = 32, LDK = 4 * DT + 2, LDV = 16 * DB + 4, DK = (4 * DT + 31) / 32, LDKB = 32 * DK + 8, LDVB = KT + 8;
  constexpr int NP = PREC == 2 ? 3 : 1;
  constexpr int KV_FLOATS = PREC ? (NP * (KT * LDKB + 16 * DB * LDVB) + 1) / 2 : KT * (LDK + LDV);
  const size_t tab = (size_t)((D.table_rows + 3) & ~3);
  const size_t tile_f = 2 * ((size_t)((KV_FLOATS + 3) & ~3) + KT), epi_f = (size_t)64 * SUB * (16 * DB + 1);
  const size_t lds = (tab + (((tile_f > epi_f ? tile_f : epi_f) + 3) & ~(size_t)3) + 2 * (size_t)((D.N + 3) & ~3)) * 4;
  DLWP_REQUIRE(lds <= 160 * 1024, DLWP_ERR_UNSUPPORTED, "window attention needs %zu bytes of LDS (bias table too large)", lds);
  const dim3 grid((D.N + 64 * SUB - 1) / (64 * SUB), D.heads, batch * nwin);
  auto kern = D.use_mask ? window_attn_kernel<DT, DB, SUB, PREC, LON4, true> : window_attn_kernel<DT, DB, SUB, PREC, LON4, false>;
  if (lds > 48 * 1024)
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, D, qkv, qkv_bias, table, out, L);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

template <int DT, int DB>
static int32_t launch_wattn(const Desc& D, const float* qkv, const float* qkv_bias, const float* table, float* out,
                            int batch, long long L, hipStream_t s, int prec) {
  // queries per workgroup = 64 * SUB.  Large windows: 128; small windows: the whole window in one
  // workgroup when it fits 256 queries, so the bias column / token map staging is paid once per
  // (window, head) instead of once per 64 queries (Pangu: N = 144 -> SUB = 3)
  int sub = D.N >= 512 ? 2 : (D.N <= 64 ? 1 : (D.N <= 128 ? 2 : (D.N <= 192 ? 3 : 4)));
  static const char* sub_env = getenv("DLWP_WATTN_SUB");   // A/B: 16-query sub-tiles per wave for large windows
  if (sub_env && D.N >= 512 && atoi(sub_env) >= 1 && atoi(sub_env) <= 4) sub = atoi(sub_env);
  const bool lon4 = (D.wlon % 4) == 0;
#define DLWP_WA2(SUB_, BF_) \
  do { if (lon4) return launch_wattn_k<DT, DB, SUB_, BF_, true>(D, qkv, qkv_bias, table, out, batch, L, s); \
       else return launch_wattn_k<DT, DB, SUB_, BF_, false>(D, qkv, qkv_bias, table, out, batch, L, s); } while (0)
#define DLWP_WA(SUB_) do { if (prec == 2) DLWP_WA2(SUB_, 2); else if (prec == 1) DLWP_WA2(SUB_, 1); else DLWP_WA2(SUB_, 0); } while (0)
  switch (sub) {
    case 1: DLWP_WA(1);
    case 2: DLWP_WA(2);
    case 3: DLWP_WA(3);
    default: DLWP_WA(4);
  }
#undef DLWP_WA
#undef DLWP_WA2
}

static int32_t window_attn_impl(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias, const float* table,
                                float* out, int32_t batch, void* stream, int prec) {
  DLWP_REQUIRE(u && qkv && table && out, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0, DLWP_ERR_INVALID_ARGUMENT, "batch must be positive");
  Desc D;
  {
    const int32_t rc = make_desc(u, qkv_bias, D);
    if (rc != DLWP_OK) return rc;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long L = (long long)D.pl * D.lat * D.lon;
  DLWP_REQUIRE(L * 3 * D.C < (1ll << 31), DLWP_ERR_UNSUPPORTED, "window attention: %lld tokens x 3 x %d channels overflow the 31-bit token offsets", L, D.C);
  const float* qb = qkv_bias ? qkv_bias : qkv;  // never dereferenced when nothing is padded
  switch (D.d / 4) {
#define DLWP_CASE(DT_) \
  case DT_: return launch_wattn<DT_, (4 * DT_ + 15) / 16>(D, qkv, qb, table, out, batch, L, s, prec);
    DLWP_CASE(2) DLWP_CASE(4) DLWP_CASE(6) DLWP_CASE(8) DLWP_CASE(12) DLWP_CASE(16)
#undef DLWP_CASE
    default: break;
  }
  return fail(DLWP_ERR_UNSUPPORTED, "head_dim %d not supported", D.d);
}

extern "C" size_t dlwp_window_attn_workspace_bytes(const dlwp_wattn_desc* u, int32_t batch, int32_t bf16) {
  if (!u || batch <= 0) return 0;
  if (!bf16 && u->form == 0) return 0;     // the fp32-MFMA form is the generic kernel
  const size_t w2 = wattn2_workspace_bytes(u, batch, bf16 ? 1 : 3);
  return w2 ? w2 : wattn3_workspace_bytes(u, batch, bf16 ? 1 : 3);
}

extern "C" int32_t dlwp_window_attn_fallbacks(const dlwp_wattn_desc* u, int32_t batch, int32_t bf16, const void* workspace,
                                              void* stream, int32_t* count) {
  DLWP_REQUIRE(u && count, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  *count = 0;
  const int* dev = wattn2_fallback_counter(u, batch, bf16 ? 1 : 3, workspace);
  if (!dev) return DLWP_OK;   // generic kernel: exact running maximum, no fallback exists
  DLWP_HIP_CHECK(hipMemcpyAsync(count, dev, sizeof(int32_t), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
  DLWP_HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  return DLWP_OK;
}

extern "C" int32_t dlwp_window_attn_f32(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias,
                                        const float* table, float* out, int32_t batch, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  // fp32-accurate either way; desc->form picks the form of the two contractions: 0 fp32 MFMA, 1 bf16x6, -1 by window
  // size as measured (whole-map windows, >= 512 tokens: fp32 MFMA; small windows: bf16x6).  Each is the other's cross-check.
  DLWP_REQUIRE(u, DLWP_ERR_INVALID_ARGUMENT, "null descriptor");
  DLWP_REQUIRE(u->form >= -1 && u->form <= 1, DLWP_ERR_INVALID_ARGUMENT, "form %d not in {-1, 0, 1}", u->form);
  const int n_win = u->window[0] * u->window[1] * u->window[2];
  const bool x6 = u->form > 0 || (u->form < 0 && n_win < 512);
  if (u->form != 0 && qkv && table && out && batch > 0) {
    // 2-D windows (every Swin block): the second-generation kernel (bf16x6 operands, bias through the accumulator,
    // masked tiles skipped); returns 1 when the descriptor or the workspace does not fit -> generic kernel below
    int32_t rc = wattn2_run(u, qkv, table, out, batch, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), 3);
    if (rc != 1) return rc;
    // small 3-D windows with the earth-specific bias (every Pangu block): the third kernel, same convention
    rc = wattn3_run(u, qkv, qkv_bias, table, out, batch, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), 3);
    if (rc != 1) return rc;
  }
  return window_attn_impl(u, qkv, qkv_bias, table, out, batch, stream, x6 ? 2 : 0);
}

extern "C" int32_t dlwp_window_attn_bf16(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias,
                                         const float* table, float* out, int32_t batch, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (u && qkv && table && out && batch > 0) {
    int32_t rc = wattn2_run(u, qkv, table, out, batch, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), 1);
    if (rc != 1) return rc;
    rc = wattn3_run(u, qkv, qkv_bias, table, out, batch, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), 1);
    if (rc != 1) return rc;
  }
  return window_attn_impl(u, qkv, qkv_bias, table, out, batch, stream, 1);
}

// bf16 form with bfloat16 TENSORS: qkv [B, L, 3 C], the qkv bias [3 C] (read for zero-padded tokens) and the output [B, L, C] are
// bf16.  Only the two fast kernels take this form (every Swin / Pangu block of the reference configurations); for any other
// descriptor the call returns DLWP_ERR_UNSUPPORTED and the caller hands fp32 tensors to dlwp_window_attn_bf16.
extern "C" int32_t dlwp_window_attn_bf16_io(const dlwp_wattn_desc* u, const void* qkv_bf16, const void* qkv_bias_bf16,
                                            const float* table, void* out_bf16, int32_t batch, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  DLWP_REQUIRE(u && qkv_bf16 && table && out_bf16, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0, DLWP_ERR_INVALID_ARGUMENT, "batch must be positive");
  int32_t rc = wattn2_run(u, reinterpret_cast<const float*>(qkv_bf16), table, reinterpret_cast<float*>(out_bf16), batch, workspace,
                          workspace_bytes, reinterpret_cast<hipStream_t>(stream), 16);
  if (rc != 1) return rc;
  rc = wattn3_run(u, reinterpret_cast<const float*>(qkv_bf16), reinterpret_cast<const float*>(qkv_bias_bf16), table,
                  reinterpret_cast<float*>(out_bf16), batch, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream), 16);
  if (rc != 1) return rc;
  return fail(DLWP_ERR_UNSUPPORTED, "window attention with bfloat16 tensors: descriptor not covered by the fast kernels");
}
