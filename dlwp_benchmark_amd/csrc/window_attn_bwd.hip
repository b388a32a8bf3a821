// Backward of the fused (shifted-)window attention (window_attn.hip) on MI355X (gfx950): what `loss.backward()` of
// reference scripts/train.py:263-271 runs through WindowAttention.forward (swin_transformer.py:122-154, with the pad / roll /
// partition / reverse / crop of :217-251) and EarthAttention3D.forward (panguweather.py:176-211, :285-316), minus their two
// Linears.  Flash-style: the N x N scores are RECOMPUTED tile by tile from q, k, v and per-row statistics and never exist in
// memory (round 2 recomputed the operator with torch operators, materialising [B, nH, N, N] three times).
//
//   s_ij = scale q_i.k_j + bias[idx(i, j)] + mask_ij        p_ij = softmax_j(s_ij)        o_i = sum_j p_ij v_j
//   dv_j = sum_i p_ij dO_i            dp_ij = dO_i.v_j            delta_i = sum_j p_ij dp_ij  (= dO_i.o_i)
//   ds_ij = p_ij (dp_ij - delta_i)    dq_i = scale sum_j ds_ij k_j    dk_j = scale sum_i ds_ij q_i    dbias[idx(i, j)] += ds_ij
//
// Two kernels over the same index arithmetic as the forward (window_attn_desc.hpp: zero-padded tokens carry the qkv bias as
// q = k = v, forward and backward rolls may differ, outputs of padded / cropped positions have no gradient):
//   stats   (query-owned): row maximum m_i, row sum l_i of exp(s - m), delta_i -- one online-softmax sweep over the keys;
//   main    (key-owned):   a workgroup keeps 32 keys' k, v and their dk, dv in registers / LDS and sweeps the query tiles;
//                          dq goes out through float atomics (each query row receives one add per key block), the bias-table
//                          gradient is accumulated in an LDS copy of the (type, head) column and flushed once per workgroup,
//                          gradients of zero-padded tokens are added to the qkv-bias gradient.
// All arithmetic is fp32 on the vector unit (5 d FMAs per score: the minimum); gradients match autograd of the reference to
// fp32 rounding (tests/test_window_attn_bwd_gpu.py, tests/test_training_gpu.py: fixtures of the real classes, bound 1e-4).
#include "common.hpp"
#include "window_attn_desc.hpp"

namespace dlwp {
namespace wattn {

constexpr int BT = 32;               // tile edge: 32 queries x 32 keys
constexpr float kLog2e = 1.4426950408889634f;

// per-row statistics: [(b * nwin + win) * heads + head][N][3] = {m (log2 domain), l, delta}
struct BwdArgs {
  const float* qkv;        // [B][L][3][heads][d]
  const float* qkv_bias;   // [3 C] or null (never read when nothing is padded)
  const float* table;
  const float* dout;       // [B][L][C]
  float* dqkv;             // [B][L][3 C]   zeroed by the entry point (dq arrives through atomics)
  float* dbias;            // [3 C] or null
  float* dtable;           // like table, zeroed by the entry point
  float* stats;
  long long L;
};

struct RowInfo {
  int src;     // element offset of the token's qkv row inside the sample, or -1 (zero-padded)
  int dst;     // element offset of the token's output row inside the sample, or -1 (cropped)
  int info;    // pack_info
};

__device__ __forceinline__ RowInfo row_info(const Desc& D, int ipl, int ilat, int ilon, int n) {
  RowInfo r;
  if (n >= D.N) { r.src = -1; r.dst = -1; r.info = -1; return r; }
  const Coord c = token_coord(D, ipl, ilat, ilon, n);
  const long long dst = token_dest(D, ipl, ilat, ilon, n);
  r.src = c.src >= 0 ? (int)c.src * 3 * D.C : -1;
  r.dst = dst >= 0 ? (int)dst * D.C : -1;
  r.info = pack_info(c);
  return r;
}

__device__ __forceinline__ void window_of(const Desc& D, int wi, int& b, int& ipl, int& ilat, int& ilon) {
  const int nwin = D.npl * D.nlat * D.nlon;      // same decomposition as window_attn_kernel (blockIdx.z)
  b = wi / nwin;
  wi -= b * nwin;
  ilat = wi % D.nlat;
  const int t2 = wi / D.nlat;
  ipl = t2 % D.npl;
  ilon = t2 / D.npl;
}

// s (log2 domain) of one (query, key) pair from the packed coordinates: + bias + mask
__device__ __forceinline__ float score_terms(const Desc& D, const float* s_tab, int qi, int ki, int& idx) {
  idx = bias_index(D, qi & 0xF, (qi >> 4) & 0xFF, (qi >> 12) & 0xFF, ki & 0xF, (ki >> 4) & 0xFF, (ki >> 12) & 0xFF);
  float v = s_tab[idx];
  if (D.use_mask && (((qi >> 20) & 0x1F) != ((ki >> 20) & 0x1F))) v += -100.0f * kLog2e;
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// stats: thread (qi = t >> 3, tj = t & 7) sweeps keys tj, tj + 8, ... of every tile for query qi of the block
// ---------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void wattn_bwd_stats_kernel(const Desc D, const BwdArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LD = DH + 1;
  const int tid = threadIdx.x, qi = tid >> 3, tj = tid & 7;
  const int head = blockIdx.y;
  int b, ipl, ilat, ilon;
  window_of(D, blockIdx.z, b, ipl, ilat, ilon);
  const int N = D.N, C = D.C;
  float* s_tab = smem;                                    // [table_rows] bias column * log2 e
  float* s_k = s_tab + ((D.table_rows + 3) & ~3);         // [BT][LD]
  float* s_v = s_k + BT * LD;
  int* s_kinfo = reinterpret_cast<int*>(s_v + BT * LD);   // [BT]
  {
    const int type = ipl * D.nlat + ilat;
    for (int i = tid; i < D.table_rows; i += 256) s_tab[i] = A.table[table_offset(D, i, type, head)] * kLog2e;
  }
  const float* qkv_b = A.qkv + (long long)b * A.L * 3 * C;
  const float* do_b = A.dout + (long long)b * A.L * C;
  const int qn = blockIdx.x * BT + qi;
  const RowInfo rq = row_info(D, ipl, ilat, ilon, qn);
  const bool qlive = qn < N;
  float q[DH], dO[DH];
  {
    const float* src = rq.src >= 0 ? qkv_b + rq.src + head * DH : A.qkv_bias + head * DH;
    const float qs = D.scale * kLog2e;
#pragma unroll
    for (int e = 0; e < DH; ++e) {
      q[e] = qlive ? src[e] * qs : 0.f;
      dO[e] = (qlive && rq.dst >= 0) ? do_b[rq.dst + head * DH + e] : 0.f;
    }
  }
  float m = -1e30f, l = 0.f, dacc = 0.f;
  const int ntile = (N + BT - 1) / BT;
  for (int kt = 0; kt < ntile; ++kt) {
    __syncthreads();
    {   // stage the key tile: thread (key = tid >> 3, dims tid & 7, + 8, ...)
      const int key = tid >> 3, kn = kt * BT + key;
      const RowInfo rk = row_info(D, ipl, ilat, ilon, kn);
      const float* src = rk.src >= 0 ? qkv_b + rk.src + head * DH : A.qkv_bias + head * DH;
      for (int e = tid & 7; e < DH; e += 8) {
        s_k[key * LD + e] = kn < N ? src[C + e] : 0.f;
        s_v[key * LD + e] = kn < N ? src[2 * C + e] : 0.f;
      }
      if ((tid & 7) == 0) s_kinfo[key] = rk.info;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < BT / 8; ++c) {
      const int jj = tj + 8 * c, ki = s_kinfo[jj];
      if (ki < 0 || !qlive) continue;
      int idx;
      float s = score_terms(D, s_tab, rq.info, ki, idx), dp = 0.f;
#pragma unroll
      for (int e = 0; e < DH; ++e) {
        s = __builtin_fmaf(q[e], s_k[jj * LD + e], s);
        dp = __builtin_fmaf(dO[e], s_v[jj * LD + e], dp);
      }
      const float mn = fmaxf(m, s);
      const float a = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
      l = l * a + p;
      dacc = dacc * a + p * dp;
      m = mn;
    }
  }
  // combine the 8 partial sweeps of a query (consecutive lanes)
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    const float mo = __shfl_xor(m, off), lo = __shfl_xor(l, off), d_o = __shfl_xor(dacc, off);
    const float mn = fmaxf(m, mo);
    const float a = __builtin_amdgcn_exp2f(m - mn), bq = __builtin_amdgcn_exp2f(mo - mn);
    l = l * a + lo * bq;
    dacc = dacc * a + d_o * bq;
    m = mn;
  }
  if (tj == 0 && qlive) {
    float* st = A.stats + (((long long)blockIdx.z * D.heads + head) * N + qn) * 3;
    st[0] = m;
    st[1] = l;
    st[2] = dacc / l;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// main: one workgroup = 32 keys of one (batch, window, head)
// ---------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void wattn_bwd_main_kernel(const Desc D, const BwdArgs A) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LD = DH + 1, LP = BT + 1, NE = DH / 8;      // NE dims per thread in the (row, dim-chunk) phases
  const int tid = threadIdx.x, r8 = tid >> 3, c8 = tid & 7;
  const int head = blockIdx.y;
  int b, ipl, ilat, ilon;
  window_of(D, blockIdx.z, b, ipl, ilat, ilon);
  const int N = D.N, C = D.C;
  const int trp = (D.table_rows + 3) & ~3;
  float* s_tab = smem;                   // [trp]  bias column * log2 e
  float* s_dtab = s_tab + trp;           // [trp]  gradient of the bias column (natural units)
  float* s_k = s_dtab + trp;             // [BT][LD]  this block's keys
  float* s_v = s_k + BT * LD;
  float* s_q = s_v + BT * LD;            // [BT][LD]  query tile, scaled by scale * log2 e
  float* s_do = s_q + BT * LD;
  float* s_p = s_do + BT * LD;           // [BT][LP]  p_ij   (query i, key j)
  float* s_ds = s_p + BT * LP;           // [BT][LP]  ds_ij
  float* s_st = s_ds + BT * LP;          // [BT][4]   m, 1 / l, delta
  int* s_kinfo = reinterpret_cast<int*>(s_st + BT * 4);    // [BT]
  int* s_ksrc = s_kinfo + BT;
  int* s_qinfo = s_ksrc + BT;
  int* s_qsrc = s_qinfo + BT;
  const int type = ipl * D.nlat + ilat;
  for (int i = tid; i < trp; i += 256) {
    s_tab[i] = i < D.table_rows ? A.table[table_offset(D, i, type, head)] * kLog2e : 0.f;
    s_dtab[i] = 0.f;
  }
  const float* qkv_b = A.qkv + (long long)b * A.L * 3 * C;
  const float* do_b = A.dout + (long long)b * A.L * C;
  float* dqkv_b = A.dqkv + (long long)b * A.L * 3 * C;
  {   // this block's keys
    const int kn = blockIdx.x * BT + r8;
    const RowInfo rk = row_info(D, ipl, ilat, ilon, kn);
    const float* src = rk.src >= 0 ? qkv_b + rk.src + head * DH : A.qkv_bias + head * DH;
    for (int e = c8; e < DH; e += 8) {
      s_k[r8 * LD + e] = kn < N ? src[C + e] : 0.f;
      s_v[r8 * LD + e] = kn < N ? src[2 * C + e] : 0.f;
    }
    if (c8 == 0) { s_kinfo[r8] = rk.info; s_ksrc[r8] = rk.src; }
  }
  float dk[NE], dv[NE];
#pragma unroll
  for (int m = 0; m < NE; ++m) { dk[m] = 0.f; dv[m] = 0.f; }
  const float qs = D.scale * kLog2e;
  const float* stats = A.stats + ((long long)blockIdx.z * D.heads + head) * N * 3;
  const int ntile = (N + BT - 1) / BT;
  for (int qt = 0; qt < ntile; ++qt) {
    __syncthreads();     // the previous tile's phases are done with s_q / s_do / s_p / s_ds
    {   // stage the query tile
      const int qn = qt * BT + r8;
      const RowInfo rq = row_info(D, ipl, ilat, ilon, qn);
      const float* src = rq.src >= 0 ? qkv_b + rq.src + head * DH : A.qkv_bias + head * DH;
      for (int e = c8; e < DH; e += 8) {
        s_q[r8 * LD + e] = qn < N ? src[e] * qs : 0.f;
        s_do[r8 * LD + e] = (qn < N && rq.dst >= 0) ? do_b[rq.dst + head * DH + e] : 0.f;
      }
      if (c8 == 0) {
        s_qinfo[r8] = rq.info;
        s_qsrc[r8] = rq.src;
        const bool live = qn < N;
        s_st[r8 * 4 + 0] = live ? stats[qn * 3 + 0] : 0.f;
        s_st[r8 * 4 + 1] = live ? 1.0f / stats[qn * 3 + 1] : 0.f;
        s_st[r8 * 4 + 2] = live ? stats[qn * 3 + 2] : 0.f;
      }
    }
    __syncthreads();
    // ---- phase A: thread (query r8, keys c8 + 8 c): p and ds of its four scores, bias-column gradient
    {
      const int qi = s_qinfo[r8];
      const float mrow = s_st[r8 * 4 + 0], linv = s_st[r8 * 4 + 1], delta = s_st[r8 * 4 + 2];
#pragma unroll
      for (int c = 0; c < BT / 8; ++c) {
        const int jj = c8 + 8 * c, ki = s_kinfo[jj];
        float p = 0.f, ds = 0.f;
        if (qi >= 0 && ki >= 0) {
          int idx;
          float s = score_terms(D, s_tab, qi, ki, idx), dp = 0.f;
#pragma unroll
          for (int e = 0; e < DH; ++e) {
            s = __builtin_fmaf(s_q[r8 * LD + e], s_k[jj * LD + e], s);
            dp = __builtin_fmaf(s_do[r8 * LD + e], s_v[jj * LD + e], dp);
          }
          p = __builtin_amdgcn_exp2f(s - mrow) * linv;
          ds = p * (dp - delta);
          atomicAdd(&s_dtab[idx], ds);
        }
        s_p[r8 * LP + jj] = p;
        s_ds[r8 * LP + jj] = ds;
      }
    }
    __syncthreads();
    // ---- phase B: thread (key r8, dims c8 + 8 m): dv_j += p_ij dO_i, dk_j += ds_ij q_i (q carries scale * log2 e)
#pragma unroll 4
    for (int i = 0; i < BT; ++i) {
      const float p = s_p[i * LP + r8], ds = s_ds[i * LP + r8];
#pragma unroll
      for (int m = 0; m < NE; ++m) {
        dv[m] = __builtin_fmaf(p, s_do[i * LD + c8 + 8 * m], dv[m]);
        dk[m] = __builtin_fmaf(ds, s_q[i * LD + c8 + 8 * m], dk[m]);
      }
    }
    // ---- phase C: thread (query r8, dims c8 + 8 m): dq_i += scale ds_ij k_j over this block's keys
    {
      float dq[NE];
#pragma unroll
      for (int m = 0; m < NE; ++m) dq[m] = 0.f;
#pragma unroll 4
      for (int jj = 0; jj < BT; ++jj) {
        const float ds = s_ds[r8 * LP + jj];
#pragma unroll
        for (int m = 0; m < NE; ++m) dq[m] = __builtin_fmaf(ds, s_k[jj * LD + c8 + 8 * m], dq[m]);
      }
      if (s_qinfo[r8] >= 0) {
        const int qsrc = s_qsrc[r8];
        float* dst = qsrc >= 0 ? dqkv_b + qsrc + head * DH : (A.dbias ? A.dbias + head * DH : nullptr);
        if (dst) {
#pragma unroll
          for (int m = 0; m < NE; ++m) atomicAdd(dst + c8 + 8 * m, dq[m] * D.scale);
        }
      }
    }
  }
  // ---- this block's dk, dv: a real token sits in exactly one window position -> plain stores; padded tokens -> qkv-bias gradient
  if (s_kinfo[r8] >= 0) {
    const int ksrc = s_ksrc[r8];
    const float kscale = 1.0f / kLog2e;        // s_q carried scale * log2 e: ds * q * scale = ds * s_q / log2 e
    if (ksrc >= 0) {
#pragma unroll
      for (int m = 0; m < NE; ++m) {
        dqkv_b[ksrc + C + head * DH + c8 + 8 * m] = dk[m] * kscale;
        dqkv_b[ksrc + 2 * C + head * DH + c8 + 8 * m] = dv[m];
      }
    } else if (A.dbias) {
#pragma unroll
      for (int m = 0; m < NE; ++m) {
        atomicAdd(A.dbias + C + head * DH + c8 + 8 * m, dk[m] * kscale);
        atomicAdd(A.dbias + 2 * C + head * DH + c8 + 8 * m, dv[m]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < D.table_rows; i += 256) {
    const float v = s_dtab[i];
    if (v != 0.f) atomicAdd(A.dtable + table_offset(D, i, type, head), v);
  }
}

template <int DH>
static int32_t launch_bwd(const Desc& D, const BwdArgs& A, int batch, hipStream_t s) {
  const int nwin = D.npl * D.nlat * D.nlon;
  const int trp = (D.table_rows + 3) & ~3;
  const size_t lds_stats = ((size_t)trp + 2 * BT * (DH + 1) + BT) * 4;
  const size_t lds_main = ((size_t)2 * trp + 4 * BT * (DH + 1) + 2 * BT * (BT + 1) + BT * 4 + 4 * BT) * 4;
  DLWP_REQUIRE(lds_main <= 160 * 1024, DLWP_ERR_UNSUPPORTED,
               "window attention backward needs %zu bytes of LDS (bias table of %d rows too large)", lds_main, D.table_rows);
  const dim3 grid((D.N + BT - 1) / BT, D.heads, batch * nwin);
  auto ks = wattn_bwd_stats_kernel<DH>;
  auto km = wattn_bwd_main_kernel<DH>;
  if (lds_stats > 48 * 1024)
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stats));
  if (lds_main > 48 * 1024)
    DLWP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(km), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_main));
  hipLaunchKernelGGL(ks, grid, dim3(256), lds_stats, s, D, A);
  DLWP_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(km, grid, dim3(256), lds_main, s, D, A);
  DLWP_HIP_CHECK(hipGetLastError());
  return DLWP_OK;
}

}  // namespace wattn
}  // namespace dlwp

using namespace dlwp;
using namespace dlwp::wattn;

extern "C" size_t dlwp_window_attn_bwd_workspace_bytes(const dlwp_wattn_desc* u, int32_t batch) {
  if (!u || batch <= 0) return 0;
  const long long n = (long long)u->window[0] * u->window[1] * u->window[2];
  const long long nwin = (long long)(u->padded[0] / (u->window[0] > 0 ? u->window[0] : 1)) *
                         (u->padded[1] / (u->window[1] > 0 ? u->window[1] : 1)) * (u->padded[2] / (u->window[2] > 0 ? u->window[2] : 1));
  return (size_t)(batch * nwin * u->heads * n * 3 * 4 + 256);
}

extern "C" int32_t dlwp_window_attn_bwd_f32(const dlwp_wattn_desc* u, const float* qkv, const float* qkv_bias, const float* table,
                                            const float* grad_out, float* grad_qkv, float* grad_qkv_bias, float* grad_table,
                                            int32_t batch, void* workspace, size_t workspace_bytes, void* stream) {
  DLWP_REQUIRE(u && qkv && table && grad_out && grad_qkv && grad_table && workspace, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0, DLWP_ERR_INVALID_ARGUMENT, "batch must be positive");
  Desc D;
  {
    const int32_t rc = make_desc(u, qkv_bias, D);
    if (rc != DLWP_OK) return rc;
  }
  const bool padded = (D.ppl != D.pl) || (D.plat != D.lat) || (D.plon != D.lon);
  DLWP_REQUIRE(!padded || grad_qkv_bias, DLWP_ERR_INVALID_ARGUMENT,
               "padded windows: zero-padded tokens carry the qkv bias, its gradient buffer is needed");
  DLWP_REQUIRE(workspace_bytes >= dlwp_window_attn_bwd_workspace_bytes(u, batch), DLWP_ERR_INVALID_ARGUMENT, "workspace too small");
  const long long L = (long long)D.pl * D.lat * D.lon;
  DLWP_REQUIRE(L * 3 * D.C < (1ll << 31), DLWP_ERR_UNSUPPORTED, "window attention: %lld tokens x 3 x %d channels overflow the 31-bit token offsets", L, D.C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t table_elems = (size_t)D.table_rows * (D.bias_mode ? D.types : 1) * D.heads;
  DLWP_HIP_CHECK(hipMemsetAsync(grad_qkv, 0, (size_t)batch * L * 3 * D.C * 4, s));
  DLWP_HIP_CHECK(hipMemsetAsync(grad_table, 0, table_elems * 4, s));
  if (grad_qkv_bias) DLWP_HIP_CHECK(hipMemsetAsync(grad_qkv_bias, 0, (size_t)3 * D.C * 4, s));
  BwdArgs A;
  A.qkv = qkv; A.qkv_bias = qkv_bias ? qkv_bias : qkv; A.table = table; A.dout = grad_out;
  A.dqkv = grad_qkv; A.dbias = grad_qkv_bias; A.dtable = grad_table; A.stats = reinterpret_cast<float*>(workspace); A.L = L;
  switch (D.d) {
    case 8: return launch_bwd<8>(D, A, batch, s);
    case 16: return launch_bwd<16>(D, A, batch, s);
    case 24: return launch_bwd<24>(D, A, batch, s);
    case 32: return launch_bwd<32>(D, A, batch, s);
    case 48: return launch_bwd<48>(D, A, batch, s);
    case 64: return launch_bwd<64>(D, A, batch, s);
    default: break;
  }
  return fail(DLWP_ERR_UNSUPPORTED, "head_dim %d not supported", D.d);
}
