// Geometry of a (shifted-)window attention call, shared by the forward kernels (window_attn.hip) and the backward kernels
// (window_attn_bwd.hip): the descriptor of include/dlwp_hip.h normalised, and the index arithmetic that replaces pad / roll /
// window_partition / window_reverse / crop and the mask build (swin_transformer.py:217-251, :383-401; panguweather.py:285-316,
// utils/shift_window_mask.py, utils/earth_position_index.py).
#pragma once
#include "common.hpp"

namespace dlwp {
namespace wattn {

struct Desc {
  int pl, lat, lon;           // un-padded token grid, L = pl*lat*lon
  int ppl, plat, plon;        // padded grid
  int pad_f, pad_t, pad_l;    // leading pads
  int wpl, wlat, wlon;        // window
  int npl, nlat, nlon;        // windows per dimension
  int sf[3];                  // forward roll:  shifted[p] = padded[(p + sf) mod dim]
  int sb[3];                  // backward roll: out_padded[(p + sb) mod dim] = attn_shifted[p]
  int use_mask;               // 0/-100 shift mask on
  int b1[3], b2[3];           // region id along a dim = (p >= b1) + (p >= b2), p in the shifted frame
  int bias_mode;              // 0: Swin 2-D relative table [(2Wh-1)(2Ww-1)][nH]
                              // 1: Pangu earth table [wpl^2 wlat^2 (2wlon-1)][types][nH]
  int heads, d, C;            // C = heads*d
  int N;                      // tokens per window
  int table_rows, types;
  float scale;
};

struct Coord {
  int zpl, zlat, zlon;  // in-window coordinates
  int region;           // shift-mask region id
  long long src;        // token index in [0, L) or -1 for a zero-padded token
};

__device__ __forceinline__ Coord token_coord(const Desc& D, int ipl, int ilat, int ilon, int n) {
  Coord c;
  c.zlon = n % D.wlon;
  const int t = n / D.wlon;
  c.zlat = t % D.wlat;
  c.zpl = t / D.wlat;
  const int P = ipl * D.wpl + c.zpl, A = ilat * D.wlat + c.zlat, O = ilon * D.wlon + c.zlon;
  const int rp = (P >= D.b1[0]) + (P >= D.b2[0]);
  const int ra = (A >= D.b1[1]) + (A >= D.b2[1]);
  const int ro = (O >= D.b1[2]) + (O >= D.b2[2]);
  c.region = (rp * 3 + ra) * 3 + ro;
  const int sp = (P + D.sf[0]) % D.ppl - D.pad_f;
  const int sa = (A + D.sf[1]) % D.plat - D.pad_t;
  const int so = (O + D.sf[2]) % D.plon - D.pad_l;
  const bool ok = sp >= 0 && sp < D.pl && sa >= 0 && sa < D.lat && so >= 0 && so < D.lon;
  c.src = ok ? ((long long)sp * D.lat + sa) * D.lon + so : -1;
  return c;
}

__device__ __forceinline__ long long token_dest(const Desc& D, int ipl, int ilat, int ilon, int n) {
  const int zlon = n % D.wlon;
  const int t = n / D.wlon;
  const int zlat = t % D.wlat, zpl = t / D.wlat;
  const int dp = (ipl * D.wpl + zpl + D.sb[0]) % D.ppl - D.pad_f;
  const int da = (ilat * D.wlat + zlat + D.sb[1]) % D.plat - D.pad_t;
  const int dq = (ilon * D.wlon + zlon + D.sb[2]) % D.plon - D.pad_l;
  const bool ok = dp >= 0 && dp < D.pl && da >= 0 && da < D.lat && dq >= 0 && dq < D.lon;
  return ok ? ((long long)dp * D.lat + da) * D.lon + dq : -1;
}

__device__ __forceinline__ int pack_info(const Coord& c) {
  return (c.zpl & 0xF) | ((c.zlat & 0xFF) << 4) | ((c.zlon & 0xFF) << 12) | ((c.region & 0x1F) << 20);
}


// dlwp_wattn_desc -> Desc, with the argument checks every entry point shares
inline int32_t make_desc(const dlwp_wattn_desc* u, const float* qkv_bias, Desc& D) {
  D.pl = u->grid[0]; D.lat = u->grid[1]; D.lon = u->grid[2];
  D.wpl = u->window[0]; D.wlat = u->window[1]; D.wlon = u->window[2];
  DLWP_REQUIRE(D.pl > 0 && D.lat > 0 && D.lon > 0 && D.wpl > 0 && D.wlat > 0 && D.wlon > 0, DLWP_ERR_INVALID_ARGUMENT,
               "grid and window must be positive");
  D.pad_f = u->pad_lead[0]; D.pad_t = u->pad_lead[1]; D.pad_l = u->pad_lead[2];
  D.ppl = u->padded[0]; D.plat = u->padded[1]; D.plon = u->padded[2];
  DLWP_REQUIRE(D.ppl % D.wpl == 0 && D.plat % D.wlat == 0 && D.plon % D.wlon == 0, DLWP_ERR_INVALID_ARGUMENT,
               "padded grid (%d,%d,%d) is not a multiple of the window (%d,%d,%d)", D.ppl, D.plat, D.plon, D.wpl,
               D.wlat, D.wlon);
  DLWP_REQUIRE(D.ppl >= D.pl + D.pad_f && D.plat >= D.lat + D.pad_t && D.plon >= D.lon + D.pad_l,
               DLWP_ERR_INVALID_ARGUMENT, "padded grid smaller than grid + leading pad");
  D.npl = D.ppl / D.wpl; D.nlat = D.plat / D.wlat; D.nlon = D.plon / D.wlon;
  for (int i = 0; i < 3; ++i) {
    const int dim = i == 0 ? D.ppl : (i == 1 ? D.plat : D.plon);
    D.sf[i] = ((u->shift_fwd[i] % dim) + dim) % dim;
    D.sb[i] = ((u->shift_back[i] % dim) + dim) % dim;
    D.b1[i] = u->mask_b1[i];
    D.b2[i] = u->mask_b2[i];
  }
  D.use_mask = u->use_mask;
  D.bias_mode = u->bias_mode;
  D.heads = u->heads; D.d = u->head_dim; D.C = u->heads * u->head_dim;
  D.N = D.wpl * D.wlat * D.wlon;
  D.scale = u->scale;
  DLWP_REQUIRE(D.wpl <= 16 && D.wlat <= 256 && D.wlon <= 256, DLWP_ERR_UNSUPPORTED, "window too large for the packed coordinates");
  if (D.bias_mode == 0) {
    DLWP_REQUIRE(D.wpl == 1, DLWP_ERR_INVALID_ARGUMENT, "Swin bias mode needs a 2-D window");
    D.table_rows = (2 * D.wlat - 1) * (2 * D.wlon - 1);
    D.types = 1;
  } else {
    D.table_rows = D.wpl * D.wpl * D.wlat * D.wlat * (2 * D.wlon - 1);
    D.types = D.npl * D.nlat;
  }
  const bool padded = (D.ppl != D.pl) || (D.plat != D.lat) || (D.plon != D.lon);
  DLWP_REQUIRE(!padded || qkv_bias, DLWP_ERR_INVALID_ARGUMENT, "padded windows need the qkv bias (zero-padded tokens carry it)");
  DLWP_REQUIRE(D.d == 8 || D.d == 16 || D.d == 24 || D.d == 32 || D.d == 48 || D.d == 64, DLWP_ERR_UNSUPPORTED,
               "head_dim %d: kernels are instantiated for 8, 16, 24, 32, 48, 64", D.d);
  return DLWP_OK;
}

// element offset of bias-table entry `idx` for (window type, head): Swin [rows][heads], Pangu [rows][types][heads]
__device__ __forceinline__ long long table_offset(const Desc& D, int idx, int type, int head) {
  return D.bias_mode ? ((long long)idx * D.types + type) * D.heads + head : (long long)idx * D.heads + head;
}

// relative-position (Swin) / earth-specific (Pangu) bias index of (query, key) from their in-window coordinates
__device__ __forceinline__ int bias_index(const Desc& D, int qpl, int qlat, int qlon, int kpl, int klat, int klon) {
  return D.bias_mode == 0
             ? (qlat - klat + D.wlat - 1) * (2 * D.wlon - 1) + (qlon - klon + D.wlon - 1)
             : ((qpl + kpl * D.wpl) * D.wlat * D.wlat + (qlat + klat * D.wlat)) * (2 * D.wlon - 1) + (qlon - klon + D.wlon - 1);
}

}  // namespace wattn
}  // namespace dlwp
