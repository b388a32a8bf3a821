// Shared host/device helpers for libdlwp_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dlwp_hip.h"

namespace dlwp {

// ---------------------------------------------------------------------------------------------
// error reporting: thread-local message, int status across the ABI
// ---------------------------------------------------------------------------------------------
std::string& last_error_slot();
int32_t fail(int32_t code, const char* fmt, ...);

#define DLWP_HIP_CHECK(expr)                                                                    \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::dlwp::fail(DLWP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                          __FILE__, __LINE__);                                                  \
  } while (0)

#define DLWP_REQUIRE(cond, code, ...)                                                           \
  do {                                                                                          \
    if (!(cond)) return ::dlwp::fail(code, __VA_ARGS__);                                        \
  } while (0)

// RAII for device allocations owned by plans
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
    o.p = nullptr;
    o.bytes = 0;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t upload(const void* host, size_t n, hipStream_t s) {
    if (p) {
      (void)hipFree(p);
      p = nullptr;
    }
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 4);
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpyAsync(p, host, n, hipMemcpyHostToDevice, s);
    return e;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 FMA chain.
//   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
//   D register r of lane l is D[row = 4*(l>>4) + r][col = l&15].
__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// GELU (exact, erf form: torch.nn.functional.gelu default) without a libm call.
//   gelu(x) = 0.5 x (1 + erf(x/sqrt2));  1 + erf(z) = erfc(-z) = 2 - erfc(z)
//   erfc(t) for t >= 0 is evaluated as exp2(t * P(t)) with a degree-8 polynomial fitted to
//   -log2(erfc(t))/t on [0, 4.5] (t clamped there: erfc(4.5) = 2e-10).  Max abs error of the
//   GELU value 1.7e-7 (fp32 round-off level of the exp), see tests/test_gelu_poly.py.
__device__ __forceinline__ float gelu_erf(float x) {
  const float t = fminf(fabsf(x) * 0.70710678118654752f, 4.5f);
  float p = 5.642222277e-06f;               // = -log2(e) * c8 ... c0 (Horner, highest first)
  p = fmaf(p, t, -9.264355322e-05f);
  p = fmaf(p, t, 6.046944181e-04f);
  p = fmaf(p, t, -1.767261187e-03f);
  p = fmaf(p, t, -5.040322430e-04f);
  p = fmaf(p, t, 2.810628898e-02f);
  p = fmaf(p, t, -1.484391242e-01f);
  p = fmaf(p, t, -9.184220433e-01f);
  p = fmaf(p, t, -1.627908349e+00f);
  const float e = __builtin_amdgcn_exp2f(p * t);  // erfc(|x|/sqrt2)
  const float hx = 0.5f * x;
  const float he = hx * e;
  return x >= 0.0f ? (x - he) : he;
}

}  // namespace dlwp
