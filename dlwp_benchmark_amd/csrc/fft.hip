// Batched 2-D real FFTs of channels-first fields for the AFNO filter (reference fourcastnet.py:87 `rfft2` and
// :122-123 `irfft2`), through hipFFT/rocFFT directly instead of torch.fft.
// Why not torch.fft: per call it adds a full-size device-to-device clone in front of the C2R transform ("C2R may
// overwrite its input"), layout copies around the multi-dimensional transforms and one elementwise pass for the
// "ortho" scaling -- three 268 MB memcpys and two scaling kernels per AFNO block at 128x256x32x64, 470 of the
// block's 2300 us.  Here the transforms are UNNORMALISED (the 1/sqrt(HW) factors ride in the mixing kernel,
// dlwp_afno2d_mix_scaled_f32), the forward one writes straight into the buffer the mixing kernel updates in place,
// and the inverse one is allowed to destroy that buffer.
// The FFT itself stays a library call by design (at 128x256 with every row and half the columns
// kept, a pruned DFT does not pay).
#include <hipfft/hipfft.h>

#include "common.hpp"

struct dlwp_fft2_plan {
  hipfftHandle r2c = 0, c2r = 0;
  bool have_r2c = false, have_c2r = false;
  int batch = 0, H = 0, W = 0;
};

using namespace dlwp;

#define DLWP_FFT_CHECK(expr)                                                                          \
  do {                                                                                                \
    hipfftResult _r = (expr);                                                                         \
    if (_r != HIPFFT_SUCCESS)                                                                         \
      return ::dlwp::fail(DLWP_ERR_HIP, "%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
  } while (0)

extern "C" int32_t dlwp_fft2_plan_create(dlwp_fft2_plan** out, int32_t batch, int32_t H, int32_t W) {
  DLWP_REQUIRE(out, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_REQUIRE(batch > 0 && H > 0 && W > 1, DLWP_ERR_INVALID_ARGUMENT, "bad shape");
  auto* pl = new dlwp_fft2_plan;
  pl->batch = batch;
  pl->H = H;
  pl->W = W;
  int n[2] = {H, W};
  int rembed[2] = {H, W}, cembed[2] = {H, W / 2 + 1};
  hipfftResult r = hipfftPlanMany(&pl->r2c, 2, n, rembed, 1, H * W, cembed, 1, H * (W / 2 + 1), HIPFFT_R2C, batch);
  if (r == HIPFFT_SUCCESS) {
    pl->have_r2c = true;
    r = hipfftPlanMany(&pl->c2r, 2, n, cembed, 1, H * (W / 2 + 1), rembed, 1, H * W, HIPFFT_C2R, batch);
    if (r == HIPFFT_SUCCESS) pl->have_c2r = true;
  }
  if (r != HIPFFT_SUCCESS) {
    if (pl->have_r2c) (void)hipfftDestroy(pl->r2c);
    delete pl;
    return fail(DLWP_ERR_HIP, "hipfftPlanMany(%d x [%d, %d]) failed: hipfftResult %d", batch, H, W, (int)r);
  }
  *out = pl;
  return DLWP_OK;
}

extern "C" int32_t dlwp_fft2_plan_destroy(dlwp_fft2_plan* plan) {
  if (!plan) return DLWP_OK;
  if (plan->have_r2c) (void)hipfftDestroy(plan->r2c);
  if (plan->have_c2r) (void)hipfftDestroy(plan->c2r);
  delete plan;
  return DLWP_OK;
}

extern "C" int32_t dlwp_rfft2_f32(const dlwp_fft2_plan* plan, const float* x_dev, float* xf_dev, void* stream) {
  DLWP_REQUIRE(plan && x_dev && xf_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_FFT_CHECK(hipfftSetStream(plan->r2c, reinterpret_cast<hipStream_t>(stream)));
  DLWP_FFT_CHECK(hipfftExecR2C(plan->r2c, const_cast<float*>(x_dev), reinterpret_cast<hipfftComplex*>(xf_dev)));
  return DLWP_OK;
}

extern "C" int32_t dlwp_irfft2_f32(const dlwp_fft2_plan* plan, float* yf_dev, float* y_dev, void* stream) {
  DLWP_REQUIRE(plan && yf_dev && y_dev, DLWP_ERR_INVALID_ARGUMENT, "null argument");
  DLWP_FFT_CHECK(hipfftSetStream(plan->c2r, reinterpret_cast<hipStream_t>(stream)));
  DLWP_FFT_CHECK(hipfftExecC2R(plan->c2r, reinterpret_cast<hipfftComplex*>(yf_dev), y_dev));
  return DLWP_OK;
}
