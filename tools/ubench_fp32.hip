// Micro-benchmark: does fp32 MFMA overlap with fp32 VALU on gfx950?  What does packed fp32 cost?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_fp32.hip -o /tmp/ubench_fp32 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 4096

__global__ void k_fma(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkfma(float* out, float a, float b) {
  f32x2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f32x2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av), "v"(bv));
  }
  f32x2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_mfma(float* out, float a, float b) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
// 4 MFMA + NV independent VALU fma per iteration, interleaved
template <int NV>
__global__ void k_mix(float* out, float a, float b) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
  for (int i = 0; i < ITERS; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 8]) : "v"(a), "v"(b));
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 8]) : "v"(a), "v"(b));
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 8]) : "v"(a), "v"(b));
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % 8]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s;
}
__global__ void k_exp(float* out, float a) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < ITERS; ++i) {
    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <class F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 256 * 16 * 1024 * 4);
  for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 1 wave/SIMD; wps blocks per CU
    const int blocks = 256 * wps;
    auto rep = [&](const char* name, float ms, double instr_per_wave) {
      // cycles per wave-instruction per SIMD, assuming 2.4 GHz: ms * 2.4e6 cycles / (instr_per_wave * wps)
      printf("wps=%d %-14s %8.3f ms  -> %6.2f cyc/instr/SIMD @2.4GHz\n", wps, name, ms, ms * 2.4e6 / (instr_per_wave * wps));
    };
    rep("v_fma_f32", timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 8.0);
    rep("v_pk_fma_f32", timeit([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 8.0);
    rep("v_exp_f32", timeit([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }), ITERS * 4.0);
    rep("mfma16x16x4", timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
    rep("mfma+0valu", timeit([&] { hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
    rep("mfma+2valu", timeit([&] { hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
    rep("mfma+4valu", timeit([&] { hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
    rep("mfma+8valu", timeit([&] { hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
    rep("mfma+16valu", timeit([&] { hipLaunchKernelGGL(k_mix<16>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), ITERS * 4.0);
  }
  return 0;
}
