// How does the dispatcher place 256 workgroups of 512 threads with ~123 KB of LDS each (one fits per CU)?
// Records (xcc, se, cu) and start/end time of every workgroup of a ~10 us dummy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned long long* rec, int spin) {
  extern __shared__ float smem[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + smem[(threadIdx.x + i) & 1023];
  smem[threadIdx.x] = acc;
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    rec[blockIdx.x * 4 + 0] = t0;
    rec[blockIdx.x * 4 + 1] = t1;
    rec[blockIdx.x * 4 + 2] = hwid;
    rec[blockIdx.x * 4 + 3] = xcc;
  }
}
int main(int argc, char** argv) {
  const int configs[][3] = {{256, 512, 123 * 1024}, {512, 256, 70 * 1024}, {512, 256, 88 * 1024}, {512, 256, 50 * 1024}, {256, 1024, 60 * 1024}};
  unsigned long long* rec; hipMalloc(&rec, 4096 * 32);
  std::vector<unsigned long long> h(4096 * 4);
  for (auto& c : configs) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, c[2]);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(c[0]), dim3(c[1]), c[2], 0, rec, 4000);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), rec, c[0] * 32, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_cu;
    unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    for (int b = 0; b < c[0]; ++b) {
      unsigned hw = (unsigned)h[b * 4 + 2], xcc = (unsigned)h[b * 4 + 3] & 0xF;
      unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
      if (h[b * 4] < tmin) tmin = h[b * 4];
      if (h[b * 4 + 1] > tmax) tmax = h[b * 4 + 1];
      if (h[b * 4] > smax) smax = h[b * 4];
    }
    int hist[8] = {0};
    for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
    printf("grid %d x %d thr, LDS %d KB: distinct CUs %zu, WGs-per-CU histogram 1:%d 2:%d 3:%d 4:%d; span %.1f us, last start at %.1f us\n",
           c[0], c[1], c[2] / 1024, per_cu.size(), hist[1], hist[2], hist[3], hist[4], (tmax - tmin) / 100.0, (smax - tmin) / 100.0);
  }
  return 0;
}
