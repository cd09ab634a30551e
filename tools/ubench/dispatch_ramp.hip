// How long does the dispatcher take to place one workgroup per CU?  256 workgroups of T threads and L bytes of LDS; thread 0 of each writes the
// 100 MHz clock at entry (and spins ~8 us so that every workgroup of the launch is resident at once, like the persistent / one-tile-per-CU
// kernels of this library).  Output: last entry - first entry per (T, L), median of 20 launches.
// build: hipcc --offload-arch=gfx950 -O3 dispatch_ramp.hip -o dispatch_ramp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
__global__ void k(unsigned long long* t, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    t[blockIdx.x] = t0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(4);
    lds[0] = 1;
  }
  __syncthreads();
}
struct Big { int v[96]; };                       // 384 bytes of kernel arguments, like srk_conv_args / srk_conv_pair_args
__global__ __launch_bounds__(512) void kv(unsigned long long* t, int spin, Big b) {   // ... and a full register file: v255 / a255 touched
  extern __shared__ char lds[];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    t[blockIdx.x] = t0 + (b.v[95] & 0);
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(4);
    lds[0] = 1;
  }
  __syncthreads();
}
int main() {
  {
    unsigned long long* d; hipMalloc(&d, 4096 * 8);
    std::vector<unsigned long long> h(4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&kv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int T : {256, 512}) {
      std::vector<double> ramps, meds;
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(kv, dim3(256), dim3(T), 152 * 1024, 0, d, 800, Big{});
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin(), h.begin() + 256);
        std::sort(v.begin(), v.end());
        ramps.push_back((v.back() - v.front()) / 100.0); meds.push_back((v[128] - v.front()) / 100.0);
      }
      std::sort(ramps.begin(), ramps.end()); std::sort(meds.begin(), meds.end());
      printf("256 workgroups x %4d threads, 152 KB LDS, 256 registers per lane, 384 B of arguments: last entry %.2f us after the first (median %.2f us)\n", T, ramps[10], meds[10]);
    }
  }
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  std::vector<unsigned long long> h(4096);
  const int Ts[] = {64, 128, 256, 512, 1024}, Ls[] = {0, 64 * 1024, 152 * 1024};
  for (int L : Ls) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int T : Ts) for (int nwg : {256, 512}) {
      if (nwg == 512 && L > 64 * 1024) continue;
      std::vector<double> ramps, meds;
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(k, dim3(nwg), dim3(T), L, 0, d, 800);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin(), h.begin() + nwg);
        std::sort(v.begin(), v.end());
        ramps.push_back((v.back() - v.front()) / 100.0); meds.push_back((v[nwg / 2] - v.front()) / 100.0);
      }
      std::sort(ramps.begin(), ramps.end()); std::sort(meds.begin(), meds.end());
      printf("%4d workgroups x %4d threads, %6d B LDS: last entry %.2f us after the first (median entry %.2f us)\n", nwg, T, L, ramps[10], meds[10]);
    }
  }
  return 0;
}
