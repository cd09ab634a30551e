// Microbenchmark: sustained issue rate of v_mfma_f32_32x32x16_bf16 with every CU busy (power-limited clocks), in
// shader cycles per MFMA (s_memtime) and in wall time.  Variants: register operands only; plus one ds_read_b128 per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int LDSREAD> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ i32x4 lds[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) lds[i] = i32x4{i, i + 1, i + 2, i + 3};
  __syncthreads();
  i32x4 a = {tid, 1, 2, 3}, b = {4, tid, 6, 7};
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 36; ++u) {
      if (LDSREAD) { a = lds[(tid + u * 7) & 1023]; }
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
      if (LDSREAD) { b = lds[(tid + u * 5 + 3) & 1023]; }
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c1, 0, 0, 0);
      if (LDSREAD) { a = lds[(tid + u * 3 + 1) & 1023]; }
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c2, 0, 0, 0);
      if (LDSREAD) { b = lds[(tid + u * 11 + 2) & 1023]; }
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  out[blockIdx.x * 256 + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

int main() {
  const int wgs = 256, iters = 40;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, wgs * 256 * 4); hipMalloc(&cyc, wgs * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) { if (v) k<1><<<wgs, 256>>>(out, cyc, iters); else k<0><<<wgs, 256>>>(out, cyc, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double sum = 0; unsigned long long mn = ~0ull, mx = 0;
      for (int i = 0; i < 1024; ++i) { sum += h[i]; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
      const double n = 144.0 * iters;
      printf("%s: cycles/MFMA avg %.2f min %.2f max %.2f | %.1f us per launch -> %.0f TFLOP/s, implied clock %.2f GHz\n",
             v ? "1 ds_read_b128 per MFMA" : "register operands      ", sum / 1024 / n, mn / n, mx / n, ms * 100,
             256.0 * 4 * n * 32768 * 2 / 2 / (ms / 10 * 1e-3) / 1e12 * 1.0, (sum / 1024) / (ms / 10 * 1e-3) / 1e9);
    }
  }
  return 0;
}
