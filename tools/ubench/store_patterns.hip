// Microbenchmark: per-CU store throughput of 16-byte-per-lane stores as a function of how a wave instruction's
// 64 pieces are laid over 128-byte lines.  Each wave writes 8 KB "tiles" (64 pixels x 128 B) with 8 dwordx4 stores.
//   mode 0: lane = pixel, instruction k writes piece k of every pixel      (64 lines touched / instruction)
//   mode 1: 2 lanes per 32 B  (lane pair contiguous)                       (32 lines / instruction)
//   mode 2: 4 lanes per 64 B                                               (16 lines / instruction)
//   mode 3: 8 lanes per 128 B line                                         (8 lines / instruction)
//   mode 4: fully contiguous 1 KB per instruction (same as 3 but lines consecutive)
// build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE> __global__ __launch_bounds__(256) void k_store(char* out, int tiles_per_wave, int nwaves_total) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  i32x4 v = {lane, wave, 3, 4};
  for (int t = 0; t < tiles_per_wave; ++t) {
    const unsigned base = (unsigned)((t * nwaves_total + wave)) * 8192u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned off;
      if (MODE == 0) off = lane * 128 + k * 16;
      else if (MODE == 1) off = ((lane >> 1) + 32 * (k >> 2)) * 128 + ((k & 3) * 2 + (lane & 1)) * 16;
      else if (MODE == 2) off = ((lane >> 2) + 16 * (k >> 1)) * 128 + ((k & 1) * 4 + (lane & 3)) * 16;
      else if (MODE == 3) off = ((lane >> 3) * 8 + k) * 128 + (lane & 7) * 16;   // 8 lines per instr, strided by 8 lines
      else off = k * 1024 + lane * 16;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + off, 0, 0);
    }
    v.z += t;
  }
}

int main(int argc, char** argv) {
  const int wpw = 4;
  const int tiles = argc > 1 ? atoi(argv[1]) : 9;      // 8 KB tiles per wave
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;      // workgroups (<= 256: one per CU)
  const size_t bytes = (size_t)wgs * wpw * tiles * 8192;
  char* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) {
        switch (mode) {
          case 0: k_store<0><<<wgs, 256>>>(d, tiles, wgs * wpw); break;
          case 1: k_store<1><<<wgs, 256>>>(d, tiles, wgs * wpw); break;
          case 2: k_store<2><<<wgs, 256>>>(d, tiles, wgs * wpw); break;
          case 3: k_store<3><<<wgs, 256>>>(d, tiles, wgs * wpw); break;
          default: k_store<4><<<wgs, 256>>>(d, tiles, wgs * wpw); break;
        }
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms / 20 < best) best = ms / 20;
    }
    printf("mode %d: %.2f us per launch, %.1f MB -> %.2f TB/s  (%.1f B/clk/CU at 2.0 GHz)\n", mode, best * 1e3, bytes / 1e6,
           bytes / (best * 1e-3) / 1e12, bytes / (double)wgs / (best * 1e-3 * 2.0e9));
  }
  return 0;
}
