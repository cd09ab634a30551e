// Prototype measurement for VERDICT r4 item 1(c): what does the halo hand-off of a PERSISTENT residual-chain kernel cost per block?
//
// The batch-16 chain today: one conv_pair launch per ResBlock (256 workgroups = 16 images x 4 x 4 tiles of 14 x 14 output pixels, one per
// CU), 17.5k in-kernel cycles + ~1.7 us between dependent launches.  A persistent form keeps a workgroup on its tile across blocks and
// replaces launch boundary + prologue DMA wait (1.7 + 1.5 us) by an exchange per block: publish the own 14 x 14 x 64-channel tile (25 KB),
// tell the neighbours, wait for the (up to) 8 neighbours' tiles of the same block, read their 2-pixel ring (16 KB).
//
// This program runs exactly that exchange between 256 resident workgroups of 512 threads (one per CU: 152 KB of dynamic LDS like conv_pair)
// around a stand-in for the block's compute (a spin of `work` shader cycles, +-`jitter`, so that neighbours arrive unevenly), in the
// placement-independent form of MI355X_MICROARCH.md's "valid forms" table:
//   producer: every byte of the tile by `buffer_store_dwordx4 sc1` (whole 128-byte pixels per 8 lanes), every storing wave
//             `s_waitcnt vmcnt(0)`, workgroup barrier, ONE lane stores the flag `sc1`
//   consumer: lanes 0..7 of wave 0 poll their neighbour's flag with `sc1` loads -- BOUNDED: after `spin_max` polls the lane sets an error
//             flag and goes on (a hang is a design choice) --, workgroup barrier, ring by `buffer_load_dwordx4 sc1`
// and checks every ring word against what the neighbour must have written for that block.  Output: microseconds per block with and
// without the exchange, the difference = what the persistent form pays per block; errors / timeouts must be 0.
//
// build: hipcc --offload-arch=gfx950 -O3 halo_exchange.hip -o halo_exchange       run: ./halo_exchange [blocks] [work_cycles] [jitter]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int TO = 14, TPI = 4, NIMG = 16, HW = 48, NWG = NIMG * TPI * TPI;
constexpr int AUX_SC1 = 16;                                // cache-policy bit 4 = sc1 on gfx940+ (agent scope: bypass / write through L1 and the XCD's L2)

__device__ __forceinline__ unsigned word_of(int blk, int n, int y, int x, int c) { return (unsigned)(blk * 0x9e3779b1u) ^ (unsigned)(((n * HW + y) * HW + x) * 8 + c); }

__global__ __launch_bounds__(512) void chain_kernel(char* act0, char* act1, unsigned* flags, unsigned* err, unsigned long long* stamps,
                                                    int nblocks, int work, int jitter, int exchange, int spin_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, n = wg / (TPI * TPI), ty = (wg / TPI) % TPI, tx = wg % TPI;
  const int y0 = ty * TO, x0 = tx * TO;
  const unsigned bytes = (unsigned)NIMG * HW * HW * 128;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(act0, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(act1, 0, bytes, 0x00020000);
  unsigned long long t_work = 0, t_pub = 0, t_wait = 0, t_ring = 0;
  unsigned bad = 0, timeouts = 0;
  for (int blk = 0; blk < nblocks; ++blk) {
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
    // ---- the block's compute: a spin of `work` cycles, unevenly long per workgroup and block ----
    const int mine = work + ((int)((unsigned)(wg * 2654435761u + blk * 40503u) >> 16) % (2 * jitter + 1)) - jitter;
    while ((long long)(__builtin_amdgcn_s_memtime() - s0) < mine) __builtin_amdgcn_s_sleep(2);
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
    if (exchange) {
      const __amdgpu_buffer_rsrc_t rw = (blk & 1) ? r1 : r0;
      // ---- publish: the 14 x 14 tile, whole pixels (8 lanes x 16 B), write-through ----
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + 512 * k, p = i >> 3, c = i & 7, row = p / TO, col = p - row * TO;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = i < TO * TO * 8 && gy < HW && gx < HW;
        const unsigned w = word_of(blk, n, gy, gx, c);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{w, w + 1, w + 2, w + 3}, rw, ok ? (unsigned)((((n * HW + gy) * HW + gx) * 8 + c) * 16) : 0x80000000u, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave, before the barrier
      __syncthreads();
      if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(blk + 1), __builtin_amdgcn_make_buffer_rsrc(flags, 0, NWG * 4, 0x00020000), wg * 4, 0, AUX_SC1);
      const unsigned long long s2 = __builtin_amdgcn_s_memtime();
      // ---- wait for the neighbours' tiles of THIS block: one lane per neighbour, bounded ----
      if (wave == 0 && lane < 8) {
        const int dy = (lane < 3) ? -1 : (lane < 5 ? 0 : 1), dx = (lane < 3) ? lane - 1 : (lane < 5 ? (lane == 3 ? -1 : 1) : lane - 6);
        const int ny = ty + dy, nx = tx + dx;
        if (ny >= 0 && ny < TPI && nx >= 0 && nx < TPI) {
          const int nb = (n * TPI + ny) * TPI + nx;
          const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(flags, 0, NWG * 4, 0x00020000);
          int spins = 0;
          while (__builtin_amdgcn_raw_buffer_load_b32(rf, nb * 4, 0, AUX_SC1) < (unsigned)(blk + 1)) {
            if (++spins >= spin_max) { ++timeouts; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      __syncthreads();                                     // between the poll and EVERY load of the handed-off bytes
      const unsigned long long s3 = __builtin_amdgcn_s_memtime();
      // ---- the 2-pixel ring of the 18 x 18 input tile around the own 14 x 14: 128 pixels x 8 chunks = 1,024 pieces, 2 per thread ----
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + 512 * k, q = i >> 3, c = i & 7;
        // ring pixel q: rows 0,1,16,17 (4 x 18 = 72 pixels) then columns 0,1,16,17 of rows 2..15 (14 x 4 = 56)
        int ry, rx;
        if (q < 72) { ry = (q / 18 < 2) ? q / 18 : q / 18 + 14; rx = q % 18; }
        else { const int u = q - 72; ry = 2 + u / 4; rx = (u % 4 < 2) ? u % 4 : u % 4 + 14; }
        const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
        const bool ok = gy >= 0 && gy < HW && gx >= 0 && gx < HW;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? (unsigned)((((n * HW + gy) * HW + gx) * 8 + c) * 16) : 0x80000000u, 0, AUX_SC1);
        *reinterpret_cast<u32x4*>(smem + i * 16) = v;
        if (ok && v.x != word_of(blk, n, gy, gx, c)) ++bad;
      }
      const unsigned long long s4 = __builtin_amdgcn_s_memtime();
      t_pub += s2 - s1; t_wait += s3 - s2; t_ring += s4 - s3;
    }
    t_work += s1 - s0;
    __syncthreads();
  }
  if (bad) atomicAdd(err, bad);
  if (timeouts) atomicAdd(err + 1, timeouts);
  if (wg == 85 && tid == 0) { stamps[0] = t_work; stamps[1] = t_pub; stamps[2] = t_wait; stamps[3] = t_ring; }   // an interior tile
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int nblocks = argc > 1 ? atoi(argv[1]) : 32, work = argc > 2 ? atoi(argv[2]) : 17500, jitter = argc > 3 ? atoi(argv[3]) : 1500;
  const size_t bytes = (size_t)NIMG * HW * HW * 128;
  char *a0, *a1; unsigned *flags, *err; unsigned long long* stamps;
  CK(hipMalloc(&a0, bytes)); CK(hipMalloc(&a1, bytes)); CK(hipMalloc(&flags, NWG * 4)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&stamps, 64));
  const int lds = 152 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%d workgroups x 512 threads, %d blocks per launch, compute stand-in %d +- %d cycles per block\n", NWG, nblocks, work, jitter);
  double us[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep)
    for (int ex = 0; ex < 2; ++ex) {
      CK(hipMemset(flags, 0, NWG * 4)); CK(hipMemset(err, 0, 8)); CK(hipMemset(stamps, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain_kernel, dim3(NWG), dim3(512), lds, 0, a0, a1, flags, err, stamps, nblocks, work, jitter, ex, 200000);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr[2]; unsigned long long hs[4];
      CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, stamps, 32, hipMemcpyDeviceToHost));
      us[ex] = ms * 1e3 / nblocks;
      printf("%s: %8.2f us per block  (wrong ring words %u, poll timeouts %u; interior tile, cycles per block: compute %llu, publish %llu, wait %llu, ring %llu)\n",
             ex ? "with the halo exchange   " : "compute stand-in only    ", us[ex], herr[0], herr[1], hs[0] / nblocks, hs[1] / nblocks, hs[2] / nblocks, hs[3] / nblocks);
    }
  printf("exchange = %.2f us per block; a persistent chain saves the launch boundary (~1.7 us) and the prologue's DMA wait (~1.5 us) per block\n", us[1] - us[0]);
  return 0;
}
