// Weight packing (OIHW fp32 -> MFMA shadow layout), weight-gradient unpacking, NCHW<->NHWC conversion,
// and the library's error/introspection entry points.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include "srk_common.h"

static thread_local char g_err[512] = "";

void srk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* srk_last_error(void) { return g_err; }
extern "C" int srk_version(void) { return 100; }
extern "C" int srk_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return p.multiProcessorCount;
}

namespace {

// PixelShuffle permutation on the output-channel axis: packed co' = (i*r+j)*C + c  <->  torch co = c*r*r + i*r + j
__device__ __forceinline__ int ps_unperm(int cop, int Cout, int r) {
  if (r <= 1) return cop;
  const int r2 = r * r, Cc = Cout / r2;
  const int ij = cop / Cc, c = cop - ij * Cc;
  return c * r2 + ij;
}

// One work item = one packed 16-byte chunk position (row, cc) for ALL taps: the thread reads the CH x KH*KW source floats of its
// chunk -- forward: CH input channels x taps are one contiguous run of w[co][ci][kh][kw]; dgrad: CH runs of `taps` floats, and
// the 64 rows of a wave are consecutive input channels, i.e. one contiguous run per output channel across the lanes -- and
// writes one 16-byte chunk per tap, neighbouring rows next to each other.  (Round 1 walked the OUTPUT elements one by one:
// every 4-byte read pulled its own cache line, 0.76 ms per step for EDSR-large's 43 M parameters.)
template <int DT> __device__ void pack_body(const srk_pack_args& a, long long total, long long first, long long stride) {
  typedef DTraits<DT> Tr;
  constexpr int CH = Tr::CH;
  typename Tr::elem* out = reinterpret_cast<typename Tr::elem*>(a.wpk);
  const int nch = a.KinP / CH;
  const int blk = (a.CoutP % 64 == 0) ? 64 : 32;
  const int taps = a.KH * a.KW;
  const long long items = (long long)a.CoutP * nch;
  if (taps <= 9) {
    for (long long it = first; it < items; it += stride) {
      const int row = (int)(it % a.CoutP), cc = (int)(it / a.CoutP);
      // MFMA row -> stored output channel (srk_common.h row_to_chan): per 64-row block, or per 32-row block when
      // the padded row count is not a multiple of 64 (the 32-row kernel tile)
      const int chan = (row / blk) * blk + row_to_chan(row % blk, blk);
      float v[9][CH];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < CH; ++e) v[t][e] = 0.f;
      if (!a.dgrad) {
        // rows = output channels (permuted for pixel shuffle), k = input channel
        if (chan < a.Cout) {
          const int co = ps_unperm(chan, a.Cout, a.ps_r);
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            const int k = cc * CH + e;
            if (k < a.Cin) {
              const float* src = a.w + ((size_t)co * a.Cin + k) * taps;
#pragma unroll
              for (int t = 0; t < 9; ++t) if (t < taps) v[t][e] = src[t];
            }
          }
        }
      } else {
        // rows = input channels, k = output channel in dy's storage order, taps flipped
        if (chan < a.Cin) {
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            const int k = cc * CH + e;
            if (k < a.Cout) {
              const int co = ps_unperm(k, a.Cout, a.ps_r);
              const float* src = a.w + ((size_t)co * a.Cin + chan) * taps;
#pragma unroll
              for (int t = 0; t < 9; ++t) if (t < taps) v[t][e] = src[taps - 1 - t];
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < taps) {
          typename Tr::elem* dst = out + (((size_t)t * nch + cc) * a.CoutP + row) * CH;
          if constexpr (Tr::IS16) {
            i32x4 q;
            q.x = (int)((uint32_t)Tr::from_f32(v[t][0]) | ((uint32_t)Tr::from_f32(v[t][1]) << 16));
            q.y = (int)((uint32_t)Tr::from_f32(v[t][2]) | ((uint32_t)Tr::from_f32(v[t][3]) << 16));
            q.z = (int)((uint32_t)Tr::from_f32(v[t][4]) | ((uint32_t)Tr::from_f32(v[t][5]) << 16));
            q.w = (int)((uint32_t)Tr::from_f32(v[t][6]) | ((uint32_t)Tr::from_f32(v[t][7]) << 16));
            *reinterpret_cast<i32x4*>(dst) = q;
          } else {
#pragma unroll
            for (int e = 0; e < CH; ++e) dst[e] = Tr::from_f32(v[t][e]);
          }
        }
      }
    }
  } else {
    // any other kernel size: element by element
    for (long long idx = first; idx < total; idx += stride) {
      // idx = ((tap*nch + cc)*CoutP + row)*CH + e
      const int e = (int)(idx % CH);
      long long t = idx / CH;
      const int row = (int)(t % a.CoutP);
      t /= a.CoutP;
      const int cc = (int)(t % nch);
      const int tap = (int)(t / nch);
      const int k = cc * CH + e;           // reduction-channel index
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      const int chan = (row / blk) * blk + row_to_chan(row % blk, blk);
      float v = 0.f;
      if (!a.dgrad) {
        if (chan < a.Cout && k < a.Cin) {
          const int co = ps_unperm(chan, a.Cout, a.ps_r);
          v = a.w[(((size_t)co * a.Cin + k) * a.KH + kh) * a.KW + kw];
        }
      } else {
        if (chan < a.Cin && k < a.Cout) {
          const int co = ps_unperm(k, a.Cout, a.ps_r);
          v = a.w[(((size_t)co * a.Cin + chan) * a.KH + (a.KH - 1 - kh)) * a.KW + (a.KW - 1 - kw)];
        }
      }
      out[idx] = Tr::from_f32(v);
    }
  }
  if (a.rows_layout && !a.dgrad) {       // the (kw, co)-rows layout of the direct large-kernel forward, behind the standard one
    typename Tr::elem* rows = out + total;
    const long long extra = (long long)a.KH * 2048;
    for (long long j = first; j < extra; j += stride) {
      const int e = (int)(j & 7), lane = (int)((j >> 3) & 63), ks = (int)((j >> 9) & 3), kh = (int)(j >> 11);
      const int m = lane & 31, kw = m / a.Cout, co = m - kw * a.Cout, ci = 16 * ks + 8 * (lane >> 5) + e;
      float v = 0.f;
      if (kw < a.KW && ci < a.Cin) v = a.w[(((size_t)co * a.Cin + ci) * a.KH + kh) * a.KW + kw];
      rows[j] = Tr::from_f32(v);
    }
  }
  if (a.bias_pk && !a.dgrad) {
    for (long long i = first; i < a.CoutP; i += stride) {
      const int chan = (int)(i / blk) * blk + row_to_chan((int)(i % blk), blk);     // bias_pk is indexed by MFMA row
      float b = 0.f;
      if (a.bias && chan < a.Cout) b = a.bias[ps_unperm(chan, a.Cout, a.ps_r)];
      a.bias_pk[i] = b;
    }
  }
}

template <int DT> __global__ void pack_kernel(const srk_pack_args a, long long total) {
  pack_body<DT>(a, total, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// one launch for a whole model: blockIdx.y = table entry, blockIdx.x strides over its elements
__global__ void pack_group_kernel(const srk_pack_args* __restrict__ table) {
  const srk_pack_args a = table[blockIdx.y];
  const long long total = (long long)a.KH * a.KW * a.KinP * a.CoutP;
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
  switch (a.dtype) {
    case SRK_BF16: pack_body<SRK_BF16>(a, total, first, stride); break;
    case SRK_F16: pack_body<SRK_F16>(a, total, first, stride); break;
    default: pack_body<SRK_F32>(a, total, first, stride); break;
  }
}

// ---- the grouped launch, tiled ---------------------------------------------------------------------------------------------------
// pack_body's threads each walk their own strided source runs: per step a model's parameters are read as millions of 4-byte loads
// (RCAN's 829 layouts: 0.38 ms per step, ~12x the time of moving the bytes).  Here a block owns one TILE of one table entry -- 16
// output channels x 64 input channels x all taps of the OIHW parameter: 16 contiguous runs -- reads it with coalesced loads into
// LDS and writes the 16-byte chunks of the packed layout (forward: the tile's 16 rows x 8 chunks per tap; dgrad: its 64 rows x 2
// chunks per tap) from there.  Blocks are assigned through a prefix table of tiles per entry (no block without work; the first
// attempt at LDS staging gave every block of a (tiles x entries) grid a whole CU's LDS and lost).  Entries the tile form does not
// cover (more than 9 taps, fp32, a pixel-shuffle permutation on a dgrad layout) get 16 blocks of pack_body.
constexpr int PT_CO = 16, PT_CI = 64, PT_PITCH = PT_CI * 9 + 1;
#ifndef SRK_PACK_TILED_MAX
#define SRK_PACK_TILED_MAX (1024LL * 1024)
#endif

// (round 3 kept layers beyond 128 x 128 channels on pack_body -- measured then, the tile form LOST on them (EDSR-large 1,376 -> 1,353 patches/s).
// What lost was the tile form's gather loop, one load in flight per thread; with nine (pack_tile) the tile form wins there too: EDSR-large's
// launch 214 -> 186 us, RDN's 112 -> 106 us on one box (tools/microbench_packgroup.py), so every 16-bit layout of at most 9 taps is tiled now)
__host__ __device__ inline bool pack_tiled_ok(const srk_pack_args& a) {
  return a.KH * a.KW <= 9 && a.dtype != SRK_F32 && !(a.dgrad && a.ps_r > 1) && (long long)a.Cout * a.Cin <= SRK_PACK_TILED_MAX;
}
// workgroups of pack_body for an entry the tile form does not take: one (row, 8-channel chunk) item of a 3x3 layout -- 72 strided loads --
// per thread (EDSR-large's 256 -> 256 layers: 8,192 items; with 16 workgroups per entry every thread walked two of them, one behind the
// other: the launch moved its 344 MB at 1.2 TB/s), 16 otherwise
__host__ __device__ inline int pack_body_blocks(const srk_pack_args& a) {
  if (a.KH * a.KW > 9 || a.dtype == SRK_F32) return 16;
  const long long items = (long long)a.CoutP * (a.KinP / 8);
  const long long b = (items + 255) / 256;
  return (int)(b < 16 ? 16 : (b > 64 ? 64 : b));
}
__host__ __device__ inline int pack_tiles_of(const srk_pack_args& a) {
  return pack_tiled_ok(a) ? ((a.Cout + PT_CO - 1) / PT_CO) * ((a.Cin + PT_CI - 1) / PT_CI) : pack_body_blocks(a);
}
// MFMA row of stored channel `chan` inside its blk-row block: the inverse of row_to_chan (srk_common.h)
__device__ __forceinline__ int chan_to_row(int chan, int blk) {
  if (blk == 64) return ((chan >> 4) & 1) * 32 + ((chan >> 2) & 3) * 8 + ((chan >> 5) & 1) * 4 + (chan & 3);
  return ((chan >> 2) & 3) * 8 + ((chan >> 4) & 1) * 4 + (chan & 3);
}

template <int DT> __device__ void pack_tile(const srk_pack_args& a, int t, float* tile) {
  typedef DTraits<DT> Tr;
  typename Tr::elem* out = reinterpret_cast<typename Tr::elem*>(a.wpk);
  const int taps = a.KH * a.KW;
  const int ncog = (a.Cout + PT_CO - 1) / PT_CO;
  const int cog = t % ncog, cig = t / ncog;
  const int co0 = cog * PT_CO, ci0 = cig * PT_CI;
  const int nco = min(PT_CO, a.Cout - co0), nci = min(PT_CI, a.Cin - ci0);
  const int run = nci * taps;                                    // contiguous floats per output channel
  // one wave per output channel at a time, NINE loads in flight per lane before the first LDS store (round 5: the loop this replaces had a
  // division per element and, with its run-time trip count, one load in flight per thread -- 36 memory latencies in a row per tile)
  for (int r16 = threadIdx.x >> 6; r16 < nco; r16 += 4) {
    const float* const src = a.w + ((size_t)(co0 + r16) * a.Cin + ci0) * taps;
    const int j0 = threadIdx.x & 63;
    float tmp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) tmp[k] = j0 + 64 * k < run ? src[j0 + 64 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (j0 + 64 * k < run) tile[r16 * PT_PITCH + j0 + 64 * k] = tmp[k];
  }
  __syncthreads();
  const int nch = a.KinP / 8;
  const int blk = (a.CoutP % 64 == 0) ? 64 : 32;
  if (!a.dgrad) {
    // item (r16, cc): source channel co0 + r16 -> packed row; 8 input channels ci0 + 8 cc ..
    for (int idx = threadIdx.x; idx < 128 * taps; idx += blockDim.x) {
      const int item = idx & 127, tap = idx >> 7;
      const int r16 = item & 15, cc = item >> 4;
      if (r16 >= nco || ci0 + cc * 8 >= a.KinP) continue;
      const int co = co0 + r16;
      int chan = co;
      if (a.ps_r > 1) {                                          // torch co = c*r*r + ij  ->  packed channel ij*Cc + c
        const int r2 = a.ps_r * a.ps_r;
        chan = (co % r2) * (a.Cout / r2) + co / r2;
      }
      const int row = (chan / blk) * blk + chan_to_row(chan % blk, blk);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = cc * 8 + e < nci ? tile[r16 * PT_PITCH + (cc * 8 + e) * taps + tap] : 0.f;
      i32x4 q;
      q.x = (int)((uint32_t)Tr::from_f32(v[0]) | ((uint32_t)Tr::from_f32(v[1]) << 16));
      q.y = (int)((uint32_t)Tr::from_f32(v[2]) | ((uint32_t)Tr::from_f32(v[3]) << 16));
      q.z = (int)((uint32_t)Tr::from_f32(v[4]) | ((uint32_t)Tr::from_f32(v[5]) << 16));
      q.w = (int)((uint32_t)Tr::from_f32(v[6]) | ((uint32_t)Tr::from_f32(v[7]) << 16));
      *reinterpret_cast<i32x4*>(out + (((size_t)tap * nch + (ci0 >> 3) + cc) * a.CoutP + row) * 8) = q;
    }
  } else {
    // item (ci_l, ccl): input channel ci0 + ci_l -> packed row; 8 output channels co0 + 8 ccl .. (dy's storage order), taps flipped
    for (int idx = threadIdx.x; idx < 128 * taps; idx += blockDim.x) {
      const int item = idx & 127, tap = idx >> 7;
      const int ci_l = item & 63, ccl = item >> 6;
      if (ci_l >= nci || co0 + ccl * 8 >= a.KinP) continue;
      const int chan = ci0 + ci_l;
      const int row = (chan / blk) * blk + chan_to_row(chan % blk, blk);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ccl * 8 + e < nco ? tile[(ccl * 8 + e) * PT_PITCH + ci_l * taps + (taps - 1 - tap)] : 0.f;
      i32x4 q;
      q.x = (int)((uint32_t)Tr::from_f32(v[0]) | ((uint32_t)Tr::from_f32(v[1]) << 16));
      q.y = (int)((uint32_t)Tr::from_f32(v[2]) | ((uint32_t)Tr::from_f32(v[3]) << 16));
      q.z = (int)((uint32_t)Tr::from_f32(v[4]) | ((uint32_t)Tr::from_f32(v[5]) << 16));
      q.w = (int)((uint32_t)Tr::from_f32(v[6]) | ((uint32_t)Tr::from_f32(v[7]) << 16));
      *reinterpret_cast<i32x4*>(out + (((size_t)tap * nch + (co0 >> 3) + ccl) * a.CoutP + row) * 8) = q;
    }
  }
  if (t == 0 && a.bias_pk && !a.dgrad) {
    for (int i = threadIdx.x; i < a.CoutP; i += blockDim.x) {
      const int chan = (i / blk) * blk + row_to_chan(i % blk, blk);       // bias_pk is indexed by MFMA row
      float b = 0.f;
      if (a.bias && chan < a.Cout) b = a.bias[ps_unperm(chan, a.Cout, a.ps_r)];
      a.bias_pk[i] = b;
    }
  }
}

__global__ __launch_bounds__(256) void pack_group_tiled_kernel(const srk_pack_args* __restrict__ table, const int* __restrict__ tile_begin, int n) {
  extern __shared__ float pack_tile_lds[];
  int lo = 0, hi = n;                                            // the entry whose tile range holds this block
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const srk_pack_args a = table[lo];
  const int t = (int)blockIdx.x - tile_begin[lo];
  if (pack_tiled_ok(a)) {
    if (a.dtype == SRK_BF16) pack_tile<SRK_BF16>(a, t, pack_tile_lds); else pack_tile<SRK_F16>(a, t, pack_tile_lds);
    return;
  }
  const long long total = (long long)a.KH * a.KW * a.KinP * a.CoutP;
  const long long first = (long long)t * blockDim.x + threadIdx.x, stride = (long long)pack_body_blocks(a) * blockDim.x;
  switch (a.dtype) {
    case SRK_BF16: pack_body<SRK_BF16>(a, total, first, stride); break;
    case SRK_F16: pack_body<SRK_F16>(a, total, first, stride); break;
    default: pack_body<SRK_F32>(a, total, first, stride); break;
  }
}

// Sums the per-workgroup slabs and converts [tap][ci][co'] -> OIHW.  Block = 128 consecutive slab elements x 8 waves;
// wave w sums slabs w, w+8, ... with float2 loads (512 contiguous bytes per wave instruction, 8 loads in flight),
// the 8 partial sums meet in LDS in a fixed order (bitwise reproducible).
__device__ __forceinline__ void wgrad_finalize_body(const srk_wgrad_fin_args& a, float (&red)[8][128], const int bx, const int gx) {
  const int taps = a.KH * a.KW;
  const int ns = a.nslabs > 1 ? a.nslabs : 1;
  const size_t per = (size_t)taps * a.CinP * a.CoutP;       // multiple of 256 (CinP, CoutP multiples of 16)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long nchunks = ((long long)per + 127) / 128;
  for (long long ch = bx; ch < nchunks; ch += gx) {
    const size_t e0 = (size_t)ch * 128 + 2 * lane;
    float s0 = 0.f, s1 = 0.f;
    if (e0 < per) {
      const float* p = a.dwp + e0;
#pragma unroll 8
      for (int sl = wv; sl < ns; sl += 8) {
        const float2 v = *reinterpret_cast<const float2*>(p + (size_t)sl * per);
        s0 += v.x;
        s1 += v.y;
      }
    }
    red[wv][2 * lane] = s0;
    red[wv][2 * lane + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 128) {
      const size_t e = (size_t)ch * 128 + threadIdx.x;
      if (e < per) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
        const int cop = (int)(e % a.CoutP);
        const size_t r2 = e / a.CoutP;
        const int ci = (int)(r2 % a.CinP), tap = (int)(r2 / a.CinP);
        if (cop < a.Cout && ci < a.Cin) {
          int co = cop;
          if (a.ps_r > 1) {          // packed co' = ij*Cc + c  ->  torch co = c*r*r + ij
            const int r2p = a.ps_r * a.ps_r, Cc = a.Cout / r2p;
            const int ij = cop / Cc, c = cop - ij * Cc;
            co = c * r2p + ij;
          }
          const size_t o = ((size_t)co * a.Cin + ci) * taps + tap;
          const float v = a.scale * t;
          if (a.accumulate) a.dw[o] += v; else a.dw[o] = v;
        }
      }
    }
    __syncthreads();
  }
  if (a.db && a.dbp && bx == 0) {
    // bias: 64 channels per pass, wave w sums slabs w, w+8, ... (loads in flight), fixed-order LDS reduce
    for (int c0 = 0; c0 < a.CoutP; c0 += 64) {
      const int cop = c0 + lane;
      float t = 0.f;
      if (cop < a.CoutP) {
#pragma unroll 8
        for (int sl = wv; sl < ns; sl += 8) t += a.dbp[(size_t)sl * a.CoutP + cop];
      }
      red[wv][lane] = t;
      __syncthreads();
      if (threadIdx.x < 64 && cop < a.Cout) {
        float u = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) u += red[g][lane];
        int co = cop;
        if (a.ps_r > 1) {
          const int r2p = a.ps_r * a.ps_r, Cc = a.Cout / r2p;
          const int ij = cop / Cc, c = cop - ij * Cc;
          co = c * r2p + ij;
        }
        const float v = a.scale * u;
        if (a.accumulate) a.db[co] += v; else a.db[co] = v;
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(512) void wgrad_finalize_kernel(const srk_wgrad_fin_args a) {
  __shared__ float red[8][128];
  wgrad_finalize_body(a, red, blockIdx.x, gridDim.x);
}

// ---- 3x3 finalize through an LDS transpose (the grouped launch) ----------------------------------------------------------------
// The slab layout is [tap][ci][co'] (co' fastest), OIHW wants [co][ci][tap]: element by element (wgrad_finalize_body)
// every 4-byte store lands in its own 64-byte segment (stride 9*Cin floats).  Here a workgroup owns 2 input channels x
// 64 output channels of one job: it reads the 18 slab rows (tap, ci) as float4 (16 lanes = one 256-byte row; the slab
// loop unrolled 8 deep, slabs added in order: bitwise reproducible) into an LDS tile and writes, per output channel, the
// 18 CONTIGUOUS floats [ci0, ci0+1][9 taps] (two lanes x 36 bytes).
__device__ __forceinline__ void wgrad_finalize3x3_tile(const srk_wgrad_fin_args& a, float (&tile)[18][68], const int ci0, const int co0) {
  const int ns = a.nslabs > 1 ? a.nslabs : 1;
  const size_t per = (size_t)9 * a.CinP * a.CoutP;
  const int tid = threadIdx.x;          // 256 threads: 16 row slots x 16 float4 columns
  const int c4 = tid & 15, rr = tid >> 4;
  for (int r = rr; r < 18; r += 16) {   // row r = tap * 2 + (ci - ci0)
    const int tap = r >> 1, ci = ci0 + (r & 1);
    float4 t = {0.f, 0.f, 0.f, 0.f};
    if (ci < a.CinP && co0 + 4 * c4 < a.CoutP) {
      const float* p = a.dwp + ((size_t)tap * a.CinP + ci) * a.CoutP + co0 + 4 * c4;
#pragma unroll 8
      for (int sl = 0; sl < ns; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sl * per);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
    }
    tile[r][4 * c4 + 0] = t.x; tile[r][4 * c4 + 1] = t.y; tile[r][4 * c4 + 2] = t.z; tile[r][4 * c4 + 3] = t.w;
  }
  __syncthreads();
  if (tid < 128) {                      // thread (co, cil): the 9 taps of one (co, ci)
    const int col = tid >> 1, cil = tid & 1;
    const int cop = co0 + col, ci = ci0 + cil;
    if (cop < a.Cout && ci < a.Cin) {
      int co = cop;
      if (a.ps_r > 1) {          // packed co' = ij*Cc + c  ->  torch co = c*r*r + ij
        const int r2p = a.ps_r * a.ps_r, Cc = a.Cout / r2p;
        const int ij = cop / Cc, c = cop - ij * Cc;
        co = c * r2p + ij;
      }
      float* o = a.dw + ((size_t)co * a.Cin + ci) * 9;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float v = a.scale * tile[tap * 2 + cil][col];
        if (a.accumulate) o[tap] += v; else o[tap] = v;
      }
    }
  }
  __syncthreads();
}

// one launch for every weight gradient of a step: blockIdx.y = table entry (the pack_group_kernel pattern)
__global__ __launch_bounds__(256) void wgrad_finalize_group_kernel(const srk_wgrad_fin_args* __restrict__ table) {
  __shared__ float tile[18][68];
  const srk_wgrad_fin_args a = table[blockIdx.y];
  const int ncob = (a.CoutP + 63) / 64, ncib = (a.CinP + 1) / 2;
  for (int t = blockIdx.x; t < ncob * ncib; t += gridDim.x) {
    const int cob = t % ncob, cib = t / ncob;
    wgrad_finalize3x3_tile(a, tile, cib * 2, cob * 64);
  }
  if (a.db && a.dbp && blockIdx.x == 0) {
    // bias: slabs summed in order per channel (a few hundred floats)
    const int ns = a.nslabs > 1 ? a.nslabs : 1;
    for (int cop = threadIdx.x; cop < a.Cout; cop += blockDim.x) {
      float u = 0.f;
#pragma unroll 8
      for (int sl = 0; sl < ns; ++sl) u += a.dbp[(size_t)sl * a.CoutP + cop];
      int co = cop;
      if (a.ps_r > 1) {
        const int r2p = a.ps_r * a.ps_r, Cc = a.Cout / r2p;
        const int ij = cop / Cc, c = cop - ij * Cc;
        co = c * r2p + ij;
      }
      const float v = a.scale * u;
      if (a.accumulate) a.db[co] += v; else a.db[co] = v;
    }
  }
}

// NCHW fp32 -> NHWC dtype; one thread per (pixel, 4-channel group)
template <int DT> __global__ void to_nhwc_kernel(const srk_to_nhwc_args a) {
  typedef DTraits<DT> Tr;
  typename Tr::elem* dst = reinterpret_cast<typename Tr::elem*>(a.dst);
  const int r = a.ps_r > 1 ? a.ps_r : 1, r2 = r * r;
  const int groups = a.Cstore / 4;
  const long long total = (long long)a.N * a.H * a.W * groups;
  const int Cs = a.C / r2;   // channels of the (shuffled) source
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    long long p = idx / groups;
    const int x = (int)(p % a.W);
    p /= a.W;
    const int y = (int)(p % a.H);
    const int n = (int)(p / a.H);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * 4 + e;
      float s = 0.f;
      if (c < a.C) {
        const int cs = c / r2, ij = c - cs * r2;
        const int si = ij / r, sj = ij - si * r;
        s = a.scale * a.src[((size_t)(n * Cs + cs) * (a.H * r) + y * r + si) * (a.W * r) + x * r + sj];
      }
      v[e] = s;
    }
    store4<DT>(dst + ((size_t)(n * a.H + y) * a.W + x) * a.dst_pitch + a.dst_coff + g * 4, v);
  }
}

// Cstore == 16 (the 3-channel image gradient): one thread per pixel, plane reads coalesced across x, ONE 32-byte
// (16-bit) / 64-byte (fp32) contiguous store per pixel
template <int DT> __global__ void to_nhwc16_kernel(const srk_to_nhwc_args a) {
  typedef DTraits<DT> Tr;
  typename Tr::elem* dst = reinterpret_cast<typename Tr::elem*>(a.dst);
  const int r = a.ps_r > 1 ? a.ps_r : 1, r2 = r * r;
  const int Cs = a.C / r2;
  const long long total = (long long)a.N * a.H * a.W;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(p % a.W);
    long long q = p / a.W;
    const int y = (int)(q % a.H);
    const int n = (int)(q / a.H);
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float s = 0.f;
      if (c < a.C) {
        const int cs = c / r2, ij = c - cs * r2;
        const int si = ij / r, sj = ij - si * r;
        s = a.scale * a.src[((size_t)(n * Cs + cs) * (a.H * r) + y * r + si) * (a.W * r) + x * r + sj];
      }
      v[c] = s;
    }
    typename Tr::elem* o = dst + (size_t)p * a.dst_pitch + a.dst_coff;
    if (Tr::IS16 && (a.dst_pitch & 7) == 0 && (a.dst_coff & 7) == 0) {
      // the pixel's 32 bytes as two 16-byte stores (was four 8-byte stores: the 302 MB gradient image of a 256 x 192 x 192
      // batch is written once per training step)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        i32x4 raw;
        raw.x = (int)pack2<DT>(v[8 * g + 0], v[8 * g + 1]); raw.y = (int)pack2<DT>(v[8 * g + 2], v[8 * g + 3]);
        raw.z = (int)pack2<DT>(v[8 * g + 4], v[8 * g + 5]); raw.w = (int)pack2<DT>(v[8 * g + 6], v[8 * g + 7]);
        *reinterpret_cast<i32x4*>(o + 8 * g) = raw;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) store4<DT>(o + 4 * g, v + 4 * g);
    }
  }
}

template <int DT> __global__ void to_nchw_kernel(const srk_to_nchw_args a) {
  typedef DTraits<DT> Tr;
  const typename Tr::elem* src = reinterpret_cast<const typename Tr::elem*>(a.src);
  const long long total = (long long)a.N * a.C * a.H * a.W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % a.W);
    long long t = idx / a.W;
    const int y = (int)(t % a.H);
    t /= a.H;
    const int c = (int)(t % a.C);
    const int n = (int)(t / a.C);
    a.dst[idx] = Tr::to_f32(src[((size_t)(n * a.H + y) * a.W + x) * a.src_pitch + a.src_coff + c]);
  }
}

// boundary im2col: NCHW fp32 (- sub) -> NHWC dtype with K = Cin*KH*KW channels; one thread per (pixel, 4 k's)
template <int DT> __global__ void unfold_kernel(const srk_unfold_args a) {
  typedef DTraits<DT> Tr;
  typename Tr::elem* dst = reinterpret_cast<typename Tr::elem*>(a.dst);
  const int groups = a.Kstore / 4, taps = a.KH * a.KW, K = a.Cin * taps;
  const int ph = a.KH / 2, pw = a.KW / 2;
  const long long total = (long long)a.N * a.H * a.W * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % groups);
    long long p = idx / groups;
    const int x = (int)(p % a.W);
    p /= a.W;
    const int y = (int)(p % a.H);
    const int n = (int)(p / a.H);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = g * 4 + e;
      float s = 0.f;
      if (k < K) {
        const int ci = k / taps, t = k - ci * taps;
        const int kh = t / a.KW, kw = t - kh * a.KW;
        const int yy = y + kh - ph, xx = x + kw - pw;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          s = a.x[((size_t)(n * a.Cin + ci) * a.H + yy) * a.W + xx];
          if (a.sub) s -= a.sub[ci];
        }
      }
      v[e] = s;
    }
    store4<DT>(dst + ((size_t)(n * a.H + y) * a.W + x) * a.dst_pitch + a.dst_coff + g * 4, v);
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 2048) g = 2048;   // cap + grid-stride (guide G11)
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int srk_pack_conv_weights(const srk_pack_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->w && a->wpk, "srk_pack_conv_weights: null pointer");
  SRK_CHECK_ARG(a->KinP % 16 == 0 && a->CoutP % 32 == 0, "srk_pack_conv_weights: KinP=%d CoutP=%d", a->KinP, a->CoutP);
  SRK_CHECK_ARG(a->dgrad ? (a->KinP >= a->Cout && a->CoutP >= a->Cin) : (a->KinP >= a->Cin && a->CoutP >= a->Cout),
                "srk_pack_conv_weights: padded sizes too small");
  if (a->ps_r > 1) SRK_CHECK_ARG(a->Cout % (a->ps_r * a->ps_r) == 0, "srk_pack_conv_weights: Cout %% r^2");
  const long long total = (long long)a->KH * a->KW * a->KinP * a->CoutP;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(total, 256);
  switch (a->dtype) {
    case SRK_BF16: hipLaunchKernelGGL(pack_kernel<SRK_BF16>, dim3(grid), dim3(256), 0, st, *a, total); break;
    case SRK_F16: hipLaunchKernelGGL(pack_kernel<SRK_F16>, dim3(grid), dim3(256), 0, st, *a, total); break;
    case SRK_F32: hipLaunchKernelGGL(pack_kernel<SRK_F32>, dim3(grid), dim3(256), 0, st, *a, total); break;
    default: SRK_CHECK_ARG(false, "srk_pack_conv_weights: dtype %d", a->dtype);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_pack_group_tiles(const srk_pack_args* host_table, int n, int* tile_begin) {
  SRK_CHECK_ARG(host_table && tile_begin && n > 0, "srk_pack_group_tiles: bad table / n=%d", n);
  long long tot = 0;
  for (int i = 0; i < n; ++i) {
    tile_begin[i] = (int)tot;
    tot += pack_tiles_of(host_table[i]);
    SRK_CHECK_ARG(tot < 0x7fffffffLL, "srk_pack_group_tiles: %lld tiles", tot);
  }
  tile_begin[n] = (int)tot;
  return 0;
}

extern "C" int srk_pack_conv_weights_group_tiled(const srk_pack_args* table, const int* tile_begin, int n, int total_tiles, srk_stream_t stream) {
  SRK_CHECK_ARG(table && tile_begin && n > 0 && total_tiles > 0, "srk_pack_conv_weights_group_tiled: bad table / n=%d tiles=%d", n, total_tiles);
  hipLaunchKernelGGL(pack_group_tiled_kernel, dim3((unsigned)total_tiles), dim3(256), PT_CO * PT_PITCH * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), table, tile_begin, n);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_pack_conv_weights_group(const srk_pack_args* table, int n, srk_stream_t stream) {
  SRK_CHECK_ARG(table && n > 0 && n <= 65535, "srk_pack_conv_weights_group: bad table / n=%d", n);
  hipLaunchKernelGGL(pack_group_kernel, dim3(16, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_wgrad_finalize(const srk_wgrad_fin_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->dwp && a->dw, "srk_wgrad_finalize: null pointer");
  SRK_CHECK_ARG(a->CinP >= a->Cin && a->CoutP >= a->Cout, "srk_wgrad_finalize: padded sizes");
  const long long rows = ((long long)a->KH * a->KW * a->CinP * a->CoutP + 127) / 128;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((unsigned)(rows > 2048 ? 2048 : (rows < 1 ? 1 : rows))), dim3(512), 0,
                     reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

// ---- small host table -> device memory through kernel arguments ------------------------------------------------------------
// The grouped launches read per-job descriptors from a device table whose contents (tensor addresses) change every step
// in eager mode.  A hipMemcpyAsync from pageable host memory cannot be captured into a hipGraph and a pinned staging
// buffer would have to outlive every graph that references it; kernel arguments are copied at launch (or capture)
// time, so the table travels as <= 3.5 KB by-value chunks and a replayed graph rewrites the same bytes.
struct UploadChunk { unsigned char b[3584]; };
__global__ void upload_kernel(unsigned char* __restrict__ dst, const UploadChunk c, int n) {
  // 16 bytes per thread where possible (dst is 16-byte aligned per chunk: chunk size is a multiple of 16)
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i + 16 <= n) {
    *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(c.b + i);
  } else {
    for (int k = i; k < n; ++k) dst[k] = c.b[k];
  }
}

extern "C" int srk_upload_small(void* dst_dev, const void* src_host, long long nbytes, srk_stream_t stream) {
  SRK_CHECK_ARG(dst_dev && src_host && nbytes >= 0, "srk_upload_small: null pointer");
  SRK_CHECK_ARG(((uintptr_t)dst_dev & 15) == 0, "srk_upload_small: destination must be 16-byte aligned");
  SRK_CHECK_ARG(nbytes <= (4 << 20), "srk_upload_small: %lld bytes is not a small table", nbytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (long long off = 0; off < nbytes; off += (long long)sizeof(UploadChunk)) {
    UploadChunk c;
    const int n = (int)((nbytes - off) < (long long)sizeof(UploadChunk) ? (nbytes - off) : (long long)sizeof(UploadChunk));
    memcpy(c.b, reinterpret_cast<const unsigned char*>(src_host) + off, (size_t)n);
    hipLaunchKernelGGL(upload_kernel, dim3((n + 16 * 256 - 1) / (16 * 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned char*>(dst_dev) + off, c, n);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

// ---- the same upload, ONCE, for a table that a hipGraph under capture will own (round 5) ---------------------------------------------
// Inside a graph the tables never change (every address a replay sees is the capture's), yet the chunks above are replayed with it: 3 launches
// per EDSR step, 30 per RCAN step (135 us of 8.7 ms at batch 16).  A capture site that keeps the table's memory alive as long as its graph
// (ops.static_tables) uploads through these instead: the chunks run NOW on a stream of the library's own -- not the capturing one, so they
// do not become graph nodes --, and srk_upload_fence(), called after the capture has ended and before the first replay, waits for them.
// One stream per DEVICE, created under a mutex (the forward thread and autograd's backward thread can both reach a capture site; a process
// may capture on several devices): the only process-wide state of the library, and it is write-once per device.
namespace {
constexpr int kMaxDev = 64;
std::mutex g_upload_mu;
hipStream_t g_upload_stream[kMaxDev] = {};
int upload_stream_of_current_device(hipStream_t* out, bool create) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDev) { srk_set_error("srk_upload: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev); return e != hipSuccess ? (int)e : SRK_E_BADARG; }
  std::lock_guard<std::mutex> lock(g_upload_mu);
  if (!g_upload_stream[dev] && create) {
    e = hipStreamCreateWithFlags(&g_upload_stream[dev], hipStreamNonBlocking);
    if (e != hipSuccess) { g_upload_stream[dev] = nullptr; srk_set_error("srk_upload_prepare: %s", hipGetErrorString(e)); return (int)e; }
  }
  *out = g_upload_stream[dev];
  return 0;
}
}  // namespace
extern "C" int srk_upload_prepare(void) {            // outside any capture: creates the current device's stream
  hipStream_t st = nullptr;
  return upload_stream_of_current_device(&st, true);
}
extern "C" int srk_upload_eager(void* dst_dev, const void* src_host, long long nbytes) {
  hipStream_t st = nullptr;
  const int rc = upload_stream_of_current_device(&st, false);
  if (rc != 0) return rc;
  SRK_CHECK_ARG(st != nullptr, "srk_upload_eager: call srk_upload_prepare() on this device first (outside the capture)");
  return srk_upload_small(dst_dev, src_host, nbytes, reinterpret_cast<srk_stream_t>(st));
}
extern "C" int srk_upload_fence(void) {
  hipStream_t st = nullptr;
  const int rc = upload_stream_of_current_device(&st, false);
  if (rc != 0) return rc;
  if (!st) return 0;
  const hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { srk_set_error("srk_upload_fence: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int srk_wgrad_finalize_group(const srk_wgrad_fin_args* table_dev, int n, int blocks_per_job, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && n > 0 && n <= 65535, "srk_wgrad_finalize_group: bad table (%d entries)", n);
  const int bx = blocks_per_job < 1 ? 1 : (blocks_per_job > 2048 ? 2048 : blocks_per_job);
  hipLaunchKernelGGL(wgrad_finalize_group_kernel, dim3(bx, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_nchw_to_nhwc(const srk_to_nhwc_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->src && a->dst, "srk_nchw_to_nhwc: null pointer");
  const int r = a->ps_r > 1 ? a->ps_r : 1;
  SRK_CHECK_ARG(a->Cstore % 4 == 0 && a->Cstore >= a->C && a->C % (r * r) == 0 && a->dst_pitch % 4 == 0 && a->dst_coff % 4 == 0,
                "srk_nchw_to_nhwc: C=%d Cstore=%d r=%d", a->C, a->Cstore, r);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->Cstore == 16) {
    const long long px = (long long)a->N * a->H * a->W;
    const int g16 = (int)((px + 255) / 256 > 8192 ? 8192 : (px + 255) / 256);
    switch (a->dtype) {
      case SRK_BF16: hipLaunchKernelGGL(to_nhwc16_kernel<SRK_BF16>, dim3(g16), dim3(256), 0, st, *a); break;
      case SRK_F16: hipLaunchKernelGGL(to_nhwc16_kernel<SRK_F16>, dim3(g16), dim3(256), 0, st, *a); break;
      case SRK_F32: hipLaunchKernelGGL(to_nhwc16_kernel<SRK_F32>, dim3(g16), dim3(256), 0, st, *a); break;
      default: SRK_CHECK_ARG(false, "srk_nchw_to_nhwc: dtype %d", a->dtype);
    }
    SRK_LAUNCH_CHECK();
    return 0;
  }
  const long long total = (long long)a->N * a->H * a->W * (a->Cstore / 4);
  const int grid = grid_for(total, 256);
  switch (a->dtype) {
    case SRK_BF16: hipLaunchKernelGGL(to_nhwc_kernel<SRK_BF16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F16: hipLaunchKernelGGL(to_nhwc_kernel<SRK_F16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F32: hipLaunchKernelGGL(to_nhwc_kernel<SRK_F32>, dim3(grid), dim3(256), 0, st, *a); break;
    default: SRK_CHECK_ARG(false, "srk_nchw_to_nhwc: dtype %d", a->dtype);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_nhwc_to_nchw(const srk_to_nchw_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->src && a->dst, "srk_nhwc_to_nchw: null pointer");
  const long long total = (long long)a->N * a->C * a->H * a->W;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(total, 256);
  switch (a->dtype) {
    case SRK_BF16: hipLaunchKernelGGL(to_nchw_kernel<SRK_BF16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F16: hipLaunchKernelGGL(to_nchw_kernel<SRK_F16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F32: hipLaunchKernelGGL(to_nchw_kernel<SRK_F32>, dim3(grid), dim3(256), 0, st, *a); break;
    default: SRK_CHECK_ARG(false, "srk_nhwc_to_nchw: dtype %d", a->dtype);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

// The head conv's case -- 3 image channels, 3x3, K = 27 stored as 32 16-bit values (64 bytes) per pixel: ONE thread per pixel (the general
// kernel spends a thread and ~10 integer divisions by run-time values per 8 bytes: 52.7 us for the 45 MB of a 256 x 48 x 48 batch, 0.85 TB/s).
// k = ci * 9 + kh * 3 + kw as above; neighbouring lanes read neighbouring columns of the same rows; 4 x 16-byte stores per pixel.
template <int DT> __global__ __launch_bounds__(256) void unfold3x3c3_kernel(const srk_unfold_args a) {
  typedef DTraits<DT> Tr;
  static_assert(Tr::IS16, "16-bit storage");
  typename Tr::elem* dst = reinterpret_cast<typename Tr::elem*>(a.dst);
  const long long total = (long long)a.N * a.H * a.W;
  const float sub[3] = {a.sub ? a.sub[0] : 0.f, a.sub ? a.sub[1] : 0.f, a.sub ? a.sub[2] : 0.f};
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % a.W);
    const long long q = p / a.W;
    const int y = (int)(q % a.H);
    const int n = (int)(q / a.H);
    float v[32];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float* const plane = a.x + (size_t)(n * 3 + ci) * a.H * a.W;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        const bool rok = yy >= 0 && yy < a.H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int xx = x + kw - 1;
          const bool ok = rok && xx >= 0 && xx < a.W;
          v[ci * 9 + kh * 3 + kw] = ok ? plane[(size_t)yy * a.W + xx] - sub[ci] : 0.f;
        }
      }
    }
#pragma unroll
    for (int k = 27; k < 32; ++k) v[k] = 0.f;
    typename Tr::elem* const o = dst + (size_t)p * a.dst_pitch + a.dst_coff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      i32x4 raw;
      raw.x = (int)pack2<DT>(v[8 * g + 0], v[8 * g + 1]); raw.y = (int)pack2<DT>(v[8 * g + 2], v[8 * g + 3]);
      raw.z = (int)pack2<DT>(v[8 * g + 4], v[8 * g + 5]); raw.w = (int)pack2<DT>(v[8 * g + 6], v[8 * g + 7]);
      *reinterpret_cast<i32x4*>(o + 8 * g) = raw;
    }
  }
}

extern "C" int srk_unfold_nchw(const srk_unfold_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->dst, "srk_unfold_nchw: null pointer");
  SRK_CHECK_ARG(a->Kstore % 16 == 0 && a->Kstore >= a->Cin * a->KH * a->KW && a->dst_pitch % 4 == 0 && a->dst_coff % 4 == 0,
                "srk_unfold_nchw: Kstore=%d for Cin=%d %dx%d", a->Kstore, a->Cin, a->KH, a->KW);
  const long long total = (long long)a->N * a->H * a->W * (a->Kstore / 4);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // (from ~200k pixels on: with fewer, the general kernel's 8 threads per pixel fill the chip better -- 5.9 against 9.1 us at 16 x 48 x 48)
  static const bool force_px = srk_dbg_getenv("SRK_UNFOLD_PX") != nullptr;      // tests: the per-pixel kernel at any size
  if (a->Cin == 3 && a->KH == 3 && a->KW == 3 && a->Kstore == 32 && a->dtype != SRK_F32 && a->dst_pitch % 8 == 0 && a->dst_coff % 8 == 0 &&
      (reinterpret_cast<uintptr_t>(a->dst) & 15) == 0 && ((long long)a->N * a->H * a->W >= 200000 || force_px)) {
    const int gpx = grid_for((long long)a->N * a->H * a->W, 256);
    if (a->dtype == SRK_BF16) hipLaunchKernelGGL(unfold3x3c3_kernel<SRK_BF16>, dim3(gpx), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(unfold3x3c3_kernel<SRK_F16>, dim3(gpx), dim3(256), 0, st, *a);
    SRK_LAUNCH_CHECK();
    return 0;
  }
  const int grid = grid_for(total, 256);
  switch (a->dtype) {
    case SRK_BF16: hipLaunchKernelGGL(unfold_kernel<SRK_BF16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F16: hipLaunchKernelGGL(unfold_kernel<SRK_F16>, dim3(grid), dim3(256), 0, st, *a); break;
    case SRK_F32: hipLaunchKernelGGL(unfold_kernel<SRK_F32>, dim3(grid), dim3(256), 0, st, *a); break;
    default: SRK_CHECK_ARG(false, "srk_unfold_nchw: dtype %d", a->dtype);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}
