// Kernels for the remaining conv models of the zoo: SRResNet (models/srresnet.py:9-36 --
// 9x9 convs, BatchNorm2d and PReLU inside `ResBlock` / `BasicBlock`, models/common.py:33-56,74-109) and DDBPN
// (models/ddbpn.py:10-137 -- Conv2d / ConvTranspose2d projections of 6x6 / 8x8 / 12x12 with stride 2 / 4 / 8, PReLU).
//
//   srk_unfold_nhwc / srk_fold_nhwc   im2col / col2im on NHWC tensors, any kernel size, stride and padding.  A strided
//       or large-kernel convolution is  unfold -> 1x1 conv on MFMA (srk_conv2d);  a transposed convolution is
//       1x1 conv -> fold.  Each is the other's adjoint, so the data gradients use the same two kernels, and every FLOP
//       of these layers runs in the 1x1 implicit-GEMM / slab weight-gradient kernels the other models already use.
//       16-byte accesses; the fold GATHERS its <= ceil(K/s)^2 contributions per output pixel in fp32 (no atomics).
//   srk_chan_stats   per-channel partial sums over pixels: {sum x, sum x^2} (BatchNorm statistics), {sum y, sum x*y}
//       (BatchNorm backward), {sum over x<=0 of x*y} (PReLU slope gradient).  One plain store per block and channel,
//       partials added by the caller in block order: no atomics, bitwise reproducible.
//   srk_chan_apply   out = post( (a[c]*x + b[c]*y + d[c]) * gate(z) ): BatchNorm apply (+ residual), BatchNorm backward
//       (two inputs), PReLU forward (post) and backward (gate).  HBM-bound streaming, 16-byte accesses.
#include <stdlib.h>
#include "srk_common.h"

namespace {

constexpr int GEN_NT = 256;

template <int DT> SRK_DEV void chunk_to_f32(i32x4 raw, float* v) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    unpack2<DT>((uint32_t)raw.x, v[0], v[1]); unpack2<DT>((uint32_t)raw.y, v[2], v[3]);
    unpack2<DT>((uint32_t)raw.z, v[4], v[5]); unpack2<DT>((uint32_t)raw.w, v[6], v[7]);
  } else {
    v[0] = __int_as_float(raw.x); v[1] = __int_as_float(raw.y); v[2] = __int_as_float(raw.z); v[3] = __int_as_float(raw.w);
  }
}
template <int DT> SRK_DEV i32x4 f32_to_chunk(const float* v) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    return i32x4{(int)pack2<DT>(v[0], v[1]), (int)pack2<DT>(v[2], v[3]), (int)pack2<DT>(v[4], v[5]), (int)pack2<DT>(v[6], v[7])};
  } else {
    return i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
  }
}

// cols[n][oy][ox][(kh*K + kw)*C + c] = x[n][oy*s + kh - p][ox*s + kw - p][c]   (0 outside the image)
template <int DT> __global__ __launch_bounds__(GEN_NT) void unfold_nhwc_kernel(const srk_unfold_nhwc_args a) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  const int nch = a.C / CH, KK = a.K * a.K;
  const long long total = (long long)a.N * a.Ho * a.Wo * KK * nch;
  const elem* x = reinterpret_cast<const elem*>(a.x);
  elem* cols = reinterpret_cast<elem*>(a.cols);
  for (long long i = (long long)blockIdx.x * GEN_NT + threadIdx.x; i < total; i += (long long)gridDim.x * GEN_NT) {
    const int cc = (int)(i % nch);
    long long q = i / nch;
    const int tap = (int)(q % KK);
    q /= KK;
    const int ox = (int)(q % a.Wo);
    q /= a.Wo;
    const int oy = (int)(q % a.Ho);
    const int n = (int)(q / a.Ho);
    const int kh = tap / a.K, kw = tap - kh * a.K;
    const int iy = oy * a.stride + kh - a.pad, ix = ox * a.stride + kw - a.pad;
    i32x4 v = {0, 0, 0, 0};
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
      v = gload16(x + ((size_t)(n * a.H + iy) * a.W + ix) * a.x_pitch + a.x_coff + cc * CH);
    *reinterpret_cast<i32x4*>(cols + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.cols_pitch + (size_t)tap * a.C + cc * CH) = v;
  }
}

// out[n][y][x][c] = bias[c] + sum over taps (kh, kw) with (y + p - kh) % s == 0, (x + p - kw) % s == 0 and the source
// pixel ((y + p - kh) / s, (x + p - kw) / s) inside the Hi x Wi grid of cols[..][(kh*K + kw)*C + c]
template <int DT> __global__ __launch_bounds__(GEN_NT) void fold_nhwc_kernel(const srk_fold_nhwc_args a) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  const int nch = a.C / CH;
  const long long total = (long long)a.N * a.Ho * a.Wo * nch;
  const elem* cols = reinterpret_cast<const elem*>(a.cols);
  elem* out = reinterpret_cast<elem*>(a.out);
  for (long long i = (long long)blockIdx.x * GEN_NT + threadIdx.x; i < total; i += (long long)gridDim.x * GEN_NT) {
    const int cc = (int)(i % nch);
    long long q = i / nch;
    const int x = (int)(q % a.Wo);
    q /= a.Wo;
    const int y = (int)(q % a.Ho);
    const int n = (int)(q / a.Ho);
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = a.bias ? a.bias[cc * CH + e] : 0.f;
    // kh = (y + p) - s*iy  for iy in [ceil((y + p - K + 1)/s), floor((y + p)/s)]
    const int ty = y + a.pad, tx = x + a.pad;
    int iy0 = ty - a.K + 1;
    iy0 = iy0 > 0 ? (iy0 + a.stride - 1) / a.stride : 0;
    int ix0 = tx - a.K + 1;
    ix0 = ix0 > 0 ? (ix0 + a.stride - 1) / a.stride : 0;
    const int iy1 = min(ty / a.stride, a.Hi - 1), ix1 = min(tx / a.stride, a.Wi - 1);
    for (int iy = iy0; iy <= iy1; ++iy) {
      const int kh = ty - iy * a.stride;
      for (int ix = ix0; ix <= ix1; ++ix) {
        const int kw = tx - ix * a.stride;
        float v[CH];
        chunk_to_f32<DT>(gload16(cols + ((size_t)(n * a.Hi + iy) * a.Wi + ix) * a.cols_pitch + (size_t)(kh * a.K + kw) * a.C + cc * CH), v);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += v[e];
      }
    }
    *reinterpret_cast<i32x4*>(out + ((size_t)(n * a.Ho + y) * a.Wo + x) * a.out_pitch + a.out_coff + cc * CH) = f32_to_chunk<DT>(acc);
  }
}

// ---- per-channel partial sums ---------------------------------------------------------------------------------------
// grid = blocks over pixels; thread = (16-byte channel chunk, pixel row); partial[block][2][C]
SRK_DEV void chan_finalize_body(const srk_chan_finalize_args& a);

// CH consecutive per-channel fp32 constants as 16-byte loads (p + c0 is 16-byte aligned: c0 is a multiple of CH >= 4)
template <int CH> SRK_DEV void load_consts(const float* p, int c0, int stride, float (&o)[CH]) {
  if (stride == 0) {
    const float v = p[0];
#pragma unroll
    for (int e = 0; e < CH; ++e) o[e] = v;
  } else {
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + c0 + 4 * q);
      o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
    }
  }
}

// FUSED: the block that finishes LAST (an arrival counter in device memory, which it resets for the next launch) goes on to do
// srk_chan_finalize's work on the partials of all blocks: one launch per BatchNorm / PReLU reduction instead of two (the second
// was one workgroup of [C]-sized arithmetic behind a launch boundary: ~7 us each, 118 per SRResNet training step)
template <int DT, bool FUSED> __global__ __launch_bounds__(GEN_NT) void chan_stats_kernel(const srk_chan_stats_args a, long long pix_per_block,
                                                                                           const srk_chan_finalize_args f, int* counter) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  __shared__ float red[3][GEN_NT * 8];
  const int tid = threadIdx.x;
  const int nch = a.C / CH, rows = GEN_NT / nch;
  const int cc = tid % nch, prow = tid / nch;
  const long long p0 = (long long)blockIdx.x * pix_per_block, p1 = min(a.P, p0 + pix_per_block);
  float s0[CH], s1[CH], s2[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    s0[e] = s1[e] = s2[e] = 0.f;
    sh[e] = (a.shift && prow < rows) ? a.shift[cc * CH + e] : 0.f;      // x is centred first: sums of (x - shift[c])
  }
  if (a.shift_out && prow < rows) {                                     // ... by the tensor's first pixel (every block reads it)
    chunk_to_f32<DT>(gload16(reinterpret_cast<const elem*>(a.x) + a.x_coff + cc * CH), sh);
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
      for (int e = 0; e < CH; ++e) __hip_atomic_store(a.shift_out + cc * CH + e, sh[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (prow < rows) {
    const elem* x = reinterpret_cast<const elem*>(a.x) + a.x_coff + cc * CH;
    const elem* y = a.y ? reinterpret_cast<const elem*>(a.y) + a.y_coff + cc * CH : nullptr;
    float ga[CH], gd[CH], gs[CH];                                       // mode 3: the thread's channels' constants, loaded once
    if (a.mode == 3) {
      load_consts<CH>(a.gate_a, cc * CH, 1, ga);
      load_consts<CH>(a.gate_d, cc * CH, 1, gd);
      load_consts<CH>(a.slope, cc * CH, a.slope_stride, gs);
    } else if (a.gate_out) {
      load_consts<CH>(a.slope, cc * CH, a.slope_stride, gs);
    }
    // U loads in flight per thread (the blocks are few: see stats_blocks), consumed in pixel order: the sums' order is fixed
    constexpr int U = 8;
    for (long long pb = p0 + prow; pb < p1; pb += (long long)rows * U) {
      i32x4 xq[U], yq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long p = pb + (long long)u * rows;
        if (p < p1) {
          xq[u] = gload16(x + (size_t)p * a.x_pitch);
          if (a.mode != 0) yq[u] = gload16(y + (size_t)p * a.y_pitch);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long p = pb + (long long)u * rows;
        if (p < p1) {
          float xv[CH];
          chunk_to_f32<DT>(xq[u], xv);
          if (a.mode == 3) {
            // BatchNorm backward BEHIND a PReLU, in one pass over (x = the BatchNorm's input, y = the gradient of the PReLU's
            // output): the BatchNorm output yb = gate_a x + gate_d is recomputed, gp = y * (yb > 0 ? 1 : slope) is the gradient the
            // BatchNorm sees: s0 = sum gp, s1 = sum (x - shift) gp, s2 = sum over yb <= 0 of yb * y (the slope's gradient)
            float yv[CH];
            chunk_to_f32<DT>(yq[u], yv);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
              const float yb = ga[e] * xv[e] + gd[e];
              const float gp = yv[e] * (yb > 0.f ? 1.f : gs[e]);
              s0[e] += gp;
              s1[e] += (xv[e] - sh[e]) * gp;
              s2[e] += yb <= 0.f ? yb * yv[e] : 0.f;
            }
            continue;
          }
#pragma unroll
          for (int e = 0; e < CH; ++e) xv[e] -= sh[e];
          if (a.mode == 0) {
#pragma unroll
            for (int e = 0; e < CH; ++e) { s0[e] += xv[e]; s1[e] += xv[e] * xv[e]; }
          } else {
            float yv[CH];
            chunk_to_f32<DT>(yq[u], yv);
            if (a.mode == 1) {
#pragma unroll
              for (int e = 0; e < CH; ++e) { s0[e] += yv[e]; s1[e] += xv[e] * yv[e]; }
            } else {
#pragma unroll
              for (int e = 0; e < CH; ++e) s0[e] += xv[e] <= 0.f ? xv[e] * yv[e] : 0.f;
              if (a.gate_out) {      // nn.PReLU's input gradient from the same read: y * (x > 0 ? 1 : slope)
                float gv[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) gv[e] = yv[e] * (xv[e] > 0.f ? 1.f : gs[e]);
                *reinterpret_cast<i32x4*>(reinterpret_cast<elem*>(a.gate_out) + (size_t)p * a.gate_pitch + cc * CH) = f32_to_chunk<DT>(gv);
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    red[0][tid * CH + e] = (prow < rows) ? s0[e] : 0.f;
    red[1][tid * CH + e] = (prow < rows) ? s1[e] : 0.f;
    if (a.mode == 3) red[2][tid * CH + e] = (prow < rows) ? s2[e] : 0.f;
  }
  __syncthreads();
  if (tid < a.C) {
    const int c_cc = tid / CH, c_e = tid % CH;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rows; ++r) {
      t0 += red[0][(r * nch + c_cc) * CH + c_e];
      t1 += red[1][(r * nch + c_cc) * CH + c_e];
      if (a.mode == 3) t2 += red[2][(r * nch + c_cc) * CH + c_e];
    }
    float* const p0 = a.partial + ((size_t)blockIdx.x * 2 + 0) * a.C + tid;
    float* const p1 = a.partial + ((size_t)blockIdx.x * 2 + 1) * a.C + tid;
    float* const p2 = a.mode == 3 ? a.partial2 + (size_t)blockIdx.x * a.C + tid : nullptr;
    if constexpr (FUSED) {       // device-scope stores (written through to where every XCD sees them): the count below then needs no
      __hip_atomic_store(p0, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // cache write-back -- a release per block cost more
      __hip_atomic_store(p1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // than the launch it saves
      if (p2) __hip_atomic_store(p2, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *p0 = t0;
      *p1 = t1;
      if (p2) *p2 = t2;
    }
  }
  if constexpr (FUSED) {
    __shared__ int last;
    // every storing thread waits for ITS OWN write-through stores to be acknowledged before the barrier (the barrier orders at
    // workgroup scope only and the compiler may drop its vmcnt wait: without this the arrival count could become visible on another
    // XCD before a wave's partial sums -- rare, nondeterministically wrong statistics).  The stores are sc1 (agent-scope
    // atomic stores): acknowledged = visible to every XCD, so the relaxed count behind the barrier needs no cache write-back.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // the last block sees every other block's
    if (tid == 0) *counter = 0;
    chan_finalize_body(f);
  }
}

// ---- per-channel affine of one or two inputs, optional PReLU gate / PReLU output ----------------------------------------
template <int DT> __global__ __launch_bounds__(GEN_NT) void chan_apply_kernel(const srk_chan_apply_args a) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  const int nch = a.C / CH;
  const long long total = a.P * nch;
  const elem* x = reinterpret_cast<const elem*>(a.x);
  const elem* y = reinterpret_cast<const elem*>(a.y);
  const elem* z = reinterpret_cast<const elem*>(a.z);
  elem* out = reinterpret_cast<elem*>(a.out);
  for (long long i = (long long)blockIdx.x * GEN_NT + threadIdx.x; i < total; i += (long long)gridDim.x * GEN_NT) {
    const int cc = (int)(i % nch);
    const long long p = i / nch;
    float xv[CH], v[CH];
    chunk_to_f32<DT>(gload16(x + (size_t)p * a.x_pitch + a.x_coff + cc * CH), xv);
    // the chunk's per-channel constants as 16-byte loads (they were 8 dword loads each: a launch of this kernel is mostly latency)
    float aa[CH], dd[CH], gs[CH];
    if (a.a) load_consts<CH>(a.a, cc * CH, 1, aa);
    if (a.d) load_consts<CH>(a.d, cc * CH, 1, dd);
    if (a.slope) load_consts<CH>(a.slope, cc * CH, a.slope_stride, gs);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] = (a.a ? aa[e] : 1.f) * xv[e] + (a.d ? dd[e] : 0.f);
    if (y) {
      float yv[CH], bb[CH];
      chunk_to_f32<DT>(gload16(y + (size_t)p * a.y_pitch + a.y_coff + cc * CH), yv);
      if (a.b) load_consts<CH>(a.b, cc * CH, 1, bb);
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += (a.b ? bb[e] : 1.f) * yv[e];
    }
    if (z && a.gate_a) {   // BatchNorm backward behind a PReLU (srk_chan_stats mode 3): only the a x term passes the gate, whose
      float zv[CH];        // argument is the recomputed BatchNorm output gate_a z + gate_d:  out = a x gate + b y + d
      chunk_to_f32<DT>(gload16(z + (size_t)p * a.z_pitch + a.z_coff + cc * CH), zv);
      float ga[CH], gd[CH];
      load_consts<CH>(a.gate_a, cc * CH, 1, ga);
      load_consts<CH>(a.gate_d, cc * CH, 1, gd);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float ax = (a.a ? aa[e] : 1.f) * xv[e];
        const float gate = (ga[e] * zv[e] + gd[e]) > 0.f ? 1.f : gs[e];
        v[e] += ax * (gate - 1.f);
      }
    } else if (z) {          // gate: PReLU backward, d/dz prelu(z) = z > 0 ? 1 : slope
      float zv[CH];
      chunk_to_f32<DT>(gload16(z + (size_t)p * a.z_pitch + a.z_coff + cc * CH), zv);
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] *= zv[e] > 0.f ? 1.f : gs[e];
    }
    if (a.post_prelu) {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * gs[e];
    }
    *reinterpret_cast<i32x4*>(out + (size_t)p * a.out_pitch + a.out_coff + cc * CH) = f32_to_chunk<DT>(v);
  }
}

inline unsigned grid_for(long long total) {
  long long b = (total + GEN_NT - 1) / GEN_NT;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline int stats_blocks(long long P) {
  // >= 256 pixels per block = ONE round of the 8 loads a thread keeps in flight (512: two dependent rounds; measured on SRResNet at
  // batch 16, 67 reductions per step: 4,846 -> 4,950 patches/s; 1,024 pixels per block: 4,498), at most 1024 blocks
  static const int ppb = [] { const char* e = srk_dbg_getenv("SRK_STATS_PPB"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();   // A/B knob
  long long b = (P + ppb - 1) / ppb;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}


// The [C]-sized step between srk_chan_stats and srk_chan_apply of a BatchNorm2d / PReLU, as ONE small launch: the ordered sum of
// the per-block partials and the vector arithmetic that was ~15 tiny elementwise launches per layer and direction (970 per
// SRResNet training step at batch 16: 4.5 ms of 14.6).  One workgroup, one thread per channel.
SRK_DEV void chan_finalize_body(const srk_chan_finalize_args& a) {
  const int c = threadIdx.x, C = a.C;
  __shared__ float red[GEN_NT];
  __shared__ __attribute__((aligned(16))) float part0[4 * GEN_NT], part1[4 * GEN_NT];
  // sum over the blocks: thread (q, c4) adds rows q, q + Q, ... of four channels with 16-byte loads (hundreds of rows: one
  // thread per channel walking them all is a chain of dependent loads, 24 us); the Q strided sums meet in LDS in a fixed order
  {
    const int c4n = C >> 2, Q = GEN_NT / c4n;
    const int q = c / c4n, c4 = c - q * c4n;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
    if (q < Q) {
      const float* p = a.partial + 4 * c4;
#pragma unroll 8
      for (int b = q; b < a.nblocks; b += Q) {
        t0 = t0 + *reinterpret_cast<const f32x4*>(p + (size_t)b * 2 * C);
        t1 = t1 + *reinterpret_cast<const f32x4*>(p + (size_t)b * 2 * C + C);
      }
      *reinterpret_cast<f32x4*>(part0 + q * C + 4 * c4) = t0;
      *reinterpret_cast<f32x4*>(part1 + q * C + 4 * c4) = t1;
    }
  }
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const int Q = GEN_NT / (C >> 2);
    for (int q = 0; q < Q; ++q) { s0 += part0[q * C + c]; s1 += part1[q * C + c]; }
  }
  const float M = a.M;
  float* const o = a.out;
  const bool real = c < a.Creal;
  if (a.mode == 4) {                       // plain sums (PReLU slope gradient); total = 1: summed over the channels too
    if (a.total) {
      red[c] = c < C ? s0 : 0.f;
      __syncthreads();
      if (c == 0) {
        float t = 0.f;
        for (int k = 0; k < C; ++k) t += red[k];
        o[0] = t;
        if (a.dgamma_acc) a.dgamma_acc[0] += t;
      }
    } else if (c < C) {
      o[c] = s0;
      if (a.dgamma_acc && real) a.dgamma_acc[c] += s0;
    }
    return;
  }
  if (c >= C) return;
  if (a.mode == 0) {                       // batch mean
    o[c] = s0 / M;
  } else if (a.mode == 1) {                // variance from the centred sums, running buffers, scale / shift of the apply pass
    const float m1 = s0 / M;
    const float var = fmaxf(s1 / M - m1 * m1, 0.f);
    const float mean = a.mean[c] + m1;                     // K + E[x - K]  (K = the first pass's mean: m1 is its rounding residue)
    if (a.nbt && c == 0) *a.nbt += 1;
    if (a.running_mean && real) {
      a.running_mean[c] = a.running_mean[c] * (1.f - a.momentum) + mean * a.momentum;
      a.running_var[c] = a.running_var[c] * (1.f - a.momentum) + var * (M / fmaxf(M - 1.f, 1.f)) * a.momentum;
    }
    const float invstd = rsqrtf(var + a.eps);
    const float gamma = real ? a.weight[c] : 0.f, beta = real ? a.bias[c] : 0.f;
    const float sc = gamma * invstd;
    o[c] = invstd;
    o[C + c] = gamma;
    o[2 * C + c] = sc;
    o[3 * C + c] = beta - mean * sc;
    o[4 * C + c] = mean;
  } else {                                 // backward: s0 = sum dy, s1 = sum (x - mean) dy
    if (a.partial2) {                      // (+ the slope gradient of the PReLU behind the BatchNorm: srk_chan_stats mode 3; out row 5)
      float t = 0.f;
#pragma unroll 8
      for (int b = 0; b < a.nblocks; ++b) t += a.partial2[(size_t)b * C + c];
      if (a.total2) {                      // one shared slope: summed over the channels too (the threads c >= C left above)
        red[c] = t;
        __syncthreads();
        if (c == 0) {
          float tt = 0.f;
          for (int k = 0; k < C; ++k) tt += red[k];
          o[5 * C] = tt;
          if (a.dslope_acc) a.dslope_acc[0] += tt;
        }
      } else {
        o[5 * C + c] = t;
        if (a.dslope_acc && real) a.dslope_acc[c] += t;
      }
    }
    const float invstd = a.invstd[c], gamma = a.gamma[c];
    const float dbeta = s0, dgamma = invstd * s1;
    const float k = gamma * invstd;
    o[c] = dgamma;
    o[C + c] = dbeta;
    if (a.dgamma_acc && real) a.dgamma_acc[c] += dgamma;
    if (a.dbeta_acc && real) a.dbeta_acc[c] += dbeta;
    o[2 * C + c] = k;
    if (a.mode == 2) {                     // batch statistics: dx = k (dy - dbeta/M - xhat dgamma/M), xhat = (x - mean) invstd
      const float bx = -k * invstd * dgamma / M;
      o[3 * C + c] = bx;
      o[4 * C + c] = -k * dbeta / M - bx * a.mean[c];
    }
  }
}
__global__ __launch_bounds__(GEN_NT) void chan_finalize_kernel(const srk_chan_finalize_args a) { chan_finalize_body(a); }

}  // namespace

#define GEN_DISPATCH(KERNEL, DTYPE, GRID, ST, ...)                                                          \
  switch (DTYPE) {                                                                                          \
    case SRK_BF16: hipLaunchKernelGGL(KERNEL<SRK_BF16>, dim3(GRID), dim3(GEN_NT), 0, ST, __VA_ARGS__); break; \
    case SRK_F16: hipLaunchKernelGGL(KERNEL<SRK_F16>, dim3(GRID), dim3(GEN_NT), 0, ST, __VA_ARGS__); break;   \
    case SRK_F32: hipLaunchKernelGGL(KERNEL<SRK_F32>, dim3(GRID), dim3(GEN_NT), 0, ST, __VA_ARGS__); break;   \
    default: srk_set_error("dtype %d", (int)(DTYPE)); return SRK_E_BADARG;                                  \
  }

extern "C" int srk_unfold_nhwc(const srk_unfold_nhwc_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->cols, "srk_unfold_nhwc: null pointer");
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0 && a->K > 0 && a->stride > 0 && a->pad >= 0, "srk_unfold_nhwc: bad dims");
  SRK_CHECK_ARG(a->C > 0 && a->C % ch == 0 && a->x_pitch % ch == 0 && a->x_coff % ch == 0 && a->cols_pitch % ch == 0 &&
                    a->cols_pitch >= a->K * a->K * a->C, "srk_unfold_nhwc: channels / pitches must be 16-byte multiples");
  SRK_CHECK_ARG(a->Ho == (a->H + 2 * a->pad - a->K) / a->stride + 1 && a->Wo == (a->W + 2 * a->pad - a->K) / a->stride + 1,
                "srk_unfold_nhwc: Ho x Wo = %d x %d does not match the conv geometry", a->Ho, a->Wo);
  const long long total = (long long)a->N * a->Ho * a->Wo * a->K * a->K * (a->C / ch);
  GEN_DISPATCH(unfold_nhwc_kernel, a->dtype, grid_for(total), reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_fold_nhwc(const srk_fold_nhwc_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->cols && a->out, "srk_fold_nhwc: null pointer");
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->N > 0 && a->Hi > 0 && a->Wi > 0 && a->K > 0 && a->stride > 0 && a->pad >= 0 && a->Ho > 0 && a->Wo > 0, "srk_fold_nhwc: bad dims");
  SRK_CHECK_ARG(a->C > 0 && a->C % ch == 0 && a->out_pitch % ch == 0 && a->out_coff % ch == 0 && a->cols_pitch % ch == 0 &&
                    a->cols_pitch >= a->K * a->K * a->C, "srk_fold_nhwc: channels / pitches must be 16-byte multiples");
  const long long total = (long long)a->N * a->Ho * a->Wo * (a->C / ch);
  GEN_DISPATCH(fold_nhwc_kernel, a->dtype, grid_for(total), reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_chan_stats_blocks(long long P) { return stats_blocks(P); }

static int chan_stats_check(const srk_chan_stats_args* a) {
  SRK_CHECK_ARG(a && a->x && a->partial && (a->mode == 0 || a->y), "srk_chan_stats: null pointer");
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->mode >= 0 && a->mode <= 3, "srk_chan_stats: mode %d", a->mode);
  SRK_CHECK_ARG(a->mode != 3 || (a->gate_a && a->gate_d && a->slope && a->partial2 && !a->shift_out && !a->gate_out &&
                                 (((uintptr_t)a->gate_a | (uintptr_t)a->gate_d | (a->slope_stride ? (uintptr_t)a->slope : 0)) & 15) == 0),
                "srk_chan_stats: mode 3 needs 16-byte aligned gate_a, gate_d, slope and partial2");
  SRK_CHECK_ARG(!a->shift_out || !a->shift, "srk_chan_stats: shift and shift_out exclude each other");
  SRK_CHECK_ARG(a->C > 0 && a->C <= GEN_NT && a->C % ch == 0 && a->x_pitch % ch == 0 && a->x_coff % ch == 0 &&
                    (!a->y || (a->y_pitch % ch == 0 && a->y_coff % ch == 0)), "srk_chan_stats: C=%d (multiple of %d, <= %d) / alignment", a->C, ch, GEN_NT);
  SRK_CHECK_ARG(!a->gate_out || (a->mode == 2 && a->slope && !a->shift && !a->shift_out && a->gate_pitch % ch == 0 && a->gate_pitch >= a->C &&
                                 (a->slope_stride == 0 || ((uintptr_t)a->slope & 15) == 0)),
                "srk_chan_stats: gate_out needs mode 2, the (16-byte aligned) slope and an aligned pitch");
  return 0;
}
static int chan_finalize_check(const srk_chan_finalize_args* a) {
  SRK_CHECK_ARG(a && a->partial && a->out && a->nblocks > 0, "srk_chan_finalize: null pointer");
  SRK_CHECK_ARG(a->C > 0 && a->C <= GEN_NT && a->Creal >= 0 && a->Creal <= a->C, "srk_chan_finalize: C=%d Creal=%d", a->C, a->Creal);
  SRK_CHECK_ARG(a->mode >= 0 && a->mode <= 4, "srk_chan_finalize: mode %d", a->mode);
  SRK_CHECK_ARG(a->mode != 1 || (a->mean && a->weight && a->bias && (!a->running_mean == !a->running_var)), "srk_chan_finalize: mode 1 needs mean, weight, bias");
  SRK_CHECK_ARG((a->mode != 2 && a->mode != 3) || (a->invstd && a->gamma && (a->mode == 3 || a->mean)), "srk_chan_finalize: backward needs mean, invstd, gamma");
  return 0;
}
template <bool FUSED> static int chan_stats_launch(const srk_chan_stats_args* a, const srk_chan_finalize_args& f, int* counter, srk_stream_t stream) {
  const int nb = stats_blocks(a->P);
  const long long ppb = (a->P + nb - 1) / nb;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (a->dtype) {
    case SRK_BF16: hipLaunchKernelGGL((chan_stats_kernel<SRK_BF16, FUSED>), dim3(nb), dim3(GEN_NT), 0, st, *a, ppb, f, counter); break;
    case SRK_F16: hipLaunchKernelGGL((chan_stats_kernel<SRK_F16, FUSED>), dim3(nb), dim3(GEN_NT), 0, st, *a, ppb, f, counter); break;
    case SRK_F32: hipLaunchKernelGGL((chan_stats_kernel<SRK_F32, FUSED>), dim3(nb), dim3(GEN_NT), 0, st, *a, ppb, f, counter); break;
    default: srk_set_error("dtype %d", (int)a->dtype); return SRK_E_BADARG;
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_chan_stats(const srk_chan_stats_args* a, srk_stream_t stream) {
  if (const int rc = chan_stats_check(a)) return rc;
  if (a->P <= 0) return 0;
  return chan_stats_launch<false>(a, srk_chan_finalize_args{}, nullptr, stream);
}

extern "C" int srk_chan_finalize(const srk_chan_finalize_args* a, srk_stream_t stream) {
  if (const int rc = chan_finalize_check(a)) return rc;
  hipLaunchKernelGGL(chan_finalize_kernel, dim3(1), dim3(GEN_NT), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_chan_stats_finalize(const srk_chan_stats_args* a, const srk_chan_finalize_args* f, int* counter, srk_stream_t stream) {
  if (const int rc = chan_stats_check(a)) return rc;
  SRK_CHECK_ARG(f && counter && a->P > 0, "srk_chan_stats_finalize: null pointer / no pixels");
  srk_chan_finalize_args g = *f;
  g.partial = a->partial;
  g.partial2 = a->mode == 3 ? a->partial2 : nullptr;
  g.nblocks = stats_blocks(a->P);
  SRK_CHECK_ARG(g.C == a->C, "srk_chan_stats_finalize: C %d vs %d", g.C, a->C);
  if (const int rc = chan_finalize_check(&g)) return rc;
  return chan_stats_launch<true>(a, g, counter, stream);
}

extern "C" int srk_chan_apply(const srk_chan_apply_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->out, "srk_chan_apply: null pointer");
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->C > 0 && a->C % ch == 0 && a->x_pitch % ch == 0 && a->x_coff % ch == 0 && a->out_pitch % ch == 0 && a->out_coff % ch == 0 &&
                    (!a->y || (a->y_pitch % ch == 0 && a->y_coff % ch == 0)) && (!a->z || (a->z_pitch % ch == 0 && a->z_coff % ch == 0)),
                "srk_chan_apply: channels / pitches must be 16-byte multiples");
  SRK_CHECK_ARG((!a->z && !a->post_prelu) || a->slope, "srk_chan_apply: gate / PReLU needs the slope");
  SRK_CHECK_ARG((((uintptr_t)a->a | (uintptr_t)a->b | (uintptr_t)a->d | (uintptr_t)a->gate_a | (uintptr_t)a->gate_d |
                  (a->slope_stride ? (uintptr_t)a->slope : 0)) & 15) == 0, "srk_chan_apply: the per-channel vectors must be 16-byte aligned");
  SRK_CHECK_ARG(!a->gate_a || (a->gate_d && a->z && a->slope), "srk_chan_apply: gate_a needs gate_d, z and the slope");
  if (a->P <= 0) return 0;
  GEN_DISPATCH(chan_apply_kernel, a->dtype, grid_for(a->P * (a->C / ch)), reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

