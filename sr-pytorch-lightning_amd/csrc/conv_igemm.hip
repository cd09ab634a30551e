// Implicit-GEMM 'same' convolution (1x1 / 3x3, stride 1) on gfx950 MFMA, NHWC activations.
//
//   Y^T[co][pixel] = sum_{tap, ci} Wpk[tap][ci][co] * X[pixel + tap][ci]
//
// MFMA 32x32 tiles with A = weights (rows = output channels) and B = pixels (columns), so every lane
// owns ONE pixel and 4-channel register groups: the epilogue (bias, ReLU, res_scale, residual add,
// ReLU-backward mask, PixelShuffle store, channel-slice store) is per-lane with 8/16-byte accesses.
//
// Workgroup = 16x16 output pixels x TC output channels.  Per 64-channel (128-byte) input block the
// (16+K-1)^2 halo tile is staged ONCE into a swizzled LDS image (srk_common.h) and all K*K taps read
// it at shifted addresses; the weight slab of one (block, tap) is double-buffered through registers
// (global loads issued before the tap's MFMAs, LDS write after them, one barrier per tap).
//
// Replaces (reference file:line): DefaultConv2d models/common.py:7-30 and the fused elementwise ops
// listed in include/srk.h.  The same kernel is the data-gradient when given dgrad-packed weights.
#include <stdlib.h>
#include "srk_common.h"

// store policy of the conv epilogues (16 = sc1, write-through).  Plain stores leave the output dirty in the XCDs' L2s until the kernel's end; the
// write-back then sits in front of the NEXT launch (all-workgroup stamps: 3.8 -> 1.8 us between two dependent launches).  At batch 256 the step's
// launches are bound by memory throughput and gain nothing (+0.2 %); a chain of small launches gains what the write-back cost per link: EDSR-baseline
// batch 64 30.4k -> 32.0k patches/s, RDN batch 16 +1.3 %, same box (conv_pair.hip: RCAN batch 16 +7.6 %).  -DSRK_ST_AUX=0 / 2 (nt) for A/B builds.
#ifndef SRK_WS_STAMPS
#define SRK_WS_STAMPS 0
#endif
#ifndef SRK_WS_ABLATE
#define SRK_WS_ABLATE 0        // timing ablations (wrong results): 1 = skip MFMAs, 2 = skip epilogue, 4 = skip halo DMA, 16 = no weight-order rotation
#endif
#ifndef SRK_ST_AUX
#define SRK_ST_AUX 16         // cache policy of the quad epilogue's stores (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif

namespace {

template <int DT, int TC, int KS> struct ConvCfg {
  typedef DTraits<DT> Tr;
  static constexpr int CH = Tr::CH;
  static constexpr int NW = (TC == 128) ? 8 : 4;
  static constexpr int NT = NW * 64;
  static constexpr int WAVES_C = (TC == 128) ? 2 : 1;
  static constexpr int WAVES_P = NW / WAVES_C;        // 4 pixel groups of 4 tile rows
  static constexpr int CB_W = (TC / WAVES_C) / 32;    // 32-channel blocks per wave (1 or 2)
  static constexpr int PB_W = 8 / WAVES_P;            // 32-pixel blocks (2 rows x 16) per wave
  static constexpr int PAD = KS / 2;
  static constexpr int TIN = 16 + KS - 1;             // halo tile edge
  static constexpr int PITCH = (TIN + 1) & ~1;        // even row pitch (pixels)
  static constexpr int XS_BYTES = TIN * PITCH * 128;
  static constexpr int WS_BYTES = 8 * TC * 16;        // one weight slab: 8 chunks x TC rows x 16 B
  static constexpr int LDS_BYTES = XS_BYTES + 2 * WS_BYTES;
  static constexpr int XPIECES = TIN * TIN * 8;       // 16-byte pieces of the halo tile
  static constexpr int WPT = (8 * TC) / NT;           // weight pieces per thread
};


// ---- epilogue shared by the streaming and the weight-stationary kernels ---------------------------------
// lane = one output pixel (gy,gx); acc[cb][pb] register e = channel co0 + cb*32 + (e&3) + 8*(e>>2) + 4*h.
// The bias is already in the accumulators (acc_init_bias).  Mode branches are taken once per tile, the
// residual / mask loads of all channel groups are issued back to back before their first use, and the
// per-group channel offsets are compile-time immediates.
template <int CB_W, int PB_W>
SRK_DEV void acc_init_bias(f32x16 (&acc)[CB_W][PB_W], const float* bias, int co_lane) {
  // bias is stored in MFMA-row order; co_lane = first ROW of this lane's register group 0 (row0 + 4*h)
#pragma unroll
  for (int cb = 0; cb < CB_W; ++cb) {
    f32x4 b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = bias ? *reinterpret_cast<const f32x4*>(bias + co_lane + cb * 32 + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pb = 0; pb < PB_W; ++pb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[cb][pb][4 * i + 0] = b[i].x; acc[cb][pb][4 * i + 1] = b[i].y;
        acc[cb][pb][4 * i + 2] = b[i].z; acc[cb][pb][4 * i + 3] = b[i].w;
      }
  }
}

// GOFF(g): channel offset (relative to the lane's first channel) of 4-channel group g = 4*cb + i
#define SRK_GOFF(g) (CB_W == 2 ? ((g) >> 2) * 16 + ((g) & 3) * 4 : ((g) & 3) * 4)
template <int DT, int CB_W, int PB_W>
SRK_DEV void conv_epilogue(const srk_conv_args& a, f32x16 (&acc)[CB_W][PB_W], int n, int y0, int x0, int co0,
                           const int (&pyb)[PB_W], int px, int h) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int NG = CB_W * 4;                 // 4-channel groups per lane (row permutation: srk_common.h row_to_chan)
  const int H = a.H, W = a.W;
  const int mode = a.out_mode;
  const int rr = a.ps_r > 1 ? a.ps_r : 1;
  const float scale = a.scale;
  const bool relu = a.relu != 0;
  const int col = co0 + (CB_W == 2 ? 32 : 16) * h;   // this lane's first channel (its 16*CB_W channels are contiguous)
#pragma unroll
  for (int pb = 0; pb < PB_W; ++pb) {
    const int gy = y0 + pyb[pb], gx = x0 + px;
    if (gy >= H || gx >= W) continue;
    float v[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[g >> 2][pb][4 * (g & 3) + e];
        if (relu) t = relu_f32(t);
        v[g][e] = t * scale;
      }

    if (mode == SRK_OUT_NHWC) {
      const size_t pix = (size_t)(n * H + gy) * W + gx;
      if (a.res) {
        const elem* rp = reinterpret_cast<const elem*>(a.res) + pix * a.res_pitch + a.res_coff + col;
        float q[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int off = SRK_GOFF(g);
          if (col + off < a.Cout) load4<DT>(rp + off, q[g]);
          else q[g][0] = q[g][1] = q[g][2] = q[g][3] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g][e] += q[g][e];
      }
      if (a.mask) {
        const elem* mp = reinterpret_cast<const elem*>(a.mask) + pix * a.mask_pitch + a.mask_coff + col;
        float q[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int off = SRK_GOFF(g);
          if (col + off < a.Cout && col + off >= a.mask_from) load4<DT>(mp + off, q[g]);
          else q[g][0] = q[g][1] = q[g][2] = q[g][3] = 1.f;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g][e] = q[g][e] > 0.f ? v[g][e] : 0.f;
      }
      elem* op = reinterpret_cast<elem*>(a.out) + pix * a.out_pitch + a.out_coff + col;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int off = SRK_GOFF(g);
        if (col + off < a.Cout) store4<DT>(op + off, v[g]);
      }
    } else if (mode == SRK_OUT_NHWC_PS) {
      // out[n][gy*r+i][gx*r+j][c], packed channel co' = (i*r+j)*Cc + c ; Cc % 4 == 0 keeps a group in one pixel
      const int Cc = a.Cout / (rr * rr);
      const bool pow2 = (Cc & (Cc - 1)) == 0;
      const int sh = 31 - __builtin_clz(Cc);
      const size_t rowpix = (size_t)(n * H * rr + gy * rr) * (W * rr) + gx * rr;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int co = col + SRK_GOFF(g);
        if (co >= a.Cout) continue;
        const int ij = pow2 ? (co >> sh) : (co / Cc);
        const int c = co - ij * Cc;
        const int si = ij / rr, sj = ij - si * rr;
        const size_t pix = rowpix + (size_t)si * (W * rr) + sj;
        if (a.res) {
          float q[4];
          load4<DT>(reinterpret_cast<const elem*>(a.res) + pix * a.res_pitch + a.res_coff + c, q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g][e] += q[e];
        }
        if (a.mask && co >= a.mask_from) {
          float q[4];
          load4<DT>(reinterpret_cast<const elem*>(a.mask) + pix * a.mask_pitch + a.mask_coff + c, q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g][e] = q[e] > 0.f ? v[g][e] : 0.f;
        }
        store4<DT>(reinterpret_cast<elem*>(a.out) + pix * a.out_pitch + a.out_coff + c, v[g]);
      }
    } else {
      // fp32 NCHW (model boundary), optional PixelShuffle: out[n][c][gy*r+i][gx*r+j], co = c*r*r + i*r + j
      const int r2 = rr * rr, Cc = a.Cout / r2;
      float* const o = reinterpret_cast<float*>(a.out);
      const float* const rs = reinterpret_cast<const float*>(a.res);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ce = col + SRK_GOFF(g) + e;
          if (ce < a.Cout) {
            const int c = ce / r2, ij = ce - c * r2;
            const int si = ij / rr, sj = ij - si * rr;
            const size_t idx = ((size_t)(n * Cc + c) * (H * rr) + gy * rr + si) * (W * rr) + gx * rr + sj;
            float val = v[g][e];
            if (rs) val += rs[idx];
            if (a.post_add) val += a.post_add[c];
            o[idx] = val;
          }
        }
      }
    }
  }
}


// ---- fast epilogue: NHWC / NHWC_PS stores of a FULL 64-channel tile through buffer instructions ----------
// One 32-bit byte offset per (lane, pixel block); the 8 channel groups of a lane are instruction immediates;
// out-of-image pixels get an out-of-range offset (loads return 0, stores are dropped), so there is no
// per-group predicate, branch or 64-bit address arithmetic.  Preconditions are checked by the launcher
// (conv_fast_ok): tensors < 2 GiB, stored channels a multiple of 64, PixelShuffle planes a multiple of 64.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned SRK_OOB = 0x80000000u;

SRK_DEV __amdgpu_buffer_rsrc_t big_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// 16 contiguous channels of one lane <-> 16 floats, as 16-byte buffer accesses (2 for 16-bit types, 4 for fp32)
template <int DT> SRK_DEV void buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int imm_bytes, float q[16]) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + imm_bytes + 16 * t, 0, 0);      // (SRK_ST_AUX is a STORE policy)
      unpack2<DT>(raw.x, q[8 * t + 0], q[8 * t + 1]);
      unpack2<DT>(raw.y, q[8 * t + 2], q[8 * t + 3]);
      unpack2<DT>(raw.z, q[8 * t + 4], q[8 * t + 5]);
      unpack2<DT>(raw.w, q[8 * t + 6], q[8 * t + 7]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + imm_bytes + 16 * t, 0, 0);      // (SRK_ST_AUX is a STORE policy)
      q[4 * t + 0] = __uint_as_float(raw.x); q[4 * t + 1] = __uint_as_float(raw.y);
      q[4 * t + 2] = __uint_as_float(raw.z); q[4 * t + 3] = __uint_as_float(raw.w);
    }
  }
}

template <int DT> SRK_DEV void buf_store16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int imm_bytes, const float v[16], bool relu_packed) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4 raw;
      raw.x = pack2<DT>(v[8 * t + 0], v[8 * t + 1]);
      raw.y = pack2<DT>(v[8 * t + 2], v[8 * t + 3]);
      raw.z = pack2<DT>(v[8 * t + 4], v[8 * t + 5]);
      raw.w = pack2<DT>(v[8 * t + 6], v[8 * t + 7]);
      if (relu_packed) { raw.x = relu_pk16<DT>(raw.x); raw.y = relu_pk16<DT>(raw.y); raw.z = relu_pk16<DT>(raw.z); raw.w = relu_pk16<DT>(raw.w); }
      __builtin_amdgcn_raw_buffer_store_b128(raw, rs, voff + imm_bytes + 16 * t, 0, SRK_ST_AUX);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4 raw;
      raw.x = __float_as_uint(v[4 * t + 0]); raw.y = __float_as_uint(v[4 * t + 1]);
      raw.z = __float_as_uint(v[4 * t + 2]); raw.w = __float_as_uint(v[4 * t + 3]);
      __builtin_amdgcn_raw_buffer_store_b128(raw, rs, voff + imm_bytes + 16 * t, 0, SRK_ST_AUX);
    }
  }
}

// opix[pb]: destination pixel index of this lane (or -1); cdst: destination channel of the lane's first channel
// (its 16*CB_W channels are contiguous: accumulator cb, register e <-> channel cdst + 16*cb + e);
// cfirst: conv output channel of that first channel (for mask_from)
template <int DT, int CB_W, int PB_W>
SRK_DEV void conv_epilogue_fast(const srk_conv_args& a, f32x16 (&acc)[CB_W][PB_W], const int (&opix)[PB_W], int cdst, int cfirst) {
  typedef DTraits<DT> Tr;
  constexpr int ESZ = 16 / Tr::CH;
  const __amdgpu_buffer_rsrc_t ro = big_rsrc(a.out);
  const bool has_res = a.res != nullptr, has_mask = a.mask != nullptr, relu = a.relu != 0;
  const float scale = a.scale;
  const bool simple = !has_res && !has_mask && scale == 1.f;     // conv (+ReLU): convert, packed ReLU, store
#pragma unroll
  for (int pb = 0; pb < PB_W; ++pb) {
    const bool ok = opix[pb] >= 0;
    const unsigned vo = ok ? (unsigned)((opix[pb] * a.out_pitch + a.out_coff + cdst) * ESZ) : SRK_OOB;
    const unsigned vr = (ok && has_res) ? (unsigned)((opix[pb] * a.res_pitch + a.res_coff + cdst) * ESZ) : SRK_OOB;
    const unsigned vm = (ok && has_mask) ? (unsigned)((opix[pb] * a.mask_pitch + a.mask_coff + cdst) * ESZ) : SRK_OOB;
#pragma unroll
    for (int cb = 0; cb < CB_W; ++cb) {
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[cb][pb][e];
      if (simple && Tr::IS16) {
        buf_store16<DT>(ro, vo, cb * 16 * ESZ, v, relu);
        continue;
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = relu_f32(v[e]);
      }
      if (scale != 1.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] *= scale;
      }
      if (has_res) {
        float q[16];
        buf_load16<DT>(big_rsrc(a.res), vr, cb * 16 * ESZ, q);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += q[e];
      }
      if (has_mask) {
        float q[16];
        buf_load16<DT>(big_rsrc(a.mask), vm, cb * 16 * ESZ, q);
        const bool use = cfirst + cb * 16 >= a.mask_from;      // mask_from is a multiple of 16
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (!use || q[e] > 0.f) ? v[e] : 0.f;
      }
      buf_store16<DT>(ro, vo, cb * 16 * ESZ, v, false);
    }
  }
}

// destination pixel / channel of a 64-aligned channel tile starting at conv channel c0 (wave-uniform part)
struct FastDst { int si, sj, rr, cbase; };
SRK_DEV FastDst fast_dst(const srk_conv_args& a, int c0) {
  FastDst d;
  d.rr = (a.out_mode == SRK_OUT_NHWC_PS && a.ps_r > 1) ? a.ps_r : 1;
  if (d.rr > 1) {
    const int Cc = a.Cout / (d.rr * d.rr);
    const int ij = c0 / Cc;
    d.cbase = c0 - ij * Cc;
    d.si = ij / d.rr;
    d.sj = ij - d.si * d.rr;
  } else {
    d.cbase = c0; d.si = 0; d.sj = 0;
  }
  return d;
}
SRK_DEV int fast_opix(const srk_conv_args& a, const FastDst& d, int n, int gy, int gx) {
  if (gy >= a.H || gx >= a.W) return -1;
  return (n * a.H * d.rr + gy * d.rr + d.si) * (a.W * d.rr) + gx * d.rr + d.sj;
}

// ---- planar epilogue for <= 4 output channels without PixelShuffle: the EDSR / RCAN / RDN tail conv (Cout = 3) -------
// fp32 NCHW store (+ planar fp32 residual, + per-channel post_add).  Only the h = 0 lanes hold real channels
// (registers 0..3 of channel block 0).  The generic epilogue walks all 16 padded channels of a lane behind
// per-channel branches and issues each post_add / residual load right before its use (one exposed memory latency
// per channel: 8-10k cycles per tile, 3x the tile's MFMA time); here post_add arrives in registers (loaded once
// per kernel), the residual loads are issued together, and padding lanes are dropped by out-of-range offsets.
template <int DT, int PB_W>
SRK_DEV void conv_epilogue_planar4(const srk_conv_args& a, f32x16 (&acc)[1][PB_W], int n, int y0, int x0,
                                   const int (&pyb)[PB_W], int px, int h, const float (&pa)[4]) {
  const int H = a.H, W = a.W, plane = H * W;
  const __amdgpu_buffer_rsrc_t ro = big_rsrc(a.out), rres = big_rsrc(a.res ? a.res : a.out);
  const bool has_res = a.res != nullptr, relu = a.relu != 0;
  const float scale = a.scale;
  unsigned vo[PB_W][4];
  float q[PB_W][4];
#pragma unroll
  for (int pb = 0; pb < PB_W; ++pb) {
    const int gy = y0 + pyb[pb], gx = x0 + px;
    const bool ok = h == 0 && gy < H && gx < W;
    const int base = (n * a.Cout * H + gy) * W + gx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vo[pb][e] = (ok && e < a.Cout) ? (unsigned)(base + e * plane) * 4u : SRK_OOB;
      q[pb][e] = has_res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, vo[pb][e], 0, 0)) : 0.f;
    }
  }
#pragma unroll
  for (int pb = 0; pb < PB_W; ++pb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[0][pb][e];
      if (relu) v = relu_f32(v);
      v = v * scale + q[pb][e];
      v += pa[e];
      if (e < a.Cout) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo[pb][e], 0, SRK_ST_AUX);
    }
}

// ---- quad-transposed epilogue (weight-stationary kernel, 16-bit types, full 64-channel tile) --------------------
// With lane = pixel, one dwordx4 store instruction drops 64 pieces of 16 bytes into 64 different 128-byte lines, and
// the memory path pays per line touched, not per byte: measured (tools/ubench/store_patterns.hip, 32 CUs active)
// 15 B/clk/CU for that pattern, 31 for 32-byte runs, 50 for 64-byte runs, 58 for whole lines; chip-wide the 16-byte
// pattern saturates the L2 request rate at 3.0-3.5 TB/s.  A lane holds 64 contiguous bytes of its pixel (4 pieces),
// so a 4x4 transpose of pieces inside each quad of lanes (4 x-adjacent pixels; two v_mov_dpp quad_perm stages) turns
// "instruction k = piece k of 64 pixels" into "instruction j = pixel j of every quad, 4 lanes covering its 64
// contiguous bytes": 4x fewer line touches for the stores and for the residual / mask loads, which are issued
// directly in the transposed layout.  Arithmetic is unchanged (the fp32 values are transposed before the
// bias/ReLU/scale/residual/mask sequence, or, for conv(+ReLU) alone, the packed 16-bit results are).
SRK_DEV uint32_t dpp_quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }   // [1,0,3,2]
SRK_DEV uint32_t dpp_quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }   // [2,3,0,1]
// in: R[k] = item k of this lane; out: R[j] = item (lane & 3) of lane (quad base + j)
SRK_DEV void quad_transpose4(uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3, bool b0, bool b1) {
  {
    const uint32_t s01 = b0 ? r0 : r1, s23 = b0 ? r2 : r3;
    const uint32_t g01 = dpp_quad_xor1(s01), g23 = dpp_quad_xor1(s23);
    r0 = b0 ? g01 : r0; r1 = b0 ? r1 : g01;
    r2 = b0 ? g23 : r2; r3 = b0 ? r3 : g23;
  }
  {
    const uint32_t s02 = b1 ? r0 : r2, s13 = b1 ? r1 : r3;
    const uint32_t g02 = dpp_quad_xor2(s02), g13 = dpp_quad_xor2(s13);
    r0 = b1 ? g02 : r0; r2 = b1 ? r2 : g02;
    r1 = b1 ? g13 : r1; r3 = b1 ? r3 : g13;
  }
}

// (quad_transpose8_dpp, the v_cndmask_b32_dpp form of this transpose for two register quartets at once: srk_common.h)

template <int DT>
SRK_DEV void conv_epilogue_quad(const srk_conv_args& a, f32x16 (&acc)[2][2], int pbase0, int pbase1, int okmask0, int okmask1,
                                int pstep, int cl, bool use_mask, int qi, int hbit = 0) {
  // pbaseX: destination pixel index of the quad's first pixel for pixel block X; okmaskX bit j: pixel j of the quad
  // is inside the image; pstep: destination pixel stride between x-neighbours (r for a pixel-shuffled store);
  // cl: destination channel of this lane's 8-channel piece AFTER the transpose; qi = lane & 3
  static_assert(DTraits<DT>::IS16, "16-bit types only");
  const __amdgpu_buffer_rsrc_t ro = big_rsrc(a.out);
  const bool has_res = a.res != nullptr, has_mask = a.mask != nullptr, relu = a.relu != 0;
  const float scale = a.scale;
  const bool simple = !has_res && !has_mask && scale == 1.f;
  const bool b0 = qi & 1, b1 = qi & 2;
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int pbase = pb ? pbase1 : pbase0, okm = pb ? okmask1 : okmask0;
    unsigned vo[4], vr[4], vm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (okm >> j) & 1;
      const int pix = pbase + j * pstep;
      vo[j] = ok ? (unsigned)((pix * a.out_pitch + a.out_coff + cl) * 2) : SRK_OOB;
      vr[j] = (ok && has_res) ? (unsigned)((pix * a.res_pitch + a.res_coff + cl) * 2) : SRK_OOB;
      vm[j] = (ok && has_mask) ? (unsigned)((pix * a.mask_pitch + a.mask_coff + cl) * 2) : SRK_OOB;
    }
    if (simple) {
      uint32_t P[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          P[k][d] = pack2<DT>(acc[k >> 1][pb][8 * (k & 1) + 2 * d], acc[k >> 1][pb][8 * (k & 1) + 2 * d + 1]);
          if (relu) P[k][d] = relu_pk16<DT>(P[k][d]);
        }
      if (a.relu_bits) {
        // ReLU sign bits of this lane's pixel (natural layout: pixel qi of the quad, channels 32 h ..): dword i = 4 k + d of the packed
        // results holds channels 32 h + 2 i, + 1 -> bits i and 16 + i (set = the stored value is > 0).  The data gradient of this conv's
        // INPUT masks with these 4 bytes per pixel and half instead of re-reading the 64-byte activation (include/srk.h: relu_bits)
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          uint32_t t;
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(P[i >> 2][i & 3]), "s"(0x00010001u));
          word |= t << i;
        }
        const bool ok = (okm >> qi) & 1;
        const unsigned off = ok ? (unsigned)(((pbase + qi * pstep) * 2 + hbit) * 4) : SRK_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(word, big_rsrc(a.relu_bits), off, 0, SRK_ST_AUX);
      }
      quad_transpose8_dpp(P[0][0], P[1][0], P[2][0], P[3][0], P[0][1], P[1][1], P[2][1], P[3][1]);
      quad_transpose8_dpp(P[0][2], P[1][2], P[2][2], P[3][2], P[0][3], P[1][3], P[2][3], P[3][3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 raw = {P[j][0], P[j][1], P[j][2], P[j][3]};
        __builtin_amdgcn_raw_buffer_store_b128(raw, ro, vo[j], 0, SRK_ST_AUX);
      }
      continue;
    }
    // residual / mask pieces, already in the transposed layout (issued first: their latency hides behind the transposes)
    u32x4 rq[4], mq[4];
    if (has_res) {
      const __amdgpu_buffer_rsrc_t rr = big_rsrc(a.res);
#pragma unroll
      for (int j = 0; j < 4; ++j) rq[j] = __builtin_amdgcn_raw_buffer_load_b128(rr, vr[j], 0, 0);
    }
    if (has_mask) {
      const __amdgpu_buffer_rsrc_t rm = big_rsrc(a.mask);
#pragma unroll
      for (int j = 0; j < 4; ++j) mq[j] = __builtin_amdgcn_raw_buffer_load_b128(rm, vm[j], 0, 0);
    }
    uint32_t F[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) F[k][e] = __float_as_uint(acc[k >> 1][pb][8 * (k & 1) + e]);
#pragma unroll
    for (int e = 0; e < 8; ++e) quad_transpose4(F[0][e], F[1][e], F[2][e], F[3][e], b0, b1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(F[j][e]);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu_f32(v[e]);
      }
      if (scale != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= scale;
      }
      if (has_res) {
        float q[8];
        unpack2<DT>(rq[j].x, q[0], q[1]); unpack2<DT>(rq[j].y, q[2], q[3]);
        unpack2<DT>(rq[j].z, q[4], q[5]); unpack2<DT>(rq[j].w, q[6], q[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += q[e];
      }
      if (has_mask) {
        float q[8];
        unpack2<DT>(mq[j].x, q[0], q[1]); unpack2<DT>(mq[j].y, q[2], q[3]);
        unpack2<DT>(mq[j].z, q[4], q[5]); unpack2<DT>(mq[j].w, q[6], q[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (!use_mask || q[e] > 0.f) ? v[e] : 0.f;
      }
      u32x4 raw;
      raw.x = pack2<DT>(v[0], v[1]); raw.y = pack2<DT>(v[2], v[3]);
      raw.z = pack2<DT>(v[4], v[5]); raw.w = pack2<DT>(v[6], v[7]);
      __builtin_amdgcn_raw_buffer_store_b128(raw, ro, vo[j], 0, SRK_ST_AUX);
    }
  }
}

// ---- the same epilogue with PREFETCHED residual / mask (conv_ws_kernel<..., EARLY = true>) ----------------------------
// vmcnt retires in order.  A residual load issued in the epilogue phase sits behind the 11 DMA pieces of the next halo
// tile, so its data only becomes usable when that whole tile has landed (3-5k cycles: s_memtime stamps showed the
// first pixel block's arithmetic finishing 5.8k cycles into a 9.3k-cycle phase with the data long there, because the
// compiler's own wait in front of the first use is vmcnt(0) whenever the loads come from an earlier loop iteration).
// So: the pieces of tile j are requested at the START of tile j's MFMA phase, from inline asm (the compiler neither
// sinks them nor waits for them) and completed by an explicit s_waitcnt at the END of that phase, when they have long
// landed; the epilogue phase then runs   DMA of the next tile | compute(0) | store(0) | compute(1) | store(1)
// on registers that are already valid.
// "early" loads the residual if there is one, else the mask (4 x 16 bytes per lane and pixel block).
struct QuadGeo {          // where this lane's quad lands: see conv_ws_kernel
  int pbase[2];           // destination pixel index of the quad's first pixel, per pixel block
  int okmask[2];          // bit j: pixel j of the quad is inside the image
  int pstep;              // destination pixel stride between x-neighbours (r for a pixel-shuffled store)
  int cl;                 // destination channel of this lane's 8-channel piece AFTER the transpose
};
struct QuadRegs { u32x4 r[4]; };

SRK_DEV unsigned quad_off(const QuadGeo& g, int pb, int j, int pitch, int coff) {
  return ((g.okmask[pb] >> j) & 1) ? (unsigned)(((g.pbase[pb] + j * g.pstep) * pitch + coff + g.cl) * 2) : SRK_OOB;
}

SRK_DEV u32x4 buf_load16_hidden(i32x4 rsrc, unsigned voff) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
  return r;      // NOT valid until the caller's own s_waitcnt (quad_early_wait)
}
// Prefetch in the NATURAL accumulator layout: piece k = channels 8k..8k+7 of this lane's own 32 channels of its own
// pixel (pixel qi of the quad).  The arithmetic then runs before the transpose, on 16-bit-packed results, which halves
// the transpose work (64 instead of 128 cross-lane moves+selects per pixel block): the epilogue phase is bound by
// vector-instruction issue next to the other group's MFMA wave (stamps: 2.4-3.1k cycles per pixel block with the fp32
// transposes, all operands already in registers).  The 16-byte-per-line load pattern this needs is the slow one for
// the memory path, but it is issued in the MFMA phase, where this group has no other memory traffic.
struct QuadPre { i32x4 rs; unsigned base[2]; };
// EM = 3: the 4-byte sign words of this lane's pixel (pixel qi of the quad) and half h: [pixel][2] u32
SRK_DEV QuadPre quad_early_setup_bits(const srk_conv_args& a, const QuadGeo& g, int qi, int h) {
  QuadPre p;
  p.rs = make_rsrc4(a.mask_bits, 0x7fffffffu);
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const bool ok = (g.okmask[pb] >> qi) & 1;
    p.base[pb] = ok ? (unsigned)(((g.pbase[pb] + qi * g.pstep) * 2 + h) * 4) : SRK_OOB;
  }
  return p;
}
SRK_DEV void quad_early_word(const QuadPre& p, int pb, QuadRegs& e) {
  uint32_t r;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(r) : "v"(p.base[pb]), "s"(p.rs) : "memory");
  e.r[0].x = r;      // NOT valid until quad_early_wait
}

SRK_DEV QuadPre quad_early_setup(const srk_conv_args& a, const QuadGeo& g, int qi) {
  QuadPre p;
  const bool has_res = a.res != nullptr, has_mask = a.mask != nullptr;
  p.rs = make_rsrc4(has_res ? a.res : a.mask, 0x7fffffffu);
  const int pitch = has_res ? a.res_pitch : a.mask_pitch, coff = has_res ? a.res_coff : a.mask_coff;
  const int c0 = g.cl - 8 * qi;                 // destination channel of this lane's first channel (g.cl is post-transpose)
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const bool ok = (g.okmask[pb] >> qi) & 1;
    p.base[pb] = ok ? (unsigned)(((g.pbase[pb] + qi * g.pstep) * pitch + coff + c0) * 2) : SRK_OOB;
  }
  return p;
}
// piece k of pixel block pb (out-of-image pixels: base is out of range and stays so with +16k)
SRK_DEV void quad_early_piece(const QuadPre& p, int pb, int k, QuadRegs& e) {
  e.r[k] = buf_load16_hidden(p.rs, p.base[pb] + 16u * k);     // unconditional: EARLY is only dispatched with a residual or a mask
}
// all vector-memory operations of this wave older than its `younger` most recent ones have completed; ties the
// prefetched registers to the wait so that no use can be scheduled in front of it
template <int YOUNGER>
SRK_DEV void quad_early_wait(QuadRegs& e0, QuadRegs& e1) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(e0.r[0]), "+v"(e0.r[1]), "+v"(e0.r[2]), "+v"(e0.r[3]), "+v"(e1.r[0]), "+v"(e1.r[1]), "+v"(e1.r[2]), "+v"(e1.r[3])
               : "n"(YOUNGER) : "memory");
}

// accumulators of pixel block pb (+ prefetched residual OR mask, natural layout) -> the 4 packed 16-byte pieces this
// lane stores after the quad transpose (out.r[j]: pixel j of the quad).  No memory operation in here: a compiler-visible
// load would put `s_waitcnt vmcnt(0)` at the merge point of every path, i.e. make every tile wait for the halo DMA issued
// just before (measured: 4.6k of a 9.4k-cycle phase); the dispatcher sends residual + mask to the plain variant.
// EM (compile time): 1 = residual, 2 = ReLU-backward mask.  Round 4: the run-time form of this function (flags tested per 16-byte
// piece) compiled to 857 vector instructions per tile and wave against 144 MFMAs (PMC: 6 VALU per MFMA) -- canonicalising maxes,
// scaled AND unscaled values with selects between them, float compares for the mask, 12 wave-uniform branches per pixel block --
// and the phase it bounds is vector-ISSUE-bound (~4 cycles per instruction beside the other group's MFMA wave).  Now: flags are
// template parameters or ONE uniform branch per pixel block, the mask is applied to the PACKED 16-bit results with packed integer
// ops (2 instead of 3 instructions per element, no unpack), the transposes take one instruction per register and stage.
// (mask_apply_pk16: srk_common.h)
template <int DT, int EM, bool SC, bool RL>
SRK_DEV void quad_compute_t(float scale, f32x16 (&acc)[2][2], int pb, const QuadRegs& e, bool use_mask_lo, bool use_mask_hi, QuadRegs& out) {
  static_assert(DTraits<DT>::IS16, "16-bit types only");
  uint32_t P[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] = acc[k >> 1][pb][8 * (k & 1) + x];
    if constexpr (RL) {
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] = relu_f32(v[x]);
    }
    // scalar v_mul_f32 / v_add_f32 from inline asm: left to hipcc, adjacent elements are SLP-packed into v_pk_mul_f32 / v_pk_add_f32, and a packed
    // f32 instruction beside the partner wave's MFMAs costs ~13 cycles more than a scalar one (MI355X_MICROARCH.md, constants table: "an
    // anti-lever beside MFMAs"); same roundings (one multiply, one add).  Residual flavour alone 45.8 -> 43.9 us (profiles/r6_experiments.txt 4)
    if constexpr (SC) {
#pragma unroll
      for (int x = 0; x < 8; ++x) asm("v_mul_f32_e32 %0, %1, %0" : "+v"(v[x]) : "s"(scale));
    }
    if constexpr (EM == 1) {
      float r8[8];
      unpack2<DT>(e.r[k].x, r8[0], r8[1]); unpack2<DT>(e.r[k].y, r8[2], r8[3]);
      unpack2<DT>(e.r[k].z, r8[4], r8[5]); unpack2<DT>(e.r[k].w, r8[6], r8[7]);
#pragma unroll
      for (int x = 0; x < 8; ++x) asm("v_add_f32_e32 %0, %1, %0" : "+v"(v[x]) : "v"(r8[x]));
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) P[k][d] = pack2<DT>(v[2 * d], v[2 * d + 1]);
    if constexpr (EM == 2) {
      if ((k >> 1) ? use_mask_hi : use_mask_lo) {      // channel block of piece k: 16 * (k >> 1); wave-uniform
        mask_apply_pk16(P[k][0], e.r[k].x); mask_apply_pk16(P[k][1], e.r[k].y);
        mask_apply_pk16(P[k][2], e.r[k].z); mask_apply_pk16(P[k][3], e.r[k].w);
      }
    }
    if constexpr (EM == 3) {
      // sign BITS (srk_conv_args.mask_bits): one dword per pixel and 32-channel half, bit i / 16 + i <-> the two elements of dword i
      const uint32_t word = e.r[0].x;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t sel = (word >> (4 * k + d)) & 0x00010001u;
        asm("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(P[k][d]) : "v"(sel));
      }
    }
  }
  quad_transpose8_dpp(P[0][0], P[1][0], P[2][0], P[3][0], P[0][1], P[1][1], P[2][1], P[3][1]);
  quad_transpose8_dpp(P[0][2], P[1][2], P[2][2], P[3][2], P[0][3], P[1][3], P[2][3], P[3][3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) out.r[j] = u32x4{P[j][0], P[j][1], P[j][2], P[j][3]};
}

// one wave-uniform branch per pixel block picks the instantiation (ReLU together with a residual or a mask is rare: it keeps one generic form)
template <int DT, int EM>
SRK_DEV void quad_compute(const srk_conv_args& a, f32x16 (&acc)[2][2], int pb, const QuadRegs& e, bool use_mask_lo,
                          bool use_mask_hi, QuadRegs& out) {
  const float scale = a.scale;
  if (a.relu) quad_compute_t<DT, EM, true, true>(scale, acc, pb, e, use_mask_lo, use_mask_hi, out);
  else if (scale != 1.f) quad_compute_t<DT, EM, true, false>(scale, acc, pb, e, use_mask_lo, use_mask_hi, out);
  else quad_compute_t<DT, EM, false, false>(scale, acc, pb, e, use_mask_lo, use_mask_hi, out);
}

SRK_DEV void quad_store(const srk_conv_args& a, const QuadGeo& g, int pb, const QuadRegs& out) {
  const __amdgpu_buffer_rsrc_t ro = big_rsrc(a.out);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_raw_buffer_store_b128(out.r[j], ro, quad_off(g, pb, j, a.out_pitch, a.out_coff), 0, SRK_ST_AUX);
}

template <int DT, int TC, int KS>
__global__ __launch_bounds__((ConvCfg<DT, TC, KS>::NT)) void conv_igemm_kernel(const srk_conv_args a, int tilesX,
                                                                              int tilesY, int ctiles, int fast) {
  typedef ConvCfg<DT, TC, KS> C;
  typedef typename C::Tr Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = C::CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ws = smem + C::XS_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wave_c = wave / C::WAVES_P, wave_p = wave % C::WAVES_P;
  const int wco0 = wave_c * (C::CB_W * 32);

  int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int ctile = lin % ctiles;
  int pt = lin / ctiles;
  const int tX = pt % tilesX;
  pt /= tilesX;
  const int tY = pt % tilesY;
  const int n = pt / tilesY;
  const int y0 = tY * 16, x0 = tX * 16;
  const int H = a.H, W = a.W;

  const int nch = a.Cin / CH;          // 16-byte chunks along Cin
  const int nblk = (nch + 7) >> 3;     // 128-byte blocks
  const elem* const xg = reinterpret_cast<const elem*>(a.x);
  const elem* const wg = reinterpret_cast<const elem*>(a.wpk);
  const int rin = a.x_ps > 1 ? a.x_ps : 1;
  const int Cs = a.Cin / (rin * rin);  // channels per stored pixel when x is pixel-shuffled

  // ---- stage the halo tile of input block cb into the swizzled LDS image -------------------------
  auto stage_x = [&](int cb) {
    const int nchb = min(8, nch - cb * 8);
#pragma unroll 1
    for (int base = tid; base < C::XPIECES; base += 4 * C::NT) {
      i32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * C::NT;
        v[u] = i32x4{0, 0, 0, 0};
        if (i < C::XPIECES) {
          const int s = i & 7, p = i >> 3;
          const int iy = p / C::TIN, ix = p - iy * C::TIN;
          const int c = s ^ swz(ix);                  // source chunk held by LDS slot s
          const int gy = y0 + iy - C::PAD, gx = x0 + ix - C::PAD;
          if (c < nchb && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int k0 = (cb * 8 + c) * CH;
            size_t off;
            if (rin == 1) {
              off = ((size_t)(n * H + gy) * W + gx) * a.x_pitch + a.x_coff + k0;
            } else {
              const int ij = k0 / Cs, c0 = k0 - ij * Cs;
              const int si = ij / rin, sj = ij - si * rin;
              off = ((size_t)(n * H * rin + gy * rin + si) * (W * rin) + gx * rin + sj) * a.x_pitch + a.x_coff + c0;
            }
            v[u] = gload16(xg + off);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * C::NT;
        if (i < C::XPIECES) {
          const int s = i & 7, p = i >> 3;
          const int iy = p / C::TIN, ix = p - iy * C::TIN;
          lds_write16(Xs + ((iy * C::PITCH + ix) << 7) + (s << 4), v[u]);
        }
      }
    }
  };

  // ---- weight slab (block cb, tap): global -> registers, registers -> LDS -----------------------
  i32x4 wreg[C::WPT];
  auto load_w = [&](int cb, int tap) {
    const int nchb = min(8, nch - cb * 8);
#pragma unroll
    for (int u = 0; u < C::WPT; ++u) {
      const int i = tid + u * C::NT;
      const int c = i / TC, co = i - c * TC;
      wreg[u] = i32x4{0, 0, 0, 0};
      if (c < nchb) {
        const size_t off = ((size_t)(tap * nch + cb * 8 + c) * a.CoutP + ctile * TC + co) * CH;
        wreg[u] = gload16(wg + off);
      }
    }
  };
  auto write_w = [&](char* Wb) {
#pragma unroll
    for (int u = 0; u < C::WPT; ++u) lds_write16(Wb + ((tid + u * C::NT) << 4), wreg[u]);
  };

  // ---- accumulators ------------------------------------------------------------------------------
  f32x16 acc[C::CB_W][C::PB_W];
  acc_init_bias<C::CB_W, C::PB_W>(acc, a.bias, ctile * TC + wco0 + 4 * h);

  const int px = r & 15;
  int pyb[C::PB_W];
#pragma unroll
  for (int pb = 0; pb < C::PB_W; ++pb) pyb[pb] = (wave_p * C::PB_W + pb) * 2 + (r >> 4);

  auto compute = [&](int tap, int nks, const char* Wb) {
    const int kh = tap / KS, kw = tap - kh * KS;
    const int g = swz(px + kw);
    const char* xb[C::PB_W];
#pragma unroll
    for (int pb = 0; pb < C::PB_W; ++pb) xb[pb] = Xs + (((pyb[pb] + kh) * C::PITCH + px + kw) << 7);
    const char* wb = Wb + ((wco0 + r) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < nks) {
        const int kc = 2 * ks + h;
        i32x4 af[C::CB_W], bf[C::PB_W];
#pragma unroll
        for (int cb = 0; cb < C::CB_W; ++cb) af[cb] = lds_read16(wb + ((kc * TC + cb * 32) << 4));
#pragma unroll
        for (int pb = 0; pb < C::PB_W; ++pb) bf[pb] = lds_read16(xb[pb] + ((kc ^ g) << 4));
#pragma unroll
        for (int cb = 0; cb < C::CB_W; ++cb)
#pragma unroll
          for (int pb = 0; pb < C::PB_W; ++pb) acc[cb][pb] = Tr::mma(af[cb], bf[pb], acc[cb][pb]);
      }
    }
  };

  // ---- main loop: Cin blocks x taps, one barrier per tap ------------------------------------------
  constexpr int NTAPS = KS * KS;
  stage_x(0);
  load_w(0, 0);
  write_w(Ws);
  __syncthreads();
  int it = 0;
  for (int cb = 0; cb < nblk; ++cb) {
    const int nks = min(8, nch - cb * 8) >> 1;
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; ++tap, ++it) {
      const bool last_tap = (tap == NTAPS - 1);
      const bool has_next = !(last_tap && cb == nblk - 1);
      if (has_next) load_w(last_tap ? cb + 1 : cb, last_tap ? 0 : tap + 1);
      compute(tap, nks, Ws + (it & 1) * C::WS_BYTES);
      if (has_next) {
        if (last_tap) {
          __syncthreads();     // every wave is done with this block's halo tile
          stage_x(cb + 1);
        }
        write_w(Ws + ((it + 1) & 1) * C::WS_BYTES);
      }
      __syncthreads();
    }
  }

  if constexpr (TC >= 64) {
    if (fast) {
      const FastDst d = fast_dst(a, ctile * TC + wco0);
      if constexpr (Tr::IS16 && C::CB_W == 2 && C::PB_W == 2 && KS == 1) {
        // quad-transposed stores (see conv_epilogue_quad) for the 1x1 convs: WDSR's 128 -> 768 expansion writes 226 MB at
        // batch 64 and was store-bound with 16-byte pieces (1.8 TB/s).  Not for 3x3: those are compute-bound and the
        // transposes cost a wave of occupancy (128 -> 166 VGPRs; EDSR-large -9 %)
        const int gxb = x0 + (px & ~3), qi = px & 3;
        const int xm = min(max(a.W - gxb, 0), 4);
        const int xmask = (1 << xm) - 1;
        const int gy0 = y0 + pyb[0], gy1 = y0 + pyb[1];
        const int pb0 = (n * a.H * d.rr + gy0 * d.rr + d.si) * (a.W * d.rr) + gxb * d.rr + d.sj;
        const int pb1 = pb0 + 2 * d.rr * (a.W * d.rr);            // pyb[1] = pyb[0] + 2
        const int cq = 32 * h + 8 * qi;
        conv_epilogue_quad<DT>(a, acc, pb0, pb1, gy0 < a.H ? xmask : 0, gy1 < a.H ? xmask : 0, d.rr, d.cbase + cq,
                               ctile * TC + wco0 + (cq & ~15) >= a.mask_from, qi);
        return;
      }
      int opix[C::PB_W];
#pragma unroll
      for (int pb = 0; pb < C::PB_W; ++pb) opix[pb] = fast_opix(a, d, n, y0 + pyb[pb], x0 + px);
      conv_epilogue_fast<DT, C::CB_W, C::PB_W>(a, acc, opix, d.cbase + 32 * h, ctile * TC + wco0 + 32 * h);
      return;
    }
  }
  conv_epilogue<DT, C::CB_W, C::PB_W>(a, acc, n, y0, x0, ctile * TC + wco0, pyb, px, h);
}


// =================================================================================================
// Weight-stationary persistent variant: 3x3, ONE 128-byte input block (Cin <= 64), 16-bit dtypes.
//
// The streaming kernel above is latency-bound on the 64->64 layers that carry EDSR-baseline / RCAN
// (a tap's 16 MFMAs are far shorter than the weight slab's load latency).  Here all 9 taps of a
// 64-channel output tile (73.7 KB) are loaded into LDS ONCE per workgroup, the workgroup is persistent
// (one per CU) and walks a contiguous range of 16x16 pixel tiles, and halo tiles are fetched by LDS-DMA
// (buffer_load ... lds: no staging registers, no wait) while the other wave group runs its MFMAs:
//     LDS = 73,728 (weights) + 2 x 41,472 (one halo image per wave group) = 156,672 B of the CU's 160 KiB.
// Out-of-image halo pixels use an out-of-range buffer offset, for which the hardware writes zeros, so no
// lane ever skips its LDS slot.  Diagnostics: SRK_NO_WS=1 (environment) routes everything to the streaming kernel;
// -DSRK_WS_ABLATE=<bits> and -DSRK_WS_STAMPS=1 (separate builds, `make stamp`) skip MFMAs / epilogue / halo DMA for
// timing ablations and write s_memtime phase stamps through `post_add` (tools/stamp_ws.py).
// =================================================================================================

struct WsCfg {
  static constexpr int NT = 512;
  static constexpr int TIN = 18, PITCH = 18;
  static constexpr int WS_BYTES = 9 * 8 * 64 * 16;     // 73,728
  static constexpr int XS_BYTES = TIN * PITCH * 128;   // 41,472
  static constexpr int LDS_BYTES = WS_BYTES + 2 * XS_BYTES;
  static constexpr int XPIECES = TIN * TIN * 8;        // 2,592
};

#ifndef SRK_WS_NOREAD
#define SRK_WS_NOREAD 0      // timing ablation (separate build only): drop 1 of 4 fragment reads per K-step
#endif
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

SRK_DEV void dma16(const void* gsrc, char* lds_wave_base) {
  // lane l writes LDS[lds_wave_base + 16*l] <- 16 bytes at its own global address
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// ---- barrier among the 4 waves of ONE wave group (the hardware barrier spans the workgroup) ----------------------------
// A monotonic arrival counter per group in LDS: every wave adds 1 and polls until the count reaches 4 x generation.
// All LDS accesses are inline asm (seen by the compiler they would be LDS reads that may alias the LDS-DMA).  The
// caller makes its own prior accesses complete first (s_waitcnt vmcnt(..) for its DMA pieces; the wave's LDS reads are
// ordered by the lgkmcnt(0) below).  Used by the FREE-running variant of conv_ws_kernel only.
SRK_DEV unsigned grp_peek(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
SRK_DEV void grp_barrier(unsigned addr, unsigned target, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) {
    const unsigned one = 1;
    asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(one) : "memory");
  }
  while (grp_peek(addr) < target) __builtin_amdgcn_s_sleep(1);
}

// The body of the weight-stationary kernel for ONE contiguous range of tiles [t0, t0 + nt) of channel tile `ctile`: called once per
// launch by conv_ws_kernel and once per (image, layer) by conv_trunk_kernel (below).  Every wave of the workgroup must call it with the
// same arguments; on return the last tile's 8 stores per lane may still be in flight.
template <int DT, int CBW, int NKS, bool FAST, bool EARLY, int EM>
SRK_DEV void conv_ws_body(const srk_conv_args& a, int tilesX, int tilesY, int ctile, int t0, int nt, unsigned x_bytes,
                          int xs_img, int xs_row, int xs_col, int wtap, const int tid) {
  // wtap: 16-byte chunks per tap in the packed weight buffer (8 for Cin = 64; 8*r*r when this launch handles one
  // 64-channel K-block of a wider reduction, a.wpk then points at that block's first chunk)
  // xs_img / xs_row / xs_col: element strides of the input between images, rows and columns of the conv-space grid
  // (plain NHWC: H*W*pitch, W*pitch, pitch; one sub-pixel lattice of a pixel-shuffled tensor: r*W*r*pitch, r*pitch)
  // dbg = SRK_WS_ABLATE (compile-time timing ablations, results are wrong): 1 = skip MFMAs, 2 = skip epilogue, 4 = skip halo DMA
  // 8 waves = two groups of 4.  A group owns every other tile of the workgroup's range and ONE halo buffer; it
  // alternates an MFMA phase (144 MFMAs per wave, LDS reads only) with an epilogue phase (issue the DMA of its
  // next halo tile, then convert/store the finished tile).  The groups run one phase apart, so each SIMD always
  // holds one wave in its matrix phase and one in its memory/VALU phase; one workgroup barrier per phase.
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  typedef WsCfg C;
  constexpr int CH = Tr::CH;   // 8
  constexpr int TCW = CBW * 32;                            // output channels per workgroup tile
  constexpr int NSTEP = 9 * NKS;                           // K-steps per tile
  constexpr int WPIECES = 9 * 2 * NKS * TCW;               // 16-byte pieces of the weight slab
  constexpr int GT = 256;                                  // threads per group
  constexpr int NPK = (C::XPIECES + GT - 1) / GT;          // 11 halo pieces per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wl = smem;
  constexpr int dbg = SRK_WS_ABLATE;      // compile-time: a run-time knob costs registers and branches in the hot loops
#if SRK_WS_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
  // every workgroup: entry / exit on the constant 100 MHz clock (comparable across XCDs), [256 + (launch & 1) * 1024 + 2 * block + {0, 1}];
  // the launch index from an arrival counter at [255]
  unsigned long long* const wgst = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.post_add));
  unsigned long long* wg_slot = nullptr;
  if (threadIdx.x == 0 && gridDim.x <= 512) {
    const unsigned long long c = atomicAdd(wgst + 255, 1ull);
    wg_slot = wgst + 256 + ((c / gridDim.x) & 1) * 1024 + 2 * blockIdx.x;
    wg_slot[0] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int gtid = tid & (GT - 1);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const elem* const xg = reinterpret_cast<const elem*>(a.x);
  const elem* const wg = reinterpret_cast<const elem*>(a.wpk);
  char* const Xg = smem + WPIECES * 16 + grp * C::XS_BYTES;

  const int nj = (nt - grp + 1) >> 1;          // tiles of this group: t0 + 2j + grp
  const int nph = 2 * ((nt + 1) >> 1) + 1;     // phases (group 0 has the most tiles)

  // ---- halo-tile DMA: everything that does not depend on the tile is computed ONCE per lane ------------------
  // every LDS-DMA of this kernel is issued from inline asm (dma16_hidden, srk_common.h): through the builtins hipcc
  // treats it as a store that may alias all later LDS reads and opens every MFMA phase with `s_waitcnt vmcnt(0)`,
  // i.e. waits for the 8 tile stores the epilogue deliberately left in flight.  Ordering is explicit: the counted
  // vmcnt wait that ends each epilogue phase covers the (older) DMA pieces, the barrier publishes them.
  // (Measured: -2 % on the plain variant; the prefetch variant gets 9 % SLOWER with it -- its epilogue phase goes back to
  // 10k cycles for a reason the ISA does not show -- so EARLY keeps the builtin.)
  const i32x4 xrsrc = make_rsrc4(xg, x_bytes), wrsrc = make_rsrc4(wg, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t xrsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<elem*>(xg), 0, x_bytes, 0x00020000);
  const unsigned xg_lds = lds_addr_of(Xg), wl_lds = lds_addr_of(Wl);
  // (filled by fill_pieces() BEHIND the first DMA requests of the prologue: 11 divisions by 18 per lane that nothing before
  // the second halo tile needs -- they used to be ~2k cycles in front of the first request)
  int pconst[NPK], pyx[NPK];
  auto fill_pieces = [&]() {
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      const int i = gtid + k * GT;
      const int sl = i & 7, p = i >> 3;
      const int iy = p / C::TIN, ix = p - iy * C::TIN;
      const int c = sl ^ swz(ix);
      pconst[k] = ((iy - 1) * xs_row + (ix - 1) * xs_col + a.x_coff + c * CH) * (int)sizeof(elem);
      pyx[k] = (i < C::XPIECES && c < 2 * NKS) ? (((iy - 1) & 0xffff) | ((ix - 1) << 16)) : (int)0x7fff7fff;   // never valid
    }
  };
  auto tile_of = [&](int j, int& n, int& y0, int& x0) {
    const int pt = t0 + 2 * j + grp;
    const int tX = pt % tilesX;
    const int q = pt / tilesX;
    const int tY = q % tilesY;
    n = q / tilesY;
    y0 = tY * 16;
    x0 = tX * 16;
  };
  auto dma_x = [&](int j) {
    int n, y0, x0;
    tile_of(j, n, y0, x0);
    const int tbase = (n * xs_img + y0 * xs_row + x0 * xs_col) * (int)sizeof(elem);   // wave-uniform (SALU)
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      const int gy = y0 + (int)(short)(pyx[k] & 0xffff), gx = x0 + (pyx[k] >> 16);
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned voff = ok ? (unsigned)(tbase + pconst[k]) : 0x80000000u;
      if (k < NPK - 1 || gtid + k * GT < C::XPIECES)
        if constexpr (EARLY) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc_b, (lds_void*)(Xg + ((k * GT + w4 * 64) << 4)), 16, voff, 0, 0, 0);
        else dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(xg_lds + ((k * GT + w4 * 64) << 4))));
    }
  };

  // bias as the C operand of the first MFMA of every tile (no accumulator initialisation pass); loaded FIRST so
  // that the counted wait of group 1 below covers it.  EARLY: the 16*CBW bias registers are needed for the prefetched
  // residual / mask pieces, so the bias is parked in LDS behind the halo buffers and re-read per tile (8 broadcast
  // ds_read_b128 per lane)
  static_assert(!EARLY || (FAST && CBW == 2), "EARLY is a variant of the quad epilogue");
  static_assert(EARLY ? (EM >= 1 && EM <= 3) : EM == 0, "EM: what the prefetch variant prefetches (1 = residual, 2 = ReLU-backward mask, 3 = its sign bits)");
  f32x16 bias16[EARLY ? 1 : CBW];
  float* const Bl = reinterpret_cast<float*>(smem + WPIECES * 16 + 2 * C::XS_BYTES);
  float bias_lane = 0.f;                                   // EARLY: this lane's bias value on its way to LDS
  auto load_bias = [&]() {                                 // requested behind the prologue's weight DMA, before a group's halo pieces
    if constexpr (EARLY) {
      if (tid < TCW && a.bias) bias_lane = a.bias[ctile * TCW + tid];
    } else {
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ctile * TCW + 4 * h + cb * 32 + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
          bias16[cb][4 * i + 0] = b.x; bias16[cb][4 * i + 1] = b.y; bias16[cb][4 * i + 2] = b.z; bias16[cb][4 * i + 3] = b.w;
        }
    }
  };
  // ---- prologue: all 9 taps of this channel tile (both groups), then each group's first halo tile ---------------
  // every workgroup needs the SAME weight bytes at the same moment: walking them in the same order would send all
  // CUs of an XCD to one L2 channel at a time, so each workgroup starts at its own 1-KB block (SRK_WS_ABLATE bit 16: off;
  // measured: no difference, the burst is bandwidth-bound)
#if SRK_WS_STAMPS
  const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
  static_assert(WPIECES % 64 == 0, "weight slab is a whole number of 1-KB blocks");
  constexpr int NBLK = WPIECES / 64;
  // STAGED prologue (the plain 64-channel variant).  Stamps: of the ~9k cycles between kernel entry and the first MFMA,
  // 2k are per-lane set-up, 3.6k are spent ISSUING the 15 DMA pieces of a wave (all 8 waves at once: the CU accepts
  // ~34 B/clk), 0.7k landing, 1.4k barrier skew.  So only the bytes the first K-steps need are issued before the first
  // barrier -- group 0's halo tile, loaded by all 8 waves together, and taps 0-2 (66 of 115 KB) -- and group 1, idle
  // through phase 0, fetches taps 3-8 behind it; group 0 meets it at a workgroup barrier just before the first
  // fragment of tap 3 / tap 6 is read (K-steps 10 / 22), then group 1 loads its own first halo tile.
  constexpr bool STAGED = !EARLY && FAST && CBW == 2 && NKS == 4;
  const int rot = (STAGED || (dbg & 16)) ? 0 : (int)((blockIdx.x * 11u) % (unsigned)NBLK);
  if constexpr (STAGED) {
    const int pt0 = t0;                                   // group 0's first tile
    const int tX = pt0 % tilesX, q0 = pt0 / tilesX, tY = q0 % tilesY, n0 = q0 / tilesY;
    const int y0 = tY * 16, x0 = tX * 16;
    const int tbase = (n0 * xs_img + y0 * xs_row + x0 * xs_col) * (int)sizeof(elem);
    const unsigned x0_lds = lds_addr_of(smem + WPIECES * 16);
#pragma unroll
    for (int k = 0; k < (C::XPIECES + 511) / 512; ++k) {
      const int i = tid + k * 512;
      if (k * 512 + wave * 64 < C::XPIECES) {             // wave-uniform
        const int sl = i & 7, p = i >> 3;
        const int iy = p / C::TIN, ix = p - iy * C::TIN;
        const int c = sl ^ swz(ix);
        const int gy = y0 + iy - 1, gx = x0 + ix - 1;
        const bool ok = i < C::XPIECES && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(tbase + ((iy - 1) * xs_row + (ix - 1) * xs_col + a.x_coff + c * CH) * (int)sizeof(elem)) : 0x80000000u;
        if (i < C::XPIECES)
          dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(x0_lds + ((k * 512 + wave * 64) << 4))));
      }
    }
  }
  auto dma_wblock = [&](int blk) {                 // one 1-KB block (64 pieces) of the weight slab, by this wave
    int rb = blk + rot;
    if (rb >= NBLK) rb -= NBLK;
    const int i = rb * 64 + lane;             // i = (tap*2*NKS + c)*TCW + co
    const int co = i % TCW, c = (i / TCW) % (2 * NKS), tap = i / (TCW * 2 * NKS);
    const size_t off = ((size_t)(tap * wtap + c) * a.CoutP + ctile * TCW + co) * CH;
    if constexpr (EARLY) dma16(wg + off, Wl + (rb << 10));
    else dma16_hidden(wrsrc, (unsigned)(off * sizeof(elem)), (unsigned)__builtin_amdgcn_readfirstlane((int)(wl_lds + (rb << 10))));
  };
  if constexpr (STAGED) {
#pragma unroll
    for (int k = 0; k < 3; ++k) dma_wblock(k * 8 + wave);          // taps 0-2, all 8 waves; taps 3-8: group 1, below
  } else {
#pragma unroll 1
    for (int k = 0; k < (NBLK + 7) / 8; ++k) {
      const int blk = k * 8 + wave;
      if (blk < NBLK) dma_wblock(blk);
    }
  }
  __builtin_amdgcn_sched_barrier(0);                        // the requests above first, the set-up below behind them
  fill_pieces();
  load_bias();
  asm volatile("" ::: "memory");
  if (!STAGED && nj > 0) dma_x(0);
#if SRK_WS_STAMPS
  const unsigned long long tB = __builtin_amdgcn_s_memtime();
#endif

  const int px = r & 15;
  int pyb[2];
  pyb[0] = w4 * 4 + (r >> 4);
  pyb[1] = w4 * 4 + 2 + (r >> 4);
  int gsw[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) gsw[kw] = swz(px + kw);
  const char* const wlane = Wl + ((h * TCW + r) << 4);
  const char* const xl0 = Xg + ((pyb[0] * C::PITCH + px) << 7);
  const char* const xl1 = Xg + ((pyb[1] * C::PITCH + px) << 7);

  const FastDst fdst = fast_dst(a, ctile * TCW);
  // tail-conv case (see conv_epilogue_planar4): wave-uniform, post_add in (scalar) registers for the whole kernel
  const bool planar4 = !FAST && CBW == 1 && a.out_mode == SRK_OUT_PLANAR && a.ps_r <= 1 && a.Cout <= 4 &&
                       (long long)a.N * a.Cout * H * W < (1ll << 29);
  float pa4[4] = {0.f, 0.f, 0.f, 0.f};
  if (planar4 && a.post_add && !SRK_WS_STAMPS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pa4[e] = e < a.Cout ? a.post_add[e] : 0.f;
  }

  // FREE (Cin = 16, i.e. 36 MFMAs per tile against a ~4k-cycle epilogue): the two groups do not alternate -- in lock
  // step only one of them would be storing at any time -- but run free of each other, each with a 4-wave barrier on an
  // LDS arrival counter after its phases.  (For the 64-channel layers free-running loses: the groups fall into step,
  // and an enforced alternation costs what the workgroup barrier costs)
  constexpr bool FREE = NKS == 1 || EARLY;
  unsigned* const sync_ctr = reinterpret_cast<unsigned*>(smem + WPIECES * 16 + 2 * C::XS_BYTES + TCW * 4);
  if (FREE && tid < 2) sync_ctr[tid] = 0;
  const unsigned my_ctr = lds_addr_of(sync_ctr + grp);
  unsigned gen = 0;

  f32x16 acc[CBW][2];
  // group 0 needs the weights and its halo tile now; group 1 idles through phase 0, so it only has to have landed its
  // share of the weights here (its 10-11 halo pieces are the youngest operations) and waits for its halo tile at the
  // end of phase 0: 41.5 KB less in the all-CUs-at-once prologue burst (~12-14 B/clk/CU)
  if constexpr (STAGED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of group 0's halo tile and of taps 0-2
  } else {
    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  }
#if SRK_WS_STAMPS
  const unsigned long long tC = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (EARLY) {                                   // (lanes 0..63 = wave 0 of group 0: its vmcnt(0) above covers the load)
    if (tid < TCW) Bl[tid] = bias_lane;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
#if SRK_WS_STAMPS
  const unsigned long long tD = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (STAGED) {
    // group 1 (idle through phase 0) fetches taps 3-8: 12 blocks per wave, in tap order
    if (grp == 1) {
#pragma unroll
      for (int kk = 0; kk < 12; ++kk) dma_wblock(24 + kk * 4 + w4);
    }
  }
  if constexpr (!EARLY) {
    // make the bias loads complete HERE: left pending into the tile loop, their first use (the first MFMA of a tile)
    // carries an `s_waitcnt vmcnt(0)` that would then run for every tile and wait for the epilogue's stores
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) asm volatile("" : "+v"(bias16[cb]));
  }

  // diagnostic stamps (separate -DSRK_WS_STAMPS=1 build, `make stamp` + tools/stamp_ws.py): s_memtime at the start and
  // end of each phase body of workgroup 0, written through the post_add pointer
#if SRK_WS_STAMPS
  unsigned long long* const stamp = blockIdx.x == 0 && (tid & 255) == 0
                                        ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.post_add)) + grp * 128 : nullptr;
#define SRK_STAMP(i) do { if (stamp) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SRK_STAMP(i) do { } while (0)
#endif
  // EARLY: geometry of this lane's quad for tile j, and the prefetched residual / mask pieces of the tile in flight
  const int qi = px & 3;
  const int cq = 32 * h + 8 * qi;
  auto quad_geo = [&](int j) {
    int n, y0, x0;
    tile_of(j, n, y0, x0);
    const int gxb = x0 + (px & ~3);
    const int xm = min(max(W - gxb, 0), 4);
    const int xmask = (1 << xm) - 1;
    const int gy0 = y0 + pyb[0], gy1 = y0 + pyb[1];
    QuadGeo g;
    g.pbase[0] = (n * H * fdst.rr + gy0 * fdst.rr + fdst.si) * (W * fdst.rr) + gxb * fdst.rr + fdst.sj;
    g.pbase[1] = g.pbase[0] + 2 * fdst.rr * (W * fdst.rr);            // pyb[1] = pyb[0] + 2
    g.okmask[0] = gy0 < H ? xmask : 0;
    g.okmask[1] = gy1 < H ? xmask : 0;
    g.pstep = fdst.rr;
    g.cl = fdst.cbase + cq;
    return g;
  };
  QuadRegs e0, e1;
  if (FREE && grp == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // its first halo tile
    grp_barrier(my_ctr, 4 * ++gen, lane);
    // start one MFMA phase late: with the epilogue the longer phase, groups that start out of phase stay out of phase
    while (grp_peek(lds_addr_of(sync_ctr)) < 4) __builtin_amdgcn_s_sleep(1);
  }
  const int nloop = FREE ? 2 * nj : nph;
#pragma unroll 1
  for (int p = 0; p < nloop; ++p) {
    const int q = FREE ? p : p - grp;            // this group's own phase counter
    const int j = q >> 1;
    SRK_STAMP(2 * p);
    if (q >= 0 && j < nj) {
      if ((q & 1) == 0) {
        if (dbg & 1) {
#pragma unroll
          for (int cb = 0; cb < CBW; ++cb) { acc[cb][0] = bias16[EARLY ? 0 : cb]; acc[cb][1] = bias16[EARLY ? 0 : cb]; }
        } else {
        // ---------------- MFMA phase: 9*NKS K-steps, fragments fetched two steps ahead ---------------------------
        // EARLY: the 12 swizzled operand addresses are rebuilt per tile (opaque copy of h): hoisted out of the tile loop
        // they would stay live through the epilogue phase, which needs the registers
        int hx = h;
        if constexpr (EARLY) {
          asm volatile("" : "+v"(hx));
        }
        // fragment piece q of K-step s: q < CBW -> weights of channel block q, q = CBW / CBW+1 -> pixel block 0 / 1
        auto frag1 = [&](int s, int q, i32x4 (&af)[CBW], i32x4& b0, i32x4& b1) {
          const int tap = s / NKS, ks = s - tap * NKS;
          const int kh = tap / 3, kw = tap - kh * 3;
          const int kc2 = 2 * ks;
          if (q < CBW) {
            af[q] = lds_read16(wlane + (((tap * 2 * NKS + kc2) * TCW + q * 32) << 4));
          } else {
            const int so = (((kc2 + hx) ^ gsw[kw]) << 4) + ((kh * C::PITCH + kw) << 7);
            if (q == CBW) b0 = lds_read16(xl0 + so);
            else b1 = lds_read16(xl1 + so);
          }
        };
        auto frag = [&](int s, i32x4 (&af)[CBW], i32x4& b0, i32x4& b1) {
#pragma unroll
          for (int q = 0; q < CBW + 2; ++q) frag1(s, q, af, b0, b1);
        };
        i32x4 fa[3][CBW], fb0[3], fb1[3];
        if constexpr (EARLY) {
#pragma unroll
          for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(Bl + 4 * h + cb * 32 + 8 * i);
              acc[cb][0][4 * i + 0] = b.x; acc[cb][0][4 * i + 1] = b.y; acc[cb][0][4 * i + 2] = b.z; acc[cb][0][4 * i + 3] = b.w;
              acc[cb][1][4 * i + 0] = b.x; acc[cb][1][4 * i + 1] = b.y; acc[cb][1][4 * i + 2] = b.z; acc[cb][1][4 * i + 3] = b.w;
            }
        }
        frag(0, fa[0], fb0[0], fb1[0]);
        frag(1, fa[1], fb0[1], fb1[1]);
        QuadPre pre;
        if constexpr (EARLY) pre = EM == 3 ? quad_early_setup_bits(a, quad_geo(j), qi, h) : quad_early_setup(a, quad_geo(j), qi);     // behind the first LDS reads: overlaps their latency
        __builtin_amdgcn_s_setprio(1);
#if SRK_WS_STAMPS
        if (p == 2) SRK_STAMP(40);
#endif
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#if SRK_WS_STAMPS
          if (p == 2 && s == 1) SRK_STAMP(41);
          if (p == 2 && s == NSTEP - 1) SRK_STAMP(42);
#endif
          const int c0 = s % 3, c2 = (s + 2) % 3;
          // ONE LDS read per MFMA gap (four waves x one ds_read_b128 = 16 of the gap's 32 LDS-array cycles); a burst of
          // 4 reads per wave in one gap oversubscribes the array while the waves run in step
          if constexpr (STAGED) {
            // first tile only: taps 3-5 / 6-8 are needed from K-steps 12 / 24 on, their fragments two steps earlier
            if (p == 0 && (s == 10 || s == 22)) __builtin_amdgcn_s_barrier();      // group 1 has landed taps 3-5 / 6-8
          }
          int q = 0;
          if constexpr (EARLY) {
            // one prefetch piece every 4th K-step (every step when the loop is short): a 16-byte-per-line load keeps the address path busy for ~64 cycles,
            // and eight of them back to back stall the wave's (in-order) issue for ~1k cycles with the matrix pipe idle
            constexpr int PST = NSTEP >= 33 ? 4 : 1;       // all 8 pieces must fit: steps 1, 1+PST, ..., 1+7*PST < NSTEP
            static_assert(1 + 7 * PST < NSTEP, "prefetch schedule does not fit the K loop");
            if constexpr (EM == 3) {
              if ((s == 1 || s == 2) && !(dbg & 2)) quad_early_word(pre, s - 1, s == 1 ? e0 : e1);      // two 4-byte loads per tile
            } else if (s >= 1 && (s - 1) % PST == 0 && (s - 1) / PST < 8 && !(dbg & 2)) {
              const int pi = (s - 1) / PST;
              quad_early_piece(pre, pi >> 2, pi & 3, pi < 4 ? e0 : e1);
            }
          }
#pragma unroll
          for (int cb = 0; cb < CBW; ++cb) {
            if (s + 2 < NSTEP && q < CBW + 2 && !SRK_WS_NOREAD) { frag1(s + 2, q, fa[c2], fb0[c2], fb1[c2]); ++q; }
            acc[cb][0] = Tr::mma(fa[c0][cb], fb0[c0], (!EARLY && s == 0) ? bias16[EARLY ? 0 : cb] : acc[cb][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < NSTEP && q < CBW + 2) { frag1(s + 2, q, fa[c2], fb0[c2], fb1[c2]); ++q; }
            if (cb == CBW - 1)
              for (; s + 2 < NSTEP && q < CBW + 2; ++q) frag1(s + 2, q, fa[c2], fb0[c2], fb1[c2]);
            acc[cb][1] = Tr::mma(fa[c0][cb], fb1[c0], (!EARLY && s == 0) ? bias16[EARLY ? 0 : cb] : acc[cb][1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
#if SRK_WS_STAMPS
        if (p == 2) SRK_STAMP(43);
#endif
        // the pieces requested 5k cycles ago are the wave's only outstanding vector-memory operations: make them
        // architecturally complete HERE, in the straight-line code that issued them.  (Waiting in the epilogue phase
        // instead lets the compiler place its loop-carried register copies of e0 / e1 in front of the wait.)
        if constexpr (EM == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(e0.r[0].x), "+v"(e1.r[0].x) : : "memory");
        else if constexpr (EARLY) quad_early_wait<0>(e0, e1);
        }
      } else {
        // ---------------- epilogue phase: next halo tile's DMA first (this group's MFMAs on the buffer are done),
        // then the finished tile; the counted wait leaves the 16 stores per lane in flight, the DMA is older ----
        if constexpr (EARLY) {
          const bool more = j + 1 < nj;
          if (more && !(dbg & 4)) dma_x(j + 1);
          if (p == 3) SRK_STAMP(30);
          if (!(dbg & 2)) {
            const QuadGeo geo = quad_geo(j);
            const bool um_lo = ctile * TCW + 32 * h >= a.mask_from, um_hi = ctile * TCW + 32 * h + 16 >= a.mask_from;
            QuadRegs o0, o1;
            quad_compute<DT, EM>(a, acc, 0, e0, um_lo, um_hi, o0);
#if SRK_WS_STAMPS
            asm volatile("" :: "v"(o0.r[0].x), "v"(o0.r[3].w));
            if (p == 3) SRK_STAMP(31);
#endif
            quad_store(a, geo, 0, o0);
            if (p == 3) SRK_STAMP(32);
            quad_compute<DT, EM>(a, acc, 1, e1, um_lo, um_hi, o1);
#if SRK_WS_STAMPS
            asm volatile("" :: "v"(o1.r[0].x), "v"(o1.r[3].w));
            if (p == 3) SRK_STAMP(33);
#endif
            quad_store(a, geo, 1, o1);
            if (p == 3) SRK_STAMP(34);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the 8 dwordx4 stores per lane stay in flight
            if (p == 3) SRK_STAMP(35);
          } else {
            asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[0][1][5]), "v"(acc[CBW - 1][0][9]), "v"(acc[CBW - 1][1][15]));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        } else {
        if (j + 1 < nj && !(dbg & 4)) dma_x(j + 1);
        if (!(dbg & 2)) {
          int n, y0, x0;
          tile_of(j, n, y0, x0);
          if constexpr (FAST && CBW == 2) {
            // quad-transposed stores: lane (quad q, position i) ends up with piece i (8 channels) of the quad's pixels
            const int gxb = x0 + (px & ~3), qi = px & 3;
            const int xm = min(max(W - gxb, 0), 4);
            const int xmask = (1 << xm) - 1;
            const int gy0 = y0 + pyb[0], gy1 = y0 + pyb[1];
            const int pb0 = (n * H * fdst.rr + gy0 * fdst.rr + fdst.si) * (W * fdst.rr) + gxb * fdst.rr + fdst.sj;
            const int pb1 = pb0 + 2 * fdst.rr * (W * fdst.rr);            // pyb[1] = pyb[0] + 2
            const int cq = 32 * h + 8 * qi;
            conv_epilogue_quad<DT>(a, acc, pb0, pb1, gy0 < H ? xmask : 0, gy1 < H ? xmask : 0, fdst.rr, fdst.cbase + cq,
                                   ctile * TCW + (cq & ~15) >= a.mask_from, qi, h);
          } else if constexpr (FAST) {
            int opix[2];
            opix[0] = fast_opix(a, fdst, n, y0 + pyb[0], x0 + px);
            opix[1] = fast_opix(a, fdst, n, y0 + pyb[1], x0 + px);
            conv_epilogue_fast<DT, CBW, 2>(a, acc, opix, fdst.cbase + 32 * h, ctile * TCW + 32 * h);
          } else if constexpr (CBW == 1) {
            if (planar4) conv_epilogue_planar4<DT, 2>(a, acc, n, y0, x0, pyb, px, h, pa4);
            else conv_epilogue<DT, CBW, 2>(a, acc, n, y0, x0, ctile * TCW, pyb, px, h);
          } else {
            conv_epilogue<DT, CBW, 2>(a, acc, n, y0, x0, ctile * TCW, pyb, px, h);
          }
          if constexpr (FAST && CBW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // 8 dwordx4 stores per lane
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[0][1][5]), "v"(acc[CBW - 1][0][9]), "v"(acc[CBW - 1][1][15]));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        }
      }
    }
    if constexpr (STAGED) {
      if (q < 0) {                                        // group 1, phase 0: the two staged weight syncs of group 0
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // its 6 blocks of taps 3-5 (taps 6-8 are the 6 younger ones)
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // taps 6-8
        __builtin_amdgcn_s_barrier();
        if (nj > 0) dma_x(0);                                 // its own first halo tile, behind everything group 0 waits for
      }
    }
    if (!FREE && q < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group 1, phase 0: its first halo tile
    SRK_STAMP(2 * p + 1);
    if constexpr (FREE) grp_barrier(my_ctr, 4 * ++gen, lane);
    else __builtin_amdgcn_s_barrier();
  }
#if SRK_WS_STAMPS
  if (wg_slot) wg_slot[1] = __builtin_amdgcn_s_memrealtime();
  if (stamp) {                       // [100] entry of the FIRST stamped launch, [101] entry / [102] exit of the last one
    if (stamp[100] == 0) stamp[100] = t_entry;
    stamp[101] = t_entry;
    stamp[102] = __builtin_amdgcn_s_memtime();
    stamp[104] = tA; stamp[105] = tB; stamp[106] = tC; stamp[107] = tD;
  }
#endif
}

template <int DT, int CBW, int NKS, bool FAST, bool EARLY, int EM>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const srk_conv_args a, int tilesX, int tilesY, int ctiles,
                                                          int nptiles, unsigned x_bytes, int tq, int trem, int /*unused*/,
                                                          int xs_img, int xs_row, int xs_col, int wtap) {
  // (tq, trem) = divmod(nptiles, slots) from the host: slot s owns tq tiles, +1 for the first trem slots
  const unsigned slot = blockIdx.x / (unsigned)ctiles;
  const int ctile = (int)(blockIdx.x - slot * (unsigned)ctiles);
  const int t0 = (int)slot * tq + min((int)slot, trem);
  const int nt = tq + ((int)slot < trem ? 1 : 0);
  if (nt <= 0) return;
  conv_ws_body<DT, CBW, NKS, FAST, EARLY, EM>(a, tilesX, tilesY, ctile, t0, nt, x_bytes, xs_img, xs_row, xs_col, wtap, (int)threadIdx.x);
}

// ---- image-stationary persistent trunk: a CHAIN of 3x3 64 -> 64 convolutions in ONE launch (srk_conv_trunk, include/srk.h) -------------
// A 'same' convolution of one image needs that image only, so a workgroup that owns an image can run layer l + 1 on it as soon as it
// has stored layer l itself: no launch boundary (entry ramp 3.2 us, 3.8 us between the last exit and the next entry, every workgroup
// waiting for the launch's slowest one: DESIGN.md 3.17) and no synchronisation with any other workgroup.  Workgroup b walks the images
// b, b + grid, ...; for each image the layers of the table in order, each with the weight-stationary body above (a layer is what one
// conv_ws_kernel launch does for that image's tiles: same code, same results, bit for bit).  Between two layers: the workgroup's own
// stores complete (vmcnt(0); they are write-through, sc1) and one workgroup barrier -- the workgroup-scope release / acquire of the gfx942+
// memory model: an image's bytes are written and read by ONE CU, whose vector L1 is coherent with its own stores.  (No `buffer_inv sc1`:
// at agent scope it also drops the XCD's L2 lines, 33 times per image: 46.9 instead of 41.2 us per convolution.)  The table lives in
// device memory; a layer's arguments are scalar loads.  A layer with KH = 0 is `out = x + res` on the image (the long skip's gradient).
// (Measured and removed: the wave group that is idle during a layer's last phase fetching the NEXT layer's weight slab -- 42.5 against 42.4 us per
// convolution, 63 more spilled SGPRs: tools/ubench/trunk_weight_prefetch.patch, profiles/r6_experiments.txt 3.)
template <int DT>
__global__ __launch_bounds__(512, 2) void conv_trunk_kernel(const srk_conv_args* __restrict__ tab, int nlayers, int nimages, int tilesX,
                                                             int tilesY, unsigned x_bytes) {
  const int tpi = tilesX * tilesY;
#pragma unroll 1
  for (int n = (int)blockIdx.x; n < nimages; n += (int)gridDim.x) {
#pragma unroll 1
    for (int l = 0; l < nlayers; ++l) {
      const srk_conv_args a = tab[l];
      // an opaque copy of the thread index per layer: every per-lane constant of a body derives from it, so none of them is hoisted
      // out of the layer loop (hoisted, the three bodies' constants are live together: 86 registers spilled)
      int tid = (int)threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int xs_col = a.x_pitch, xs_row = a.W * a.x_pitch, xs_img = a.H * a.W * a.x_pitch;
      if (a.KH == 0) {
        // pseudo-layer `out = x + res` on this image (the long skip's gradient: the two contributions to the trunk input's gradient,
        // models/edsr.py:46-47 backward): fp32 add of the two 16-bit values, rounded once -- what torch's add does
        const int npc = a.H * a.W * 8;                                   // 16-byte pieces of the image
        const uint4* const px = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.x) + (size_t)n * npc * 16);
        const uint4* const pr = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.res) + (size_t)n * npc * 16);
        uint4* const po = reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out) + (size_t)n * npc * 16);
        for (int i = tid; i < npc; i += 512) {
          const uint4 u = px[i], v = pr[i];
          const uint32_t uu[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
          uint32_t oo[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float a0, a1, b0, b1;
            unpack2<DT>(uu[k], a0, a1);
            unpack2<DT>(vv[k], b0, b1);
            oo[k] = pack2<DT>(a0 + b0, a1 + b1);
          }
          po[i] = uint4{oo[0], oo[1], oo[2], oo[3]};
        }
      } else if (a.mask_bits) conv_ws_body<DT, 2, 4, true, true, 3>(a, tilesX, tilesY, 0, n * tpi, tpi, x_bytes, xs_img, xs_row, xs_col, 8, tid);
      else if (a.res) conv_ws_body<DT, 2, 4, true, true, 1>(a, tilesX, tilesY, 0, n * tpi, tpi, x_bytes, xs_img, xs_row, xs_col, 8, tid);
      else conv_ws_body<DT, 2, 4, true, false, 0>(a, tilesX, tilesY, 0, n * tpi, tpi, x_bytes, xs_img, xs_row, xs_col, 8, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// launcher-side preconditions of conv_epilogue_fast
static bool conv_fast_ok(const srk_conv_args& a, int esz) {
  if (a.out_mode == SRK_OUT_PLANAR) return false;
  if (a.Cout % 64 != 0) return false;
  const int al = 16 / esz;                       // 16-byte accesses
  if (a.out_pitch % al || a.out_coff % al) return false;
  if (a.res && (a.res_pitch % al || a.res_coff % al)) return false;
  if (a.mask && (a.mask_pitch % al || a.mask_coff % al || a.mask_from % 16)) return false;
  const int rr = (a.out_mode == SRK_OUT_NHWC_PS && a.ps_r > 1) ? a.ps_r : 1;
  if (rr > 1 && (a.Cout / (rr * rr)) % 64 != 0) return false;
  const long long px = (long long)a.N * a.H * a.W * rr * rr;
  long long mx = px * a.out_pitch;
  if (a.res && px * a.res_pitch > mx) mx = px * a.res_pitch;
  if (a.mask && px * a.mask_pitch > mx) mx = px * a.mask_pitch;
  return mx * esz < 0x7fff0000LL;
}

// one launch; `early` picks the variant that prefetches the residual / mask (see quad_compute)
template <int DT, int CBW, int NKS, bool FAST, bool EARLY, int EM>
static int launch_ws_one(const srk_conv_args& b, hipStream_t st, unsigned grid, int tilesX, int tilesY, int ctiles, int nptiles,
                         unsigned xb, int tq, int trem, int xs_img, int xs_row, int xs_col, int wtap) {
  typedef WsCfg C;
  constexpr int TCW = CBW * 32;
  constexpr int LDS = 9 * 2 * NKS * TCW * 16 + 2 * C::XS_BYTES + TCW * 4 + 16;      // weights, two halo buffers, bias, group-barrier counters
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<DT, CBW, NKS, FAST, EARLY, EM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d(ws): cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return (int)attr;
  }
  hipLaunchKernelGGL((conv_ws_kernel<DT, CBW, NKS, FAST, EARLY, EM>), dim3(grid), dim3(C::NT), LDS, st, b, tilesX, tilesY, ctiles, nptiles,
                     xb, tq, trem, 0, xs_img, xs_row, xs_col, wtap);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT, int CBW, int NKS, bool FAST> int launch_ws(const srk_conv_args& a, hipStream_t st) {
  constexpr int TCW = CBW * 32;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const bool no_early = srk_dbg_getenv("SRK_NO_EARLY") != nullptr;
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16, ctiles = a.CoutP / TCW;
  const long long nptiles = (long long)a.N * tilesX * tilesY;
  if (nptiles <= 0 || nptiles > 0x7fffffffLL) {
    srk_set_error("srk_conv2d(ws): bad tile count %lld", nptiles);
    return SRK_E_BADARG;
  }
  long long slots = cus / ctiles;            // workgroups per channel tile
  if (slots < 1) slots = 1;
  if (slots > nptiles) slots = nptiles;
  const unsigned grid = (unsigned)(slots * ctiles);
  const int tq = (int)(nptiles / slots), trem = (int)(nptiles % slots);
  const int rin = a.x_ps > 1 ? a.x_ps : 1;
  const unsigned xb = (unsigned)(((long long)a.N * a.H * a.W * rin * rin * a.x_pitch) * 2);
  // residual OR ReLU mask in the epilogue (not both): the variant that prefetches it during the MFMA phase
  auto one = [&](const srk_conv_args& b, int xs_img, int xs_row, int xs_col, int wtap) -> int {
    if constexpr (FAST && CBW == 2) {
      if (b.mask_bits && !b.res && !b.mask)
        return launch_ws_one<DT, CBW, NKS, FAST, true, 3>(b, st, grid, tilesX, tilesY, ctiles, (int)nptiles, xb, tq, trem, xs_img, xs_row, xs_col, wtap);
      if (((b.res != nullptr) != (b.mask != nullptr)) && !no_early) {
        if (b.res) return launch_ws_one<DT, CBW, NKS, FAST, true, 1>(b, st, grid, tilesX, tilesY, ctiles, (int)nptiles, xb, tq, trem, xs_img, xs_row, xs_col, wtap);
        return launch_ws_one<DT, CBW, NKS, FAST, true, 2>(b, st, grid, tilesX, tilesY, ctiles, (int)nptiles, xb, tq, trem, xs_img, xs_row, xs_col, wtap);
      }
    }
    return launch_ws_one<DT, CBW, NKS, FAST, false, 0>(b, st, grid, tilesX, tilesY, ctiles, (int)nptiles, xb, tq, trem, xs_img, xs_row, xs_col, wtap);
  };
  if (rin == 1) return one(a, a.H * a.W * a.x_pitch, a.W * a.x_pitch, a.x_pitch, 2 * NKS);
  // Input stored pixel-shuffled (the dgrad of a conv + PixelShuffle(r)): channel k = (i*r+j)*64 + c lives at sub-pixel
  // (i,j), so the K = r*r*64 reduction is r*r weight-stationary 64->64 convs, one per sub-pixel lattice of x, each
  // with its own 9x64x64 weight slab (K-block ij of every tap of the packed weights), accumulating into `out`:
  //   launch 0: out = scale*conv_0 + res ; launch ij>0: out = scale*conv_ij + out ; the mask runs on the last one.
  const int r2 = rin * rin;
  for (int ij = 0; ij < r2; ++ij) {
    srk_conv_args b = a;
    const int si = ij / rin, sj = ij - si * rin;
    b.x = reinterpret_cast<const char*>(a.x) + ((size_t)si * (a.W * rin) + sj) * a.x_pitch * 2;
    b.x_ps = 0;
    b.Cin = 64;
    b.wpk = reinterpret_cast<const char*>(a.wpk) + (size_t)ij * 8 * a.CoutP * 8 * 2;
    if (ij > 0) {
      b.bias = nullptr;
      b.res = a.out; b.res_pitch = a.out_pitch; b.res_coff = a.out_coff;
    }
    if (ij < r2 - 1) b.mask = nullptr;
    const int rc = one(b, a.H * rin * a.W * rin * a.x_pitch, rin * a.W * rin * a.x_pitch, rin * a.x_pitch, 8 * r2);
    if (rc != 0) return rc;
  }
  return 0;
}

template <int DT, int TC, int KS> int launch(const srk_conv_args& a, hipStream_t st) {
  typedef ConvCfg<DT, TC, KS> C;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<DT, TC, KS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d: cannot reserve %d bytes of LDS: %s", C::LDS_BYTES, hipGetErrorString(attr));
    return (int)attr;
  }
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16, ctiles = a.CoutP / TC;
  const long long nb = (long long)a.N * tilesX * tilesY * ctiles;
  if (nb <= 0 || nb > 0x7fffffffLL) {
    srk_set_error("srk_conv2d: bad grid %lld", nb);
    return SRK_E_BADARG;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<DT, TC, KS>), dim3((unsigned)nb), dim3(C::NT), C::LDS_BYTES, st, a, tilesX,
                     tilesY, ctiles, (int)conv_fast_ok(a, 16 / C::CH));
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT> int dispatch_tc(const srk_conv_args& a, hipStream_t st) {
  if constexpr (DTraits<DT>::IS16) {
    // 3x3 with ONE 128-byte input block: weights stay in LDS, persistent workgroups (conv_ws_kernel)
    const int rin = a.x_ps > 1 ? a.x_ps : 1;
    const long long xbytes = ((long long)a.N * a.H * a.W * rin * rin * a.x_pitch) * 2;
    static const bool no_ws = srk_dbg_getenv("SRK_NO_WS") != nullptr;      // diagnostics knob, read once
    if (a.KH == 3 && xbytes < 0x7fffffffLL && !no_ws) {
      if (a.CoutP % 64 == 0 && conv_fast_ok(a, 2)) {
        if (rin == 1 && a.Cin == 64) return launch_ws<DT, 2, 4, true>(a, st);
        if (rin == 1 && a.Cin == 16) return launch_ws<DT, 2, 1, true>(a, st);          // e.g. dgrad of the 3-channel tail conv
        // the data gradient of a 64-feature upsampler conv (256 gradient channels read through the PixelShuffle): ONE launch of the
        // many-input-channel kernel (conv_ks.hip: K blocks streamed, fp32 partial sums stay in registers) instead of four weight-stationary
        // passes that add their 16-bit partial results through `res` -- 4 x 10.3 us of launch latency at batch 16 (+2.0 % on the EDSR-baseline
        // step, neutral at batch 256) and three roundings of the sum less.  SRK_PS_DGRAD_WS=1 (under SRK_DEBUG=1): the four passes.
        static const bool ps_ws = [] { const char* e = srk_dbg_getenv("SRK_PS_DGRAD_WS"); return e && e[0] == '1'; }();      // A/B knob
        if (rin > 1 && a.Cin == 64 * rin * rin && !a.relu && a.out_mode == SRK_OUT_NHWC && (ps_ws || !srk_conv_ks_ok(a))) return launch_ws<DT, 2, 4, true>(a, st);
      }
      if (a.CoutP == 32 && rin == 1 && a.Cin == 64) return launch_ws<DT, 1, 4, false>(a, st);   // e.g. the 64->3 tail conv
    }
  }
  if constexpr (DTraits<DT>::IS16) {
    if (srk_conv_ks_ok(a)) return srk_conv_ks_launch(a, st);      // many input channels: conv_ks.hip
    if (srk_conv1x1_ok(a)) return srk_conv1x1_launch(a, st);      // pointwise convs: conv1x1.hip
  }
  const int tc = (a.CoutP % 128 == 0) ? 128 : (a.CoutP % 64 == 0) ? 64 : 32;
  if (a.KH == 3) {
    if (tc == 128) return launch<DT, 128, 3>(a, st);
    if (tc == 64) return launch<DT, 64, 3>(a, st);
    return launch<DT, 32, 3>(a, st);
  }
  if (tc == 128) return launch<DT, 128, 1>(a, st);
  if (tc == 64) return launch<DT, 64, 1>(a, st);
  return launch<DT, 32, 1>(a, st);
}

}  // namespace

// relu_bits / mask_bits: only the weight-stationary 64 -> 64 kernel implements them
static bool conv_bits_ok(const srk_conv_args& a) {
  static const bool no_ws = srk_dbg_getenv("SRK_NO_WS") != nullptr;
  if (no_ws || a.dtype == SRK_F32 || a.KH != 3 || a.KW != 3 || a.Cin != 64 || a.CoutP != 64 || a.Cout != 64 || a.x_ps > 1 || a.out_mode != SRK_OUT_NHWC) return false;
  // the bit words are indexed by the conv's own output pixel: no shuffled store (ps_r) and no planar-only post_add on this path
  if (a.ps_r > 1 || a.post_add) return false;
  if (!conv_fast_ok(a, 2) || ((long long)a.N * a.H * a.W * a.x_pitch) * 2 >= 0x7fffffffLL || (long long)a.N * a.H * a.W * 8 >= 0x7fffffffLL) return false;
  if (a.relu_bits && !(a.relu && !a.res && !a.mask && a.scale == 1.f)) return false;
  if (a.mask_bits && (a.res || a.mask)) return false;
  return true;
}
extern "C" int srk_conv_bits_ok(const srk_conv_args* a) { return a && conv_bits_ok(*a) ? 1 : 0; }

// one layer of a trunk table: exactly what dispatch_tc sends to launch_ws<DT, 2, 4, true> with one of the three flavours conv_trunk_kernel holds
static bool conv_trunk_layer_ok(const srk_conv_args& a, const srk_conv_args& a0) {
  if (a.dtype != a0.dtype || a.N != a0.N || a.H != a0.H || a.W != a0.W) return false;
  if (a.KH == 0)      // pseudo-layer out = x + res: dense 64-channel tensors
    return a.KW == 0 && a.x && a.res && a.out && a.x_pitch == 64 && a.res_pitch == 64 && a.out_pitch == 64 && !a.x_coff && !a.res_coff && !a.out_coff &&
           a.dtype != SRK_F32 && !a.mask && !a.mask_bits && !a.relu_bits && !a.post_add;
  if (a.dtype == SRK_F32 || a.KH != 3 || a.KW != 3 || a.Cin != 64 || a.CoutP != 64 || a.Cout != 64 || a.x_ps > 1) return false;
  if (a.out_mode != SRK_OUT_NHWC || a.ps_r > 1 || a.post_add || a.mask || a.x_coff % 8 || a.x_pitch % 8) return false;
  if (!a.x || !a.wpk || !a.out || !conv_fast_ok(a, 2)) return false;
  if (((long long)a.N * a.H * a.W * a.x_pitch) * 2 >= 0x7fffffffLL || (long long)a.N * a.H * a.W * 8 >= 0x7fffffffLL) return false;
  if (a.relu_bits && !(a.relu && !a.res && a.scale == 1.f)) return false;
  if (a.mask_bits && (a.res || a.relu_bits)) return false;
  if (a.x_pitch != a0.x_pitch) return false;                      // one buffer size (x_bytes) for the whole chain
  return true;
}
extern "C" int srk_conv_trunk_ok(const srk_conv_args* layers, int nlayers) {
  if (!layers || nlayers < 1) return 0;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const bool off = [] { const char* e = srk_dbg_getenv("SRK_NO_TRUNK"); return e && e[0] == '1'; }();      // diagnostics knob (SRK_DEBUG=1), read once
  if (off) return 0;
  // every CU one image at a time: worth it when the batch fills the chip in whole rounds (else the per-layer launches balance better)
  if (layers[0].N < cus || layers[0].N % cus != 0) return 0;
  for (int l = 0; l < nlayers; ++l)
    if (!conv_trunk_layer_ok(layers[l], layers[0])) return 0;
  return 1;
}
extern "C" int srk_conv_trunk(const srk_conv_args* layers_host, const void* table_dev, int nlayers, srk_stream_t stream) {
  SRK_CHECK_ARG(layers_host && table_dev && nlayers >= 1, "srk_conv_trunk: null table / no layers");
  SRK_CHECK_ARG(srk_conv_trunk_ok(layers_host, nlayers), "srk_conv_trunk: a layer is not a 16-bit 3x3 64 -> 64 NHWC convolution of the chain's shape, "
                "or the batch is not a multiple of the CU count (srk_conv_trunk_ok)");
  const srk_conv_args& a = layers_host[0];
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  const unsigned xb = (unsigned)(((long long)a.N * a.H * a.W * a.x_pitch) * 2);
  const unsigned grid = (unsigned)(a.N < cus ? a.N : cus);
  constexpr int LDS = 9 * 2 * 4 * 64 * 16 + 2 * WsCfg::XS_BYTES + 64 * 4 + 16;      // as launch_ws_one<.., 2, 4, ..>
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const srk_conv_args* tab = reinterpret_cast<const srk_conv_args*>(table_dev);
  if (a.dtype == SRK_BF16) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<SRK_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr != hipSuccess) { srk_set_error("srk_conv_trunk: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr)); return (int)attr; }
    hipLaunchKernelGGL((conv_trunk_kernel<SRK_BF16>), dim3(grid), dim3(WsCfg::NT), LDS, st, tab, nlayers, a.N, tilesX, tilesY, xb);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<SRK_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr != hipSuccess) { srk_set_error("srk_conv_trunk: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr)); return (int)attr; }
    hipLaunchKernelGGL((conv_trunk_kernel<SRK_F16>), dim3(grid), dim3(WsCfg::NT), LDS, st, tab, nlayers, a.N, tilesX, tilesY, xb);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_conv_tile(int Cout) {
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  const int up = (Cout + 127) / 128 * 128;
  return (up - Cout < 64) ? 128 : 64;
}

extern "C" int srk_conv2d(const srk_conv_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->wpk && a->out, "srk_conv2d: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "srk_conv2d: bad dims N=%d H=%d W=%d", a->N, a->H, a->W);
  const bool lk = srk_conv_lk_ok(*a);                   // direct 5x5 / 7x7 / 9x9 conv (conv_lk.hip)
  SRK_CHECK_ARG(a->KH == a->KW && (a->KH == 1 || a->KH == 3 || lk), "srk_conv2d: kernel %dx%d not supported for this shape (1x1, 3x3; "
                "5x5 .. 9x9 with 64 -> <= 32 or 16 -> 64 channels, 16-bit)", a->KH, a->KW);
  SRK_CHECK_ARG(a->Cin > 0 && a->Cin % 16 == 0, "srk_conv2d: Cin=%d must be a positive multiple of 16", a->Cin);
  SRK_CHECK_ARG(a->CoutP > 0 && a->CoutP % 32 == 0, "srk_conv2d: CoutP=%d must be a multiple of 32", a->CoutP);
  SRK_CHECK_ARG(a->Cout > 0 && a->Cout <= a->CoutP, "srk_conv2d: Cout=%d CoutP=%d", a->Cout, a->CoutP);
  SRK_CHECK_ARG(a->dtype >= SRK_BF16 && a->dtype <= SRK_F32, "srk_conv2d: dtype %d", a->dtype);
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->x_pitch % ch == 0 && a->x_coff % ch == 0, "srk_conv2d: x pitch/offset must be 16-byte aligned");
  const int rin = a->x_ps > 1 ? a->x_ps : 1;
  SRK_CHECK_ARG(a->Cin % (rin * rin) == 0 && (a->Cin / (rin * rin)) % ch == 0, "srk_conv2d: x_ps=%d incompatible with Cin=%d", a->x_ps, a->Cin);
  const int rr = a->ps_r > 1 ? a->ps_r : 1;
  if (a->out_mode == SRK_OUT_NHWC) {
    SRK_CHECK_ARG(a->Cout % 4 == 0 && a->out_pitch % 4 == 0 && a->out_coff % 4 == 0, "srk_conv2d: NHWC store needs 4-channel alignment");
  } else if (a->out_mode == SRK_OUT_NHWC_PS) {
    SRK_CHECK_ARG(rr > 1 && a->Cout % (rr * rr) == 0 && (a->Cout / (rr * rr)) % 4 == 0 && a->out_pitch % 4 == 0 && a->out_coff % 4 == 0,
                  "srk_conv2d: pixel-shuffle store needs Cout %% r^2 == 0 and 4-channel alignment");
  } else if (a->out_mode == SRK_OUT_PLANAR) {
    SRK_CHECK_ARG(a->Cout % (rr * rr) == 0 && a->mask == nullptr, "srk_conv2d: planar store: Cout %% r^2, no mask");
  } else {
    SRK_CHECK_ARG(false, "srk_conv2d: out_mode %d", a->out_mode);
  }
  if (a->res && a->out_mode != SRK_OUT_PLANAR)
    SRK_CHECK_ARG(a->res_pitch % 4 == 0 && a->res_coff % 4 == 0, "srk_conv2d: residual alignment");
  if (a->mask) SRK_CHECK_ARG(a->mask_pitch % 4 == 0 && a->mask_coff % 4 == 0 && a->mask_from % 4 == 0, "srk_conv2d: mask alignment");
  if (a->relu_bits || a->mask_bits)
    SRK_CHECK_ARG(!lk && conv_bits_ok(*a), "srk_conv2d: relu_bits / mask_bits are implemented by the weight-stationary 3x3 64 -> 64 kernel only (srk_conv_bits_ok)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (lk) return srk_conv_lk_launch(*a, st);
  switch (a->dtype) {
    case SRK_BF16: return dispatch_tc<SRK_BF16>(*a, st);
    case SRK_F16: return dispatch_tc<SRK_F16>(*a, st);
    default: return dispatch_tc<SRK_F32>(*a, st);
  }
}
