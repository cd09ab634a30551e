// Device-side helpers shared by the gfx950 kernels (wave64, MFMA 32x32, 16-byte LDS chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/srk.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;      // one 16-byte chunk
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;

#define SRK_DEV __device__ __forceinline__

// thread-local error text (host)
void srk_set_error(const char* fmt, ...);
// A/B switches of tools/ (SRK_NO_WS, SRK_NO_EARLY, SRK_NO_LK5, ...): honoured only under SRK_DEBUG=1 (a stray variable in a user's
// environment must not route the product path through a diagnostic form)
#include <stdlib.h>
#include <string.h>
static inline const char* srk_dbg_getenv(const char* name) {
  const char* d = getenv("SRK_DEBUG");
  return (d && d[0] == '1') ? getenv(name) : nullptr;
}
#define SRK_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      srk_set_error(__VA_ARGS__);           \
      return SRK_E_BADARG;                  \
    }                                       \
  } while (0)
#define SRK_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) {                                  \
      srk_set_error("%s: %s", __func__, hipGetErrorString(e_)); \
      return (int)e_;                                        \
    }                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// dtype traits.  CH = elements per 16-byte chunk.  A "chunk" is the unit of the K dimension:
// lane half h of a 32x32 MFMA consumes chunk (2*kstep + h) of both operands.
// ---------------------------------------------------------------------------------------------
template <int DT> struct DTraits;

template <> struct DTraits<SRK_BF16> {
  typedef uint16_t elem;
  static constexpr int CH = 8;
  static constexpr bool IS16 = true;
  static SRK_DEV float to_f32(elem v) { return __uint_as_float(((uint32_t)v) << 16); }
  static SRK_DEV elem from_f32(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
  }
  static SRK_DEV f32x16 mma(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

template <> struct DTraits<SRK_F16> {
  typedef uint16_t elem;
  static constexpr int CH = 8;
  static constexpr bool IS16 = true;
  static SRK_DEV float to_f32(elem v) { return (float)__builtin_bit_cast(_Float16, v); }
  static SRK_DEV elem from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  static SRK_DEV f32x16 mma(i32x4 a, i32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <> struct DTraits<SRK_F32> {
  typedef float elem;
  static constexpr int CH = 4;
  static constexpr bool IS16 = false;
  static SRK_DEV float to_f32(elem v) { return v; }
  static SRK_DEV elem from_f32(float f) { return f; }
  // one chunk = 4 floats -> four 32x32x2 MFMAs; lane half h supplies k = h of each
  static SRK_DEV f32x16 mma(i32x4 a, i32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.x), __int_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.y), __int_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.z), __int_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.w), __int_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

// ---------------------------------------------------------------------------------------------
// LDS activation image: [row][col] pixels, 128 bytes (8 chunks) per pixel, chunk slot XOR-swizzled
// by the pixel's column so that (a) the 32-pixel ds_read_b128 operand reads of the implicit GEMM
// (2 tile rows x 16 columns, any tap shift) and (b) the 4-pixel x 4-chunk ds_read_b64_tr_b16
// blocks of the weight-gradient kernel are both bank-conflict free (DESIGN.md "LDS image").
//   slot = chunk ^ g(col),  g(col) = ((u & 1) << 2) | (u >> 1),  u = (col >> 1) & 7
// The row pitch (in pixels) must be even so that the 256-byte bank-row half is col & 1.
// ---------------------------------------------------------------------------------------------
SRK_DEV int swz(int col) {
  int u = (col >> 1) & 7;
  return ((u & 1) << 2) | (u >> 1);
}

SRK_DEV i32x4 lds_read16(const char* p) { return *reinterpret_cast<const i32x4*>(p); }
SRK_DEV void lds_write16(char* p, i32x4 v) { *reinterpret_cast<i32x4*>(p) = v; }

SRK_DEV i32x4 gload16(const void* p) { return *reinterpret_cast<const i32x4*>(p); }

// ---------------------------------------------------------------------------------------------
// LDS-DMA the compiler does not see.  hipcc treats `buffer_load ... lds` (the builtin) as a store to LDS that
// may alias EVERY later LDS read, and puts `s_waitcnt vmcnt(0)` in front of the next ds_read: a tile prefetched into
// the other half of a double buffer is then waited for before the current tile's MFMA loop starts (no overlap at
// all).  Issued from inline asm, the transfer is invisible to that pass; the kernel orders it by hand
// (`s_waitcnt vmcnt(0)` + s_barrier before the buffer is read).  The compiler's own vmcnt bookkeeping stays safe:
// the counter retires in order, so operations it does not know about can only make its waits stricter.
//   rsrc: buffer descriptor in SGPRs (make_rsrc4), voff: per-lane byte offset (out of range -> zero fill),
//   lds_addr: wave-uniform LDS byte address; lane l lands at lds_addr + 16*l.
// ---------------------------------------------------------------------------------------------
SRK_DEV i32x4 make_rsrc4(const void* base, unsigned bytes) {
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uintptr_t)base);
  r.y = __builtin_amdgcn_readfirstlane((int)(((uintptr_t)base >> 32) & 0xffff));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
SRK_DEV unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// Store policy "write-through" (sc1) for the buffer-store builtins' `aux` operand.  A kernel's plain stores leave their lines dirty in the XCDs' L2s;
// they are written back when the kernel ends, in front of the next launch -- ~1-2 us that a chain of small dependent launches pays per link
// (round 5: all-workgroup stamps).  The conv kernels hand their activations to the next launch this way.  (Measured and NOT
// used in pw_chain.hip: WDSR-B 1.5 % slower with it at batch 16 and 256.)
constexpr int SRK_AUX_WT = 16;          // the same policy for the buffer-store builtins' `aux` operand
SRK_DEV void dma16_hidden(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :: "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory", "m0");
}

// Sum of one value per lane over the 64 lanes of a wave in ONE fixed order, returned uniformly (from lane 63): the GFX9 DPP
// reduction -- quad swaps, half-row and row mirrors, then row_bcast15 / row_bcast31 across the four rows of 16 -- six vector
// instructions and no LDS traffic (a ds_bpermute butterfly measured SLOWER than the serial 64-term sum it was to replace: six
// dependent LDS round trips per sum).  The channel-attention MLP's 64-term sums use it in the stand-alone kernels (ca.hip) AND
// inside the conv-pair launches (conv_pair.hip), so both forms stay bit-identical.
template <int CTRL, int ROWS> SRK_DEV float dpp_take(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
SRK_DEV float wave_sum64(float v) {
  v += dpp_take<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(v);      // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);      // row_mirror: every lane of a row holds the row's sum
  v += dpp_take<0x142, 0xa>(v);      // row_bcast15 into rows 1, 3
  v += dpp_take<0x143, 0xc>(v);      // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// XCD-contiguous block remap: blocks b and b+8 share an XCD under round-robin dispatch, so give each
// XCD a contiguous range of the linear work index (neighbouring tiles then share that XCD's L2).
// Bijective for every grid size; placement only affects speed, never results.
SRK_DEV int xcd_remap(int bid, int nb) {
  int q = nb >> 3, rem = nb & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
}

// ---------------------------------------------------------------------------------------------
// Output-channel (MFMA row) permutation of the packed weights.  A 32x32 MFMA leaves lane half h with rows
// {8i + 4h + e}: four 4-row groups 8 rows apart.  Packing the weights so that MFMA row rho of a channel block
// carries channel  c = 32h + 16cb + 4i + e  (64-channel blocks: rho = 32cb + 8i + 4h + e)  or
// c = 16h + 4i + e  (32-channel blocks)  makes the 16 accumulator registers of a lane 16 CONTIGUOUS channels
// (register index = channel offset), so the epilogue moves 16-byte pieces: 4x fewer store instructions than
// 4-channel groups (the store tail is issue-bound, guide T21).
// ---------------------------------------------------------------------------------------------
SRK_DEV int row_to_chan(int rho, int blk) {   // rho in [0, blk), blk = 64 or 32
  const int e = rho & 3, hh = (rho >> 2) & 1, i = (rho >> 3) & 3;
  if (blk == 64) return 32 * hh + 16 * (rho >> 5) + 4 * i + e;
  return 16 * hh + 4 * i + e;
}

// packed pair conversion: two floats -> one dword of two 16-bit elements (ONE v_cvt_pk instruction, RNE)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <int DT> SRK_DEV uint32_t pack2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  if constexpr (DT == SRK_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// one dword of two 16-bit elements -> two floats (bf16: one shift, one and)
template <int DT> SRK_DEV void unpack2(uint32_t w, float& lo, float& hi) {
  if constexpr (DT == SRK_BF16) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
  } else {
    const f16x2 h = __builtin_bit_cast(f16x2, w);
    lo = (float)h.x;
    hi = (float)h.y;
  }
}
// ---- ReLU (models/common.py:89,99-100: torch.relu) ---------------------------------------------------------------------------------
// Default build: one instruction per value -- fp32: v_max_f32 with 0 (IEEE maxNum: a NaN becomes 0); packed 16-bit: integer max with 0
// (negative <=> sign bit <=> negative as int16; -0.0 -> +0.0; a NaN keeps its payload unless its SIGN bit is set, then it becomes 0).
// torch.relu returns NaN for a NaN of either sign.  -DSRK_EXACT_RELU=1 (`make exact` -> libsrk_gfx950_exact.so, loaded when the process
// runs with SRK_EXACT_RELU=1) builds the NaN-preserving forms: `v < 0 ? 0 : v` (compare + select) and its packed equivalent.  Finite
// values: identical bits in both builds.  Cost of the exact build: profiles/r6_exact_relu.txt.
#ifndef SRK_EXACT_RELU
#define SRK_EXACT_RELU 0
#endif
SRK_DEV float relu_f32(float v) {
#if SRK_EXACT_RELU
  return v < 0.f ? 0.f : v;
#else
  return fmaxf(v, 0.f);
#endif
}
template <int DT> SRK_DEV uint32_t relu_pk16(uint32_t w) {
  typedef __attribute__((ext_vector_type(2))) short i16x2;
  const i16x2 v = __builtin_bit_cast(i16x2, w);
  const i16x2 z = {0, 0};
  const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, z));
#if SRK_EXACT_RELU
  // a NaN (magnitude bits above the exponent mask) passes whatever its sign: per half n = 1 for a NaN, t | w * n
  constexpr uint32_t EXP = DT == SRK_BF16 ? 0x7f80u : 0x7c00u;
  const uint32_t lo = w & 0x7fffu, hi = (w >> 16) & 0x7fffu;
  const uint32_t keep = (lo > EXP ? 0x0000ffffu : 0u) | (hi > EXP ? 0xffff0000u : 0u);
  return t | (w & keep);
#else
  return t;
#endif
}

// ReLU-backward mask on PACKED 16-bit results (conv_igemm.hip's prefetch variant, conv_pair.hip's intermediate epilogue).
// P (two packed 16-bit results) keeps the lanes whose mask element m is > 0 as a float (sign clear, magnitude non-zero; a NaN passes),
// the others become +0: three packed-integer instructions per TWO elements (max with 0: negatives and -0.0 -> 0; min with 1: 0 / 1;
// multiply).  Inline asm: from the vector builtins hipcc builds compares, selects and v_perm instead (6 instructions per dword).
SRK_DEV void mask_apply_pk16(uint32_t& P, uint32_t m) {
  uint32_t t;
  asm("v_pk_max_i16 %0, %2, 0\n\t"
      "v_pk_min_u16 %0, %0, %3\n\t"
      "v_pk_mul_lo_u16 %1, %1, %0"
      : "=&v"(t), "+v"(P) : "v"(m), "s"(0x00010001u));
}

// element <-> float helpers on 4-element groups (8 B for 16-bit types, 16 B for fp32)
template <int DT> SRK_DEV void load4(const typename DTraits<DT>::elem* p, float v[4]) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    i32x2 raw = *reinterpret_cast<const i32x2*>(p);
    uint32_t a = (uint32_t)raw.x, b = (uint32_t)raw.y;
    v[0] = Tr::to_f32((uint16_t)(a & 0xffff));
    v[1] = Tr::to_f32((uint16_t)(a >> 16));
    v[2] = Tr::to_f32((uint16_t)(b & 0xffff));
    v[3] = Tr::to_f32((uint16_t)(b >> 16));
  } else {
    f32x4 raw = *reinterpret_cast<const f32x4*>(p);
    v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
  }
}

template <int DT> SRK_DEV void store4(typename DTraits<DT>::elem* p, const float v[4]) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    uint32_t a = (uint32_t)Tr::from_f32(v[0]) | ((uint32_t)Tr::from_f32(v[1]) << 16);
    uint32_t b = (uint32_t)Tr::from_f32(v[2]) | ((uint32_t)Tr::from_f32(v[3]) << 16);
    i32x2 raw; raw.x = (int)a; raw.y = (int)b;
    *reinterpret_cast<i32x2*>(p) = raw;
  } else {
    f32x4 raw; raw.x = v[0]; raw.y = v[1]; raw.z = v[2]; raw.w = v[3];
    *reinterpret_cast<f32x4*>(p) = raw;
  }
}

// 16-bit operand fetch with the transposing LDS read.  For a 32x32x16 MFMA operand whose M/N index is a
// channel (32 channels of 32-block ch32) and whose K index is 16 consecutive pixels of one image row,
// lane l of 16-lane group G = l>>4 supplies the address of pixel k = 8*(G>>1) + 4*rd + q (q = (l&15)>>2),
// channels 16*(G&1) + 4p .. +3 (p = l&3), and receives channel 16*(G&1) + (l&15) of pixels 8*(G>>1)+4*rd+0..3.
// tr_lane_off() is the per-lane byte offset inside an image row for column shift col0 and read rd;
// the row offset (row * pitch * 128) is a compile-time immediate in the unrolled K loop.
SRK_DEV int tr_lane_off(int col0, int rd, int ch32, int lane) {
  const int G = lane >> 4, hh = G >> 1, rowblk = G & 1, q = (lane & 15) >> 2, p = lane & 3;
  const int chunk = ch32 * 4 + rowblk * 2 + (p >> 1);
  const int col = col0 + 8 * hh + 4 * rd + q;
  return (col << 7) + ((chunk ^ swz(col)) << 4) + ((p & 1) << 3);
}

SRK_DEV i32x4 tr_read2(const char* a0, const char* a1) {
  typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(a0));
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(a1));
  const i32x2 l2 = __builtin_bit_cast(i32x2, lo), h2 = __builtin_bit_cast(i32x2, hi);
  return i32x4{l2.x, l2.y, h2.x, h2.y};
}

// conv_ks.hip: 3x3 conv with >= 2 input blocks of 64 channels (K-streaming kernel); srk_conv2d (conv_igemm.hip) dispatches to it
bool srk_conv_ks_ok(const srk_conv_args& a);
int srk_conv_ks_launch(const srk_conv_args& a, hipStream_t st);
// conv_lk.hip: direct 5x5 / 7x7 / 9x9 convs with few channels on one side (forward, dgrad, weight gradient)
bool srk_conv_lk_ok(const srk_conv_args& a);
int srk_conv_lk_launch(const srk_conv_args& a, hipStream_t st);
bool srk_wgrad_lk_ok(const srk_wgrad_args& a);
int srk_wgrad_lk_slabs(const srk_wgrad_args& a);
int srk_wgrad_lk_slab_cout(const srk_wgrad_args& a);
int srk_wgrad_lk_launch(const srk_wgrad_args& a, hipStream_t st);
// conv1x1.hip: 1x1 conv with Cin <= 384 (all operands in LDS behind one wait)
bool srk_conv1x1_ok(const srk_conv_args& a);
int srk_conv1x1_launch(const srk_conv_args& a, hipStream_t st);


// ---------------------------------------------------------------------------------------------
// 4 x 4 transpose of 16-byte pieces inside every quad of lanes (quad-transposed stores: lane j of a quad then holds piece j of the
// quad's four pixels, so the four lanes of a store instruction cover 64 contiguous bytes of ONE pixel instead of 16 bytes of four).
// The same transpose for TWO register quartets at once with v_cndmask_b32_dpp: on gfx9 the DPP lane permutation is a modifier of the
// select's first source (VOP2 encoding, lane mask in VCC), so one instruction per register and stage does what the form above needs
// two for (select + v_mov_dpp + two selects per register pair): 16 vector instructions per 8 registers instead of 32.  The epilogue
// phase is bound by vector-instruction ISSUE beside the other group's MFMA wave, and the chip is power-limited under this kernel, so
// every instruction not issued counts twice.  The lane masks are constants (qi = lane & 3): b0 = odd lanes, b1 = lanes 2, 3 of a quad.
// v_cndmask_b32: D = VCC ? src1 : dpp(src0).  Hazards (inline asm is invisible to the compiler's hazard recogniser): a DPP read needs two
// wait states behind the VALU write of its source -- `s_nop 1` covers the inputs, the instruction order below covers the temporaries
// (every stage-2 DPP source is written at least two instructions earlier).
SRK_DEV void quad_transpose8_dpp(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3,
                                 uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
  const unsigned long long mb0 = 0xaaaaaaaaaaaaaaaaull, mn0 = 0x5555555555555555ull;
  const unsigned long long mb1 = 0xccccccccccccccccull, mn1 = 0x3333333333333333ull;
  uint32_t t0, t1, t2, t3, u0, u1, u2, u3, oa0, oa1, oa2, oa3, oc0, oc1, oc2, oc3;
  asm("s_nop 1\n\t"
      "s_mov_b64 vcc, %[mn0]\n\t"                                         // even lanes keep r0 / r2, odd lanes take the neighbour's r1 / r3
      "v_cndmask_b32_dpp %[t0], %[a1], %[a0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t2], %[a3], %[a2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[u0], %[c1], %[c0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[u2], %[c3], %[c2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mb0]\n\t"                                         // odd lanes keep r1 / r3, even lanes take the neighbour's r0 / r2
      "v_cndmask_b32_dpp %[t1], %[a0], %[a1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t3], %[a2], %[a3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[u1], %[c0], %[c1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[u3], %[c2], %[c3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mn1]\n\t"                                         // lanes 0, 1 keep r0 / r1, lanes 2, 3 take r2 / r3 from two lanes over
      "v_cndmask_b32_dpp %[oa0], %[t2], %[t0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oa1], %[t3], %[t1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oc0], %[u2], %[u0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oc1], %[u3], %[u1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mb1]\n\t"
      "v_cndmask_b32_dpp %[oa2], %[t0], %[t2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oa3], %[t1], %[t3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oc2], %[u0], %[u2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[oc3], %[u1], %[u3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3),
        [oa0] "=&v"(oa0), [oa1] "=&v"(oa1), [oa2] "=&v"(oa2), [oa3] "=&v"(oa3), [oc0] "=&v"(oc0), [oc1] "=&v"(oc1), [oc2] "=&v"(oc2), [oc3] "=&v"(oc3)
      : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3),
        [mb0] "s"(mb0), [mn0] "s"(mn0), [mb1] "s"(mb1), [mn1] "s"(mn1)
      : "vcc");
  a0 = oa0; a1 = oa1; a2 = oa2; a3 = oa3;
  c0 = oc0; c1 = oc1; c2 = oc2; c3 = oc3;
}

