// Two chained pointwise (1x1) convolutions with the WIDE intermediate kept on chip: WDSR-B's block opens with
//     h = relu(conv1x1(x; F -> 6F) + b1);   z = conv1x1(h; 6F -> int(0.8 F)) + b2          (models/wdsr.py:30-51)
// Launched one by one the 6F-channel tensor (768 channels at F = 128: 1.5 KB per pixel) is written and re-read, ~6x the traffic of
// the block's input and output together, and the two GEMMs run HBM-bound.  Here the hidden channels are walked in SLICES of 64:
// a wave owns 32 pixels, keeps their input fragments in registers for the whole tile, computes one slice of h with MFMAs
// (lane = pixel, accumulator registers = hidden channels), applies bias / ReLU, packs it -- and the packed registers ARE the
// B operand of the second GEMM's next K-steps (the weight ROWS of conv 1 are permuted at pack time so that the accumulator
// registers of a lane half are two 8-channel K-chunks): h never exists outside the register file.
//   forward  (pw_fwd_kernel)  : z = W2 relu(W1 x + b1) + b2                               32 MFMAs per wave and slice
//   backward (pw_bwd_kernel)  : recompute the slice of h (its sign is the ReLU mask), gh = (W2^T gz) * (h > 0), gx = W1^T gh [+ res];
//                               optionally h and gh leave for the weight-gradient GEMMs               48 MFMAs per wave and slice
// Workgroup = 8 waves x 32 pixels = 256 pixels (pixels are a flat list: a 1x1 conv has no geometry).  The weights of a slice
// (forward 32 KB, backward 48 KB at F = 128) stream through an LDS ring by hidden LDS-DMA (srk_common.h), all waves issuing and
// all waves computing (two per SIMD: one's convert / wait gaps are the other's MFMA time); the pixel tiles arrive once by
// LDS-DMA into ring slots that are still empty and go to registers.  Results leave by per-lane 16-byte stores (a lane holds 32
// contiguous channels of its pixel).
#include "srk_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// diagnostics build only (make stamp: -DSRK_PW_STAMPS=1, tools/stamp_pw.py): s_memtime of workgroup 0's waves 0 and 4 at the stages
// of pw_fwd_kernel, into a __device__ array nothing else reads
#if SRK_PW_STAMPS
__device__ unsigned long long pw_stamp_buf[2][64];
#define PW_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (i) < 64) pw_stamp_buf[threadIdx.x >> 8][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define PW_STAMP2(i) do { if (blockIdx.x == 0 && (threadIdx.x & 127) == 0 && (i) < 64) pw_stamp_buf[threadIdx.x >> 7][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PW_STAMP(i) do { } while (0)
#define PW_STAMP2(i) do { } while (0)
#endif

// MFMA row rho (0..63) of a 64-row hidden slice -> hidden channel inside the slice.  After a 32x32 MFMA lane half hh holds rows
// 8i + 4hh + e (i, e = 0..3) in accumulator register 4i + e; the next GEMM wants lane half hh to supply channels 16j + 8hh + t
// (t = 0..7) of K-step j.  So: block b = rho >> 5 feeds K-steps 2b (registers i = 0, 1) and 2b + 1 (i = 2, 3).
__host__ __device__ inline int pw_hid_of_row(int rho) {
  const int b = rho >> 5, rp = rho & 31;
  const int i = rp >> 3, hh = (rp >> 2) & 1, e = rp & 3;
  return 16 * (2 * b + (i >> 1)) + 8 * hh + 4 * (i & 1) + e;
}
// MFMA row -> stored channel for output rows (64-row blocks, as every conv kernel here: a lane half holds 32 contiguous channels)
__host__ __device__ inline int pw_chan_of_row(int rho) {
  const int e = rho & 3, hh = (rho >> 2) & 1, i = (rho >> 3) & 3, r64 = rho & 63;
  return (rho >> 6) * 64 + 32 * hh + 16 * (r64 >> 5) + 4 * i + e;
}

template <int KC1, int NRB> struct PwCfg {
  static constexpr int NT = 512;
  static constexpr int PXW = 32, PX = 8 * PXW;                   // pixels per wave / workgroup
  static constexpr int RI = 16 * KC1;                            // input channels = rows of the W1^T GEMM
  static constexpr int R2 = 32 * NRB;                            // padded output rows of conv 2 = K of the W2^T GEMM
  static constexpr int W1_BYTES = 2 * KC1 * 64 * 16;             // [chunk][64 hidden rows][16 B]
  static constexpr int W2_BYTES = 8 * R2 * 16;                   // [chunk (64 hidden / 8)][R2 rows][16 B]
  static constexpr int W2T_BYTES = (R2 / 8) * 64 * 16;           // [chunk (R2 / 8)][64 hidden rows][16 B]
  static constexpr int W1T_BYTES = 8 * RI * 16;                  // [chunk (64 hidden / 8)][RI rows][16 B]
  static constexpr int FWD_SLICE = W1_BYTES + W2_BYTES;
  static constexpr int BWD_SLICE = W1_BYTES + W2T_BYTES + W1T_BYTES;
  static constexpr int FWD_SLOTS = 4, BWD_SLOTS = 3;
  static constexpr int XT_BYTES = PX * RI * 2;                   // staged input tile
  static constexpr int ZT_BYTES = PX * R2 * 2;                   // staged gz tile (backward)
  static_assert(FWD_SLICE % 8192 == 0 && BWD_SLICE % 8192 == 0, "whole 1 KB pieces per wave");
  static_assert(XT_BYTES <= 2 * FWD_SLICE, "the input tile is staged in two empty ring slots");
};

// swizzle of the staged pixel tiles: 16-byte slot of chunk c of pixel r = c ^ f(r), chosen so that the 16 lanes one
// ds_read_b128 group serves ({0-3,12-15,20-27} and its shifts) hit 16 different 16-byte columns of the 256-byte bank row
template <int CHUNKS> SRK_DEV constexpr int pw_swz(int r) {
  if constexpr (CHUNKS >= 16) return r & 15;
  else return (r >> 1) & 7;        // 8 chunks (128 B) per pixel: two pixels per bank row
}

// one pixel tile (this wave's 32 pixels x CHUNKS 16-byte chunks) -> LDS at `dst` (wave-uniform), swizzled
template <int CHUNKS> SRK_DEV void pw_dma_pixels(i32x4 rsrc, long long p0, long long P, int pitch, int coff, int cvalid, unsigned dst, int lane) {
  constexpr int PIECES = 32 * CHUNKS / 64;                       // 1 KB pieces
#pragma unroll
  for (int k = 0; k < PIECES; ++k) {
    const int idx = k * 64 + lane;
    const int pl = idx / CHUNKS, slot = idx % CHUNKS;
    const int c = slot ^ pw_swz<CHUNKS>(pl);
    const long long p = p0 + pl;
    const bool ok = p < P && c * 8 < cvalid;
    const unsigned voff = ok ? (unsigned)((p * pitch + coff + c * 8) * 2) : 0x80000000u;
    dma16_hidden(rsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + k * 1024)));
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------------
template <int DT, int KC1, int NRB>
__global__ __launch_bounds__(512) void pw_fwd_kernel(const srk_pw_args a, unsigned x_bytes, unsigned w_bytes, int cst_pieces) {
  typedef DTraits<DT> Tr;
  typedef PwCfg<KC1, NRB> C;
  constexpr int SL = C::FWD_SLICE, NSLOT = C::FWD_SLOTS, PPW = SL / 8192;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const cst = smem + NSLOT * SL;                          // b1 (permuted, Chid floats) | b2 (permuted, R2 floats)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int NS = a.Chid >> 6;
  const long long P = a.P;
  const long long p0 = (long long)blockIdx.x * C::PX + wave * C::PXW;

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned ring_lds = lds_addr_of(ring), cst_lds = lds_addr_of(cst);
  const unsigned cst_bytes = (unsigned)cst_pieces * 1024u;
  PW_STAMP(0);

  // The weight slices are fetched by waves 4-7 alone, 2 PPW pieces each: an LDS-DMA piece costs its issuer ~60 cycles, and with all
  // eight waves issuing right behind the barrier nobody computed meanwhile (stamps: 350-470 cycles per slice).  Now each SIMD's
  // other wave (0-3) has the MFMA pipe to itself while its partner issues.  (Waves 0-3 have no transfers of their own in the
  // loop: its counted waits pass at once for them.)
  const bool issuer = wave >= 4;
  auto dma_slice = [&](int s) {
    if (!issuer) return;
    const unsigned dst = ring_lds + (unsigned)((s % NSLOT) * SL);
#pragma unroll
    for (int k = 0; k < 2 * PPW; ++k) {
      const int piece = (wave - 4) * 2 * PPW + k;
      dma16_hidden(wrsrc, cst_bytes + (unsigned)s * SL + piece * 1024 + lane * 16,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));
    }
  };

  // ---- prologue: constants, this wave's pixels (into ring slots 2, 3: still empty), slices 0 and 1 ------------------------------
  if (wave < cst_pieces)
    dma16_hidden(wrsrc, (unsigned)(wave * 1024 + lane * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(cst_lds + wave * 1024)));
  const unsigned xs_lds = ring_lds + 2 * SL + (unsigned)wave * (C::PXW * C::RI * 2);
  pw_dma_pixels<2 * KC1>(xrsrc, p0, P, a.x_pitch, a.x_coff, a.Cin, xs_lds, lane);
  dma_slice(0);
  dma_slice(1);
  PW_STAMP(1);
  if (!issuer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // constants and pixels; an issuer: everything but slice 1
  else if (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  PW_STAMP(2);
  __builtin_amdgcn_s_barrier();
  PW_STAMP(3);

  i32x4 xf[KC1];
  {
    const char* xs = ring + 2 * SL + wave * (C::PXW * C::RI * 2) + r * (C::RI * 2);
    const int g = pw_swz<2 * KC1>(r);
#pragma unroll
    for (int j = 0; j < KC1; ++j) xf[j] = lds_read16(xs + (((2 * j + h) ^ g) << 4));
  }
  f32x16 acc2[NRB];
  {
    const float* b2 = reinterpret_cast<const float*>(cst) + a.Chid;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(b2 + (rb * 2 + h) * 16 + 4 * q);
        acc2[rb][4 * q] = v.x; acc2[rb][4 * q + 1] = v.y; acc2[rb][4 * q + 2] = v.z; acc2[rb][4 * q + 3] = v.w;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has its pixels in registers: slots 2, 3 are free
  dma_slice(2);
  dma_slice(3);

  const int lane_a = (h * 64 + r) << 4;                          // A fragments of the 64-row W1 part
  const int lane_a2 = (h * C::R2 + r) << 4;                      // ... of the R2-row W2 part
  // One iteration = ONE stream of MFMAs: conv 1 of slice s + 1 (2 KC1 steps: K-step j = k >> 1, row block k & 1), then conv 2 of
  // slice s (4 NRB steps).  The weight fragment of step k + PF is requested before the MFMA of step k and the order is pinned
  // (sched_barrier), so the compiler's counted lgkmcnt waits leave PF reads in flight; the bias / ReLU / pack of slice s + 1
  // rides in the gaps of conv 2 (two 16-bit pairs per step, starting two steps in: the accumulators are complete by then).
  constexpr int G1 = 2 * KC1, G2 = 4 * NRB, PF = 3, NBF = 4;
  constexpr int CPS = (16 + G2 - 3) / (G2 - 2);                   // packed dwords per conv-2 step
  f32x16 acc1[2];
  i32x4 hfa[4], hfb[4];
  auto bias1 = [&](int s) {
    const float* b1 = reinterpret_cast<const float*>(cst) + s * 64;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(b1 + (b * 2 + h) * 16 + 4 * q);
        acc1[b][4 * q] = v.x; acc1[b][4 * q + 1] = v.y; acc1[b][4 * q + 2] = v.z; acc1[b][4 * q + 3] = v.w;
      }
  };
  auto pack_piece = [&](int d, i32x4 (&hf)[4]) {                 // dword d of the 16 that hold the slice: block d >> 3, K-step half (d >> 2) & 1
    const int b = d >> 3, m = (d >> 2) & 1, w = d & 3;
    const int v = (int)relu_pk16<DT>(pack2<DT>(acc1[b][8 * m + 2 * w], acc1[b][8 * m + 2 * w + 1]));
    if (w == 0) hf[2 * b + m].x = v; else if (w == 1) hf[2 * b + m].y = v; else if (w == 2) hf[2 * b + m].z = v; else hf[2 * b + m].w = v;
  };
  // first: first step of the stream (0: conv 1 + conv 2; G1: conv 2 only, the last iteration); w1 / w2: fragment bases of the lane
  auto stream = [&](const int first, const char* w1, const char* w2, const i32x4 (&hc)[4], i32x4 (&hn)[4]) {
    auto rd = [&](int k) {
      return k < G1 ? lds_read16(w1 + (k >> 1) * 2048 + (k & 1) * 512)
                    : lds_read16(w2 + ((k - G1) / NRB) * (2 * C::R2 * 16) + ((k - G1) % NRB) * 512);
    };
    i32x4 fa[NBF];
#pragma unroll
    for (int k = 0; k < PF; ++k) fa[k] = rd(first + k);
#pragma unroll
    for (int k = 0; k < G1 + G2; ++k) {
      if (k >= first) {
        if (k + PF < G1 + G2) fa[(k - first + PF) % NBF] = rd(k + PF);
        if (k < G1) acc1[k & 1] = Tr::mma(fa[(k - first) % NBF], xf[k >> 1], acc1[k & 1]);
        else {
          const int kk = k - G1;
          acc2[kk % NRB] = Tr::mma(fa[(k - first) % NBF], hc[kk / NRB], acc2[kk % NRB]);
          if (first == 0 && kk >= 2) {
#pragma unroll
            for (int c = 0; c < CPS; ++c)
              if ((kk - 2) * CPS + c < 16) pack_piece((kk - 2) * CPS + c, hn);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // slice 0's conv 1 alone (its weights are there: the prologue waited for slice 0)
  bias1(0);
  {
    const char* w1 = ring + lane_a;
#pragma unroll
    for (int k = 0; k < G1; ++k) acc1[k & 1] = Tr::mma(lds_read16(w1 + (k >> 1) * 2048 + (k & 1) * 512), xf[k >> 1], acc1[k & 1]);
#pragma unroll
    for (int d = 0; d < 16; ++d) pack_piece(d, hfa);
  }
  PW_STAMP(4);
  for (int s = 0; s < NS; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ss = s + u;
      if (ss < NS) {
        // slice ss + 1 must have landed; younger transfers: slices ss + 2 (and, in the first iteration, 3)
        if (ss == 0) { if (PPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (ss + 2 < NS) { if (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PW_STAMP(8 + 4 * ss);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PW_STAMP(9 + 4 * ss);
        if (ss >= 1 && ss + 3 < NS) dma_slice(ss + 3);           // into the slot slice ss - 1 left
        PW_STAMP(10 + 4 * ss);
        const char* const w1 = ring + ((ss + 1) % NSLOT) * SL + lane_a;
        const char* const w2 = ring + (ss % NSLOT) * SL + C::W1_BYTES + lane_a2;
        if (ss + 1 < NS) {
          bias1(ss + 1);
          if (u == 0) stream(0, w1, w2, hfa, hfb); else stream(0, w1, w2, hfb, hfa);
        } else {
          if (u == 0) stream(G1, w1, w2, hfa, hfb); else stream(G1, w1, w2, hfb, hfa);
        }
        PW_STAMP(11 + 4 * ss);
      }
    }
  }
  PW_STAMP(5);

  // ---- store: lane (pixel r, half h) holds channels 64 B + 32 h .. + 32 of its pixel ---------------------------------------------
  const long long p = p0 + r;
  if (p < P) {
    typename Tr::elem* const o = reinterpret_cast<typename Tr::elem*>(a.out) + p * a.out_pitch + a.out_coff;
#pragma unroll
    for (int B = 0; B < NRB / 2; ++B)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ch = 64 * B + 32 * h + 8 * k;
        if (ch < a.Cout) {
          const f32x16& v = acc2[2 * B + (k >> 1)];
          const int q0 = 8 * (k & 1);
          i32x4 q;
          q.x = (int)pack2<DT>(v[q0 + 0], v[q0 + 1]);
          q.y = (int)pack2<DT>(v[q0 + 2], v[q0 + 3]);
          q.z = (int)pack2<DT>(v[q0 + 4], v[q0 + 5]);
          q.w = (int)pack2<DT>(v[q0 + 6], v[q0 + 7]);
          *reinterpret_cast<i32x4*>(o + ch) = q;
        }
      }
  }
  PW_STAMP(6);
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward, persistent form: 4 waves x 64 pixels (two 32-pixel column blocks per wave: every weight fragment read from LDS feeds
// TWO MFMAs, which halves the LDS read traffic that bounded the 8 x 32 form -- there each wave read the whole 32 KB slice for 32
// MFMAs of 32 cycles: 128 B / clk / CU, the LDS's whole bandwidth), one wave per SIMD with the conv-2 accumulators in the
// accumulation registers.  A workgroup walks tiles blockIdx.x, + gridDim.x, ... and the weight ring NEVER drains between them: ring
// element i = [W2 of slice i | W1 of slice i + 1] (what ONE iteration uses: conv 2 of slice i, then conv 1 of slice i + 1, which
// for the last slice is slice 0 of the NEXT tile), and the next tile's pixels travel through the ring too, as two elements
// (column block 0 / 1 of every wave) placed before the tile's last element and read into the (by then dead) input registers during
// that iteration's conv 2.  The tile's results leave during the conv 1 that follows.  One barrier per iteration; the LDS-DMA pieces
// of the elements that barrier frees are issued one per MFMA step instead of in a burst.
// ------------------------------------------------------------------------------------------------------------------------------
// MFMAs with the accumulator's register file chosen by hand (the compiler's own choice put BOTH accumulator sets of the persistent
// forward into the accumulation registers and copied conv 1's out again for every pack: 64 extra instructions per iteration of a
// stream that has five issue slots per MFMA).  The compiler cannot see an MFMA inside the asm, so it inserts no wait states for
// it: the streams below keep every VALU / store read of an accumulator two or more MFMAs behind its last write, and MFMA -> MFMA
// on the same accumulator needs none.
// GUARD: the wait states a read of the result needs ride inside the statement -- for the accumulators that stay live across code the
// register allocator may spill around (it puts a spill right behind the defining instruction; tools/isa_mfma_hazards.py checks
// the listing for such reads)
template <int DT, bool GUARD = false> SRK_DEV void pw_mma_v(f32x16& acc, i32x4 a, i32x4 b) {
  if constexpr (GUARD) {
    if constexpr (DT == SRK_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 12" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 12" : "+v"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (DT == SRK_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}
template <int DT> SRK_DEV void pw_mma_vc(f32x16& acc, i32x4 a, i32x4 b, const f32x16& c) {   // acc = a b + c, acc and c different registers
  if constexpr (DT == SRK_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}
template <int V> struct pw_int { static constexpr int value = V; };
template <int I, int N, class F> SRK_DEV void pw_static_for(F&& f) {
  if constexpr (I < N) { f(pw_int<I>{}); pw_static_for<I + 1, N>(f); }
}
// the hidden LDS-DMA with the wave-uniform part of the address in the scalar offset (NOT part of the range check: a lane is
// masked by voff = 0x80000000)
SRK_DEV void dma16_hidden_s(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
template <int DT, int KC1, int NRB>
__global__ __launch_bounds__(256) void pw_fwd2_kernel(const srk_pw_args a, unsigned x_bytes, unsigned out_bytes, unsigned w_bytes, int cst_pieces, int ntiles) {
  typedef DTraits<DT> Tr;
  typedef PwCfg<KC1, NRB> C;
  constexpr int SL = C::FWD_SLICE, PPW = SL / 4096;               // 1 KB pieces per wave and ring element
  constexpr int PX = 256, CHUNKS = 2 * KC1;
  static_assert(C::W1_BYTES == C::W2_BYTES && PPW == KC1, "waves 0, 1 carry the W2 half of an element, waves 2, 3 the W1 half; a column block is one wave share");
  constexpr int NA = 4 * NRB, NB = 2 * KC1, PF = 3, NBF = 4;
  constexpr int PER_A = 8 / NRB, PER_B = (16 + KC1 - 2) / (KC1 - 1);
  constexpr int NSTORE = 4 * NRB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const cst = smem + 4 * SL;                               // b1 (permuted, Chid floats) | b2 (permuted, R2 floats)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int NS = a.Chid >> 6;
  const long long P = a.P;
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;
  const int m = (ntiles - bid + G - 1) / G;                      // tiles of this workgroup

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  // results leave by buffer stores (a lane outside the tile or the stored channels gets an out-of-range offset: every store is
  // issued, which the hand-counted vmcnt waits rely on); the builtin, not asm: the data registers of a 16-byte store must not be
  // rewritten in the next cycles, and only the compiler can see to that
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned ring_lds = lds_addr_of(ring), cst_lds = lds_addr_of(cst);
  const int cst_bytes = cst_pieces * 1024;

  // ---- the loader: ring position q -> (tile k, place u in the tile's cycle  E(0) .. E(NS-2), X0, X1, E(NS-1)) ----------------------
  int ld_q = 0, ld_k = -1, ld_u = NS - 1;                         // the stream opens with tile 0's pixels and E(NS-1) (for its W1 of slice 0)
  // per-lane parts of the addresses (everything else is wave-uniform and rides in the scalar offset): a pixel piece covers PPL
  // pixels, lane -> pixel q = lane / CHUNKS of the piece and 16-byte slot lane % CHUNKS, which holds chunk slot ^ swz(pixel); the
  // swizzle of pixel pc * PPL + q splits into a lane part and a piece part (K below).  ONE branch-free piece sequence serves
  // weight and pixel elements (a branch per piece would cut the pinned MFMA stream into basic blocks, and the compiler then
  // sinks the riding VALU work out of its steps): the element descriptor carries the differences.
  constexpr int PPL = 64 / CHUNKS;
  const int xq = lane / CHUNKS;
  const unsigned x_row = (unsigned)((xq * a.x_pitch + a.x_coff) * 2);
  const unsigned x_c0 = (unsigned)(((lane % CHUNKS) ^ pw_swz<CHUNKS>(xq)) << 4);
  const unsigned w_lane = (unsigned)(lane * 16);
  struct El {
    i32x4 rsrc;                                                   // the pixels' or the weights' buffer
    bool isx;                                                     // pixels (per-lane offset x_row + swizzled chunk) or weights (lane * 16)
    int lim;                                                      // pixels that exist from the element's first one on (weights: all)
    unsigned km, soff, sstep, dst;                                // swizzle mask (weights: 0), scalar offset of piece 0, its step per piece, LDS address
  };
  El cur, cur1, cur2;                                             // the elements whose pieces ride in the current iteration's stream
  auto describe = [&](El& e) __attribute__((always_inline)) {
    e.dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + (unsigned)((ld_q & 3) * SL + wave * (PPW * 1024))));
    const bool isx = ld_u == NS - 1 || ld_u == NS || ld_k >= m;   // pixels of tile k + 1: this wave's column block ld_u - (NS - 1); past the
    int src;                                                      // last tile every element is a masked transfer (zeros into a free slot)
    if (isx) {
      src = ld_k + 1 < m && ld_k < m ? (bid + (ld_k + 1) * G) * PX + wave * 64 + (ld_u - (NS - 1)) * 32 : -1;
      const long long left = src < 0 ? 0 : P - src;
      e.lim = __builtin_amdgcn_readfirstlane((int)(left > 64 ? 64 : left));
      e.soff = (unsigned)__builtin_amdgcn_readfirstlane(src < 0 ? 0 : src * a.x_pitch * 2);
      e.sstep = (unsigned)__builtin_amdgcn_readfirstlane(PPL * a.x_pitch * 2);
      e.km = ~0u;
    } else {
      const int s = ld_u == NS + 1 ? NS - 1 : ld_u;
      src = wave < 2 ? cst_bytes + s * SL + C::W1_BYTES + wave * (PPW * 1024)
                     : cst_bytes + (s + 1 == NS ? 0 : s + 1) * SL + (wave - 2) * (PPW * 1024);
      e.lim = 0x7fffffff;
      e.soff = (unsigned)__builtin_amdgcn_readfirstlane(src);
      e.sstep = 1024u;
      e.km = 0u;
    }
    e.rsrc = isx ? xrsrc : wrsrc;
    e.isx = isx;
    ++ld_q;
    if (++ld_u == NS + 2) { ld_u = 0; ++ld_k; }
  };
  auto issue_piece = [&](const El& e, auto pcc) __attribute__((always_inline)) {
    constexpr int pc = decltype(pcc)::value;
    constexpr unsigned K = (unsigned)(pw_swz<CHUNKS>(pc * PPL) << 4);
    static_assert(pw_swz<CHUNKS>(pc * PPL + PPL - 1) == (pw_swz<CHUNKS>(pc * PPL) | pw_swz<CHUNKS>(PPL - 1)), "swizzle splits");
    const unsigned voff = xq < e.lim - pc * PPL ? (e.isx ? x_row + (x_c0 ^ K) : w_lane) : 0x80000000u;
    dma16_hidden_s(e.rsrc, voff, e.soff + pc * e.sstep, e.dst + pc * 1024);
  };
  auto burst = [&]() __attribute__((always_inline)) {          // one whole element now
    El e;
    describe(e);
    pw_static_for<0, PPW>([&](auto pcc) __attribute__((always_inline)) { issue_piece(e, pcc); });
  };

  // ---- state of the tile in flight ------------------------------------------------------------------------------------------------
  i32x4 xf[KC1][2];
  i32x4 hf[4][2];
  f32x16 acc1[2][2], acc2[NRB][2];
  const int lane_a = (h * 64 + r) << 4;                          // A fragments of the 64-row W1 part
  const int lane_a2 = (h * C::R2 + r) << 4;                      // ... of the R2-row W2 part
  const float* const b1c = reinterpret_cast<const float*>(cst);
  const float* const b2c = b1c + a.Chid;

  auto pack1 = [&](auto bc, auto dc) __attribute__((always_inline)) {                          // dword d (0..15) of row block b: K-step 2b + (d >> 3), column block (d >> 2) & 1
    constexpr int b = decltype(bc)::value, d = decltype(dc)::value;
    constexpr int mm = d >> 3, c = (d >> 2) & 1, w = d & 3;
    const int v = (int)relu_pk16<DT>(pack2<DT>(acc1[b][c][8 * mm + 2 * w], acc1[b][c][8 * mm + 2 * w + 1]));
    i32x4& t = hf[2 * b + mm][c];
    if constexpr (w == 0) t.x = v; else if constexpr (w == 1) t.y = v; else if constexpr (w == 2) t.z = v; else t.w = v;
  };
  auto bias_vec = [&](const float* bp) __attribute__((always_inline)) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(bp + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
  };
  // Results leave through LDS: a lane holds 8 channels of ITS pixel, and stored from there every lane of a 16-byte store hits a
  // different pixel row -- 64 partial lines per instruction, the address path takes them one per cycle (measured: ~4,000 cycles
  // per tile for the whole CU, 10 % of the tile).  The wave instead writes its 32 x R2 block (+ b2) into its share of a pixel
  // element's ring slot (free once the next tile's fragments are in registers; 16-byte chunks swizzled like the input tile) and
  // reads it back row-major: a store then covers 64 / CH2 whole pixel rows.
  constexpr int CH2 = C::R2 / 8, ROWB = C::R2 * 2, RPI = 64 / CH2;
  static_assert(32 * ROWB == PPW * 1024, "a column block's results fit its pixel-element share");
  auto out_write = [&](auto rbc, auto cc, char* region) __attribute__((always_inline)) {
    constexpr int rb = decltype(rbc)::value, c = decltype(cc)::value;
    pw_static_for<0, 2>([&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      const int qch = 8 * (rb >> 1) + 2 * (rb & 1) + e + 4 * h;    // the 8-channel chunk these registers are
      const f32x16& v = acc2[rb][c];
      i32x4 q;
      q.x = (int)pack2<DT>(v[8 * e + 0], v[8 * e + 1]);
      q.y = (int)pack2<DT>(v[8 * e + 2], v[8 * e + 3]);
      q.z = (int)pack2<DT>(v[8 * e + 4], v[8 * e + 5]);
      q.w = (int)pack2<DT>(v[8 * e + 6], v[8 * e + 7]);
      lds_write16(region + r * ROWB + ((qch ^ pw_swz<CH2>(r)) << 4), q);
    });
  };
  // read-back / store i of a column block covers rows RPI i .. + RPI: lane -> row o_row of those, 16-byte position o_pos, which holds
  // chunk o_pos ^ swz(row) = (o_pos ^ swz(o_row)) ^ swz(RPI i) (the swizzle splits as for the input pieces)
  const int o_row = lane / CH2, o_pos = lane % CH2;
  const unsigned o_lds = (unsigned)(o_row * ROWB + (o_pos << 4));
  const unsigned o_c0 = (unsigned)((o_pos ^ pw_swz<CH2>(o_row)) << 4);
  const unsigned o_base = (unsigned)((o_row * a.out_pitch + a.out_coff) * 2);
  const unsigned o_cbytes = (unsigned)(a.Cout * 2);
  auto out_read = [&](auto ic, const char* region) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    return lds_read16(region + RPI * i * ROWB + o_lds);
  };
  auto out_send = [&](auto ic, int pxc, i32x4 q) __attribute__((always_inline)) {                  // pxc: the column block's first pixel
    constexpr int i = decltype(ic)::value;
    constexpr unsigned K = (unsigned)(pw_swz<CH2>(RPI * i) << 4);
    static_assert(pw_swz<CH2>(RPI * i + RPI - 1) == (pw_swz<CH2>(RPI * i) | pw_swz<CH2>(RPI - 1)), "swizzle splits");
    const unsigned cb = o_c0 ^ K;                                 // byte offset of the lane's chunk in its pixel row
    const int left = (int)(P - pxc) - RPI * i;                    // rows of this store that exist
    const bool ok = o_row < left && cb < o_cbytes;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, q), orsrc, (int)(ok ? o_base + cb : 0x80000000u),
                                           __builtin_amdgcn_readfirstlane((pxc + RPI * i) * a.out_pitch * 2), 0);
  };

  // One iteration = ONE pinned stream of steps (a weight fragment read PF steps ahead, two MFMAs, a share of the side work):
  //   part A (HAS_A): conv 2 of slice s, K-step outer; rides: the second half of slice s's hidden values (acc1[1] -> hf[2], hf[3], wanted
  //                   from K-step 2 on), with WRAP the next tile's input fragments, with BIAS_A (s = 0) the accumulators start from b2
  //   part B (HAS_B): conv 1 of slice sn (bias in the first MFMA's C operand), row block outer; rides: acc1[0] -> hf[0], hf[1] during
  //                   row block 1, with WRAP the tile's results (one accumulator block per step of row block 0)
  constexpr int HAS_A = 1, HAS_B = 2, WRAP = 4;
  auto iteration = [&](auto modec, const char* w2, const char* w1, int sn, bool s0, char* xs0, char* xs1, int px0) __attribute__((always_inline)) {
    constexpr int mode = decltype(modec)::value;
    constexpr int first = (mode & HAS_A) ? 0 : NA, last = (mode & HAS_B) ? NA + NB : NA;
    auto rd = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t < NA) return lds_read16(w2 + (t / NRB) * (2 * C::R2 * 16) + (t % NRB) * 512);
      else return lds_read16(w1 + ((t - NA) % KC1) * 2048 + ((t - NA) / KC1) * 512);
    };
    i32x4 fa[NBF];
    f32x16 bv0, bv1;                                              // conv 1's biases, the C operands of the row blocks' first MFMAs
    pw_static_for<0, PF>([&](auto tc) __attribute__((always_inline)) { fa[decltype(tc)::value] = rd(pw_int<first + decltype(tc)::value>{}); });
    if constexpr (!(mode & HAS_A)) bv0 = bias_vec(b1c + sn * 64 + h * 16);
    pw_static_for<first, last>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t + PF < last) fa[(t - first + PF) % NBF] = rd(pw_int<t + PF>{});
      const i32x4 f = fa[(t - first) % NBF];
      if constexpr (t < PPW) issue_piece(cur, pw_int<t>{});
      if constexpr (t < NA) {
        constexpr int j2 = t / NRB, rb = t % NRB;
        acc2[rb][0] = Tr::mma(f, hf[j2][0], acc2[rb][0]);
        acc2[rb][1] = Tr::mma(f, hf[j2][1], acc2[rb][1]);
        if constexpr (t >= 1 && (t - 1) * PER_A < 16)
          pw_static_for<(t - 1) * PER_A, (t * PER_A < 16 ? t * PER_A : 16)>([&](auto dc) __attribute__((always_inline)) { pack1(pw_int<1>{}, dc); });
        if constexpr ((mode & WRAP) != 0 && (t >> 1) < KC1)
          xf[t >> 1][t & 1] = lds_read16(((t & 1) ? xs1 : xs0) + r * (C::RI * 2) + (((2 * (t >> 1) + h) ^ pw_swz<CHUNKS>(r)) << 4));
        if constexpr ((mode & HAS_B) != 0 && t == NA - 4) bv0 = bias_vec(b1c + sn * 64 + h * 16);
      } else {
        constexpr int u = t - NA, b = u / KC1, j = u % KC1;
        if constexpr (j == 0) {
          pw_mma_vc<DT>(acc1[b][0], f, xf[0][0], b == 0 ? bv0 : bv1);
          pw_mma_vc<DT>(acc1[b][1], f, xf[0][1], b == 0 ? bv0 : bv1);
        } else {
          // outside the steady loop the last results stay live across the tile's output code or the loop's entry
          constexpr bool guard = mode != (HAS_A | HAS_B) && b == 1 && j == KC1 - 1;
          pw_mma_v<DT>(acc1[b][0], f, xf[j][0]);
          pw_mma_v<DT, guard>(acc1[b][1], f, xf[j][1]);
        }
        // an MFMA reads its C operand LATE: the registers must not be handed to anything else (the compiler does not know the asm
        // is an MFMA) until two more steps have gone by
        if constexpr (j == 2 && b == 0) asm volatile("" :: "v"(bv0));
        if constexpr (j == 2 && b == 1) asm volatile("" :: "v"(bv1));
        if constexpr (b == 0 && j == (KC1 >= 8 ? 3 : 2)) bv1 = bias_vec(b1c + sn * 64 + (2 + h) * 16);
        if constexpr (b == 1 && j >= 1 && (j - 1) * PER_B < 16)
          pw_static_for<(j - 1) * PER_B, (j * PER_B < 16 ? j * PER_B : 16)>([&](auto dc) __attribute__((always_inline)) { pack1(pw_int<0>{}, dc); });
        // s = 0: the two more elements the tile switch made room for, in the steps that carry no other side work (a branch cuts the
        // stream into basic blocks, and side work whose results are used beyond the cut is sunk there by the compiler)
        if constexpr (mode == (HAS_A | HAS_B) && b == 0) {
          if (s0) {
            issue_piece(2 * j < PPW ? cur1 : cur2, pw_int<(2 * j) % PPW>{});
            issue_piece(2 * j + 1 < PPW ? cur1 : cur2, pw_int<(2 * j + 1) % PPW>{});
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // the tile's results: + b2 -> LDS -> memory, the accumulators back to zero.  NOT inside the stream: with its temporaries there the
  // kernel no longer fits the register file, and a spill is a vector-memory instruction the hand-counted vmcnt waits do not know
  auto finish_tile = [&](char* xs0, char* xs1, int px0) __attribute__((always_inline)) {
    pw_static_for<0, NRB>([&](auto rbc) __attribute__((always_inline)) {
      constexpr int rb = decltype(rbc)::value;
      out_write(rbc, pw_int<0>{}, xs0);
      out_write(rbc, pw_int<1>{}, xs1);
      acc2[rb][0] = acc2[rb][1] = bias_vec(b2c + (rb * 2 + h) * 16);      // the next tile's accumulators start from b2
    });
    pw_static_for<0, 2>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      i32x4 od[2 * NRB];
      pw_static_for<0, 2 * NRB>([&](auto ic) __attribute__((always_inline)) { od[decltype(ic)::value] = out_read(ic, c ? xs1 : xs0); });
      pw_static_for<0, 2 * NRB>([&](auto ic) __attribute__((always_inline)) { out_send(ic, px0 + 32 * c, od[decltype(ic)::value]); });
    });
  };

  // ---- start: constants, tile 0's pixels, E(NS-1) (for W1 of slice 0), E(0) ----------------------------------------------------------
  PW_STAMP2(0);
  for (int pc = wave; pc < cst_pieces; pc += 4)
    dma16_hidden(wrsrc, (unsigned)(pc * 1024 + lane * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(cst_lds + pc * 1024)));
#pragma unroll 1
  for (int e = 0; e < 4; ++e) burst();
  if (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const char* const xs0 = ring + wave * (PPW * 1024) + r * (C::RI * 2);
    const int g = pw_swz<CHUNKS>(r);
#pragma unroll
    for (int j = 0; j < KC1; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) xf[j][c] = lds_read16(xs0 + c * SL + (((2 * j + h) ^ g) << 4));
  }
  PW_STAMP2(1);
  pw_static_for<0, NRB>([&](auto rbc) __attribute__((always_inline)) {
    constexpr int rb = decltype(rbc)::value;
    acc2[rb][0] = acc2[rb][1] = bias_vec(b2c + (rb * 2 + h) * 16);
  });
  iteration(pw_int<HAS_B>{}, ring, ring + 2 * SL + C::W2_BYTES + lane_a, 0, false, ring, ring, 0);
  PW_STAMP2(2);

  for (int k = 0; k < m; ++k) {
    const int base = 3 + k * (NS + 2);
    const int px0 = (bid + k * G) * PX + wave * 64;
    if (k == 2) PW_STAMP2(52);
    // Before each iteration: its element (ring position p) must have landed -- younger transfers: two elements (s = 1 .. NS - 2), the
    // previous tile's stores (s = 0), nothing (s = NS - 1) -- and every wave must be done with the slot the new transfers go to.
    // Ring positions up to p + 3 may be filled: one element per iteration rides in its stream, at s = 0 three (the tile switch
    // freed the pixel elements' slots too).
#pragma unroll 1
    for (int s_ = 0; s_ < NS - 1; ++s_) {
      const int p = base + s_;
      if (s_ == 0) {
        if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NSTORE == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (PPW == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (k == 1) PW_STAMP2(4 + 4 * s_);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (k == 1) PW_STAMP2(5 + 4 * s_);
      describe(cur);
      if (s_ == 0) { describe(cur1); describe(cur2); }
      const char* const slot = ring + (p & 3) * SL;
      if (k == 1) PW_STAMP2(6 + 4 * s_);
      iteration(pw_int<(HAS_A | HAS_B)>{}, slot + lane_a2, slot + C::W2_BYTES + lane_a, s_ + 1, s_ == 0, ring, ring, px0);
      if (k == 1) PW_STAMP2(7 + 4 * s_);
    }
    const int p = base + NS + 1;
    if (k == 1) PW_STAMP2(4 + 4 * (NS - 1));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (k == 1) PW_STAMP2(5 + 4 * (NS - 1));
    describe(cur);                               // position p + 1: the next tile's E(0)
    if (k == 1) PW_STAMP2(6 + 4 * (NS - 1));
    const char* const slot = ring + (p & 3) * SL;
    char* const xs0 = ring + ((p - 2) & 3) * SL + wave * (PPW * 1024);      // this wave's shares of the two pixel elements
    char* const xs1 = ring + ((p - 1) & 3) * SL + wave * (PPW * 1024);
    // (after the last tile the conv 1 of "the next tile's slice 0" runs on the zeros of the masked pixel elements: 32 MFMAs per
    // workgroup, and one copy of the stream less -- every copy is one more place where the compiler moves the accumulators)
    iteration(pw_int<(HAS_A | HAS_B | WRAP)>{}, slot + lane_a2, slot + C::W2_BYTES + lane_a, 0, false, xs0, xs1, px0);
    if (k == 1) PW_STAMP2(7 + 4 * (NS - 1));
    finish_tile(xs0, xs1, px0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the masked transfers of the last iterations still target this LDS
  PW_STAMP2(3);
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward (data): gx = W1^T ((W2^T gz) * (W1 x + b1 > 0)) [+ res]; WH: the slices of h and gh also go to HBM (operands of the
// weight-gradient GEMMs).  The hidden pre-activation is RE-computed from x with the forward's instruction sequence (same
// fragments, same accumulation order: bit-identical), so its sign is exactly the forward's ReLU mask.
// ------------------------------------------------------------------------------------------------------------------------------
template <int DT, int KC1, int NRB, bool WH>
__global__ __launch_bounds__(512) void pw_bwd_kernel(const srk_pw_bwd_args a, unsigned x_bytes, unsigned gz_bytes, unsigned w_bytes, int cst_pieces) {
  typedef DTraits<DT> Tr;
  typedef PwCfg<KC1, NRB> C;
  constexpr int SL = C::BWD_SLICE, NSLOT = C::BWD_SLOTS, PPW = SL / 8192;
  constexpr int KCZ = NRB * 2;                                   // K-steps of the W2^T GEMM (R2 / 16)
  constexpr int NIB = KC1 / 2;                                   // 32-row blocks of the W1^T GEMM (RI / 32)
  constexpr int NST = WH ? 8 : 0;                                // per-lane stores per slice (they count in vmcnt too)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const cst = smem + NSLOT * SL;                          // b1 (permuted)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int NS = a.Chid >> 6;
  const long long P = a.P;
  const long long p0 = (long long)blockIdx.x * C::PX + wave * C::PXW;

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  const i32x4 zrsrc = make_rsrc4(a.gz, gz_bytes);
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned ring_lds = lds_addr_of(ring), cst_lds = lds_addr_of(cst);
  const unsigned cst_bytes = (unsigned)cst_pieces * 1024u;

  const bool issuer = wave >= 4;                                  // the weight slices: waves 4-7 alone, 2 PPW pieces each (see pw_fwd_kernel)
  auto dma_slice = [&](int s) {
    if (!issuer) return;
    const unsigned dst = ring_lds + (unsigned)((s % NSLOT) * SL);
#pragma unroll
    for (int k = 0; k < 2 * PPW; ++k) {
      const int piece = (wave - 4) * 2 * PPW + k;
      dma16_hidden(wrsrc, cst_bytes + (unsigned)s * SL + piece * 1024 + lane * 16,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));
    }
  };

  // ---- prologue: constants, slice 0, then this wave's x and gz pixels one after the other through its own staging area (slots 1, 2) --
  if (wave < cst_pieces)
    dma16_hidden(wrsrc, (unsigned)(wave * 1024 + lane * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(cst_lds + wave * 1024)));
  dma_slice(0);
  constexpr int STG = C::PXW * (C::RI > C::R2 ? C::RI : C::R2) * 2;      // per-wave staging bytes
  static_assert(8 * STG <= 2 * SL, "staging fits the two empty slots");
  const unsigned st_lds = ring_lds + SL + (unsigned)wave * STG;
  const char* const st = ring + SL + wave * STG;
  pw_dma_pixels<2 * KC1>(xrsrc, p0, P, a.x_pitch, a.x_coff, a.Cin, st_lds, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  i32x4 xf[KC1], zf[KCZ];
  {
    const int g = pw_swz<2 * KC1>(r);
#pragma unroll
    for (int j = 0; j < KC1; ++j) xf[j] = lds_read16(st + r * (C::RI * 2) + (((2 * j + h) ^ g) << 4));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pw_dma_pixels<2 * KCZ>(zrsrc, p0, P, a.gz_pitch, a.gz_coff, a.Cz, st_lds, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const int g = pw_swz<2 * KCZ>(r);
#pragma unroll
    for (int j = 0; j < KCZ; ++j) zf[j] = lds_read16(st + r * (C::R2 * 2) + (((2 * j + h) ^ g) << 4));
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // staging areas free, slice 0 and the constants have landed everywhere
  dma_slice(1);
  dma_slice(2);

  const int lane_a = (h * 64 + r) << 4;
  const int lane_ai = (h * C::RI + r) << 4;
  f32x16 gx[NIB];
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
    for (int q = 0; q < 16; ++q) gx[ib][q] = 0.f;

  const long long p = p0 + r;
  typename Tr::elem* const ho = WH ? reinterpret_cast<typename Tr::elem*>(a.h_out) + p * a.Chid : nullptr;
  typename Tr::elem* const go = WH ? reinterpret_cast<typename Tr::elem*>(a.gh_out) + p * a.Chid : nullptr;

  for (int s = 0; s < NS; ++s) {
    const char* const slot = ring + (s % NSLOT) * SL;
    // (1) pre-activation of the slice, exactly as the forward computed it
    f32x16 acc[2];
    {
      const float* b1 = reinterpret_cast<const float*>(cst) + s * 64;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(b1 + (b * 2 + h) * 16 + 4 * q);
          acc[b][4 * q] = v.x; acc[b][4 * q + 1] = v.y; acc[b][4 * q + 2] = v.z; acc[b][4 * q + 3] = v.w;
        }
      const char* w = slot + lane_a;
#pragma unroll
      for (int j = 0; j < KC1; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = Tr::mma(lds_read16(w + j * 2048 + b * 512), xf[j], acc[b]);
    }
    bool on[2][16];
    i32x4 hf[4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) on[b][q] = acc[b][q] > 0.f;
      if (WH) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          i32x4 q;
          q.x = (int)relu_pk16<DT>(pack2<DT>(acc[b][8 * m + 0], acc[b][8 * m + 1]));
          q.y = (int)relu_pk16<DT>(pack2<DT>(acc[b][8 * m + 2], acc[b][8 * m + 3]));
          q.z = (int)relu_pk16<DT>(pack2<DT>(acc[b][8 * m + 4], acc[b][8 * m + 5]));
          q.w = (int)relu_pk16<DT>(pack2<DT>(acc[b][8 * m + 6], acc[b][8 * m + 7]));
          hf[2 * b + m] = q;
        }
      }
    }
    if (WH && p < P) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<i32x4*>(ho + 64 * s + 16 * j + 8 * h) = hf[j];
    }
    // (2) gh = (W2^T gz) masked
    {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
      const char* w = slot + C::W1_BYTES + lane_a;
#pragma unroll
      for (int j = 0; j < KCZ; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = Tr::mma(lds_read16(w + j * 2048 + b * 512), zf[j], acc[b]);
    }
    i32x4 gf[4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = on[b][8 * m + t] ? acc[b][8 * m + t] : 0.f;
        i32x4 q;
        q.x = (int)pack2<DT>(v[0], v[1]); q.y = (int)pack2<DT>(v[2], v[3]); q.z = (int)pack2<DT>(v[4], v[5]); q.w = (int)pack2<DT>(v[6], v[7]);
        gf[2 * b + m] = q;
      }
    if (WH && p < P) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<i32x4*>(go + 64 * s + 16 * j + 8 * h) = gf[j];
    }
    // (3) gx += W1^T gh
    {
      const char* w = slot + C::W1_BYTES + C::W2T_BYTES + lane_ai;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) gx[ib] = Tr::mma(lds_read16(w + j * (2 * C::RI * 16) + ib * 512), gf[j], gx[ib]);
    }
    // hand-over: slice s + 1 has landed (younger: slice s + 2 and this iteration's stores), slot s % 3 is free for slice s + 3
    if (s + 2 < NS) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW + NST) : "memory");      // (waves 0-3: only their NST stores can be out)
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (s + 3 < NS) dma_slice(s + 3);
  }

  // ---- store: gx (+ res); lane (pixel r, half h) holds channels 64 B + 32 h .. + 32 ----------------------------------------------
  if (p < P) {
    typename Tr::elem* const o = reinterpret_cast<typename Tr::elem*>(a.gx) + p * a.gx_pitch + a.gx_coff;
    const typename Tr::elem* const rs = a.res ? reinterpret_cast<const typename Tr::elem*>(a.res) + p * a.res_pitch + a.res_coff : nullptr;
#pragma unroll
    for (int B = 0; B < NIB / 2; ++B)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ch = 64 * B + 32 * h + 8 * k;
        const f32x16& v = gx[2 * B + (k >> 1)];
        const int q0 = 8 * (k & 1);
        float f[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) f[t] = v[q0 + t];
        if (rs) {
          const i32x4 q = *reinterpret_cast<const i32x4*>(rs + ch);
          const int qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f0, f1;
            unpack2<DT>((uint32_t)qw[e], f0, f1);
            f[2 * e] += f0; f[2 * e + 1] += f1;
          }
        }
        i32x4 q;
        q.x = (int)pack2<DT>(f[0], f[1]); q.y = (int)pack2<DT>(f[2], f[3]); q.z = (int)pack2<DT>(f[4], f[5]); q.w = (int)pack2<DT>(f[6], f[7]);
        *reinterpret_cast<i32x4*>(o + ch) = q;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward (weights): dW2[hid][z] = sum_p h[p][hid] gz[p][z],  dW1[hid][in] = sum_p gh[p][hid] x[p][in],  db1[hid] = sum_p gh[p][hid],
// db2[z] = sum_p gz[p][z].  Neither h nor gh exists in HBM: a workgroup owns TWO slices (128 hidden channels) and a range of pixels,
// which it walks in tiles of 64; x and gz arrive by LDS-DMA as images in the conv kernels' swizzled format (three buffers).
// The eight waves are PRODUCERS and CONSUMERS one tile apart (one of each per SIMD, so one's vector work runs beside the other's MFMAs):
//   waves 0-3, tile t    : re-compute h and gh of the two slices for the tile (lane = pixel; the W1 / W2^T fragments of the wave's two
//                          32-row blocks stay in registers for the whole launch), write both as 16-bit images        32 MFMAs per tile
//   waves 4-7, tile t - 1: read the four images with the transposing LDS read (K = pixels) and accumulate the slices' rows of
//                          dW2 and dW1 in registers (each wave the same 64-column tile of both slices)               32 MFMAs per tile
// One workgroup barrier per tile; 32 KB of DMA per 2k MFMA cycles (the one-slice version needed 64 KB: at the ~31 B/clk a CU accepts).
// At the end every workgroup stores its partial sums to its own slab (no atomics: fixed order, reproducible);
// pw_wgrad_finalize_kernel adds the slabs of the pixel ranges.
// ------------------------------------------------------------------------------------------------------------------------------
// swizzle of pw_wgrad_kernel's h / gh images: slot = chunk ^ hsw(pixel); a bit permutation of the pixel's low three bits (bit 1 -> bit 2)
#ifndef SRK_PW_OLD_SWZ          // diagnostics build only (tools/ab_pw.sh; `make pwold` builds tools/ubench/libsrk_pwold.so): round 4's swizzle of these images, for the same-box A/B
#define SRK_PW_OLD_SWZ 0
#endif
SRK_DEV int hsw(int pl) { return SRK_PW_OLD_SWZ ? swz(pl & 15) : ((pl & 1) | (((pl >> 1) & 1) << 2) | (((pl >> 2) & 1) << 1)); }
// tr_lane_off (srk_common.h) for such an image: 16 consecutive pixels of a K-step, pixel pitch 128 B
SRK_DEV int tr_lane_off_h(int rd, int ch32, int lane) {
  const int G = lane >> 4, hh = G >> 1, rowblk = G & 1, q = (lane & 15) >> 2, p = lane & 3;
  const int chunk = ch32 * 4 + rowblk * 2 + (p >> 1);
  const int col = 8 * hh + 4 * rd + q;
  return (col << 7) + ((chunk ^ hsw(col)) << 4) + ((p & 1) << 3);
}

template <int DT, int KC1, int NRB>
__global__ __launch_bounds__(512) void pw_wgrad_kernel(const srk_pw_wgrad_args a, unsigned x_bytes, unsigned gz_bytes, int NR, int tq, int trem) {
  typedef DTraits<DT> Tr;
  typedef PwCfg<KC1, NRB> C;
  constexpr int TP = 64;                                         // pixels per tile
  constexpr int PLANE = TP * 128;                                // one 64-channel block of a tile: 8 KB
  constexpr int XPL = C::RI / 64, ZPL = C::R2 / 64;              // planes of x / gz
  constexpr int BUF = (XPL + ZPL) * PLANE, NXB = 3;
  constexpr int HB = 4 * PLANE;                                  // h (2 slices) | gh (2 slices) of one tile
  constexpr int NTS = XPL + ZPL;                                 // 64 x 64 tiles of [dW2 | dW1] per slice
  constexpr int TPW = 2 * NTS / 4;                               // ... per consumer wave
  constexpr int KCZ = NRB * 2;
  constexpr int PPW = (XPL + ZPL) * 8 / 8;                       // DMA pieces per wave and tile (8 pieces per plane)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const hbase = smem + NXB * BUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int NSP = a.Chid >> 7;                                   // slice pairs
  // Which (slice pair, pixel range) this workgroup owns.  The NSP workgroups of one pixel range read the SAME x / gz tiles: when the
  // number of ranges is a multiple of 8 they are given block ids that differ by multiples of 8, i.e. the same XCD under round-robin
  // dispatch (placement only affects speed), so that XCD's L2 serves five of the six reads -- round 3's mapping (consecutive ids)
  // spread them over six XCDs: 1,734 MB of L2-fabric traffic per launch against 283 MB of x + gz (profiles/r3_pw_wgrad_n256_pmc.txt).
  int sp, rg;
  if ((NR & 7) == 0) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    sp = k % NSP;
    rg = (k / NSP) * 8 + xcd;
  } else {
    sp = blockIdx.x % NSP;
    rg = blockIdx.x / NSP;
  }
  const long long P = a.P;
  const int t0 = rg * tq + min(rg, trem), nt = tq + (rg < trem ? 1 : 0);

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes), zrsrc = make_rsrc4(a.gz, gz_bytes);
  const unsigned lds0 = lds_addr_of(smem);
  auto dma_tile = [&](int tile, int b) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave * PPW + k, plane = q >> 3, pc = q & 7;
      const int pl = 8 * pc + (lane >> 3), c8 = (lane & 7) ^ swz(pl & 15);
      const long long p = (long long)tile * TP + pl;
      const bool isx = plane < XPL;
      const int ch = (isx ? plane : plane - XPL) * 64 + c8 * 8;
      const bool ok = p < P && ch < (isx ? a.Cin : a.Cz);
      const unsigned voff = ok ? (unsigned)((p * (isx ? a.x_pitch : a.gz_pitch) + (isx ? a.x_coff : a.gz_coff) + ch) * 2) : 0x80000000u;
      dma16_hidden(isx ? xrsrc : zrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * BUF + plane * PLANE + pc * 1024)));
    }
  };

  const bool producer = wave < 4;
  const int cst_bytes = (a.Chid * 4 + 1023) / 1024 * 1024;
  // Both roles run the SAME sequence of nt + 1 steps, each opened by `step_sync` (wait for the own DMA pieces, workgroup barrier, issue
  // the next tile's pieces); the role bodies are separate code paths so that the register allocation is the larger of the two, not
  // their sum (producers: 128 registers of weight fragments; consumers: 128 accumulator registers).
  auto step_sync = [&](int st) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // tile st landed; h / gh of tile st - 1 written; tile st - 2 read
    if (st + 1 < nt) dma_tile(t0 + st + 1, (st + 1) % NXB);
  };
  if (producer) {
    // ---- weight fragments and bias of the wave's two 32-row blocks (global row block 2 rq + u of the 4: slice rbg >> 1) -------------
    const int pb = wave & 1, rq = (wave >> 1) & 1;
    i32x4 w1r[2][KC1], w2r[2][KCZ];
    float b1r[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rbg = 2 * rq + u, sl = rbg >> 1, rb = rbg & 1;
      const char* const slice = reinterpret_cast<const char*>(a.wpk) + cst_bytes + (size_t)(2 * sp + sl) * C::BWD_SLICE;
#pragma unroll
      for (int j = 0; j < KC1; ++j) w1r[u][j] = gload16(slice + (((2 * j + h) * 64 + rb * 32 + r) << 4));
#pragma unroll
      for (int j = 0; j < KCZ; ++j) w2r[u][j] = gload16(slice + C::W1_BYTES + (((2 * j + h) * 64 + rb * 32 + r) << 4));
      const float* b1 = reinterpret_cast<const float*>(a.wpk) + (2 * sp + sl) * 64 + (rb * 2 + h) * 16;
#pragma unroll
      for (int q = 0; q < 16; ++q) b1r[u][q] = b1[q];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // (the loads are waited for here, before any hidden DMA is queued)
#pragma unroll
      for (int j = 0; j < KC1; ++j) asm volatile("" : "+v"(w1r[u][j]));
#pragma unroll
      for (int j = 0; j < KCZ; ++j) asm volatile("" : "+v"(w2r[u][j]));
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(b1r[u][q]));
    }
    const int pl = 32 * pb + r, g = swz(pl & 15);
    // h / gh images (written here, read by the consumers with the transposing LDS read): their OWN swizzle.  With the activation
    // image's swizzle (two neighbouring pixels share one) the eight lanes of a ds_write_b128 group -- eight consecutive pixels, one
    // chunk -- hit four 16-byte slots twice: every one of these stores was a 2-way bank conflict, 256 conflict cycles per tile =
    // exactly one per MFMA of the kernel (SQ_LDS_BANK_CONFLICT == SQ_INSTS_MFMA in profiles/r4_pw_wgrad_*_pmc.txt).  hsw() gives
    // eight consecutive pixels eight different slots and keeps the consumers' 4-pixel x 4-chunk blocks conflict-free
    // (tools/lds_conflicts_pw.py).
    const int g2 = hsw(pl);
    if (nt > 0) dma_tile(t0, 0);
#pragma unroll 1
    for (int st = 0; st <= nt; ++st) {
      step_sync(st);
      if (st < nt) {
        const char* const B0 = smem + (st % NXB) * BUF;
        char* const hb = hbase + (st & 1) * HB;
        const char* const xp = B0 + (pl << 7);
        const char* const zp = B0 + XPL * PLANE + (pl << 7);
        const bool valid = (long long)(t0 + st) * TP + pl < P;   // pixels beyond P: x and gz are zeros, but h = relu(b1) is not
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rbg = 2 * rq + u, sl = rbg >> 1, rb = rbg & 1;
          f32x16 pre, gh;
#pragma unroll
          for (int q = 0; q < 16; ++q) { pre[q] = b1r[u][q]; gh[q] = 0.f; }
#pragma unroll
          for (int j = 0; j < KC1; ++j)
            pre = Tr::mma(w1r[u][j], lds_read16(xp + (j >> 2) * PLANE + (((((2 * j) & 7) + h) ^ g) << 4)), pre);
#pragma unroll
          for (int j = 0; j < KCZ; ++j)
            gh = Tr::mma(w2r[u][j], lds_read16(zp + (j >> 2) * PLANE + (((((2 * j) & 7) + h) ^ g) << 4)), gh);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float hv[8], gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const bool on = pre[8 * m + t] > 0.f;
              hv[t] = (on && valid) ? pre[8 * m + t] : 0.f;
              gv[t] = on ? gh[8 * m + t] : 0.f;
            }
            const int off = sl * PLANE + (pl << 7) + (((4 * rb + 2 * m + h) ^ g2) << 4);
            lds_write16(hb + off, i32x4{(int)pack2<DT>(hv[0], hv[1]), (int)pack2<DT>(hv[2], hv[3]), (int)pack2<DT>(hv[4], hv[5]), (int)pack2<DT>(hv[6], hv[7])});
            lds_write16(hb + 2 * PLANE + off, i32x4{(int)pack2<DT>(gv[0], gv[1]), (int)pack2<DT>(gv[2], gv[3]), (int)pack2<DT>(gv[4], gv[5]), (int)pack2<DT>(gv[6], gv[7])});
          }
        }
      }
    }
  } else {
    // ---- consumers: accumulator tiles tt = cw + 4 u: slice tt / NTS, column tile tt % NTS of [gz planes | x planes] -------------------
    const int cw = wave & 3;
    f32x16 acc[TPW][2][2];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[u][i][o][e] = 0.f;
    int toff[2][2], hoff[2][2];                                   // lane offsets into the x / gz planes (swz) and the h / gh images (hsw)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        toff[i][rd] = tr_lane_off(0, rd, i, lane);
        hoff[i][rd] = tr_lane_off_h(rd, i, lane);
      }
    float dbz[2] = {0.f, 0.f}, dbh[TPW][2];
#pragma unroll
    for (int u = 0; u < TPW; ++u) dbh[u][0] = dbh[u][1] = 0.f;
    if (nt > 0) dma_tile(t0, 0);
#pragma unroll 1
    for (int st = 0; st <= nt; ++st) {
      step_sync(st);
      if (st >= 1) {
        const char* const B0 = smem + ((st - 1) % NXB) * BUF;
        const char* const hb = hbase + ((st - 1) & 1) * HB;
#pragma unroll
        for (int rr = 0; rr < TP / 16; ++rr) {
#pragma unroll
          for (int u = 0; u < TPW; ++u) {
            const int tt = cw + 4 * u, sl = tt / NTS, t2 = tt % NTS;
            const bool is2 = t2 < ZPL;                            // dW2 tiles read h against gz plane t2, dW1 tiles gh against x plane t2 - ZPL
            const char* const a_img = hb + (is2 ? 0 : 2 * PLANE) + sl * PLANE + rr * 2048;
            const char* const b_img = B0 + (is2 ? XPL + t2 : t2 - ZPL) * PLANE + rr * 2048;
            i32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              af[i] = tr_read2(a_img + hoff[i][0], a_img + hoff[i][1]);
              bf[i] = tr_read2(b_img + toff[i][0], b_img + toff[i][1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int o = 0; o < 2; ++o) acc[u][i][o] = Tr::mma(af[i], bf[o], acc[u][i][o]);
            if (is2 && sl == 0 && sp == 0 && a.db2p != nullptr) { // db2 = sum_p gz: lane = channel, 8 pixels per read
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                const int qw[4] = {bf[o].x, bf[o].y, bf[o].z, bf[o].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { float f0, f1; unpack2<DT>((uint32_t)qw[e], f0, f1); dbz[o] += f0 + f1; }
              }
            }
            if (t2 == ZPL) {                                      // db1 = sum_p gh: from the fragments of the first x column tile's waves
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int qw[4] = {af[i].x, af[i].y, af[i].z, af[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { float f0, f1; unpack2<DT>((uint32_t)qw[e], f0, f1); dbh[u][i] += f0 + f1; }
              }
            }
          }
        }
      }
    }
    // ---- the consumers store the workgroup's slab -----------------------------------------------------------------------------------
    const int hq = lane >> 5;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int tt = cw + 4 * u, sl = tt / NTS, t2 = tt % NTS;
      const bool is2 = t2 < ZPL;
      const int ncol = is2 ? C::R2 : C::RI;
      float* const slab = (is2 ? a.dw2p : a.dw1p) + (size_t)rg * a.Chid * ncol;
      const int col0 = 64 * (is2 ? t2 : t2 - ZPL), row0 = 64 * (2 * sp + sl);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = row0 + 32 * i + 4 * hq + (e & 3) + 8 * (e >> 2);
            slab[(size_t)row * ncol + col0 + 32 * o + (lane & 31)] = acc[u][i][o][e];
          }
      if (t2 == ZPL) {                                            // lanes l and l + 32 hold the two K halves of a read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float v = dbh[u][i] + __shfl_xor(dbh[u][i], 32, 64);
          if (lane < 32) a.db1p[(size_t)rg * a.Chid + row0 + 32 * i + lane] = v;
        }
      }
      if (is2 && sl == 0 && sp == 0 && a.db2p != nullptr) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const float v = dbz[o] + __shfl_xor(dbz[o], 32, 64);
          if (lane < 32) a.db2p[(size_t)rg * C::R2 + 64 * t2 + 32 * o + lane] = v;
        }
      }
    }
  }
}

// dW1 [Chid][Cin] and dW2 [Cmid][Chid] (the parameters' row-major fp32 layouts), db1 and db2 from the NR slabs of pw_wgrad_kernel:
// one thread per 4 consecutive slab elements, the NR loads of a thread independent of each other, slabs added in index order
SRK_DEV void pw_fin_item(const srk_pw_wgrad_args& a, int NR, int RI, int R2, long long gi) {
  const long long n1 = (long long)a.Chid * RI / 4, n2 = (long long)a.Chid * R2 / 4;
  if (gi < n1 + n2) {
    const bool one = gi < n1;
    const long long j = 4 * (one ? gi : gi - n1);
    const float* src = (one ? a.dw1p : a.dw2p) + j;
    const size_t stride = (size_t)a.Chid * (one ? RI : R2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < NR; ++k) v += *reinterpret_cast<const f32x4*>(src + (size_t)k * stride);
    if (one) *reinterpret_cast<f32x4*>(a.dw1 + j) = v;            // [hid][in], RI == Cin
    else {
      const int hid = (int)(j / R2), z = (int)(j % R2);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (z + t < a.Cmid) a.dw2[(size_t)(z + t) * a.Chid + hid] = e[t];
    }
  } else {
    const long long i = gi - n1 - n2;
    if (i < a.Chid) {
      float v = 0.f;
      for (int k = 0; k < NR; ++k) v += a.db1p[(size_t)k * a.Chid + i];
      if (a.db1) a.db1[i] = v;
    } else if (i < a.Chid + a.Cmid && a.db2) {
      const int z = (int)(i - a.Chid);
      float v = 0.f;
      for (int k = 0; k < NR; ++k) v += a.db2p[(size_t)k * R2 + z];
      a.db2[z] = v;
    }
  }
}

__global__ __launch_bounds__(256) void pw_wgrad_finalize_kernel(const srk_pw_wgrad_args a, int NR, int RI, int R2) {
  pw_fin_item(a, NR, RI, R2, (long long)blockIdx.x * 256 + threadIdx.x);
}

// The finalize steps of SEVERAL pointwise pairs in one launch (round 5; VERDICT r4 weak #4: WDSR-B's 16 blocks issued 16 of these,
// 12.8 us each = 6.8 % of the batch-16 step; every other finalize of the library was grouped already): blockIdx.y = job (its
// srk_pw_wgrad_args from a device table; RI = Cin, R2 = CoutP, NR = nranges), blockIdx.x strides over the job's items.
__global__ __launch_bounds__(256) void pw_wgrad_finalize_group_kernel(const srk_pw_wgrad_args* __restrict__ jobs) {
  const srk_pw_wgrad_args a = jobs[blockIdx.y];
  const long long fin = (long long)a.Chid * (a.Cin + a.CoutP) / 4 + a.Chid + a.CoutP;
  for (long long gi = (long long)blockIdx.x * 256 + threadIdx.x; gi < fin; gi += (long long)gridDim.x * 256)
    pw_fin_item(a, a.nranges, a.Cin, a.CoutP, gi);
}

// ------------------------------------------------------------------------------------------------------------------------------
// packing: fp32 [Chid][Cin] / [Cmid][Chid] weights (+ biases) -> constant block + per-slice blocks of both directions
//   fwd : cst = b1p[Chid] | b2p[R2] (padded to whole KB) ; slice s = W1 part | W2 part
//   bwd : cst = b1p[Chid]           (padded to whole KB) ; slice s = W1 part | W2^T part | W1^T part
// one work item = one 16-byte chunk
// ------------------------------------------------------------------------------------------------------------------------------
template <int DT> __device__ void pw_pack_body(const srk_pw_pack_args& a, long long stride, long long first) {
  typedef DTraits<DT> Tr;
  const int KC1 = a.Cin / 16, R2 = a.CoutP, RI = a.Cin, NS = a.Chid / 64;
  const int w1c = 2 * KC1 * 64, w2c = 8 * R2, w2tc = (R2 / 8) * 64, w1tc = 8 * RI;       // chunks per part
  const int fsl = w1c + w2c, bsl = w1c + w2tc + w1tc;
  const int fcst = ((a.Chid + R2) * 4 + 1023) / 1024 * 1024, bcst = (a.Chid * 4 + 1023) / 1024 * 1024;
  const long long nf = (long long)NS * fsl, nb = (long long)NS * bsl;
  auto put = [&](void* base, long long byte_off, const float (&v)[8]) {
    i32x4 q;
    q.x = (int)pack2<DT>(v[0], v[1]); q.y = (int)pack2<DT>(v[2], v[3]); q.z = (int)pack2<DT>(v[4], v[5]); q.w = (int)pack2<DT>(v[6], v[7]);
    *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(base) + byte_off) = q;
  };
  auto w1_chunk = [&](int s, int q, float (&v)[8]) {            // q = c * 64 + rho
    const int c = q >> 6, hid = 64 * s + pw_hid_of_row(q & 63);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = a.w1[(size_t)hid * a.Cin + 8 * c + t];
  };
  for (long long it = first; it < nf + nb; it += stride) {
    float v[8];
    if (it < nf) {
      const int s = (int)(it / fsl), q = (int)(it % fsl);
      if (q < w1c) w1_chunk(s, q, v);
      else {
        const int q2 = q - w1c, c = q2 / R2, ch = pw_chan_of_row(q2 % R2);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = ch < a.Cmid ? a.w2[(size_t)ch * a.Chid + 64 * s + 8 * c + t] : 0.f;
      }
      put(a.fwd, fcst + it * 16, v);
    } else {
      const long long ib = it - nf;
      const int s = (int)(ib / bsl), q = (int)(ib % bsl);
      if (q < w1c) w1_chunk(s, q, v);
      else if (q < w1c + w2tc) {
        const int q2 = q - w1c, c = q2 >> 6, hid = 64 * s + pw_hid_of_row(q2 & 63);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (8 * c + t) < a.Cmid ? a.w2[(size_t)(8 * c + t) * a.Chid + hid] : 0.f;
      } else {
        const int q2 = q - w1c - w2tc, c = q2 / RI, ch = pw_chan_of_row(q2 % RI);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = a.w1[(size_t)(64 * s + 8 * c + t) * a.Cin + ch];
      }
      put(a.bwd, bcst + ib * 16, v);
    }
  }
  // constants: b1 in accumulator order [s][b][h][4i + e], b2 [rb][h][4i + e]
  for (long long i = first; i < a.Chid + R2; i += stride) {
    if (i < a.Chid) {
      const int s = (int)i >> 6, k = (int)i & 63, b = k >> 5, hh = (k >> 4) & 1, q = k & 15, ii = q >> 2, e = q & 3;
      const float bv = a.b1 ? a.b1[64 * s + pw_hid_of_row(32 * b + 8 * ii + 4 * hh + e)] : 0.f;
      reinterpret_cast<float*>(a.fwd)[i] = bv;
      reinterpret_cast<float*>(a.bwd)[i] = bv;
    } else {
      const int k = (int)(i - a.Chid), rb = k >> 5, hh = (k >> 4) & 1, q = k & 15, ii = q >> 2, e = q & 3;
      const int ch = pw_chan_of_row(32 * rb + 8 * ii + 4 * hh + e);
      reinterpret_cast<float*>(a.fwd)[i] = (a.b2 && ch < a.Cmid) ? a.b2[ch] : 0.f;
    }
  }
}

template <int DT> __global__ void pw_pack_kernel(const srk_pw_pack_args a) {
  pw_pack_body<DT>(a, (long long)gridDim.x * blockDim.x, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void pw_pack_group_kernel(const srk_pw_pack_args* __restrict__ table) {
  const srk_pw_pack_args a = table[blockIdx.y];
  const long long stride = (long long)gridDim.x * blockDim.x, first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.dtype == SRK_BF16) pw_pack_body<SRK_BF16>(a, stride, first);
  else pw_pack_body<SRK_F16>(a, stride, first);
}

bool pw_shape_ok(int Cin, int Chid, int CoutP) {
  return ((Cin == 128 && CoutP == 128) || (Cin == 64 && CoutP == 64)) && Chid % 128 == 0 && Chid >= 256 && Chid <= 4096;
}

template <int DT, int KC1, int NRB> int pw_fwd_launch(const srk_pw_args& a, hipStream_t st) {
  typedef PwCfg<KC1, NRB> C;
  const int cst_bytes = ((a.Chid + C::R2) * 4 + 1023) / 1024 * 1024;
  const int lds = C::FWD_SLOTS * C::FWD_SLICE + cst_bytes;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_fwd_kernel<DT, KC1, NRB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_pw_forward: cannot reserve LDS"); return (int)attr; }
  SRK_CHECK_ARG(lds <= 160 * 1024 && cst_bytes <= 8 * 1024, "srk_pw_forward: %d bytes of LDS", lds);
  const long long nb = (a.P + C::PX - 1) / C::PX;
  const unsigned xb = (unsigned)(a.P * a.x_pitch * 2);
  const unsigned wb = (unsigned)(cst_bytes + (long long)(a.Chid / 64) * C::FWD_SLICE);
  hipLaunchKernelGGL((pw_fwd_kernel<DT, KC1, NRB>), dim3((unsigned)nb), dim3(C::NT), lds, st, a, xb, wb, cst_bytes / 1024);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT, int KC1, int NRB> int pw_fwd2_launch(const srk_pw_args& a, hipStream_t st) {
  typedef PwCfg<KC1, NRB> C;
  const int cst_bytes = ((a.Chid + C::R2) * 4 + 1023) / 1024 * 1024;
  const int lds = 4 * C::FWD_SLICE + cst_bytes;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_fwd2_kernel<DT, KC1, NRB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_pw_forward: cannot reserve LDS"); return (int)attr; }
  SRK_CHECK_ARG(lds <= 160 * 1024 && cst_bytes <= 8 * 1024, "srk_pw_forward: %d bytes of LDS", lds);
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  const long long nt = (a.P + 255) / 256;
  const unsigned xb = (unsigned)(a.P * a.x_pitch * 2), ob = (unsigned)(a.P * a.out_pitch * 2);
  const unsigned wb = (unsigned)(cst_bytes + (long long)(a.Chid / 64) * C::FWD_SLICE);
  hipLaunchKernelGGL((pw_fwd2_kernel<DT, KC1, NRB>), dim3((unsigned)(nt < cus ? nt : cus)), dim3(256), lds, st, a, xb, ob, wb, cst_bytes / 1024, (int)nt);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT, int KC1, int NRB> int pw_bwd_launch(const srk_pw_bwd_args& a, hipStream_t st) {
  typedef PwCfg<KC1, NRB> C;
  const int cst_bytes = (a.Chid * 4 + 1023) / 1024 * 1024;
  const int lds = C::BWD_SLOTS * C::BWD_SLICE + cst_bytes;
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_kernel<DT, KC1, NRB, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_kernel<DT, KC1, NRB, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr0 != hipSuccess || attr1 != hipSuccess) { srk_set_error("srk_pw_backward: cannot reserve LDS"); return (int)(attr0 != hipSuccess ? attr0 : attr1); }
  SRK_CHECK_ARG(lds <= 160 * 1024 && cst_bytes <= 8 * 1024, "srk_pw_backward: %d bytes of LDS", lds);
  const long long nb = (a.P + C::PX - 1) / C::PX;
  const unsigned xb = (unsigned)(a.P * a.x_pitch * 2), zb = (unsigned)(a.P * a.gz_pitch * 2);
  const unsigned wb = (unsigned)(cst_bytes + (long long)(a.Chid / 64) * C::BWD_SLICE);
  if (a.h_out) hipLaunchKernelGGL((pw_bwd_kernel<DT, KC1, NRB, true>), dim3((unsigned)nb), dim3(C::NT), lds, st, a, xb, zb, wb, cst_bytes / 1024);
  else hipLaunchKernelGGL((pw_bwd_kernel<DT, KC1, NRB, false>), dim3((unsigned)nb), dim3(C::NT), lds, st, a, xb, zb, wb, cst_bytes / 1024);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT, int KC1, int NRB> int pw_wgrad_launch(const srk_pw_wgrad_args& a, hipStream_t st, int NR, bool finalize) {
  typedef PwCfg<KC1, NRB> C;
  constexpr int PLANE = 64 * 128, BUF = (C::RI / 64 + C::R2 / 64) * PLANE, LDS = 3 * BUF + 2 * 4 * PLANE;
  static_assert(LDS <= 160 * 1024, "LDS");
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_kernel<DT, KC1, NRB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_pw_wgrad: cannot reserve LDS"); return (int)attr; }
  const long long ntiles = (a.P + 63) / 64;
  const int NS = a.Chid / 128;                                   // slice pairs
  hipLaunchKernelGGL((pw_wgrad_kernel<DT, KC1, NRB>), dim3((unsigned)(NS * NR)), dim3(512), LDS, st, a,
                     (unsigned)(a.P * a.x_pitch * 2), (unsigned)(a.P * a.gz_pitch * 2), NR, (int)(ntiles / NR), (int)(ntiles % NR));
  if (finalize) {
    const long long fin = (long long)a.Chid * (C::RI + C::R2) / 4 + a.Chid + C::R2;
    hipLaunchKernelGGL(pw_wgrad_finalize_kernel, dim3((unsigned)((fin + 255) / 256)), dim3(256), 0, st, a, NR, C::RI, C::R2);
  }
  SRK_LAUNCH_CHECK();
  return 0;
}

int pw_wgrad_ranges(long long P, int Chid) {
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  const long long ntiles = (P + 63) / 64;
  const int nsp = Chid / 128;
  long long nr = cus / nsp;
  // a multiple of 8 ranges with all workgroups of a range on one XCD (cus / 8 workgroups fit there: one per CU)
  const long long nr8 = (long long)((cus / 8) / nsp) * 8;
  if (nr8 >= 8 && nr8 <= ntiles) nr = nr8;
  if (nr < 1) nr = 1;
  if (nr > ntiles) nr = ntiles;
  return (int)nr;
}

}  // namespace

extern "C" int srk_pw_shape_ok(int Cin, int Chid, int CoutP) { return pw_shape_ok(Cin, Chid, CoutP) ? 1 : 0; }

extern "C" long long srk_pw_pack_bytes(int Cin, int Chid, int CoutP, int bwd) {
  if (!pw_shape_ok(Cin, Chid, CoutP)) return -1;
  const int KC1 = Cin / 16, R2 = CoutP, RI = Cin, NS = Chid / 64;
  const long long w1 = 2LL * KC1 * 64 * 16, w2 = 8LL * R2 * 16, w2t = (R2 / 8) * 64LL * 16, w1t = 8LL * RI * 16;
  if (!bwd) return ((Chid + R2) * 4 + 1023) / 1024 * 1024 + NS * (w1 + w2);
  return (Chid * 4 + 1023) / 1024 * 1024 + NS * (w1 + w2t + w1t);
}

extern "C" int srk_pw_pack(const srk_pw_pack_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->w1 && a->w2 && a->fwd && a->bwd, "srk_pw_pack: null pointer");
  SRK_CHECK_ARG(pw_shape_ok(a->Cin, a->Chid, a->CoutP) && a->Cmid <= a->CoutP && a->dtype != SRK_F32,
                "srk_pw_pack: unsupported shape %d -> %d -> %d (rows %d)", a->Cin, a->Chid, a->Cmid, a->CoutP);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(pw_pack_kernel<SRK_BF16>, dim3(128), dim3(256), 0, st, *a);
  else hipLaunchKernelGGL(pw_pack_kernel<SRK_F16>, dim3(128), dim3(256), 0, st, *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_pw_pack_group(const srk_pw_pack_args* table_dev, int n, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && n > 0 && n <= 65535, "srk_pw_pack_group: bad table (%d entries)", n);
  hipLaunchKernelGGL(pw_pack_group_kernel, dim3(64, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_pw_forward(const srk_pw_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->wpk && a->out, "srk_pw_forward: null pointer");
  SRK_CHECK_ARG(pw_shape_ok(a->Cin, a->Chid, a->CoutP) && a->dtype != SRK_F32, "srk_pw_forward: unsupported shape %d -> %d -> rows %d",
                a->Cin, a->Chid, a->CoutP);
  SRK_CHECK_ARG(a->P > 0 && a->P * (long long)a->x_pitch * 2 < 0x7fff0000LL && a->x_pitch % 8 == 0 && a->x_coff % 8 == 0 &&
                a->out_pitch % 8 == 0 && a->out_coff % 8 == 0 && a->Cout % 8 == 0 && a->Cout <= a->CoutP,
                "srk_pw_forward: addressing (P=%lld pitch %d)", a->P, a->x_pitch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // up to one 256-pixel tile per CU: the one-tile-per-workgroup kernel (8 waves x 32 pixels: two waves per SIMD hide each other's
  // waits, which is worth more than the persistent kernel's streamlined tile switch when there is no second tile); beyond: the
  // persistent kernel (4 waves x 64 pixels, the weight ring kept running across a workgroup's tiles)
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  if ((a->P + 255) / 256 <= cus) {
    if (a->Cin == 128) return a->dtype == SRK_BF16 ? pw_fwd_launch<SRK_BF16, 8, 4>(*a, st) : pw_fwd_launch<SRK_F16, 8, 4>(*a, st);
    return a->dtype == SRK_BF16 ? pw_fwd_launch<SRK_BF16, 4, 2>(*a, st) : pw_fwd_launch<SRK_F16, 4, 2>(*a, st);
  }
  SRK_CHECK_ARG(a->P * (long long)a->out_pitch * 2 < 0x7fff0000LL, "srk_pw_forward: output addressing (P=%lld pitch %d)", a->P, a->out_pitch);
  if (a->Cin == 128) return a->dtype == SRK_BF16 ? pw_fwd2_launch<SRK_BF16, 8, 4>(*a, st) : pw_fwd2_launch<SRK_F16, 8, 4>(*a, st);
  return a->dtype == SRK_BF16 ? pw_fwd2_launch<SRK_BF16, 4, 2>(*a, st) : pw_fwd2_launch<SRK_F16, 4, 2>(*a, st);
}

extern "C" int srk_pw_backward(const srk_pw_bwd_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->gz && a->wpk && a->gx, "srk_pw_backward: null pointer");
  SRK_CHECK_ARG(pw_shape_ok(a->Cin, a->Chid, a->CoutP) && a->dtype != SRK_F32, "srk_pw_backward: unsupported shape %d -> %d -> rows %d",
                a->Cin, a->Chid, a->CoutP);
  SRK_CHECK_ARG((a->h_out == nullptr) == (a->gh_out == nullptr), "srk_pw_backward: h_out and gh_out come together");
  SRK_CHECK_ARG(a->P > 0 && a->P * (long long)a->x_pitch * 2 < 0x7fff0000LL && a->P * (long long)a->gz_pitch * 2 < 0x7fff0000LL &&
                a->x_pitch % 8 == 0 && a->x_coff % 8 == 0 && a->gz_pitch % 8 == 0 && a->gz_coff % 8 == 0 && a->Cz % 8 == 0 && a->Cz <= a->CoutP &&
                a->gx_pitch % 8 == 0 && a->gx_coff % 8 == 0 && (!a->res || (a->res_pitch % 8 == 0 && a->res_coff % 8 == 0)),
                "srk_pw_backward: addressing (P=%lld)", a->P);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->Cin == 128) return a->dtype == SRK_BF16 ? pw_bwd_launch<SRK_BF16, 8, 4>(*a, st) : pw_bwd_launch<SRK_F16, 8, 4>(*a, st);
  return a->dtype == SRK_BF16 ? pw_bwd_launch<SRK_BF16, 4, 2>(*a, st) : pw_bwd_launch<SRK_F16, 4, 2>(*a, st);
}

extern "C" int srk_pw_wgrad_ranges(long long P, int Chid) { return (P > 0 && Chid >= 64) ? pw_wgrad_ranges(P, Chid) : 0; }

static int pw_wgrad_entry(const srk_pw_wgrad_args* a, srk_stream_t stream, bool finalize, const char* who) {
  SRK_CHECK_ARG(a && a->x && a->gz && a->wpk && a->dw1p && a->dw2p && a->db1p && a->dw1 && a->dw2, "%s: null pointer", who);
  SRK_CHECK_ARG(pw_shape_ok(a->Cin, a->Chid, a->CoutP) && a->dtype != SRK_F32 && a->Cmid <= a->CoutP, "%s: unsupported shape %d -> %d -> rows %d",
                who, a->Cin, a->Chid, a->CoutP);
  SRK_CHECK_ARG(a->P > 0 && a->P * (long long)a->x_pitch * 2 < 0x7fff0000LL && a->P * (long long)a->gz_pitch * 2 < 0x7fff0000LL &&
                a->x_pitch % 8 == 0 && a->x_coff % 8 == 0 && a->gz_pitch % 8 == 0 && a->gz_coff % 8 == 0 && a->Cz % 8 == 0 && a->Cz <= a->CoutP,
                "%s: addressing (P=%lld)", who, a->P);
  const int NR = pw_wgrad_ranges(a->P, a->Chid);
  SRK_CHECK_ARG(a->nranges == NR, "%s: nranges=%d but srk_pw_wgrad_ranges() is %d", who, a->nranges, NR);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->Cin == 128) return a->dtype == SRK_BF16 ? pw_wgrad_launch<SRK_BF16, 8, 4>(*a, st, NR, finalize) : pw_wgrad_launch<SRK_F16, 8, 4>(*a, st, NR, finalize);
  return a->dtype == SRK_BF16 ? pw_wgrad_launch<SRK_BF16, 4, 2>(*a, st, NR, finalize) : pw_wgrad_launch<SRK_F16, 4, 2>(*a, st, NR, finalize);
}

extern "C" int srk_pw_wgrad(const srk_pw_wgrad_args* a, srk_stream_t stream) { return pw_wgrad_entry(a, stream, true, "srk_pw_wgrad"); }

extern "C" int srk_pw_wgrad_partial(const srk_pw_wgrad_args* a, srk_stream_t stream) { return pw_wgrad_entry(a, stream, false, "srk_pw_wgrad_partial"); }

extern "C" int srk_pw_wgrad_finalize_group(const srk_pw_wgrad_args* jobs_dev, int njobs, int blocks_per_job, srk_stream_t stream) {
  SRK_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535 && blocks_per_job > 0, "srk_pw_wgrad_finalize_group: %d jobs, %d blocks each", njobs, blocks_per_job);
  hipLaunchKernelGGL(pw_wgrad_finalize_group_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}

#if SRK_PW_STAMPS
extern "C" int srk_pw_read_stamps(unsigned long long* host128) {
  return (int)hipMemcpyFromSymbol(host128, HIP_SYMBOL(pw_stamp_buf), sizeof(unsigned long long) * 128);
}
#endif
