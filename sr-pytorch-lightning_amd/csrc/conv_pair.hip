// Two chained 3x3 64->64 convolutions in ONE launch, for the small-batch regime (the reference's batch of 16).
//
//     mid = epi_mid( conv3x3(x, W1) + b1 )          epi_mid: ReLU | ReLU-backward mask, * scale_mid
//     out = conv3x3(mid, W2) + b2, * scale_out, + res
//
// i.e. a whole ResBlock (models/common.py:74-109: conv -> ReLU -> conv, * res_scale, += x), the conv pair of an RCAB
// (models/rcan.py:33-55) and, with data-gradient weights, the backward chain of either (dgrad of conv 2 with the ReLU
// mask, dgrad of conv 1, + upstream gradient).
//
// Why: at 16 patches of 48x48 a 64->64 layer is 144 tiles of 16x16 -- one per workgroup of the weight-stationary kernel,
// on 144 of 256 CUs -- and a launch spends 7.2k cycles before its first MFMA (per-lane set-up, 115 KB of LDS-DMA issue), 5.5k
// in MFMAs, 2.4k in the epilogue, 1.2k leaving and 3.8k between launches: 8.1 us in the kernel for 2.6 us of matrix work
// (DESIGN.md section 7).  The chain of 70 such launches is 54 % of an EDSR-baseline training step at that batch.
// Here one workgroup owns a 14x14 OUTPUT tile and recomputes the intermediate on the 16x16 pixels around it (= exactly 8
// MFMA pixel blocks; 31 % more conv-1 work) from an 18x18 input tile, so nothing is exchanged between workgroups: one
// launch, one prologue and one drain per PAIR of layers, the second layer's weights arrive while the first computes, the
// intermediate never leaves LDS on its way to conv 2 and the residual comes from the input tile already in LDS.
// 48x48 x 16 images = 256 tiles = every CU.
//
// LDS (152,064 B): input tile 18x18x128 B (swizzled image format of srk_common.h), intermediate 16 rows x pitch 18,
// and a 3-slot ring of weight slabs of 3 taps each (24 KB: six slabs a0 a1 a2 b0 b1 b2 stream through it by LDS-DMA, two
// slabs ahead of the MFMAs).  8 waves: wave w owns pixel block w (2 tile rows x 16 columns = 32 pixels) x all 64
// channels (2 accumulator tiles): per K-step 2 weight fragments + 1 pixel fragment -> 2 MFMAs.
#include "srk_common.h"
#include <type_traits>

// diagnostics build only (make stamp, tools/stamp_pair.py): s_memtime stamps of workgroup 0 through `b2` (conv 2 then runs
// without bias)
#ifndef SRK_PAIR_STAMPS
#define SRK_PAIR_STAMPS 0
#endif
#ifndef SRK_PAIR_PD
#define SRK_PAIR_PD 2
#endif
#ifndef SRK_PAIR_ST_AUX
#define SRK_PAIR_ST_AUX 16      // cache policy of the tile stores (gfx940+: 16 = sc1, write-through).  With plain stores the tiles sit dirty in the XCDs' L2s
                                // and are written back when the kernel ends -- ~1 us in front of the NEXT launch of a chain of dependent launches, which at the
                                // batch sizes this kernel serves is all the step is (all-workgroup stamps: 3.0 us between last exit and next entry).
                                // Same box, batch 16: RCAN 2,014 / 2,024 -> 2,168 / 2,176 patches/s, EDSR-baseline 19.25k / 19.29k -> 20.31k / 20.13k.
#endif

namespace {

struct PairCfg {
  static constexpr int NT = 512;
  static constexpr int TO = 14;                        // output tile edge
  static constexpr int XT = 18, XP = 18;               // input tile edge / row pitch (pixels)
  static constexpr int MT = 16, MP = 18;               // intermediate tile edge / row pitch
  static constexpr int XS_BYTES = XT * XP * 128;       // 41,472
  static constexpr int MS_BYTES = MT * MP * 128;       // 36,864
  static constexpr int WG_BYTES = 3 * 8 * 64 * 16;     // one slab = 3 taps: 24,576
  static constexpr int LDS_BYTES = XS_BYTES + MS_BYTES + 3 * WG_BYTES;
  static constexpr int XPIECES = XT * XT * 8;          // 2,592
  static constexpr int XK = (XPIECES + NT - 1) / NT;   // 6 pieces per lane
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

SRK_DEV __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int DT>
__global__ __launch_bounds__(512) void conv_pair_kernel(const srk_conv_pair_args a, int tilesX, int tilesY, unsigned x_bytes) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  typedef PairCfg C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ms = smem + C::XS_BYTES;
  char* const Wr = smem + C::XS_BYTES + C::MS_BYTES;      // ring of 3 slabs

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
#if SRK_PAIR_STAMPS
  unsigned long long* const stamp = (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && (tid & 255) == 0) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.b2)) + (wave >> 2) * 32 : nullptr;
#define SRK_PSTAMP(i) do { if (stamp) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
  const float* const bias2 = nullptr;
  // every workgroup: entry / exit on the constant 100 MHz clock, [64 + (launch & 1) * 1024 + 2 * block + {0, 1}], launch index from an arrival counter at [63]
  unsigned long long* wg_slot = nullptr;
  {
    unsigned long long* const wgst = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.b2));
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    if (tid == 0 && nwg <= 512) {
      const unsigned long long c = atomicAdd(wgst + 63, 1ull);
      wg_slot = wgst + 64 + ((c / nwg) & 1) * 1024 + 2 * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
      wg_slot[0] = __builtin_amdgcn_s_memrealtime();
    }
  }
#else
#define SRK_PSTAMP(i) do { } while (0)
  const float* const bias2 = a.b2;
#endif
  SRK_PSTAMP(0);

  // both biases: one early load by waves 0 / 1, parked in LDS ahead of the first barrier -- the accumulators are initialised from
  // there (as global loads in front of each conv they were ~0.5k exposed cycles each in the channel-attention modes and before conv 2)
  __shared__ __attribute__((aligned(16))) float sbias[128];
  float bias_in = 0.f;
  if (tid < 64) { if (a.b1) bias_in = a.b1[tid]; }
  else if (tid < 128) { if (bias2) bias_in = bias2[tid - 64]; }
  // grid = (tilesX, tilesY, N): no integer division on the way to the first DMA (two scalar divisions cost ~0.8k cycles of the
  // ~1.6k a wave spent before its first transfer, tools/stamp_pair.py)
  const int tX = blockIdx.x, tY = blockIdx.y, n = blockIdx.z;
  const int bid = (n * tilesY + tY) * tilesX + tX;        // linear tile index (pool rows)
  const int y0 = tY * C::TO, x0 = tX * C::TO;            // output tile origin; intermediate origin (y0-1, x0-1); input (y0-2, x0-2)

  const elem* const xg = reinterpret_cast<const elem*>(a.x);
  const i32x4 xrsrc = make_rsrc4(xg, x_bytes);
  const i32x4 w1rsrc = make_rsrc4(a.w1, 9 * 8 * 64 * 16), w2rsrc = make_rsrc4(a.w2, 9 * 8 * 64 * 16);
  const unsigned xs_lds = lds_addr_of(Xs), wr_lds = lds_addr_of(Wr);

  // ---- prologue: slab a0, input tile, slabs a1 a2 -------------------------------------------------------------------------
  // slab g (0..5): taps 3*(g%3)..+2 of conv (g/3); 24 pieces of 1 KB; a straight copy of the packed layout.
  // Prologue form: 3 pieces per wave, all 8 waves (the fastest way to get 72 pieces out).
  auto dma_slab = [&](int g) {
    const i32x4 rs = g < 3 ? w1rsrc : w2rsrc;
    const int gg = g < 3 ? g : g - 3;
    const unsigned dst = wr_lds + (unsigned)((g % 3) * C::WG_BYTES);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int piece = wave * 3 + k;
      dma16_hidden(rs, (unsigned)(gg * C::WG_BYTES + piece * 1024 + lane * 16),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));
    }
  };
  // In-loop form: 6 pieces per wave, waves 4..7 only -- the compute waves issue NO transfer between their MFMAs (a piece costs
  // the issuing wave ~60 cycles next to MFMAs; 7 of them per hand-over stretched conv 1 to 41 cycles per MFMA against conv 2's 37)
  auto dma_slab_dw = [&](int g) {
    const i32x4 rs = g < 3 ? w1rsrc : w2rsrc;
    const int gg = g < 3 ? g : g - 3;
    const unsigned dst = wr_lds + (unsigned)((g % 3) * C::WG_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int piece = (wave - 4) * 6 + k;
      dma16_hidden(rs, (unsigned)(gg * C::WG_BYTES + piece * 1024 + lane * 16),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));
    }
  };
  const bool ca = a.ca_mode != 0;                          // the input is transformed on its way in (below): not by DMA
  // a0 first: it needs nothing but the weight pointer, so it flies while the tile's addresses are computed.  The CU takes one
  // 1 KB piece per ~16-24 cycles whoever asks (64 B/clk), so what is issued AHEAD of the first barrier is what the first MFMA
  // waits for: a0 and the tile (65 pieces) by all eight waves; a1 a2 (48 pieces, not needed before K-step 10) by the DMA waves
  // behind that barrier (tools/stamp_pair.py: with a1 a2 issued by every wave right behind its tile pieces the barrier was
  // passed at 3.6k cycles, the tile's last piece issued at 2.7k).
  if (!ca) {
    dma_slab(0);
    SRK_PSTAMP(1);
    unsigned xoff[C::XK];
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int i = tid + k * C::NT;
      const int sl = i & 7, p = i >> 3;
      const int iy = p / C::XT, ix = p - iy * C::XT;
      const int c = sl ^ swz(ix);
      const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      xoff[k] = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + c * Tr::CH) * (int)sizeof(elem)) : 0x80000000u;
    }
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      if (k * C::NT + wave * 64 < C::XPIECES) {             // wave-uniform
        // the last 1 KB piece is half tile: its upper lanes are switched off (EXEC), they would land on the intermediate tile
        if (tid + k * C::NT < C::XPIECES)
          dma16_hidden(xrsrc, xoff[k], (unsigned)__builtin_amdgcn_readfirstlane((int)(xs_lds + ((k * C::NT + wave * 64) << 4))));
      }
    }
  }
  SRK_PSTAMP(2);

  // ---- per-lane constants -----------------------------------------------------------------------------------------------
  // Waves 0..3 compute (one per SIMD), each a 64-channel x 64-pixel tile = pixel blocks 2w, 2w+1 (tile rows 4w .. 4w+3):
  // 2 weight + 2 pixel fragments per K-step feed 4 MFMAs, the 1:1 read:MFMA ratio at which the CU's 128 B/clk of LDS
  // keeps up (32-pixel wave tiles on all 8 waves need 1.5 reads per MFMA and ran LDS-bound).  Waves 4..7 only issue DMA.
  const bool cw = wave < 4;
  const int px = r & 15;
  int gsw[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) gsw[kw] = swz(px + kw);
  const char* const wlane = Wr + ((h * 64 + r) << 4);
  int prow[2];
  const char* xl[2];
  const char* ml[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    prow[pb] = 4 * (wave & 3) + 2 * pb + (r >> 4);
    xl[pb] = Xs + ((prow[pb] * C::XP + px) << 7);
    ml[pb] = Ms + ((prow[pb] * C::MP + px) << 7);
  }

  // copies between an LDS tile and HBM (8 lanes = one 128-byte pixel, consecutive lane octets = consecutive pixels): pixels 2 and 3
  // of every four swap places.  A ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: pixel
  // octets 0 and 2 (same 128-byte half of the bank row) then ask for chunks 0-3 and 4-7 ^ swizzle, which collide whenever the
  // swizzles differ in bit 2 -- always, two columns apart.  Swapped, octets 0 and 3 / 1 and 2 meet instead (opposite halves).
  // tools/lds_conflicts_pair.py: 206 -> 34 conflict cycles per workgroup in the copy loops.
  auto pswap = [](int p) { return p ^ ((p >> 1) & 1); };

  f32x16 acc[2][2];                                       // [channel block][pixel block]
  auto init_acc = [&](int which) {                       // 0: conv 1, 1: conv 2
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sbias + which * 64 + 4 * h + cb * 32 + 8 * i);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          acc[cb][pb][4 * i + 0] = b.x; acc[cb][pb][4 * i + 1] = b.y; acc[cb][pb][4 * i + 2] = b.z; acc[cb][pb][4 * i + 3] = b.w;
        }
      }
  };
  if (!ca && tid < 128) sbias[tid] = bias_in;
  if (ca) {
    __shared__ __attribute__((aligned(16))) float cW1[512], cW2[512], cA[64], cB[64], cmean[64], cd2[64], cz[8], cd1[8];
    const bool bwd = a.ca_mode == 1;
    // workgroup barrier for LDS traffic only: __syncthreads() would also wait for the weight slabs in flight (vmcnt)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int Cr = a.ca_cr;                                  // <= 8 (launcher)
    const float invHW = 1.f / (float)(H * W);
    const bool slot_owner = tX == 0 && tY == 0;             // one workgroup per sample writes the per-sample results
    // The MLP's operands FIRST: they head the longest chain of this prologue (operands -> staging -> pooled means -> squeeze / excite ->
    // scale -> tile transform); the tile's own loads follow and stay in flight behind the staging barrier (requested ahead of the
    // operands, their 12 vector-memory instructions per wave delayed the chain by ~0.7k cycles).
    // The per-block partial sums of the pooled vectors ([rows <= 64][64] floats each): 16-byte loads spread over the whole
    // workgroup -- a CU issues roughly one vector-memory instruction per 16 cycles whoever asks, so their NUMBER is what
    // costs -- staged raw in the (still unused) intermediate tile's LDS
    const bool want_s = !bwd || slot_owner;
    const int rg = bwd ? a.ca_gsum_rows : 0, rs = want_s ? a.ca_sums_rows : 0;
    float* const rawG = reinterpret_cast<float*>(Ms);
    float* const rawS = rawG + 64 * 64;
    f32x4 vg[2], vs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + C::NT * u;                         // float4 index: row f >> 4, channels 4 (f & 15) ..
      vg[u] = f < rg * 16 ? *reinterpret_cast<const f32x4*>(a.ca_gsum + (size_t)n * rg * 64 + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
      vs[u] = f < rs * 16 ? *reinterpret_cast<const f32x4*>(a.ca_sums + (size_t)n * rs * 64 + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float w1r = 0.f, w2r = 0.f, sg_in = 0.f, z_in = 0.f, b1_in = 0.f, b2_in = 0.f;
    if (tid < 64 * Cr) { w1r = a.ca_w1[tid]; w2r = a.ca_w2[tid]; }
    if (bwd) {
      if (tid < 64) sg_in = a.ca_s[(size_t)n * 64 + tid];
      if (tid < Cr) z_in = a.ca_z[(size_t)n * Cr + tid];
    } else {
      if (tid < Cr) b1_in = a.ca_b1[tid];
      if (tid < 64) b2_in = a.ca_b2[tid];
    }
    SRK_PSTAMP(13);
    // input pieces (and, forward, the second operand) through registers.  Lane `tid` carries CHUNK tid & 7 of pixels (tid >> 3) +
    // 64 k (eight lanes = one 128-byte pixel, in order) and writes it to the swizzled slot: the eight scale / shift values of a
    // lane are then the same for all its pieces (round 4 mapped lanes to SLOTS: four LDS reads of s / dmean per piece).
    u32x4_t xin[C::XK], x2in[C::XK] = {};
    unsigned xdst[C::XK];
    unsigned okbits = 0;
    const int cch = tid & 7;
    {
      const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x);
      const __amdgpu_buffer_rsrc_t r2 = rsrc_of(bwd ? a.x : a.ca_x2);
#pragma unroll
      for (int k = 0; k < C::XK; ++k) {
        const int p = (tid >> 3) + 64 * k;
        const int iy = p / C::XT, ix = p - iy * C::XT;
        const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
        const bool ok = p < C::XT * C::XT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int pix = (n * H + gy) * W + gx;
        okbits |= ok ? (1u << k) : 0u;
        xdst[k] = (unsigned)((p << 7) + ((cch ^ swz(ix)) << 4));
        xin[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)((pix * a.x_pitch + a.x_coff + cch * Tr::CH) * 2) : 0x80000000u, 0, 0);
        if (!bwd)
          x2in[k] = __builtin_amdgcn_raw_buffer_load_b128(r2, ok ? (unsigned)((pix * a.ca_x2_pitch + a.ca_x2_coff + cch * Tr::CH) * 2) : 0x80000000u, 0, 0);
      }
    }
    SRK_PSTAMP(14);
    // the MLP's operands are used (= waited for) HERE; the tile's pieces, requested after them, keep flying (the counter retires in
    // order: their wait at the transform also covers the weight slabs waves 1..7 request below, which have landed by then)
    asm volatile("" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vs[0]), "+v"(vs[1]), "+v"(w1r), "+v"(w2r), "+v"(sg_in), "+v"(z_in), "+v"(b1_in), "+v"(b2_in), "+v"(bias_in));
    SRK_PSTAMP(15);
    // (everything beyond the real rows / hidden units is staged as ZERO, so that the sums below run over fixed ranges with
    // no per-element branch: a +-0 term leaves a sum as it is)
    cW1[tid] = w1r;
    cW2[tid] = w2r;
    if (tid < 128) sbias[tid] = bias_in;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + C::NT * u;
      *reinterpret_cast<f32x4*>(rawG + 4 * f) = vg[u];
      *reinterpret_cast<f32x4*>(rawS + 4 * f) = vs[u];
    }
    lds_barrier();
    SRK_PSTAMP(19);
    // slabs a0 a1 a2 (72 pieces) by waves 1..7 only, AFTER the staging barrier: wave 0 carries the MLP below, the longest chain
    // of this prologue, and must not wait for their issue (~1.3k cycles); the transfers land well before the tile transform ends
    if (wave) {
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const int piece = (wave - 1) + 7 * k;                // 0 .. 71 = slab piece / 24, piece % 24
        if (piece < 72)
          dma16_hidden(w1rsrc, (unsigned)(piece * 1024 + lane * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(wr_lds + piece * 1024)));
      }
    }
    // sum of the partials of channel c the way ca_sum_partials forms it: four strided sums (rows q, q+4, ...), then those in order
    // (all <= 64 operands are read before the first add: a dependent LDS read per add costs ~100 cycles each; rows beyond
    // `rows` contribute +0.0, which leaves every partial sum as it is)
    auto pooled = [&](const float* raw, int rows, int c) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += 4) {                   // 16 rows at a time (rows q + 4k, k = k0 .. k0+3), reads before adds
        if (4 * k0 < rows) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = raw[(4 * k0 + i) * 64 + c];
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += v[4 * k + q];
        }
      }
      return ((t[0] + t[1]) + t[2]) + t[3] + 0.f;
    };
    // The MLP itself on wave 0 alone (LDS operations of one wave execute in order, so its stages need no workgroup barrier):
    // lane = channel; the 64-term sums are wave butterflies (wave_sum64, srk_common.h: the order of the stand-alone kernels
    // for 64 channels), their results uniform over the wave, so the hidden units live in registers of every lane.  The NJ
    // butterflies are INDEPENDENT chains in one basic block (no per-unit branch): their DPP stages interleave instead of
    // each waiting out its own read-after-write distance.  Units beyond Cr have zero weights and biases: z = 0, +0.0 terms.
    auto mlp = [&](auto njc) {
      constexpr int NJ = decltype(njc)::value;
      // column `lane` of W1 [Cr][64], row `lane` of W2 [64][Cr] into registers, all reads issued at once (round 4 read one element
      // per hidden unit inside the dependent chain: ~500 cycles per unit)
      float w1c[NJ], w2c[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) { w1c[j] = j < Cr ? cW1[j * 64 + lane] : 0.f; w2c[j] = j < Cr ? cW2[lane * Cr + j] : 0.f; }
      const float mean_c = pooled(rawS, rs, lane) * invHW;
      cmean[lane] = mean_c;
      SRK_PSTAMP(20);
      float zz[8] = {}, dd[8] = {};
      if (bwd) {
        const float u = pooled(rawG, rg, lane);
        const float d2 = u * (sg_in * (1.f - sg_in));
        cA[lane] = sg_in;
        cd2[lane] = d2;
        // dz[j] = sum_c W2[c][j] dpre2[c], dpre1 = relu'(z) dz; dmean[c] = sum_j W1[j][c] dpre1[j] (j ascending)
        float pr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) pr[j] = w2c[j] * d2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) pr[j] = wave_sum64(pr[j]);
        float dm = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          zz[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z_in), j));
          dd[j] = zz[j] > 0.f ? pr[j] : 0.f;
          if (j < Cr) dm += w1c[j] * dd[j];
        }
        cB[lane] = dm / (float)(H * W);
      } else {
        // forward: z = relu(b1 + W1 mean), s = sigmoid(b2 + W2 z) (j ascending)
        float pr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) pr[j] = w1c[j] * mean_c;
#pragma unroll
        for (int j = 0; j < NJ; ++j) pr[j] = wave_sum64(pr[j]);
        float sg = b2_in;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          zz[j] = relu_f32(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(b1_in), j)) + pr[j]);
          if (j < Cr) sg += w2c[j] * zz[j];
        }
        sg = 1.f / (1.f + expf(-sg));
        cA[lane] = sg;
        cB[lane] = 0.f;
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cz[j] = zz[j]; cd1[j] = dd[j]; }
      }
      SRK_PSTAMP(21);
    };
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);                         // the chain everyone waits for: first pick over its SIMD partner (a DMA wave issuing slab pieces)
      if (Cr <= 4) mlp(std::integral_constant<int, 4>());
      else mlp(std::integral_constant<int, 8>());
      __builtin_amdgcn_s_setprio(0);
    }
    lds_barrier();
    if (slot_owner) {                                        // per-sample results: one workgroup per sample
      if (bwd && a.ca_slots) {
        float* const sl0 = a.ca_slots + (size_t)n * (2 * 64 * Cr + Cr + 64);      // [dW1 | db1 | dW2 | db2]
        float* const dw1 = sl0, *const db1 = sl0 + 64 * Cr, *const dw2 = db1 + Cr, *const db2 = dw2 + 64 * Cr;
        for (int i = tid; i < 64 * Cr; i += C::NT) {
          dw2[i] = cd2[i / Cr] * cz[i % Cr];
          dw1[i] = cd1[i / 64] * cmean[i % 64];
        }
        if (tid < 64) db2[tid] = cd2[tid];
        if (tid < Cr) db1[tid] = cd1[tid];
      }
      if (!bwd) {
        if (a.ca_z_out && tid < Cr) a.ca_z_out[(size_t)n * Cr + tid] = cz[tid];
        if (a.ca_s_out && tid < 64) a.ca_s_out[(size_t)n * 64 + tid] = cA[tid];
      }
    }
    SRK_PSTAMP(16);
    {
      // packed pairs: v * s + b as v_pk_mul_f32, v_pk_add_f32 (separate roundings, like the stand-alone kernels).  Forward: the
      // out-of-image pieces were loaded as zeros from both tensors and 0 * s + 0 = 0 needs no select; backward: dmean != 0 there.
      const f32x4 A0 = *reinterpret_cast<const f32x4*>(cA + 8 * cch), A1 = *reinterpret_cast<const f32x4*>(cA + 8 * cch + 4);
      const f32x2 Av[4] = {{A0.x, A0.y}, {A0.z, A0.w}, {A1.x, A1.y}, {A1.z, A1.w}};
      f32x2 Bv[4] = {};
      if (bwd) {
        const f32x4 B0 = *reinterpret_cast<const f32x4*>(cB + 8 * cch), B1 = *reinterpret_cast<const f32x4*>(cB + 8 * cch + 4);
        Bv[0] = f32x2{B0.x, B0.y}; Bv[1] = f32x2{B0.z, B0.w}; Bv[2] = f32x2{B1.x, B1.y}; Bv[3] = f32x2{B1.z, B1.w};
      }
#pragma unroll
      for (int k = 0; k < C::XK; ++k) {
        if ((tid >> 3) + 64 * k < C::XT * C::XT) {
          const bool ok = (okbits >> k) & 1;
          const uint32_t w4[4] = {xin[k].x, xin[k].y, xin[k].z, xin[k].w};
          const uint32_t r4[4] = {x2in[k].x, x2in[k].y, x2in[k].z, x2in[k].w};
          uint32_t o4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t0, t1, q0, q1;
            unpack2<DT>(w4[e], t0, t1);
            f32x2 v = {t0, t1}, b = Bv[e];
            if (!bwd) { unpack2<DT>(r4[e], q0, q1); b = f32x2{q0, q1}; }
            v = v * Av[e];
            v = v + b;
            const uint32_t pk = pack2<DT>(v.x, v.y);
            o4[e] = (!bwd || ok) ? pk : 0u;
          }
          lds_write16(Xs + xdst[k], i32x4{(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]});
        }
      }
    }
    SRK_PSTAMP(17);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SRK_PSTAMP(18);
  }

  // One conv = 36 K-steps (slab = kernel row kh = s / 12, tap column kw, K-step ks), 4 MFMAs each; the fragments of
  // step s+2 are read during step s, one ds_read_b128 per MFMA gap (a burst of four per gap oversubscribes the LDS
  // array while the four waves run in step).  `at_step(s)` runs the slab hand-over (barriers, waits, DMA) between steps.
  // SPLIT (round 5): the LAST kernel row (K-steps 24..35) runs pixel block 0 first (pass A: 24 MFMAs), then pixel block 1
  // (pass B: 24 MFMAs), and `epi(piece)`, piece 0..23, is issued behind the MFMAs of pass B: the epilogue of pixel block 0 --
  // pack, ReLU / mask, residual, the LDS stores: ~0.6k cycles of vector instructions per conv that used to follow the loop --
  // rides in the shadow of the matrix pipe.  Every accumulator still sees its K-steps in the same order (bit-identical
  // results); the last row reads its weight fragments twice (1.5 LDS reads per MFMA there instead of 1).
  auto conv_loop = [&](auto split_c, int g0, const char* const (&pix)[2], int pitch, auto&& at_step, auto&& epi) {
    constexpr bool SPLIT = decltype(split_c)::value;
    constexpr int NV = SPLIT ? 48 : 36;                    // virtual steps: 0..23 | pass A 24..35 | pass B 36..47 (K-steps 24..35 again)
    constexpr int PD = SRK_PAIR_PD;                        // fragments are read PD steps ahead (ring of PD + 1 sets)
    i32x4 fa[PD + 1][2], fb[PD + 1][2];
    auto needed = [&](int v, int q) { return v < NV && (!SPLIT || v < 24 || (v < 36 ? q != 3 : q != 2)); };
    auto frag1 = [&](int v, int q, i32x4 (&af)[2], i32x4 (&bf)[2]) {
      const int s = v < 36 ? v : v - 12;
      const int kh = s / 12, kw = (s % 12) / 4, ks = s % 4;
      if (q < 2) af[q] = lds_read16(wlane + ((g0 + kh) % 3) * C::WG_BYTES + (((kw * 8 + 2 * ks) * 64 + q * 32) << 4));
      else bf[q - 2] = lds_read16(pix[q - 2] + ((kh * pitch + kw) << 7) + (((2 * ks + h) ^ gsw[kw]) << 4));
    };
    // the MFMA gap of step v in which fragment q of step v + PD is read: one per gap in a 4-MFMA step; in a 2-MFMA step the
    // (at most three) fragments of step v + 2 go two into the first gap, one into the second
    auto gap_of = [&](int v, int q) {
      if (!needed(v + PD, q)) return -1;
      if (!SPLIT || v < 24) return q;
      int idx = 0;
      for (int qq = 0; qq < q; ++qq) idx += needed(v + PD, qq) ? 1 : 0;
      return idx >> 1;
    };
#pragma unroll
    for (int v0 = 0; v0 < PD; ++v0)
#pragma unroll
      for (int q = 0; q < 4; ++q) frag1(v0, q, fa[v0], fb[v0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v < 36) at_step(v);
      const int c0 = v % (PD + 1), c2 = (v + PD) % (PD + 1);
      if (!SPLIT || v < 24) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gap_of(v, q) == m) frag1(v + PD, q, fa[c2], fb[c2]);
          const int cb = m >> 1, pb = m & 1;
          acc[cb][pb] = Tr::mma(fa[c0][cb], fb[c0][pb], acc[cb][pb]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const int pb = v < 36 ? 0 : 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gap_of(v, q) == m) frag1(v + PD, q, fa[c2], fb[c2]);
          acc[m][pb] = Tr::mma(fa[c0][m], fb[c0][pb], acc[m][pb]);
          if (v >= 36) epi(2 * (v - 36) + m);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // every LDS read of this wave has returned, then the workgroup barrier: the ring slot / tile behind it may be rewritten
  auto drain_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  int mpix[2];
  bool m_in[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int my = y0 - 1 + prow[pb], mx = x0 - 1 + px;   // this lane's intermediate pixels in the image
    m_in[pb] = (unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W;
    mpix[pb] = (n * H + my) * W + mx;
  }
  // ReLU-backward mask: its 16x16 tile goes by LDS-DMA to WHERE THE INTERMEDIATE WILL BE WRITTEN -- each lane reads its own
  // mask chunks right before it overwrites them.  (Per-lane 16-byte loads of one pixel each occupy the address path for
  // ~64 cycles apiece: eight of them cost more than the tile transfer.)  32 pieces of 8 pixels, 8 per DMA wave (4..7).
  const unsigned ms_lds = lds_addr_of(Ms);
  auto dma_mask = [&]() {
    if (a.mask) {
      const i32x4 mrsrc = make_rsrc4(a.mask, 0x7fffffffu);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int blk = (wave - 4) + 4 * k;
        const int iy = blk >> 1, ix = (blk & 1) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(ix);
        const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.mask_pitch + a.mask_coff + c * Tr::CH) * 2) : 0x80000000u;
        dma16_hidden(mrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(ms_lds + ((iy * C::MP + (blk & 1) * 8) << 7))));
      }
    }
  };
  // Only the input tile and kernel row 0 of conv 1 (66 of the 115 KB) are requested and waited for before the first MFMA.
  SRK_PSTAMP(3);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the tile and a0 have landed; this wave's LDS stores (the biases) have completed
  SRK_PSTAMP(4);
  __builtin_amdgcn_s_barrier();
  SRK_PSTAMP(5);
  if (cw) init_acc(0);
  // a1 is needed from K-step 10 on (~1.5k cycles from here), a2 from K-step 22: requested NOW, not in the prologue, where every
  // piece issued ahead of the barrier delays it by the ~24 cycles the CU needs per piece
  if (!ca && wave >= 4) { dma_slab_dw(1); dma_slab_dw(2); }

  // ---- conv 1 on the 16x16 intermediate pixels ------------------------------------------------------------------------------
  // Hand-over at K-step 10 / 22 (fragments are read two steps ahead, so the reads of kernel row 0 / 1 have all been issued
  // and, after the drain, returned): kernel row 1 / 2 has landed (own prologue pieces by the counted wait, the others' by the
  // barrier) and ring slot 0 / 1 is free for conv 2's slab b0 / b1, which the DMA waves (4..7) then request.  The mask tile
  // is requested at the first hand-over, ahead of b0, so that the DMA waves' second wait (all but b0's six pieces) covers it.
  // vector-memory operations in flight, per wave, oldest first --
  //   compute waves: none after the prologue
  //   DMA waves:     [a1 a2 | 6 + 6] [mask 8] [b0 6] [b1 6]       hand 1: vmcnt(6)   hand 2: vmcnt(6) = a2 and the mask have landed
  // (channel-attention modes: nothing is in flight when conv 1 starts, the counts are merely stricter than needed)
  auto hand1_cw = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto hand2_cw = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto hand1_dw = [&]() { asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); dma_mask(); dma_slab_dw(3); };
  auto hand2_dw = [&]() { asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); dma_slab_dw(4); };
  // intermediate epilogue: [ReLU | mask], * scale_mid, zero outside the image (conv 2 pads the IMAGE with zeros), into the
  // LDS tile in the image format (this lane: 32 contiguous channels 32h .. 32h+31 of its pixel = chunks 4h .. 4h+3).
  // Pixel block 0's epilogue rides behind the MFMAs of pixel block 1's last kernel row (conv_loop, SPLIT) in the two common
  // flavours -- forward (ReLU, no scale, no mask) and backward (mask, scale) --, as 24 pieces of at most 6 vector instructions.
  const float sm = a.scale_mid;
  const f32x2 sm2 = {sm, sm};
  const int gmid = swz(px);
  auto mid_full = [&](int pb) {
    char* const mp = const_cast<char*>(ml[pb]);
    uint32_t P[16];
    if (a.relu_mid && sm == 1.f) {                         // forward: ReLU on the packed pairs (the form srk_conv2d uses)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int d = 0; d < 8; ++d) P[8 * cb + d] = relu_pk16<DT>(pack2<DT>(acc[cb][pb][2 * d], acc[cb][pb][2 * d + 1]));
    } else {
      f32x2 v[16];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int d = 0; d < 8; ++d) v[8 * cb + d] = f32x2{acc[cb][pb][2 * d], acc[cb][pb][2 * d + 1]};
      if (a.relu_mid) {
#pragma unroll
        for (int d = 0; d < 16; ++d) v[d] = f32x2{relu_f32(v[d].x), relu_f32(v[d].y)};
      }
#pragma unroll
      for (int d = 0; d < 16; ++d) { const f32x2 t = v[d] * sm2; P[d] = pack2<DT>(t.x, t.y); }
    }
    if (a.mask) {
      // on the PACKED results, three packed-integer instructions per two elements (mask_apply_pk16, the form the
      // weight-stationary kernel's data gradient uses; unpack + float compare + select was 7 per element pair)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const i32x4 m = lds_read16(mp + (((4 * h + j) ^ gmid) << 4));
        mask_apply_pk16(P[4 * j + 0], (uint32_t)m.x);
        mask_apply_pk16(P[4 * j + 1], (uint32_t)m.y);
        mask_apply_pk16(P[4 * j + 2], (uint32_t)m.z);
        mask_apply_pk16(P[4 * j + 3], (uint32_t)m.w);
      }
    }
    // pixels outside the image: conv 2 pads the IMAGE with zeros.  Interior tiles (most) have none: one uniform test
    if (__builtin_amdgcn_ballot_w64(!m_in[pb]) != 0) {
#pragma unroll
      for (int d = 0; d < 16; ++d) P[d] = m_in[pb] ? P[d] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      lds_write16(mp + (((4 * h + j) ^ gmid) << 4), i32x4{(int)P[4 * j], (int)P[4 * j + 1], (int)P[4 * j + 2], (int)P[4 * j + 3]});
  };
  // the same arithmetic for pixel block 0, cut into the pieces conv_loop places behind pass B's MFMAs
  uint32_t P0[16];
  i32x4 mreg[4];
  char* const mp0 = const_cast<char*>(ml[0]);
  auto mid_piece_relu = [&](int p) {                       // 16 x (pack, ReLU, zero outside the image), 4 stores
    if (p < 16) {
      const int cb = p >> 3, d = p & 7;
      const uint32_t w = relu_pk16<DT>(pack2<DT>(acc[cb][0][2 * d], acc[cb][0][2 * d + 1]));
      P0[p] = m_in[0] ? w : 0u;
    } else if (p < 20) {
      const int j = p - 16;
      lds_write16(mp0 + (((4 * h + j) ^ gmid) << 4), i32x4{(int)P0[4 * j], (int)P0[4 * j + 1], (int)P0[4 * j + 2], (int)P0[4 * j + 3]});
    }
  };
  auto mid_piece_mask = [&](int p) {                       // 4 mask chunks, 16 x (scale, pack, mask, zero outside), 4 stores
    if (p < 4) {
      mreg[p] = lds_read16(mp0 + (((4 * h + p) ^ gmid) << 4));
    } else if (p < 20) {
      const int e = p - 4, cb = e >> 3, d = e & 7;
      const f32x2 t = f32x2{acc[cb][0][2 * d], acc[cb][0][2 * d + 1]} * sm2;
      uint32_t w = pack2<DT>(t.x, t.y);
      const i32x4 m = mreg[e >> 2];
      mask_apply_pk16(w, (uint32_t)((e & 3) == 0 ? m.x : (e & 3) == 1 ? m.y : (e & 3) == 2 ? m.z : m.w));
      P0[e] = m_in[0] ? w : 0u;
    } else {
      const int j = p - 20;
      lds_write16(mp0 + (((4 * h + j) ^ gmid) << 4), i32x4{(int)P0[4 * j], (int)P0[4 * j + 1], (int)P0[4 * j + 2], (int)P0[4 * j + 3]});
    }
  };
  auto hand_cw = [&](int s) {
    if (s == 12 - SRK_PAIR_PD) hand1_cw();                 // the last step whose prefetch still reads kernel row 0 / 1 has issued its reads
    if (s == 24 - SRK_PAIR_PD) hand2_cw();
  };
  const std::integral_constant<bool, true> split_on;
  const std::integral_constant<bool, false> split_off;
  if (cw) {
    if (a.relu_mid && sm == 1.f && !a.mask) {
      conv_loop(split_on, 0, xl, C::XP, hand_cw, mid_piece_relu);
      SRK_PSTAMP(6);
      mid_full(1);
    } else if (a.mask && !a.relu_mid) {
      conv_loop(split_on, 0, xl, C::XP, hand_cw, mid_piece_mask);
      SRK_PSTAMP(6);
      mid_full(1);
    } else {
      conv_loop(split_off, 0, xl, C::XP, hand_cw, [](int) {});
      SRK_PSTAMP(6);
      mid_full(0);
      mid_full(1);
    }
  } else {
    hand1_dw();
    hand2_dw();
    // transformed input: this workgroup's 14x14 to HBM, whole 128-byte pixels (the input tile stays until conv 1 is done)
    if (ca && a.xo) {
      const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.xo);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int i = (tid - 256) + 256 * k;
        const int p = pswap(i >> 3), c = i & 7;
        const int row = p / C::TO, col = p - row * C::TO;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = i < C::TO * C::TO * 8 && gy < H && gx < W;
        const i32x4 q = lds_read16(Xs + (((row + 2) * C::XP + col + 2) << 7) + ((c ^ swz(col + 2)) << 4));
        const unsigned vo = ok ? (unsigned)((((n * H + gy) * W + gx) * a.xo_pitch + a.xo_coff + c * Tr::CH) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, vo, 0, SRK_PAIR_ST_AUX);
      }
    }
  }

  if (!cw) SRK_PSTAMP(6);
  SRK_PSTAMP(7);
  // ONE barrier between the convs.  Behind it: the intermediate tile is written, the input tile and ring slot 2 are free, and
  // slabs b0 b1 have landed -- the DMA waves asked for them 5k and 2.5k cycles ago and wait for everything they have in flight
  // (also their copy of the transformed input) before they arrive.  They then request b2 (+ an external residual tile, in the
  // input tile's place), which conv 2 needs from K-step 22 on, while the compute waves are already in conv 2.
  if (!cw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  drain_barrier();
  SRK_PSTAMP(8);
  // a residual that is not the input: its tile replaces the input tile (28 pieces of 8 pixels: rows 2..15, columns 2..17)
  const bool xs_res = a.res && !a.res_from_x;
  if (!cw) {
    dma_slab_dw(5);
    if (xs_res) {
      const i32x4 rrsrc = make_rsrc4(a.res, 0x7fffffffu);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int blk = (wave - 4) + 4 * k;                  // 0..27
        const int iy = 2 + (blk >> 1), ix = 2 + (blk & 1) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(ix);
        const int gy = y0 + iy - 2, gx = x0 + ix - 2;
        const bool ok = gy < H && gx < W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.res_pitch + a.res_coff + c * Tr::CH) * 2) : 0x80000000u;
        dma16_hidden(rrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(xs_lds + ((iy * C::XP + 2 + (blk & 1) * 8) << 7))));
      }
    }
  }

  // ---- conv 2 on the 14x14 output tile: rows 0..13 x 16 column slots (14 used).  Wave 3's second pixel block is rows
  // 14, 15: computed (it reads past the intermediate tile into the weight ring, inside the allocation) and dropped. ----------
  SRK_PSTAMP(9);
  init_acc(1);
  // slabs b0 b1 have landed (barrier above); b2 and an external residual tile are needed from step 22 on (fragments are read two
  // steps ahead): the DMA waves wait for them, the barrier tells the compute waves
  // pooling with a second factor (pool_aux): this lane's four 16-byte pieces of it, requested once nothing else is in
  // flight (conv-2 hand-over at step 22) and used when the output tile leaves
  // the output tile's way out (after conv 2): lane `tid` copies pieces i = tid + 512 k: chunk c = tid & 7 of pixels pswap((tid >> 3)
  // + 64 k) -- the same 8 channels each time.  Addresses now, while this wave waits at a barrier anyway, not in the kernel's tail.
  unsigned out_lds[4], out_off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + C::NT * k;
    const int p = pswap(i >> 3), c = i & 7;
    const int row = p / C::TO, col = p - row * C::TO;
    const int gy = y0 + row, gx = x0 + col;
    const bool ok = i < C::TO * C::TO * 8 && gy < H && gx < W;
    out_lds[k] = (unsigned)((((row + 2) * C::XP + col + 2) << 7) + ((c ^ swz(col + 2)) << 4));
    out_off[k] = ok ? (unsigned)((((n * H + gy) * W + gx) * a.out_pitch + a.out_coff + c * Tr::CH) * 2) : 0x80000000u;
  }
  u32x4_t au[4] = {};
  auto load_aux = [&]() {
    if (a.pool && a.pool_aux) {
      const __amdgpu_buffer_rsrc_t ra = rsrc_of(a.pool_aux);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + C::NT * k;
        const int p = pswap(i >> 3), c = i & 7;
        const int row = p / C::TO, col = p - row * C::TO;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = i < C::TO * C::TO * 8 && gy < H && gx < W;
        const unsigned vo = ok ? (unsigned)((((n * H + gy) * W + gx) * a.pool_aux_pitch + a.pool_aux_coff + c * Tr::CH) * 2) : 0x80000000u;
        au[k] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, 0, 0);
      }
    }
  };
  // ---- output epilogue: * scale_out, + residual (in the input tile's place either way), the result written back over the
  // residual it used; pixel block 0's part behind the MFMAs of pixel block 1's last kernel row (conv_loop, SPLIT) ----------
  const float so = a.scale_out;
  const f32x2 so2 = {so, so};
  const int gout = swz(px + 2);
  auto out_full = [&](int pb) {
    f32x2 v[16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int d = 0; d < 8; ++d) v[8 * cb + d] = f32x2{acc[cb][pb][2 * d], acc[cb][pb][2 * d + 1]} * so2;
    char* const xr = Xs + (((prow[pb] + 2) * C::XP + px + 2) << 7);
    if (a.res) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const i32x4 q = lds_read16(xr + (((4 * h + j) ^ gout) << 4));
        const int qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f0, f1;
          unpack2<DT>((uint32_t)qw[e], f0, f1);
          v[4 * j + e] = v[4 * j + e] + f32x2{f0, f1};
        }
      }
    }
    if (prow[pb] < C::TO && px < C::TO) {                  // rows 14, 15 / columns 14, 15 are not part of the tile
#pragma unroll
      for (int j = 0; j < 4; ++j)
        lds_write16(xr + (((4 * h + j) ^ gout) << 4),
                    i32x4{(int)pack2<DT>(v[4 * j].x, v[4 * j].y), (int)pack2<DT>(v[4 * j + 1].x, v[4 * j + 1].y),
                          (int)pack2<DT>(v[4 * j + 2].x, v[4 * j + 2].y), (int)pack2<DT>(v[4 * j + 3].x, v[4 * j + 3].y)});
    }
  };
  uint32_t Q0[16];
  i32x4 rreg[4];
  char* const xr0 = Xs + (((prow[0] + 2) * C::XP + px + 2) << 7);
  const bool keep0 = prow[0] < C::TO && px < C::TO;
  auto out_piece_res = [&](int p) {                        // 4 residual chunks, 16 x (scale, + residual, pack), 4 stores
    if (p < 4) {
      rreg[p] = lds_read16(xr0 + (((4 * h + p) ^ gout) << 4));
    } else if (p < 20) {
      const int e = p - 4, cb = e >> 3, d = e & 7;
      const i32x4 q = rreg[e >> 2];
      float f0, f1;
      unpack2<DT>((uint32_t)((e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w), f0, f1);
      const f32x2 t = f32x2{acc[cb][0][2 * d], acc[cb][0][2 * d + 1]} * so2 + f32x2{f0, f1};
      Q0[e] = pack2<DT>(t.x, t.y);
    } else if (keep0) {
      const int j = p - 20;
      lds_write16(xr0 + (((4 * h + j) ^ gout) << 4), i32x4{(int)Q0[4 * j], (int)Q0[4 * j + 1], (int)Q0[4 * j + 2], (int)Q0[4 * j + 3]});
    }
  };
  auto out_piece_plain = [&](int p) {                      // 16 x (scale, pack), 4 stores
    if (p < 16) {
      const int cb = p >> 3, d = p & 7;
      const f32x2 t = f32x2{acc[cb][0][2 * d], acc[cb][0][2 * d + 1]} * so2;
      Q0[p] = pack2<DT>(t.x, t.y);
    } else if (p < 20 && keep0) {
      const int j = p - 16;
      lds_write16(xr0 + (((4 * h + j) ^ gout) << 4), i32x4{(int)Q0[4 * j], (int)Q0[4 * j + 1], (int)Q0[4 * j + 2], (int)Q0[4 * j + 3]});
    }
  };
  auto hand_c2 = [&](int s) {
    if (s == 24 - SRK_PAIR_PD) { __builtin_amdgcn_s_barrier(); load_aux(); }
  };
  if (cw) {
    if (a.res) conv_loop(split_on, 3, ml, C::MP, hand_c2, out_piece_res);
    else conv_loop(split_on, 3, ml, C::MP, hand_c2, out_piece_plain);
    SRK_PSTAMP(10);
    out_full(1);
  } else {
    // the idle waves copy this workgroup's 14x14 of the intermediate to HBM while conv 2 runs: whole 128-byte pixels,
    // 8 lanes each, neighbouring pixels of a row contiguous
    if (a.mid) {
      const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.mid);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int i = (tid - 256) + 256 * k;
        const int p = pswap(i >> 3), c = i & 7;
        const int row = p / C::TO, col = p - row * C::TO;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = i < C::TO * C::TO * 8 && gy < H && gx < W;
        const i32x4 q = lds_read16(Ms + (((row + 1) * C::MP + col + 1) << 7) + ((c ^ swz(col + 1)) << 4));
        const unsigned vo = ok ? (unsigned)((((n * H + gy) * W + gx) * a.mid_pitch + a.mid_coff + c * Tr::CH) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, vo, 0, SRK_PAIR_ST_AUX);
      }
    }
    // b2 (+ residual tile) have landed; the copy's seven stores (the youngest operations) may still be on their way -- waiting for
    // them too made the compute waves wait at K-step 22 (conv 2 at 40 cycles per MFMA against conv 1's 37.6)
    if (a.mid) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_aux();
  }

  if (!cw) SRK_PSTAMP(10);
  // all eight waves copy the 14x14 tile to HBM in whole pixels
  drain_barrier();
  {
    const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out);
    float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = out_off[k] != 0x80000000u;
      const i32x4 q = lds_read16(Xs + out_lds[k]);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, out_off[k], 0, SRK_PAIR_ST_AUX);
      if (a.pool && ok) {
        const int qw[4] = {q.x, q.y, q.z, q.w};
        const uint32_t aw[4] = {au[k].x, au[k].y, au[k].z, au[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0, v1, t0 = 1.f, t1 = 1.f;
          unpack2<DT>((uint32_t)qw[e], v0, v1);
          if (a.pool_aux) unpack2<DT>(aw[e], t0, t1);
          ps[2 * e] += v0 * t0;
          ps[2 * e + 1] += v1 * t1;
        }
      }
    }
    if (a.pool) {
      // per-lane partials -> LDS (the intermediate tile's place), summed per channel over the 64 lanes that carry it in two
      // steps of eight, each in a fixed order (bitwise reproducible).  Row pitch NT + 4 words: the 32 lanes of a ds_read_b32
      // group (chunks 0..3 or 4..7 x 8 elements) then hit 32 different banks (pitch NT: 8-way conflicts, and ONE wave walked
      // all 64 terms behind a barrier that also waited for the tile's stores: ~2k cycles at the end of every RCAB launch).
      constexpr int PP = C::NT + 4;
      float* const P = reinterpret_cast<float*>(Ms);
      float* const P2 = P + 8 * PP;
#pragma unroll
      for (int e = 0; e < 8; ++e) P[e * PP + tid] = ps[e];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the stores above stay in flight
      {
        const int ch = tid & 63, part = tid >> 6;            // channel = chunk ch >> 3, element ch & 7; lanes 8 (8 part + j) + chunk
        const float* const src = P + (ch & 7) * PP + (ch >> 3) + 64 * part;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[8 * j];
        float u = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) u += v[j];
        P2[part * 64 + ch] = u;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = P2[j * 64 + tid];
        float u = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) u += v[j];
        a.pool[(size_t)bid * 64 + tid] = u;
      }
    }
  }
#if SRK_PAIR_STAMPS
  SRK_PSTAMP(11);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SRK_PSTAMP(12);
  if (wg_slot) wg_slot[1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

extern "C" int srk_conv_pair_tiles(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const long long t = (long long)N * ((H + PairCfg::TO - 1) / PairCfg::TO) * ((W + PairCfg::TO - 1) / PairCfg::TO);
  return t > 0x7fffffffLL ? 0x7fffffff : (int)t;
}

extern "C" int srk_conv_pair(const srk_conv_pair_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->w1 && a->w2 && a->out, "srk_conv_pair: null pointer");
  SRK_CHECK_ARG(a->dtype == SRK_BF16 || a->dtype == SRK_F16, "srk_conv_pair: 16-bit dtypes only");
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "srk_conv_pair: bad dims N=%d H=%d W=%d", a->N, a->H, a->W);
  SRK_CHECK_ARG(a->x_pitch % 8 == 0 && a->x_coff % 8 == 0 && a->out_pitch % 8 == 0 && a->out_coff % 8 == 0, "srk_conv_pair: 16-byte alignment of x / out");
  SRK_CHECK_ARG(!a->mid || (a->mid_pitch % 8 == 0 && a->mid_coff % 8 == 0), "srk_conv_pair: alignment of mid");
  SRK_CHECK_ARG(!a->mask || (a->mask_pitch % 8 == 0 && a->mask_coff % 8 == 0), "srk_conv_pair: alignment of mask");
  SRK_CHECK_ARG(!a->res || (a->res_pitch % 8 == 0 && a->res_coff % 8 == 0), "srk_conv_pair: alignment of res");
  SRK_CHECK_ARG(!(a->relu_mid && a->mask), "srk_conv_pair: relu_mid and mask are exclusive");
  SRK_CHECK_ARG(!a->pool_aux || (a->pool && a->pool_aux_pitch % 8 == 0 && a->pool_aux_coff % 8 == 0), "srk_conv_pair: pool_aux needs pool and 16-byte alignment");
  SRK_CHECK_ARG(!a->res_from_x || (a->res == a->x && a->res_pitch == a->x_pitch && a->res_coff == a->x_coff),
                "srk_conv_pair: res_from_x needs res to BE x");
  SRK_CHECK_ARG(a->ca_mode >= 0 && a->ca_mode <= 2, "srk_conv_pair: ca_mode %d", a->ca_mode);
  if (a->ca_mode) {
    SRK_CHECK_ARG(!a->res_from_x, "srk_conv_pair: the LDS input tile is transformed in ca_mode, a residual comes from memory");
    SRK_CHECK_ARG(a->ca_mode != 1 || (a->ca_gsum && a->ca_sums && a->ca_s && a->ca_z && a->ca_w1 && a->ca_w2 && a->ca_gsum_rows > 0 && a->ca_sums_rows > 0),
                  "srk_conv_pair: ca_mode 1 needs gsum, sums, s, z, w1, w2");
    SRK_CHECK_ARG(a->ca_mode != 2 || (a->ca_x2 && a->ca_sums && a->ca_w1 && a->ca_w2 && a->ca_b1 && a->ca_b2 && a->ca_sums_rows > 0 &&
                                      a->ca_x2_pitch % 8 == 0 && a->ca_x2_coff % 8 == 0),
                  "srk_conv_pair: ca_mode 2 needs x2, sums, w1, b1, w2, b2");
    SRK_CHECK_ARG(a->ca_cr > 0 && a->ca_cr <= 8, "srk_conv_pair: ca_cr=%d (1..8)", a->ca_cr);
    SRK_CHECK_ARG(a->ca_sums_rows <= 64 && a->ca_gsum_rows <= 64, "srk_conv_pair: more than 64 partial rows per sample");
    SRK_CHECK_ARG(!a->xo || (a->xo_pitch % 8 == 0 && a->xo_coff % 8 == 0), "srk_conv_pair: alignment of xo");
  }
  const long long px = (long long)a->N * a->H * a->W;
  long long mx = px * a->x_pitch;
  if (px * a->out_pitch > mx) mx = px * a->out_pitch;
  if (a->mid && px * a->mid_pitch > mx) mx = px * a->mid_pitch;
  if (a->mask && px * a->mask_pitch > mx) mx = px * a->mask_pitch;
  if (a->res && px * a->res_pitch > mx) mx = px * a->res_pitch;
  if (a->pool_aux && px * a->pool_aux_pitch > mx) mx = px * a->pool_aux_pitch;
  if (a->ca_mode && a->xo && px * a->xo_pitch > mx) mx = px * a->xo_pitch;
  if (a->ca_mode == 2 && px * a->ca_x2_pitch > mx) mx = px * a->ca_x2_pitch;
  SRK_CHECK_ARG(mx * 2 < 0x7fff0000LL, "srk_conv_pair: tensors of 2 GiB and more are not supported (small-batch kernel)");
  typedef PairCfg C;
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<SRK_BF16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<SRK_F16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  if (attr0 != hipSuccess || attr1 != hipSuccess) {
    srk_set_error("srk_conv_pair: cannot reserve %d bytes of LDS", C::LDS_BYTES);
    return (int)(attr0 != hipSuccess ? attr0 : attr1);
  }
  const int tilesX = (a->W + C::TO - 1) / C::TO, tilesY = (a->H + C::TO - 1) / C::TO;
  const long long nb = (long long)a->N * tilesX * tilesY;
  SRK_CHECK_ARG(nb <= 0x7fffffffLL, "srk_conv_pair: %lld tiles", nb);
  const unsigned xb = (unsigned)(px * a->x_pitch * 2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  SRK_CHECK_ARG(tilesY <= 65535 && a->N <= 65535, "srk_conv_pair: grid (%d, %d, %d)", tilesX, tilesY, a->N);
  const dim3 grid((unsigned)tilesX, (unsigned)tilesY, (unsigned)a->N);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL((conv_pair_kernel<SRK_BF16>), grid, dim3(C::NT), C::LDS_BYTES, st, *a, tilesX, tilesY, xb);
  else hipLaunchKernelGGL((conv_pair_kernel<SRK_F16>), grid, dim3(C::NT), C::LDS_BYTES, st, *a, tilesX, tilesY, xb);
  SRK_LAUNCH_CHECK();
  return 0;
}
