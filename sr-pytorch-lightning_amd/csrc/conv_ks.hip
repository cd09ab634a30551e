// 3x3 convolution with MANY input channels (Cin = 64 k, k >= 2: EDSR-large's 256 -> 256 layers, models/edsr.py with
// n_feats = 256; RDN's dense layers, models/rdn.py:9-40): forward and, with data-gradient packs, dgrad.
//
// The weight-stationary kernel (conv_igemm.hip) keeps one 64-channel input block's weights in LDS; with 256 input channels
// a 64-output-channel block alone is 295 KB of weights, and the streaming kernel that handles it re-stages a 41 KB halo tile
// AND its weights per 64-channel block through one LDS buffer pair, at 24 % of the MFMA peak.  This kernel uses the machinery
// of conv_pair.hip instead: a workgroup owns one 16x16 output tile x 64 output channels and walks the input channels in
// blocks of 64; per block an 18x18 halo tile (double-buffered) and three weight slabs of one kernel row each (24 KB) that
// stream through a 3-slot ring by LDS-DMA, two slabs ahead of the MFMAs -- the MFMA loop never stops between blocks.
//   LDS (156,672 B): 2 x 41,472 halo tiles + 3 x 24,576 ring.  DMA per K-block: 41 + 74 KB per 5.5k cycles of MFMA = 21 B/clk
//   (the CU accepts ~31).  Waves 0-3 compute, one per SIMD, a 64-channel x 64-pixel tile each (2 weight + 2 pixel fragments
//   per K-step feed 4 MFMAs: the 1:1 read:MFMA ratio LDS sustains); waves 4-7 only issue DMA.
// The residual / ReLU-mask tile is fetched by LDS-DMA into the halo buffer that is idle during the last K-block; results are
// written back over it and all eight waves copy the tile to HBM in whole 128-byte pixels (no per-lane 16-byte global access).
// Epilogue arithmetic and order are srk_conv2d's: v = acc + bias; relu; * scale; + res; mask.
#include "srk_common.h"

namespace {

struct KsCfg {
  static constexpr int NT = 512;
  static constexpr int T = 16;                          // output tile edge
  static constexpr int XT = 18;                         // halo tile edge = row pitch (pixels)
  static constexpr int XS_BYTES = XT * XT * 128;        // 41,472
  static constexpr int WG_BYTES = 3 * 8 * 64 * 16;      // one slab = 3 taps x 64 input channels x 64 rows: 24,576
  static constexpr int LDS_BYTES = 2 * XS_BYTES + 3 * WG_BYTES;
  static constexpr int XPIECES = XT * XT * 8;           // 2,592
  static constexpr int DT_ = 256;                       // threads that issue the transfers: waves 4-7
  static constexpr int XK = (XPIECES + DT_ - 1) / DT_;  // 11 halo pieces per DMA-wave lane (the last one: 32 lanes of wave 4)
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// diagnostics build only (make stamp: -DSRK_KS_STAMPS=1, tools/stamp_ks.py): s_memtime of workgroup 0's waves 0 (compute) and 4 (DMA)
#if SRK_KS_STAMPS
__device__ unsigned long long ks_stamp_buf[2][32];
#define KS_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (i) < 32) ks_stamp_buf[threadIdx.x >> 8][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KS_STAMP(i) do { } while (0)
#endif

SRK_DEV __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// compile-time loop: the K-block's 36 steps must be straight-line code (the fragment buffers are indexed by the step; a `#pragma unroll`
// gives up silently once the body has grown enough, and then nothing below holds)
template <int V> struct ks_int { static constexpr int value = V; };
template <int I, int N, class F> SRK_DEV void ks_static_for(F&& f) {
  if constexpr (I < N) { f(ks_int<I>{}); ks_static_for<I + 1, N>(f); }
}

// hidden 16-byte load into registers (the residual / mask values of a lane's pixel): the compiler does not see a vector-memory
// instruction, so it neither counts it nor waits for it -- the kernel does, by hand (see the site table below)
SRK_DEV void load16_hidden(i32x4& d, i32x4 rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rsrc) : "memory");
}

// PSIN: pixel-shuffled input (the dgrad of an upsampler conv): the halo pieces' addresses depend on the channel block
template <int DT, bool PSIN>
__global__ __launch_bounds__(512) void conv_ks_kernel(const srk_conv_args a, int tilesX, int tilesY, int ncob, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef DTraits<DT> Tr;
  typedef KsCfg C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const X0 = smem;
  char* const Wr = smem + 2 * C::XS_BYTES;
  float* const bias_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES);     // this block's 64 biases (zeros without a bias)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const int nkb = (a.Cin + 63) >> 6, nch = a.Cin >> 3;  // K-blocks of 64 input channels (the last one may be partial: Cin % 16 == 0);
                                                        // 16-byte chunks per pixel / tap

  // PERSISTENT: the workgroup keeps its output-channel block and walks tiles q, q + stride, ... (the workgroups that share a halo
  // tile -- the ncob blocks of one tile -- have neighbouring LOGICAL indices: one XCD, see below).  The K-blocks of all its tiles form ONE
  // stream Gk = 0 .. total - 1 (tile Gk / nkb, channel block Gk % nkb): halo tiles alternate between the two buffers by Gk's
  // parity, weight slabs keep cycling through the ring -- the next tile's first halo tile and slabs arrive during the current
  // tile's last K-block.  (One tile per workgroup, as before round 3's last change: 6k cycles of prologue, a first K-block twice as
  // long as the later ones -- the ring not yet ahead -- and 3.5k of epilogue around 9.2k cycles of MFMA for 112 -> 128 channels.)
  // The hardware hands consecutive block ids to the eight XCDs in turn, each with an L2 of its own: taken as they come, the ncob blocks of one tile
  // would sit on ncob DIFFERENT XCDs and every one of them would fetch the tile's halo from HBM again (round 6, PMC in the EDSR-large step:
  // 133 MB per 256 -> 256 launch at batch 16 against 39 MB algorithmic).  Logical index = (XCD, slot in the XCD): the blocks of a tile -- and
  // neighbouring tiles -- share one L2.  (SRK_KS_XCD_REMAP=0: A/B builds.)
#ifndef SRK_KS_XCD_REMAP
#define SRK_KS_XCD_REMAP 1
#endif
  unsigned lb = blockIdx.x;
  if (SRK_KS_XCD_REMAP && (gridDim.x & 7u) == 0 && ((gridDim.x >> 3) % (unsigned)ncob) == 0) lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cob = (int)(lb % (unsigned)ncob);
  const int q0 = (int)(lb / (unsigned)ncob), qstride = (int)gridDim.x / ncob;
  const int mytiles = (ntiles - q0 + qstride - 1) / qstride;
  const int total = mytiles * nkb;
  auto tile_origin = [&](int ti, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    int pt = q0 + ti * qstride;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    n = pt / tilesY;
    y0 = tY * C::T;
    x0 = tX * C::T;
  };

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned x_lds = lds_addr_of(X0), wr_lds = lds_addr_of(Wr);

  // halo tile of stream K-block Gk -> X[Gk & 1]: pieces in the tile's LDS order (pixel-major, chunk slot XOR-swizzled by column)
  const int rin = PSIN ? a.x_ps : 1, Cs = a.Cin / (rin * rin);
  // Per lane and piece of a halo tile: xo[k] = byte offset of the piece's 16 bytes for channel block 0, | its chunk index c in the
  // low bits (0x80000000 | c outside the image): the part of the address arithmetic that changes only with the TILE, kept in
  // registers from one tile switch to the next (re-derived at every site it was 150 vector instructions per K-block and wave;
  // the one-tile-per-workgroup kernel had the compiler hoist all of it).  Plain NHWC input only (rin == 1).
  constexpr bool USE_XO = !PSIN;
  const int dtid = tid - C::DT_, dwave = wave - 4;        // the transfers are the DMA waves' (4 - 7) alone: an LDS-DMA piece costs the
                                                          // issuing wave ~60 cycles -- cycles a compute wave's MFMA stream does not have
  int xo[USE_XO ? C::XK : 1];
  auto xo_set = [&](int n, int y0, int x0) __attribute__((always_inline)) {
    if constexpr (USE_XO) {
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int i = dtid + k * C::DT_;
      const int sl = i & 7, p = i >> 3;
      const int iy = p / C::XT, ix = p - iy * C::XT;
      const int c = sl ^ swz(ix);
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      xo[k] = (int)((ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + c * Tr::CH) * 2) : 0x80000000u) | (unsigned)c);
    }
    }
  };
  auto dma_xtile = [&](int par, int kb, int n, int y0, int x0) __attribute__((always_inline)) {      // par: the stream K-block's parity
    const unsigned dst = x_lds + (unsigned)(par * C::XS_BYTES);
    if constexpr (USE_XO) {
#pragma unroll
      for (int k = 0; k < C::XK; ++k) {
        if (k * C::DT_ + dwave * 64 < C::XPIECES) {       // wave-uniform
          const int c = xo[k] & 7;
          const unsigned voff = kb * 8 + c < nch ? ((unsigned)xo[k] & ~15u) + (unsigned)(kb * 128) : 0x80000000u;     // chunks beyond Cin: zeros
          if (dtid + k * C::DT_ < C::XPIECES)              // the last piece is half a wave: its upper lanes are switched off
            dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + ((k * C::DT_ + dwave * 64) << 4))));
        }
      }
    } else {
    const int xij = (kb * 64) / Cs, xc0 = kb * 64 - xij * Cs, xsi = xij / rin, xsj = xij - xsi * rin;
    int tid_ = dtid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int i = tid_ + k * C::DT_;
      if (k * C::DT_ + dwave * 64 < C::XPIECES) {         // wave-uniform
        const int sl = i & 7, p = i >> 3;
        const int iy = p / C::XT, ix = p - iy * C::XT;
        const int c = sl ^ swz(ix);
        const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && kb * 8 + c < nch;      // chunks beyond Cin: zeros
        // (pixel-shuffled input, the dgrad of a conv -> PixelShuffle(r): logical channel k = (i*r+j)*Cs + c lives in pixel
        //  (gy*r+i, gx*r+j), channel c; a 64-channel K-block lies inside one (i, j) plane)
        const int pix = !PSIN ? (n * H + gy) * W + gx : (n * H * rin + gy * rin + xsi) * (W * rin) + gx * rin + xsj;
        const unsigned voff = ok ? (unsigned)((pix * a.x_pitch + a.x_coff + xc0 + c * Tr::CH) * 2) : 0x80000000u;
        if (i < C::XPIECES)                                // the last piece is half tile: its upper lanes are switched off
          dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + ((k * C::DT_ + dwave * 64) << 4))));
      }
    }
    }
  };
  // slab g = (stream K-block g / 3, kernel row g % 3): 3 taps x 8 chunks = 24 pieces of 64 rows x 16 B (contiguous in the packed
  // layout wpk[tap][chunk][CoutP][CH]), 6 per DMA wave, into ring slot g % 3 as [tap][chunk][row]
  auto dma_slab = [&](int kb, int kh) __attribute__((always_inline)) {
    const unsigned dst = wr_lds + (unsigned)(kh * C::WG_BYTES);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int piece = dwave * 6 + k;                     // tap kh*3 + piece / 8, chunk kb*8 + piece % 8
      const int tap = kh * 3 + (piece >> 3), cc = kb * 8 + (piece & 7);
      dma16_hidden(wrsrc, cc < nch ? (unsigned)((((tap * nch + cc) * a.CoutP + cob * 64) << 4) + lane * 16) : 0x80000000u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));      // chunks beyond Cin: zeros
    }
  };

  // bias: the accumulators' initial value at every tile start, from LDS (a global load per tile would enter the hand-counted
  // vector-memory sequence; 32 registers per lane would not fit)
  if (tid < 64) bias_lds[tid] = a.bias ? a.bias[cob * 64 + tid] : 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  f32x16 acc[2][2];                                       // [channel block][pixel block]
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + 4 * h + cb * 32 + 8 * i);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          acc[cb][pb][4 * i + 0] = b.x; acc[cb][pb][4 * i + 1] = b.y; acc[cb][pb][4 * i + 2] = b.z; acc[cb][pb][4 * i + 3] = b.w;
        }
      }
  };
  acc_init();

  // the stream's position, kept incrementally (no division inside the stream): K-block Gk = (tile ti, channel block kb), the
  // tile's origin and the next tile's
  int kb = 0, ti = 0, n0, y0c, x0c, n1 = 0, y1c = 0, x1c = 0;
  tile_origin(0, n0, y0c, x0c);
  if (mytiles > 1) tile_origin(1, n1, y1c, x1c);
  KS_STAMP(0);
  const bool cw = wave < 4;
  if (!cw) {
    xo_set(n0, y0c, x0c);
    dma_xtile(0, 0, n0, y0c, x0c);
    dma_slab(0, 0);
    dma_slab(0, 1);
    dma_slab(0, 2);
  }

  // ---- per-lane constants (as conv_pair's first conv: output pixel (row, col) reads halo pixels (row + kh, col + kw)) -----
  const int px = r & 15;
  int gsw[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) gsw[kw] = swz(px + kw);
  const char* const wlane = Wr + ((h * 64 + r) << 4);
  int prow[2];
  int xoff[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    prow[pb] = 4 * (wave & 3) + 2 * pb + (r >> 4);
    xoff[pb] = (prow[pb] * C::XT + px) << 7;
  }
  // LDS has no room for a residual / mask tile during the K-blocks any more (the buffer it used to wait in now receives the next
  // tile's first halo tile).  The DMA waves fetch both into REGISTERS -- their 8 copy-out pieces' worth each, coalesced (a thread
  // holds 16 B of a whole pixel row: 8 rows per instruction), hidden loads behind site B of a tile's last K-block:
  //   residual: needed where the fp32 accumulators are, so at the tile's end the DMA waves write it into the halo buffer that has
  //             just become free, as the image the epilogue reads (one more barrier);
  //   mask    : applied to the finished 16-bit values, which is exact, during the copy-out.
  // (First form: the compute waves loaded their own lanes' residual values -- 8 instructions that each touch 64 places.  Stamps:
  // +2.7k cycles on the K-block that carried them; a vector-memory instruction holds its wave until the address path has taken
  // all its lanes' requests.)
  const bool has_res = a.res != nullptr, has_mask = a.mask != nullptr;
  const int NL = cw ? 0 : (has_res ? 8 : 0) + (has_mask ? 8 : 0);
  const i32x4 rrsrc = make_rsrc4(has_res ? a.res : a.x, 0x7fffffffu), mrsrc = make_rsrc4(has_mask ? a.mask : a.x, 0x7fffffffu);
  i32x4 rq[8], mq[8];                                      // (not initialised: a value from the kernel's start would occupy the registers all along)
  auto load_tiles = [&](int n, int y0, int x0) __attribute__((always_inline)) {
    int tid_ = dtid;
    asm volatile("" : "+v"(tid_));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid_ + 256 * k;                        // the copy-out's pieces: pixel i >> 3 (row-major 16 x 16), chunk i & 7
      const int p = i >> 3, c = i & 7;
      const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
      const bool ok = gy < H && gx < W && cob * 64 + c * Tr::CH < a.Cout;       // Cout may end inside the last 64-row block
      const int pix = (n * H + gy) * W + gx;
      if (has_res) load16_hidden(rq[k], rrsrc, ok ? (unsigned)((pix * a.res_pitch + a.res_coff + cob * 64 + c * Tr::CH) * 2) : 0x80000000u);
      if (has_mask) load16_hidden(mq[k], mrsrc, ok ? (unsigned)((pix * a.mask_pitch + a.mask_coff + cob * 64 + c * Tr::CH) * 2) : 0x80000000u);
    }
  };

  // Hand-overs inside stream K-block Gk (slab g = 3 Gk + kh), at K-steps 10 / 22 / 34 of its 36: fragments are read two steps
  // ahead, so the reads of the slab that ends at step 12 / 24 / 36 have all been issued and, after the drain, returned --
  // its ring slot is free -- and the next slab must have landed.  The DMA waves own the transfers: each waits for its own
  // pieces by a counted wait (the vector-memory counter retires in order; the counts are what each site leaves in flight), the
  // barrier makes that collective; a compute wave only drains its LDS reads and joins the barrier.  Per DMA wave, in issue order:
  //   site A (step 10): needs slab 3Gk+1; younger: slab 3Gk+2 (6) and, in a tile's FIRST K-block, the previous tile's 8 stores.
  //                     Then issues slab 3Gk+3 (6) and the next halo tile (10, wave 4: 11).
  //   site B (step 22): needs slab 3Gk+2; younger: slab 3Gk+3 + halo tile (>= 16).  Then issues slab 3Gk+4 and, in a tile's LAST
  //                     K-block, its NL = 8 residual and / or 8 mask loads.  (In a tile's first K-block the stores are older than
  //                     these transfers and by now long done: not counted = a stricter wait.)
  //   site C (step 34): needs slab 3Gk+3 and the halo tile; younger: slab 3Gk+4 (6) + NL.  Then issues slab 3Gk+5.
  //   tile end        : needs the residual / mask values: younger: slab 3Gk+5 (6).  Then 8 stores.
  //   Nothing is issued for K-blocks beyond the stream's end (the last K-block's waits: everything).
  // The compute waves issue no vector-memory instruction at all.
  auto wait_barrier = [&](int n) __attribute__((always_inline)) {                          // s_waitcnt vmcnt(n) lgkmcnt(0); s_barrier   (n: a few wave-uniform values)
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 14: asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 22: asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;      // 16
    }
  };
  auto siteA = [&](int Gk, auto computec) __attribute__((always_inline)) {
    if constexpr (decltype(computec)::value != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      const bool more = Gk + 1 < total, lastkb = kb == nkb - 1;
      wait_barrier(Gk > 0 && kb == 0 ? 14 : 6);
      if (more) {
        dma_slab(lastkb ? 0 : kb + 1, 0);
        if (lastkb) xo_set(n1, y1c, x1c);                  // from here on the halo tiles are the next tile's
        dma_xtile((Gk + 1) & 1, lastkb ? 0 : kb + 1, lastkb ? n1 : n0, lastkb ? y1c : y0c, lastkb ? x1c : x0c);
      }
    }
  };
  auto siteB = [&](int Gk, auto computec) __attribute__((always_inline)) {
    const bool lastkb = kb == nkb - 1;
    if constexpr (decltype(computec)::value != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      const bool more = Gk + 1 < total;
      wait_barrier(more ? 16 : 0);
      if (more) dma_slab(lastkb ? 0 : kb + 1, 1);
      if (NL && lastkb) load_tiles(n0, y0c, x0c);
    }
  };
  auto siteC = [&](int Gk, auto computec) __attribute__((always_inline)) {
    if constexpr (decltype(computec)::value != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      const bool more = Gk + 1 < total, lastkb = kb == nkb - 1;
      wait_barrier(more ? 6 + (lastkb ? NL : 0) : 0);
      if (more) dma_slab(lastkb ? 0 : kb + 1, 2);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {     // to the next stream K-block (after tile_end where a tile ends)
    if (++kb == nkb) {
      kb = 0;
      ++ti;
      n0 = n1; y0c = y1c; x0c = x1c;
      if (ti + 1 < mytiles) tile_origin(ti + 1, n1, y1c, x1c);
    }
  };

  // ---- tile end: relu, * scale, + res, mask (channels >= mask_from) -> the halo buffer the tile's last K-block has just left, as
  // a 16 x 16-pixel image; all eight waves copy it out in whole 128-byte pixels (no per-lane 16-byte global access) ------------------
  const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out);
  const int orr = (a.out_mode == SRK_OUT_NHWC_PS && a.ps_r > 1) ? a.ps_r : 1, Cc = a.Cout / (orr * orr);
  const int oij = (cob * 64) / Cc, ocb = cob * 64 - oij * Cc, osi = oij / orr, osj = oij - osi * orr;
  auto tile_end = [&](int Gk, auto computec) __attribute__((always_inline)) {
    char* const stage = X0 + (Gk & 1) * C::XS_BYTES;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every compute wave is done reading that buffer
    if (has_res) {                                                             // the residual tile: DMA waves' registers -> the image the epilogue reads
      if constexpr (decltype(computec)::value == 0) {
        if (Gk + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int tid_ = dtid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          asm volatile("" : "+v"(rq[k]));
          const int i = tid_ + 256 * k, p = i >> 3, c = i & 7;
          lds_write16(stage + (p << 7) + ((c ^ swz(p & 15)) << 4), rq[k]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr (decltype(computec)::value != 0) {
      const float sc = a.scale;
      const f32x2 sc2 = {sc, sc};
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int po = ((prow[pb] * C::T + px) << 7);
        const int g = swz(px);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {                   // 16 of the lane's 32 channels at a time (registers)
          f32x2 v[8];
#pragma unroll
          for (int d = 0; d < 8; ++d) v[d] = f32x2{acc[cb][pb][2 * d], acc[cb][pb][2 * d + 1]};
          if (a.relu) {
#pragma unroll
            for (int d = 0; d < 8; ++d) v[d] = f32x2{relu_f32(v[d].x), relu_f32(v[d].y)};
          }
#pragma unroll
          for (int d = 0; d < 8; ++d) v[d] = v[d] * sc2;
          if (has_res) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const i32x4 tv = lds_read16(stage + po + (((4 * h + 2 * cb + jj) ^ g) << 4));
              const int qw[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float f0, f1;
                unpack2<DT>((uint32_t)qw[e], f0, f1);
                v[4 * jj + e] = v[4 * jj + e] + f32x2{f0, f1};
              }
            }
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            lds_write16(stage + po + (((4 * h + 2 * cb + jj) ^ g) << 4),
                        i32x4{(int)pack2<DT>(v[4 * jj].x, v[4 * jj].y), (int)pack2<DT>(v[4 * jj + 1].x, v[4 * jj + 1].y),
                              (int)pack2<DT>(v[4 * jj + 2].x, v[4 * jj + 2].y), (int)pack2<DT>(v[4 * jj + 3].x, v[4 * jj + 3].y)});
        }
      }
      acc_init();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // the tile's results are in LDS: the compute waves go on
    if constexpr (decltype(computec)::value == 0) {
      const int n = n0, y0 = y0c, x0 = x0c;
      if (has_mask) {                                                          // the mask registers are in
        if (Gk + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(mq[k]));
      }
      int tid_ = tid - 256;
      asm volatile("" : "+v"(tid_));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = tid_ + 256 * k;                      // 2,048 pieces: pixel i >> 3 (row-major 16 x 16), chunk i & 7
        const int p = i >> 3, c = i & 7;
        const int row = p >> 4, col = p & 15;
        const int gy = y0 + row, gx = x0 + col;
        const bool ok = gy < H && gx < W && cob * 64 + c * Tr::CH < a.Cout;
        i32x4 q = lds_read16(stage + (p << 7) + ((c ^ swz(col)) << 4));
        if (has_mask && cob * 64 + c * Tr::CH >= a.mask_from) {            // ReLU mask: keep where the mask value is > 0 (mask_from: a multiple of 16)
          int qw[4] = {q.x, q.y, q.z, q.w};
          const int mw[4] = {mq[k].x, mq[k].y, mq[k].z, mq[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float m0, m1;
            unpack2<DT>((uint32_t)mw[e], m0, m1);
            qw[e] = (m0 > 0.f ? qw[e] & 0xffff : 0) | (m1 > 0.f ? qw[e] & (int)0xffff0000 : 0);
          }
          q = i32x4{qw[0], qw[1], qw[2], qw[3]};
        }
        // (fused PixelShuffle(r) store: packed channel co' = (i*r+j)*Cc + c goes to pixel (gy*r+i, gx*r+j), channel c)
        const int opix = orr == 1 ? (n * H + gy) * W + gx : (n * H * orr + gy * orr + osi) * (W * orr) + gx * orr + osj;
        const unsigned vo = ok ? (unsigned)((opix * a.out_pitch + a.out_coff + ocb + c * Tr::CH) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, vo, 0, SRK_AUX_WT);
      }
    }
  };

  if (!cw) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // halo tile 0 and slab 0 (slabs 1, 2 are a DMA wave's 12 youngest transfers)
  __builtin_amdgcn_s_barrier();
  KS_STAMP(1);

  if (cw) {
    // fragments of K-step s of stream K-block Gk (s >= 36: step s - 36 of the next one, other halo buffer)
    i32x4 fa[3][2], fb[3][2];
    auto frag1 = [&](int Gk, auto sc, auto qc, i32x4 (&af)[2], i32x4 (&bf)[2]) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, q = decltype(qc)::value;
      constexpr int ss = s >= 36 ? s - 36 : s;
      constexpr int kh = ss / 12, kw = (ss % 12) / 4, ks = ss % 4;
      const int kbb = s >= 36 ? Gk + 1 : Gk;
      if constexpr (q < 2) af[q] = lds_read16(wlane + kh * C::WG_BYTES + (((kw * 8 + 2 * ks) * 64 + q * 32) << 4));     // ring slot = kernel row
      else bf[q - 2] = lds_read16(X0 + (kbb & 1) * C::XS_BYTES + xoff[q - 2] + ((kh * C::XT + kw) << 7) + (((2 * ks + h) ^ gsw[kw]) << 4));
    };
    ks_static_for<0, 4>([&](auto qc) __attribute__((always_inline)) { frag1(0, ks_int<0>{}, qc, fa[0], fb[0]); });
    ks_static_for<0, 4>([&](auto qc) __attribute__((always_inline)) { frag1(0, ks_int<1>{}, qc, fa[1], fb[1]); });
    __builtin_amdgcn_s_setprio(1);
    for (int Gk = 0; Gk < total; ++Gk) {
      const bool more = Gk + 1 < total;
      ks_static_for<0, 36>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s == 10) siteA(Gk, ks_int<1>{});
        if constexpr (s == 22) siteB(Gk, ks_int<1>{});
        if constexpr (s == 34) siteC(Gk, ks_int<1>{});
        constexpr int c0 = s % 3, c2 = (s + 2) % 3;
        ks_static_for<0, 4>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          if (s + 2 < 36 || more) frag1(Gk, ks_int<s + 2>{}, mc, fa[c2], fb[c2]);
          constexpr int cb = m >> 1, pb = m & 1;
          acc[cb][pb] = Tr::mma(fa[c0][cb], fb[c0][pb], acc[cb][pb]);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      if (Gk < 8) KS_STAMP(2 + Gk);
      if (kb == nkb - 1) {
        __builtin_amdgcn_s_setprio(0);
        tile_end(Gk, ks_int<1>{});
        __builtin_amdgcn_s_setprio(1);
      }
      advance();
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    for (int Gk = 0; Gk < total; ++Gk) {
      siteA(Gk, ks_int<0>{});
      siteB(Gk, ks_int<0>{});
      siteC(Gk, ks_int<0>{});
      if (kb == nkb - 1) tile_end(Gk, ks_int<0>{});
      advance();
    }
  }
  KS_STAMP(13);
}

}  // namespace

#if SRK_KS_STAMPS
extern "C" int srk_ks_read_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(ks_stamp_buf), sizeof(unsigned long long) * 64);
}
#endif


// Whether srk_conv2d takes this kernel for `a` (16-bit, 3x3, more than one 64-channel input block -- Cin >= 96, a multiple of 16 --,
// the stored channels ending inside the last 64-row block of the packed weights,
// NHWC in and out, optionally a pixel-shuffled input -- dgrad of an upsampler conv -- or the fused PixelShuffle store).
// SRK_NO_KS=1 keeps the streaming kernel (A/B runs).
bool srk_conv_ks_ok(const srk_conv_args& a) {
  static const bool off = [] { const char* e = srk_dbg_getenv("SRK_NO_KS"); return e && e[0] == '1'; }();
  if (off || a.dtype == SRK_F32 || a.KH != 3 || a.KW != 3) return false;
  if (a.out_mode == SRK_OUT_PLANAR || a.post_add) return false;
  if (a.Cin < 96 || a.Cin % 16 != 0 || a.CoutP % 64 != 0 || a.Cout % 8 != 0 || a.Cout > a.CoutP || a.Cout <= a.CoutP - 64) return false;
  const int rin = a.x_ps > 1 ? a.x_ps : 1, rr = (a.out_mode == SRK_OUT_NHWC_PS && a.ps_r > 1) ? a.ps_r : 1;
  if (rin > 1 && (a.Cin % 64 != 0 || a.Cin % (rin * rin) != 0 || (a.Cin / (rin * rin)) % 64 != 0)) return false;   // a K-block inside one (i, j) plane
  if (rr > 1 && (a.Cout != a.CoutP || a.Cout % (rr * rr) != 0 || (a.Cout / (rr * rr)) % 64 != 0 || a.res || a.mask)) return false;
  if (a.x_pitch % 8 || a.x_coff % 8 || a.out_pitch % 8 || a.out_coff % 8) return false;
  if (a.res && (a.res_pitch % 8 || a.res_coff % 8)) return false;
  if (a.mask && (a.mask_pitch % 8 || a.mask_coff % 8 || a.mask_from % 16)) return false;
  const long long px = (long long)a.N * a.H * a.W;
  long long mx = px * rin * rin * a.x_pitch;
  if (px * rr * rr * a.out_pitch > mx) mx = px * rr * rr * a.out_pitch;
  if (a.res && px * a.res_pitch > mx) mx = px * a.res_pitch;
  if (a.mask && px * a.mask_pitch > mx) mx = px * a.mask_pitch;
  if (mx * 2 >= 0x7fff0000LL) return false;
  const long long wb = 9LL * (a.Cin / 8) * a.CoutP * 16;
  return wb < 0x7fff0000LL;
}

int srk_conv_ks_launch(const srk_conv_args& a, hipStream_t st) {
  typedef KsCfg C;
  constexpr int LDS = C::LDS_BYTES + 256;                  // + the block's biases
  typedef void (*ks_fn)(const srk_conv_args, int, int, int, unsigned, unsigned, int);
  static const ks_fn fns[2][2] = {{conv_ks_kernel<SRK_BF16, false>, conv_ks_kernel<SRK_BF16, true>}, {conv_ks_kernel<SRK_F16, false>, conv_ks_kernel<SRK_F16, true>}};
  static const hipError_t attr = [] {
    for (int d = 0; d < 2; ++d)
      for (int v = 0; v < 2; ++v) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[d][v]), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
      }
    return hipSuccess;
  }();
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d: cannot reserve %d bytes of LDS", LDS);
    return (int)attr;
  }
  const int tilesX = (a.W + C::T - 1) / C::T, tilesY = (a.H + C::T - 1) / C::T, ncob = a.CoutP / 64;
  const long long ntiles = (long long)a.N * tilesX * tilesY;
  SRK_CHECK_ARG(ntiles * ncob <= 0x7fffffffLL, "srk_conv2d: %lld tiles", ntiles * ncob);
  // one workgroup per CU (156 KB of LDS each), every one with the same number of output-channel blocks' worth of neighbours
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  long long nb = ntiles * ncob;
  if (nb > cus) nb = cus >= ncob ? (long long)(cus / ncob) * ncob : ncob;
  const int rin = a.x_ps > 1 ? a.x_ps : 1;
  const unsigned xb = (unsigned)((long long)a.N * a.H * a.W * rin * rin * a.x_pitch * 2);
  const unsigned wb = (unsigned)(9LL * (a.Cin / 8) * a.CoutP * 16);
  hipLaunchKernelGGL(fns[a.dtype == SRK_BF16 ? 0 : 1][rin > 1 ? 1 : 0], dim3((unsigned)nb), dim3(C::NT), LDS, st, a, tilesX, tilesY, ncob, xb, wb, (int)ntiles);
  SRK_LAUNCH_CHECK();
  return 0;
}
