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
  static constexpr int XK = (XPIECES + NT - 1) / NT;    // 6 pieces per lane
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

SRK_DEV __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int DT>
__global__ __launch_bounds__(512) void conv_ks_kernel(const srk_conv_args a, int tilesX, int tilesY, int ncob, unsigned x_bytes, unsigned w_bytes) {
  typedef DTraits<DT> Tr;
  typedef KsCfg C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const X0 = smem;
  char* const Wr = smem + 2 * C::XS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const int nkb = (a.Cin + 63) >> 6, nch = a.Cin >> 3;  // K-blocks of 64 input channels (the last one may be partial: Cin % 16 == 0);
                                                        // 16-byte chunks per pixel / tap

  // output-channel block fastest: the workgroups that share a halo tile run together
  int pt = blockIdx.x;
  const int cob = pt % ncob;
  pt /= ncob;
  const int tX = pt % tilesX;
  pt /= tilesX;
  const int tY = pt % tilesY;
  const int n = pt / tilesY;
  const int y0 = tY * C::T, x0 = tX * C::T;

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned x_lds = lds_addr_of(X0), wr_lds = lds_addr_of(Wr);

  // halo tile of K-block kb -> X[kb & 1]: pieces in the tile's LDS order (pixel-major, chunk slot XOR-swizzled by column)
  const int rin = a.x_ps > 1 ? a.x_ps : 1, Cs = a.Cin / (rin * rin);
  auto dma_xtile = [&](int kb) {
    const unsigned dst = x_lds + (unsigned)((kb & 1) * C::XS_BYTES);
    const int xij = (kb * 64) / Cs, xc0 = kb * 64 - xij * Cs, xsi = xij / rin, xsj = xij - xsi * rin;
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int i = tid + k * C::NT;
      if (k * C::NT + wave * 64 < C::XPIECES) {           // wave-uniform
        const int sl = i & 7, p = i >> 3;
        const int iy = p / C::XT, ix = p - iy * C::XT;
        const int c = sl ^ swz(ix);
        const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && kb * 8 + c < nch;      // chunks beyond Cin: zeros
        // (pixel-shuffled input, the dgrad of a conv -> PixelShuffle(r): logical channel k = (i*r+j)*Cs + c lives in pixel
        //  (gy*r+i, gx*r+j), channel c; a 64-channel K-block lies inside one (i, j) plane)
        const int pix = rin == 1 ? (n * H + gy) * W + gx : (n * H * rin + gy * rin + xsi) * (W * rin) + gx * rin + xsj;
        const unsigned voff = ok ? (unsigned)((pix * a.x_pitch + a.x_coff + xc0 + c * Tr::CH) * 2) : 0x80000000u;
        if (i < C::XPIECES)                                // the last piece is half tile: its upper lanes are switched off
          dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + ((k * C::NT + wave * 64) << 4))));
      }
    }
  };
  // slab g = (K-block g / 3, kernel row g % 3): 3 taps x 8 chunks = 24 pieces of 64 rows x 16 B (contiguous in the packed
  // layout wpk[tap][chunk][CoutP][CH]), 3 per wave, into ring slot g % 3 as [tap][chunk][row]
  auto dma_slab = [&](int g) {
    const int kb = g / 3, kh = g - 3 * kb;
    const unsigned dst = wr_lds + (unsigned)((g % 3) * C::WG_BYTES);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int piece = wave * 3 + k;                      // tap kh*3 + piece / 8, chunk kb*8 + piece % 8
      const int tap = kh * 3 + (piece >> 3), cc = kb * 8 + (piece & 7);
      dma16_hidden(wrsrc, cc < nch ? (unsigned)((((tap * nch + cc) * a.CoutP + cob * 64) << 4) + lane * 16) : 0x80000000u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));      // chunks beyond Cin: zeros
    }
  };
  // residual / mask tile (16x16 pixels x this block's 64 channels) -> a halo buffer, image format with row pitch 16
  auto dma_tile16 = [&](const void* src, int pitch, int coff, char* buf) {
    const i32x4 rs = make_rsrc4(src, 0x7fffffffu);
    const unsigned dst = lds_addr_of(buf);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int blk = wave + 8 * k;                        // 32 pieces of 8 pixels: row blk >> 1, columns (blk & 1) * 8 ..
      const int iy = blk >> 1, ix = (blk & 1) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz(ix);
      const int gy = y0 + iy, gx = x0 + ix;
      const bool ok = gy < H && gx < W && cob * 64 + c * Tr::CH < a.Cout;      // Cout may end inside the last 64-row block
      const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * pitch + coff + cob * 64 + c * Tr::CH) * 2) : 0x80000000u;
      dma16_hidden(rs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (blk << 10))));
    }
  };

  // bias = initial accumulators; loaded BEFORE the first DMA (the vector-memory counter retires in order: a wait for these
  // loads placed behind 115 KB of transfers would wait for all of them)
  f32x16 acc[2][2];                                       // [channel block][pixel block]
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cob * 64 + 4 * h + cb * 32 + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc[cb][pb][4 * i + 0] = b.x; acc[cb][pb][4 * i + 1] = b.y; acc[cb][pb][4 * i + 2] = b.z; acc[cb][pb][4 * i + 3] = b.w;
      }
    }

  asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));

  dma_xtile(0);
  dma_slab(0);
  dma_slab(1);
  dma_slab(2);

  // ---- per-lane constants (as conv_pair's first conv: output pixel (row, col) reads halo pixels (row + kh, col + kw)) -----
  const bool cw = wave < 4;
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
  // Hand-overs inside K-block kb (slab g = 3 kb + kh), at K-steps 10 / 22 / 34 of its 36: fragments are read two steps
  // ahead, so the reads of the slab that ends at step 12 / 24 / 36 have all been issued and, after the drain, returned --
  // its ring slot is free -- and the next slab must have landed (own pieces by the counted wait: the vector-memory counter
  // retires in order and the counts are what each site leaves in flight; the others' by the barrier).
  //   site A (step 10): needs slab 3kb+1; in flight: slab 3kb+2 (3).  Then issues slab 3kb+3 and the NEXT halo tile (<= 6) --
  //                     in the last K-block the residual / mask tile (4) instead, into the idle halo buffer.
  //   site B (step 22): needs slab 3kb+2; in flight: slab 3kb+3 + halo tile (>= 8 with 5 halo pieces; last K-block: waits for
  //                     everything).  Then issues slab 3kb+4.
  //   site C (step 34): needs slab 3kb+3 and the tile; in flight: slab 3kb+4 (3).  Then issues slab 3kb+5.
  const int nslab = 3 * nkb;
  const void* const tile_src = a.res ? a.res : a.mask;     // the tile fetched during the last K-block (the mask comes later if both)
  char* const stage = X0 + (nkb & 1) * C::XS_BYTES;        // idle during the last K-block (which reads X[(nkb-1) & 1])
  auto siteA = [&](int kb) {
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (3 * kb + 3 < nslab) dma_slab(3 * kb + 3);
    if (kb + 1 < nkb) dma_xtile(kb + 1);
    else if (tile_src) dma_tile16(tile_src, a.res ? a.res_pitch : a.mask_pitch, a.res ? a.res_coff : a.mask_coff, stage);
  };
  auto siteB = [&](int kb) {
    // (last K-block: nothing but the residual / mask tile is younger than the slab that is needed)
    if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (3 * kb + 4 < nslab) dma_slab(3 * kb + 4);
  };
  auto siteC = [&](int kb) {
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (3 * kb + 5 < nslab) dma_slab(3 * kb + 5);
  };

  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         // halo tile 0 and slab 0 (slabs 1, 2 are the 6 youngest transfers)
  __builtin_amdgcn_s_barrier();

  if (cw) {
    // fragments of K-step s of the current K-block (s >= 36: step s - 36 of the next one, other halo buffer)
    i32x4 fa[3][2], fb[3][2];
    auto frag1 = [&](int kb, int s, int q, i32x4 (&af)[2], i32x4 (&bf)[2]) {
      const int kbb = s >= 36 ? kb + 1 : kb, ss = s >= 36 ? s - 36 : s;
      const int kh = ss / 12, kw = (ss % 12) / 4, ks = ss % 4;
      if (q < 2) af[q] = lds_read16(wlane + kh * C::WG_BYTES + (((kw * 8 + 2 * ks) * 64 + q * 32) << 4));     // ring slot = kernel row
      else bf[q - 2] = lds_read16(X0 + (kbb & 1) * C::XS_BYTES + xoff[q - 2] + ((kh * C::XT + kw) << 7) + (((2 * ks + h) ^ gsw[kw]) << 4));
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) frag1(0, 0, q, fa[0], fb[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) frag1(0, 1, q, fa[1], fb[1]);
    __builtin_amdgcn_s_setprio(1);
    for (int kb = 0; kb < nkb; ++kb) {
      const bool more = kb + 1 < nkb;
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        if (s == 10) siteA(kb);
        if (s == 22) siteB(kb);
        if (s == 34) siteC(kb);
        const int c0 = s % 3, c2 = (s + 2) % 3;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (s + 2 < 36 || more) frag1(kb, s + 2, m, fa[c2], fb[c2]);
          const int cb = m >> 1, pb = m & 1;
          acc[cb][pb] = Tr::mma(fa[c0][cb], fb[c0][pb], acc[cb][pb]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    for (int kb = 0; kb < nkb; ++kb) {
      siteA(kb);
      siteB(kb);
      siteC(kb);
    }
  }
  // everything has landed (the residual / mask tile included); the last K-block's halo buffer is free
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  char* const other = X0 + ((nkb - 1) & 1) * C::XS_BYTES;
  const bool both = a.res && a.mask;
  if (both) {                                              // residual AND mask (RDN's dense-block dgrad): the mask tile now
    dma_tile16(a.mask, a.mask_pitch, a.mask_coff, other);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- epilogue: relu, * scale, + res, mask (channels >= mask_from), written over the residual tile in `stage` ------------
  if (cw) {
    const float sc = a.scale;
    const f32x2 sc2 = {sc, sc};
    const char* const mbuf = both ? other : stage;
    const bool mask_on = a.mask && cob * 64 + 32 * h + 32 > a.mask_from;      // some of this lane's 32 channels are masked
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      f32x2 v[16];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int d = 0; d < 8; ++d) v[8 * cb + d] = f32x2{acc[cb][pb][2 * d], acc[cb][pb][2 * d + 1]};
      if (a.relu) {
#pragma unroll
        for (int d = 0; d < 16; ++d) v[d] = f32x2{fmaxf(v[d].x, 0.f), fmaxf(v[d].y, 0.f)};
      }
#pragma unroll
      for (int d = 0; d < 16; ++d) v[d] = v[d] * sc2;
      const int po = ((prow[pb] * C::T + px) << 7);
      const int g = swz(px);
      if (a.res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const i32x4 q = lds_read16(stage + po + (((4 * h + j) ^ g) << 4));
          const int qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f0, f1;
            unpack2<DT>((uint32_t)qw[e], f0, f1);
            v[4 * j + e] = v[4 * j + e] + f32x2{f0, f1};
          }
        }
      }
      if (mask_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const i32x4 q = lds_read16(mbuf + po + (((4 * h + j) ^ g) << 4));
          const int qw[4] = {q.x, q.y, q.z, q.w};
          const bool on = cob * 64 + 32 * h + 8 * j >= a.mask_from;            // mask_from is a multiple of 16
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float m0, m1;
            unpack2<DT>((uint32_t)qw[e], m0, m1);
            if (on) v[4 * j + e] = f32x2{m0 > 0.f ? v[4 * j + e].x : 0.f, m1 > 0.f ? v[4 * j + e].y : 0.f};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        lds_write16(stage + po + (((4 * h + j) ^ g) << 4),
                    i32x4{(int)pack2<DT>(v[4 * j].x, v[4 * j].y), (int)pack2<DT>(v[4 * j + 1].x, v[4 * j + 1].y),
                          (int)pack2<DT>(v[4 * j + 2].x, v[4 * j + 2].y), (int)pack2<DT>(v[4 * j + 3].x, v[4 * j + 3].y)});
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out);
    const int orr = (a.out_mode == SRK_OUT_NHWC_PS && a.ps_r > 1) ? a.ps_r : 1, Cc = a.Cout / (orr * orr);
    const int oij = (cob * 64) / Cc, ocb = cob * 64 - oij * Cc, osi = oij / orr, osj = oij - osi * orr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + C::NT * k;                       // 2,048 pieces: pixel i >> 3 (row-major 16 x 16), chunk i & 7
      const int p = i >> 3, c = i & 7;
      const int row = p >> 4, col = p & 15;
      const int gy = y0 + row, gx = x0 + col;
      const bool ok = gy < H && gx < W && cob * 64 + c * Tr::CH < a.Cout;
      const i32x4 q = lds_read16(stage + (p << 7) + ((c ^ swz(col)) << 4));
      // (fused PixelShuffle(r) store: packed channel co' = (i*r+j)*Cc + c goes to pixel (gy*r+i, gx*r+j), channel c)
      const int opix = orr == 1 ? (n * H + gy) * W + gx : (n * H * orr + gy * orr + osi) * (W * orr) + gx * orr + osj;
      const unsigned vo = ok ? (unsigned)((opix * a.out_pitch + a.out_coff + ocb + c * Tr::CH) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, vo, 0, 0);
    }
  }
}

}  // namespace

// Whether srk_conv2d takes this kernel for `a` (16-bit, 3x3, more than one 64-channel input block -- Cin >= 96, a multiple of 16 --,
// the stored channels ending inside the last 64-row block of the packed weights,
// NHWC in and out, optionally a pixel-shuffled input -- dgrad of an upsampler conv -- or the fused PixelShuffle store).
// SRK_NO_KS=1 keeps the streaming kernel (A/B runs).
bool srk_conv_ks_ok(const srk_conv_args& a) {
  static const bool off = [] { const char* e = getenv("SRK_NO_KS"); return e && e[0] == '1'; }();
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
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ks_kernel<SRK_BF16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ks_kernel<SRK_F16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  if (attr0 != hipSuccess || attr1 != hipSuccess) {
    srk_set_error("srk_conv2d: cannot reserve %d bytes of LDS", C::LDS_BYTES);
    return (int)(attr0 != hipSuccess ? attr0 : attr1);
  }
  const int tilesX = (a.W + C::T - 1) / C::T, tilesY = (a.H + C::T - 1) / C::T, ncob = a.CoutP / 64;
  const long long nb = (long long)a.N * tilesX * tilesY * ncob;
  SRK_CHECK_ARG(nb <= 0x7fffffffLL, "srk_conv2d: %lld workgroups", nb);
  const int rin = a.x_ps > 1 ? a.x_ps : 1;
  const unsigned xb = (unsigned)((long long)a.N * a.H * a.W * rin * rin * a.x_pitch * 2);
  const unsigned wb = (unsigned)(9LL * (a.Cin / 8) * a.CoutP * 16);
  if (a.dtype == SRK_BF16) hipLaunchKernelGGL((conv_ks_kernel<SRK_BF16>), dim3((unsigned)nb), dim3(C::NT), C::LDS_BYTES, st, a, tilesX, tilesY, ncob, xb, wb);
  else hipLaunchKernelGGL((conv_ks_kernel<SRK_F16>), dim3((unsigned)nb), dim3(C::NT), C::LDS_BYTES, st, a, tilesX, tilesY, ncob, xb, wb);
  SRK_LAUNCH_CHECK();
  return 0;
}
