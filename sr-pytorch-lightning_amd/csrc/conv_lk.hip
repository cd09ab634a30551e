// Direct large-kernel convolution (5x5, 7x7, 9x9, stride 1, 'same' padding) for few channels on one side: SRResNet's 9x9 tail
// conv 64 -> 3 at HR resolution (models/srresnet.py:29: DefaultConv2d(n_feats, channels, 9)) forward, data gradient and weight
// gradient.  The generic path (generic.hip) materialises the K*K-times larger column tensor in HBM (6.1 GB per batch of 16 for
// this layer: 8 of SRResNet's 11.9 ms per step); here the (16 + K - 1)^2 halo tile sits in LDS once and all K*K taps read it at
// shifted addresses, exactly as the 3x3 kernels do, with the weights of one KERNEL ROW (K taps) at a time streaming through a
// two-slot LDS ring by hidden LDS-DMA.
//   lk_conv_kernel<DT, CPP, NRB>  forward (CPP = 8: 64 input channels, NRB = 1: <= 32 output rows) and, with data-gradient packs,
//                                 dgrad (CPP = 2: 16 stored gradient channels, NRB = 2: 64 rows).  4 waves x 64 pixels of a 16x16 tile.
//   lk_wgrad_kernel<DT>           dW[tap][ci][co]: a workgroup owns ONE kernel row (K taps x 64 input channels x 16 gradient
//                                 channels = 18 accumulator tiles over its 4 waves) and a range of tiles; K = the 16 pixels of a tile
//                                 row, both operands by the transposing LDS read; partial sums to per-workgroup slabs
//                                 (srk_wgrad_finalize's layout).
// Packed weights, epilogue order and slab format are srk_conv2d's / srk_conv2d_wgrad's: those entry points dispatch here.
#include <stdlib.h>
#include <type_traits>
#include "srk_common.h"

// diagnostics build only (make stamp, tools/stamp_lk5.py): s_memtime stamps of wave 0 of workgroup 0 of lk5_dgrad_kernel through `post_add`
#ifndef SRK_LK5_STAMPS
#define SRK_LK5_STAMPS 0
#endif

namespace {

template <int DT, int CPP, int NRB, int K>
__global__ __launch_bounds__(256) void lk_conv_kernel(const srk_conv_args a, int tilesX, int tilesY, unsigned x_bytes, unsigned w_bytes) {
  typedef DTraits<DT> Tr;
  constexpr int PXB = CPP * 16;                                  // bytes per pixel in the LDS tile
  constexpr int ROWS = 32 * NRB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XT = 16 + K - 1, XTP = (XT + 1) & ~1;           // halo tile edge, even row pitch
  constexpr int xs_bytes = XT * XTP * PXB;
  constexpr int slab = K * CPP * ROWS * 16;                      // one kernel row of packed weights
  char* const Xs = smem;
  char* const Wr = smem + ((xs_bytes + 1023) & ~1023);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, P = K / 2;
  int pt = blockIdx.x;
  const int tX = pt % tilesX;
  pt /= tilesX;
  const int tY = pt % tilesY;
  const int n = pt / tilesY;
  const int y0 = tY * 16, x0 = tX * 16;

  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes), wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned xs_lds = lds_addr_of(Xs), wr_lds = lds_addr_of(Wr);

  // bias = initial accumulators (loaded before any hidden DMA is queued: the vector-memory counter retires in order)
  f32x16 acc[NRB][2];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rb * 32 + 4 * h + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc[rb][pb][4 * i + 0] = b.x; acc[rb][pb][4 * i + 1] = b.y; acc[rb][pb][4 * i + 2] = b.z; acc[rb][pb][4 * i + 3] = b.w;
      }
    }
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) asm volatile("" : "+v"(acc[rb][0]), "+v"(acc[rb][1]));

  // halo tile: 1 KB pieces of 64 / CPP pixels; chunk slot XOR-swizzled by the tile column when a pixel is a whole 128-byte row
  {
    constexpr int npieces = (XT * XTP * CPP + 63) / 64;
#pragma unroll 4
    for (int k = wave; k < npieces; k += 4) {
      const int i = k * 64 + lane;
      const int sl = i % CPP, p = i / CPP;
      const int iy = p / XTP, ix = p - iy * XTP;
      const int c = CPP == 8 ? (sl ^ swz(ix)) : sl;
      const int gy = y0 - P + iy, gx = x0 - P + ix;
      const bool ok = iy < XT && ix < XT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c * 8 < a.Cin;
      const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + c * 8) * 2) : 0x80000000u;
      dma16_hidden(xrsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(xs_lds + (k << 10))));
    }
  }
  constexpr int spieces = slab >> 10;                            // whole KB: K * CPP * ROWS * 16 is a multiple of 1024 for ROWS >= 32, CPP >= 2
  auto dma_slab = [&](int kh) {
    const unsigned dst = wr_lds + (unsigned)((kh & 1) * slab);
    for (int k = wave; k < spieces; k += 4)
      dma16_hidden(wrsrc, (unsigned)(kh * slab + (k << 10) + lane * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (k << 10))));
  };
  dma_slab(0);

  const int px = r & 15;
  int prow[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) prow[pb] = 4 * wave + 2 * pb + (r >> 4);
  const char* const wl = Wr + ((h * ROWS + r) << 4);

  for (int kh = 0; kh < K; ++kh) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // slab kh (and, first, the tile) landed; slot (kh + 1) & 1 is free
    if (kh + 1 < K) dma_slab(kh + 1);
    const char* const ws = wl + (kh & 1) * slab;
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int g = CPP == 8 ? swz(px + kw) : 0;
#pragma unroll
      for (int ks = 0; ks < CPP / 2; ++ks) {
        i32x4 bf[2], af[NRB];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
          bf[pb] = lds_read16(Xs + ((prow[pb] + kh) * XTP + px + kw) * PXB + (((2 * ks + h) ^ g) << 4));
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) af[rb] = lds_read16(ws + (((kw * CPP + 2 * ks) * ROWS + rb * 32) << 4));
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[rb][pb] = Tr::mma(af[rb], bf[pb], acc[rb][pb]);
      }
    }
  }

  // epilogue: v = acc (+ bias already); relu; * scale; + res; store.  Lane (pixel, half h) holds 16 (NRB = 1) / 32 (NRB = 2) contiguous
  // channels of its pixel: channel = 16 h + 4 i + e  resp.  32 h + 16 rb + 4 i + e  (row_to_chan, srk_common.h)
  const float sc = a.scale;
  if constexpr (NRB == 1) {
    if (a.out_mode == SRK_OUT_PLANAR) {
      // fp32 NCHW behind a PixelShuffle(2): channel k = o*4 + i*2 + j of pixel (gy, gx) is out[n][o][2 gy + i][2 gx + j] (+ post_add[o]); the
      // collapsed HR stage's 5x5 conv (hr_tail.hip) writes the image this way.  Only the h = 0 lanes hold stored channels (Cout <= 16);
      // a lane's (j = 0, 1) pair is one 8-byte store, 16 lanes cover 128 contiguous bytes of an output row.
      const int O = a.Cout >> 2, H2 = 2 * H, W2 = 2 * W;
      if (h == 0) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int gy = y0 + prow[pb], gx = x0 + px;
          if (gy >= H || gx >= W) continue;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            if (o >= O) break;
            const float pa = a.post_add ? a.post_add[o] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              f32x2 v;
              v.x = acc[0][pb][o * 4 + i * 2] * sc + pa;
              v.y = acc[0][pb][o * 4 + i * 2 + 1] * sc + pa;
              *reinterpret_cast<f32x2*>(reinterpret_cast<float*>(a.out) + (((size_t)n * O + o) * H2 + 2 * gy + i) * W2 + 2 * gx) = v;
            }
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int gy = y0 + prow[pb], gx = x0 + px;
    if (gy >= H || gx >= W) continue;
    const size_t pix = (size_t)(n * H + gy) * W + gx;
    typename Tr::elem* const o = reinterpret_cast<typename Tr::elem*>(a.out) + pix * a.out_pitch + a.out_coff;
    const typename Tr::elem* const rs = a.res ? reinterpret_cast<const typename Tr::elem*>(a.res) + pix * a.res_pitch + a.res_coff : nullptr;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int q = 0; q < 2; ++q) {                              // 8 channels = one 16-byte store
        const int ch = (NRB == 1 ? 16 * h : 32 * h + 16 * rb) + 8 * q;
        if (ch >= a.Cout) continue;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float u = acc[rb][pb][8 * q + t];
          if (a.relu) u = relu_f32(u);
          v[t] = u * sc;
        }
        if (rs) {
          const i32x4 qv = *reinterpret_cast<const i32x4*>(rs + ch);
          const int qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f0, f1;
            unpack2<DT>((uint32_t)qw[e], f0, f1);
            v[2 * e] += f0; v[2 * e + 1] += f1;
          }
        }
        i32x4 ov;
        ov.x = (int)pack2<DT>(v[0], v[1]); ov.y = (int)pack2<DT>(v[2], v[3]); ov.z = (int)pack2<DT>(v[4], v[5]); ov.w = (int)pack2<DT>(v[6], v[7]);
        *reinterpret_cast<i32x4*>(o + ch) = ov;
      }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------
// blockIdx = (slab, kernel row kh).  Per tile: the 16 x (16 + K - 1) pixels of x the kernel row touches (128 bytes per pixel,
// swizzled image) and the 16 x 16 gradient tile as a 128-byte-per-pixel image whose channels beyond Cout are zero-filled by the
// DMA; one K-step = one 16-pixel tile row; pair p = 2 kw + rb of (tap column, 32-row input-channel block) belongs to wave p & 3.
template <int DT, int K>
__global__ __launch_bounds__(256) void lk_wgrad_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int ntiles, int tq, int trem,
                                                       unsigned x_bytes, unsigned dy_bytes) {
  typedef DTraits<DT> Tr;
  constexpr int MAXP = (2 * K + 3) / 4;                          // pairs per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 + K - 1, XWP = (XW + 1) & ~1;
  constexpr int xbuf = 16 * XWP * 128, dbuf = 16 * 16 * 32 + 1024, buf = xbuf + dbuf;   // gradient tile: 32 bytes per pixel (16 channels) + 1 KB of zeros
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x, kh = blockIdx.y;
  const int H = a.H, W = a.W, P = K / 2;
  const int t0 = slot * tq + min(slot, trem), nt = tq + (slot < trem ? 1 : 0);
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);

  // per-lane piece constants (tile-independent): x piece k = wave + 4 j covers pixels (iy, ix) of the 16 x XWP rows, chunk c;
  // packed iy | ix << 8 | c << 16 | valid << 24
  constexpr int NPX = 16 * XWP * 8 / 64, XPW = (NPX + 3) / 4, NPD = 9, DPW = (NPD + 3) / 4;   // + 8 gradient pieces + 1 of zeros
  int xdesc[XPW], ddesc[DPW];
#pragma unroll
  for (int j = 0; j < XPW; ++j) {
    const int k = wave + 4 * j, i = k * 64 + lane, sl = i & 7, p = i >> 3;
    const int iy = p / XWP, ix = p - iy * XWP, c = sl ^ swz(ix);
    xdesc[j] = iy | (ix << 8) | (c << 16) | ((k < NPX && ix < XW && c * 8 < a.Cin) ? 1 << 24 : 0);
  }
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int k = wave + 4 * j, i = k * 64 + lane, c = i & 1, pp = i >> 1;          // 2 chunks per pixel, pixels row-major 16 x 16
    ddesc[j] = (pp >> 4) | ((pp & 15) << 8) | (c << 16) | ((k < 8 && c * 8 < a.Cout) ? 1 << 24 : 0);
  }
  auto dma_tile = [&](int tile, int b) {
    int pt = tile;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    const int n = pt / tilesY;
    const int y0 = tY * 16, x0 = tX * 16;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
      const int k = wave + 4 * j;
      if (k < NPX) {
        const int d = xdesc[j], gy = y0 + (d & 255) + kh - P, gx = x0 + ((d >> 8) & 255) - P;
        const bool ok = (d >> 24) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
        dma16_hidden(xrs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + (k << 10))));
      }
    }
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int k = wave + 4 * j;
      if (k < NPD) {
        const int d = ddesc[j], gy = y0 + (d & 255), gx = x0 + ((d >> 8) & 255);
        const bool ok = (d >> 24) && gy < H && gx < W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.dy_pitch + a.dy_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
        dma16_hidden(drs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + xbuf + (k << 10))));
      }
    }
  };

  constexpr int npairs = 2 * K;
  f32x16 acc[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  int aoff[MAXP][2], boff[2];
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    const int p = wave + 4 * j;                                  // kw = p >> 1, rb = p & 1
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) aoff[j][rd] = tr_lane_off(p >> 1, rd, p & 1, lane);
  }
  // gradient fragment: lane l of 16-lane group G supplies pixel 8 (G >> 1) + 4 rd + q, channels 4 p .. + 3 of block G & 1; block 1
  // (channels 16 .. 31) does not exist: those lanes read the zero piece behind the tile
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int G = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int col = 8 * (G >> 1) + 4 * rd + q;
    boff[rd] = (G & 1) ? (16 * 16 * 32 + pq * 8) : (col * 32 + pq * 8);
  }

  const bool do_bias = a.dbp != nullptr && kh == 0 && wave == 0;     // db = sum of the gradient: from the fragments wave 0 fetches anyway
  float dbz = 0.f;
  if (nt > 0) dma_tile(t0, 0);
  for (int it = 0; it < nt; ++it) {
    const char* const X = smem + (it & 1) * buf;
    const char* const D = X + xbuf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (it + 1 < nt) dma_tile(t0 + it + 1, (it + 1) & 1);
    // fragments of tile row y + 1 are requested before the MFMAs of row y
    i32x4 bfn, afn[MAXP];
    auto fetch = [&](int y) {
      const int yd = (lane & 16) ? 0 : y * 512;                  // (the zero piece has no rows)
      bfn = tr_read2(D + yd + boff[0], D + yd + boff[1]);
#pragma unroll
      for (int j = 0; j < MAXP; ++j)
        if (wave + 4 * j < npairs) afn[j] = tr_read2(X + y * (XWP * 128) + aoff[j][0], X + y * (XWP * 128) + aoff[j][1]);
    };
    fetch(0);
#pragma unroll 2
    for (int y = 0; y < 16; ++y) {
      const i32x4 bf = bfn;
      i32x4 af[MAXP];
#pragma unroll
      for (int j = 0; j < MAXP; ++j) af[j] = afn[j];
      if (y + 1 < 16) fetch(y + 1);
      if (do_bias) {                                             // lane = channel (lane & 31), 8 pixels of it per read
        const int qw[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f0, f1;
          unpack2<DT>((uint32_t)qw[e], f0, f1);
          dbz += f0 + f1;
        }
      }
#pragma unroll
      for (int j = 0; j < MAXP; ++j)
        if (wave + 4 * j < npairs) acc[j] = Tr::mma(af[j], bf, acc[j]);
    }
  }

  if (do_bias) {
    dbz += __shfl_xor(dbz, 32, 64);                              // the two K halves of a read
    if (lane < a.Cout) a.dbp[(size_t)slot * a.Cout + lane] = dbz;
  }
  // slab [tap][ci][co] fp32 (srk_wgrad_finalize's layout, Cout = the gradient's stored channels): rows = input channels
  {
    const int hq = lane >> 5, co = lane & 31;
    const size_t per = (size_t)K * K * a.Cin * a.Cout;
    float* const sl = a.dwp + (size_t)slot * per;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const int p = wave + 4 * j;
      if (p >= npairs) continue;
      const int kw = p >> 1, rb = p & 1, tap = kh * K + kw;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = rb * 32 + 4 * hq + (e & 3) + 8 * (e >> 2);
        if (ci < a.Cin && co < a.Cout) sl[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[j][e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward with FEW real output channels (SRResNet's 9x9 tail: 64 -> 3; cr * K <= 32): (kw, co) pairs on the MFMA ROWS.  For an output
// row y and halo column u,
//     acc[(kw, co)][u] = sum_{kh, ci} W[co][ci][kh][kw] * X[y + kh - P][u][ci]           (K kernel rows x 4 channel steps = 4 K MFMAs per row)
// and the output is the diagonal sum  out[y][x][co] = bias[co] + sum_kw acc[(kw, co)][x + kw]  (through a wave-private LDS tile).
// lk_conv_kernel spends K * K * 4 MFMAs on the same 16 pixels with 3 of its 32 rows real.  The weights ([kh][ci / 16] fragments in the
// rows layout behind the standard pack: srk_pack_args.rows_layout) stay in registers; tiles of 8 x 16 output pixels, 16 x (16 + K - 1)
// halo pixels in LDS, two workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------
template <int DT, int K>
__global__ __launch_bounds__(256, 2) void lk_conv_rows_kernel(const srk_conv_args a, int tilesX, int tilesY, unsigned x_bytes, unsigned rows_off) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TR = 8, XW = 16 + K - 1, XWP = (XW + 1) & ~1, XR = TR + K - 1;
  constexpr int xbuf = (XR * XWP * 128 + 1023) & ~1023;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const Pw = reinterpret_cast<float*>(smem + xbuf) + wave * 1024;          // [32 rows][32 columns] fp32
  char* const Ow = smem + xbuf + 4 * 4096 + wave * 512;                            // [16 pixels][16 stored channels] of one output row
  const int H = a.H, W = a.W, P = K / 2, cr = a.cout_real;
  int pt = blockIdx.x;
  const int tX = pt % tilesX;
  pt /= tilesX;
  const int tY = pt % tilesY;
  const int n = pt / tilesY;
  const int y0 = tY * TR, x0 = tX * 16;

  // halo tile: 1 KB pieces, chunk slot XOR-swizzled by the tile column
  {
    const i32x4 xrs = make_rsrc4(a.x, x_bytes);
    const unsigned lds0 = lds_addr_of(smem);
    constexpr int NPX = (XR * XWP * 8 + 63) / 64;
#pragma unroll
    for (int j = 0; j < (NPX + 3) / 4; ++j) {
      const int k = wave + 4 * j;
      if (k < NPX) {
        const int i = k * 64 + lane, sl = i & 7, p = i >> 3;
        const int iy = p / XWP, ix = p - iy * XWP, c = sl ^ swz(ix);
        const int gy = y0 + iy - P, gx = x0 + ix - P;
        const bool ok = iy < XR && ix < XW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + c * 8) * 2) : 0x80000000u;
        dma16_hidden(xrs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (k << 10))));
      }
    }
  }
  // weights: K x 4 fragments per lane, straight from the packed buffer (L2)
  i32x4 wf[K * 4];
  {
    const i32x4* wp = reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(a.wpk) + rows_off) + lane;
#pragma unroll
    for (int f = 0; f < K * 4; ++f) wf[f] = wp[f * 64];
  }
  if (lane < 32) lds_write16(Ow + lane * 16, i32x4{0, 0, 0, 0});                   // the pad channels stay zero
  const int u = lane & 31, g = lane >> 5;
  const int ox = lane & 15, oc = lane >> 4;                                       // epilogue: lane = (pixel, real channel)
  float bias = 0.f;
  if (a.bias && oc < cr) bias = a.bias[((oc >> 2) & 3) * 8 + ((oc >> 4) & 1) * 4 + (oc & 3)];      // bias_pk is indexed by MFMA row
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll 1
  for (int yy = 0; yy < TR / 4; ++yy) {
    const int y = (TR / 4) * wave + yy;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const char* xr = smem + (y * XWP + u) * 128;
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const i32x4 b = lds_read16(xr + kh * (XWP * 128) + (((2 * ks + g) ^ swz(u)) << 4));
        acc = Tr::mma(wf[kh * 4 + ks], b, acc);
      }
#pragma unroll
    for (int e = 0; e < 16; ++e) Pw[(8 * (e >> 2) + 4 * g + (e & 3)) * 32 + u] = acc[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (oc < cr) {
      float sum = bias;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) sum += Pw[(kw * cr + oc) * 32 + ox + kw];
      *reinterpret_cast<uint16_t*>(Ow + ox * 32 + oc * 2) = Tr::from_f32(sum);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 32) {
      const int px = lane >> 1, hf = lane & 1;
      const i32x4 o = lds_read16(Ow + lane * 16);
      const int gy = y0 + y, gx = x0 + px;
      if (gy < H && gx < W)
        *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(a.out) + ((size_t)((n * H + gy) * W + gx) * a.out_pitch + a.out_coff + hf * 8) * 2) = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient when the conv has FEW real output channels (SRResNet's 9x9 tail: 3; cr * K <= 32): the MFMA columns carry
// (kw, co) pairs instead of 16 stored gradient channels of which 3 are real.  With u = x + kw - P (a column of the input halo):
//     dW[co][ci][kh][kw] = sum_{y, u} X[y + kh - P][u][ci] * dY[y][u - kw + P][co]
// so per tile row ONE operand of the input (K index = halo column u, no tap shift) meets ONE operand built from the gradient row,
// B[u][(kw, co)] = dY[y][u - kw][co]: 2 column blocks x 2 channel blocks = 4 MFMAs per tile row and kernel row instead of 2 K,
// one per wave.  The gradient tile is transposed once per tile into channel planes with K - 1 zero columns on both sides
// (DT[co][y][col + K - 1], + one all-zero plane for the unused MFMA columns), so the 8 two-byte reads of a lane's operand need
// no bounds logic.  Slab layout, bias partials and the finalize launch are lk_wgrad_kernel's.
// ------------------------------------------------------------------------------------------------------------------
template <int DT, int K>
__global__ __launch_bounds__(256) void lk_wgrad_packed_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int ntiles, int tq, int trem,
                                                              unsigned x_bytes, unsigned dy_bytes) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 + K - 1, XWP = (XW + 1) & ~1;
  constexpr int xbuf = 16 * XWP * 128, dbuf = 16 * 16 * 32, buf = xbuf + dbuf;
  constexpr int DTW = 16 + 2 * (K - 1) + 16, DTP = DTW * 2;      // plane row: columns -(K-1) .. 16 + (K-1) + 15, two bytes each
  constexpr int NPL = 8;                                          // planes: <= 6 real + the zero plane (index cr)
  constexpr int dt_bytes = NPL * 16 * DTP;
  char* const DTb = smem + 2 * buf;
  float* const red = reinterpret_cast<float*>(DTb + dt_bytes);    // 2 x 16 x 64 floats: the s = 1 waves' sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x, kh = blockIdx.y;
  const int H = a.H, W = a.W, P = K / 2, cr = a.cout_real;
  const int t0 = slot * tq + min(slot, trem), nt = tq + (slot < trem ? 1 : 0);
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);

  constexpr int NPX = 16 * XWP * 8 / 64, XPW = (NPX + 3) / 4, DPW = 2;          // + 8 gradient pieces (two per wave)
  int xdesc[XPW], ddesc[DPW];
#pragma unroll
  for (int j = 0; j < XPW; ++j) {
    const int k = wave + 4 * j, i = k * 64 + lane, sl = i & 7, p = i >> 3;
    const int iy = p / XWP, ix = p - iy * XWP, c = sl ^ swz(ix);
    xdesc[j] = iy | (ix << 8) | (c << 16) | ((k < NPX && ix < XW && c * 8 < a.Cin) ? 1 << 24 : 0);
  }
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int k = wave + 4 * j, i = k * 64 + lane, c = i & 1, pp = i >> 1;
    ddesc[j] = (pp >> 4) | ((pp & 15) << 8) | (c << 16) | ((c * 8 < a.Cout) ? 1 << 24 : 0);
  }
  auto dma_tile = [&](int tile, int b) {
    int pt = tile;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    const int n = pt / tilesY;
    const int y0 = tY * 16, x0 = tX * 16;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
      const int k = wave + 4 * j;
      if (k < NPX) {
        const int d = xdesc[j], gy = y0 + (d & 255) + kh - P, gx = x0 + ((d >> 8) & 255) - P;
        const bool ok = (d >> 24) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
        dma16_hidden(xrs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + (k << 10))));
      }
    }
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int k = wave + 4 * j;
      const int d = ddesc[j], gy = y0 + (d & 255), gx = x0 + ((d >> 8) & 255);
      const bool ok = (d >> 24) && gy < H && gx < W;
      const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.dy_pitch + a.dy_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
      dma16_hidden(drs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + xbuf + (k << 10))));
    }
  };

  // zero the planes once: the pad columns and the zero plane are never written again
  for (int i = tid; i < dt_bytes / 16; i += 256) lds_write16(DTb + i * 16, i32x4{0, 0, 0, 0});

  const int s = wave >> 1, rb = wave & 1;                         // column block (halo columns 16 s ..), channel block
  int aoff[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) aoff[rd] = tr_lane_off(16 * s, rd, rb, lane);
  // B operand: lane (n = lane & 31 -> (kw, co), g = lane >> 5) reads gradient columns 16 s + 8 g - kw + j, j = 0..7, of plane co
  const int nn = lane & 31, kwn = nn / cr, con = nn - kwn * cr;
  const int plane = kwn < K ? con : cr;                           // columns beyond K * cr: the zero plane
  const int boff = plane * 16 * DTP + (16 * s + 8 * (lane >> 5) - (kwn < K ? kwn : 0) + (K - 1)) * 2;
  // transposition pass: thread = pixel (ty, tx) of the gradient tile
  const int ty = tid >> 4, tx = tid & 15;
  const bool do_bias = a.dbp != nullptr && kh == 0;
  float dbs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (nt > 0) dma_tile(t0, 0);
  for (int it = 0; it < nt; ++it) {
    const char* const X = smem + (it & 1) * buf;
    const char* const D = X + xbuf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (it + 1 < nt) dma_tile(t0 + it + 1, (it + 1) & 1);
    {
      const i32x4 v = lds_read16(D + (ty * 16 + tx) * 32);        // channels 0..7 of the pixel (cr <= 6)
      const uint32_t w4[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
      for (int c = 0; c < 6; ++c)
        if (c < cr) {
          const uint16_t hv = (uint16_t)(w4[c >> 1] >> ((c & 1) * 16));
          *reinterpret_cast<uint16_t*>(DTb + (c * 16 + ty) * DTP + (tx + K - 1) * 2) = hv;
          if (do_bias) dbs[c] += Tr::to_f32(hv);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int y = 0; y < 16; ++y) {
      const i32x4 af = tr_read2(X + y * (XWP * 128) + aoff[0], X + y * (XWP * 128) + aoff[1]);
      const uint16_t* bp = reinterpret_cast<const uint16_t*>(DTb + boff + y * DTP);
      i32x4 bf;
      bf.x = (int)((uint32_t)bp[0] | ((uint32_t)bp[1] << 16));
      bf.y = (int)((uint32_t)bp[2] | ((uint32_t)bp[3] << 16));
      bf.z = (int)((uint32_t)bp[4] | ((uint32_t)bp[5] << 16));
      bf.w = (int)((uint32_t)bp[6] | ((uint32_t)bp[7] << 16));
      acc = Tr::mma(af, bf, acc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  if (do_bias) {                                                  // db partial of this slot: block sum of the per-thread sums
    float* rs = reinterpret_cast<float*>(smem);                   // (the tile buffers are free now)
#pragma unroll
    for (int c = 0; c < 6; ++c) rs[c * 256 + tid] = dbs[c];
    __syncthreads();
    if (tid < a.Cout) {
      float t = 0.f;
      if (tid < cr)
        for (int i = 0; i < 256; ++i) t += rs[tid * 256 + i];
      a.dbp[(size_t)slot * a.Cout + tid] = t;
    }
    __syncthreads();
  }
  // the two column blocks' sums meet in LDS; rows = input channels rb * 32 + 8 (e >> 2) + 4 (lane >> 5) + (e & 3), column = (kw, co)
  if (s == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(rb * 16 + e) * 64 + lane] = acc[e];
  }
  __syncthreads();
  const size_t per = (size_t)K * K * a.Cin * a.Cout;
  float* const sl = a.dwp + (size_t)slot * per + (size_t)kh * K * a.Cin * a.Cout;
  if (s == 0 && kwn < K) {
    const int hq = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = rb * 32 + 4 * hq + (e & 3) + 8 * (e >> 2);
      if (ci < a.Cin) sl[((size_t)kwn * a.Cin + ci) * a.Cout + con] = acc[e] + red[(rb * 16 + e) * 64 + lane];
    }
  }
  // the stored channels that are padding: zeros (the finalize step adds whole slabs)
  for (int i = tid; i < K * a.Cin * a.Cout; i += 256)
    if (i % a.Cout >= cr) sl[i] = 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// The same with ALL kernel rows in one workgroup: tiles of 8 x 16 output pixels, whose 16 x (16 + K - 1)-pixel input halo serves the
// K kernel rows of the 8 output rows (the one-row-per-workgroup form reads the input K times: 1.2 GB into LDS per launch for
// SRResNet's 9x9 tail at 16 x 192 x 192).  K accumulators per wave; slabs are COMPACT: [tap][ci][4] (cout_real <= 4), one per
// workgroup (srk_wgrad_slab_cout tells the caller).
// ------------------------------------------------------------------------------------------------------------------
constexpr int LK_TR = 8, LK_CS = 4;
template <int DT, int K>
__global__ __launch_bounds__(256) void lk_wgrad_allrows_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int ntiles, int tq, int trem,
                                                               unsigned x_bytes, unsigned dy_bytes) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TR = LK_TR, CS = LK_CS;
  constexpr int XW = 16 + K - 1, XWP = (XW + 1) & ~1, XR = TR + K - 1;
  constexpr int xbuf = (XR * XWP * 128 + 1023) & ~1023, dbuf = TR * 16 * 32, buf = xbuf + dbuf;     // whole 1 KB DMA pieces
  constexpr int DTW = 16 + 2 * (K - 1) + 16, DTP = DTW * 2;
  constexpr int NPL = 8;
  constexpr int dt_bytes = NPL * TR * DTP;
  static_assert(K * 2 * 16 * 64 * 4 <= 2 * buf, "the final reduction reuses the tile buffers");
  char* const DTb = smem + 2 * buf;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x;
  const int H = a.H, W = a.W, P = K / 2, cr = a.cout_real;
  const int t0 = slot * tq + min(slot, trem), nt = tq + (slot < trem ? 1 : 0);
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);

  constexpr int NPX = (XR * XWP * 8 + 63) / 64, XPW = (NPX + 3) / 4;      // + 4 gradient pieces (one per wave)
  int xdesc[XPW], ddesc;
#pragma unroll
  for (int j = 0; j < XPW; ++j) {
    const int k = wave + 4 * j, i = k * 64 + lane, sl = i & 7, p = i >> 3;
    const int iy = p / XWP, ix = p - iy * XWP, c = sl ^ swz(ix);
    xdesc[j] = iy | (ix << 8) | (c << 16) | ((k < NPX && iy < XR && ix < XW && c * 8 < a.Cin) ? 1 << 24 : 0);
  }
  {
    const int i = wave * 64 + lane, c = i & 1, pp = i >> 1;        // 2 chunks per pixel, pixels row-major 8 x 16
    ddesc = (pp >> 4) | ((pp & 15) << 8) | (c << 16) | ((c * 8 < a.Cout) ? 1 << 24 : 0);
  }
  auto dma_tile = [&](int tile, int b) {
    int pt = tile;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    const int n = pt / tilesY;
    const int y0 = tY * TR, x0 = tX * 16;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
      const int k = wave + 4 * j;
      if (k < NPX) {
        const int d = xdesc[j], gy = y0 + (d & 255) - P, gx = x0 + ((d >> 8) & 255) - P;
        const bool ok = (d >> 24) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.x_pitch + a.x_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
        dma16_hidden(xrs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + (k << 10))));
      }
    }
    {
      const int d = ddesc, gy = y0 + (d & 255), gx = x0 + ((d >> 8) & 255);
      const bool ok = (d >> 24) && gy < H && gx < W;
      const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * a.dy_pitch + a.dy_coff + ((d >> 16) & 255) * 8) * 2) : 0x80000000u;
      dma16_hidden(drs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * buf + xbuf + (wave << 10))));
    }
  };

  for (int i = tid; i < dt_bytes / 16; i += 256) lds_write16(DTb + i * 16, i32x4{0, 0, 0, 0});

  const int s = wave >> 1, rb = wave & 1;
  int aoff[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) aoff[rd] = tr_lane_off(16 * s, rd, rb, lane);
  const int nn = lane & 31, kwn = nn / cr, con = nn - kwn * cr;
  const int plane = kwn < K ? con : cr;
  const int boff = plane * TR * DTP + (16 * s + 8 * (lane >> 5) - (kwn < K ? kwn : 0) + (K - 1)) * 2;
  const int ty = tid >> 4, tx = tid & 15;                          // transposition pass: threads 0 .. 127 = the tile's pixels
  const bool do_bias = a.dbp != nullptr;
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[K];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[kh][e] = 0.f;
  if (nt > 0) dma_tile(t0, 0);
  for (int it = 0; it < nt; ++it) {
    const char* const X = smem + (it & 1) * buf;
    const char* const D = X + xbuf;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (it + 1 < nt) dma_tile(t0 + it + 1, (it + 1) & 1);
    if (tid < TR * 16) {
      const i32x4 v = lds_read16(D + (ty * 16 + tx) * 32);
      const uint32_t w4[4] = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
#pragma unroll
      for (int c = 0; c < CS; ++c)
        if (c < cr) {
          const uint16_t hv = (uint16_t)(w4[c >> 1] >> ((c & 1) * 16));
          *reinterpret_cast<uint16_t*>(DTb + (c * TR + ty) * DTP + (tx + K - 1) * 2) = hv;
          if (do_bias) dbs[c] += Tr::to_f32(hv);
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int y = 0; y < TR; ++y) {
      const uint16_t* bp = reinterpret_cast<const uint16_t*>(DTb + boff + y * DTP);
      i32x4 bf;
      bf.x = (int)((uint32_t)bp[0] | ((uint32_t)bp[1] << 16));
      bf.y = (int)((uint32_t)bp[2] | ((uint32_t)bp[3] << 16));
      bf.z = (int)((uint32_t)bp[4] | ((uint32_t)bp[5] << 16));
      bf.w = (int)((uint32_t)bp[6] | ((uint32_t)bp[7] << 16));
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const char* xr = X + (y + kh) * (XWP * 128);
        const i32x4 af = tr_read2(xr + aoff[0], xr + aoff[1]);
        acc[kh] = Tr::mma(af, bf, acc[kh]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  if (do_bias) {
    float* rs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int c = 0; c < CS; ++c) rs[c * 256 + tid] = dbs[c];
    __syncthreads();
    if (tid < CS) {
      float t = 0.f;
      if (tid < cr)
        for (int i = 0; i < 256; ++i) t += rs[tid * 256 + i];
      a.dbp[(size_t)slot * CS + tid] = t;
    }
    __syncthreads();
  }
  float* const red = reinterpret_cast<float*>(smem);               // [kh][rb][16][64]
  if (s == 1) {
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[((kh * 2 + rb) * 16 + e) * 64 + lane] = acc[kh][e];
  }
  __syncthreads();
  const size_t per = (size_t)K * K * a.Cin * CS;
  float* const sl = a.dwp + (size_t)slot * per;
  if (s == 0 && kwn < K) {
    const int hq = lane >> 5;
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = rb * 32 + 4 * hq + (e & 3) + 8 * (e >> 2);
        if (ci < a.Cin) sl[(((size_t)kh * K + kwn) * a.Cin + ci) * CS + con] = acc[kh][e] + red[((kh * 2 + rb) * 16 + e) * 64 + lane];
      }
  }
  for (int i = tid; i < K * K * a.Cin * CS; i += 256)
    if ((i & (CS - 1)) >= cr) sl[i] = 0.f;
}

// whether the all-rows form takes this weight gradient (and its slabs are the compact [tap][ci][4] ones)
// ---- 5x5 forward, 64 input -> <= 16 stored output channels, fp32 NCHW store behind PixelShuffle(2): the collapsed HR stage -------------
// (hr_tail.hip: the image = one 5x5 conv of the upsampler's input.)  lk_conv_kernel gives every 16 x 16 tile its own workgroup and streams
// the weights per kernel row: one workgroup per CU (91 KB of LDS), nothing of tile t + 1 overlaps tile t -- 375 us at 256 x 96 x 96 for
// 160 us of MFMA work.  Here: PERSISTENT workgroups (one per CU, a contiguous range of tiles), the weights of all 25 taps stationary in
// LDS -- only the 16 MFMA rows that carry real channels (row_to_chan: rows 8 i + e, i, e < 4), 51 KB; the other 16 row lanes read a
// duplicate and their results are never stored -- and the 20 x 20 halo tile double-buffered by hidden LDS-DMA (2 x 51 KB): the next
// tile's DMA is issued right behind the barrier that releases its buffer and lands under the current tile's 200 MFMAs per wave.
template <int DT>
__global__ __launch_bounds__(256) void lk5_fwd_kernel(const srk_conv_args a, int tilesX, int tilesY, int tq, int trem, unsigned x_bytes) {
  typedef DTraits<DT> Tr;
  constexpr int XT = 20, XB = XT * XT * 128, WB = 25 * 8 * 16 * 16;      // halo tile 51,200 B; compact weights 51,200 B
  constexpr int NPC = XB / 1024;                                         // 50 pieces of 1 KB per halo tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wl = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const int slot = blockIdx.x;
  const int t0 = slot * tq + min(slot, trem), nt = tq + (slot < trem ? 1 : 0);
  if (nt <= 0) return;
  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes), wrsrc = make_rsrc4(a.wpk, 25u * 8u * 32u * 16u);
  const unsigned lds0 = lds_addr_of(smem);

  // bias -> registers first (the vector-memory counter retires in order: nothing hidden may be queued in front of a visible load)
  f32x16 bias16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 4 * h + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    bias16[4 * i + 0] = b.x; bias16[4 * i + 1] = b.y; bias16[4 * i + 2] = b.z; bias16[4 * i + 3] = b.w;
  }
  float pa[4] = {0.f, 0.f, 0.f, 0.f};
  const int O = a.Cout >> 2;
  if (a.post_add) {
#pragma unroll
    for (int o = 0; o < 4; ++o) pa[o] = o < O ? a.post_add[o] : 0.f;
  }
  asm volatile("" : "+v"(bias16), "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // stationary weights: compact row j = 4 i + e  <-  packed row 8 i + e of (tap, chunk); piece p = (tap*8 + chunk)*16 + j
#pragma unroll 1
  for (int k = wave; k < WB / 1024; k += 4) {
    const int p = k * 64 + lane, j = p & 15, tc = p >> 4;
    dma16_hidden(wrsrc, (unsigned)((tc * 32 + 8 * (j >> 2) + (j & 3)) * 16), (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (k << 10))));
  }
  // halo pieces of this lane: piece k = wave + 4 m (m < 13): slot sl of halo pixel (iy, ix), source chunk sl ^ swz(ix)
  constexpr int NPW = (NPC + 3) / 4;
  int hconst[NPW], hyx[NPW];
#pragma unroll
  for (int m = 0; m < NPW; ++m) {
    const int k = wave + 4 * m, i = k * 64 + lane, sl = i & 7, p = i >> 3;
    const int iy = p / XT, ix = p - iy * XT, c = sl ^ swz(ix);
    hconst[m] = (((iy - 2) * W + (ix - 2)) * a.x_pitch + a.x_coff + c * 8) * 2;
    hyx[m] = ((iy - 2) & 0xffff) | ((ix - 2) << 16);
  }
  auto tile_of = [&](int t, int& n, int& y0, int& x0) {
    int pt = t0 + t;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    n = pt / tilesY; y0 = tY * 16; x0 = tX * 16;
  };
  auto dma_tile = [&](int t, int b) {
    int n, y0, x0;
    tile_of(t, n, y0, x0);
    const int tbase = ((n * H + y0) * W + x0) * a.x_pitch * 2;
#pragma unroll
    for (int m = 0; m < NPW; ++m) {
      const int k = wave + 4 * m;
      if (k >= NPC) continue;                                            // wave-uniform
      const int gy = y0 + (int)(short)(hyx[m] & 0xffff), gx = x0 + (hyx[m] >> 16);
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      dma16_hidden(xrsrc, ok ? (unsigned)(tbase + hconst[m]) : 0x80000000u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + WB + b * XB + (k << 10))));
    }
  };
  dma_tile(0, 0);

  const int px = r & 15;
  int prow[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) prow[pb] = 4 * wave + 2 * pb + (r >> 4);
  // A fragment: row lane r -> compact row (real rows 8 i + e; the others take a duplicate), chunk half h
  const int jrow = ((r >> 2) & 1) ? 0 : 4 * (r >> 3) + (r & 3);
  const char* const wl = Wl + ((h * 16 + jrow) << 4);
  int gsw[5];
#pragma unroll
  for (int kw = 0; kw < 5; ++kw) gsw[kw] = swz(px + kw);

  const int H2 = 2 * H, W2 = 2 * W;
  const float sc = a.scale;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
  for (int t = 0; t < nt; ++t) {
    const char* const Xs = smem + WB + (t & 1) * XB;
    // tile t (and, first, the weights) landed; the other buffer is free.  The counted wait leaves the previous tile's 4 O image stores
    // (always issued, out-of-range offsets for pixels outside the image) in flight: they are the youngest vector-memory operations,
    // the DMA pieces of tile t are older.  (vmcnt(0) here made every tile wait for its predecessor's stores to be acknowledged.)
    if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (O == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (O == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (O == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < nt) dma_tile(t + 1, (t + 1) & 1);
    f32x16 acc[2] = {bias16, bias16};
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const i32x4 af = lds_read16(wl + ((((kh * 5 + kw) * 8 + 2 * ks) * 16) << 4));
          i32x4 bf[2];
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
            bf[pb] = lds_read16(Xs + ((prow[pb] + kh) * XT + px + kw) * 128 + (((2 * ks + h) ^ gsw[kw]) << 4));
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[pb] = Tr::mma(af, bf[pb], acc[pb]);
        }
      }
    }
    // fp32 NCHW behind PixelShuffle(2): channel k = o*4 + i*2 + j of pixel (gy, gx) -> out[n][o][2 gy + i][2 gx + j] (+ post_add[o]); the h = 0
    // lanes hold the stored channels (register index = channel); a lane's (j = 0, 1) pair is one 8-byte store
    {
      int n, y0, x0;
      tile_of(t, n, y0, x0);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int gy = y0 + prow[pb], gx = x0 + px;
        const bool ok = h == 0 && gy < H && gx < W;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (o >= O) break;                                             // wave-uniform
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
            u32x2_t v;
            v.x = __float_as_uint(acc[pb][o * 4 + i * 2] * sc + pa[o]);
            v.y = __float_as_uint(acc[pb][o * 4 + i * 2 + 1] * sc + pa[o]);
            const unsigned off = ok ? (unsigned)((((n * O + o) * H2 + 2 * gy + i) * W2 + 2 * gx) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, off, 0, SRK_AUX_WT);
          }
        }
      }
    }
  }
}

// ---- the same conv with (kernel column, channel) PAIRS on the MFMA rows (<= 12 stored channels = 3 colours x 4 sub-pixels) ---------------
// lk5_fwd_kernel gives every tap its own MFMA although only 12 of its 32 rows carry a channel, and reads 1.5 fragments from LDS per MFMA:
// LDS-bound (313 us at 256 x 96 x 96; 36 tiles x 800 MFMAs per image).  Here the 5 kernel columns share an MFMA pair: row R = kw * 12 + co of a
// 64-row block (60 used), column = one INPUT pixel of a 32-pixel run of one image row, K = 16 input channels:
//     D[kw * 12 + co][c] = sum_{kh, ci} W[co][ci][kh][kw] * X[y + kh - 2][x0 - 2 + c][ci]        out[co][y][x0 + x] = sum_kw D[kw * 12 + co][x + kw]
// -- the kernel rows and the input channels accumulate inside the MFMA chain (5 x 4 steps x 2 row blocks = 40 MFMAs per output row of 28
// pixels: 1.9x fewer than the tap form), the five column taps meet in a 8.5 KB per-wave fp32 scratch (written in the accumulator layout, read
// back shifted).  The 40 weight fragments of a lane are STATIONARY IN REGISTERS (160 VGPRs, loaded once per persistent workgroup), so an MFMA
// pair costs ONE 16-byte LDS read per lane (0.5 per MFMA).  A workgroup walks a 28-column band of an image downwards, four output rows per
// step (one per wave), over a 16-row ring of the swizzled 128-byte image (srk_common.h) fed by hidden LDS-DMA two steps ahead: every input row
// is read once per band (x 32 / 28 for the column halo).  The epilogue of step s - 1 (scratch, shifted sums, PixelShuffle(2) store) is
// issued BETWEEN the MFMAs of step s.
// NW waves = NW output rows per step.  NW = 4 (one wave per SIMD, rows requested two steps ahead: a 16-row ring) or NW = 8 (two waves per
// SIMD -- one wave's scratch traffic, address arithmetic and waits under the other's MFMAs; rows one step ahead: a 20-row ring).
template <int DT, int NW>
__global__ __launch_bounds__(NW * 64) void lk5_rows_fwd_kernel(const srk_conv_args a, int nb, int segs, int seg_rows, int units, unsigned x_bytes) {
  typedef DTraits<DT> Tr;
  constexpr int AHEAD = NW == 4 ? 2 : 1, PRO = NW + 4 + (AHEAD - 1) * NW, RING = PRO + NW;
  constexpr int ROWB = 32 * 128, SP = 68, SCR = 32 * SP * 4;      // ring row 4,096 B; scratch [column][row], pitch 68 floats: 8,704 B per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, O = a.Cout >> 2;
  float* const scr = reinterpret_cast<float*>(smem + RING * ROWB + wave * SCR);
  const unsigned lds0 = lds_addr_of(smem);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpk), 0, 25 * 8 * 32 * 16, 0x00020000);
  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);

  // ---- stationary operands: bias / post_add of this lane's six outputs, the 40 weight fragments ----------------------------------------
  // epilogue lane (x = lane & 31, i = lane >> 5) owns out[o][2 y + i][2 (x0 + x) + j], channel co = o*4 + i*2 + j (packed bias row 8 (co >> 2) + (co & 3))
  float bias6[6], pa3[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    pa3[o] = (a.post_add && o < O) ? a.post_add[o] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) bias6[o * 2 + j] = (a.bias && o < O) ? a.bias[8 * o + h * 2 + j] : 0.f;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  i32x4 A[5][4][2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int R = m * 32 + r, kw = R / 12, co = R - kw * 12;
    const bool ok = R < 60 && co < a.Cout;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned off = ok ? (unsigned)(((((kh * 5 + kw) * 8 + 2 * ks + h) * 32 + 8 * (co >> 2) + (co & 3)) << 4)) : 0x80000000u;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0);
        A[kh][ks][m] = i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
      }
  }
  // per-lane constants of the four 1 KB pieces of an input row: piece pc holds pixels 8 pc .. 8 pc + 7, lane -> (pixel, chunk slot)
  int ccol[4], cpx[4];
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int px = pc * 8 + (lane >> 3), sl = lane & 7, c = sl ^ swz(px);
    cpx[pc] = px - 2;
    ccol[pc] = ((px - 2) * a.x_pitch + a.x_coff + c * 8) * 2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing visible may be queued behind a hidden load (in-order retirement)

  const int bsw = swz(r);
  const int H2 = 2 * H, W2 = 2 * W;
  const float sc = a.scale;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);

  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int sgm = u % segs, bb = (u / segs) % nb, n = u / (segs * nb);
    const int x0 = bb * 28, ys = sgm * seg_rows;
    const int ye = min(H, ys + seg_rows), nx = min(28, W - x0);
    const int steps = (ye - ys + NW - 1) / NW, qmax = (ye - ys) + 4;      // input rows q = 0 .. qmax - 1 <-> image rows ys - 2 + q
    auto dma_row = [&](int q) {                                       // wave-uniform q; ALWAYS four operations (the counted waits rely on it):
      const int gy = ys - 2 + q;                                      // rows behind the segment's last land as zeros in a free slot
      const bool rok = (unsigned)gy < (unsigned)H && q < qmax;
      const int base = ((n * H + gy) * W + x0) * a.x_pitch * 2;
      const unsigned dst = lds0 + (unsigned)((q % RING) * ROWB);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const bool ok = rok && (unsigned)(x0 + cpx[pc]) < (unsigned)W;
        dma16_hidden(xrsrc, ok ? (unsigned)(base + ccol[pc]) : 0x80000000u, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + pc * 1024)));
      }
    };
    // the previous unit's last MFMAs have read the ring and its last (zero) rows have landed -- LDS-DMA of different waves is not ordered --
    // before the first rows of this unit are requested
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int q = wave; q < PRO; q += NW) dma_row(q);

    f32x16 accp[2];                                      // the step whose epilogue is still owed
    int gyp = 0;
    // scratch[c][R]: the four registers 4 i .. 4 i + 3 of an accumulator are rows 8 i + 4 h .. + 3 of column r: one 16-byte write
    auto epi_write = [&](int g0, int cnt) {              // register groups g0 .. g0 + cnt - 1 (of 8: m = g >> 2, i = g & 3)
#pragma unroll
      for (int g = g0; g < g0 + cnt; ++g) {
        const int m = g >> 2, i = g & 3;
        *reinterpret_cast<f32x4*>(scr + r * SP + m * 32 + 8 * i + 4 * h) = f32x4{accp[m][4 * i], accp[m][4 * i + 1], accp[m][4 * i + 2], accp[m][4 * i + 3]};
      }
    };
    float ev[6];
    auto epi_read = [&](int t0, int cnt) {               // terms t = o*5 + kw: both sub-pixel columns j = 0, 1 of colour o in one 8-byte read
#pragma unroll
      for (int t = t0; t < t0 + cnt; ++t) {
        const int o = t / 5, kw = t - o * 5;
        const f32x2 v = *reinterpret_cast<const f32x2*>(scr + (r + kw) * SP + kw * 12 + o * 4 + h * 2);
        ev[o * 2] = kw == 0 ? v.x : ev[o * 2] + v.x;
        ev[o * 2 + 1] = kw == 0 ? v.y : ev[o * 2 + 1] + v.y;
      }
    };
    auto epi_store = [&]() {
      const bool ok = r < nx && gyp < ye;
#pragma unroll
      for (int o = 0; o < 3; ++o) {                      // always three stores: the counted waits below rely on it
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
        u32x2_t v;
        v.x = __float_as_uint((ev[o * 2] + bias6[o * 2]) * sc + pa3[o]);
        v.y = __float_as_uint((ev[o * 2 + 1] + bias6[o * 2 + 1]) * sc + pa3[o]);
        const unsigned off = (ok && o < O) ? (unsigned)((((n * O + o) * H2 + 2 * gyp + h) * W2 + 2 * (x0 + r)) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, off, 0, SRK_AUX_WT);
      }
    };
    auto step = [&](int s, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      // rows q < NW s + NW + 4 have landed.  Issue order per wave: P (the first PRO rows), then per step s: D_s (one row: four pieces) and,
      // from step 1 on, E_s (the three image stores of step s - 1's epilogue).  Step s needs D_(s-AHEAD): behind it sit E_(s-AHEAD) and the
      // D, E of the steps between.
      if (!EPI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (AHEAD == 1) { if (s == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
      else if (s == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (s == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma_row(NW * s + PRO + wave);
      const int q0 = NW * s + wave;
      const char* rowp[5];
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) rowp[kh] = smem + ((q0 + kh) % RING) * ROWB + r * 128;
      f32x16 acc[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
      // the pixel fragments run PF MFMA pairs ahead of their use (PF x 4 registers: the 160 registers of weights must stay in VGPRs)
      constexpr int PF = 6;
      i32x4 bf[PF];
      auto bfrag = [&](int k) { return lds_read16(rowp[k >> 2] + (((2 * (k & 3) + h) ^ bsw) << 4)); };
#pragma unroll
      for (int k = 0; k < PF; ++k) bf[k] = bfrag(k);
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        const i32x4 b = bf[k % PF];
        if (k + PF < 20) bf[k % PF] = bfrag(k + PF);
        if (EPI) {                                        // the previous step's epilogue, a few LDS operations per MFMA pair
          if (k < 8) epi_write(k, 1);
          else if (k == 8) asm volatile("" ::: "memory");      // lanes read what OTHER lanes wrote: the compiler must not carry scratch values across
          else if (k >= 9 && k < 19) epi_read((k - 9) * 3 / 2, (k - 8) * 3 / 2 - (k - 9) * 3 / 2);
          else if (k == 19) epi_store();
        }
        acc[0] = Tr::mma(A[k >> 2][k & 3][0], b, acc[0]);
        acc[1] = Tr::mma(A[k >> 2][k & 3][1], b, acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      accp[0] = acc[0]; accp[1] = acc[1];
      gyp = ys + q0;
    };
    step(0, std::false_type{});
#pragma unroll 1
    for (int s = 1; s < steps; ++s) step(s, std::true_type{});
    epi_write(0, 8);
    asm volatile("" ::: "memory");      // (the LDS itself executes a wave's operations in order: no wait is needed, only the compiler's)
    epi_read(0, 15);
    epi_store();
  }
}

// ---- 5x5 data gradient of the same stage: 16 stored gradient channels -> 64 channels, NHWC ------------------------------------------------
// lk_conv_kernel<DT, 2, 2, 5> streams the weights through LDS per kernel row and reads 1.5 fragments per MFMA pair from a 32-byte-per-pixel
// tile whose rows are 640 bytes apart (7.4 M bank conflicts per launch at 256 x 96 x 96: 151 us).  K is ONE MFMA step here (16 channels), so
// a tap is an MFMA pair (2 x 32 output rows) and the 50 weight fragments of a lane stay in registers (200 VGPRs) for the life of a
// persistent workgroup; a pixel fragment (32 pixels of one image row x 16 channels, 1 KB) feeds both MFMAs: 0.5 LDS reads per MFMA, and
// consecutive lanes read consecutive 32-byte pixels (conflict-free without a swizzle).  Same walk as lk5_rows_fwd_kernel: a workgroup
// owns a 32-column band, four output rows per step (one per wave), a 16-row ring of 36-pixel rows fed two steps ahead by hidden LDS-DMA.
template <int DT>
__global__ __launch_bounds__(256) void lk5_dgrad_kernel(const srk_conv_args a, int nb, int segs, int seg_rows, int units, unsigned x_bytes) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int RING = 16, ROWB = 36 * 32;               // ring row: 36 pixels x 32 bytes = 72 chunks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const unsigned lds0 = lds_addr_of(smem);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpk), 0, 25 * 2 * 64 * 16, 0x00020000);
  const i32x4 xrsrc = make_rsrc4(a.x, x_bytes);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  i32x4 A[25][2];
#pragma unroll
  for (int t = 0; t < 25; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)((((t * 2 + h) * 64 + m * 32 + r) << 4)), 0, 0);
      A[t][m] = i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  // a ring row is 72 chunks of 16 bytes: chunk ch = pixel (ch >> 1), channel half (ch & 1); DMA 0 moves chunks 0 .. 63, DMA 1 (lanes 0 .. 7) 64 .. 71
  const int c0 = ((lane >> 1) - 2) * a.x_pitch * 2 + (a.x_coff + (lane & 1) * 8) * 2, p0 = (lane >> 1) - 2;
  const int c1 = ((32 + (lane >> 1)) - 2) * a.x_pitch * 2 + (a.x_coff + (lane & 1) * 8) * 2, p1 = 32 + (lane >> 1) - 2;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing visible may be queued behind a hidden load (in-order retirement)

  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int sgm = u % segs, bb = (u / segs) % nb, n = u / (segs * nb);
    const int x0 = bb * 32, ys = sgm * seg_rows;
    const int ye = min(H, ys + seg_rows);
    const int steps = (ye - ys + 3) >> 2, qmax = (ye - ys) + 4;      // input rows q = 0 .. qmax - 1 <-> image rows ys - 2 + q
    auto dma_row = [&](int q) {                                       // wave-uniform q; ALWAYS two operations (the counted waits rely on it)
      const int gy = ys - 2 + q;
      const bool rok = (unsigned)gy < (unsigned)H && q < qmax;
      const int base = ((n * H + gy) * W + x0) * a.x_pitch * 2;
      const unsigned dst = lds0 + (unsigned)((q % RING) * ROWB);
      dma16_hidden(xrsrc, (rok && (unsigned)(x0 + p0) < (unsigned)W) ? (unsigned)(base + c0) : 0x80000000u, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
      if (lane < 8)
        dma16_hidden(xrsrc, (rok && (unsigned)(x0 + p1) < (unsigned)W) ? (unsigned)(base + c1) : 0x80000000u, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 1024)));
    };
    // the previous unit's last MFMAs have read the ring and its last rows have landed (LDS-DMA of different waves is not ordered)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dma_row(wave);
    dma_row(4 + wave);
    dma_row(8 + wave);
#if SRK_LK5_STAMPS
    unsigned long long* const stamp = (blockIdx.x == 0 && tid == 0 && u == blockIdx.x) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.post_add)) : nullptr;
#define SRK_LSTAMP(i) do { if (stamp && s < 24) stamp[s * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SRK_LSTAMP(i) do { } while (0)
#endif
    // The epilogue of step s - 1 is issued BETWEEN the MFMAs of step s (stand-alone it was 910 of a step's 3,220 cycles: the wave is alone
    // on its SIMD).  Lane (pixel r, half h) holds channels 32 h + 8 k + 0 .. 7, k = 0 .. 3, of its pixel (row_to_chan).  Stored as they
    // sit, a store instruction is 64 separate 16-byte requests (16.6 M per launch at 256 x 96 x 96); after a 4 x 4 transpose of the pieces
    // inside every quad of lanes, lane qi holds piece qi of the quad's four pixels and four neighbouring lanes write 64 contiguous bytes.
    f32x16 accp[2];
    int gyp = 0;
    uint32_t P[4][4];
    auto epi_cvt = [&](int k) {
#pragma unroll
      for (int d = 0; d < 4; ++d) P[k][d] = pack2<DT>(accp[k >> 1][8 * (k & 1) + 2 * d], accp[k >> 1][8 * (k & 1) + 2 * d + 1]);
    };
    auto epi_store = [&](int j) {                          // always issued (the counted waits rely on it)
      const int qi = r & 3, gxq = x0 + (r & ~3);
      const bool ok = gyp < ye && gxq + j < W;
      const unsigned off = (unsigned)((((n * H + gyp) * W + gxq + j) * a.out_pitch + a.out_coff + 32 * h + 8 * qi) * 2);
      const u32x4_t ov = {P[j][0], P[j][1], P[j][2], P[j][3]};
      __builtin_amdgcn_raw_buffer_store_b128(ov, orsrc, ok ? off : 0x80000000u, 0, SRK_AUX_WT);
    };
    auto step = [&](int s, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      SRK_LSTAMP(0);
      // rows q < 4 s + 8 have landed.  Issue order per wave: P (6 operations), then per step s: D_s (2) and, from step 1 on, E_s (the four
      // stores of step s - 1's epilogue).  Step s needs D_(s-2): behind it sit E_(s-2), D_(s-1), E_(s-1).
      if (!EPI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (s == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (s == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      SRK_LSTAMP(1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      SRK_LSTAMP(2);
      dma_row(4 * s + 12 + wave);
      const int q0 = 4 * s + wave;
      const char* rowp[5];
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) rowp[kh] = smem + ((q0 + kh) % RING) * ROWB + r * 32 + h * 16;
      f32x16 acc[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
      constexpr int PF = 6;
      i32x4 bf[PF];
      auto bfrag = [&](int t) { return lds_read16(rowp[t / 5] + (t % 5) * 32); };
#pragma unroll
      for (int t = 0; t < PF; ++t) bf[t] = bfrag(t);
      // taps in pairs, a0 a0 a1 a1: an MFMA that accumulates onto the result of the one issued right before it starts without a bubble,
      // one that follows a DIFFERENT accumulator's waits for its own predecessor to drain (alternating a0 a1 measured 37.6 cycles per MFMA)
#pragma unroll
      for (int t = 0; t < 25; t += 2) {
        const i32x4 b0 = bf[t % PF], b1 = bf[(t + 1) % PF];
        if (t + PF < 25) bf[t % PF] = bfrag(t + PF);
        if (t + 1 + PF < 25) bf[(t + 1) % PF] = bfrag(t + 1 + PF);
        if (EPI) {
          if (t >= 2 && t < 10) epi_cvt((t - 2) >> 1);
          else if (t == 10) quad_transpose8_dpp(P[0][0], P[1][0], P[2][0], P[3][0], P[0][1], P[1][1], P[2][1], P[3][1]);
          else if (t == 12) quad_transpose8_dpp(P[0][2], P[1][2], P[2][2], P[3][2], P[0][3], P[1][3], P[2][3], P[3][3]);
          else if (t >= 14 && t < 22) epi_store((t - 14) >> 1);
        }
        acc[0] = Tr::mma(A[t][0], b0, acc[0]);
        if (t + 1 < 25) acc[0] = Tr::mma(A[t + 1][0], b1, acc[0]);
        acc[1] = Tr::mma(A[t][1], b0, acc[1]);
        if (t + 1 < 25) acc[1] = Tr::mma(A[t + 1][1], b1, acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      SRK_LSTAMP(3);
      accp[0] = acc[0]; accp[1] = acc[1];
      gyp = ys + q0;
    };
    step(0, std::false_type{});
#pragma unroll 1
    for (int s = 1; s < steps; ++s) step(s, std::true_type{});
#pragma unroll
    for (int k = 0; k < 4; ++k) epi_cvt(k);
    quad_transpose8_dpp(P[0][0], P[1][0], P[2][0], P[3][0], P[0][1], P[1][1], P[2][1], P[3][1]);
    quad_transpose8_dpp(P[0][2], P[1][2], P[2][2], P[3][2], P[0][3], P[1][3], P[2][3], P[3][3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) epi_store(j);
  }
}

// ---- 5x5 weight gradient, 64 input x 16 stored gradient channels: the collapsed HR stage (hr_tail.hip) -----------------------------
// dW[f][ci][co] = sum_q X[q][ci] dY[q - f][co]: the TAP SHIFT IS ON THE GRADIENT, not on x.  So a tile is the 16 x 16 pixels of x
// WITHOUT a halo (32 KB instead of the 51 KB of a 20 x 20 halo: x is the operand that costs bandwidth, 302 MB per launch at 256 x 96 x 96)
// plus the 20 x 20 halo of the 32-byte-per-pixel gradient (12.8 KB), and one workgroup owns ALL 25 taps (lk_wgrad_kernel gives a
// workgroup one kernel row: five workgroups re-read every x tile, 828 us for this shape, DMA-bound).  The 16 gradient channels fill
// only half of a 32-column MFMA, so each MFMA carries TWO taps: columns 0-15 read the gradient shifted by tap A, columns 16-31 by
// tap B (the transposing LDS read takes its address per 16-lane group).  Pairs are (fy, fx) & (fy + 1, fx): the two groups' 128-byte
// runs then sit 640 bytes apart = in opposite halves of the 256-byte bank row (conflict-free); (2,-2) & (2,2) likewise (128 bytes
// apart); the last three taps of row fy = 2 ride alone.  14 pair tiles x 2 input-channel blocks = 28 accumulator tiles, 7 per wave.
// K = the 16 pixels of a tile row; x fragments by the same transposing read from the swizzled 128-byte image (srk_common.h).
// Double-buffered by hidden LDS-DMA; partial sums to per-workgroup slabs in srk_wgrad_finalize's layout [tap][ci][16].
SRK_DEV void lk5_pair(int t, int& fyA, int& fxA, int& fyB, int& fxB, bool& hasB) {
  hasB = true;
  if (t < 5) { fyA = -2; fxA = t - 2; fyB = -1; fxB = t - 2; }
  else if (t < 10) { fyA = 0; fxA = t - 7; fyB = 1; fxB = t - 7; }
  else if (t == 10) { fyA = 2; fxA = -2; fyB = 2; fxB = 2; }
  else { fyA = 2; fxA = t - 12; fyB = 2; fxB = t - 12; hasB = false; }
}

template <int DT>
__global__ __launch_bounds__(256) void lk5_wgrad_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int tq, int trem,
                                                        unsigned x_bytes, unsigned dy_bytes) {
  typedef DTraits<DT> Tr;
  constexpr int XB = 16 * 16 * 128, DP = 20, DB = 13 * 1024, BUF = XB + DB;      // gradient halo: 20 x 20 x 32 B = 12,800 (+ DMA slack)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = blockIdx.x;
  const int H = a.H, W = a.W;
  const int t0 = slot * tq + min(slot, trem), nt = tq + (slot < trem ? 1 : 0);
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);

  // per-lane DMA constants.  x piece i = tid + 256 k: chunk slot i & 7 of tile pixel i >> 3 (the image IS piece order);
  // gradient piece i = tid + 256 j (< 800): chunk i & 1 of halo pixel i >> 1
  int xconst[8], xyx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = tid + 256 * k, sl = i & 7, p = i >> 3, iy = p >> 4, ix = p & 15, c = sl ^ swz(ix);
    xconst[k] = ((iy * W + ix) * a.x_pitch + a.x_coff + c * 8) * 2;
    xyx[k] = iy | (ix << 8);
  }
  int dconst[4], dyx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, c = i & 1, pp = i >> 1, iy = pp / DP, ix = pp - iy * DP;
    dconst[j] = (((iy - 2) * W + (ix - 2)) * a.dy_pitch + a.dy_coff + c * 8) * 2;
    dyx[j] = i < 2 * DP * DP ? (((iy - 2) & 0xffff) | ((ix - 2) << 16)) : (int)0x7fff7fff;      // never inside
  }
  auto dma_tile = [&](int tile, int b) {
    int pt = tile;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    const int n = pt / tilesY;
    const int y0 = tY * 16, x0 = tX * 16;
    const int xbase = ((n * H + y0) * W + x0) * a.x_pitch * 2, dbase = ((n * H + y0) * W + x0) * a.dy_pitch * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = y0 + (xyx[k] & 255) < H && x0 + (xyx[k] >> 8) < W;
      dma16_hidden(xrs, ok ? (unsigned)(xbase + xconst[k]) : 0x80000000u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * BUF + ((k * 256 + wave * 64) << 4))));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j * 256 + wave * 64 >= 2 * DP * DP) continue;        // wave-uniform: nothing of this piece row is inside the halo image
      const int gy = y0 + (int)(short)(dyx[j] & 0xffff), gx = x0 + (dyx[j] >> 16);
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      dma16_hidden(drs, ok ? (unsigned)(dbase + dconst[j]) : 0x80000000u,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + b * BUF + XB + ((j * 256 + wave * 64) << 4))));
    }
  };

  const int rb = wave & 1, half = wave >> 1;
  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  int aoff[2], boff[7][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) aoff[rd] = tr_lane_off(0, rd, rb, lane);
  {
    const int G = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      int fyA, fxA, fyB, fxB; bool hasB;
      lk5_pair(half * 7 + j, fyA, fxA, fyB, fxB, hasB);
      const int fy = (G & 1) ? fyB : fyA, fx = (G & 1) ? fxB : fxA;      // (a lone tap: its columns 16-31 repeat tap A and are not stored)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int col = 8 * (G >> 1) + 4 * rd + q;
        boff[j][rd] = ((2 - fy) * DP + (col - fx + 2)) * 32 + pq * 8;
      }
    }
  }
  const bool do_bias = a.dbp != nullptr && wave == 2;      // wave 2's first pair is (0,0) & (1,0): its group-0 lanes see the unshifted gradient
  float dbz = 0.f;

  // THREE buffers, the DMA two tiles ahead: a tile's 45 KB need 3-5k cycles to land under load, its MFMAs take 3.6k -- one tile ahead
  // every tile waited for its data (6.6k cycles per tile).  Per tile and wave 8 + 3 or 4 DMA pieces (pieces_of: wave-uniform).
  const int pieces_of = 8 + (wave == 0 ? 4 : 3);      // gradient piece rows j = 0..2 all waves, j = 3 (pieces 768..799): wave 0 only
  if (nt > 0) dma_tile(t0, 0);
  if (nt > 1) dma_tile(t0 + 1, 1);
#if SRK_LK5_STAMPS
  unsigned long long* const wstamp = (blockIdx.x == 0 && tid == 0 && a.dbp) ? reinterpret_cast<unsigned long long*>(a.dbp) + (size_t)a.nslabs * 8 : nullptr;
#define SRK_WSTAMP(i) do { if (wstamp && it < 24) wstamp[it * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SRK_WSTAMP(i) do { } while (0)
#endif
  for (int it = 0; it < nt; ++it) {
    SRK_WSTAMP(0);
    const char* const X = smem + (it % 3) * BUF;
    const char* const D = X + XB;
    // tile `it` landed (tile it + 1 may still be in flight: the younger pieces_of operations)
    if (it + 1 < nt) { if (pieces_of == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SRK_WSTAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... for every wave; buffer (it + 2) % 3 (tile it - 1's) is free
    SRK_WSTAMP(2);
    if (it + 2 < nt) dma_tile(t0 + it + 2, (it + 2) % 3);
    SRK_WSTAMP(3);
    i32x4 afn, bfn[7];
    auto fetch = [&](int y) {
      afn = tr_read2(X + y * 2048 + aoff[0], X + y * 2048 + aoff[1]);
#pragma unroll
      for (int j = 0; j < 7; ++j) bfn[j] = tr_read2(D + y * (DP * 32) + boff[j][0], D + y * (DP * 32) + boff[j][1]);
    };
    fetch(0);
#pragma unroll 2
    for (int y = 0; y < 16; ++y) {
      const i32x4 af = afn;
      i32x4 bf[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) bf[j] = bfn[j];
      if (y + 1 < 16) fetch(y + 1);
      if (do_bias) {
        const int qw[4] = {bf[0].x, bf[0].y, bf[0].z, bf[0].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f0, f1;
          unpack2<DT>((uint32_t)qw[e], f0, f1);
          dbz += f0 + f1;
        }
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = Tr::mma(af, bf[j], acc[j]);
    }
    SRK_WSTAMP(4);
  }

  if (do_bias) {
    dbz += __shfl_xor(dbz, 32, 64);                              // the two K halves of a read
    if (lane < 16) a.dbp[(size_t)slot * 16 + lane] = dbz;
  }
  {
    const int hq = lane >> 5, n = lane & 31, co = n & 15;
    float* const sl = a.dwp + (size_t)slot * (25 * 64 * 16);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      int fyA, fxA, fyB, fxB; bool hasB;
      lk5_pair(half * 7 + j, fyA, fxA, fyB, fxB, hasB);
      if (n >= 16 && !hasB) continue;
      const int tap = n < 16 ? (fyA + 2) * 5 + fxA + 2 : (fyB + 2) * 5 + fxB + 2;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = rb * 32 + 4 * hq + (e & 3) + 8 * (e >> 2);
        sl[((size_t)tap * 64 + ci) * 16 + co] = acc[j][e];
      }
    }
  }
}

bool lk5_wgrad_ok(const srk_wgrad_args& a) {
  static const bool off = [] { const char* e = srk_dbg_getenv("SRK_NO_LK5"); return e && e[0] == '1'; }();      // A/B knob
  return !off && a.KH == 5 && a.Cin == 64 && a.Cout == 16 && (a.cout_real == 0 || a.cout_real > 6) && a.x_pitch % 8 == 0 && a.dy_pitch % 8 == 0 &&
         a.x_coff % 8 == 0 && a.dy_coff % 8 == 0;
}

bool lk_all_rows(const srk_wgrad_args& a) {
  static const bool off = [] { const char* e = srk_dbg_getenv("SRK_NO_LK_ALLROWS"); const char* p = srk_dbg_getenv("SRK_NO_LK_PACKED"); return (e && e[0] == '1') || (p && p[0] == '1'); }();
  return !off && a.cout_real > 0 && a.cout_real <= LK_CS && a.cout_real * a.KW <= 32 && a.KH >= 5;
}

int lk_wgrad_slabs_for(const srk_wgrad_args& a) {
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  if (lk_all_rows(a)) {                        // one workgroup per slab, all kernel rows inside
    const long long nt8 = (long long)a.N * ((a.H + LK_TR - 1) / LK_TR) * ((a.W + 15) / 16);
    return (int)(nt8 < cus ? nt8 : cus);
  }
  const long long ntiles = (long long)a.N * ((a.H + 15) / 16) * ((a.W + 15) / 16);
  if (lk5_wgrad_ok(a)) return (int)(ntiles < cus ? ntiles : cus);      // one workgroup per slab, all 25 taps inside
  long long s = cus / a.KH;
  if (s < 1) s = 1;
  if (s > ntiles) s = ntiles;
  return (int)s;
}

}  // namespace

// 16-bit, K in {5, 7, 9}, plain NHWC in / out, (Cin <= 64, <= 32 output rows) or (16 input channels, 64 rows): srk_conv2d
bool srk_conv_lk_ok(const srk_conv_args& a) {
  if (a.dtype == SRK_F32 || a.KH != a.KW || a.KH < 5 || a.KH > 9 || !(a.KH & 1)) return false;
  if (a.x_ps > 1 || a.mask) return false;
  const bool fwd = a.Cin == 64 && a.CoutP == 32;
  const bool bwd = a.Cin == 16 && a.CoutP == 64;
  if (!fwd && !bwd) return false;
  if (a.out_mode == SRK_OUT_PLANAR) {
    // the image store of the collapsed HR stage: fp32 NCHW behind PixelShuffle(2), <= 16 channels (4 sub-pixels x <= 4 colours)
    if (!fwd || a.ps_r != 2 || a.res || a.relu || a.Cout % 4 || a.Cout > 16 || a.x_pitch % 8 || a.x_coff % 8) return false;
    return (long long)a.N * a.H * a.W * a.x_pitch * 2 < 0x7fff0000LL;
  }
  if (a.out_mode != SRK_OUT_NHWC || (a.post_add && !SRK_LK5_STAMPS)) return false;
  if (a.x_pitch % 8 || a.x_coff % 8 || a.out_pitch % 8 || a.out_coff % 8 || a.Cout % 8 || (a.res && (a.res_pitch % 8 || a.res_coff % 8))) return false;
  const long long px = (long long)a.N * a.H * a.W;
  long long mx = px * a.x_pitch;
  if (px * a.out_pitch > mx) mx = px * a.out_pitch;
  return mx * 2 < 0x7fff0000LL;
}

template <int DT, int CPP, int NRB, int K> static int lk_launch_k(const srk_conv_args& a, hipStream_t st) {
  constexpr int XT = 16 + K - 1, XTP = (XT + 1) & ~1;
  constexpr int xs = (XT * XTP * CPP * 16 + 1023) & ~1023, slab = K * CPP * 32 * NRB * 16;
  constexpr int lds = xs + 2 * slab;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk_conv_kernel<DT, CPP, NRB, K>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_conv2d: cannot reserve LDS for the large-kernel conv"); return (int)attr; }
  SRK_CHECK_ARG(lds <= 160 * 1024, "srk_conv2d: %dx%d kernel needs %d bytes of LDS", K, K, lds);
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  const long long nb = (long long)a.N * tilesX * tilesY;
  SRK_CHECK_ARG(nb <= 0x7fffffffLL, "srk_conv2d: %lld workgroups", nb);
  hipLaunchKernelGGL((lk_conv_kernel<DT, CPP, NRB, K>), dim3((unsigned)nb), dim3(256), lds, st, a, tilesX, tilesY,
                     (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2), (unsigned)((long long)K * slab));
  SRK_LAUNCH_CHECK();
  return 0;
}
template <int DT, int CPP, int NRB> static int lk_launch(const srk_conv_args& a, hipStream_t st) {
  switch (a.KH) {
    case 5: return lk_launch_k<DT, CPP, NRB, 5>(a, st);
    case 7: return lk_launch_k<DT, CPP, NRB, 7>(a, st);
    default: return lk_launch_k<DT, CPP, NRB, 9>(a, st);
  }
}

template <int DT, int K> static int lk_rows_launch_k(const srk_conv_args& a, hipStream_t st) {
  constexpr int XWP = (16 + K - 1 + 1) & ~1, XR = 8 + K - 1;
  constexpr int lds = ((XR * XWP * 128 + 1023) & ~1023) + 4 * 4096 + 4 * 512;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk_conv_rows_kernel<DT, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) { srk_set_error("srk_conv2d: cannot reserve LDS for the large-kernel conv"); return (int)attr; }
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 7) / 8;
  const long long nb = (long long)a.N * tilesX * tilesY;
  SRK_CHECK_ARG(nb <= 0x7fffffffLL, "srk_conv2d: %lld workgroups", nb);
  hipLaunchKernelGGL((lk_conv_rows_kernel<DT, K>), dim3((unsigned)nb), dim3(256), lds, st, a, tilesX, tilesY,
                     (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2), (unsigned)((long long)K * K * 64 * a.CoutP * 2));
  SRK_LAUNCH_CHECK();
  return 0;
}
template <int DT> static int lk_rows_launch(const srk_conv_args& a, hipStream_t st) {
  switch (a.KH) {
    case 5: return lk_rows_launch_k<DT, 5>(a, st);
    case 7: return lk_rows_launch_k<DT, 7>(a, st);
    default: return lk_rows_launch_k<DT, 9>(a, st);
  }
}

template <int DT, int NW> static int lk5_rows_fwd_launch_w(const srk_conv_args& a, hipStream_t st) {
  constexpr int PRO = NW + 4 + (NW == 4 ? NW : 0), lds = (PRO + NW) * 32 * 128 + NW * 32 * 68 * 4;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk5_rows_fwd_kernel<DT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) { srk_set_error("srk_conv2d: cannot reserve LDS for the 5x5 image conv"); return (int)attr; }
  const int nb = (a.W + 27) / 28;
  // row segments per band: as many units as there are CUs at small batches, but not more (a second round of units doubles the launch;
  // a segment re-reads four halo rows), at least 8 rows each
  int segs = (int)(cus / ((long long)a.N * nb));
  const int max_segs = (a.H + 7) / 8;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  const int seg_rows = (((a.H + segs - 1) / segs) + NW - 1) / NW * NW;
  segs = (a.H + seg_rows - 1) / seg_rows;
  const long long units = (long long)a.N * nb * segs;
  SRK_CHECK_ARG(units <= 0x7fffffffLL, "srk_conv2d: %lld units", units);
  const int grid = (int)(units < cus ? units : cus);
  hipLaunchKernelGGL((lk5_rows_fwd_kernel<DT, NW>), dim3(grid), dim3(NW * 64), lds, st, a, nb, segs, seg_rows, (int)units,
                     (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2));
  SRK_LAUNCH_CHECK();
  return 0;
}
// (NW = 8 -- two waves per SIMD -- needs 160 weight + 64 accumulator + 24 fragment registers of a wave's 256: hipcc spills 83 of them, and a
// scratch access is a vector-memory operation the hand-counted waits do not include.  One wave per SIMD it is.)
template <int DT> static int lk5_rows_fwd_launch(const srk_conv_args& a, hipStream_t st) { return lk5_rows_fwd_launch_w<DT, 4>(a, st); }

// the data gradient of the collapsed HR stage: 5x5, 16 stored channels -> 64, plain NHWC store
static bool lk5_dgrad_ok(const srk_conv_args& a) {
  return a.KH == 5 && a.Cin == 16 && a.CoutP == 64 && a.Cout == 64 && a.out_mode == SRK_OUT_NHWC && !a.bias && !a.res && !a.relu && !a.mask &&
         (SRK_LK5_STAMPS || !a.post_add) && a.scale == 1.f && (long long)a.N * a.H * a.W * a.out_pitch * 2 < 0x7fff0000LL;
}
template <int DT> static int lk5_dgrad_launch(const srk_conv_args& a, hipStream_t st) {
  constexpr int lds = 16 * 36 * 32;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  const int nb = (a.W + 31) / 32;
  int segs = (int)(cus / ((long long)a.N * nb));      // (as in lk5_rows_fwd_launch)
  const int max_segs = (a.H + 7) / 8;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  const int seg_rows = (((a.H + segs - 1) / segs) + 3) & ~3;
  segs = (a.H + seg_rows - 1) / seg_rows;
  const long long units = (long long)a.N * nb * segs;
  SRK_CHECK_ARG(units <= 0x7fffffffLL, "srk_conv2d: %lld units", units);
  const int grid = (int)(units < cus ? units : cus);
  hipLaunchKernelGGL((lk5_dgrad_kernel<DT>), dim3(grid), dim3(256), lds, st, a, nb, segs, seg_rows, (int)units,
                     (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2));
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT> static int lk5_fwd_launch(const srk_conv_args& a, hipStream_t st) {
  constexpr int lds = 25 * 8 * 16 * 16 + 2 * 20 * 20 * 128;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk5_fwd_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_conv2d: cannot reserve LDS for the 5x5 image conv"); return (int)attr; }
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  const long long ntiles = (long long)a.N * tilesX * tilesY;
  SRK_CHECK_ARG(ntiles <= 0x7fffffffLL, "srk_conv2d: %lld tiles", ntiles);
  const int slots = (int)(ntiles < cus ? ntiles : cus);
  hipLaunchKernelGGL((lk5_fwd_kernel<DT>), dim3(slots), dim3(256), lds, st, a, tilesX, tilesY, (int)(ntiles / slots), (int)(ntiles % slots),
                     (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2));
  SRK_LAUNCH_CHECK();
  return 0;
}

int srk_conv_lk_launch(const srk_conv_args& a, hipStream_t st) {
  static const bool no_lk5 = [] { const char* e = srk_dbg_getenv("SRK_NO_LK5"); return e && e[0] == '1'; }();      // A/B knob
  static const bool no_lk5_rows = [] { const char* e = srk_dbg_getenv("SRK_NO_LK5_ROWS"); return e && e[0] == '1'; }();      // A/B knob
  if (a.out_mode == SRK_OUT_PLANAR && a.KH == 5 && !no_lk5 && !no_lk5_rows && a.Cout <= 12)
    return a.dtype == SRK_BF16 ? lk5_rows_fwd_launch<SRK_BF16>(a, st) : lk5_rows_fwd_launch<SRK_F16>(a, st);
  if (a.out_mode == SRK_OUT_PLANAR && a.KH == 5 && !no_lk5) return a.dtype == SRK_BF16 ? lk5_fwd_launch<SRK_BF16>(a, st) : lk5_fwd_launch<SRK_F16>(a, st);
  static const bool no_rows = [] { const char* e = srk_dbg_getenv("SRK_NO_LK_ROWS"); return e && e[0] == '1'; }();      // A/B knob
  if (!no_rows && a.out_mode == SRK_OUT_NHWC && a.Cin == 64 && a.cout_real > 0 && a.cout_real <= 4 && a.cout_real * a.KW <= 32 && a.Cout == 16 && !a.relu && !a.res && a.scale == 1.f)
    return a.dtype == SRK_BF16 ? lk_rows_launch<SRK_BF16>(a, st) : lk_rows_launch<SRK_F16>(a, st);
  static const bool no_lk5_dgrad = [] { const char* e = srk_dbg_getenv("SRK_NO_LK5_DGRAD"); return e && e[0] == '1'; }();      // A/B knob
  if (!no_lk5 && !no_lk5_dgrad && lk5_dgrad_ok(a)) return a.dtype == SRK_BF16 ? lk5_dgrad_launch<SRK_BF16>(a, st) : lk5_dgrad_launch<SRK_F16>(a, st);
  if (a.Cin == 16) return a.dtype == SRK_BF16 ? lk_launch<SRK_BF16, 2, 2>(a, st) : lk_launch<SRK_F16, 2, 2>(a, st);
  return a.dtype == SRK_BF16 ? lk_launch<SRK_BF16, 8, 1>(a, st) : lk_launch<SRK_F16, 8, 1>(a, st);
}

bool srk_wgrad_lk_ok(const srk_wgrad_args& a) {
  if (a.dtype == SRK_F32 || a.KH != a.KW || a.KH < 5 || a.KH > 9 || !(a.KH & 1)) return false;
  if (a.x_ps > 1 || a.dy_ps > 1 || a.Cin > 64 || a.Cin % 16 || a.Cout > 32 || a.Cout % 8) return false;
  const long long px = (long long)a.N * a.H * a.W;
  return px * a.x_pitch * 2 < 0x7fff0000LL && px * a.dy_pitch * 2 < 0x7fff0000LL;
}
int srk_wgrad_lk_slabs(const srk_wgrad_args& a) { return lk_wgrad_slabs_for(a); }
int srk_wgrad_lk_slab_cout(const srk_wgrad_args& a) { return lk_all_rows(a) ? LK_CS : a.Cout; }

template <int DT, int K> static int lk_wgrad_launch_k(const srk_wgrad_args& a, hipStream_t st) {
  constexpr int XWP = (16 + K - 1 + 1) & ~1;
  constexpr int lds = 2 * (16 * XWP * 128 + 16 * 16 * 32 + 1024);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk_wgrad_kernel<DT, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) { srk_set_error("srk_conv2d_wgrad: cannot reserve LDS"); return (int)attr; }
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  const long long ntiles = (long long)a.N * tilesX * tilesY;
  const int slabs = a.nslabs;
  const unsigned xb = (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2), db = (unsigned)((long long)a.N * a.H * a.W * a.dy_pitch * 2);
  if (lk_all_rows(a)) {
    constexpr int XR = LK_TR + K - 1;
    constexpr int alds = 2 * (((XR * XWP * 128 + 1023) & ~1023) + LK_TR * 16 * 32) + 8 * LK_TR * (16 + 2 * (K - 1) + 16) * 2;
    static_assert(alds <= 160 * 1024, "LDS");
    static const hipError_t aattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk_wgrad_allrows_kernel<DT, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (aattr != hipSuccess) { srk_set_error("srk_conv2d_wgrad: cannot reserve LDS"); return (int)aattr; }
    const int tY8 = (a.H + LK_TR - 1) / LK_TR;
    const long long nt8 = (long long)a.N * tY8 * tilesX;
    hipLaunchKernelGGL((lk_wgrad_allrows_kernel<DT, K>), dim3(slabs), dim3(256), alds, st, a, tilesX, tY8, (int)nt8, (int)(nt8 / slabs), (int)(nt8 % slabs), xb, db);
    SRK_LAUNCH_CHECK();
    return 0;
  }
  if constexpr (K == 5) {
    if (lk5_wgrad_ok(a)) {
      constexpr int l5 = 3 * (16 * 16 * 128 + 13 * 1024);
      static const hipError_t a5 = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk5_wgrad_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (a5 != hipSuccess) { srk_set_error("srk_conv2d_wgrad: cannot reserve LDS"); return (int)a5; }
      hipLaunchKernelGGL((lk5_wgrad_kernel<DT>), dim3(slabs), dim3(256), l5, st, a, tilesX, tilesY, (int)(ntiles / slabs), (int)(ntiles % slabs), xb, db);
      SRK_LAUNCH_CHECK();
      return 0;
    }
  }
  static const bool no_packed = [] { const char* e = srk_dbg_getenv("SRK_NO_LK_PACKED"); return e && e[0] == '1'; }();      // A/B knob
  if (a.cout_real > 0 && a.cout_real <= 6 && a.cout_real * K <= 32 && !no_packed) {
    // few real output channels (SRResNet's tail: 3): MFMA columns = (kw, co) pairs
    constexpr int plds = 2 * (16 * XWP * 128 + 16 * 16 * 32) + 8 * 16 * (16 + 2 * (K - 1) + 16) * 2 + 2 * 16 * 64 * 4;
    static_assert(plds <= 160 * 1024, "LDS");
    static const hipError_t pattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&lk_wgrad_packed_kernel<DT, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (pattr != hipSuccess) { srk_set_error("srk_conv2d_wgrad: cannot reserve LDS"); return (int)pattr; }
    hipLaunchKernelGGL((lk_wgrad_packed_kernel<DT, K>), dim3(slabs, K), dim3(256), plds, st, a, tilesX, tilesY, (int)ntiles, (int)(ntiles / slabs), (int)(ntiles % slabs), xb, db);
    SRK_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((lk_wgrad_kernel<DT, K>), dim3(slabs, K), dim3(256), lds, st, a, tilesX, tilesY, (int)ntiles, (int)(ntiles / slabs), (int)(ntiles % slabs), xb, db);
  SRK_LAUNCH_CHECK();
  return 0;
}
template <int DT> static int lk_wgrad_launch_dt(const srk_wgrad_args& a, hipStream_t st) {
  switch (a.KH) {
    case 5: return lk_wgrad_launch_k<DT, 5>(a, st);
    case 7: return lk_wgrad_launch_k<DT, 7>(a, st);
    default: return lk_wgrad_launch_k<DT, 9>(a, st);
  }
}
int srk_wgrad_lk_launch(const srk_wgrad_args& a, hipStream_t st) {
  return a.dtype == SRK_BF16 ? lk_wgrad_launch_dt<SRK_BF16>(a, st) : lk_wgrad_launch_dt<SRK_F16>(a, st);
}
