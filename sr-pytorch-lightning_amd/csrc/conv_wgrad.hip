// Weight / bias gradient of the 'same' convolution on gfx950 MFMA.
//
//   dWp[tap][ci][co] += sum_{n,y,x} X[n][y+kh-p][x+kw-p][ci] * dY[n][y][x][co]
//
// GEMM view: M = ci (A = X^T), N = co (B = dY), K = pixels.  Both operands live in LDS as
// [pixel][channel] images (the same swizzled halo image the forward kernel uses), i.e. K-major, so the
// 16-bit path fetches MFMA fragments with the gfx950 transposing LDS read `ds_read_b64_tr_b16`
// (4 pixels x 16 channels per 16-lane group); the fp32 path reads one float per lane (32x32x2 MFMA).
// One K-step = one 16-pixel tile row; the K*K taps are K*K accumulator tiles fed from shifted rows /
// columns of the halo image.  A workgroup owns one (ci-block, co-block) pair of 128-byte channel
// blocks, walks a strided set of 16x16 pixel tiles accumulating in registers, and adds its partial
// result into the fp32 scratch with row-contiguous float atomics.
//
// Replaces: autograd's conv2d weight/bias gradient for models/common.py:7-30 convs.
#include <stdlib.h>
#include "srk_common.h"

namespace {

template <int DT, int KS> struct WgCfg {
  typedef DTraits<DT> Tr;
  static constexpr int CH = Tr::CH;
  static constexpr int ESZ = 16 / CH;                 // bytes per element
  static constexpr int CBLK = 128 / ESZ;              // channels per 128-byte block (64 / 32)
  static constexpr int RB = CBLK / 32;                // 32-row MFMA blocks per channel block (2 / 1)
  static constexpr int PAIRS = RB * RB;               // (ci32, co32) pairs per workgroup (4 / 1)
  static constexpr int KGROUPS = 4 / PAIRS;           // waves that split the K (tile-row) loop (1 / 4)
  static constexpr int NT = 256;
  static constexpr int NTAPS = KS * KS;
  static constexpr int PAD = KS / 2;
  static constexpr int TIN = 16 + KS - 1;
  static constexpr int PITCH = (TIN + 1) & ~1;
  static constexpr int XS_BYTES = TIN * PITCH * 128;
  static constexpr int DYS_BYTES = 256 * 128;
  static constexpr int LDS_BYTES = XS_BYTES + DYS_BYTES;
  static constexpr int XPIECES = TIN * TIN * 8;
  static constexpr int DYPIECES = 256 * 8;
};

// element offset of chunk k0 (first channel) of conv-space pixel (n,gy,gx); r>1: tensor stored pixel-shuffled
SRK_DEV size_t chunk_off(int n, int gy, int gx, int H, int W, int r, int Cs, int k0, int pitch, int coff) {
  if (r == 1) return ((size_t)(n * H + gy) * W + gx) * pitch + coff + k0;
  const int ij = k0 / Cs, c0 = k0 - ij * Cs;
  const int si = ij / r, sj = ij - si * r;
  return ((size_t)(n * H * r + gy * r + si) * (W * r) + gx * r + sj) * pitch + coff + c0;
}

SRK_DEV float lds_f32(const char* img, int pitch, int row, int col, int c) {
  return *reinterpret_cast<const float*>(img + ((row * pitch + col) << 7) + (((c >> 2) ^ swz(col)) << 4) + ((c & 3) << 2));
}

template <int DT, int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int ntiles) {
  typedef WgCfg<DT, KS> C;
  typedef typename C::Tr Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = C::CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ds = smem + C::XS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % C::PAIRS, kgroup = wave / C::PAIRS;
  const int rb = pair / C::RB, cbk = pair % C::RB;   // 32-blocks of ci (rows) and co (cols) inside the channel block
  const int cib = blockIdx.y, cob = blockIdx.z;
  const int H = a.H, W = a.W;
  const int nch_x = a.Cin / CH, nch_d = a.Cout / CH;
  const int rx = a.x_ps > 1 ? a.x_ps : 1, rd = a.dy_ps > 1 ? a.dy_ps : 1;
  const int Csx = a.Cin / (rx * rx), Csd = a.Cout / (rd * rd);
  const elem* const xg = reinterpret_cast<const elem*>(a.x);
  const elem* const dg = reinterpret_cast<const elem*>(a.dy);

  f32x16 acc[C::NTAPS];
#pragma unroll
  for (int t = 0; t < C::NTAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float bsum = 0.f;
  constexpr int BPARTS = C::NT / C::CBLK;                 // bias-grad: pixel partitions per channel
  const int b_c = tid % C::CBLK, b_part = tid / C::CBLK;
  int xoff[KS][2], doff[2];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xoff[kw][0] = tr_lane_off(kw, 0, rb, lane);
    xoff[kw][1] = tr_lane_off(kw, 1, rb, lane);
  }
  doff[0] = tr_lane_off(0, 0, cbk, lane);
  doff[1] = tr_lane_off(0, 1, cbk, lane);

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int pt = tile;
    const int tX = pt % tilesX;
    pt /= tilesX;
    const int tY = pt % tilesY;
    const int n = pt / tilesY;
    const int y0 = tY * 16, x0 = tX * 16;

    __syncthreads();   // previous tile's fragment reads are done
    // ---- stage X halo tile (ci block) ------------------------------------------------------------
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
          const int c = cib * 8 + (s ^ swz(ix));
          const int gy = y0 + iy - C::PAD, gx = x0 + ix - C::PAD;
          if (c < nch_x && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v[u] = gload16(xg + chunk_off(n, gy, gx, H, W, rx, Csx, c * CH, a.x_pitch, a.x_coff));
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
    // ---- stage dY tile (co block), zero outside the image ------------------------------------------
#pragma unroll 1
    for (int base = tid; base < C::DYPIECES; base += 4 * C::NT) {
      i32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * C::NT;
        const int s = i & 7, p = i >> 3;
        const int iy = p >> 4, ix = p & 15;
        const int c = cob * 8 + (s ^ swz(ix));
        const int gy = y0 + iy, gx = x0 + ix;
        v[u] = i32x4{0, 0, 0, 0};
        if (c < nch_d && gy < H && gx < W)
          v[u] = gload16(dg + chunk_off(n, gy, gx, H, W, rd, Csd, c * CH, a.dy_pitch, a.dy_coff));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) lds_write16(Ds + ((base + u * C::NT) << 4), v[u]);
    }
    __syncthreads();

    // ---- bias gradient from the dY image ------------------------------------------------------------
    if (a.dbp && cib == 0) {
#pragma unroll 4
      for (int p = b_part; p < 256; p += BPARTS) {
        const char* ad = Ds + (p << 7) + ((((b_c / CH)) ^ swz(p & 15)) << 4) + (b_c % CH) * C::ESZ;
        bsum += Tr::to_f32(*reinterpret_cast<const elem*>(ad));
      }
    }

    // ---- K loop over tile rows ---------------------------------------------------------------------------
    if constexpr (Tr::IS16) {
      // fully unrolled: row offsets become ds_read immediates; the halo rows are rotated through
      // registers so each K-step fetches only ONE new image row (KS fragments) plus the dY fragment
      i32x4 xf[KS][KS];
#pragma unroll
      for (int rr = 0; rr < KS - 1; ++rr)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
          xf[rr % KS][kw] = tr_read2(Xs + xoff[kw][0] + rr * (C::PITCH * 128), Xs + xoff[kw][1] + rr * (C::PITCH * 128));
#pragma unroll
      for (int y = 0; y < 16; ++y) {
        const int nr = y + KS - 1;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
          xf[nr % KS][kw] = tr_read2(Xs + xoff[kw][0] + nr * (C::PITCH * 128), Xs + xoff[kw][1] + nr * (C::PITCH * 128));
        const i32x4 bf = tr_read2(Ds + doff[0] + y * (16 * 128), Ds + doff[1] + y * (16 * 128));
#pragma unroll
        for (int t = 0; t < C::NTAPS; ++t) {
          const int kh = t / KS, kw = t - kh * KS;
          acc[t] = Tr::mma(xf[(y + kh) % KS][kw], bf, acc[t]);
        }
      }
    } else {
      const int i = lane & 31, hk = lane >> 5;
#pragma unroll 1
      for (int y = kgroup; y < 16; y += C::KGROUPS) {
#pragma unroll 2
        for (int m = 0; m < 8; ++m) {
          const int col = 2 * m + hk;
          const float b = lds_f32(Ds, 16, y, col, i);
#pragma unroll
          for (int t = 0; t < C::NTAPS; ++t) {
            const int kh = t / KS, kw = t - kh * KS;
            const float av = lds_f32(Xs, C::PITCH, y + kh, col + kw, i);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- add the partial result into the fp32 scratch (row-contiguous float atomics) -------------------
  const int co = cob * C::CBLK + cbk * 32 + (lane & 31);
  const int hq = lane >> 5;
  if (co < a.Cout) {
#pragma unroll
    for (int t = 0; t < C::NTAPS; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = cib * C::CBLK + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hq;
        if (ci < a.Cin) atomicAdd(a.dwp + ((size_t)t * a.Cin + ci) * a.Cout + co, acc[t][e]);
      }
    }
  }
  if (a.dbp && cib == 0) {
    const int c = cob * C::CBLK + b_c;
    if (c < a.Cout) atomicAdd(a.dbp + c, bsum);
  }
}


// =================================================================================================
// Slab-mode kernel (16-bit dtypes, 3x3): same MFMA core, but
//  * halo / dY tiles arrive by LDS-DMA (buffer_load ... lds; out-of-image pieces use an out-of-range offset and
//    are zero-filled by the hardware), with all per-lane address constants computed once per kernel;
//  * the (X, dY) image pair is double-buffered: the DMA of tile t+1 is in flight during the 144 MFMAs per wave
//    of tile t; there is no per-tile epilogue, so one wave per SIMD with the whole 512-register file
//    (144 accumulators + 9 rotating X fragments) keeps the matrix pipe busy; one barrier per tile;
//  * every workgroup STORES its partial sum to its own slab: no atomics, nothing to zero, bitwise
//    reproducible; srk_wgrad_finalize sums the slabs.
// =================================================================================================
typedef __attribute__((address_space(3))) void lds_void_t;

// The 16 x 16-tile form of rounds 1-5 (two buffers, one tile ahead): kept for SMALL problems, where a workgroup walks few tiles and the three-tile fill of the ring
// below costs more than it returns (same box, patches/s 16-row / 8-row: RCAN batch 16 2,064 / 2,058, EDSR-large batch 16 1,391 / 1,371, EDSR-baseline batch 64 32.4k / 32.2k;
// batch 256 sustained 48.1k / 49.7k: profiles/r6_experiments.txt 10).  ws_tile_height() picks per launch.
template <int DT>
SRK_DEV void wgrad_ws_body16(const srk_wgrad_args& a, const int slot, const int cib, const int cob, const int tilesX, const int tilesY,
                           const unsigned x_bytes, const unsigned dy_bytes, const int tq, const int trem, char* const smem) {
  typedef WgCfg<DT, 3> C;
  typedef typename C::Tr Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = C::CH, KS = 3, GT = 256, ESZ = C::ESZ;
  constexpr int NPK = (C::XPIECES + GT - 1) / GT;      // 11 halo pieces per lane
  constexpr int NDK = C::DYPIECES / GT;                // 8 dY pieces per lane
  constexpr int BUF_BYTES = C::XS_BYTES + C::DYS_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, cbk = wave & 1;
  const int H = a.H, W = a.W;
  const int nch_x = a.Cin / CH, nch_d = a.Cout / CH;

  const int t0 = slot * tq + min(slot, trem);
  const int nt = tq + (slot < trem ? 1 : 0);

  // ---- per-lane DMA constants --------------------------------------------------------------------------
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);
  int pconst[NPK], pyx[NPK];
#pragma unroll
  for (int k = 0; k < NPK; ++k) {
    const int i = tid + k * GT;
    const int sl = i & 7, p = i >> 3;
    const int iy = p / C::TIN, ix = p - iy * C::TIN;
    const int c = cib * 8 + (sl ^ swz(ix));
    pconst[k] = (((iy - 1) * W + (ix - 1)) * a.x_pitch + a.x_coff + c * CH) * ESZ;
    pyx[k] = (i < C::XPIECES && c < nch_x) ? (((iy - 1) & 0xffff) | ((ix - 1) << 16)) : (int)0x7fff7fff;
  }
  // dY piece i = tid + 256k: slot, column and channel chunk do not depend on k, the row advances by 2 per k
  const int rd = a.dy_ps > 1 ? a.dy_ps : 1;
  const int d_ix = (tid >> 3) & 15, d_iy0 = tid >> 7;
  const int d_c = cob * 8 + ((tid & 7) ^ swz(d_ix));
  const bool d_cok = d_c < nch_d;
  int dconst, dkstride;
  {
    const int Csd = a.Cout / (rd * rd);
    const int k0 = d_c * CH;
    const int ij = k0 / Csd, c0 = k0 - ij * Csd;
    const int si = ij / rd, sj = ij - si * rd;
    dconst = (((d_iy0 * rd + si) * (W * rd) + d_ix * rd + sj) * a.dy_pitch + a.dy_coff + c0) * ESZ;
    dkstride = 2 * rd * (W * rd) * a.dy_pitch * ESZ;
  }
  // one tile's DMA = 11 halo pieces + 8 dY pieces per lane.  The pieces are issued one or two per K-step from inside
  // the MFMA loop of the previous tile (an LDS-DMA issue costs 60-100 cycles of the wave's time in front of the loop,
  // with the matrix pipe idle: one wave per SIMD; between MFMAs most of that hides behind the running MFMA)
  struct TileAddr { int y0, x0, xbase, dbase; bool colok; unsigned Xb; };   // Xb: LDS byte address of the buffer
  auto tile_addr = [&](int pt, char* Xb) {
    TileAddr t;
    const int tX = pt % tilesX;
    const int q = pt / tilesX;
    const int tY = q % tilesY, n = q / tilesY;
    t.y0 = tY * 16; t.x0 = tX * 16;
    t.xbase = (((n * H + t.y0) * W + t.x0) * a.x_pitch) * ESZ;
    t.dbase = (((n * H * rd + t.y0 * rd) * (W * rd) + t.x0 * rd) * a.dy_pitch) * ESZ;
    t.colok = d_cok && (t.x0 + d_ix < W);
    t.Xb = lds0 + (unsigned)(Xb - smem);
    return t;
  };
  auto dma_x_piece = [&](const TileAddr& t, int k) {
    const int gy = t.y0 + (int)(short)(pyx[k] & 0xffff), gx = t.x0 + (pyx[k] >> 16);
    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const unsigned voff = ok ? (unsigned)(t.xbase + pconst[k]) : 0x80000000u;
    if (k < NPK - 1 || tid + k * GT < C::XPIECES)
      dma16_hidden(xrs, voff, __builtin_amdgcn_readfirstlane(t.Xb + ((k * GT + wave * 64) << 4)));
  };
  auto dma_d_piece = [&](const TileAddr& t, int k) {
    const bool ok = t.colok && (t.y0 + d_iy0 + 2 * k < H);
    const unsigned voff = ok ? (unsigned)(t.dbase + dconst + k * dkstride) : 0x80000000u;
    dma16_hidden(drs, voff, __builtin_amdgcn_readfirstlane(t.Xb + C::XS_BYTES + ((k * GT + wave * 64) << 4)));
  };
  auto dma_tile = [&](int pt, char* Xb) {
    const TileAddr t = tile_addr(pt, Xb);
#pragma unroll
    for (int k = 0; k < NPK; ++k) dma_x_piece(t, k);
#pragma unroll
    for (int k = 0; k < NDK; ++k) dma_d_piece(t, k);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float bsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.dbp != nullptr && cib == 0;
  int xoff[KS][2], doff[2];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xoff[kw][0] = tr_lane_off(kw, 0, rb, lane);
    xoff[kw][1] = tr_lane_off(kw, 1, rb, lane);
  }
  doff[0] = tr_lane_off(0, 0, cbk, lane);
  doff[1] = tr_lane_off(0, 1, cbk, lane);

  dma_tile(t0, smem);
#pragma unroll 1
  for (int it = 0; it < nt; ++it) {
    char* const Xs = smem + (it & 1) * BUF_BYTES;
    char* const Ds = Xs + C::XS_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile `it` landed
    __builtin_amdgcn_s_barrier();                         // ... for every wave; the other buffer is free
    const bool more = it + 1 < nt;
    const TileAddr nxt = tile_addr(more ? t0 + it + 1 : t0 + it, smem + ((it + 1) & 1) * BUF_BYTES);

    if (do_bias) {
      // bias gradient: lane owns LDS slot (tid&7) of pixels (tid>>3) + 32k; the slot's channel chunk is the same
      // for all 8 of them (the swizzle depends on the column, and 32 pixels = 2 full rows)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const i32x4 raw = lds_read16(Ds + ((tid + 256 * k) << 4));
        const uint32_t w4[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bsum8[2 * i] += Tr::to_f32((uint16_t)(w4[i] & 0xffff));
          bsum8[2 * i + 1] += Tr::to_f32((uint16_t)(w4[i] >> 16));
        }
      }
    }
    // K loop over the 16 tile rows; the fragments of step y+1 (one new halo row, one dY row) are fetched while
    // the 9 MFMAs of step y run (4 rotating halo-row slots), sched_barrier keeps the groups apart
    i32x4 xf[4][KS];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
        xf[rr][kw] = tr_read2(Xs + xoff[kw][0] + rr * (C::PITCH * 128), Xs + xoff[kw][1] + rr * (C::PITCH * 128));
    i32x4 bf = tr_read2(Ds + doff[0], Ds + doff[1]);
#pragma unroll
    for (int y = 0; y < 16; ++y) {
      i32x4 bfn = bf;
      if (y + 1 < 16) {
        const int nr = y + 3;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
          xf[nr & 3][kw] = tr_read2(Xs + xoff[kw][0] + nr * (C::PITCH * 128), Xs + xoff[kw][1] + nr * (C::PITCH * 128));
        bfn = tr_read2(Ds + doff[0] + (y + 1) * (16 * 128), Ds + doff[1] + (y + 1) * (16 * 128));
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int kh = t / KS, kw = t - kh * KS;
        acc[t] = Tr::mma(xf[(y + kh) & 3][kw], bf, acc[t]);
        // next tile's DMA, front-loaded (3 pieces per step in steps 0-5, the rest by step 7): a piece takes 3-5k cycles
        // to land under load and the whole tile must be there when this loop (5.3k cycles) ends
        if (t == 1 && 2 * y < NPK && more) dma_x_piece(nxt, 2 * y);
        if (t == 4 && 2 * y + 1 < NPK && more) dma_x_piece(nxt, 2 * y + 1);
        if (t == 7 && y < NDK && more) dma_d_piece(nxt, y);
      }
      bf = bfn;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- store the workgroup's slab (row-contiguous 128-byte segments per half wave) ------------------------
  // buffer stores: one 32-bit offset per lane, the (tap, row) part is uniform; rows / columns beyond the real
  // channel counts get an out-of-range offset (dropped by the hardware) instead of a branch
  {
    const int co = cob * 64 + cbk * 32 + (lane & 31);
    const int hq = lane >> 5;
    const size_t slab_elems = (size_t)9 * a.Cin * a.Cout;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(a.dwp + (size_t)slot * slab_elems, 0, (unsigned)(slab_elems * 4), 0x00020000);
    const int ci0 = cib * 64 + rb * 32 + 4 * hq;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + (e & 3) + 8 * (e >> 2);
        const unsigned voff = (ci < a.Cin && co < a.Cout) ? (unsigned)(((t * a.Cin + ci) * a.Cout + co) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][e]), srs, voff, 0, 0);
      }
    }
  }
  if (do_bias) {
    // fixed-order reduction of the per-lane chunk sums: channel c lives in chunk c>>3, held by the 32 lanes
    // (hi, q16) with slot (c>>3) ^ swz(q16)
    float* const bred = reinterpret_cast<float*>(smem);      // [256][8]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) bred[tid * 8 + e] = bsum8[e];
    __syncthreads();
    if (tid < 64) {
      const int ch = tid >> 3, e = tid & 7;
      float t = 0.f;
      for (int p0 = 0; p0 < 32; ++p0) t += bred[((p0 << 3) + (ch ^ swz(p0 & 15))) * 8 + e];
      const int c = cob * 64 + tid;
      if (c < a.Cout) a.dbp[(size_t)slot * a.Cout + c] = t;
    }
  }
}


// Round 6: tiles of WS_TH = 8 rows x 16 columns in a ring of FOUR buffers, THREE tiles in flight.  With 16 x 16 tiles the LDS held two (X, dY) pairs: the next
// tile's 73.5 KB were requested during the first half of a tile and the stream then paused until the next tile began -- on average half a tile's bytes in
// flight per CU, and the kernel ran at the byte rate that allows (4.7 TB/s chip-wide, profiles/r6_experiments.txt 9; MFMA issue was not the limit: two waves
// per SIMD changed nothing).  Half-height tiles are 40 KB each (10 x 18 halo pixels + 8 x 16 gradient pixels, the halo image padded to whole 256-lane
// pieces so that EVERY lane issues the same 10 pieces per tile: out-of-image and past-the-end pieces use out-of-range offsets), four of them fill the 160 KB,
// and the wait in front of a tile is the constant vmcnt(2 x 10): the two younger tiles stay in flight.  Same MFMAs per pixel; the halo rows cost 10 / 8
// instead of 18 / 16 of X.  The slabs differ from the 16-row form in summation order only (a tile row is still one K step).
constexpr int WS_TH = 8;                                 // tile height of the slab-mode 3x3 weight gradient (host: tilesY = ceil(H / WS_TH))

template <int DT>
SRK_DEV void wgrad_ws_body8(const srk_wgrad_args& a, const int slot, const int cib, const int cob, const int tilesX, const int tilesY,
                           const unsigned x_bytes, const unsigned dy_bytes, const int tq, const int trem, char* const smem) {
  typedef WgCfg<DT, 3> C;
  typedef typename C::Tr Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = C::CH, KS = 3, GT = 256, ESZ = C::ESZ;
  constexpr int TH = WS_TH, NBUF = 4, AHEAD = NBUF - 1;
  constexpr int XPIECES = (TH + 2) * C::TIN * 8;       // 16-byte pieces of the halo image: (TH + 2) rows x 18 columns x 8 chunks
  constexpr int NPK = (XPIECES + GT - 1) / GT;         // 6 halo pieces per lane (the image is padded to NPK * GT pieces)
  constexpr int NDK = TH * 16 * 8 / GT;                // 4 dY pieces per lane
  constexpr int PPT = NPK + NDK;                       // vector-memory operations per lane and tile: ALWAYS issued (the counted waits rely on it)
  constexpr int XS_PAD = NPK * GT * 16, DYS = TH * 16 * 128;
  constexpr int BUF_BYTES = XS_PAD + DYS;
  static_assert(NBUF * BUF_BYTES <= 160 * 1024, "the ring must fit the LDS");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, cbk = wave & 1;
  const int H = a.H, W = a.W;
  const int nch_x = a.Cin / CH, nch_d = a.Cout / CH;

  const int t0 = slot * tq + min(slot, trem);
  const int nt = tq + (slot < trem ? 1 : 0);

  // ---- per-lane DMA constants --------------------------------------------------------------------------
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);
  int pconst[NPK], pyx[NPK];
#pragma unroll
  for (int k = 0; k < NPK; ++k) {
    const int i = tid + k * GT;
    const int sl = i & 7, p = i >> 3;
    const int iy = p / C::TIN, ix = p - iy * C::TIN;
    const int c = cib * 8 + (sl ^ swz(ix));
    pconst[k] = (((iy - 1) * W + (ix - 1)) * a.x_pitch + a.x_coff + c * CH) * ESZ;
    pyx[k] = (i < XPIECES && c < nch_x) ? (((iy - 1) & 0xffff) | ((ix - 1) << 16)) : (int)0x7fff7fff;      // (padding pieces: never inside the image)
  }
  // dY piece i = tid + 256k: slot, column and channel chunk do not depend on k, the row advances by 2 per k
  const int rd = a.dy_ps > 1 ? a.dy_ps : 1;
  const int d_ix = (tid >> 3) & 15, d_iy0 = tid >> 7;
  const int d_c = cob * 8 + ((tid & 7) ^ swz(d_ix));
  const bool d_cok = d_c < nch_d;
  int dconst, dkstride;
  {
    const int Csd = a.Cout / (rd * rd);
    const int k0 = d_c * CH;
    const int ij = k0 / Csd, c0 = k0 - ij * Csd;
    const int si = ij / rd, sj = ij - si * rd;
    dconst = (((d_iy0 * rd + si) * (W * rd) + d_ix * rd + sj) * a.dy_pitch + a.dy_coff + c0) * ESZ;
    dkstride = 2 * rd * (W * rd) * a.dy_pitch * ESZ;
  }
  // one tile's DMA = 11 halo pieces + 8 dY pieces per lane.  The pieces are issued one or two per K-step from inside
  // the MFMA loop of the previous tile (an LDS-DMA issue costs 60-100 cycles of the wave's time in front of the loop,
  // with the matrix pipe idle: one wave per SIMD; between MFMAs most of that hides behind the running MFMA)
  struct TileAddr { int y0, x0, xbase, dbase; bool colok, live; unsigned Xb; };   // Xb: LDS byte address of the buffer; live: a tile of this workgroup
  auto tile_addr = [&](int it_, char* Xb) {
    TileAddr t;
    t.live = it_ < nt;
    const int pt = t0 + (t.live ? it_ : 0);
    const int tX = pt % tilesX;
    const int q = pt / tilesX;
    const int tY = q % tilesY, n = q / tilesY;
    t.y0 = tY * TH; t.x0 = tX * 16;
    t.xbase = (((n * H + t.y0) * W + t.x0) * a.x_pitch) * ESZ;
    t.dbase = (((n * H * rd + t.y0 * rd) * (W * rd) + t.x0 * rd) * a.dy_pitch) * ESZ;
    t.colok = t.live && d_cok && (t.x0 + d_ix < W);
    t.Xb = lds0 + (unsigned)(Xb - smem);
    return t;
  };
  auto dma_x_piece = [&](const TileAddr& t, int k) {
    const int gy = t.y0 + (int)(short)(pyx[k] & 0xffff), gx = t.x0 + (pyx[k] >> 16);
    const bool ok = t.live && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const unsigned voff = ok ? (unsigned)(t.xbase + pconst[k]) : 0x80000000u;
    dma16_hidden(xrs, voff, __builtin_amdgcn_readfirstlane(t.Xb + ((k * GT + wave * 64) << 4)));
  };
  auto dma_d_piece = [&](const TileAddr& t, int k) {
    const bool ok = t.colok && (t.y0 + d_iy0 + 2 * k < H);
    const unsigned voff = ok ? (unsigned)(t.dbase + dconst + k * dkstride) : 0x80000000u;
    dma16_hidden(drs, voff, __builtin_amdgcn_readfirstlane(t.Xb + XS_PAD + ((k * GT + wave * 64) << 4)));
  };
  auto dma_tile = [&](int it_, char* Xb) {
    const TileAddr t = tile_addr(it_, Xb);
#pragma unroll
    for (int k = 0; k < NPK; ++k) dma_x_piece(t, k);
#pragma unroll
    for (int k = 0; k < NDK; ++k) dma_d_piece(t, k);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float bsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.dbp != nullptr && cib == 0;
  int xoff[KS][2], doff[2];
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
    xoff[kw][0] = tr_lane_off(kw, 0, rb, lane);
    xoff[kw][1] = tr_lane_off(kw, 1, rb, lane);
  }
  doff[0] = tr_lane_off(0, 0, cbk, lane);
  doff[1] = tr_lane_off(0, 1, cbk, lane);

#pragma unroll
  for (int k = 0; k < AHEAD; ++k) dma_tile(k, smem + k * BUF_BYTES);      // tiles 0 .. 2 (past-the-end ones as out-of-range pieces)
#pragma unroll 1
  for (int it = 0; it < nt; ++it) {
    char* const Xs = smem + (it & (NBUF - 1)) * BUF_BYTES;
    char* const Ds = Xs + XS_PAD;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * PPT) : "memory");     // tile `it` landed; the two younger tiles stay in flight
    __builtin_amdgcn_s_barrier();                         // ... for every wave; the buffer of tile it - 1 is free
    constexpr bool more = true;                           // (a tile is ALWAYS requested: past the end as out-of-range pieces)
    const TileAddr nxt = tile_addr(it + AHEAD, smem + ((it + AHEAD) & (NBUF - 1)) * BUF_BYTES);

    if (do_bias) {
      // bias gradient: lane owns LDS slot (tid&7) of pixels (tid>>3) + 32k; the slot's channel chunk is the same
      // for all 8 of them (the swizzle depends on the column, and 32 pixels = 2 full rows)
#pragma unroll
      for (int k = 0; k < NDK; ++k) {
        const i32x4 raw = lds_read16(Ds + ((tid + 256 * k) << 4));
        const uint32_t w4[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bsum8[2 * i] += Tr::to_f32((uint16_t)(w4[i] & 0xffff));
          bsum8[2 * i + 1] += Tr::to_f32((uint16_t)(w4[i] >> 16));
        }
      }
    }
    // K loop over the 16 tile rows; the fragments of step y+1 (one new halo row, one dY row) are fetched while
    // the 9 MFMAs of step y run (4 rotating halo-row slots), sched_barrier keeps the groups apart
    i32x4 xf[4][KS];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
        xf[rr][kw] = tr_read2(Xs + xoff[kw][0] + rr * (C::PITCH * 128), Xs + xoff[kw][1] + rr * (C::PITCH * 128));
    i32x4 bf = tr_read2(Ds + doff[0], Ds + doff[1]);
#pragma unroll
    for (int y = 0; y < TH; ++y) {
      i32x4 bfn = bf;
      if (y + 1 < TH) {
        const int nr = y + 3;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
          xf[nr & 3][kw] = tr_read2(Xs + xoff[kw][0] + nr * (C::PITCH * 128), Xs + xoff[kw][1] + nr * (C::PITCH * 128));
        bfn = tr_read2(Ds + doff[0] + (y + 1) * (16 * 128), Ds + doff[1] + (y + 1) * (16 * 128));
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int kh = t / KS, kw = t - kh * KS;
        acc[t] = Tr::mma(xf[(y + kh) & 3][kw], bf, acc[t]);
        // next tile's DMA, front-loaded (3 pieces per step in steps 0-5, the rest by step 7): a piece takes 3-5k cycles
        // to land under load and the whole tile must be there when this loop (5.3k cycles) ends
        if (t == 1 && y < NPK && more) dma_x_piece(nxt, y);
        if (t == 5 && y < NDK && more) dma_d_piece(nxt, y);
      }
      bf = bfn;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the past-the-end pieces (zero fill) are LDS writes too: done before the buffers are reused
  // ---- store the workgroup's slab (row-contiguous 128-byte segments per half wave) ------------------------
  // buffer stores: one 32-bit offset per lane, the (tap, row) part is uniform; rows / columns beyond the real
  // channel counts get an out-of-range offset (dropped by the hardware) instead of a branch
  {
    const int co = cob * 64 + cbk * 32 + (lane & 31);
    const int hq = lane >> 5;
    const size_t slab_elems = (size_t)9 * a.Cin * a.Cout;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(a.dwp + (size_t)slot * slab_elems, 0, (unsigned)(slab_elems * 4), 0x00020000);
    const int ci0 = cib * 64 + rb * 32 + 4 * hq;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + (e & 3) + 8 * (e >> 2);
        const unsigned voff = (ci < a.Cin && co < a.Cout) ? (unsigned)(((t * a.Cin + ci) * a.Cout + co) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][e]), srs, voff, 0, 0);
      }
    }
  }
  if (do_bias) {
    // fixed-order reduction of the per-lane chunk sums: channel c lives in chunk c>>3, held by the 32 lanes
    // (hi, q16) with slot (c>>3) ^ swz(q16)
    float* const bred = reinterpret_cast<float*>(smem);      // [256][8]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) bred[tid * 8 + e] = bsum8[e];
    __syncthreads();
    if (tid < 64) {
      const int ch = tid >> 3, e = tid & 7;
      float t = 0.f;
      for (int p0 = 0; p0 < 32; ++p0) t += bred[((p0 << 3) + (ch ^ swz(p0 & 15))) * 8 + e];
      const int c = cob * 64 + tid;
      if (c < a.Cout) a.dbp[(size_t)slot * a.Cout + c] = t;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256, 1) void conv_wgrad_ws_kernel(const srk_wgrad_args a, int tilesX, int tilesY, int th,
                                                                unsigned x_bytes, unsigned dy_bytes, int tq, int trem) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // th: tile height the host cut the images with (ws_tile_height: 8 = the four-buffer ring, 16 = two buffers); uniform over the launch
  if (th == WS_TH) wgrad_ws_body8<DT>(a, blockIdx.x, blockIdx.y, blockIdx.z, tilesX, tilesY, x_bytes, dy_bytes, tq, trem, smem);
  else wgrad_ws_body16<DT>(a, blockIdx.x, blockIdx.y, blockIdx.z, tilesX, tilesY, x_bytes, dy_bytes, tq, trem, smem);
}

// ---- grouped launch: the weight gradients of MANY convolutions in one dispatch ---------------------------------------------
// A training step's weight gradients do not depend on each other, only on (x_l, dy_l) pairs that all exist once the
// data-gradient chain has run.  Launched per layer they are 37 (EDSR-baseline) to 411 (RCAN) dispatches of 144 tiles
// at the reference's batch of 16 -- each one pays its own pipeline fill, slab store and a finalize launch for ~2.5 us
// of matrix work per workgroup.  Here one dispatch walks a device table: block b belongs to job block_job[b], inside
// that job it is (slot, ci block, co block) like the single launch, and `srk_wgrad_group_plan` sizes the slabs so
// that every block gets the same number of tiles whatever its layer.  Same body, same slabs, same finalize.
struct WgJob {
  srk_wgrad_args a;
  int tilesX, tilesY;
  unsigned x_bytes, dy_bytes;
  int tq, trem;
  int ncib, ncob;
  int block0, th;           // th: tile height of this launch (all jobs of a launch: ws_tile_height of the largest one)
};

template <int DT>
__global__ __launch_bounds__(256, 1) void conv_wgrad_ws_group_kernel(const WgJob* __restrict__ jobs, const int* __restrict__ block_job) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The (ci block, co block) workgroups of one tile range read the SAME X and dY tiles.  The hardware deals consecutive block ids to the eight XCDs in turn
  // (each with its own L2), so the logical index is (XCD, slot in the XCD): the ncib x ncob blocks of a range share one L2 (cf. conv_ks.hip).  Any grid size:
  // XCD x holds ceil((G - x) / 8) blocks.  (SRK_WGRAD_XCD_REMAP=0: A/B builds.)
#ifndef SRK_WGRAD_XCD_REMAP
#define SRK_WGRAD_XCD_REMAP 1
#endif
  unsigned lb = blockIdx.x;
  if (SRK_WGRAD_XCD_REMAP) {
    const unsigned q = gridDim.x >> 3, r = gridDim.x & 7u, x = blockIdx.x & 7u;
    lb = x * q + (x < r ? x : r) + (blockIdx.x >> 3);
  }
  const int jid = block_job[lb];
  const WgJob j = jobs[jid];
  const int local = (int)lb - j.block0;
  const int per = j.ncib * j.ncob;
  const int slot = local / per, rem = local - slot * per;
  const int cib = rem / j.ncob, cob = rem - cib * j.ncob;
  if (j.th == WS_TH) wgrad_ws_body8<DT>(j.a, slot, cib, cob, j.tilesX, j.tilesY, j.x_bytes, j.dy_bytes, j.tq, j.trem, smem);
  else wgrad_ws_body16<DT>(j.a, slot, cib, cob, j.tilesX, j.tilesY, j.x_bytes, j.dy_bytes, j.tq, j.trem, smem);
}

// =================================================================================================
// 1x1 weight gradient, slab mode (16-bit):  dW[ci][co] = sum_p X[p][ci] * dY[p][co],  a plain GEMM with K = pixels.
// Used by WDSR's 1x1 expand / reduce convs, RDN's local / global feature fusion and the unfolded 3-channel head conv.
// Arithmetic intensity is M*N/(M+N) FLOP per byte of 16-bit operand (110 for 128 x 768): HBM-bound, so the kernel is
// built around re-reading as little as possible rather than around the matrix pipe:
//  * a workgroup owns CIB x COB channel blocks of 64 (128 x 256 or 256 x 128, the larger side on the larger channel
//    count), i.e. the activations are read Cout/256 (resp. Cin/256) times instead of Cout/64: the v1 atomic kernel
//    (64 x 64 per workgroup) moved 908 MB for WDSR-B's 128 -> 768 layer at batch 64, this one 340 MB;
//  * K tiles of 64 consecutive pixels (NHWC: contiguous), each channel block a 64 x 128 B LDS plane in the swizzled
//    4 x 16-pixel image format of the 3x3 kernel, so both operands are fetched with ds_read_b64_tr_b16;
//    planes double-buffered by hidden LDS-DMA (dma16_hidden), 12 pieces per lane and tile issued between the MFMAs;
//  * wave (rb, cbk) owns the (32 x 32) quadrant of every block pair: CIB*COB accumulator tiles (128 registers);
//  * partial sums go to the workgroup's own slab, the bias gradient is summed from the dY planes on the side
//    (ci-tile 0 only): same scratch layout and srk_wgrad_finalize as the 3x3 slab kernel (bitwise reproducible).
// =================================================================================================
template <int DT, int CIB, int COB>
__global__ __launch_bounds__(256, 1) void wgrad1x1_ws_kernel(const srk_wgrad_args a, int ntiles, unsigned x_bytes, unsigned dy_bytes,
                                                              int tq, int trem) {
  typedef DTraits<DT> Tr;
  constexpr int CH = Tr::CH, ESZ = 2;
  constexpr int NPL = CIB + COB;                  // planes per buffer
  constexpr int PLANE = 64 * 128;                 // 64 pixels x 128 bytes
  constexpr int BUF_BYTES = NPL * PLANE;
  constexpr int NPK = NPL * 2;                    // pieces per lane and tile: plane k >> 1, pixel (tid >> 3) + 32 (k & 1)
  static_assert(Tr::IS16, "16-bit types only");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, cbk = wave & 1;
  const int slot = blockIdx.x, cit = blockIdx.y, cot = blockIdx.z;
  const long long P = (long long)a.N * a.H * a.W;
  const int nch_x = a.Cin / CH, nch_d = a.Cout / CH;
  const int t0 = slot * tq + min(slot, trem);
  const int nt = tq + (slot < trem ? 1 : 0);

  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);
  // per-lane piece constants: LDS slot sl, pixel column (px & 15) fixes the swizzle, so the channel chunk is per lane
  const int px0 = tid >> 3, sl = tid & 7;
  const int cchunk = sl ^ swz(px0 & 15);          // chunk inside the 64-channel block (px0 + 32 has the same column)
  auto dma_tile = [&](int pt, unsigned buf, int k) {            // piece k of tile pt into buffer at LDS address buf
    const int plane = k >> 1;
    const long long p = (long long)pt * 64 + px0 + 32 * (k & 1);
    const bool isx = plane < CIB;
    const int blk = isx ? cit * CIB + plane : cot * COB + (plane - CIB);
    const int chunk = blk * 8 + cchunk;
    const bool ok = p < P && chunk < (isx ? nch_x : nch_d);
    const unsigned voff = ok ? (unsigned)((p * (isx ? a.x_pitch : a.dy_pitch) + (isx ? a.x_coff : a.dy_coff) + chunk * CH) * ESZ) : 0x80000000u;
    dma16_hidden(isx ? xrs : drs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(buf + plane * PLANE + (((k & 1) * 256 + wave * 64) << 4))));
  };

  f32x16 acc[CIB][COB];
#pragma unroll
  for (int i = 0; i < CIB; ++i)
#pragma unroll
    for (int o = 0; o < COB; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][o][e] = 0.f;
  float bsum[COB][8];
#pragma unroll
  for (int o = 0; o < COB; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[o][e] = 0.f;
  const bool do_bias = a.dbp != nullptr && cit == 0;
  int aoff[2], boff[2];
  aoff[0] = tr_lane_off(0, 0, rb, lane);  aoff[1] = tr_lane_off(0, 1, rb, lane);
  boff[0] = tr_lane_off(0, 0, cbk, lane); boff[1] = tr_lane_off(0, 1, cbk, lane);

  if (nt > 0) {
#pragma unroll
    for (int k = 0; k < NPK; ++k) dma_tile(t0, lds0, k);
  }
#pragma unroll 1
  for (int it = 0; it < nt; ++it) {
    const char* const B0 = smem + (it & 1) * BUF_BYTES;
    const unsigned nb = lds0 + ((it + 1) & 1) * BUF_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile `it` landed
    __builtin_amdgcn_s_barrier();                         // ... for every wave; the other buffer is free
    const bool more = it + 1 < nt;
    if (do_bias) {
      // lane owns LDS slot sl of pixels px0 and px0 + 32 of every dY plane
#pragma unroll
      for (int o = 0; o < COB; ++o)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const i32x4 raw = lds_read16(B0 + (CIB + o) * PLANE + ((tid + 256 * hf) << 4));
          const uint32_t w4[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bsum[o][2 * i] += Tr::to_f32((uint16_t)(w4[i] & 0xffff));
            bsum[o][2 * i + 1] += Tr::to_f32((uint16_t)(w4[i] >> 16));
          }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                 // K-step = one 16-pixel row of the planes
      i32x4 af[CIB], bf[COB];
#pragma unroll
      for (int i = 0; i < CIB; ++i) af[i] = tr_read2(B0 + i * PLANE + r * 2048 + aoff[0], B0 + i * PLANE + r * 2048 + aoff[1]);
#pragma unroll
      for (int o = 0; o < COB; ++o) bf[o] = tr_read2(B0 + (CIB + o) * PLANE + r * 2048 + boff[0], B0 + (CIB + o) * PLANE + r * 2048 + boff[1]);
      int k = r * (NPK / 4);
#pragma unroll
      for (int i = 0; i < CIB; ++i)
#pragma unroll
        for (int o = 0; o < COB; ++o) {
          acc[i][o] = Tr::mma(af[i], bf[o], acc[i][o]);
          if (more && k < (r + 1) * (NPK / 4) && ((i * COB + o) & 1) == 0) { dma_tile(t0 + it + 1, nb, k); ++k; }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the workgroup's slab: [Cin][Cout] fp32, rows / columns beyond the real channel counts are dropped -------
  {
    const int hq = lane >> 5;
    const size_t slab_elems = (size_t)a.Cin * a.Cout;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(a.dwp + (size_t)slot * slab_elems, 0, (unsigned)(slab_elems * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < CIB; ++i)
#pragma unroll
      for (int o = 0; o < COB; ++o) {
        const int ci0 = (cit * CIB + i) * 64 + rb * 32 + 4 * hq;
        const int co = (cot * COB + o) * 64 + cbk * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + (e & 3) + 8 * (e >> 2);
          const unsigned voff = (ci < a.Cin && co < a.Cout) ? (unsigned)((ci * a.Cout + co) * 4) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][o][e]), srs, voff, 0, 0);
        }
      }
  }
  if (do_bias) {
    // fixed-order reduction of the per-lane chunk sums (same scheme as the 3x3 slab kernel), one dY plane at a time
    float* const bred = reinterpret_cast<float*>(smem);      // [256][8]
#pragma unroll
    for (int o = 0; o < COB; ++o) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) bred[tid * 8 + e] = bsum[o][e];
      __syncthreads();
      if (tid < 64) {
        const int ch = tid >> 3, e = tid & 7;
        float t = 0.f;
        for (int p0 = 0; p0 < 32; ++p0) t += bred[((p0 << 3) + (ch ^ swz(p0 & 15))) * 8 + e];
        const int c = (cot * COB + o) * 64 + tid;
        if (c < a.Cout) a.dbp[(size_t)slot * a.Cout + c] = t;
      }
    }
  }
}

// ---- at most 64 x 64 channels (the unfolded 3-channel head conv: 27 (32) x 64; RCAN / RDN never get here with fewer than 128) ---------
// The kernel above moves 12 KB of real operands per 64-pixel tile there (6 of its 8 planes lie beyond the channel counts) and keeps ONE tile
// in flight: 256 CUs x 12 KB per ~2 us of memory latency = 1.7 TB/s, 65.5 us for the head conv of a 256 x 48 x 48 batch (rocprofv3, round 5).
// Same tiles, same planes, same MFMA order (bit-identical slabs), but ONE 64 x 64 block pair and a ring of NBUF tile buffers of 16 KB:
// NBUF - 1 tiles in flight.  Every iteration requests exactly NPK pieces (tiles beyond the slot's range as out-of-range offsets: zeros
// into a buffer nobody reads), so the counted wait in front of a tile is a constant.
template <int DT, int NBUF>
__global__ __launch_bounds__(256, 1) void wgrad1x1_small_kernel(const srk_wgrad_args a, int ntiles, unsigned x_bytes, unsigned dy_bytes,
                                                                 int tq, int trem) {
  typedef DTraits<DT> Tr;
  constexpr int CH = Tr::CH, ESZ = 2;
  constexpr int PLANE = 64 * 128, BUF_BYTES = 2 * PLANE, NPK = 4;
  static_assert(Tr::IS16, "16-bit types only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, cbk = wave & 1;
  const int slot = blockIdx.x;
  const long long P = (long long)a.N * a.H * a.W;
  const int nch_x = a.Cin / CH, nch_d = a.Cout / CH;
  const int t0 = slot * tq + min(slot, trem);
  const int nt = tq + (slot < trem ? 1 : 0);
  const i32x4 xrs = make_rsrc4(a.x, x_bytes), drs = make_rsrc4(a.dy, dy_bytes);
  const unsigned lds0 = lds_addr_of(smem);
  const int px0 = tid >> 3, sl = tid & 7;
  const int cchunk = sl ^ swz(px0 & 15);
  auto dma_tile = [&](int j) {                          // the slot's tile j (none: j >= nt) into ring buffer j % NBUF
    const unsigned buf = lds0 + (unsigned)(j % NBUF) * BUF_BYTES;
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
      const int plane = k >> 1;
      const long long p = (long long)(t0 + j) * 64 + px0 + 32 * (k & 1);
      const bool isx = plane == 0;
      const bool ok = j < nt && p < P && cchunk < (isx ? nch_x : nch_d);
      const unsigned voff = ok ? (unsigned)((p * (isx ? a.x_pitch : a.dy_pitch) + (isx ? a.x_coff : a.dy_coff) + cchunk * CH) * ESZ) : 0x80000000u;
      dma16_hidden(isx ? xrs : drs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(buf + plane * PLANE + (((k & 1) * 256 + wave * 64) << 4))));
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.dbp != nullptr;
  const int aoff0 = tr_lane_off(0, 0, rb, lane), aoff1 = tr_lane_off(0, 1, rb, lane);
  const int boff0 = tr_lane_off(0, 0, cbk, lane), boff1 = tr_lane_off(0, 1, cbk, lane);
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j) dma_tile(j);
#pragma unroll 1
  for (int it = 0; it < nt; ++it) {
    const char* const B0 = smem + (it % NBUF) * BUF_BYTES;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * NPK) : "memory");     // tile `it` landed (the NBUF - 2 younger tiles may be on their way)
    __builtin_amdgcn_s_barrier();                                                // ... for every wave; tile it - 1's buffer is free
    dma_tile(it + NBUF - 1);
    if (do_bias) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const i32x4 raw = lds_read16(B0 + PLANE + ((tid + 256 * hf) << 4));
        const uint32_t w4[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bsum[2 * i] += Tr::to_f32((uint16_t)(w4[i] & 0xffff));
          bsum[2 * i + 1] += Tr::to_f32((uint16_t)(w4[i] >> 16));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const i32x4 af = tr_read2(B0 + r * 2048 + aoff0, B0 + r * 2048 + aoff1);
      const i32x4 bf = tr_read2(B0 + PLANE + r * 2048 + boff0, B0 + PLANE + r * 2048 + boff1);
      acc = Tr::mma(af, bf, acc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's trailing (empty) requests, before the LDS is reused below
  {
    const int hq = lane >> 5;
    const size_t slab_elems = (size_t)a.Cin * a.Cout;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(a.dwp + (size_t)slot * slab_elems, 0, (unsigned)(slab_elems * 4), 0x00020000);
    const int ci0 = rb * 32 + 4 * hq, co = cbk * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = ci0 + (e & 3) + 8 * (e >> 2);
      const unsigned voff = (ci < a.Cin && co < a.Cout) ? (unsigned)((ci * a.Cout + co) * 4) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e]), srs, voff, 0, 0);
    }
  }
  if (do_bias) {
    float* const bred = reinterpret_cast<float*>(smem);      // [256][8]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) bred[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 64) {
      const int ch = tid >> 3, e = tid & 7;
      float t = 0.f;
      for (int p0 = 0; p0 < 32; ++p0) t += bred[((p0 << 3) + (ch ^ swz(p0 & 15))) * 8 + e];
      if (tid < a.Cout) a.dbp[(size_t)slot * a.Cout + tid] = t;
    }
  }
}

// pixel slabs used by the slab-mode kernel for these arguments (0: atomic-mode kernel)
// 1x1 slab kernel: tile shape (ci blocks x co blocks of 64) and slab count
static void wgrad1x1_shape(const srk_wgrad_args& a, int& cib, int& cob) {
  if (a.Cout >= a.Cin) { cib = 2; cob = 4; } else { cib = 4; cob = 2; }
}
static int wgrad1x1_slabs(const srk_wgrad_args& a) {
  if (a.dtype == SRK_F32 || a.KH != 1 || a.KW != 1 || a.x_ps > 1 || a.dy_ps > 1) return 0;
  static const bool no_ws = srk_dbg_getenv("SRK_NO_WS") != nullptr;      // diagnostics knob, read once
  if (no_ws) return 0;
  const long long P = (long long)a.N * a.H * a.W;
  if (P * a.x_pitch * 2 >= 0x7fff0000LL || P * a.dy_pitch * 2 >= 0x7fff0000LL) return 0;
  int cib, cob;
  wgrad1x1_shape(a, cib, cob);
  const int cit = (a.Cin + 64 * cib - 1) / (64 * cib), cot = (a.Cout + 64 * cob - 1) / (64 * cob);
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  const long long ntiles = (P + 63) / 64;
  long long slabs = cus / (cit * cot);
  if (slabs < 1) slabs = 1;
  if (slabs > ntiles) slabs = ntiles;
  return (int)slabs;
}

// Tile height of a slab-mode 3x3 launch: the 8-row four-buffer ring where a workgroup walks many tiles (>= 1,024 16-row tiles per convolution: batch 256 of
// 48 x 48), else the 16-row two-buffer form.  SRK_WGRAD_TH=8 / 16 (under SRK_DEBUG=1) forces one (A/B).
static int ws_tile_height(long long n_images, int H, int W) {
  static const int forced = [] { const char* e = srk_dbg_getenv("SRK_WGRAD_TH"); return e ? atoi(e) : 0; }();
  if (forced == 8 || forced == 16) return forced;
  return n_images * ((H + 15) / 16) * ((W + 15) / 16) >= 1024 ? WS_TH : 16;
}

static int wgrad_ws_slabs(const srk_wgrad_args& a) {
  if (a.KH == 1 && a.KW == 1) return wgrad1x1_slabs(a);
  if (a.KH > 3) return srk_wgrad_lk_ok(a) ? srk_wgrad_lk_slabs(a) : 0;
  if (a.dtype == SRK_F32 || a.KH != 3 || a.KW != 3 || a.x_ps > 1) return 0;
  static const bool no_ws = srk_dbg_getenv("SRK_NO_WS") != nullptr;      // diagnostics knob, read once
  if (no_ws) return 0;
  const int rd = a.dy_ps > 1 ? a.dy_ps : 1;
  const long long xb = (long long)a.N * a.H * a.W * a.x_pitch * 2, db = (long long)a.N * a.H * a.W * rd * rd * a.dy_pitch * 2;
  if (xb >= 0x7fff0000LL || db >= 0x7fff0000LL) return 0;
  const int th = ws_tile_height(a.N, a.H, a.W);
  const long long ntiles = (long long)a.N * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
  const int cib = (a.Cin + 63) / 64, cob = (a.Cout + 63) / 64;
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  long long slabs = cus / (cib * cob);
  if (slabs < 1) slabs = 1;
  if (slabs > ntiles) slabs = ntiles;
  return (int)slabs;
}

constexpr int WS_LDS_BYTES = 160 * 1024;                 // four (padded halo image, gradient tile) buffers: wgrad_ws_body

template <int DT> int launch_ws(const srk_wgrad_args& a, hipStream_t st, int slabs) {
  typedef WgCfg<DT, 3> C;
  constexpr int LDS = WS_LDS_BYTES;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ws_kernel<DT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d_wgrad(ws): cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return (int)attr;
  }
  const int th = ws_tile_height(a.N, a.H, a.W);
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + th - 1) / th;
  const long long ntiles = (long long)a.N * tilesX * tilesY;
  const int cib = (a.Cin + 63) / 64, cob = (a.Cout + 63) / 64;
  const int rd = a.dy_ps > 1 ? a.dy_ps : 1;
  const unsigned xb = (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2);
  const unsigned db = (unsigned)((long long)a.N * a.H * a.W * rd * rd * a.dy_pitch * 2);
  hipLaunchKernelGGL((conv_wgrad_ws_kernel<DT>), dim3(slabs, cib, cob), dim3(256), LDS, st, a, tilesX, tilesY, th, xb, db,
                     (int)(ntiles / slabs), (int)(ntiles % slabs));
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT, int CIB, int COB> int launch_1x1(const srk_wgrad_args& a, hipStream_t st, int slabs) {
  constexpr int LDS = 2 * (CIB + COB) * 64 * 128;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_ws_kernel<DT, CIB, COB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d_wgrad(1x1): cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return (int)attr;
  }
  const long long P = (long long)a.N * a.H * a.W;
  const long long ntiles = (P + 63) / 64;
  const int cit = (a.Cin + 64 * CIB - 1) / (64 * CIB), cot = (a.Cout + 64 * COB - 1) / (64 * COB);
  hipLaunchKernelGGL((wgrad1x1_ws_kernel<DT, CIB, COB>), dim3(slabs, cit, cot), dim3(256), LDS, st, a, (int)ntiles,
                     (unsigned)(P * a.x_pitch * 2), (unsigned)(P * a.dy_pitch * 2), (int)(ntiles / slabs), (int)(ntiles % slabs));
  SRK_LAUNCH_CHECK();
  return 0;
}
template <int DT> int launch_1x1_small(const srk_wgrad_args& a, hipStream_t st, int slabs) {
  constexpr int NBUF = 8, LDS = NBUF * 2 * 64 * 128;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_small_kernel<DT, NBUF>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d_wgrad(1x1 small): cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(attr));
    return (int)attr;
  }
  const long long P = (long long)a.N * a.H * a.W;
  const long long ntiles = (P + 63) / 64;
  hipLaunchKernelGGL((wgrad1x1_small_kernel<DT, NBUF>), dim3(slabs), dim3(256), LDS, st, a, (int)ntiles,
                     (unsigned)(P * a.x_pitch * 2), (unsigned)(P * a.dy_pitch * 2), (int)(ntiles / slabs), (int)(ntiles % slabs));
  SRK_LAUNCH_CHECK();
  return 0;
}
template <int DT> int launch_1x1_any(const srk_wgrad_args& a, hipStream_t st, int slabs) {
  static const bool no_small = srk_dbg_getenv("SRK_NO_WGRAD1X1_SMALL") != nullptr;      // A/B knob, read once
  if (a.Cin <= 64 && a.Cout <= 64 && !no_small) return launch_1x1_small<DT>(a, st, slabs);
  int cib, cob;
  wgrad1x1_shape(a, cib, cob);
  return cib == 2 ? launch_1x1<DT, 2, 4>(a, st, slabs) : launch_1x1<DT, 4, 2>(a, st, slabs);
}

template <int DT, int KS> int launch(const srk_wgrad_args& a, hipStream_t st) {
  typedef WgCfg<DT, KS> C;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<DT, KS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
  if (attr != hipSuccess) {
    srk_set_error("srk_conv2d_wgrad: cannot reserve %d bytes of LDS: %s", C::LDS_BYTES, hipGetErrorString(attr));
    return (int)attr;
  }
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  const long long ntiles = (long long)a.N * tilesX * tilesY;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) {
    srk_set_error("srk_conv2d_wgrad: bad tile count %lld", ntiles);
    return SRK_E_BADARG;
  }
  const int cib = (a.Cin + C::CBLK - 1) / C::CBLK, cob = (a.Cout + C::CBLK - 1) / C::CBLK;
  int slabs = 256 / (cib * cob);
  if (slabs < 1) slabs = 1;
  if (slabs > ntiles) slabs = (int)ntiles;
  hipLaunchKernelGGL((conv_wgrad_kernel<DT, KS>), dim3(slabs, cib, cob), dim3(C::NT), C::LDS_BYTES, st, a, tilesX, tilesY,
                     (int)ntiles);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT> int dispatch(const srk_wgrad_args& a, hipStream_t st) {
  return a.KH == 3 ? launch<DT, 3>(a, st) : launch<DT, 1>(a, st);
}

}  // namespace

extern "C" int srk_conv2d_wgrad(const srk_wgrad_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->dy && a->dwp, "srk_conv2d_wgrad: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "srk_conv2d_wgrad: bad dims");
  SRK_CHECK_ARG(a->KH == a->KW && (a->KH == 1 || a->KH == 3 || srk_wgrad_lk_ok(*a)), "srk_conv2d_wgrad: kernel %dx%d not supported", a->KH, a->KW);
  SRK_CHECK_ARG(a->Cin % 16 == 0 && a->Cout % 16 == 0 && a->Cin > 0 && a->Cout > 0, "srk_conv2d_wgrad: Cin=%d Cout=%d must be multiples of 16", a->Cin, a->Cout);
  SRK_CHECK_ARG(a->dtype >= SRK_BF16 && a->dtype <= SRK_F32, "srk_conv2d_wgrad: dtype %d", a->dtype);
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->x_pitch % ch == 0 && a->x_coff % ch == 0 && a->dy_pitch % ch == 0 && a->dy_coff % ch == 0,
                "srk_conv2d_wgrad: pitch/offset must be 16-byte aligned");
  const int rx = a->x_ps > 1 ? a->x_ps : 1, rd = a->dy_ps > 1 ? a->dy_ps : 1;
  SRK_CHECK_ARG(a->Cin % (rx * rx) == 0 && (a->Cin / (rx * rx)) % ch == 0 && a->Cout % (rd * rd) == 0 && (a->Cout / (rd * rd)) % ch == 0,
                "srk_conv2d_wgrad: pixel-shuffled operand incompatible with channel count");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = wgrad_ws_slabs(*a);
  SRK_CHECK_ARG(a->nslabs == slabs, "srk_conv2d_wgrad: nslabs=%d but srk_wgrad_slabs() is %d for these arguments", a->nslabs, slabs);
  if (a->KH > 3) return srk_wgrad_lk_launch(*a, st);
  if (slabs > 0 && a->KH == 1) return a->dtype == SRK_BF16 ? launch_1x1_any<SRK_BF16>(*a, st, slabs) : launch_1x1_any<SRK_F16>(*a, st, slabs);
  if (slabs > 0) return a->dtype == SRK_BF16 ? launch_ws<SRK_BF16>(*a, st, slabs) : launch_ws<SRK_F16>(*a, st, slabs);
  switch (a->dtype) {
    case SRK_BF16: return dispatch<SRK_BF16>(*a, st);
    case SRK_F16: return dispatch<SRK_F16>(*a, st);
    default: return dispatch<SRK_F32>(*a, st);
  }
}

extern "C" int srk_wgrad_slabs(const srk_wgrad_args* a) { return a ? wgrad_ws_slabs(*a) : 0; }
extern "C" int srk_wgrad_slab_cout(const srk_wgrad_args* a) {
  if (!a) return 0;
  return (a->KH > 3 && srk_wgrad_lk_ok(*a)) ? srk_wgrad_lk_slab_cout(*a) : a->Cout;
}

// ---- grouped launch: planning (host) and dispatch --------------------------------------------------------------------------
extern "C" int srk_wgrad_group_job_bytes(void) { return (int)sizeof(WgJob); }

extern "C" int srk_wgrad_group_ok(const srk_wgrad_args* a) {
  // jobs the grouped kernel takes: what the 3x3 slab kernel takes
  return (a && a->KH == 3 && a->KW == 3 && a->N > 0 && wgrad_ws_slabs(*a) > 0) ? 1 : 0;
}

extern "C" int srk_wgrad_group_plan(srk_wgrad_args* jobs, int n, float* scratch, void* table_host, int* block_job_host,
                                    int* nblocks_out, long long* scratch_floats_out) {
  SRK_CHECK_ARG(jobs && n > 0 && nblocks_out && scratch_floats_out, "srk_wgrad_group_plan: null pointer / no jobs");
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  // one tile height for the whole launch: the largest job's (jobs of a backward pass share their batch)
  int th = 16;
  for (int i = 0; i < n; ++i)
    if (ws_tile_height(jobs[i].N, jobs[i].H, jobs[i].W) == WS_TH) th = WS_TH;
  long long units = 0, maxnt = 0;
  for (int i = 0; i < n; ++i) {
    const srk_wgrad_args& a = jobs[i];
    SRK_CHECK_ARG(srk_wgrad_group_ok(&a), "srk_wgrad_group_plan: job %d is not a 16-bit 3x3 slab-mode weight gradient", i);
    SRK_CHECK_ARG(a.dtype == jobs[0].dtype, "srk_wgrad_group_plan: job %d has another dtype", i);
    const long long nt = (long long)a.N * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
    units += nt * ((a.Cin + 63) / 64) * ((a.Cout + 63) / 64);
    if (nt > maxnt) maxnt = nt;
  }
  // tiles per workgroup T: every block gets <= T tiles of ONE job; blocks are dispatched as CUs free up, so the launch
  // takes about ceil(blocks / CUs) rounds of (T + fill/store overhead) tile times.  Scan T upwards from the even split.
  auto blocks_for = [&](long long T) {
    long long nb = 0;
    for (int i = 0; i < n; ++i) {
      const srk_wgrad_args& a = jobs[i];
      const long long nt = (long long)a.N * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
      nb += ((nt + T - 1) / T) * ((a.Cin + 63) / 64) * ((a.Cout + 63) / 64);
    }
    return nb;
  };
  // candidates from 8 tiles per block up to the largest job (geometric steps) plus the even split; more blocks than CUs
  // is fine -- RCAN has 411 jobs -- as long as the rounds fill: cost = rounds x (T + 3 tiles of fill / slab-store overhead)
  long long bestT = 0;
  double best = 1e300;
  auto consider = [&](long long t) {
    if (t < 1) t = 1;
    if (t > maxnt) t = maxnt;
    const long long nb = blocks_for(t);
    const double cost = (double)((nb + cus - 1) / cus) * (double)(t + (th == WS_TH ? 6 : 3));      // (fill / slab-store overhead: 3 tiles of 16 rows)
    if (cost < best || (cost == best && t > bestT)) { best = cost; bestT = t; }
  };
  consider((units + cus - 1) / cus);
  for (long long t = (th == WS_TH ? 16 : 8); t <= maxnt; t = t + 1 + t / 12) consider(t);
  consider(maxnt);
  if (bestT == 0) bestT = maxnt;
  long long off = 0, nb = 0;
  WgJob* tab = reinterpret_cast<WgJob*>(table_host);
  for (int i = 0; i < n; ++i) {
    srk_wgrad_args& a = jobs[i];
    const int tilesX = (a.W + 15) / 16, tilesY = (a.H + th - 1) / th;
    const long long nt = (long long)a.N * tilesX * tilesY;
    const int slabs = (int)((nt + bestT - 1) / bestT);
    const int ncib = (a.Cin + 63) / 64, ncob = (a.Cout + 63) / 64;
    const long long per = 9LL * a.Cin * a.Cout;
    a.nslabs = slabs;
    const bool want_b = a.dbp != nullptr;          // in: non-NULL = bias gradient wanted
    if (scratch) {                                 // sizing pass (scratch == NULL): only nslabs and the totals
      a.dwp = scratch + off;
      a.dbp = want_b ? scratch + off + (long long)slabs * per : nullptr;
    }
    off += (long long)slabs * (per + a.Cout);
    if (tab) {
      SRK_CHECK_ARG(scratch, "srk_wgrad_group_plan: the table needs the scratch pointer");
      WgJob& j = tab[i];
      j.a = a;
      const int rd = a.dy_ps > 1 ? a.dy_ps : 1;
      j.tilesX = tilesX; j.tilesY = tilesY;
      j.x_bytes = (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2);
      j.dy_bytes = (unsigned)((long long)a.N * a.H * a.W * rd * rd * a.dy_pitch * 2);
      j.tq = (int)(nt / slabs); j.trem = (int)(nt % slabs);
      j.ncib = ncib; j.ncob = ncob; j.block0 = (int)nb; j.th = th;
    }
    const long long cnt = (long long)slabs * ncib * ncob;
    if (block_job_host)
      for (long long b = 0; b < cnt; ++b) block_job_host[nb + b] = i;
    nb += cnt;
  }
  SRK_CHECK_ARG(nb < 0x7fffffffLL, "srk_wgrad_group_plan: %lld blocks", nb);
  *nblocks_out = (int)nb;
  *scratch_floats_out = off;
  return 0;
}

extern "C" int srk_conv2d_wgrad_group(const void* table_dev, const int* block_job_dev, int nblocks, int dtype, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && block_job_dev && nblocks > 0, "srk_conv2d_wgrad_group: null pointer / no blocks");
  SRK_CHECK_ARG(dtype == SRK_BF16 || dtype == SRK_F16, "srk_conv2d_wgrad_group: 16-bit dtypes only");
  typedef WgCfg<SRK_BF16, 3> C;
  constexpr int LDS = WS_LDS_BYTES;
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ws_group_kernel<SRK_BF16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ws_group_kernel<SRK_F16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr0 != hipSuccess || attr1 != hipSuccess) {
    srk_set_error("srk_conv2d_wgrad_group: cannot reserve %d bytes of LDS", LDS);
    return (int)(attr0 != hipSuccess ? attr0 : attr1);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const WgJob* tab = reinterpret_cast<const WgJob*>(table_dev);
  if (dtype == SRK_BF16) hipLaunchKernelGGL((conv_wgrad_ws_group_kernel<SRK_BF16>), dim3(nblocks), dim3(256), LDS, st, tab, block_job_dev);
  else hipLaunchKernelGGL((conv_wgrad_ws_group_kernel<SRK_F16>), dim3(nblocks), dim3(256), LDS, st, tab, block_job_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}
