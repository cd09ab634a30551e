// D-DBPN's projection convolutions at scale 4, directly (no column tensor in HBM): nn.Conv2d / nn.ConvTranspose2d with kernel 8,
// stride 4, padding 2 and 32 channels on both sides (reference: models/ddbpn.py:10-24 `projection_conv`, used by
// `DenseProjection`, ddbpn.py:27-64).  16-bit storage, fp32 accumulation on the MFMA pipe.
//
// One weight tensor convention serves both module kinds:  W4[cl][ch][ky][kx]  with cl the channel on the LOW-resolution side and
// ch the channel on the HIGH-resolution side -- Conv2d's [out][in][ky][kx] and ConvTranspose2d's [in][out][ky][kx] both read that
// way -- and three kernels cover forward, data gradient and weight gradient of both:
//
//   down  (HR -> LR):  out[q][cl]        = b[cl] + sum_{k, ch} xh[4q - 2 + k][ch] * W4[cl][ch][k]      Conv2d forward, ConvTranspose2d dgrad
//   up    (LR -> HR):  out[4q - 2 + k][ch] += x[q][cl] * W4[cl][ch][k]  (+ b[ch])                       ConvTranspose2d forward, Conv2d dgrad
//   wgrad           :  dW4[cl][ch][k]    = sum_q g[q][cl] * xh[4q - 2 + k][ch]                          (xh, g) = (x, dy) resp. (dy, x)
//
// All three are bound by the 32-channel HR tensor (64 bytes per pixel, 16 x the LR tensor): 4.8 GFLOP against 37.7 MB at the
// reference's batch (16 patches of 48 x 48 LR pixels), i.e. ~10 us of HBM time and ~3 us of MFMA time per launch; the im2col form
// they replace moved a 151 MB column tensor per launch through a 32-column GEMM (107 + 39 us).
//
//  * down / up keep the WHOLE weight tensor in registers: wave w of a workgroup owns kernel rows 2w, 2w+1 (down) resp. the output
//    phase row ry = w (up) -- 32 MFMA operand fragments = 128 registers -- and the persistent workgroups walk 8 x 4-pixel LR tiles.
//  * down: the (20 x 36)-pixel HR halo of a tile sits in LDS (quads of 4 pixels 272 bytes apart, rows 2464: the 32 pixel lanes of
//    an operand read are 4 HR pixels apart and land in 16 different 16-byte bank groups); every wave runs its 32 MFMAs over the
//    tile's 32 pixels, the four partial sums meet in LDS, 256 threads add them with the bias and store 8 bytes each.
//  * up: an output pixel (4q + r) takes the 2 x 2 input pixels {q, q + (r < 2 ? -1 : +1)} per dimension: 4 taps x 32 channels = 8
//    MFMAs per phase and 32-pixel tile; the wave's 4 x 32 x 32 results go through a wave-private, XOR-swizzled 8 KB LDS stage and
//    leave as 16-byte stores covering whole 2 KB row segments of the HR tensor.
//  * wgrad: K = pixels, so both operands are fetched with the transposing LDS read (ds_read_b64_tr_b16); a workgroup owns one
//    kernel row ky and a slice of the tiles (it reads only the HR rows = ky - 2 mod 4 of them), its waves own two kx each; the
//    partial sums of the slices are added by srk_proj_wgrad's second launch, in slice order (bitwise reproducible).
#include <stdlib.h>
#include "srk_common.h"

namespace {

constexpr int PJ_TX = 8, PJ_TY = 4;                 // LR pixels per tile (one 32-pixel MFMA operand)
// ---- down ----
constexpr int D_ROWS = 4 * PJ_TY + 4, D_COLS = 4 * PJ_TX + 4;          // 20 x 36 HR pixels
constexpr int D_QP = 272, D_RP = (D_COLS / 4) * D_QP + 16;             // quad pitch, row pitch (2464)
constexpr int D_HALO = D_ROWS * D_RP;                                  // 49,280
constexpr int D_LDS = D_HALO + 4 * 32 * 32 * 4;                        // + the four partial sums
constexpr int D_CHUNKS = D_ROWS * D_COLS * 4, D_NST = (D_CHUNKS + 255) / 256;     // 2,880 chunks, 12 per thread
// ---- up ----
constexpr int U_PP = 80, U_RP = 896;                                   // pixel pitch, row pitch of the (6 x 10)-pixel LR halo
constexpr int U_CONST = 256;                                           // bias[32], slope[32] (fp32) in front of the halo
constexpr int U_HALO = U_CONST + (PJ_TY + 2) * U_RP;                   // 5,632
constexpr int U_LDS = U_HALO + 4 * 8192;                                   // + 4 x 8192 more with the fused PReLU (pre-activation stage)
// ---- wgrad ----
constexpr int G_QP = 320, G_RP = (D_COLS / 4) * G_QP;                  // 2,880
constexpr int G_TY = 8;                                                // wgrad tiles: 8 x 8 LR pixels
constexpr int G_ROWS = G_TY + 1;                                       // HR rows per tile: tap ky + 4 of LR row q = tap ky of row q + 1
constexpr int G_X = G_ROWS * G_RP, G_G = G_TY * 8 * 64;
constexpr int G_LDS = G_X + G_G;
constexpr int G_CHUNKS = G_ROWS * D_COLS * 4, G_NST = (G_CHUNKS + 255) / 256;

template <int DT> SRK_DEV uint16_t cvt16(float f) { return DTraits<DT>::from_f32(f); }

// ------------------------------------------------------------------------------------------------------------------
// weights -> MFMA fragment order (both directions in one launch).  Fragment = 64 lanes x 8 elements; lane l supplies
// MFMA row l % 32 and k = 8 (l / 32) + e.
//   down: [wave w][kyl 2][kx 8][kb 2]   row rho -> cl = row_to_chan(rho, 32) (a lane's 16 accumulators = 16 adjacent channels),
//                                       k -> ch = 16 kb + ..,  ky = 2w + kyl
//   up  : [wave ry][rx 4][ty 2][tx 2][kb 2]   row rho -> ch = row_to_chan(rho, 32), k -> cl,
//                                       ky = ry + 2 (ty = 0) or ry + 6 / ry - 2 (ty = 1: the neighbour q - 1 resp. q + 1)
// ------------------------------------------------------------------------------------------------------------------
template <int DT> __global__ void proj_pack_kernel(const float* __restrict__ w4, uint16_t* __restrict__ down, uint16_t* __restrict__ up) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // 0 .. 2 * 65536
  const int which = i >> 16, j = i & 65535;
  const int e = j & 7, lane = (j >> 3) & 63, f = (j >> 9) & 31, w = j >> 14;
  const int c = row_to_chan(lane & 31, 32), kk = (lane >> 5) * 8 + e;
  if (which == 0) {
    const int kb = f & 1, kx = (f >> 1) & 7, kyl = f >> 4;
    const int cl = c, ch = kb * 16 + kk, ky = 2 * w + kyl;
    down[j] = cvt16<DT>(w4[((cl * 32 + ch) * 8 + ky) * 8 + kx]);
  } else {
    const int kb = f & 1, tx = (f >> 1) & 1, ty = (f >> 2) & 1, rx = f >> 3, ry = w;
    const int ch = c, cl = kb * 16 + kk;
    const int ky = ty == 0 ? ry + 2 : (ry < 2 ? ry + 6 : ry - 2);
    const int kx = tx == 0 ? rx + 2 : (rx < 2 ? rx + 6 : rx - 2);
    up[j] = cvt16<DT>(w4[((cl * 32 + ch) * 8 + ky) * 8 + kx]);
  }
}

// nn.PReLU on two stored (already rounded) 16-bit values: what a separate PReLU pass over the stored tensor computes
template <int DT> SRK_DEV uint32_t prelu_pk(uint32_t w, float s0, float s1) {
  float lo, hi;
  unpack2<DT>(w, lo, hi);
  return pack2<DT>(lo > 0.f ? lo : lo * s0, hi > 0.f ? hi : hi * s1);
}

// the same for a table of (fp32 weights, packed buffer) pairs: blockIdx.y = entry (all of D-DBPN's 33 projections in one launch)
template <int DT> __global__ void proj_pack_group_kernel(const srk_proj_pack_job* __restrict__ jobs) {
  const srk_proj_pack_job j = jobs[blockIdx.y];
  const float* w4 = j.w4;
  uint16_t* down = reinterpret_cast<uint16_t*>(j.wpk);
  uint16_t* up = down + 65536;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int which = i >> 16, jj = i & 65535;
  const int e = jj & 7, lane = (jj >> 3) & 63, f = (jj >> 9) & 31, w = jj >> 14;
  const int c = row_to_chan(lane & 31, 32), kk = (lane >> 5) * 8 + e;
  if (which == 0) {
    const int kb = f & 1, kx = (f >> 1) & 7, kyl = f >> 4;
    down[jj] = cvt16<DT>(w4[((c * 32 + kb * 16 + kk) * 8 + 2 * w + kyl) * 8 + kx]);
  } else {
    const int kb = f & 1, tx = (f >> 1) & 1, ty = (f >> 2) & 1, rx = f >> 3, ry = w;
    const int ky = ty == 0 ? ry + 2 : (ry < 2 ? ry + 6 : ry - 2);
    const int kx = tx == 0 ? rx + 2 : (rx < 2 ? rx + 6 : rx - 2);
    up[jj] = cvt16<DT>(w4[(((kb * 16 + kk) * 32 + c) * 8 + ky) * 8 + kx]);
  }
}

SRK_DEV void tile_coords(int tile, int tilesX, int tilesY, int& n, int& ty0, int& tx0) {
  const int per = tilesX * tilesY;
  n = tile / per;
  const int r = tile - n * per, ty = r / tilesX;
  ty0 = ty * PJ_TY;
  tx0 = (r - ty * tilesX) * PJ_TX;
}

// ------------------------------------------------------------------------------------------------------------------
// down
// ------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256, 2) void proj_down_kernel(const srk_proj_args a, int tilesX, int tilesY, int ntiles) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  float* red = reinterpret_cast<float*>(smem + D_HALO);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  i32x4 wf[32];
  {
    const i32x4* wp = reinterpret_cast<const i32x4*>(a.wpk) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
    for (int f = 0; f < 32; ++f) wf[f] = wp[f * 64];
  }
  const int px = lane & 31, half = lane >> 5, mx = px & 7, my = px >> 3;
  const char* bbase = halo + (4 * my + 2 * wave) * D_RP + mx * D_QP + half * 16;
  const int HH = 4 * a.H, WH = 4 * a.W;
  const char* xb = reinterpret_cast<const char*>(a.x);
  const size_t xpb = (size_t)a.x_pitch * 2;

  i32x4 st[D_NST];
  auto fetch = [&](int tile) {
    int n, ty0, tx0;
    tile_coords(tile, tilesX, tilesY, n, ty0, tx0);
    const int hy0 = 4 * ty0 - 2, hx0 = 4 * tx0 - 2;
#pragma unroll
    for (int i = 0; i < D_NST; ++i) {
      const int u = tid + 256 * i;
      const int row = u / (D_COLS * 4), rem = u - row * (D_COLS * 4), p = rem >> 2, c = rem & 3;
      const int y = hy0 + row, x = hx0 + p;
      const bool ok = u < D_CHUNKS && y >= 0 && y < HH && x >= 0 && x < WH;
      st[i] = i32x4{0, 0, 0, 0};
      if (ok) st[i] = gload16(xb + ((size_t)(n * HH + y) * WH + x) * xpb + c * 16);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < D_NST; ++i) {
      const int u = tid + 256 * i;
      const int row = u / (D_COLS * 4), rem = u - row * (D_COLS * 4), p = rem >> 2, c = rem & 3;
      if (u < D_CHUNKS) lds_write16(halo + row * D_RP + (p >> 2) * D_QP + (p & 3) * 64 + c * 16, st[i]);
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    stash();
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) fetch(nxt);                    // in flight during this tile's MFMAs
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kyl = 0; kyl < 2; ++kyl)
#pragma unroll
      for (int kx = 0; kx < 8; ++kx)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const i32x4 b = lds_read16(bbase + kyl * D_RP + (kx >> 2) * D_QP + (kx & 3) * 64 + kb * 32);
          acc = Tr::mma(wf[(kyl * 8 + kx) * 2 + kb], b, acc);
        }
    {
      float* rp = red + wave * 1024 + px * 32 + half * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(rp + 4 * j) = f32x4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
    }
    __syncthreads();
    {
      int n, ty0, tx0;
      tile_coords(tile, tilesX, tilesY, n, ty0, tx0);
      const int p = tid >> 3, cg = tid & 7;
      const float* rp = red + p * 32 + cg * 4;
      f32x4 s = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(rp + w * 1024);
        s += t;
      }
      if (a.bias) s += *reinterpret_cast<const f32x4*>(a.bias + cg * 4);
      const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
      if (y < a.H && x < a.W) {
        i32x2 o;
        o.x = (int)pack2<DT>(s.x, s.y);
        o.y = (int)pack2<DT>(s.z, s.w);
        const size_t pix = (size_t)(n * a.H + y) * a.W + x;
        if (a.slope) {         // fused nn.PReLU: the pre-activation is kept for its backward, the activation goes to `out`
          if (a.pre) *reinterpret_cast<i32x2*>(reinterpret_cast<char*>(a.pre) + pix * (size_t)a.pre_pitch * 2 + cg * 8) = o;
          const float* sl = a.slope + cg * 4 * a.slope_stride;
          o.x = (int)prelu_pk<DT>((uint32_t)o.x, sl[0], sl[a.slope_stride]);
          o.y = (int)prelu_pk<DT>((uint32_t)o.y, sl[2 * a.slope_stride], sl[3 * a.slope_stride]);
        }
        *reinterpret_cast<i32x2*>(reinterpret_cast<char*>(a.out) + pix * (size_t)a.out_pitch * 2 + cg * 8) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// up
// ------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256, 2) void proj_up_kernel(const srk_proj_args a, int tilesX, int tilesY, int ntiles) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem + U_CONST;
  const float* cbias = reinterpret_cast<const float*>(smem);
  const float* cslope = cbias + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 32) reinterpret_cast<float*>(smem)[tid] = a.bias ? a.bias[tid] : 0.f;
  else if (tid < 64) reinterpret_cast<float*>(smem)[tid] = a.slope ? a.slope[(tid - 32) * a.slope_stride] : 1.f;
  char* stage = smem + U_HALO + wave * 8192;
  char* stage2 = smem + U_LDS + wave * 8192;             // fused PReLU only: the pre-activations
  i32x4 wf[32];
  {
    const i32x4* wp = reinterpret_cast<const i32x4*>(a.wpk) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
    for (int f = 0; f < 32; ++f) wf[f] = wp[f * 64];
  }
  const int ry = wave, sy = ry < 2 ? -1 : 1;
  const int px = lane & 31, half = lane >> 5, mx = px & 7, my = px >> 3;
  const char* bbase = halo + (my + 1) * U_RP + (mx + 1) * U_PP + half * 16;
  const bool act = a.slope != nullptr;
  const int HH = 4 * a.H, WH = 4 * a.W;
  const char* xb = reinterpret_cast<const char*>(a.x);
  const size_t xpb = (size_t)a.x_pitch * 2, opb = (size_t)a.out_pitch * 2, ppb = (size_t)a.pre_pitch * 2;
  // staging of the (6 x 10)-pixel halo: 240 chunks
  const int hrow = tid / 40, hrem = tid - hrow * 40, hp = hrem >> 2, hc = hrem & 3;
  const int skey = (mx << 1) | (my & 1);                 // XOR key of the stage's 16-byte slots (bank spread of the 32 pixel lanes)

  i32x4 st = i32x4{0, 0, 0, 0};
  auto fetch = [&](int tile) {
    int n, ty0, tx0;
    tile_coords(tile, tilesX, tilesY, n, ty0, tx0);
    const int y = ty0 - 1 + hrow, x = tx0 - 1 + hp;
    st = i32x4{0, 0, 0, 0};
    if (tid < 240 && y >= 0 && y < a.H && x >= 0 && x < a.W) st = gload16(xb + ((size_t)(n * a.H + y) * a.W + x) * xpb + hc * 16);
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    if (tid < 240) lds_write16(halo + hrow * U_RP + hp * U_PP + hc * 16, st);
    __syncthreads();
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) fetch(nxt);
    i32x4 bf[2][3][2];                                   // [row: q, q + sy][column: q - 1, q, q + 1][channel block]
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) bf[ty][dx][kb] = lds_read16(bbase + (ty ? sy * U_RP : 0) + (dx - 1) * U_PP + kb * 32);
    __syncthreads();                                     // the halo is free for the next tile; everything below is wave-private
    int n, ty0, tx0;
    tile_coords(tile, tilesX, tilesY, n, ty0, tx0);
#pragma unroll
    for (int rx = 0; rx < 4; ++rx) {
      const int sx = rx < 2 ? -1 : 1;
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(cbias + 16 * half + 4 * j);
        acc[4 * j] = b4.x; acc[4 * j + 1] = b4.y; acc[4 * j + 2] = b4.z; acc[4 * j + 3] = b4.w;
      }
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) acc = Tr::mma(wf[((rx * 2 + ty) * 2 + tx) * 2 + kb], bf[ty][tx ? 1 + sx : 1][kb], acc);
      // 16 adjacent channels of HR pixel (4 my' + ry, 4 mx + rx): two 16-byte slots of the stage row of LR row my
      i32x4 lo, hi;
      lo.x = (int)pack2<DT>(acc[0], acc[1]);   lo.y = (int)pack2<DT>(acc[2], acc[3]);
      lo.z = (int)pack2<DT>(acc[4], acc[5]);   lo.w = (int)pack2<DT>(acc[6], acc[7]);
      hi.x = (int)pack2<DT>(acc[8], acc[9]);   hi.y = (int)pack2<DT>(acc[10], acc[11]);
      hi.z = (int)pack2<DT>(acc[12], acc[13]); hi.w = (int)pack2<DT>(acc[14], acc[15]);
      const int slot = mx * 16 + rx * 4 + half * 2;
      if (act) {             // fused nn.PReLU on the stored values; the pre-activations leave through the second stage
        lds_write16(stage2 + my * 2048 + (((slot) ^ skey) << 4), lo);
        lds_write16(stage2 + my * 2048 + (((slot + 1) ^ skey) << 4), hi);
        const f32x4* sl = reinterpret_cast<const f32x4*>(cslope + 16 * half);
        const f32x4 s0 = sl[0], s1 = sl[1], s2 = sl[2], s3 = sl[3];
        lo.x = (int)prelu_pk<DT>((uint32_t)lo.x, s0.x, s0.y); lo.y = (int)prelu_pk<DT>((uint32_t)lo.y, s0.z, s0.w);
        lo.z = (int)prelu_pk<DT>((uint32_t)lo.z, s1.x, s1.y); lo.w = (int)prelu_pk<DT>((uint32_t)lo.w, s1.z, s1.w);
        hi.x = (int)prelu_pk<DT>((uint32_t)hi.x, s2.x, s2.y); hi.y = (int)prelu_pk<DT>((uint32_t)hi.y, s2.z, s2.w);
        hi.z = (int)prelu_pk<DT>((uint32_t)hi.z, s3.x, s3.y); hi.w = (int)prelu_pk<DT>((uint32_t)hi.w, s3.z, s3.w);
      }
      lds_write16(stage + my * 2048 + (((slot) ^ skey) << 4), lo);
      lds_write16(stage + my * 2048 + (((slot + 1) ^ skey) << 4), hi);
      __builtin_amdgcn_sched_barrier(0);                 // one phase's accumulators at a time (the weights take 128 registers)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int u = lane + 64 * i, srow = u >> 7, vpos = u & 127;
      const int key = (((vpos >> 4) & 7) << 1) | (srow & 1);
      const int v = vpos ^ key;                          // which (pixel, chunk) of the row segment lives in slot vpos
      const i32x4 d = lds_read16(stage + u * 16);
      i32x4 d2 = d;
      if (act && a.pre) d2 = lds_read16(stage2 + u * 16);
      const int yl = ty0 + srow, xl = tx0 + (v >> 4);
      if (yl < a.H && xl < a.W) {
        const int y = 4 * yl + ry, x = 4 * tx0 + (v >> 2);
        const size_t pix = (size_t)(n * HH + y) * WH + x;
        *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(a.out) + pix * opb + (v & 3) * 16) = d;
        if (act && a.pre) *reinterpret_cast<i32x4*>(reinterpret_cast<char*>(a.pre) + pix * ppb + (v & 3) * 16) = d2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: scratch[slice][ky][kx][ch][cl], then bpart[slice][ky][32].  Tiles of 8 x 8 LR pixels (four 16-pixel K-steps).
// ------------------------------------------------------------------------------------------------------------------
SRK_DEV void gtile_coords(int tile, int tilesX, int tilesY, int& n, int& ty0, int& tx0) {
  const int per = tilesX * tilesY;
  n = tile / per;
  const int r = tile - n * per, ty = r / tilesX;
  ty0 = ty * G_TY;
  tx0 = (r - ty * tilesX) * PJ_TX;
}

template <int DT>
__global__ __launch_bounds__(256, 2) void proj_wgrad_kernel(const srk_proj_wgrad_args a, int tilesX, int tilesY, int ntiles, int nslices) {
  typedef DTraits<DT> Tr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* gs = smem + G_X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a workgroup owns the kernel rows ky and ky + 4: both read the HR rows = ky - 2 (mod 4), so every HR row is fetched by ONE of the
  // four row groups (and 9 rows serve the 8 LR rows of a tile)
  const int ky = blockIdx.x & 3, slice = blockIdx.x >> 2;
  const int per = (ntiles + nslices - 1) / nslices;
  const int t0 = slice * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  const int HH = 4 * a.H, WH = 4 * a.W;
  const char* xb = reinterpret_cast<const char*>(a.xh);
  const char* gb = reinterpret_cast<const char*>(a.g);
  const size_t xpb = (size_t)a.xh_pitch * 2, gpb = (size_t)a.g_pitch * 2;
  // transposing reads: lane l of 16-lane group G gives the address of pixel 4 rd + q (q = (l & 15) >> 2) of LR row 2 s + (G >> 1),
  // channels 16 (G & 1) + 4 p .. + 3 (p = l & 3), and receives channel 16 (G & 1) + (l & 15) of that row's pixels 4 rd .. 4 rd + 3
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int choff = (16 * (G & 1) + 4 * p) * 2;
  const char* abase = xs + (G >> 1) * G_RP + q * G_QP + choff;
  const char* bbase = gs + ((G >> 1) * 8 + q) * 64 + choff;

  // three tiles in flight per workgroup (a tile's MFMAs take ~0.3 us, a load under this traffic ~2 us): register stages 0..2
  i32x4 sx[3][G_NST], sg[3];
  auto fetch = [&](int tile, i32x4 (&fx)[G_NST], i32x4& fg) {
    int n, ty0, tx0;
    gtile_coords(tile, tilesX, tilesY, n, ty0, tx0);
    const int hx0 = 4 * tx0 - 2;
#pragma unroll
    for (int i = 0; i < G_NST; ++i) {
      const int u = tid + 256 * i;                       // 9 rows x 36 pixels x 4 chunks
      const int row = u / 144, rem = u - row * 144, pp = rem >> 2, c = rem & 3;
      const int y = 4 * (ty0 + row) - 2 + ky, x = hx0 + pp;
      fx[i] = i32x4{0, 0, 0, 0};
      if (u < G_CHUNKS && y >= 0 && y < HH && x >= 0 && x < WH) fx[i] = gload16(xb + ((size_t)(n * HH + y) * WH + x) * xpb + c * 16);
    }
    fg = i32x4{0, 0, 0, 0};
    {
      const int pp = tid >> 2, c = tid & 3;              // 64 pixels x 4 chunks
      const int y = ty0 + (pp >> 3), x = tx0 + (pp & 7);
      if (y < a.H && x < a.W) fg = gload16(gb + ((size_t)(n * a.H + y) * a.W + x) * gpb + c * 16);
    }
  };
  f32x16 acc[2][2];                                      // [ky, ky + 4][kx = 2 wave + j]
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][j][r] = 0.f;
  // bias gradient = per-channel sum of the upstream gradient over all its pixels, taken from the tiles in LDS: the LR operand by the
  // ky = 0 workgroups; the HR operand from the rows 4q .. 4q + 3 (kernel rows 2..5: LDS rows 0..7 of the groups ky = 2, 3, LDS rows
  // 1..8 of the groups ky = 0, 1 -- their kernel rows 4, 5), columns 2..33 of the 36 (the tile's own): every HR pixel once
  const bool bsum = a.db != nullptr && (a.bias_side == 1 ? ky == 0 : true);
  const int brow0 = ky < 2 ? 1 : 0;
  const int bc4 = tid & 7, bpg = tid >> 3;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  // one tile: its register stage -> LDS, the tile two ahead -> the stage that was consumed last, MFMAs
  auto step = [&](int tile, i32x4 (&cx)[G_NST], i32x4& cg, i32x4 (&nx)[G_NST], i32x4& ng) {
#pragma unroll
    for (int i = 0; i < G_NST; ++i) {
      const int u = tid + 256 * i;
      const int row = u / 144, rem = u - row * 144, pp = rem >> 2, c = rem & 3;
      if (u < G_CHUNKS) lds_write16(xs + row * G_RP + (pp >> 2) * G_QP + (pp & 3) * 64 + c * 16, cx[i]);
    }
    lds_write16(gs + tid * 16, cg);
    __syncthreads();
    if (tile + 2 < t1) fetch(tile + 2, nx, ng);
#pragma unroll
    for (int s = 0; s < G_TY / 2; ++s) {
      const i32x4 b = tr_read2(bbase + s * 16 * 64, bbase + s * 16 * 64 + 4 * 64);
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kx = 2 * wave + j;
          const char* ap = abase + (s * 2 + d) * G_RP + (kx >> 2) * G_QP + (kx & 3) * 64;
          const i32x4 av = tr_read2(ap, ap + 4 * G_QP);
          acc[d][j] = Tr::mma(av, b, acc[d][j]);
        }
    }
    if (bsum) {
      if (a.bias_side == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[4];
          load4<DT>(reinterpret_cast<const typename Tr::elem*>(gs + (bpg + 32 * h) * 64 + bc4 * 8), v);
#pragma unroll
          for (int e = 0; e < 4; ++e) bs[e] += v[e];
        }
      } else {
        const int pp = 2 + bpg;
#pragma unroll
        for (int row = 0; row < G_TY; ++row) {
          float v[4];
          load4<DT>(reinterpret_cast<const typename Tr::elem*>(xs + (row + brow0) * G_RP + (pp >> 2) * G_QP + (pp & 3) * 64 + bc4 * 8), v);
#pragma unroll
          for (int e = 0; e < 4; ++e) bs[e] += v[e];
        }
      }
    }
    __syncthreads();
  };

  if (t0 < t1) fetch(t0, sx[0], sg[0]);
  if (t0 + 1 < t1) fetch(t0 + 1, sx[1], sg[1]);
  for (int tile = t0; tile < t1; tile += 3) {
    step(tile, sx[0], sg[0], sx[2], sg[2]);
    if (tile + 1 < t1) step(tile + 1, sx[1], sg[1], sx[0], sg[0]);
    if (tile + 2 < t1) step(tile + 2, sx[2], sg[2], sx[1], sg[1]);
  }
  // rows = ch (8 (r / 4) + 4 (lane / 32) + r % 4), column = cl (lane % 32)
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    float* out = a.scratch + ((size_t)(slice * 8 + ky + 4 * d) * 8) * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* o = out + (2 * wave + j) * 1024 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 32] = acc[d][j][r];
    }
  }
  if (a.db) {           // every workgroup writes its (maybe zero) partial: bpart[slice][ky][32] (the ky + 4 rows of the table stay zero)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[bpg * 32 + bc4 * 4 + e] = bs[e];
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int g2 = 0; g2 < 32; ++g2) t += red[g2 * 32 + tid];
      float* bp = a.scratch + (size_t)nslices * 65536 + (size_t)(slice * 8 + ky) * 32 + tid;
      bp[0] = t;
      bp[4 * 32] = 0.f;
    }
  }
}

// dW4[cl][ch][ky][kx] (=, +=) sum over the slices: a block of 1,024 threads owns 256 outputs, thread (part, i) adds slices part, part + 4, ..
// and the four parts meet in LDS (one fixed order: bitwise reproducible).  Block 256: the bias gradient from bpart, likewise.
__global__ __launch_bounds__(1024) void proj_wgrad_finalize_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int nslices, int accumulate,
                                                                  float* __restrict__ db, int db_accumulate) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  if (blockIdx.x == 256) {
    const int c = tid & 31, part = tid >> 5;             // 32 parts
    const float* bp = scratch + (size_t)nslices * 65536 + c;
    float s = 0.f;
    for (int j = part; j < nslices * 8; j += 32) s += bp[j * 32];
    red[tid] = s;
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int k = 0; k < 32; ++k) t += red[k * 32 + tid];
      db[tid] = db_accumulate ? db[tid] + t : t;
    }
    return;
  }
  const int part = tid >> 8, i = blockIdx.x * 256 + (tid & 255);     // ((ky * 8 + kx) * 32 + ch) * 32 + cl
  float s = 0.f;
#pragma unroll 4
  for (int sl = part; sl < nslices; sl += 4) s += scratch[(size_t)sl * 65536 + i];
  red[tid] = s;
  __syncthreads();
  if (tid < 256) {
    const float t = (red[tid] + red[tid + 256]) + (red[tid + 512] + red[tid + 768]);
    const int cl = i & 31, ch = (i >> 5) & 31, kx = (i >> 10) & 7, ky = i >> 13;
    float* o = dw + ((cl * 32 + ch) * 8 + ky) * 8 + kx;
    *o = accumulate ? *o + t : t;
  }
}

int grid_for(int ntiles) {
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const int env = [] { const char* e = srk_dbg_getenv("SRK_PROJ_WGS_PER_CU"); return e ? atoi(e) : 0; }();     // A/B knob
  const int g = (env > 0 ? env : 2) * cus;
  return ntiles < g ? ntiles : g;
}

template <typename K> int set_lds(K k, int bytes, const char* what) {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    srk_set_error("%s: cannot reserve %d bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

int check_proj(const srk_proj_args* a, const char* what, bool up) {
  SRK_CHECK_ARG(a && a->x && a->out && a->wpk, "%s: null pointer", what);
  SRK_CHECK_ARG(a->dtype == SRK_BF16 || a->dtype == SRK_F16, "%s: 16-bit storage only (dtype %d)", what, a->dtype);
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "%s: bad dims N=%d H=%d W=%d", what, a->N, a->H, a->W);
  SRK_CHECK_ARG(a->x_pitch >= 32 && a->x_pitch % 8 == 0 && a->out_pitch >= 32 && a->out_pitch % 8 == 0, "%s: pitches %d / %d (>= 32, multiples of 8)",
                what, a->x_pitch, a->out_pitch);
  SRK_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->out | (uintptr_t)a->wpk) & 15) == 0, "%s: 16-byte alignment", what);
  const long long hr = 16LL * a->N * a->H * a->W;
  SRK_CHECK_ARG(hr < 0x7fffffffLL, "%s: %lld HR pixels", what, hr);
  SRK_CHECK_ARG(!a->pre || (a->slope && a->pre_pitch >= 32 && a->pre_pitch % 8 == 0 && ((uintptr_t)a->pre & 15) == 0), "%s: pre needs the slope and an aligned pitch", what);
  SRK_CHECK_ARG(!a->slope || a->slope_stride == 0 || a->slope_stride == 1, "%s: slope_stride %d", what, a->slope_stride);
  (void)up;
  return 0;
}

}  // namespace

extern "C" long long srk_proj_pack_bytes(void) { return 2LL * 65536 * 2; }

extern "C" int srk_proj_pack(const float* w4, void* wpk, int dtype, srk_stream_t stream) {
  SRK_CHECK_ARG(w4 && wpk && (((uintptr_t)wpk) & 15) == 0, "srk_proj_pack: null / unaligned pointer");
  SRK_CHECK_ARG(dtype == SRK_BF16 || dtype == SRK_F16, "srk_proj_pack: 16-bit storage only (dtype %d)", dtype);
  uint16_t* d = reinterpret_cast<uint16_t*>(wpk);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SRK_BF16) hipLaunchKernelGGL(proj_pack_kernel<SRK_BF16>, dim3(512), dim3(256), 0, st, w4, d, d + 65536);
  else hipLaunchKernelGGL(proj_pack_kernel<SRK_F16>, dim3(512), dim3(256), 0, st, w4, d, d + 65536);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_proj_pack_group(const srk_proj_pack_job* table_dev, int n, int dtype, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && n > 0 && n <= 65535, "srk_proj_pack_group: bad table (%d entries)", n);
  SRK_CHECK_ARG(dtype == SRK_BF16 || dtype == SRK_F16, "srk_proj_pack_group: 16-bit storage only (dtype %d)", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SRK_BF16) hipLaunchKernelGGL(proj_pack_group_kernel<SRK_BF16>, dim3(512, n), dim3(256), 0, st, table_dev);
  else hipLaunchKernelGGL(proj_pack_group_kernel<SRK_F16>, dim3(512, n), dim3(256), 0, st, table_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_proj_down(const srk_proj_args* a, srk_stream_t stream) {
  if (int rc = check_proj(a, "srk_proj_down", false)) return rc;
  static const int attr = [] {
    int r = set_lds(proj_down_kernel<SRK_BF16>, D_LDS, "srk_proj_down");
    return r ? r : set_lds(proj_down_kernel<SRK_F16>, D_LDS, "srk_proj_down");
  }();
  if (attr) return attr;
  const int tilesX = (a->W + PJ_TX - 1) / PJ_TX, tilesY = (a->H + PJ_TY - 1) / PJ_TY;
  const long long nt = (long long)a->N * tilesX * tilesY;
  SRK_CHECK_ARG(nt < 0x7fffffffLL, "srk_proj_down: %lld tiles", nt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(proj_down_kernel<SRK_BF16>, dim3(grid_for((int)nt)), dim3(256), D_LDS, st, *a, tilesX, tilesY, (int)nt);
  else hipLaunchKernelGGL(proj_down_kernel<SRK_F16>, dim3(grid_for((int)nt)), dim3(256), D_LDS, st, *a, tilesX, tilesY, (int)nt);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_proj_up(const srk_proj_args* a, srk_stream_t stream) {
  if (int rc = check_proj(a, "srk_proj_up", true)) return rc;
  static const int attr = [] {
    int r = set_lds(proj_up_kernel<SRK_BF16>, U_LDS + 4 * 8192, "srk_proj_up");
    return r ? r : set_lds(proj_up_kernel<SRK_F16>, U_LDS + 4 * 8192, "srk_proj_up");
  }();
  if (attr) return attr;
  const int tilesX = (a->W + PJ_TX - 1) / PJ_TX, tilesY = (a->H + PJ_TY - 1) / PJ_TY;
  const long long nt = (long long)a->N * tilesX * tilesY;
  SRK_CHECK_ARG(nt < 0x7fffffffLL, "srk_proj_up: %lld tiles", nt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int lds = U_LDS + (a->slope ? 4 * 8192 : 0);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(proj_up_kernel<SRK_BF16>, dim3(grid_for((int)nt)), dim3(256), lds, st, *a, tilesX, tilesY, (int)nt);
  else hipLaunchKernelGGL(proj_up_kernel<SRK_F16>, dim3(grid_for((int)nt)), dim3(256), lds, st, *a, tilesX, tilesY, (int)nt);
  SRK_LAUNCH_CHECK();
  return 0;
}

static int wgrad_slices(long long ntiles) {
  static const int cus = [] { int c = srk_device_cus(); return c > 0 ? c : 256; }();
  static const int env = [] { const char* e = srk_dbg_getenv("SRK_PROJ_WG_SLICES"); return e ? atoi(e) : 0; }();     // A/B knob
  // 4 kernel-row pairs x slices workgroups: 2 per CU when every slice still gets >= 16 tiles, else 1 per CU (the slices' partial sums
  // are 256 KB each: measured at 16 x 48 x 48, 64 slices 26.3 us, 128 slices 30.3; at 256 x 48 x 48, 254 vs 218 us)
  long long s = env > 0 ? env : (ntiles >= 16LL * (cus / 2) ? cus / 2 : cus / 4);
  if (s > ntiles) s = ntiles;
  return s < 1 ? 1 : (int)s;
}

extern "C" long long srk_proj_wgrad_scratch_floats(int N, int H, int W) {
  const long long nt = (long long)N * ((W + PJ_TX - 1) / PJ_TX) * ((H + G_TY - 1) / G_TY);
  return (long long)wgrad_slices(nt) * (65536 + 8 * 32);
}

extern "C" int srk_proj_wgrad(const srk_proj_wgrad_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->xh && a->g && a->scratch && a->dw, "srk_proj_wgrad: null pointer");
  SRK_CHECK_ARG(a->dtype == SRK_BF16 || a->dtype == SRK_F16, "srk_proj_wgrad: 16-bit storage only (dtype %d)", a->dtype);
  SRK_CHECK_ARG(a->db == nullptr || a->bias_side == 1 || a->bias_side == 2, "srk_proj_wgrad: bias_side %d (1: low-resolution operand, 2: high-resolution)", a->bias_side);
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "srk_proj_wgrad: bad dims N=%d H=%d W=%d", a->N, a->H, a->W);
  SRK_CHECK_ARG(a->xh_pitch >= 32 && a->xh_pitch % 8 == 0 && a->g_pitch >= 32 && a->g_pitch % 8 == 0, "srk_proj_wgrad: pitches %d / %d", a->xh_pitch, a->g_pitch);
  SRK_CHECK_ARG((((uintptr_t)a->xh | (uintptr_t)a->g) & 15) == 0, "srk_proj_wgrad: 16-byte alignment");
  SRK_CHECK_ARG(16LL * a->N * a->H * a->W < 0x7fffffffLL, "srk_proj_wgrad: too many pixels");
  const int tilesX = (a->W + PJ_TX - 1) / PJ_TX, tilesY = (a->H + G_TY - 1) / G_TY;
  const long long nt = (long long)a->N * tilesX * tilesY;
  const int ns = wgrad_slices(nt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(proj_wgrad_kernel<SRK_BF16>, dim3(4 * ns), dim3(256), G_LDS, st, *a, tilesX, tilesY, (int)nt, ns);
  else hipLaunchKernelGGL(proj_wgrad_kernel<SRK_F16>, dim3(4 * ns), dim3(256), G_LDS, st, *a, tilesX, tilesY, (int)nt, ns);
  SRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(proj_wgrad_finalize_kernel, dim3(a->db ? 257 : 256), dim3(1024), 0, st, a->scratch, a->dw, ns, a->accumulate, a->db, a->db_accumulate);
  SRK_LAUNCH_CHECK();
  return 0;
}
