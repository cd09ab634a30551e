// 1x1 convolution (a [pixels x Cin] x [Cin x Cout] product) onto 64 output channels: RDN's local / global feature fusion
// (models/rdn.py:36-40,104-108: 1x1 convs from 64 k channels down to 64), RCAN-style bottlenecks and their data gradients.  On the
// streaming implicit-GEMM kernel these launches are latency-bound (64-pixel tiles, one K-block staged at a time).
// One phase, no pipeline: a workgroup owns 8 x 16 pixels x 64 output channels; ALL of its operands -- nkb = Cin / 64 pixel
// tiles of 16 KB and weight blocks of 8 KB (<= 6 blocks: 144 KB), the residual / ReLU-mask tile -- arrive by LDS-DMA behind one
// wait, the four waves run their MFMAs (64 channels x 32 pixels each), the result is staged in LDS over the residual tile and
// copied out in whole 128-byte pixels.  Epilogue arithmetic and order are srk_conv2d's: v = acc + bias; relu; * scale; + res; mask.
#include "srk_common.h"

namespace {

struct P1Cfg {
  static constexpr int NT = 256;
  static constexpr int TR = 8, TCOL = 16, PX = TR * TCOL;   // 128 pixels
  static constexpr int X_BYTES = PX * 128;                  // one 64-channel block of the pixel tile: 16,384
  static constexpr int W_BYTES = 8 * 64 * 16;               // one 64-channel block of the weights for 64 rows: 8,192
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

SRK_DEV __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int DT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const srk_conv_args a, int tilesX, int tilesY, int ncob, int PB, unsigned x_bytes, unsigned w_bytes) {
  typedef DTraits<DT> Tr;
  typedef P1Cfg C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nkb = (a.Cin + 63) >> 6;                        // 64-channel K-blocks (the last one may be partial: Cin % 16 == 0)
  char* const Xs = smem;                                    // [PB][128 px][128 B], chunk slot XOR-swizzled by column
  char* const Ws = smem + PB * C::X_BYTES;                  // [PB][8 chunks][64 rows][16 B]
  char* const stage = Ws + PB * C::W_BYTES;                 // residual / output tile [128 px][128 B]
  char* const mbuf = stage + C::X_BYTES;                    // mask tile when there is a residual too

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  int pt = blockIdx.x;
  const int cob = pt % ncob;
  pt /= ncob;
  const int tX = pt % tilesX;
  pt /= tilesX;
  const int tY = pt % tilesY;
  const int n = pt / tilesY;
  const int y0 = tY * C::TR, x0 = tX * C::TCOL;

  // bias first (the vector-memory counter retires in order: a wait for it behind the transfers would wait for them all)
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cob * 64 + 4 * h + cb * 32 + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[cb][4 * i + 0] = b.x; acc[cb][4 * i + 1] = b.y; acc[cb][4 * i + 2] = b.z; acc[cb][4 * i + 3] = b.w;
    }
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));

  // ---- every operand by LDS-DMA: 1 KB pieces = 8 pixels of one row x 128 B (16 per 64-channel block, 4 per wave) -------------
  auto dma_tile = [&](const void* src, int pitch, int coff, int nchan, unsigned bytes, char* buf) {     // nchan: valid channels from coff
    const i32x4 rs = make_rsrc4(src, bytes);
    const unsigned dst = lds_addr_of(buf);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int blk = wave + 4 * k;                          // row blk >> 1, columns (blk & 1) * 8 ..
      const int iy = blk >> 1, ix = (blk & 1) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz(ix);
      const int gy = y0 + iy, gx = x0 + ix;
      const bool ok = gy < H && gx < W && c * Tr::CH < nchan;      // chunks beyond the tensor's channels: zeros
      const unsigned voff = ok ? (unsigned)((((n * H + gy) * W + gx) * pitch + coff + c * Tr::CH) * 2) : 0x80000000u;
      dma16_hidden(rs, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (blk << 10))));
    }
  };
  const i32x4 wrsrc = make_rsrc4(a.wpk, w_bytes);
  const unsigned ws_lds = lds_addr_of(Ws);
  const bool both = a.res && a.mask;
  const int ovalid = a.Cout - cob * 64;                     // stored channels of this block (Cout may be < CoutP)
  const int px = r & 15, prow = 2 * wave + (r >> 4);
  const int g = swz(px);
  const char* const xl = Xs + ((prow * C::TCOL + px) << 7);
  const char* const wl = Ws + ((h * 64 + r) << 4);
  // passes of PB K-blocks (as many as LDS holds; WDSR's 768 input channels are two passes): transfers, one wait, MFMAs
  for (int k0 = 0; k0 < nkb; k0 += PB) {
    const int nb = min(PB, nkb - k0);
    if (k0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous pass's fragments have been read
    for (int kb = 0; kb < nb; ++kb) {
      dma_tile(a.x, a.x_pitch, a.x_coff + (k0 + kb) * 64, a.Cin - (k0 + kb) * 64, x_bytes, Xs + kb * C::X_BYTES);
#pragma unroll
      for (int k = 0; k < 2; ++k) {                          // 8 pieces of 64 rows x 16 B: chunk (k0 + kb) * 8 + piece (beyond KinP: zeros)
        const int piece = wave * 2 + k;
        dma16_hidden(wrsrc, (unsigned)(((((k0 + kb) * 8 + piece) * a.CoutP + cob * 64) << 4) + lane * 16),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(ws_lds + kb * C::W_BYTES + piece * 1024)));
      }
    }
    if (k0 == 0) {
      if (a.res) dma_tile(a.res, a.res_pitch, a.res_coff + cob * 64, ovalid, 0x7fffffffu, stage);
      if (a.mask) dma_tile(a.mask, a.mask_pitch, a.mask_coff + cob * 64, ovalid, 0x7fffffffu, both ? mbuf : stage);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // MFMAs: wave w owns pixel block w (rows 2w, 2w+1) x 64 channels
    for (int kb = 0; kb < nb; ++kb) {
      i32x4 bf[4], af[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf[ks] = lds_read16(xl + kb * C::X_BYTES + (((2 * ks + h) ^ g) << 4));
        af[ks][0] = lds_read16(wl + kb * C::W_BYTES + (((2 * ks) * 64) << 4));
        af[ks][1] = lds_read16(wl + kb * C::W_BYTES + (((2 * ks) * 64 + 32) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc[0] = Tr::mma(af[ks][0], bf[ks], acc[0]);
        acc[1] = Tr::mma(af[ks][1], bf[ks], acc[1]);
      }
    }
  }

  // ---- epilogue: relu, * scale, + res, mask (channels >= mask_from), staged over the residual tile ---------------------------------
  {
    const float sc = a.scale;
    const f32x2 sc2 = {sc, sc};
    const char* const mb = both ? mbuf : stage;
    const bool mask_on = a.mask && cob * 64 + 32 * h + 32 > a.mask_from;
    f32x2 v[16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int d = 0; d < 8; ++d) v[8 * cb + d] = f32x2{acc[cb][2 * d], acc[cb][2 * d + 1]};
    if (a.relu) {
#pragma unroll
      for (int d = 0; d < 16; ++d) v[d] = f32x2{relu_f32(v[d].x), relu_f32(v[d].y)};
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = v[d] * sc2;
    const int po = ((prow * C::TCOL + px) << 7);
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
        const i32x4 q = lds_read16(mb + po + (((4 * h + j) ^ g) << 4));
        const int qw[4] = {q.x, q.y, q.z, q.w};
        const bool on = cob * 64 + 32 * h + 8 * j >= a.mask_from;              // mask_from is a multiple of 16
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
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.out);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + C::NT * k;                         // 1,024 pieces: pixel i >> 3 (row-major 8 x 16), chunk i & 7
      const int p = i >> 3, c = i & 7;
      const int row = p >> 4, col = p & 15;
      const int gy = y0 + row, gx = x0 + col;
      const bool ok = gy < H && gx < W && c * Tr::CH < ovalid;
      const i32x4 q = lds_read16(stage + (p << 7) + ((c ^ swz(col)) << 4));
      const unsigned vo = ok ? (unsigned)((((n * H + gy) * W + gx) * a.out_pitch + a.out_coff + cob * 64 + c * Tr::CH) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w}, ro, vo, 0, 0);
    }
  }
}

}  // namespace

// Whether srk_conv2d takes this kernel for `a`: 16-bit 1x1, Cin >= 64 (a multiple of 16), plain NHWC, the stored channels
// inside the last 64-row block of the packed weights.
// SRK_NO_P1=1 keeps the streaming kernel (A/B runs).
bool srk_conv1x1_ok(const srk_conv_args& a) {
  static const bool off = [] { const char* e = srk_dbg_getenv("SRK_NO_P1"); return e && e[0] == '1'; }();
  if (off || a.dtype == SRK_F32 || a.KH != 1 || a.KW != 1) return false;
  if (a.x_ps > 1 || a.out_mode != SRK_OUT_NHWC || a.post_add) return false;
  if (a.Cin < 64 || a.Cin % 16 != 0 || a.CoutP % 64 != 0 || a.Cout % 8 != 0 || a.Cout > a.CoutP || a.Cout <= a.CoutP - 64) return false;
  // Measured (tools/microbench_conv.py, 16 x 48x48): it wins where ONE 64-channel output block covers the layer (64 -> 64: 5.1 vs
  // 7.1 us, 256 -> 64: 10.2 vs 15.5, 576 -> 64: 18.2 vs 30.9) and loses on wide outputs, where every output block re-reads the
  // pixel tile (128 -> 768: 34.7 vs 28.9 us): those stay on the streaming kernel.
  if (a.CoutP != 64 || a.Cin > 640) return false;
  if (a.x_pitch % 8 || a.x_coff % 8 || a.out_pitch % 8 || a.out_coff % 8) return false;
  if (a.res && (a.res_pitch % 8 || a.res_coff % 8)) return false;
  if (a.mask && (a.mask_pitch % 8 || a.mask_coff % 8 || a.mask_from % 16)) return false;
  const long long px = (long long)a.N * a.H * a.W;
  long long mx = px * a.x_pitch;
  if (px * a.out_pitch > mx) mx = px * a.out_pitch;
  if (a.res && px * a.res_pitch > mx) mx = px * a.res_pitch;
  if (a.mask && px * a.mask_pitch > mx) mx = px * a.mask_pitch;
  return mx * 2 < 0x7fff0000LL;
}

int srk_conv1x1_launch(const srk_conv_args& a, hipStream_t st) {
  typedef P1Cfg C;
  const int nkb = (a.Cin + 63) / 64;
  const int fixed = C::X_BYTES + ((a.res && a.mask) ? C::X_BYTES : 0);                   // output / residual tile (+ mask tile)
  int PB = (160 * 1024 - fixed) / (C::X_BYTES + C::W_BYTES);                                // K-blocks per pass that fit (5 or 6)
  if (PB > nkb) PB = nkb;
  const int lds = PB * (C::X_BYTES + C::W_BYTES) + fixed;
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kernel<SRK_BF16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kernel<SRK_F16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr0 != hipSuccess || attr1 != hipSuccess) {
    srk_set_error("srk_conv2d: cannot reserve LDS for the 1x1 kernel");
    return (int)(attr0 != hipSuccess ? attr0 : attr1);
  }
  SRK_CHECK_ARG(lds <= 160 * 1024, "srk_conv2d: 1x1 kernel needs %d bytes of LDS", lds);
  const int tilesX = (a.W + C::TCOL - 1) / C::TCOL, tilesY = (a.H + C::TR - 1) / C::TR, ncob = a.CoutP / 64;
  const long long nb = (long long)a.N * tilesX * tilesY * ncob;
  SRK_CHECK_ARG(nb <= 0x7fffffffLL, "srk_conv2d: %lld workgroups", nb);
  const unsigned xb = (unsigned)((long long)a.N * a.H * a.W * a.x_pitch * 2);
  const unsigned wb = (unsigned)((long long)((a.Cin + 15) / 16 * 2) * a.CoutP * 16);      // KinP / 8 chunks: later chunks read as zeros
  if (a.dtype == SRK_BF16) hipLaunchKernelGGL((conv1x1_kernel<SRK_BF16>), dim3((unsigned)nb), dim3(C::NT), lds, st, a, tilesX, tilesY, ncob, PB, xb, wb);
  else hipLaunchKernelGGL((conv1x1_kernel<SRK_F16>), dim3((unsigned)nb), dim3(C::NT), lds, st, a, tilesX, tilesY, ncob, PB, xb, wb);
  SRK_LAUNCH_CHECK();
  return 0;
}
