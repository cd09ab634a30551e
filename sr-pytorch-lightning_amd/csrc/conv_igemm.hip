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
#include "srk_common.h"

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

template <int DT, int TC, int KS>
__global__ __launch_bounds__((ConvCfg<DT, TC, KS>::NT)) void conv_igemm_kernel(const srk_conv_args a, int tilesX,
                                                                              int tilesY, int ctiles) {
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
#pragma unroll
  for (int cb = 0; cb < C::CB_W; ++cb)
#pragma unroll
    for (int pb = 0; pb < C::PB_W; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[cb][pb][e] = 0.f;

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

  // ---- epilogue ------------------------------------------------------------------------------------
  const int mode = a.out_mode;
  const int rr = a.ps_r > 1 ? a.ps_r : 1;
  const float scale = a.scale;
  const bool relu = a.relu != 0;
#pragma unroll
  for (int pb = 0; pb < C::PB_W; ++pb) {
    const int gy = y0 + pyb[pb], gx = x0 + px;
    if (gy >= H || gx >= W) continue;
#pragma unroll
    for (int cb = 0; cb < C::CB_W; ++cb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = ctile * TC + wco0 + cb * 32 + 8 * i + 4 * h;   // first of 4 consecutive channels
        if (co >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[cb][pb][4 * i + e];
        if (a.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= scale;

        if (mode == SRK_OUT_PLANAR) {
          const int r2 = rr * rr, Cc = a.Cout / r2;
          float* const o = reinterpret_cast<float*>(a.out);
          const float* const rs = reinterpret_cast<const float*>(a.res);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ce = co + e;
            if (ce < a.Cout) {
              const int c = ce / r2, ij = ce - c * r2;
              const int si = ij / rr, sj = ij - si * rr;
              const size_t idx = ((size_t)(n * Cc + c) * (H * rr) + gy * rr + si) * (W * rr) + gx * rr + sj;
              float val = v[e];
              if (rs) val += rs[idx];
              if (a.post_add) val += a.post_add[c];
              o[idx] = val;
            }
          }
        } else {
          size_t pix;   // pixel index in the output tensor
          int c = co;
          if (mode == SRK_OUT_NHWC_PS) {
            const int Cc = a.Cout / (rr * rr);
            const int ij = co / Cc;
            c = co - ij * Cc;
            const int si = ij / rr, sj = ij - si * rr;
            pix = (size_t)(n * H * rr + gy * rr + si) * (W * rr) + gx * rr + sj;
          } else {
            pix = (size_t)(n * H + gy) * W + gx;
          }
          if (a.res) {
            float q[4];
            load4<DT>(reinterpret_cast<const elem*>(a.res) + pix * a.res_pitch + a.res_coff + c, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += q[e];
          }
          if (a.mask && co >= a.mask_from) {
            float q[4];
            load4<DT>(reinterpret_cast<const elem*>(a.mask) + pix * a.mask_pitch + a.mask_coff + c, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = q[e] > 0.f ? v[e] : 0.f;
          }
          store4<DT>(reinterpret_cast<elem*>(a.out) + pix * a.out_pitch + a.out_coff + c, v);
        }
      }
    }
  }
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
                     tilesY, ctiles);
  SRK_LAUNCH_CHECK();
  return 0;
}

template <int DT> int dispatch_tc(const srk_conv_args& a, hipStream_t st) {
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

extern "C" int srk_conv_tile(int Cout) {
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  const int up = (Cout + 127) / 128 * 128;
  return (up - Cout < 64) ? 128 : 64;
}

extern "C" int srk_conv2d(const srk_conv_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->wpk && a->out, "srk_conv2d: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "srk_conv2d: bad dims N=%d H=%d W=%d", a->N, a->H, a->W);
  SRK_CHECK_ARG(a->KH == a->KW && (a->KH == 1 || a->KH == 3), "srk_conv2d: kernel %dx%d not supported (1x1, 3x3)", a->KH, a->KW);
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
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (a->dtype) {
    case SRK_BF16: return dispatch_tc<SRK_BF16>(*a, st);
    case SRK_F16: return dispatch_tc<SRK_F16>(*a, st);
    default: return dispatch_tc<SRK_F32>(*a, st);
  }
}
