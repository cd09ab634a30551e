// The last upsampling stage and the tail conv as ONE linear map (the "collapsed HR stage").
//
// Replaces, on the 16-bit path, the end of EDSR / RCAN / RDN:  UpscaleBlock's last stage + the tail conv
//     V = conv3x3(X; Wu, bu)  (Ci -> 4C at the resolution of X)      models/common.py:112-139 (act = None on this path)
//     U = PixelShuffle(2)(V)  (C channels at twice the resolution)   common.py:133
//     T = conv3x3(U; Wt, bt)  (C -> O, O <= 4: the image)            edsr.py:37-38,49-52, rcan.py tail, rdn.py:85-95
// There is no activation between the two convolutions, so T is a LINEAR function of X: written at the resolution of X with the
// four sub-pixel positions (a, b) of an output pixel as channels k = o*4 + a*2 + b,
//     T[(2y+a, 2x+b)][o] = beff[k] + sum_{ci, f in [-2,2]^2} Weff[k][ci][f] X[(y,x)+f][ci]           (X zero-padded)
//     Weff[(o,a,b)][ci][s+e] = sum_{d,e} sum_c Wt[o][c][d] Wu[(c,i,j)][ci][e],   (i, sy) = ((a+dy) mod 2, floor((a+dy)/2)), same for x
// a 5x5 convolution Ci -> 4 O with 8x fewer multiply-adds than the two layers (4 O Ci 25 against 4 C Ci 9 + 4 O C 9 per pixel of X) and,
// above all, WITHOUT the C-channel tensor at twice the resolution: at EDSR-baseline's batch of 256 that tensor is 1.2 GB, written by
// the upsampler, read by the tail conv, and its gradient written by the tail's data gradient and read twice more -- 55 % of the
// training step for 38 % of its multiply-adds.  The 5x5 convolution itself, its data gradient and its weight
// gradient run on the direct large-kernel kernels (conv_lk.hip); this file holds what is new:
//   srk_hrtail_collapse    (Wt, bt, Wu, bu) -> Weff, beff and the border terms below             [parameter-sized]
//   srk_hrtail_edge_fwd    T -= border terms                                                      [the outermost ring of output pixels]
//   srk_hrtail_edge_bwd_x  dX -= (border terms)^T g                                               [the outermost ring of X]
//   srk_hrtail_edge_bwd_w  the border-restricted correlations of g and X                          [rows / columns 0 and last]
//   srk_hrtail_expand      dWeff, dbeff (+ the border correlations) -> dWt, dbt, dWu, dbu         [parameter-sized]
// Exactness at the border.  The tail conv pads U with ZEROS, the formula above continues U beyond the image (with what the
// upsampler conv would produce there from the zero-padded X).  The difference lives on the outermost ring of output pixels only and is
// itself linear in X: for an output pixel in the top row (y = 0, a = 0) the taps dy = -1 must go, i.e. T -= Ctop with
//     Ctop[(o,b)] = bedge + sum_{ci, fx} Wedge_top[(o,b)][ci][fx] X[(0, x+fx)][ci]       (only fy = 0 reaches inside the image),
// likewise bottom / left / right, and a corner pixel gets the (dy, dx) = (-1,-1)-type tap back that both of its edges removed.  All
// of these are sums of the same products Wt Wu over subsets of the taps d, so `collapse` builds them with the same routine, and the
// chain rule back to (Wt, bt, Wu, bu) is the same routine transposed, fed with  G = (correlation over all pixels) - (over the
// edge's pixels) [+ corner].  tests/collapse_ref.py states all of this in float64 torch and checks it against autograd of the
// two-layer form (also for 1 x 1, 1 x n and 2 x 1 images, where a pixel is on two opposite edges at once).
#include "srk_common.h"

namespace {

// (sub-pixel a in {0,1}, tap d in {-1,0,1}) -> sub-pixel i of the source and the offset s of its pixel
SRK_DEV void sub_of(int a, int d, int& i, int& s) {
  const int ap = a + d;          // -1 .. 2
  i = ap & 1;
  s = (ap - i) >> 1;             // floor(ap / 2)
}

struct Wts {
  const float* wt; const float* bt; const float* wu; const float* bu;
  int O, C, Ci;
};

// sum of one value per lane over an aligned group of 16 lanes (fixed order), returned in every lane of the group
SRK_DEV float sum16(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// One output element per group of 16 lanes; lane `sub` of the group takes the channels c = sub, sub + 16, ... of the contraction
// (one thread per output walked 64 x 9 strided products on its own: 121 us for 28k outputs, all of it load latency).
// sum over the taps d in [dy0, dy1] x [dx0, dx1] of  sum_c Wt[o][c][d] Wu[(c,i,j)][ci][f - s(d)]   (terms whose e = f - s is not a tap drop out)
SRK_DEV float collapse_w(const Wts& p, int sub, int o, int a, int b, int ci, int dy0, int dy1, int dx0, int dx1, int fy, int fx) {
  float acc = 0.f;
  for (int dy = dy0; dy <= dy1; ++dy) {
    int i, sy; sub_of(a, dy, i, sy);
    const int ey = fy - sy;
    if (ey < -1 || ey > 1) continue;
    for (int dx = dx0; dx <= dx1; ++dx) {
      int j, sx; sub_of(b, dx, j, sx);
      const int ex = fx - sx;
      if (ex < -1 || ex > 1) continue;
      const float* wt = p.wt + (size_t)o * p.C * 9 + (dy + 1) * 3 + (dx + 1);
      const float* wu = p.wu + ((size_t)(i * 2 + j) * p.Ci + ci) * 9 + (ey + 1) * 3 + (ex + 1);
      // (same order of additions as `for (c = sub; c < C; c += 16) acc += wt[..] * wu[..]`, but the eight loads of four channels are
      // requested together: with the run-time trip count one pair was in flight at a time, and the kernel was ~36 L2 latencies long)
      for (int c0 = sub; c0 < p.C; c0 += 64) {
        float x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = c0 + 16 * k;
          const bool ok = c < p.C;
          x[k] = ok ? wt[(size_t)c * 9] : 0.f;
          y[k] = ok ? wu[(size_t)c * 4 * p.Ci * 9] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c0 + 16 * k < p.C) acc += x[k] * y[k];
      }
    }
  }
  return sum16(acc);
}
SRK_DEV float collapse_b(const Wts& p, int sub, int o, int a, int b, int dy0, int dy1, int dx0, int dx1) {
  float acc = 0.f;
  if (p.bu) {
    for (int dy = dy0; dy <= dy1; ++dy) {
      int i, sy; sub_of(a, dy, i, sy);
      for (int dx = dx0; dx <= dx1; ++dx) {
        int j, sx; sub_of(b, dx, j, sx);
        const float* wt = p.wt + (size_t)o * p.C * 9 + (dy + 1) * 3 + (dx + 1);
        for (int c = sub; c < p.C; c += 16) acc += wt[(size_t)c * 9] * p.bu[c * 4 + i * 2 + j];
      }
    }
  }
  return sum16(acc);
}

// edge type 0 top (a = 0, dy = -1), 1 bottom (a = 1, dy = +1): kk = o*2 + b;  2 left (b = 0, dx = -1), 3 right (b = 1, dx = +1): kk = o*2 + a
// corner c = a*2 + b: 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right
__global__ __launch_bounds__(256) void hrtail_collapse_kernel(const srk_hrtail_args a) {
  const Wts p{a.wt, a.bt, a.wu, a.bu, a.O, a.C, a.Ci};
  const int O = a.O, Ci = a.Ci;
  const int n_eff = 4 * O * Ci * 25, n_edge = 4 * 2 * O * Ci * 5, n_cor = 4 * O * Ci;
  const int n_b = 4 * O + 4 * 2 * O + 4 * O;
  const int sub = threadIdx.x & 15;
  int t = blockIdx.x * 16 + (threadIdx.x >> 4);          // output element of this 16-lane group (uniform in the group: no divergence inside)
  if (t >= n_eff + n_edge + n_cor + n_b) t = n_eff + n_edge + n_cor + n_b;      // idle groups still take part in the shuffles of their wave
  float v = 0.f;
  float* dst = nullptr;
  if (t < n_eff) {
    const int f = t % 25, ci = (t / 25) % Ci, k = t / (25 * Ci);
    v = collapse_w(p, sub, k >> 2, (k >> 1) & 1, k & 1, ci, -1, 1, -1, 1, f / 5 - 2, f % 5 - 2);
    dst = a.weff + t;
  } else if (t < n_eff + n_edge) {
    const int u = t - n_eff;
    const int tt = u % 5, ci = (u / 5) % Ci, kk = (u / (5 * Ci)) % (2 * O), ty = u / (5 * Ci * 2 * O);
    const int o = kk >> 1, q = kk & 1;
    if (ty == 0) v = collapse_w(p, sub, o, 0, q, ci, -1, -1, -1, 1, 0, tt - 2);
    else if (ty == 1) v = collapse_w(p, sub, o, 1, q, ci, 1, 1, -1, 1, 0, tt - 2);
    else if (ty == 2) v = collapse_w(p, sub, o, q, 0, ci, -1, 1, -1, -1, tt - 2, 0);
    else v = collapse_w(p, sub, o, q, 1, ci, -1, 1, 1, 1, tt - 2, 0);
    dst = a.wedge + u;
  } else if (t < n_eff + n_edge + n_cor) {
    const int u = t - n_eff - n_edge;
    const int ci = u % Ci, o = (u / Ci) % O, c = u / (Ci * O);
    const int ca = c >> 1, cb = c & 1;
    v = collapse_w(p, sub, o, ca, cb, ci, ca ? 1 : -1, ca ? 1 : -1, cb ? 1 : -1, cb ? 1 : -1, 0, 0);
    dst = a.wcor + u;
  } else if (t < n_eff + n_edge + n_cor + n_b) {
    const int w = t - n_eff - n_edge - n_cor;
    if (w < 4 * O) {
      const int k = w;
      v = (a.bt ? a.bt[k >> 2] : 0.f) + collapse_b(p, sub, k >> 2, (k >> 1) & 1, k & 1, -1, 1, -1, 1);
      dst = a.beff + k;
    } else if (w < 4 * O + 8 * O) {
      const int u = w - 4 * O, kk = u % (2 * O), ty = u / (2 * O), o = kk >> 1, q = kk & 1;
      if (ty == 0) v = collapse_b(p, sub, o, 0, q, -1, -1, -1, 1);
      else if (ty == 1) v = collapse_b(p, sub, o, 1, q, 1, 1, -1, 1);
      else if (ty == 2) v = collapse_b(p, sub, o, q, 0, -1, 1, -1, -1);
      else v = collapse_b(p, sub, o, q, 1, -1, 1, 1, 1);
      dst = a.bedge + u;
    } else {
      const int u = w - 12 * O, o = u % O, c = u / O, ca = c >> 1, cb = c & 1;
      v = collapse_b(p, sub, o, ca, cb, ca ? 1 : -1, ca ? 1 : -1, cb ? 1 : -1, cb ? 1 : -1);
      dst = a.bcor + u;
    }
  }
  if (dst && sub == 0) *dst = v;
}

template <int DT> SRK_DEV float ld_act(const void* base, size_t idx) {
  return DTraits<DT>::to_f32(reinterpret_cast<const typename DTraits<DT>::elem*>(base)[idx]);
}

// ---- border terms, forward and data gradient: one workgroup per (image, edge) ----------------------------------------------------
// An edge is a LINE of `len` pixels of X (row 0 / H-1 or column 0 / W-1) and the 2 len output pixels beside it; position s along the
// line, sub-pixel q of the output.  The line of X (fp32, two zero pixels on both ends) and the edge's weights sit in LDS.  Row edges
// (launch 0) and column edges (launch 1) are separate launches because the four corner pixels belong to one of each -- in sequence
// their read-modify-writes cannot meet -- and the column launch also applies the corner terms.  (First form: one thread, then one
// wave per ring pixel, 411 / 242 us forward and 216 us backward at 256 x 96 x 96: ~200k waves of a few dependent round trips each.)
constexpr int MAXO = 4;
SRK_DEV float wave_total(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
struct EdgeGeo {
  int ty, len; bool row;      // edge type, pixels along the line, row edge?
  int fix_lr, fix_hr;         // the fixed coordinate of the line in X / in the output
};
SRK_DEV EdgeGeo edge_geo(int ty, int H, int W) {
  EdgeGeo e;
  e.ty = ty; e.row = ty < 2; e.len = e.row ? W : H;
  e.fix_lr = ty == 0 ? 0 : ty == 1 ? H - 1 : ty == 2 ? 0 : W - 1;
  e.fix_hr = ty == 0 ? 0 : ty == 1 ? 2 * H - 1 : ty == 2 ? 0 : 2 * W - 1;
  return e;
}
// n floats from global memory to LDS by a 256-thread workgroup, EIGHT loads in flight per thread (the plain strided loop keeps one: its
// trip count is a run-time value; 1,920 edge weights were 8 memory latencies in a row at the head of every edge kernel)
SRK_DEV void stage_copy256(float* __restrict__ dst, const float* __restrict__ src, int n) {
  int i = threadIdx.x;
  for (; i + 256 * 7 < n; i += 256 * 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[i + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[i + 256 * k] = v[k];
  }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = i + 256 * k < n ? src[i + 256 * k] : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (i + 256 * k < n) dst[i + 256 * k] = v[k];
}
// stage the line of X: Xl[(s + 2) * pitch + ci], s = -2 .. len + 1 (zeros outside); 16-byte loads (8 channels) where the layout allows
template <int DT>
SRK_DEV void stage_line(const srk_hrtail_args& a, const EdgeGeo& e, int n, float* Xl, int pitch) {
  typedef DTraits<DT> Tr;
  const int Ci = a.Ci;
  if ((Ci & 7) == 0 && (a.x_pitch & 7) == 0) {
    const int nch = Ci >> 3, total = (e.len + 4) * nch;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 4) {          // four 16-byte loads in flight per thread
      i32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const int sp = i / nch, c8 = i - sp * nch, s = sp - 2;
        raw[u] = i32x4{0, 0, 0, 0};
        if (i < total && s >= 0 && s < e.len) {
          const int yy = e.row ? e.fix_lr : s, xx = e.row ? s : e.fix_lr;
          raw[u] = gload16(reinterpret_cast<const typename Tr::elem*>(a.x) + ((size_t)(n * a.H + yy) * a.W + xx) * a.x_pitch + c8 * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        if (i >= total) continue;
        const int sp = i / nch, c8 = i - sp * nch;
        const int w4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) unpack2<DT>((uint32_t)w4[k], v[2 * k], v[2 * k + 1]);     // (16 zero bits unpack to 0.0 in both 16-bit types)
#pragma unroll
        for (int k = 0; k < 8; ++k) Xl[sp * pitch + c8 * 8 + k] = v[k];
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < (e.len + 4) * Ci; i += 256) {
    const int s = i / Ci - 2, ci = i % Ci;
    float v = 0.f;
    if (s >= 0 && s < e.len) {
      const int yy = e.row ? e.fix_lr : s, xx = e.row ? s : e.fix_lr;
      v = ld_act<DT>(a.x, ((size_t)(n * a.H + yy) * a.W + xx) * a.x_pitch + ci);
    }
    Xl[(s + 2) * pitch + ci] = v;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_fwd_kernel(const srk_hrtail_args a, int phase) {
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const int n = blockIdx.x;
  const EdgeGeo e = edge_geo(2 * phase + blockIdx.y, H, W);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int pitch = Ci + 1;                                        // consecutive pixels in consecutive banks
  float* const Xl = reinterpret_cast<float*>(smem_raw);           // [len + 4][Ci + 1]
  float* const Wl = Xl + (size_t)(e.len + 4) * pitch;             // [2 O][Ci][5]
  stage_line<DT>(a, e, n, Xl, pitch);
  stage_copy256(Wl, a.wedge + (size_t)e.ty * 2 * O * Ci * 5, 2 * O * Ci * 5);
  __syncthreads();
  const size_t oplane = (size_t)H2 * W2;
  float* const out = a.out + (size_t)n * O * oplane;
  // (gridDim.z workgroups share an edge: at the reference's batch of 16 there are only 2 N = 32 edges per launch -- every workgroup stages
  // the whole line, a few KB from L2, and takes its slice of the outputs)
  // (gridDim.z workgroups share an edge: at the reference's batch of 16 there are only 2 N = 32 edges per launch -- every workgroup stages
  // the whole line, a few KB from L2, and takes its slice of the outputs.  One OUTPUT per thread: a thread that took all O colours of
  // its pixel read a third fewer operands and ran 2.5x longer -- these loops are bound by their dependent chain, not by LDS traffic.)
  for (int idx = threadIdx.x + 256 * blockIdx.z; idx < 2 * e.len * O; idx += 256 * gridDim.z) {
    const int p = idx % (2 * e.len), o = idx / (2 * e.len);        // consecutive threads: consecutive output pixels of one colour
    const int s = p >> 1, q = p & 1;
    const float* const w = Wl + (size_t)(o * 2 + q) * Ci * 5;
    const float* const x = Xl + (size_t)s * pitch;                 // tap tt reads pixel s + tt - 2 = Xl row s + tt
    float a0 = a.bedge[e.ty * 2 * O + o * 2 + q], a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;      // one chain per tap
    for (int ci = 0; ci < Ci; ++ci) {
      a0 += w[ci * 5 + 0] * x[0 * pitch + ci]; a1 += w[ci * 5 + 1] * x[1 * pitch + ci]; a2 += w[ci * 5 + 2] * x[2 * pitch + ci];
      a3 += w[ci * 5 + 3] * x[3 * pitch + ci]; a4 += w[ci * 5 + 4] * x[4 * pitch + ci];
    }
    float acc = ((a0 + a1) + (a2 + a3)) + a4;
    if (!e.row && (p == 0 || p == 2 * e.len - 1)) {
      // corner output pixel (column launch): give back the tap both of its edges removed
      const int ca = p == 0 ? 0 : 1, cb = e.ty == 2 ? 0 : 1, c = ca * 2 + cb;
      float cc = a.bcor[c * O + o];
      const float* const xc = Xl + (size_t)(s + 2) * pitch;
      for (int ci = 0; ci < Ci; ++ci) cc += a.wcor[((size_t)c * O + o) * Ci + ci] * xc[ci];
      acc -= cc;
    }
    const size_t at = e.row ? (size_t)e.fix_hr * W2 + p : (size_t)p * W2 + e.fix_hr;
    out[(size_t)o * oplane + at] -= acc;
  }
}

// data gradient: dX[line pixel s'][ci] -= sum_{kk, tt} wedge[kk][ci][tt] g[kk][s' - (tt - 2)]  (+ the corner term in the column launch)
template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_bwd_x_kernel(const srk_hrtail_args a, int phase) {
  typedef DTraits<DT> Tr;
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const int n = blockIdx.x;
  // a 1-pixel-high (wide) image: both row (column) edges are the SAME line of dX -- one workgroup takes them one after the other
  const bool shared_line = phase == 0 ? H == 1 : W == 1;
  if (shared_line && blockIdx.y == 1) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const size_t gplane = (size_t)H2 * W2;
  const float* const g = a.g + (size_t)n * O * gplane;
  for (int rep = 0; rep < (shared_line ? 2 : 1); ++rep) {
  const EdgeGeo e = edge_geo(2 * phase + (shared_line ? rep : (int)blockIdx.y), H, W);
  float* const Gl = reinterpret_cast<float*>(smem_raw);           // [2 O][len + 4]: g of sub-pixel row kk along the line, zeros outside
  float* const Wl = Gl + (size_t)2 * O * (e.len + 4);             // [2 O][Ci][5]
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < 2 * O * (e.len + 4); i0 += 256 * 4) {      // four loads in flight per thread
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      const int kk = i / (e.len + 4), s = i % (e.len + 4) - 2, o = kk >> 1, q = kk & 1;
      v[u] = 0.f;
      if (i < 2 * O * (e.len + 4) && s >= 0 && s < e.len)
        v[u] = e.row ? g[(size_t)o * gplane + (size_t)e.fix_hr * W2 + 2 * s + q] : g[(size_t)o * gplane + (size_t)(2 * s + q) * W2 + e.fix_hr];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + 256 * u < 2 * O * (e.len + 4)) Gl[i0 + 256 * u] = v[u];
  }
  stage_copy256(Wl, a.wedge + (size_t)e.ty * 2 * O * Ci * 5, 2 * O * Ci * 5);
  __syncthreads();
  // one thread = 8 channels of one pixel: ONE 16-byte read-modify-write of dX (2-byte scalar updates were 24 dependent global round
  // trips per thread)
  const int nch = (Ci + 7) >> 3;
  const bool vec = (Ci & 7) == 0 && (a.dx_pitch & 7) == 0;
  for (int idx = threadIdx.x + 256 * blockIdx.z; idx < e.len * nch; idx += 256 * gridDim.z) {
    const int c8 = idx % nch, s = idx / nch;
    float corr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) corr[k] = 0.f;
    for (int kk = 0; kk < 2 * O; ++kk) {
      const float* const gl = Gl + (size_t)kk * (e.len + 4) + s + 2;      // the edge pixel whose tap tt reads pixel s: s - (tt - 2)
      float gv[5];
#pragma unroll
      for (int tt = 0; tt < 5; ++tt) gv[tt] = gl[2 - tt];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ci = c8 * 8 + k;
        if (ci < Ci) {
          const float* const w = Wl + ((size_t)kk * Ci + ci) * 5;
#pragma unroll
          for (int tt = 0; tt < 5; ++tt) corr[k] += w[tt] * gv[tt];
        }
      }
    }
    if (!e.row && (s == 0 || s == e.len - 1)) {
      for (int ca = 0; ca < 2; ++ca) {                               // (H == 1: the pixel is the top AND the bottom corner)
        if (!(ca ? s == e.len - 1 : s == 0)) continue;
        const int cb = e.ty == 2 ? 0 : 1, c = ca * 2 + cb;
        const size_t at = (size_t)(ca ? H2 - 1 : 0) * W2 + e.fix_hr;
        for (int o = 0; o < O; ++o) {
          const float gc = g[(size_t)o * gplane + at];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (c8 * 8 + k < Ci) corr[k] -= a.wcor[((size_t)c * O + o) * Ci + c8 * 8 + k] * gc;
        }
      }
    }
    const int yy = e.row ? e.fix_lr : s, xx = e.row ? s : e.fix_lr;
    typename Tr::elem* const d = reinterpret_cast<typename Tr::elem*>(a.dx) + ((size_t)(n * H + yy) * W + xx) * a.dx_pitch + c8 * 8;
    if (vec) {
      const i32x4 raw = *reinterpret_cast<const i32x4*>(d);
      const int w4[4] = {raw.x, raw.y, raw.z, raw.w};
      int o4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f0, f1;
        unpack2<DT>((uint32_t)w4[k], f0, f1);
        o4[k] = (int)pack2<DT>(f0 - corr[2 * k], f1 - corr[2 * k + 1]);
      }
      *reinterpret_cast<i32x4*>(d) = i32x4{o4[0], o4[1], o4[2], o4[3]};
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c8 * 8 + k < Ci) d[k] = Tr::from_f32(Tr::to_f32(d[k]) - corr[k]);
    }
  }
  }
}

// ---- weight-gradient side: correlations of g and X restricted to one edge (types 0..3: 2 O x 5 combos) or one corner (types 4..7:
// O combos).  One workgroup per (type, chunk of EW_CHUNK images): per image the edge's row / column of X (fp32, zero-padded by two
// pixels on both sides) and its 2 O rows of g go to LDS, then thread (ci, combo group) walks them.  Column Ci of a combo row holds
// the sum of g alone (the bias terms; kept once, under the centre tap).  Partial sums per chunk; a second launch adds the chunks in a
// fixed order (reproducible). ------------------------------------------------------------------------------------------------------------
// images per workgroup: srk_hrtail_ew_chunk(N) (1 up to 64 images, then N / 64); the items of an edge are split over gridDim.z workgroups
constexpr int EW_MAXLEN = 512;       // longest edge the LDS staging takes (longer: the direct-from-memory loop)
template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_bwd_w_kernel(const srk_hrtail_args a, int nchunks, int ew_chunk) {
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const int ty = blockIdx.y, chunk = blockIdx.x;
  const int n0 = chunk * ew_chunk, n1 = min(a.N, n0 + ew_chunk);
  const int ncombo = ty < 4 ? 2 * O * 5 : O;
  const size_t gplane = (size_t)H2 * W2;
  float* const dst = a.scratch + ((size_t)ty * nchunks + chunk) * (size_t)(2 * MAXO * 5) * (Ci + 1);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* const Xl = reinterpret_cast<float*>(smem_raw);          // [len + 4][Ci]
  if (ty >= 4) {
    if (blockIdx.z != 0) return;
    const int c = ty - 4, ca = c >> 1, cbb = c & 1;
    const int Py = ca ? H2 - 1 : 0, Px = cbb ? W2 - 1 : 0, yy = ca ? H - 1 : 0, xx = cbb ? W - 1 : 0;
    for (int item = threadIdx.x; item < ncombo * (Ci + 1); item += 256) {
      const int ci = item % (Ci + 1), o = item / (Ci + 1);
      float acc = 0.f;
      for (int n = n0; n < n1; ++n) {
        const float gv = a.g[((size_t)n * O + o) * gplane + (size_t)Py * W2 + Px];
        const float xv = ci < Ci ? ld_act<DT>(a.x, ((size_t)(n * H + yy) * W + xx) * a.x_pitch + ci) : 1.f;
        acc += gv * xv;
      }
      dst[item] = acc;
    }
    return;
  }
  const bool rowedge = ty < 2;
  const int len = rowedge ? W : H;
  float* const Gl = Xl + (size_t)(len + 4) * Ci;                  // [2 O][len]
  // thread -> items: ci = tid % (Ci + 1) would not divide 256; walk items as (combo, ci) pairs with a fixed per-thread set instead
  constexpr int MAXIT = (2 * MAXO * 5 * 65 + 255) / 256;          // items per thread for Ci = 64 (larger Ci: the loop below just runs longer)
  const int nitems = ncombo * (Ci + 1);
  const int ipb = (nitems + gridDim.z - 1) / gridDim.z, item0 = blockIdx.z * ipb, item1 = min(nitems, item0 + ipb);      // this workgroup's items
  float acc[MAXIT];
#pragma unroll
  for (int k = 0; k < MAXIT; ++k) acc[k] = 0.f;
  for (int n = n0; n < n1; ++n) {
    __syncthreads();
    stage_line<DT>(a, edge_geo(ty, H, W), n, Xl, Ci);          // Xl[(s + 2) * Ci + ci], zeros outside (16-byte loads)
    for (int i = threadIdx.x; i < 2 * O * len; i += 256) {
      const int kk = i / len, s = i % len, o = kk >> 1, q = kk & 1;
      const float* const g = a.g + ((size_t)n * O + o) * gplane;
      Gl[i] = rowedge ? g[(size_t)(ty == 0 ? 0 : H2 - 1) * W2 + 2 * s + q] : g[(size_t)(2 * s + q) * W2 + (ty == 2 ? 0 : W2 - 1)];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      const int item = item0 + threadIdx.x + k * 256;
      if (item >= item1) break;
      const int ci = item % (Ci + 1), cb = item / (Ci + 1), tt = cb % 5, kk = cb / 5;
      const float* const gr = Gl + kk * len;
      // four interleaved partial sums (fixed order): the dependent add chain, not the LDS reads, paced the single-sum form
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      int p = 0;
      if (ci < Ci) {
        const float* const xr = Xl + (size_t)tt * Ci + ci;        // X[s + tt - 2] = Xl[(s + tt) * Ci + ci]
        for (; p + 3 < len; p += 4) {
          t0 += gr[p] * xr[(size_t)p * Ci]; t1 += gr[p + 1] * xr[(size_t)(p + 1) * Ci];
          t2 += gr[p + 2] * xr[(size_t)(p + 2) * Ci]; t3 += gr[p + 3] * xr[(size_t)(p + 3) * Ci];
        }
        for (; p < len; ++p) t0 += gr[p] * xr[(size_t)p * Ci];
      } else if (tt == 2) {
        for (; p + 3 < len; p += 4) { t0 += gr[p]; t1 += gr[p + 1]; t2 += gr[p + 2]; t3 += gr[p + 3]; }
        for (; p < len; ++p) t0 += gr[p];
      }
      acc[k] += (t0 + t1) + (t2 + t3);
    }
  }
#pragma unroll
  for (int k = 0; k < MAXIT; ++k) {
    const int item = item0 + threadIdx.x + k * 256;
    if (item < item1) dst[item] = acc[k];
  }
}

// chunk partials -> eedge / e0 / ecor / k0 (fixed order: four interleaved partial sums, then their sum: reproducible)
__global__ __launch_bounds__(256) void hrtail_edge_reduce_kernel(const srk_hrtail_args a, int nchunks) {
  const int O = a.O, Ci = a.Ci;
  const int per = 2 * MAXO * 5 * (Ci + 1);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int n_edge = 4 * 2 * O * 5 * (Ci + 1), n_cor = 4 * O * (Ci + 1);
  if (t >= n_edge + n_cor) return;
  int ty, item;
  if (t < n_edge) { ty = t / (2 * O * 5 * (Ci + 1)); item = t % (2 * O * 5 * (Ci + 1)); }
  else { const int u = t - n_edge; ty = 4 + u / (O * (Ci + 1)); item = u % (O * (Ci + 1)); }
  const float* const src = a.scratch + (size_t)ty * nchunks * per + item;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 3 < nchunks; c += 4) {
    s0 += src[(size_t)c * per]; s1 += src[(size_t)(c + 1) * per]; s2 += src[(size_t)(c + 2) * per]; s3 += src[(size_t)(c + 3) * per];
  }
  for (; c < nchunks; ++c) s0 += src[(size_t)c * per];
  const float s = (s0 + s1) + (s2 + s3);
  const int ci = item % (Ci + 1), cb = item / (Ci + 1);
  if (ty < 4) {
    const int tt = cb % 5, kk = cb / 5;
    if (ci < Ci) a.eedge[(((size_t)ty * 2 * O + kk) * Ci + ci) * 5 + tt] = s;
    else if (tt == 2) a.e0[ty * 2 * O + kk] = s;
  } else {
    if (ci < Ci) a.ecor[((size_t)(ty - 4) * O + cb) * Ci + ci] = s;
    else a.k0[(ty - 4) * O + cb] = s;
  }
}

// ---- the chain rule back to the two layers' parameters ----------------------------------------------------------------------------
struct Gr {
  const float* r; const float* r0; const float* ee; const float* e0; const float* ec; const float* k0;
  int O, Ci;
};
SRK_DEV int row_type(int a, int dy) { return (a == 0 && dy == -1) ? 0 : ((a == 1 && dy == 1) ? 1 : -1); }
SRK_DEV int col_type(int b, int dx) { return (b == 0 && dx == -1) ? 2 : ((b == 1 && dx == 1) ? 3 : -1); }
// correlation of g and X over the output pixels whose tap d stays inside the image: all - row edge - column edge + corner
SRK_DEV float g_w(const Gr& q, int o, int a, int b, int dy, int dx, int ci, int fy, int fx) {
  float v = q.r[(((size_t)(o * 4 + a * 2 + b)) * q.Ci + ci) * 25 + (fy + 2) * 5 + (fx + 2)];
  const int rt = row_type(a, dy), ct = col_type(b, dx);
  if (rt >= 0 && fy == 0) v -= q.ee[(((size_t)rt * 2 * q.O + o * 2 + b) * q.Ci + ci) * 5 + fx + 2];
  if (ct >= 0 && fx == 0) v -= q.ee[(((size_t)ct * 2 * q.O + o * 2 + a) * q.Ci + ci) * 5 + fy + 2];
  if (rt >= 0 && ct >= 0 && fy == 0 && fx == 0) v += q.ec[((size_t)(a * 2 + b) * q.O + o) * q.Ci + ci];
  return v;
}
SRK_DEV float g_b(const Gr& q, int o, int a, int b, int dy, int dx) {
  float v = q.r0[o * 4 + a * 2 + b];
  const int rt = row_type(a, dy), ct = col_type(b, dx);
  if (rt >= 0) v -= q.e0[rt * 2 * q.O + o * 2 + b];
  if (ct >= 0) v -= q.e0[ct * 2 * q.O + o * 2 + a];
  if (rt >= 0 && ct >= 0) v += q.k0[(a * 2 + b) * q.O + o];
  return v;
}

// dWu / dbu / dbt: one thread per element (27 O terms each).  dWt: one WAVE per element, lane = input channel (4 x Ci x 9 terms each:
// as one thread per element this was 240 of the launch's 361 us on 27 waves).
__global__ __launch_bounds__(256) void hrtail_expand_kernel(const srk_hrtail_args a) {
  const Gr q{a.r, a.r0, a.eedge, a.e0, a.ecor, a.k0, a.O, a.Ci};
  const int O = a.O, C = a.C, Ci = a.Ci;
  const int n_wu = 4 * C * Ci * 9, n_bu = 4 * C, n_bt = O;
  const int nb_thread = (n_wu + n_bu + n_bt + 255) / 256;         // blocks of the per-thread part; the rest: 4 dWt elements per block
  if ((int)blockIdx.x >= nb_thread) {
    const int t = ((int)blockIdx.x - nb_thread) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= O * C * 9) return;
    // dWt[o][c][d] = sum_{a,b} ( sum_{ci,e} G(a,b,d)[o][ci][s+e] Wu[(c,i,j)][ci][e]  +  Gb(a,b,d)[o] bu[(c,i,j)] )
    const int d = t % 9, c = (t / 9) % C, o = t / (9 * C);
    const int dy = d / 3 - 1, dx = d % 3 - 1;
    float acc = 0.f;
    for (int pa = 0; pa < 2; ++pa) {
      int i, sy; sub_of(pa, dy, i, sy);
      for (int pb = 0; pb < 2; ++pb) {
        int j, sx; sub_of(pb, dx, j, sx);
        const int cu = c * 4 + i * 2 + j;
        if (a.bu && lane == 0) acc += g_b(q, o, pa, pb, dy, dx) * a.bu[cu];
        for (int ci = lane; ci < Ci; ci += 64)
#pragma unroll
          for (int e = 0; e < 9; ++e)
            acc += g_w(q, o, pa, pb, dy, dx, ci, sy + e / 3 - 1, sx + e % 3 - 1) * a.wu[((size_t)cu * Ci + ci) * 9 + e];
      }
    }
    acc = wave_total(acc);
    if (lane == 0) a.dwt[t] = acc;
    return;
  }
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_wu) {
    // dWu[(c,i,j)][ci][e] = sum over the (a, dy) that land on sub-pixel i, the (b, dx) on j, and o of  Wt[o][c][d] G(a,b,d)[o][ci][s+e]
    const int e = t % 9, ci = (t / 9) % Ci, cu = t / (9 * Ci), c = cu >> 2, i = (cu >> 1) & 1, j = cu & 1;
    const int ey = e / 3 - 1, ex = e % 3 - 1;
    float acc = 0.f;
    for (int pa = 0; pa < 2; ++pa)
      for (int dy = -1; dy <= 1; ++dy) {
        int ii, sy; sub_of(pa, dy, ii, sy);
        if (ii != i) continue;
        for (int pb = 0; pb < 2; ++pb)
          for (int dx = -1; dx <= 1; ++dx) {
            int jj, sx; sub_of(pb, dx, jj, sx);
            if (jj != j) continue;
            for (int o = 0; o < O; ++o)
              acc += a.wt[((size_t)o * C + c) * 9 + (dy + 1) * 3 + dx + 1] * g_w(q, o, pa, pb, dy, dx, ci, sy + ey, sx + ex);
          }
      }
    a.dwu[t] = acc;
    return;
  }
  t -= n_wu;
  if (t < n_bu) {
    if (!a.dbu) return;
    const int cu = t, c = cu >> 2, i = (cu >> 1) & 1, j = cu & 1;
    float acc = 0.f;
    for (int pa = 0; pa < 2; ++pa)
      for (int dy = -1; dy <= 1; ++dy) {
        int ii, sy; sub_of(pa, dy, ii, sy);
        if (ii != i) continue;
        for (int pb = 0; pb < 2; ++pb)
          for (int dx = -1; dx <= 1; ++dx) {
            int jj, sx; sub_of(pb, dx, jj, sx);
            if (jj != j) continue;
            for (int o = 0; o < O; ++o) acc += a.wt[((size_t)o * C + c) * 9 + (dy + 1) * 3 + dx + 1] * g_b(q, o, pa, pb, dy, dx);
          }
      }
    a.dbu[cu] = acc;
    return;
  }
  t -= n_bu;
  if (t < n_bt && a.dbt) a.dbt[t] = a.r0[t * 4] + a.r0[t * 4 + 1] + a.r0[t * 4 + 2] + a.r0[t * 4 + 3];
}

int check_common(const srk_hrtail_args* a, const char* who) {
  SRK_CHECK_ARG(a, "%s: null argument", who);
  SRK_CHECK_ARG(a->O >= 1 && a->O <= MAXO && a->C >= 1 && a->Ci >= 1, "%s: O=%d (1..%d) C=%d Ci=%d", who, a->O, MAXO, a->C, a->Ci);
  return 0;
}
int check_act(const srk_hrtail_args* a, const char* who) {
  SRK_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0, "%s: bad dims N=%d H=%d W=%d", who, a->N, a->H, a->W);
  SRK_CHECK_ARG(a->dtype == SRK_BF16 || a->dtype == SRK_F16, "%s: 16-bit activations only", who);
  SRK_CHECK_ARG((long long)a->N * a->O * 4 * a->H * a->W < (1ll << 31), "%s: output too large", who);
  return 0;
}

}  // namespace

extern "C" int srk_hrtail_collapse(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_collapse")) return rc;
  SRK_CHECK_ARG(a->wt && a->wu && a->weff && a->beff && a->wedge && a->bedge && a->wcor && a->bcor, "srk_hrtail_collapse: null pointer");
  const int total = 4 * a->O * a->Ci * 25 + 4 * 2 * a->O * a->Ci * 5 + 4 * a->O * a->Ci + 16 * a->O;      // outputs: 16 per workgroup
  hipLaunchKernelGGL(hrtail_collapse_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

// workgroups per edge: enough of them to reach ~512 per launch at small batches
static unsigned edge_split(int N) {
  const int z = (512 + 2 * N - 1) / (2 * N);
  return (unsigned)(z < 1 ? 1 : (z > 8 ? 8 : z));
}

template <typename K> static int edge_lds_attr(K kernel, const char* who) {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { srk_set_error("%s: cannot reserve LDS: %s", who, hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int srk_hrtail_edge_fwd(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_fwd")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_fwd")) return rc;
  SRK_CHECK_ARG(a->x && a->out && a->wedge && a->bedge && a->wcor && a->bcor, "srk_hrtail_edge_fwd: null pointer");
  const int len = a->H > a->W ? a->H : a->W;
  const size_t lds = ((size_t)(len + 4) * (a->Ci + 1) + (size_t)2 * a->O * a->Ci * 5) * 4;
  SRK_CHECK_ARG(lds <= 160 * 1024, "srk_hrtail_edge_fwd: edge of %d pixels x %d channels does not fit the LDS", len, a->Ci);
  static const int rc0 = edge_lds_attr(&hrtail_edge_fwd_kernel<SRK_BF16>, "srk_hrtail_edge_fwd");
  static const int rc1 = edge_lds_attr(&hrtail_edge_fwd_kernel<SRK_F16>, "srk_hrtail_edge_fwd");
  if (rc0 || rc1) return rc0 ? rc0 : rc1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int phase = 0; phase < 2; ++phase) {      // row edges, then column edges + corners (see the kernel)
    if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_fwd_kernel<SRK_BF16>, dim3((unsigned)a->N, 2, edge_split(a->N)), dim3(256), lds, st, *a, phase);
    else hipLaunchKernelGGL(hrtail_edge_fwd_kernel<SRK_F16>, dim3((unsigned)a->N, 2, edge_split(a->N)), dim3(256), lds, st, *a, phase);
    SRK_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int srk_hrtail_edge_bwd_x(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_bwd_x")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_bwd_x")) return rc;
  SRK_CHECK_ARG(a->g && a->dx && a->wedge && a->wcor, "srk_hrtail_edge_bwd_x: null pointer");
  const int len = a->H > a->W ? a->H : a->W;
  const size_t lds = ((size_t)2 * a->O * (len + 4) + (size_t)2 * a->O * a->Ci * 5) * 4;
  SRK_CHECK_ARG(lds <= 160 * 1024, "srk_hrtail_edge_bwd_x: edge of %d pixels does not fit the LDS", len);
  static const int rc0 = edge_lds_attr(&hrtail_edge_bwd_x_kernel<SRK_BF16>, "srk_hrtail_edge_bwd_x");
  static const int rc1 = edge_lds_attr(&hrtail_edge_bwd_x_kernel<SRK_F16>, "srk_hrtail_edge_bwd_x");
  if (rc0 || rc1) return rc0 ? rc0 : rc1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int phase = 0; phase < 2; ++phase) {
    if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_bwd_x_kernel<SRK_BF16>, dim3((unsigned)a->N, 2, edge_split(a->N)), dim3(256), lds, st, *a, phase);
    else hipLaunchKernelGGL(hrtail_edge_bwd_x_kernel<SRK_F16>, dim3((unsigned)a->N, 2, edge_split(a->N)), dim3(256), lds, st, *a, phase);
    SRK_LAUNCH_CHECK();
  }
  return 0;
}

static int ew_chunk_of(int N) { return N <= 64 ? 1 : (N + 63) / 64; }

extern "C" long long srk_hrtail_scratch_floats(int N, int Ci) {
  const int ewc = ew_chunk_of(N);
  const long long nchunks = (N + ewc - 1) / ewc;
  return 8 * nchunks * (long long)(2 * MAXO * 5) * (Ci + 1);
}

extern "C" int srk_hrtail_edge_bwd_w(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_bwd_w")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_bwd_w")) return rc;
  SRK_CHECK_ARG(a->x && a->g && a->scratch && a->eedge && a->e0 && a->ecor && a->k0, "srk_hrtail_edge_bwd_w: null pointer");
  const int ewc = ew_chunk_of(a->N);
  const int nchunks = (a->N + ewc - 1) / ewc;
  const unsigned nz = nchunks * 8 >= 512 ? 2 : (nchunks * 8 >= 256 ? 4 : 8);      // workgroups per (chunk, edge): their items are sliced
  const int len = a->H > a->W ? a->H : a->W;
  SRK_CHECK_ARG(a->Ci <= 64 && len <= EW_MAXLEN, "srk_hrtail_edge_bwd_w: Ci=%d (<= 64) edge length %d (<= %d)", a->Ci, len, EW_MAXLEN);
  const size_t lds = ((size_t)(len + 4) * a->Ci + (size_t)2 * a->O * len) * 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&hrtail_edge_bwd_w_kernel<SRK_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&hrtail_edge_bwd_w_kernel<SRK_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr0 != hipSuccess || attr1 != hipSuccess) { srk_set_error("srk_hrtail_edge_bwd_w: cannot reserve LDS"); return (int)(attr0 != hipSuccess ? attr0 : attr1); }
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_bwd_w_kernel<SRK_BF16>, dim3((unsigned)nchunks, 8, nz), dim3(256), lds, st, *a, nchunks, ewc);
  else hipLaunchKernelGGL(hrtail_edge_bwd_w_kernel<SRK_F16>, dim3((unsigned)nchunks, 8, nz), dim3(256), lds, st, *a, nchunks, ewc);
  SRK_LAUNCH_CHECK();
  const int total = (4 * 2 * a->O * 5 + 4 * a->O) * (a->Ci + 1);
  hipLaunchKernelGGL(hrtail_edge_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, nchunks);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_hrtail_expand(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_expand")) return rc;
  SRK_CHECK_ARG(a->wt && a->wu && a->r && a->r0 && a->eedge && a->e0 && a->ecor && a->k0 && a->dwt && a->dwu, "srk_hrtail_expand: null pointer");
  const int per_thread = 4 * a->C * a->Ci * 9 + 4 * a->C + a->O, per_wave = a->O * a->C * 9;
  const int nblocks = (per_thread + 255) / 256 + (per_wave + 3) / 4;
  hipLaunchKernelGGL(hrtail_expand_kernel, dim3((unsigned)nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}
