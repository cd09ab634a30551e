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

// sum over the taps d in [dy0, dy1] x [dx0, dx1] of  sum_c Wt[o][c][d] Wu[(c,i,j)][ci][f - s(d)]   (terms whose e = f - s is not a tap drop out)
SRK_DEV float collapse_w(const Wts& p, int o, int a, int b, int ci, int dy0, int dy1, int dx0, int dx1, int fy, int fx) {
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
      float s = 0.f;
      for (int c = 0; c < p.C; ++c) s += wt[(size_t)c * 9] * wu[(size_t)c * 4 * p.Ci * 9];
      acc += s;
    }
  }
  return acc;
}
SRK_DEV float collapse_b(const Wts& p, int o, int a, int b, int dy0, int dy1, int dx0, int dx1) {
  if (!p.bu) return 0.f;
  float acc = 0.f;
  for (int dy = dy0; dy <= dy1; ++dy) {
    int i, sy; sub_of(a, dy, i, sy);
    for (int dx = dx0; dx <= dx1; ++dx) {
      int j, sx; sub_of(b, dx, j, sx);
      const float* wt = p.wt + (size_t)o * p.C * 9 + (dy + 1) * 3 + (dx + 1);
      float s = 0.f;
      for (int c = 0; c < p.C; ++c) s += wt[(size_t)c * 9] * p.bu[c * 4 + i * 2 + j];
      acc += s;
    }
  }
  return acc;
}

// edge type 0 top (a = 0, dy = -1), 1 bottom (a = 1, dy = +1): kk = o*2 + b;  2 left (b = 0, dx = -1), 3 right (b = 1, dx = +1): kk = o*2 + a
// corner c = a*2 + b: 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right
__global__ __launch_bounds__(256) void hrtail_collapse_kernel(const srk_hrtail_args a) {
  const Wts p{a.wt, a.bt, a.wu, a.bu, a.O, a.C, a.Ci};
  const int O = a.O, Ci = a.Ci;
  const int n_eff = 4 * O * Ci * 25, n_edge = 4 * 2 * O * Ci * 5, n_cor = 4 * O * Ci;
  const int n_b = 4 * O + 4 * 2 * O + 4 * O;
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_eff) {
    const int f = t % 25, ci = (t / 25) % Ci, k = t / (25 * Ci);
    a.weff[t] = collapse_w(p, k >> 2, (k >> 1) & 1, k & 1, ci, -1, 1, -1, 1, f / 5 - 2, f % 5 - 2);
    return;
  }
  t -= n_eff;
  if (t < n_edge) {
    const int tt = t % 5, ci = (t / 5) % Ci, kk = (t / (5 * Ci)) % (2 * O), ty = t / (5 * Ci * 2 * O);
    const int o = kk >> 1, q = kk & 1;
    float v;
    if (ty == 0) v = collapse_w(p, o, 0, q, ci, -1, -1, -1, 1, 0, tt - 2);
    else if (ty == 1) v = collapse_w(p, o, 1, q, ci, 1, 1, -1, 1, 0, tt - 2);
    else if (ty == 2) v = collapse_w(p, o, q, 0, ci, -1, 1, -1, -1, tt - 2, 0);
    else v = collapse_w(p, o, q, 1, ci, -1, 1, 1, 1, tt - 2, 0);
    a.wedge[t] = v;
    return;
  }
  t -= n_edge;
  if (t < n_cor) {
    const int ci = t % Ci, o = (t / Ci) % O, c = t / (Ci * O);
    const int ca = c >> 1, cb = c & 1;
    a.wcor[t] = collapse_w(p, o, ca, cb, ci, ca ? 1 : -1, ca ? 1 : -1, cb ? 1 : -1, cb ? 1 : -1, 0, 0);
    return;
  }
  t -= n_cor;
  if (t < n_b) {
    if (t < 4 * O) {
      const int k = t;
      a.beff[k] = (a.bt ? a.bt[k >> 2] : 0.f) + collapse_b(p, k >> 2, (k >> 1) & 1, k & 1, -1, 1, -1, 1);
    } else if (t < 4 * O + 8 * O) {
      const int u = t - 4 * O, kk = u % (2 * O), ty = u / (2 * O), o = kk >> 1, q = kk & 1;
      float v;
      if (ty == 0) v = collapse_b(p, o, 0, q, -1, -1, -1, 1);
      else if (ty == 1) v = collapse_b(p, o, 1, q, 1, 1, -1, 1);
      else if (ty == 2) v = collapse_b(p, o, q, 0, -1, 1, -1, -1);
      else v = collapse_b(p, o, q, 1, -1, 1, 1, 1);
      a.bedge[u] = v;
    } else {
      const int u = t - 12 * O, o = u % O, c = u / O, ca = c >> 1, cb = c & 1;
      a.bcor[u] = collapse_b(p, o, ca, cb, ca ? 1 : -1, ca ? 1 : -1, cb ? 1 : -1, cb ? 1 : -1);
    }
  }
}

template <int DT> SRK_DEV float ld_act(const void* base, size_t idx) {
  return DTraits<DT>::to_f32(reinterpret_cast<const typename DTraits<DT>::elem*>(base)[idx]);
}

// ---- forward: one thread per (image, ring pixel of the 2H x 2W output); all O channels -----------------------------------------
constexpr int MAXO = 4;
template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_fwd_kernel(const srk_hrtail_args a, int ring) {
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.N * ring) return;
  const int n = (int)(t / ring);
  int r = (int)(t % ring), Py, Px;
  if (r < W2) { Py = 0; Px = r; }
  else if (r < 2 * W2) { Py = H2 - 1; Px = r - W2; }
  else { r -= 2 * W2; if (r < H2 - 2) { Py = 1 + r; Px = 0; } else { Py = 1 + r - (H2 - 2); Px = W2 - 1; } }
  const int y = Py >> 1, pa = Py & 1, x = Px >> 1, pb = Px & 1;
  const bool top = Py == 0, bot = Py == H2 - 1, left = Px == 0, right = Px == W2 - 1;
  float corr[MAXO];
#pragma unroll
  for (int o = 0; o < MAXO; ++o) corr[o] = 0.f;
  const size_t xn = (size_t)n * H * W;
  for (int e = 0; e < 2; ++e) {                 // e = 0: the row edge of this pixel (if any), e = 1: its column edge
    const int ty = e == 0 ? (top ? 0 : (bot ? 1 : -1)) : (left ? 2 : (right ? 3 : -1));
    if (ty < 0) continue;
    const int q = e == 0 ? pb : pa;             // the free sub-pixel index of kk
    for (int tt = 0; tt < 5; ++tt) {
      const int yy = e == 0 ? y : y + tt - 2, xx = e == 0 ? x + tt - 2 : x;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const size_t px = (xn + (size_t)yy * W + xx) * a.x_pitch;
      for (int ci = 0; ci < Ci; ++ci) {
        const float xv = ld_act<DT>(a.x, px + ci);
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
          if (o < O) corr[o] += a.wedge[(((size_t)ty * 2 * O + o * 2 + q) * Ci + ci) * 5 + tt] * xv;
      }
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
      if (o < O) corr[o] += a.bedge[ty * 2 * O + o * 2 + q];
  }
  if ((top || bot) && (left || right)) {
    const int c = pa * 2 + pb;                  // top-left: (a, b) = (0, 0) ...
    const size_t px = (xn + (size_t)y * W + x) * a.x_pitch;
    for (int ci = 0; ci < Ci; ++ci) {
      const float xv = ld_act<DT>(a.x, px + ci);
#pragma unroll
      for (int o = 0; o < MAXO; ++o)
        if (o < O) corr[o] -= a.wcor[((size_t)c * O + o) * Ci + ci] * xv;
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
      if (o < O) corr[o] -= a.bcor[c * O + o];
  }
#pragma unroll
  for (int o = 0; o < MAXO; ++o)
    if (o < O) a.out[(((size_t)n * O + o) * H2 + Py) * W2 + Px] -= corr[o];
}

// ---- data gradient: one wave per (image, ring pixel of X), lane = input channel (+64 per pass) ---------------------------------
template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_bwd_x_kernel(const srk_hrtail_args a, int ring) {
  typedef DTraits<DT> Tr;
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (long long)a.N * ring) return;
  const int lane = threadIdx.x & 63;
  const int n = (int)(wv / ring);
  int r = (int)(wv % ring), y, x;
  // ring of the H x W grid: row 0, row H-1 (if H > 1), then column 0 and column W-1 (if W > 1) of rows 1 .. H-2
  if (r < W) { y = 0; x = r; }
  else if (H > 1 && r < 2 * W) { y = H - 1; x = r - W; }
  else { r -= (H > 1 ? 2 : 1) * W; if (r < H - 2) { y = 1 + r; x = 0; } else { y = 1 + r - (H - 2); x = W - 1; } }
  const size_t gplane = (size_t)H2 * W2;
  const float* const g = a.g + (size_t)n * O * gplane;
  for (int ci = lane; ci < Ci; ci += 64) {
    float corr = 0.f;
    for (int e = 0; e < 2; ++e) {               // row edges (0: top, 1: bottom), both can hold when H == 1
      if (!(e == 0 ? y == 0 : y == H - 1)) continue;
      const int Py = e == 0 ? 0 : H2 - 1;
      for (int tt = 0; tt < 5; ++tt) {
        const int xs = x - (tt - 2);            // the edge pixel whose tap tt reads this pixel
        if (xs < 0 || xs >= W) continue;
        for (int kk = 0; kk < 2 * O; ++kk)
          corr += a.wedge[(((size_t)e * 2 * O + kk) * Ci + ci) * 5 + tt] * g[(size_t)(kk >> 1) * gplane + (size_t)Py * W2 + 2 * xs + (kk & 1)];
      }
    }
    for (int e = 0; e < 2; ++e) {               // column edges (2: left, 3: right)
      if (!(e == 0 ? x == 0 : x == W - 1)) continue;
      const int Px = e == 0 ? 0 : W2 - 1;
      for (int tt = 0; tt < 5; ++tt) {
        const int ys = y - (tt - 2);
        if (ys < 0 || ys >= H) continue;
        for (int kk = 0; kk < 2 * O; ++kk)
          corr += a.wedge[(((size_t)(2 + e) * 2 * O + kk) * Ci + ci) * 5 + tt] * g[(size_t)(kk >> 1) * gplane + (size_t)(2 * ys + (kk & 1)) * W2 + Px];
      }
    }
    for (int c = 0; c < 4; ++c) {               // corners: this pixel IS the corner pixel (the corner term has the single tap f = 0)
      const int ca = c >> 1, cb = c & 1;
      if ((ca ? y == H - 1 : y == 0) && (cb ? x == W - 1 : x == 0)) {
        const int Py = ca ? H2 - 1 : 0, Px = cb ? W2 - 1 : 0;
        for (int o = 0; o < O; ++o) corr -= a.wcor[((size_t)c * O + o) * Ci + ci] * g[(size_t)o * gplane + (size_t)Py * W2 + Px];
      }
    }
    typename Tr::elem* const d = reinterpret_cast<typename Tr::elem*>(a.dx) + ((size_t)(n * H + y) * W + x) * a.dx_pitch + ci;
    *d = Tr::from_f32(Tr::to_f32(*d) - corr);
  }
}

// ---- weight-gradient side: correlations of g and X restricted to one edge (types 0..3: 2 O x 5 combos) or one corner (types 4..7:
// O combos), summed over a chunk of images by one workgroup; thread = (input channel, combo group) ----------------------------------
constexpr int EW_CHUNK = 1;         // images per workgroup (2048 workgroups at 256 images: the loops are latency-bound)
template <int DT>
__global__ __launch_bounds__(256) void hrtail_edge_bwd_w_kernel(const srk_hrtail_args a, int nchunks) {
  const int H = a.H, W = a.W, O = a.O, Ci = a.Ci, H2 = 2 * H, W2 = 2 * W;
  const int ty = blockIdx.y, chunk = blockIdx.x;
  const int n0 = chunk * EW_CHUNK, n1 = min(a.N, n0 + EW_CHUNK);
  const int ncombo = ty < 4 ? 2 * O * 5 : O;
  const size_t gplane = (size_t)H2 * W2;
  // scratch row of this (type, chunk): [ncombo][Ci + 1] (the last column: the sums of g alone, for the bias terms)
  float* const dst = a.scratch + ((size_t)ty * nchunks + chunk) * (size_t)(2 * MAXO * 5) * (Ci + 1);
  for (int item = threadIdx.x; item < ncombo * (Ci + 1); item += 256) {
    const int ci = item % (Ci + 1), cb = item / (Ci + 1);
    float acc = 0.f;
    if (ty < 4) {
      const int tt = cb % 5, kk = cb / 5, o = kk >> 1, q = kk & 1;
      const bool rowedge = ty < 2;
      const int len = rowedge ? W : H;
      if (ci == Ci && tt != 2) { dst[item] = 0.f; continue; }        // the g-only sums are kept once (under the centre tap)
      for (int n = n0; n < n1; ++n) {
        const float* const g = a.g + ((size_t)n * O + o) * gplane;
        for (int s = 0; s < len; ++s) {
          const int sx = s + tt - 2;
          if (ci < Ci && (sx < 0 || sx >= len)) continue;
          float gv, xv = 1.f;
          if (rowedge) {
            const int Py = ty == 0 ? 0 : H2 - 1, yy = ty == 0 ? 0 : H - 1;
            gv = g[(size_t)Py * W2 + 2 * s + q];
            if (ci < Ci) xv = ld_act<DT>(a.x, ((size_t)(n * H + yy) * W + sx) * a.x_pitch + ci);
          } else {
            const int Px = ty == 2 ? 0 : W2 - 1, xx = ty == 2 ? 0 : W - 1;
            gv = g[(size_t)(2 * s + q) * W2 + Px];
            if (ci < Ci) xv = ld_act<DT>(a.x, ((size_t)(n * H + sx) * W + xx) * a.x_pitch + ci);
          }
          acc += gv * xv;
        }
      }
    } else {
      const int c = ty - 4, ca = c >> 1, cbb = c & 1, o = cb;
      const int Py = ca ? H2 - 1 : 0, Px = cbb ? W2 - 1 : 0, yy = ca ? H - 1 : 0, xx = cbb ? W - 1 : 0;
      for (int n = n0; n < n1; ++n) {
        const float gv = a.g[((size_t)n * O + o) * gplane + (size_t)Py * W2 + Px];
        const float xv = ci < Ci ? ld_act<DT>(a.x, ((size_t)(n * H + yy) * W + xx) * a.x_pitch + ci) : 1.f;
        acc += gv * xv;
      }
    }
    dst[item] = acc;
  }
}

// chunk partials -> eedge / e0 / ecor / k0 (fixed order: reproducible)
__global__ __launch_bounds__(256) void hrtail_edge_reduce_kernel(const srk_hrtail_args a, int nchunks) {
  const int O = a.O, Ci = a.Ci;
  const int per = 2 * MAXO * 5 * (Ci + 1);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int n_edge = 4 * 2 * O * 5 * (Ci + 1), n_cor = 4 * O * (Ci + 1);
  if (t >= n_edge + n_cor) return;
  int ty, item;
  if (t < n_edge) { ty = t / (2 * O * 5 * (Ci + 1)); item = t % (2 * O * 5 * (Ci + 1)); }
  else { const int u = t - n_edge; ty = 4 + u / (O * (Ci + 1)); item = u % (O * (Ci + 1)); }
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += a.scratch[((size_t)ty * nchunks + c) * per + item];
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

__global__ __launch_bounds__(256) void hrtail_expand_kernel(const srk_hrtail_args a) {
  const Gr q{a.r, a.r0, a.eedge, a.e0, a.ecor, a.k0, a.O, a.Ci};
  const int O = a.O, C = a.C, Ci = a.Ci;
  const int n_wu = 4 * C * Ci * 9, n_bu = 4 * C, n_wt = O * C * 9, n_bt = O;
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
  if (t < n_wt) {
    // dWt[o][c][d] = sum_{a,b} ( sum_{ci,e} G(a,b,d)[o][ci][s+e] Wu[(c,i,j)][ci][e]  +  Gb(a,b,d)[o] bu[(c,i,j)] )
    const int d = t % 9, c = (t / 9) % C, o = t / (9 * C);
    const int dy = d / 3 - 1, dx = d % 3 - 1;
    float acc = 0.f;
    for (int pa = 0; pa < 2; ++pa) {
      int i, sy; sub_of(pa, dy, i, sy);
      for (int pb = 0; pb < 2; ++pb) {
        int j, sx; sub_of(pb, dx, j, sx);
        const int cu = c * 4 + i * 2 + j;
        if (a.bu) acc += g_b(q, o, pa, pb, dy, dx) * a.bu[cu];
        for (int ci = 0; ci < Ci; ++ci)
          for (int e = 0; e < 9; ++e)
            acc += g_w(q, o, pa, pb, dy, dx, ci, sy + e / 3 - 1, sx + e % 3 - 1) * a.wu[((size_t)cu * Ci + ci) * 9 + e];
      }
    }
    a.dwt[t] = acc;
    return;
  }
  t -= n_wt;
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
int out_ring(int H, int W) { return 2 * (2 * W) + 2 * (2 * H - 2); }
int in_ring(int H, int W) { return (H > 1 ? 2 : 1) * W + (W > 1 ? 2 : 1) * (H > 2 ? H - 2 : 0); }

}  // namespace

extern "C" int srk_hrtail_collapse(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_collapse")) return rc;
  SRK_CHECK_ARG(a->wt && a->wu && a->weff && a->beff && a->wedge && a->bedge && a->wcor && a->bcor, "srk_hrtail_collapse: null pointer");
  const int total = 4 * a->O * a->Ci * 25 + 4 * 2 * a->O * a->Ci * 5 + 4 * a->O * a->Ci + 16 * a->O;
  hipLaunchKernelGGL(hrtail_collapse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_hrtail_edge_fwd(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_fwd")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_fwd")) return rc;
  SRK_CHECK_ARG(a->x && a->out && a->wedge && a->bedge && a->wcor && a->bcor, "srk_hrtail_edge_fwd: null pointer");
  const int ring = out_ring(a->H, a->W);
  const long long total = (long long)a->N * ring;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_fwd_kernel<SRK_BF16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, ring);
  else hipLaunchKernelGGL(hrtail_edge_fwd_kernel<SRK_F16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, ring);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_hrtail_edge_bwd_x(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_bwd_x")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_bwd_x")) return rc;
  SRK_CHECK_ARG(a->g && a->dx && a->wedge && a->wcor, "srk_hrtail_edge_bwd_x: null pointer");
  const int ring = in_ring(a->H, a->W);
  const long long waves = (long long)a->N * ring;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_bwd_x_kernel<SRK_BF16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, *a, ring);
  else hipLaunchKernelGGL(hrtail_edge_bwd_x_kernel<SRK_F16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, *a, ring);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" long long srk_hrtail_scratch_floats(int N, int Ci) {
  const long long nchunks = (N + EW_CHUNK - 1) / EW_CHUNK;
  return 8 * nchunks * (long long)(2 * MAXO * 5) * (Ci + 1);
}

extern "C" int srk_hrtail_edge_bwd_w(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_edge_bwd_w")) return rc;
  if (int rc = check_act(a, "srk_hrtail_edge_bwd_w")) return rc;
  SRK_CHECK_ARG(a->x && a->g && a->scratch && a->eedge && a->e0 && a->ecor && a->k0, "srk_hrtail_edge_bwd_w: null pointer");
  const int nchunks = (a->N + EW_CHUNK - 1) / EW_CHUNK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SRK_BF16) hipLaunchKernelGGL(hrtail_edge_bwd_w_kernel<SRK_BF16>, dim3((unsigned)nchunks, 8), dim3(256), 0, st, *a, nchunks);
  else hipLaunchKernelGGL(hrtail_edge_bwd_w_kernel<SRK_F16>, dim3((unsigned)nchunks, 8), dim3(256), 0, st, *a, nchunks);
  SRK_LAUNCH_CHECK();
  const int total = (4 * 2 * a->O * 5 + 4 * a->O) * (a->Ci + 1);
  hipLaunchKernelGGL(hrtail_edge_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a, nchunks);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_hrtail_expand(const srk_hrtail_args* a, srk_stream_t stream) {
  if (int rc = check_common(a, "srk_hrtail_expand")) return rc;
  SRK_CHECK_ARG(a->wt && a->wu && a->r && a->r0 && a->eedge && a->e0 && a->ecor && a->k0 && a->dwt && a->dwu, "srk_hrtail_expand: null pointer");
  const int total = 4 * a->C * a->Ci * 9 + 4 * a->C + a->O * a->C * 9 + a->O;
  hipLaunchKernelGGL(hrtail_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}
