// The steps on either side of the convolutional path, on the device:
//   srk_sample_patches  random-crop + rot90 + flips + uint8->float for a whole batch (srdata.py:64-91,137-169)
//   srk_image_sse       clamped squared-error reduction behind PSNR / PSNR-Y (srmodel.py:224-232,582)
// Both are tiny HBM-bound gathers / reductions; images and batches never leave the GPU.
#include "srk_common.h"

namespace {

// output pixel (y, x) of the final S x S patch -> pixel (i, j) of the cropped patch, undoing vflip, hflip and
// the counter-clockwise quarter turns in reverse order:  rot90(P,1)[i][j] = P[j][S-1-i]
__device__ __forceinline__ void src_of(int y, int x, int S, int rot, int hf, int vf, int& i, int& j) {
  if (vf) y = S - 1 - y;
  if (hf) x = S - 1 - x;
  switch (rot & 3) {
    case 0: i = y; j = x; break;
    case 1: i = x; j = S - 1 - y; break;
    case 2: i = S - 1 - y; j = S - 1 - x; break;
    default: i = S - 1 - x; j = y; break;
  }
}

__global__ __launch_bounds__(256) void patch_kernel(const srk_patch_args a) {
  const int n = blockIdx.y;
  const srk_patch_desc d = a.table[n];
  const int pl = a.patch_lr, ph = a.patch_lr * a.scale;
  const int npl = pl * pl, nph = ph * ph;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < npl + nph; t += gridDim.x * blockDim.x) {
    const bool is_hr = t >= npl;
    const int q = is_hr ? t - npl : t;
    const int S = is_hr ? ph : pl;
    const int y = q / S, x = q - y * S;
    int i, j;
    src_of(y, x, S, d.rot, d.hflip, d.vflip, i, j);
    const int top = is_hr ? d.top * a.scale : d.top, left = is_hr ? d.left * a.scale : d.left;
    const int w = is_hr ? d.hr_w : d.lr_w;
    const uint8_t* src = (is_hr ? d.hr : d.lr) + ((size_t)(top + i) * w + (left + j)) * a.C;
    float* dst = (is_hr ? a.hr_out : a.lr_out) + (size_t)n * a.C * S * S + (size_t)y * S + x;
    for (int c = 0; c < a.C; ++c) dst[(size_t)c * S * S] = (float)src[c] / 255.0f;     // TF.to_tensor: x / 255
  }
}

__global__ __launch_bounds__(256) void sse_kernel(const srk_sse_args a) {
  __shared__ double red[4];
  const int n = blockIdx.y;
  const int h = a.H - 2 * a.shave, w = a.W - 2 * a.shave;
  const long long plane = (long long)a.H * a.W;
  const float* sr = a.sr + (size_t)n * a.C * plane;
  const float* hr = a.hr + (size_t)n * a.C * plane;
  double acc = 0.0;
  const long long npix = (long long)h * w;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(p / w) + a.shave, x = (int)(p % w) + a.shave;
    const long long o = (long long)y * a.W + x;
    if (a.luma) {
      float ys = 16.f, yh = 16.f;
      const float k[3] = {65.481f, 128.553f, 24.966f};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ys += k[c] * fminf(fmaxf(sr[c * plane + o], 0.f), 1.f);
        yh += k[c] * fminf(fmaxf(hr[c * plane + o], 0.f), 1.f);
      }
      const double dd = (double)(ys - yh) / 255.0;
      acc += dd * dd;
    } else {
      for (int c = 0; c < a.C; ++c) {
        const float dd = fminf(fmaxf(sr[c * plane + o], 0.f), 1.f) - fminf(fmaxf(hr[c * plane + o], 0.f), 1.f);
        acc += (double)dd * dd;
      }
    }
  }
  // wave reduction (64 lanes), then the 4 waves through LDS, one double atomic per block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(a.sse + n, red[0] + red[1] + red[2] + red[3]);
}

// SSIM map of one 16x16 tile per workgroup: (26 x 26) pooled pixels of both images in LDS, horizontal 11-tap pass for
// the five moments (x, y, xx, yy, xy) on 26 rows x 16 columns, vertical pass + SSIM formula per output pixel, wave /
// LDS reduction, one double atomic per workgroup.  HBM-bound (each input pixel is read ~1.6x), fp32 like piq.
__global__ __launch_bounds__(256) void ssim_kernel(const srk_ssim_args a, int tilesX, int Hp, int Wp) {
  __shared__ float X[26][27], Y[26][27];
  __shared__ float Hm[5][26][16];
  __shared__ float g[11];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int plane = blockIdx.y;
  const int tX = blockIdx.x % tilesX, tY = blockIdx.x / tilesX;
  const int y0 = tY * 16, x0 = tX * 16;
  const float* xs = a.x + (size_t)plane * a.H * a.W;
  const float* ys = a.y + (size_t)plane * a.H * a.W;
  if (tid < 11) {
    const float co = (float)tid - 5.0f;
    g[tid] = expf(-(co * co) / (2.0f * a.sigma * a.sigma));
  }
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) sum += g[i];
    for (int i = 0; i < 11; ++i) g[i] /= sum;
  }
  const int f = a.pool;
  const float inv = 1.0f / (float)(f * f);
  for (int i = tid; i < 26 * 26; i += 256) {
    const int r = i / 26, c = i - r * 26;
    const int py = y0 + r, px = x0 + c;
    float vx = 0.f, vy = 0.f;
    if (py < Hp && px < Wp) {
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
          const size_t o = (size_t)(py * f + dy) * a.W + (px * f + dx);
          vx += xs[o];
          vy += ys[o];
        }
      vx *= inv; vy *= inv;
    }
    X[r][c] = vx; Y[r][c] = vy;
  }
  __syncthreads();
  for (int i = tid; i < 26 * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = X[r][c + k], yv = Y[r][c + k], w = g[k];
      m[0] += w * xv; m[1] += w * yv; m[2] += w * (xv * xv); m[3] += w * (yv * yv); m[4] += w * (xv * yv);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) Hm[q][r][c] = m[q];
  }
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  double acc = 0.0;
  if (y0 + r + 10 < Hp && x0 + c + 10 < Wp) {
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = g[k];
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] += w * Hm[q][r + k][c];
    }
    const float c1 = a.k1 * a.k1, c2 = a.k2 * a.k2;
    const float sxx = m[2] - m[0] * m[0], syy = m[3] - m[1] * m[1], sxy = m[4] - m[0] * m[1];
    const float cs = (2.f * sxy + c2) / (sxx + syy + c2);
    const float ss = (2.f * m[0] * m[1] + c1) / (m[0] * m[0] + m[1] * m[1] + c1) * cs;
    acc = (double)ss;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(a.sums + plane, red[0] + red[1] + red[2] + red[3]);
}

// fused L1 loss.  Forward: grid-stride float4 loads of both tensors, |d| summed per thread in fp32 over at most
// L1_PER_THREAD elements then in double, wave + LDS reduction, one partial per block (no atomics: the caller's sum over
// the partials has a fixed order); sign(d) packed 4 per dword.  Backward: one int32 of signs -> one float4 of gradient.
constexpr int L1_BLOCKS_MAX = 2048;
__global__ __launch_bounds__(256) void l1_fwd_kernel(const srk_l1_args a) {
  __shared__ double red[4];
  const long long n4 = a.n >> 2;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 s = reinterpret_cast<const float4*>(a.sr)[i], h = reinterpret_cast<const float4*>(a.hr)[i];
    const float d[4] = {s.x - h.x, s.y - h.y, s.z - h.z, s.w - h.w};
    unsigned pk = 0;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t += fabsf(d[k]);
      const int sg = (d[k] > 0.f) - (d[k] < 0.f);
      pk |= (unsigned)(sg & 0xff) << (8 * k);
    }
    acc += (double)t;
    reinterpret_cast<unsigned*>(a.sign)[i] = pk;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {            // tail (n not a multiple of 4)
    const long long i = (n4 << 2) + threadIdx.x;
    const float d = a.sr[i] - a.hr[i];
    acc += (double)fabsf(d);
    a.sign[i] = (signed char)((d > 0.f) - (d < 0.f));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// mean = (sum of the partials in index order) / n as ONE small launch (torch: a double reduction, a division and a cast)
__global__ __launch_bounds__(64) void l1_mean_kernel(const double* __restrict__ partial, int nb, long long n, float* __restrict__ out) {
  double t = 0.0;
  // lane-strided, then a fixed shuffle tree: reproducible.  Eight loads in flight per lane, added in index order (one at a time this
  // 64-thread kernel took 8-9 us for 2,048 partial sums: 32 memory latencies in a row)
  int i = threadIdx.x;
  for (; i + 64 * 7 < nb; i += 64 * 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = partial[i + 64 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; i < nb; i += 64) t += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (threadIdx.x == 0) *out = (float)(t / (double)n);
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(const srk_l1_args a) {
  const float g = *a.gout * a.scale;
  const long long n4 = a.n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const unsigned pk = reinterpret_cast<const unsigned*>(a.sign)[i];
    float4 o;
    o.x = g * (float)(signed char)(pk & 0xff);
    o.y = g * (float)(signed char)((pk >> 8) & 0xff);
    o.z = g * (float)(signed char)((pk >> 16) & 0xff);
    o.w = g * (float)(signed char)(pk >> 24);
    reinterpret_cast<float4*>(a.grad)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    a.grad[i] = g * (float)a.sign[i];
  }
}

}  // namespace

extern "C" int srk_l1_blocks(long long n) {
  const long long b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > L1_BLOCKS_MAX ? L1_BLOCKS_MAX : b));
}
extern "C" int srk_l1_loss_fwd(const srk_l1_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->sr && a->hr && a->sign && a->partial && a->n > 0, "srk_l1_loss_fwd: null pointer / empty");
  SRK_CHECK_ARG(((uintptr_t)a->sr | (uintptr_t)a->hr | (uintptr_t)a->sign) % 16 == 0, "srk_l1_loss_fwd: 16-byte alignment");
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(srk_l1_blocks(a->n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}
extern "C" int srk_l1_loss_mean(const double* partial, int nb, long long n, float* out, srk_stream_t stream) {
  SRK_CHECK_ARG(partial && out && nb > 0 && n > 0, "srk_l1_loss_mean: null pointer / empty");
  hipLaunchKernelGGL(l1_mean_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), partial, nb, n, out);
  SRK_LAUNCH_CHECK();
  return 0;
}
extern "C" int srk_l1_loss_bwd(const srk_l1_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->sign && a->gout && a->grad && a->n > 0, "srk_l1_loss_bwd: null pointer / empty");
  SRK_CHECK_ARG(((uintptr_t)a->grad | (uintptr_t)a->sign) % 16 == 0, "srk_l1_loss_bwd: 16-byte alignment");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(srk_l1_blocks(a->n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_image_ssim(const srk_ssim_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->x && a->y && a->sums, "srk_image_ssim: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->C > 0 && (long long)a->N * a->C <= 65535 && a->pool >= 1 && a->sigma > 0.f, "srk_image_ssim: bad sizes");
  const int Hp = a->H / a->pool, Wp = a->W / a->pool;
  SRK_CHECK_ARG(Hp >= 11 && Wp >= 11, "srk_image_ssim: image %dx%d (pooled %dx%d) is smaller than the 11x11 window", a->H, a->W, Hp, Wp);
  const int tilesX = (Wp - 10 + 15) / 16, tilesY = (Hp - 10 + 15) / 16;
  hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)(tilesX * tilesY), (unsigned)(a->N * a->C)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), *a, tilesX, Hp, Wp);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_sample_patches(const srk_patch_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->table && a->lr_out && a->hr_out, "srk_sample_patches: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->N <= 65535 && a->C > 0 && a->C <= 4 && a->patch_lr > 0 && a->scale > 0, "srk_sample_patches: bad sizes");
  const int total = a->patch_lr * a->patch_lr * (1 + a->scale * a->scale);
  int gx = (total + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(patch_kernel, dim3(gx, a->N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_image_sse(const srk_sse_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->sr && a->hr && a->sse, "srk_image_sse: null pointer");
  SRK_CHECK_ARG(a->N > 0 && a->N <= 65535 && a->C > 0 && a->H > 2 * a->shave && a->W > 2 * a->shave && a->shave >= 0, "srk_image_sse: bad sizes");
  SRK_CHECK_ARG(!a->luma || a->C == 3, "srk_image_sse: luma needs 3 channels");
  const long long npix = (long long)(a->H - 2 * a->shave) * (a->W - 2 * a->shave);
  long long gx = (npix + 255) / 256;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(sse_kernel, dim3((unsigned)gx, a->N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  SRK_LAUNCH_CHECK();
  return 0;
}
