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

}  // namespace

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
