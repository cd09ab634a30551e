// Weight normalisation of every weight-normed convolution of a model in ONE launch per direction.
//
// Replaces nn.utils.weight_norm as the reference applies it to every conv of WDSR (models/wdsr.py:62: `wn = lambda x:
// torch.nn.utils.weight_norm(x)`; head, 16 x 3 block convs, tail, skip = 51 convs):
//     forward : w[o][...] = v[o][...] * (g[o] / ||v[o]||)                       (torch._weight_norm, dim = 0)
//     backward: s = <dw[o], v[o]>;  dg[o] = s / ||v[o]||;  dv[o] = (g[o] / ||v[o]||) * (dw[o] - v[o] * s / ||v[o]||^2)
// torch issues one small kernel per conv and direction (~100 launches of ~5 us per training step, next to ~100 more that re-pack
// the resulting non-leaf weights one by one).  Here a device table lists the tensors; one wave owns one output row.  The
// effective weights are written to buffers with STABLE addresses, so the grouped pack launch (srk_pack_conv_weights_group) and a
// captured hipGraph can name them.
#include "srk_common.h"

namespace {

SRK_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void wn_group_kernel(const srk_wn_job* __restrict__ jobs, int njobs, int total_rows, int backward) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int lo = 0, hi = njobs - 1;                                // last job with row0 <= row
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row0 <= row) lo = mid; else hi = mid - 1;
  }
  const srk_wn_job j = jobs[lo];
  const int r = row - j.row0, n = j.cols;
  const float* const v = j.v + (size_t)r * n;
  if (!backward) {
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += v[i] * v[i];
    const float norm = sqrtf(wave_sum(s));
    const float k = j.g[r] / norm;
    float* const w = j.w + (size_t)r * n;
    for (int i = lane; i < n; i += 64) w[i] = v[i] * k;
    if (lane == 0) j.inv[r] = 1.f / norm;
  } else {
    const float* const dw = j.dw + (size_t)r * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += dw[i] * v[i];
    s = wave_sum(s);
    const float inv = j.inv[r], g = j.g[r];
    const float a = g * inv, b = s * inv * inv;
    float* const dv = j.dv + (size_t)r * n;
    for (int i = lane; i < n; i += 64) dv[i] = a * (dw[i] - v[i] * b);
    if (lane == 0) j.dg[r] = s * inv;
  }
}

}  // namespace

extern "C" int srk_weight_norm_group(const srk_wn_job* table_dev, int njobs, int total_rows, int backward, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && njobs > 0 && total_rows > 0, "srk_weight_norm_group: empty table");
  hipLaunchKernelGGL(wn_group_kernel, dim3((unsigned)((total_rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     table_dev, njobs, total_rows, backward);
  SRK_LAUNCH_CHECK();
  return 0;
}
