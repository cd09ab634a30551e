// RCAN channel attention (models/rcan.py:10-29) fused with RCAB's residual add (rcan.py:52-54), NHWC.
//
//   srk_ca_pool      sums[n][c] += sum_{hw} t (or t*u): every thread owns one 16-byte channel chunk and a
//                    strided set of pixels; partials meet in LDS, one fp32 atomic per (block, channel).
//   srk_ca_apply     each block recomputes the tiny squeeze/excite MLP of its sample from `sums`
//                    (z = relu(W1 mean + b1), s = sigmoid(W2 z + b2)) and streams out = t*s + res.
//   srk_ca_bwd_apply backward of the MLP per sample (+ fp32 atomics for dW1,db1,dW2,db2) and
//                    gt = g*s + dmean/HW streamed.
// All three are HBM-bound streaming kernels: 16-byte loads/stores, grid = N x splits.
#include "srk_common.h"

namespace {

constexpr int CA_NT = 256;
constexpr int CA_MAXC = 256;
constexpr int CA_MAXCR = 32;

template <int DT> SRK_DEV void chunk_to_f32(i32x4 raw, float* v) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    const uint32_t w[4] = {(uint32_t)raw.x, (uint32_t)raw.y, (uint32_t)raw.z, (uint32_t)raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = Tr::to_f32((uint16_t)(w[i] & 0xffff));
      v[2 * i + 1] = Tr::to_f32((uint16_t)(w[i] >> 16));
    }
  } else {
    v[0] = __int_as_float(raw.x); v[1] = __int_as_float(raw.y); v[2] = __int_as_float(raw.z); v[3] = __int_as_float(raw.w);
  }
}

template <int DT> SRK_DEV i32x4 f32_to_chunk(const float* v) {
  typedef DTraits<DT> Tr;
  if constexpr (Tr::IS16) {
    i32x4 o;
    o.x = (int)((uint32_t)Tr::from_f32(v[0]) | ((uint32_t)Tr::from_f32(v[1]) << 16));
    o.y = (int)((uint32_t)Tr::from_f32(v[2]) | ((uint32_t)Tr::from_f32(v[3]) << 16));
    o.z = (int)((uint32_t)Tr::from_f32(v[4]) | ((uint32_t)Tr::from_f32(v[5]) << 16));
    o.w = (int)((uint32_t)Tr::from_f32(v[6]) | ((uint32_t)Tr::from_f32(v[7]) << 16));
    return o;
  } else {
    return i32x4{__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
  }
}

template <int DT> __global__ __launch_bounds__(CA_NT) void ca_pool_kernel(const srk_ca_pool_args a, int pix_per_block) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  __shared__ float red[CA_NT * 8];
  const int n = blockIdx.x, split = blockIdx.y;
  const int nch = a.C / CH;
  const int rows = CA_NT / nch;
  const int tid = threadIdx.x;
  const int cc = tid % nch, prow = tid / nch;
  const int p0 = split * pix_per_block, p1 = min(a.HW, p0 + pix_per_block);
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
  if (prow < rows) {
    const elem* t = reinterpret_cast<const elem*>(a.t) + (size_t)n * a.HW * a.t_pitch + a.t_coff + cc * CH;
    const elem* u = a.u ? reinterpret_cast<const elem*>(a.u) + (size_t)n * a.HW * a.u_pitch + a.u_coff + cc * CH : nullptr;
    for (int p = p0 + prow; p < p1; p += rows) {
      float tv[CH];
      chunk_to_f32<DT>(gload16(t + (size_t)p * a.t_pitch), tv);
      if (u) {
        float uv[CH];
        chunk_to_f32<DT>(gload16(u + (size_t)p * a.u_pitch), uv);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += tv[e] * uv[e];
      } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += tv[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) red[tid * CH + e] = (prow < rows) ? acc[e] : 0.f;
  __syncthreads();
  // thread c (< C) sums its channel over the pixel rows
  if (tid < a.C) {
    const int c_cc = tid / CH, c_e = tid % CH;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[(r * nch + c_cc) * CH + c_e];
    a.sums[((size_t)n * gridDim.y + split) * a.C + tid] = s;       // this block's partial: no atomics, nothing to zero
  }
}

// out[c] = scale * sum of the per-block partials of sample n, channel c (c < C), computed by the whole workgroup:
// thread (q, c), q = tid / C, adds blocks q, q + Q, ... (loads independent: all in flight together), the Q strided sums
// meet in LDS in a fixed order (bitwise reproducible).  Ends with a barrier; `red` holds CA_NT floats.
SRK_DEV void ca_sum_partials(const float* __restrict__ sums, int n, int nsplit, int C, float scale, float* red, float* out) {
  const int tid = threadIdx.x;
  const int Q = CA_NT / C;                       // C <= CA_MAXC = CA_NT
  const int q = tid / C, c = tid - q * C;
  float t = 0.f;
  if (q < Q) {
    const float* p = sums + (size_t)n * nsplit * C + c;
#pragma unroll 4
    for (int sp = q; sp < nsplit; sp += Q) t += p[(size_t)sp * C];
  }
  red[tid] = t;
  __syncthreads();
  if (tid < C) {
    float u = 0.f;
    for (int k = 0; k < Q; ++k) u += red[k * C + tid];
    out[tid] = u * scale;
  }
  __syncthreads();
}

template <int DT> __global__ __launch_bounds__(CA_NT) void ca_apply_kernel(const srk_ca_apply_args a, int pix_per_block) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  __shared__ float mean[CA_MAXC], sv[CA_MAXC], zv[CA_MAXCR], red[CA_NT];
  const int n = blockIdx.x, split = blockIdx.y, tid = threadIdx.x;
  const int C = a.C, Cr = a.Cr;
  ca_sum_partials(a.sums, n, a.sums_rows > 0 ? a.sums_rows : (int)gridDim.y, C, 1.f / (float)a.HW, red, mean);
  if (C == 64) {
    // 64 channels (RCAN): lane = channel, the 64-term sum as a wave butterfly (srk_common.h wave_sum64: the order the conv-pair
    // launches use too), z = relu(b1 + sum)
    if (tid < 64) {
      const float m = mean[tid];
      for (int j = 0; j < Cr; ++j) {
        const float z = relu_f32(a.b1[j] + wave_sum64(a.w1[j * 64 + tid] * m));
        if (tid == 0) {
          zv[j] = z;
          if (a.z_out && split == 0) a.z_out[(size_t)n * Cr + j] = z;
        }
      }
    }
  } else if (tid < Cr) {
    float z = a.b1[tid];
    for (int c = 0; c < C; ++c) z += a.w1[tid * C + c] * mean[c];
    z = relu_f32(z);
    zv[tid] = z;
    if (a.z_out && split == 0) a.z_out[(size_t)n * Cr + tid] = z;
  }
  __syncthreads();
  if (tid < C) {
    float s = a.b2[tid];
    for (int j = 0; j < Cr; ++j) s += a.w2[tid * Cr + j] * zv[j];
    s = 1.f / (1.f + expf(-s));
    sv[tid] = s;
    if (a.s_out && split == 0) a.s_out[(size_t)n * C + tid] = s;
  }
  __syncthreads();
  const int nch = C / CH, rows = CA_NT / nch;
  const int cc = tid % nch, prow = tid / nch;
  if (prow >= rows) return;
  const int p0 = split * pix_per_block, p1 = min(a.HW, p0 + pix_per_block);
  const elem* t = reinterpret_cast<const elem*>(a.t) + (size_t)n * a.HW * a.t_pitch + a.t_coff + cc * CH;
  const elem* rs = a.res ? reinterpret_cast<const elem*>(a.res) + (size_t)n * a.HW * a.res_pitch + a.res_coff + cc * CH : nullptr;
  elem* o = reinterpret_cast<elem*>(a.out) + (size_t)n * a.HW * a.out_pitch + a.out_coff + cc * CH;
  float sc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) sc[e] = sv[cc * CH + e];
  for (int p = p0 + prow; p < p1; p += rows) {
    float tv[CH];
    chunk_to_f32<DT>(gload16(t + (size_t)p * a.t_pitch), tv);
#pragma unroll
    for (int e = 0; e < CH; ++e) tv[e] *= sc[e];
    if (rs) {
      float rv[CH];
      chunk_to_f32<DT>(gload16(rs + (size_t)p * a.res_pitch), rv);
#pragma unroll
      for (int e = 0; e < CH; ++e) tv[e] += rv[e];
    }
    *reinterpret_cast<i32x4*>(o + (size_t)p * a.out_pitch) = f32_to_chunk<DT>(tv);
  }
}

template <int DT> __global__ __launch_bounds__(CA_NT) void ca_bwd_kernel(const srk_ca_bwd_args a, int pix_per_block) {
  typedef DTraits<DT> Tr;
  typedef typename Tr::elem elem;
  constexpr int CH = Tr::CH;
  __shared__ float dpre2[CA_MAXC], dmean[CA_MAXC], sv[CA_MAXC], mean[CA_MAXC], dpre1[CA_MAXCR], zv[CA_MAXCR], red[CA_NT];
  const int n = blockIdx.x, split = blockIdx.y, tid = threadIdx.x;
  const int C = a.C, Cr = a.Cr;
  ca_sum_partials(a.gsum, n, a.gsum_rows > 0 ? a.gsum_rows : (int)gridDim.y, C, 1.f, red, dpre2);
  if (split == 0) ca_sum_partials(a.sums, n, a.sums_rows > 0 ? a.sums_rows : (int)gridDim.y, C, 1.f / (float)a.HW, red, mean);        // block-uniform branch
  if (tid < C) {
    const float s = a.s[(size_t)n * C + tid];
    sv[tid] = s;
    dpre2[tid] *= s * (1.f - s);
  }
  if (tid < Cr) zv[tid] = a.z[(size_t)n * Cr + tid];
  __syncthreads();
  if (C == 64) {
    if (tid < 64) {
      const float d2 = dpre2[tid];
      for (int j = 0; j < Cr; ++j) {
        const float dz = wave_sum64(a.w2[tid * Cr + j] * d2);
        if (tid == 0) dpre1[j] = zv[j] > 0.f ? dz : 0.f;
      }
    }
  } else if (tid < Cr) {
    float dz = 0.f;
    for (int c = 0; c < C; ++c) dz += a.w2[c * Cr + tid] * dpre2[c];
    dpre1[tid] = zv[tid] > 0.f ? dz : 0.f;
  }
  __syncthreads();
  if (tid < C) {
    float dm = 0.f;
    for (int j = 0; j < Cr; ++j) dm += a.w1[j * C + tid] * dpre1[j];
    dmean[tid] = dm / (float)a.HW;
  }
  __syncthreads();
  if (split == 0) {
    // this SAMPLE's contribution to the parameter gradients of the two 1x1 convs, stored to the sample's own slot
    // (stride 2*C*Cr + Cr + C floats); the caller sums the slots over n: no atomics, nothing to zero, reproducible
    const size_t so = (size_t)n * (2 * C * Cr + Cr + C);
    for (int i = tid; i < C * Cr; i += CA_NT) {
      const int c = i / Cr, j = i % Cr;
      a.dw2[so + i] = dpre2[c] * zv[j];                                              // dW2[c][j]
      const int j1 = i / C, c1 = i % C;
      a.dw1[so + i] = dpre1[j1] * mean[c1];                                          // dW1[j][c]
    }
    if (tid < C) a.db2[so + tid] = dpre2[tid];
    if (tid < Cr) a.db1[so + tid] = dpre1[tid];
  }
  const int nch = C / CH, rows = CA_NT / nch;
  const int cc = tid % nch, prow = tid / nch;
  if (prow >= rows) return;
  const int p0 = split * pix_per_block, p1 = min(a.HW, p0 + pix_per_block);
  const elem* g = reinterpret_cast<const elem*>(a.g) + (size_t)n * a.HW * a.g_pitch + a.g_coff + cc * CH;
  elem* o = reinterpret_cast<elem*>(a.gt) + (size_t)n * a.HW * a.gt_pitch + a.gt_coff + cc * CH;
  float sc[CH], dm[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { sc[e] = sv[cc * CH + e]; dm[e] = dmean[cc * CH + e]; }
  for (int p = p0 + prow; p < p1; p += rows) {
    float gv[CH];
    chunk_to_f32<DT>(gload16(g + (size_t)p * a.g_pitch), gv);
#pragma unroll
    for (int e = 0; e < CH; ++e) gv[e] = gv[e] * sc[e] + dm[e];
    *reinterpret_cast<i32x4*>(o + (size_t)p * a.gt_pitch) = f32_to_chunk<DT>(gv);
  }
}

// dst[i] = sum over rows r < n of src[r*k + i], for every job of a device table in one launch (blockIdx.y = job): the sums
// over the batch of the per-sample parameter-gradient slots of all channel-attention layers of a backward pass
__global__ __launch_bounds__(CA_NT) void rowsum_group_kernel(const srk_rowsum_job* __restrict__ table) {
  const srk_rowsum_job j = table[blockIdx.y];
  for (int i = blockIdx.x * CA_NT + threadIdx.x; i < j.k; i += gridDim.x * CA_NT) {
    float u = 0.f;
#pragma unroll 8
    for (int r = 0; r < j.n; ++r) u += j.src[(size_t)r * j.k + i];
    j.dst[i] = u;
  }
}

inline void split_for(int N, int HW, int* splits, int* ppb) {
  // aim at >= ~1024 blocks but at least 64 pixels per block
  int s = (1024 + N - 1) / N;
  int maxs = (HW + 63) / 64;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  *ppb = (HW + s - 1) / s;
  *splits = (HW + *ppb - 1) / *ppb;
}

int check_c(int C, int Cr, int dtype, const char* who) {
  const int ch = dtype == SRK_F32 ? 4 : 8;
  if (C <= 0 || C > CA_MAXC || C % ch != 0) {
    srk_set_error("%s: C=%d unsupported (multiple of %d, <= %d)", who, C, ch, CA_MAXC);
    return SRK_E_BADARG;
  }
  if (Cr < 0 || Cr > CA_MAXCR) {
    srk_set_error("%s: Cr=%d unsupported (<= %d)", who, Cr, CA_MAXCR);
    return SRK_E_BADARG;
  }
  return 0;
}

}  // namespace

#define CA_DISPATCH(KERNEL, ARGS, GRID, ST, PPB)                                                          \
  switch ((ARGS).dtype) {                                                                                 \
    case SRK_BF16: hipLaunchKernelGGL(KERNEL<SRK_BF16>, GRID, dim3(CA_NT), 0, ST, ARGS, PPB); break;      \
    case SRK_F16: hipLaunchKernelGGL(KERNEL<SRK_F16>, GRID, dim3(CA_NT), 0, ST, ARGS, PPB); break;        \
    case SRK_F32: hipLaunchKernelGGL(KERNEL<SRK_F32>, GRID, dim3(CA_NT), 0, ST, ARGS, PPB); break;        \
    default: srk_set_error("channel attention: dtype %d", (ARGS).dtype); return SRK_E_BADARG;             \
  }

extern "C" int srk_ca_splits(int N, int HW) {
  if (N <= 0 || HW <= 0) return 1;
  int splits, ppb;
  split_for(N, HW, &splits, &ppb);
  return splits;
}

extern "C" int srk_ca_pool(const srk_ca_pool_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->t && a->sums, "srk_ca_pool: null pointer");
  if (int e = check_c(a->C, 0, a->dtype, "srk_ca_pool")) return e;
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->t_pitch % ch == 0 && a->t_coff % ch == 0 && (!a->u || (a->u_pitch % ch == 0 && a->u_coff % ch == 0)), "srk_ca_pool: alignment");
  int splits, ppb;
  split_for(a->N, a->HW, &splits, &ppb);
  CA_DISPATCH(ca_pool_kernel, *a, dim3(a->N, splits), reinterpret_cast<hipStream_t>(stream), ppb);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_ca_apply(const srk_ca_apply_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->t && a->sums && a->out && a->w1 && a->b1 && a->w2 && a->b2, "srk_ca_apply: null pointer");
  if (int e = check_c(a->C, a->Cr, a->dtype, "srk_ca_apply")) return e;
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->t_pitch % ch == 0 && a->t_coff % ch == 0 && a->out_pitch % ch == 0 && a->out_coff % ch == 0 &&
                    (!a->res || (a->res_pitch % ch == 0 && a->res_coff % ch == 0)), "srk_ca_apply: alignment");
  int splits, ppb;
  split_for(a->N, a->HW, &splits, &ppb);
  CA_DISPATCH(ca_apply_kernel, *a, dim3(a->N, splits), reinterpret_cast<hipStream_t>(stream), ppb);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_ca_bwd_apply(const srk_ca_bwd_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->g && a->gsum && a->sums && a->s && a->z && a->w1 && a->w2 && a->dw1 && a->db1 && a->dw2 && a->db2 && a->gt,
                "srk_ca_bwd_apply: null pointer");
  if (int e = check_c(a->C, a->Cr, a->dtype, "srk_ca_bwd_apply")) return e;
  const int ch = a->dtype == SRK_F32 ? 4 : 8;
  SRK_CHECK_ARG(a->g_pitch % ch == 0 && a->g_coff % ch == 0 && a->gt_pitch % ch == 0 && a->gt_coff % ch == 0, "srk_ca_bwd_apply: alignment");
  int splits, ppb;
  split_for(a->N, a->HW, &splits, &ppb);
  CA_DISPATCH(ca_bwd_kernel, *a, dim3(a->N, splits), reinterpret_cast<hipStream_t>(stream), ppb);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_rowsum_group(const srk_rowsum_job* table_dev, int njobs, int max_k, srk_stream_t stream) {
  SRK_CHECK_ARG(table_dev && njobs > 0 && njobs <= 65535 && max_k > 0, "srk_rowsum_group: bad table (%d jobs, k <= %d)", njobs, max_k);
  int bx = (max_k + CA_NT - 1) / CA_NT;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(rowsum_group_kernel, dim3(bx, njobs), dim3(CA_NT), 0, reinterpret_cast<hipStream_t>(stream), table_dev);
  SRK_LAUNCH_CHECK();
  return 0;
}
