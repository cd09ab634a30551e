// Adam update of every parameter tensor of a model in ONE launch.
//
// Replaces torch.optim.Adam.step() as the reference configures it (models/srmodel.py:145-154: `optim.Adam(trainable)` at
// torch's defaults; :602-603 drops every user-supplied hyper-parameter).  torch's fused implementation packs at most a few
// dozen tensors into one launch's arguments: RCAN's ~1,600 parameter tensors are 45 launches of ~28 us (1.3 ms of a 15 ms
// batch-16 step), EDSR-baseline's 4 launches are 7 % of its step.  Here the tensor list lives in a device table that is
// uploaded when the set of (parameter, gradient) addresses changes, and one grid walks all of it: HBM-bound
// (7 floats of traffic per parameter), ~0.1 ms for RCAN.
//
//   m = beta1*m + (1-beta1)*g;  v = beta2*v + (1-beta2)*g*g;
//   p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)          (t = step count after this step, fp32 like torch)
// `weight_decay` is the L2 form (g += wd*p), `maximize` negates g: torch.optim.Adam's semantics.  The step counts (per tensor, as
// torch keeps them) live on the device so that hipGraph replays advance them: the update reads its tensor's count, a second
// tiny launch increments the counts of all tensors in the table.
#include "srk_common.h"

namespace {

constexpr int ADAM_NT = 256;

// Dynamic loss scaling on the DEVICE (fp16 training, the reference's `precision: 16` = Lightning's "16-mixed": autocast +
// torch.amp.GradScaler, configs/all.yml:122): state = {scale, growth tracker, found_inf, growth factor, backoff factor, growth
// interval, skipped steps}.  GradScaler reads found_inf on the HOST every step (`optimizer.step` is skipped from Python), which a
// hipGraph replay cannot do; here the check, the skip and the scale update are launches of the captured step:
//   adam_check_kernel  : found_inf = any gradient not finite           (the gradients are still multiplied by `scale`)
//   adam_group_kernel  : found_inf ? nothing : the update on g / scale
//   adam_bump_kernel   : found_inf ? nothing : step counts += 1
//   scaler_update      : found_inf ? (scale *= backoff, tracker = 0) : (++tracker == interval ? scale *= growth, tracker = 0); found_inf = 0
// -- torch.amp.GradScaler.step / update, same constants, same order.
__global__ __launch_bounds__(ADAM_NT) void adam_check_kernel(const srk_adam_args a, float* __restrict__ state) {
  const srk_adam_block blk = a.blocks[blockIdx.x];
  const srk_adam_slot sl = a.slots[blk.slot];
  const long long e0 = blk.start, e1 = (e0 + blk.count < sl.n) ? e0 + blk.count : sl.n;
  bool bad = false;
  for (long long e = e0 + threadIdx.x; e < e1; e += ADAM_NT) bad |= !isfinite(sl.g[e]);
  if (bad) state[2] = 1.f;                  // every writer stores the same value
}

__global__ void scaler_update_kernel(float* __restrict__ state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[2] != 0.f) {
    state[0] *= state[4];
    state[1] = 0.f;
    state[6] += 1.f;
  } else {
    const float t = state[1] + 1.f;
    if (t >= state[5]) { state[0] *= state[3]; state[1] = 0.f; }
    else state[1] = t;
  }
  state[2] = 0.f;
}

template <bool SCALED>
__global__ __launch_bounds__(ADAM_NT) void adam_group_kernel(const srk_adam_args a, const float* __restrict__ state) {
  float inv_scale = 1.f;
  if constexpr (SCALED) {
    if (state[2] != 0.f) return;            // a non-finite gradient somewhere: the whole step is skipped
    inv_scale = 1.f / state[0];
  }
  const srk_adam_block blk = a.blocks[blockIdx.x];
  const srk_adam_slot sl = a.slots[blk.slot];
  const float t = a.steps[sl.step_idx] + 1.f;
  const float bc1 = 1.f - powf(a.beta1, t), bc2 = 1.f - powf(a.beta2, t);
  const float step_size = a.lr / bc1, bc2s = sqrtf(bc2);
  const float b2 = a.beta2, eps = a.eps, wd = a.weight_decay, omb1 = a.one_minus_beta1, omb2 = a.one_minus_beta2;
  const long long e0 = blk.start, e1 = (e0 + blk.count < sl.n) ? e0 + blk.count : sl.n;
  float* const p = sl.p;
  const float* const g = sl.g;
  float* const m = a.m + sl.state_off;
  float* const v = a.v + sl.state_off;
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    if constexpr (SCALED) gv *= inv_scale;
    if (a.maximize) gv = -gv;
    if (wd != 0.f) gv += wd * pv;
    mv = mv + (gv - mv) * omb1;                             // torch: exp_avg.lerp_(grad, 1 - beta1)
    vv = b2 * vv + omb2 * gv * gv;                          //        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vv) / bc2s + eps;
    pv -= step_size * (mv / denom);
  };
  // 16-byte accesses when all four arrays allow it (tensor starts inside the flat state buffers are 4-float aligned)
  const bool vec = ((((uintptr_t)p | (uintptr_t)g) & 15) == 0) && (sl.state_off & 3) == 0 && (e0 & 3) == 0;
  if (vec) {
    // 4 x float4 per array and thread, all 16 loads issued before the first use (a 4096-element block is one pass)
    const long long n4 = (e1 - e0) >> 2;
    for (long long i0 = 0; i0 < n4; i0 += 4 * ADAM_NT) {
      f32x4 p4[4], m4[4], v4[4], g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long i = i0 + u * ADAM_NT + threadIdx.x;
        if (i < n4) {
          const long long e = e0 + 4 * i;
          p4[u] = *reinterpret_cast<const f32x4*>(p + e); g4[u] = *reinterpret_cast<const f32x4*>(g + e);
          m4[u] = *reinterpret_cast<const f32x4*>(m + e); v4[u] = *reinterpret_cast<const f32x4*>(v + e);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long i = i0 + u * ADAM_NT + threadIdx.x;
        if (i < n4) {
          const long long e = e0 + 4 * i;
          float pv[4] = {p4[u].x, p4[u].y, p4[u].z, p4[u].w}, mv[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
          float vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
          const float gv[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) upd(pv[k], gv[k], mv[k], vv[k]);
          *reinterpret_cast<f32x4*>(p + e) = f32x4{pv[0], pv[1], pv[2], pv[3]};
          *reinterpret_cast<f32x4*>(m + e) = f32x4{mv[0], mv[1], mv[2], mv[3]};
          *reinterpret_cast<f32x4*>(v + e) = f32x4{vv[0], vv[1], vv[2], vv[3]};
        }
      }
    }
    for (long long e = e0 + 4 * n4 + threadIdx.x; e < e1; e += ADAM_NT) upd(p[e], g[e], m[e], v[e]);
  } else {
    for (long long e = e0 + threadIdx.x; e < e1; e += ADAM_NT) upd(p[e], g[e], m[e], v[e]);
  }
}

// second launch of a step: the update has read every count, advance the counts of the tensors in the table.  (A "last block
// bumps" ticket inside the update costs one same-address atomic per block: 5k blocks serialise to 0.3 ms.)
__global__ __launch_bounds__(ADAM_NT) void adam_bump_kernel(const srk_adam_args a, const float* __restrict__ state) {
  if (state && state[2] != 0.f) return;
  for (int i = blockIdx.x * ADAM_NT + threadIdx.x; i < a.nslots; i += gridDim.x * ADAM_NT) a.steps[a.slots[i].step_idx] += 1.f;
}

}  // namespace

extern "C" int srk_adam_step(const srk_adam_args* a, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->slots && a->blocks && a->m && a->v && a->steps, "srk_adam_step: null pointer");
  SRK_CHECK_ARG(a->nblocks > 0 && a->nslots > 0, "srk_adam_step: %d blocks, %d tensors", a->nblocks, a->nslots);
  SRK_CHECK_ARG(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f && a->lr >= 0.f,
                "srk_adam_step: lr=%g betas=(%g, %g) eps=%g", a->lr, a->beta1, a->beta2, a->eps);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_group_kernel<false>, dim3((unsigned)a->nblocks), dim3(ADAM_NT), 0, st, *a, (const float*)nullptr);
  hipLaunchKernelGGL(adam_bump_kernel, dim3((unsigned)((a->nslots + ADAM_NT - 1) / ADAM_NT)), dim3(ADAM_NT), 0, st, *a, (const float*)nullptr);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_adam_step_scaled(const srk_adam_args* a, float* scaler_state, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->slots && a->blocks && a->m && a->v && a->steps && scaler_state, "srk_adam_step_scaled: null pointer");
  SRK_CHECK_ARG(a->nblocks > 0 && a->nslots > 0, "srk_adam_step_scaled: %d blocks, %d tensors", a->nblocks, a->nslots);
  SRK_CHECK_ARG(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f && a->lr >= 0.f,
                "srk_adam_step_scaled: lr=%g betas=(%g, %g) eps=%g", a->lr, a->beta1, a->beta2, a->eps);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_check_kernel, dim3((unsigned)a->nblocks), dim3(ADAM_NT), 0, st, *a, scaler_state);
  hipLaunchKernelGGL(adam_group_kernel<true>, dim3((unsigned)a->nblocks), dim3(ADAM_NT), 0, st, *a, (const float*)scaler_state);
  hipLaunchKernelGGL(adam_bump_kernel, dim3((unsigned)((a->nslots + ADAM_NT - 1) / ADAM_NT)), dim3(ADAM_NT), 0, st, *a, (const float*)scaler_state);
  SRK_LAUNCH_CHECK();
  return 0;
}

/* Several parameter groups: GradScaler.step unscales and checks EVERY group before any is updated -- an inf in group 1
 * must not leave group 0 already stepped.  srk_adam_check_scaled over all groups first, then srk_adam_update_scaled per group. */
extern "C" int srk_adam_check_scaled(const srk_adam_args* a, float* scaler_state, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->slots && a->blocks && scaler_state, "srk_adam_check_scaled: null pointer");
  SRK_CHECK_ARG(a->nblocks > 0 && a->nslots > 0, "srk_adam_check_scaled: %d blocks, %d tensors", a->nblocks, a->nslots);
  hipLaunchKernelGGL(adam_check_kernel, dim3((unsigned)a->nblocks), dim3(ADAM_NT), 0, reinterpret_cast<hipStream_t>(stream), *a, scaler_state);
  SRK_LAUNCH_CHECK();
  return 0;
}

extern "C" int srk_adam_update_scaled(const srk_adam_args* a, const float* scaler_state, srk_stream_t stream) {
  SRK_CHECK_ARG(a && a->slots && a->blocks && a->m && a->v && a->steps && scaler_state, "srk_adam_update_scaled: null pointer");
  SRK_CHECK_ARG(a->nblocks > 0 && a->nslots > 0, "srk_adam_update_scaled: %d blocks, %d tensors", a->nblocks, a->nslots);
  SRK_CHECK_ARG(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f && a->lr >= 0.f,
                "srk_adam_update_scaled: lr=%g betas=(%g, %g) eps=%g", a->lr, a->beta1, a->beta2, a->eps);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_group_kernel<true>, dim3((unsigned)a->nblocks), dim3(ADAM_NT), 0, st, *a, scaler_state);
  hipLaunchKernelGGL(adam_bump_kernel, dim3((unsigned)((a->nslots + ADAM_NT - 1) / ADAM_NT)), dim3(ADAM_NT), 0, st, *a, scaler_state);
  SRK_LAUNCH_CHECK();
  return 0;
}

/* after the LAST parameter group's srk_adam_step_scaled of a step: the scale / growth-tracker update (one thread) */
extern "C" int srk_loss_scale_update(float* scaler_state, srk_stream_t stream) {
  SRK_CHECK_ARG(scaler_state, "srk_loss_scale_update: null pointer");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), scaler_state);
  SRK_LAUNCH_CHECK();
  return 0;
}
