// Internal (not part of the C ABI): the in-kernel-draw entry points with the optional device-side offset base that
// snr_render_rays_fused_forward passes through (include/spinnerf_hip.h: snr_step_state).
#pragma once
#include <stdint.h>

#include "../../include/spinnerf_hip.h"

namespace snr {
int sample_coarse_rng_impl(const float* rays, int ld, int64_t n_rays, int N, int lindisp, uint64_t seed, uint64_t offset,
                           const uint64_t* base, float* z_vals, snr_stream_t stream);
int sample_fine_rng_impl(const float* z_coarse, const float* weights, int64_t n_rays, int Nc, int Nf, uint64_t seed,
                         uint64_t offset, const uint64_t* base, float* z_out, float* z_samples, float* z_std,
                         snr_stream_t stream);
// the loss terms of a training render (include/spinnerf_hip.h: snr_loss_terms), as the kernels take them: by value
// slot / slot_final are ELEMENT offsets from the kernels' `loss` pointer: 0..3 for the library routes (one 4-float block); the
// public snr_composite_train passes the distance to its independent second accumulator in slot_final (any sign): kNoSlot = none
constexpr int64_t kNoSlot = INT64_MIN;
struct LossTerm { int64_t first, n; int kind; const float* target; float inv_count; int slot; int64_t slot_final; };
struct LossSpec { int n; LossTerm t[4]; };
// one term over rays [0, n_rays): mean((rgb - target)^2) over 3 * n_rays_global elements into loss[0] (and loss[1] from the final pass)
inline LossSpec plain_rgb_loss(const float* target, int64_t n_rays, int64_t n_rays_global) {
  LossSpec s{};
  s.n = 1;
  s.t[0] = LossTerm{0, n_rays, 0, target, 1.f / (3.f * (float)n_rays_global), 0, 1};
  return s;
}
int composite_train_impl(const float* raw, int C, const float* z, const float* rays, int ld, const float* noise,
                         float noise_std, uint64_t seed, uint64_t offset, const uint64_t* base, int64_t n_rays, int S,
                         int white, const LossSpec& spec, int final_pass, float* rgb_map, float* disp_map,
                         float* acc_map, float* depth_map, float* weights, float* d_raw, float* loss, snr_stream_t stream);
int composite_train_sample_impl(const float* raw, int C, const float* z, const float* rays, int ld, const float* noise,
                                float noise_std, uint64_t seed, uint64_t offset, const uint64_t* base, int64_t n_rays, int Nc,
                                int white, const LossSpec& spec, float* rgb_map, float* disp_map,
                                float* acc_map, float* depth_map, float* weights, float* d_raw, float* loss, const float* u,
                                int use_rng_u, uint64_t offset_u, int Nf, float* z_out, float* z_samples, float* z_std,
                                snr_stream_t stream);
// behind the final pass of a render with a NaN-guarded term (run_nerf.py:1518-1521): loss[slot] NaN -> the term's rows of
// both passes' d raw become 0 and the term is dropped; else loss[0] += loss[slot]
int loss_guard_impl(float* loss, int slot, int64_t first_ray, int64_t n_rays, float* d_raw0, int64_t row0, float* d_raw,
                    int64_t row1, snr_stream_t stream);
int pack_rays_sample_impl(const float* rays_o, const float* rays_d, int64_t n_rays, int H, int W, float focal, int ndc, float near,
                          float far, int use_viewdirs, float* rays, int ld, int N, int lindisp, const float* t_rand, int use_rng,
                          uint64_t seed, uint64_t offset, const uint64_t* base, float* z_vals, float* zero, int n_zero,
                          snr_stream_t stream);
}  // namespace snr
