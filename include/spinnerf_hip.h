/* libspinnerf_hip — C ABI of the MI355X-native volumetric-render hot path.
 *
 * Drop-in scope: the arithmetic behind DS_NeRF/run_nerf.py's render() / render_rays() /
 * network_query_fn surface of SamsungLabs/SPIn-NeRF.  The reference has NO native
 * interface on this path (it is stock torch ops; its only native code, DS_NeRF/torchsearchsorted,
 * is dead code — SURVEY.md §2 row 5), so each entry point below names the reference *Python*
 * function whose arithmetic it replaces.  The Python host in spin-nerf_amd/ binds these with
 * ctypes and reproduces the reference's signatures; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; tensors are
 *     contiguous row-major fp32 unless stated; the caller owns all memory (no allocation inside)
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *   - return: 0 = ok, <0 = bad argument (SNR_ERR_*), >0 = hipError_t from the launch
 *   - re-entrant per stream.  Process-wide state is limited to: the SNR_* environment switches (A/B experiments) and the
 *     CU count of each device, read ONCE at the first call that needs them (snr_tunables_reload re-reads the environment);
 *     the per-kernel LDS attribute set once per device; and the optional profiling log (snr_prof_*).  Nothing a launch
 *     computes depends on a previous launch.
 */
#ifndef SPINNERF_HIP_H
#define SPINNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNR_ABI_VERSION 4   /* 2: snr_mlp_backward / snr_pack_rays argument lists, snr_net / snr_step_state (round 2)
                             * 3: snr_mlp_backward_multi, snr_adam_pack_multi, snr_render_step_prepare, snr_render_config.flags,
                             *    snr_render_ws_layout.bwd_ws0, snr_tunables_reload (round 4)
                             * 4: snr_render_rays_fused_forward_terms, snr_loss_terms, SNR_RENDER_LOSS4 (round 5) */

#define SNR_OK 0
#define SNR_ERR_NULL (-1)         /* a required pointer is NULL */
#define SNR_ERR_SHAPE (-2)        /* a size is out of the supported range */
#define SNR_ERR_UNSUPPORTED (-3)  /* configuration not implemented by the HIP path */

#define SNR_PREC_BF16 0 /* bf16 MFMA inputs, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
#define SNR_PREC_FP32 1 /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32) — the parity mode */

typedef void* snr_stream_t;

/* Shape of one NeRF MLP (DS_NeRF/run_nerf_helpers.py:74-127; built by create_nerf,
 * run_nerf.py:380-425).  netdepth=8, netwidth=256, skips=[4] are fixed (the reference's
 * defaults and every BASELINE config); anything else returns SNR_ERR_UNSUPPORTED. */
typedef struct snr_mlp_config {
  int multires;       /* --multires, 0..10            (run_nerf.py:383) */
  int multires_views; /* --multires_views, 0..4       (run_nerf.py:388) */
  int i_embed;        /* 0 = positional encoding, -1 = identity (helpers:56-57) */
  int use_viewdirs;   /* --use_viewdirs */
  int out_ch;         /* output_ch of output_linear when !use_viewdirs (4 or 5, run_nerf.py:390); 4 with viewdirs */
  int precision;      /* SNR_PREC_* */
} snr_mlp_config;

int snr_abi_version(void);
const char* snr_status_string(int status);

/* ---- NeRF MLP: replaces run_network()/network_query_fn + NeRF.forward + their autograd ----
 * (run_nerf.py:56-71, 427-430; helpers:22-70, 104-127) */

/* number of fp32 parameters, in the reference module's state-dict order (flat buffer layout) */
int64_t snr_mlp_param_count(const snr_mlp_config* cfg);
/* bytes of the packed (MFMA fragment order) weight blob produced by snr_mlp_pack */
int64_t snr_mlp_packed_bytes(const snr_mlp_config* cfg);
/* bytes of activations snr_mlp_forward saves for backward when `act` != NULL */
int64_t snr_mlp_act_bytes(const snr_mlp_config* cfg, int64_t n_samples);
/* bytes of scratch snr_mlp_backward needs */
int64_t snr_mlp_bwd_ws_bytes(const snr_mlp_config* cfg, int64_t n_samples);

/* flat fp32 parameters -> packed blob (call after every optimizer step) */
int snr_mlp_pack(const snr_mlp_config* cfg, const float* params, void* packed, snr_stream_t stream);

/* raw[n_samples, out_ch] = NeRF(embed(pts), embed(viewdirs of the sample's ray)).
 * Sample positions come from `pts` [n_samples,3] if non-NULL, else pts = o + d*z is formed
 * in-kernel from `rays` (row r = o(3) d(3) ..., leading dimension ray_ld floats) and
 * `z_vals` [n_rays, samples_per_ray] (run_nerf.py:670-671).  `viewdirs` = n_rays rows of 3 floats
 * with leading dimension viewdirs_ld (so the last three columns of the packed ray batch can be
 * passed in place, run_nerf.py:642; required iff use_viewdirs); sample m belongs to ray
 * m / samples_per_ray (the expand at run_nerf.py:63).
 * `act` NULL = inference; else snr_mlp_act_bytes() of workspace that backward consumes. */
int snr_mlp_forward(const snr_mlp_config* cfg, const void* packed, const float* pts, const float* rays,
                    int ray_ld, const float* z_vals, const float* viewdirs, int viewdirs_ld, int64_t n_samples,
                    int samples_per_ray, float* raw, void* act, snr_stream_t stream);

/* d(loss)/d(params) from d(loss)/d(raw).  No gradient flows to pts/viewdirs.
 * `params` = the flat fp32 parameters `packed` was built from (with use_viewdirs the gradients of feature_linear and of
 * views_linears.0's feature columns are formed from d z9^T h7 and these weights instead of streaming the feature
 * tensor; may be NULL without viewdirs).  grad_params (flat, fp32): overwritten if accumulate == 0, else += . */
int snr_mlp_backward(const snr_mlp_config* cfg, const void* packed, const float* params, const float* d_raw,
                     int64_t n_samples, const void* act, void* ws, float* grad_params, int accumulate,
                     snr_stream_t stream);

/* The backward passes of SEVERAL networks (the coarse and the fine network of one render_rays, run_nerf.py:593-737, whose
 * gradients loss.backward() produces together, :1611) as ONE launch sequence: one chain launch, one weight-gradient launch
 * and one reduce cover all of them — one accumulator flush and one launch ramp instead of one per network.  Every item is
 * what snr_mlp_backward takes; 1 <= n_items <= 2.  Items that cannot share a launch sequence (fp32 mode, different
 * use_viewdirs, SNR_RECOMPUTE=0) are run one after the other; the results are the same either way. */
typedef struct snr_mlp_bwd_item {
  const snr_mlp_config* cfg;
  const void* packed;
  const float* params;
  const float* d_raw;
  int64_t n_samples;
  const void* act;
  void* ws;               /* snr_mlp_bwd_ws_bytes(cfg, n_samples) of scratch, distinct per item */
  float* grad_params;
  int accumulate;
} snr_mlp_bwd_item;
int snr_mlp_backward_multi(const snr_mlp_bwd_item* items, int n_items, snr_stream_t stream);

/* ---- hash-grid radiance network: replaces NeRF_TCNN.forward and its autograd (run_nerf_helpers_tcnn.py:13-113; the
 * reference's default network, create_nerf_tcnn, run_nerf.py:499-590).  The reference delegates this arithmetic to
 * tiny-cuda-nn (tcnn.Encoding "HashGrid" / "SphericalHarmonics", tcnn.Network "FullyFusedMLP"): parity unpinned, the
 * definition followed is restated in oracle/hashgrid_oracle.py.  All configuration values are the reference's constants
 * (16 levels x 2 features, 2^19 entries, base 16, bound 100, SH degree 4, 32->64->16 and 32->64->64->16 bias-free MLPs).
 * `params` = one flat fp32 buffer [table entries x 2 | sigma_net.params | color_net.params] (state-dict order). */
int64_t snr_hashgrid_table_entries(void);
int64_t snr_hashgrid_param_count(void);
int64_t snr_hashgrid_packed_bytes(void);
int64_t snr_hashgrid_act_bytes(int64_t n_samples);
int64_t snr_hashgrid_bwd_ws_bytes(int64_t n_samples);
/* MLP weights -> MFMA fragment order (call after every optimizer step; the table is read in place) */
int snr_hashgrid_pack(const float* params, void* packed, snr_stream_t stream);
/* raw[n_samples,4] = (colour 3 — no activation, sigma) for sample positions `pts` [n,3] or o + d*z formed in-kernel from
 * `rays` / `z_vals` (as snr_mlp_forward); `viewdirs` (unit, per ray) is required: the network reads input[:, 3:]
 * (tcnn.py:88).  `act` NULL = inference, else snr_hashgrid_act_bytes() that backward consumes. */
int snr_hashgrid_forward(const float* params, const void* packed, const float* pts, const float* rays, int ray_ld,
                         const float* z_vals, const float* viewdirs, int viewdirs_ld, int64_t n_samples,
                         int samples_per_ray, float* raw, void* act, snr_stream_t stream);
/* d(loss)/d(params) from d(loss)/d(raw) [n,4] for the same inputs as the forward call.  grad_params (flat, fp32):
 * overwritten if accumulate == 0 (the table part is cleared, then accumulated with atomics), else += . */
int snr_hashgrid_backward(const float* params, const void* packed, const float* pts, const float* rays, int ray_ld,
                          const float* z_vals, const float* viewdirs, int viewdirs_ld, const float* d_raw,
                          int64_t n_samples, int samples_per_ray, const void* act, void* ws, float* grad_params,
                          int accumulate, snr_stream_t stream);

/* ---- stratified sampling: replaces run_nerf.py:646-668 ----
 * z_vals[n_rays, n_samples] from near/far = rays[:,6], rays[:,7]; lindisp per run_nerf.py:647-650;
 * t_rand [n_rays, n_samples] non-NULL = perturb (run_nerf.py:654-668). */
int snr_sample_coarse(const float* rays, int ray_ld, int64_t n_rays, int n_samples, int lindisp,
                      const float* t_rand, float* z_vals, snr_stream_t stream);

/* ---- alpha compositing: replaces raw2outputs (helpers:350-401) ----
 * raw [n_rays, S, raw_ch] (channels 0..2 rgb, 3 density), z_vals [n_rays,S], rays_d = rays + 3
 * (leading dimension ray_ld), noise [n_rays,S] pre-scaled or NULL.  Outputs: rgb_map [n,3],
 * disp_map, acc_map, depth_map [n], weights [n,S], alpha [n,S] or NULL. */
int snr_composite_forward(const float* raw, int raw_ch, const float* z_vals, const float* rays, int ray_ld,
                          const float* noise, int64_t n_rays, int S, int white_bkgd, float* rgb_map,
                          float* disp_map, float* acc_map, float* depth_map, float* weights, float* alpha,
                          snr_stream_t stream);

/* autograd of the above w.r.t. raw.  Upstream grads may be NULL (= zero).  detach_weights:
 * rgb_map does not back-propagate into weights (helpers:385-388).  d_raw [n_rays,S,raw_ch] is
 * fully overwritten (channels >= 4 get 0). */
int snr_composite_backward(const float* raw, int raw_ch, const float* z_vals, const float* rays, int ray_ld,
                           const float* noise, int64_t n_rays, int S, int white_bkgd, int detach_weights,
                           const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_depth,
                           const float* g_weights, const float* g_alpha, float* d_raw, snr_stream_t stream);

/* Compositing from opacities the caller computed: MVSeg's raw2outputs post-processes alpha before the transmittance
 * product when only_object is set (MVSeg/DS_NeRF/run_nerf_helpers.py:383-397).  alpha [n_rays,S]; colours from
 * raw[..., :3].  Backward: d_raw gets the colour gradients (channels >= 3 are 0), d_alpha [n_rays,S] = d loss / d alpha. */
int snr_composite_alpha_forward(const float* raw, int raw_ch, const float* z_vals, const float* rays, int ray_ld,
                                const float* alpha, int64_t n_rays, int S, int white_bkgd, float* rgb_map,
                                float* disp_map, float* acc_map, float* depth_map, float* weights, snr_stream_t stream);
int snr_composite_alpha_backward(const float* raw, int raw_ch, const float* z_vals, const float* rays, int ray_ld,
                                 const float* alpha, int64_t n_rays, int S, int white_bkgd, int detach_weights,
                                 const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_depth,
                                 const float* g_weights, float* d_raw, float* d_alpha, snr_stream_t stream);

/* ---- hierarchical sampling: replaces sample_pdf + sort + z_std
 * (helpers:304-347, run_nerf.py:697-702, 726) ----
 * bins = midpoints of z_coarse, pdf from weights[:,1:-1]; u [n_rays, n_fine] or NULL (= the
 * deterministic linspace(0,1,n_fine) of det=True).  z_out [n_rays, n_coarse+n_fine] = sorted union,
 * z_samples [n_rays, n_fine] (may be NULL), z_std [n_rays] = population std of the new samples. */
int snr_sample_fine(const float* z_coarse, const float* weights, const float* u, int64_t n_rays,
                    int n_coarse, int n_fine, float* z_out, float* z_samples, float* z_std,
                    snr_stream_t stream);

/* sample_pdf alone (helpers:304-347) for callers that hold their own bins: bins [n_rays, n_bins], weights
 * [n_rays, n_bins-1], u [n_rays, n_samples] or NULL (= linspace(0,1,n_samples)); samples [n_rays, n_samples]. */
int snr_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins,
                   int n_samples, float* samples, snr_stream_t stream);

/* ---- the training step's production forms (run_nerf.py:593-737 + 1482-1490 + autograd's first step): the same
 * arithmetic with the random draws made in-kernel and the loss folded into the compositing kernel.
 * Random numbers: Philox4x32-10, element i of a call = counter (i, 0, offset) under key `seed`; the caller advances
 * `offset` by one per call.  (The reference draws torch.rand / torch.randn, run_nerf.py:660-668, 699 and helpers:376-381;
 * the entry points above take injected draws and are what the parity tests use.) */
/* stratified z_vals with t_rand ~ U[0,1) drawn in-kernel */
int snr_sample_coarse_rng(const float* rays, int ray_ld, int64_t n_rays, int n_samples, int lindisp, uint64_t seed,
                          uint64_t offset, float* z_vals, snr_stream_t stream);
/* hierarchical sampling with u ~ U[0,1) drawn in-kernel */
int snr_sample_fine_rng(const float* z_coarse, const float* weights, int64_t n_rays, int n_coarse, int n_fine,
                        uint64_t seed, uint64_t offset, float* z_out, float* z_samples, float* z_std,
                        snr_stream_t stream);
/* raw2outputs forward + this network's loss term mean((rgb_map - target)^2) over the GLOBAL batch (3 * n_rays_global
 * elements, so that data-parallel shards sum to the global mean) + the backward of both, in one kernel: maps and weights
 * as snr_composite_forward, d_raw as snr_composite_backward would give for g_rgb = d loss / d rgb_map.  Density noise:
 * `noise` [n_rays,S] pre-scaled, or NULL with noise_std > 0 = N(0,1) * noise_std drawn in-kernel (the same numbers in the
 * forward and the backward half).  loss[0] += the term, and loss_also[0] too when non-NULL (zero them first); the two are
 * independent accumulators: any two 4-byte aligned device floats, not necessarily of one allocation. */
int snr_composite_train(const float* raw, int raw_ch, const float* z_vals, const float* rays, int ray_ld,
                        const float* noise, float noise_std, uint64_t seed, uint64_t offset, int64_t n_rays, int S,
                        int white_bkgd, int detach_weights, const float* target, int64_t n_rays_global, float* rgb_map,
                        float* disp_map, float* acc_map, float* depth_map, float* weights, float* d_raw, float* loss,
                        float* loss_also, snr_stream_t stream);

/* ---- per-step scalars in DEVICE memory, so that a captured HIP graph of the training step can be replayed: the draw
 * counter the in-kernel random numbers are offset by, and Adam's rate / bias corrections (run_nerf.py:1611-1622).  The
 * struct describes the step about to run; snr_step_state_advance turns it into the next one on the device exactly as
 * the host would (lr = lrate * 0.1^(global_step / decay_steps) from the not yet incremented global_step). ---- */
typedef struct snr_step_state {
  uint64_t offset_base;  /* added to every call offset of this step's draws */
  int64_t opt_step;      /* Adam's step number of this step (1-based) */
  int64_t global_step;   /* completed optimisation steps */
  float lr;              /* rate this step uses */
  float bc1;             /* 1 - beta1^opt_step */
  float bc2_sqrt;        /* sqrt(1 - beta2^opt_step) */
  float reserved;
} snr_step_state;
int snr_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const snr_step_state* state, float beta1, float beta2, float eps, float grad_scale,
                      snr_stream_t stream);
int snr_step_state_advance(snr_step_state* state, double lrate, double decay_steps, float beta1, float beta2,
                           uint64_t n_offsets, snr_stream_t stream);

/* ---- render_rays as ONE call (render_rays_fused_forward / _backward): the launch sequence of
 * run_nerf.py:593-737 — stratified sampling, coarse network, compositing, hierarchical sampling + sort, fine network,
 * compositing — enqueued by the library on `stream`, every intermediate in one caller-provided workspace; with `target`
 * also the loss terms of run_nerf.py:1482-1490 and the compositing backward (snr_composite_train), after which
 * snr_render_rays_fused_backward runs autograd's part: both networks' parameter gradients.  One network evaluation =
 * snr_mlp_forward (kind SNR_NET_MLP) or snr_hashgrid_forward (SNR_NET_HASHGRID); fine == NULL with n_importance > 0
 * evaluates the coarse network twice (run_nerf.py:705). ---- */
#define SNR_NET_MLP 0
#define SNR_NET_HASHGRID 1
typedef struct snr_net {
  int kind;              /* SNR_NET_* */
  snr_mlp_config mlp;    /* kind == SNR_NET_MLP */
  const void* packed;    /* snr_mlp_pack / snr_hashgrid_pack output */
  const float* params;   /* the flat fp32 parameters `packed` was built from */
} snr_net;
typedef struct snr_render_config {
  int n_samples;         /* N_samples */
  int n_importance;      /* N_importance (0 = coarse pass only) */
  int lindisp;           /* --lindisp */
  int white_bkgd;        /* --white_bkgd */
  int perturb;           /* perturb > 0: stratified / hierarchical draws are random */
  float raw_noise_std;   /* --raw_noise_std */
  int flags;             /* SNR_RENDER_* */
} snr_render_config;
#define SNR_RENDER_Z_COARSE_READY 1  /* the workspace already holds the stratified z_vals of the coarse samples
                                        (snr_render_step_prepare wrote them): the forward does not sample them again */
#define SNR_RENDER_LOSS4 2           /* snr_render_step_prepare zero-fills the four loss slots of a forward with loss terms */
/* byte offsets inside the workspace of the tensors render_rays returns in its dict (run_nerf.py:715-726) and of the
 * intermediates the backward consumes; -1 = absent in this configuration */
typedef struct snr_render_ws_layout {
  int64_t z_coarse, raw0, weights0, depth0;  /* [n,Nc], [n,Nc,C], [n,Nc], [n] */
  int64_t z_vals, raw, weights, z_samples;   /* [n,Nc+Nf], [n,Nc+Nf,C], [n,Nc+Nf], [n,Nf] (n_importance > 0) */
  int64_t d_raw0, d_raw, act0, act, bwd_ws;  /* training only; bwd_ws = the final pass's backward scratch */
  int64_t bwd_ws0;                           /* the coarse pass's backward scratch (n_importance > 0; else bwd_ws serves it):
                                              * the two backward passes run as one launch sequence and cannot share one */
  int64_t total;                             /* bytes for inference (train == 0) or training */
} snr_render_ws_layout;
int snr_render_rays_fused_layout(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine, int64_t n_rays,
                                 int train, snr_render_ws_layout* out);
/* rays: packed rows (snr_pack_rays / snr_make_rays).  Random draws: t_rand [n,Nc], u [n,Nf], noise0 [n,Nc], noise
 * [n,Nc+Nf] (pre-scaled) when non-NULL, else Philox draws with offsets offset+1 (t_rand), +2 (noise0), +3 (u), +4 (noise)
 * under `seed` when the configuration asks for random numbers (the call always consumes four offsets); offset_base
 * (device memory, may be NULL) is added to them at run time — snr_step_state.offset_base of a captured step.
 * target [n,3] non-NULL = training: loss[0] += mse(rgb, target) + mse(rgb0, target) terms over 3 * n_rays_global
 * elements, loss[1] += the final map's term alone (zero both first); the workspace then holds what the backward needs.
 * Outputs (all required): rgb/disp/acc/depth maps of the final pass, rgb0/disp0/acc0 of the coarse pass and z_std [n]
 * (written only when n_importance > 0). */
int snr_render_rays_fused_forward(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                  const float* rays, int ray_ld, int64_t n_rays, const float* t_rand, const float* u,
                                  const float* noise0, const float* noise, uint64_t seed, uint64_t offset,
                                  const uint64_t* offset_base, const float* target, int64_t n_rays_global, void* ws, float* rgb_map, float* disp_map,
                                  float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0, float* z_std,
                                  float* loss, snr_stream_t stream);
/* The same call with the loss as a LIST OF TERMS over ray ranges (round 5: the SPIn-NeRF iteration, run_nerf.py:1455-1521, as
 * one launch sequence — its three renders are one render of the concatenated rays, and its loss is
 *   mse(rgb, target_clf) + mse(rgb0, target_clf)                         rays of the unmasked pixels     SNR_LOSS_RGB
 *   mse(rgb, target_s) + mse(rgb0, target_s), detach_weights=True        rays of all pixels              SNR_LOSS_RGB_DETACHED
 *   MSE(disp, depth_inp) + MSE(disp0, depth_inp), dropped when NaN       rays with an inpainted depth    SNR_LOSS_DISP, guarded).
 * A term covers rays [first_ray, first_ray + n_rays); its mean runs over `count` rays (the GLOBAL batch of a data-parallel
 * step); both passes add their part to loss[slot], the final pass also to loss[slot_final] (-1: nowhere).  Rays no term
 * covers get no gradient.  guard_term >= 0 names a term with slot != 0 that follows run_nerf.py:1518-1521: once both passes
 * have run, a NaN in loss[slot] zeroes that term's rays in the workspace's d raw tensors (the term and its gradient are
 * dropped), otherwise loss[0] += loss[slot].  `loss` has 4 slots, zeroed by the caller (snr_render_step_prepare with
 * SNR_RENDER_LOSS4 in cfg->flags does it). */
#define SNR_LOSS_RGB 0
#define SNR_LOSS_RGB_DETACHED 1
#define SNR_LOSS_DISP 2
typedef struct snr_loss_term {
  int64_t first_ray, n_rays;
  int kind;              /* SNR_LOSS_* */
  const float* target;   /* [n_rays, 3] (rgb kinds) or [n_rays] (disparity), device memory */
  int64_t count;         /* rays in the term's mean */
  int slot, slot_final;
} snr_loss_term;
typedef struct snr_loss_terms {
  int n_terms;           /* 1..4 */
  snr_loss_term term[4];
  int guard_term;        /* -1 or the index of the NaN-guarded term */
} snr_loss_terms;
int snr_render_rays_fused_forward_terms(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                        const float* rays, int ray_ld, int64_t n_rays, const float* t_rand, const float* u,
                                        const float* noise0, const float* noise, uint64_t seed, uint64_t offset,
                                        const uint64_t* offset_base, const snr_loss_terms* terms, void* ws, float* rgb_map,
                                        float* disp_map, float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0,
                                        float* z_std, float* loss, snr_stream_t stream);
/* The head of a training step as ONE launch (pack_rays + stratified sampling + the zero fill of the loss accumulator were
 * three): the packed ray rows of render(rays=...)'s plain case (run_nerf.py:117-153: no c2w_staticcam, scalar near / far, no
 * depth column), the stratified z_vals of the coarse samples (run_nerf.py:646-668; t_rand [n, n_samples] when non-NULL, else
 * the Philox draws of offset + 1 like the fused forward when cfg->perturb) written to `z_coarse` = the forward's workspace +
 * snr_render_ws_layout.z_coarse, and loss[0] = loss[1] = 0.  The forward that follows is called with
 * SNR_RENDER_Z_COARSE_READY set in cfg->flags. */
int snr_render_step_prepare(const snr_render_config* cfg, const float* rays_o, const float* rays_d, int64_t n_rays, int H, int W,
                            float focal, int ndc, float near, float far, int use_viewdirs, float* rays, int ray_ld,
                            const float* t_rand, uint64_t seed, uint64_t offset, const uint64_t* offset_base, float* z_coarse,
                            float* loss, snr_stream_t stream);
/* parameter gradients of the training forward above (same cfg / networks / rays / ws): grad_coarse and grad_fine (flat
 * fp32) overwritten if accumulate == 0, else += ; fine == NULL: both passes accumulate into grad_coarse.
 * passes: SNR_PASS_FINE | SNR_PASS_COARSE — a data-parallel caller runs the fine pass, starts that network's all-reduce
 * and then runs the coarse pass.  Both passes asked for together run as ONE launch sequence (snr_mlp_backward_multi). */
#define SNR_PASS_COARSE 1
#define SNR_PASS_FINE 2
int snr_render_rays_fused_backward(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                   const float* rays, int ray_ld, int64_t n_rays, void* ws, float* grad_coarse,
                                   float* grad_fine, int accumulate, int passes, snr_stream_t stream);

/* ---- rays: replaces get_rays + ndc_rays + the ray packing of render() (helpers:249-300,
 * run_nerf.py:117-153).  Writes rows [o(3) d(3) near far (viewdirs(3))] for the pixel rectangle
 * [i0,i0+h) x [j0,j0+w) of an H x W pinhole camera with pose c2w_host (12 floats, HOST memory). */
int snr_make_rays(int H, int W, float focal, const float* c2w_host, int i0, int j0, int h, int w, int ndc,
                  float near, float far, int use_viewdirs, float* rays, int ray_ld, snr_stream_t stream);

/* Same rows from rays the caller already holds (render(rays=...), run_nerf.py:117-153 incl. the viewdir
 * normalisation :128-135 and ndc_rays :140): rays_o, rays_d [n_rays,3] contiguous device memory.  Optional (NULL =
 * absent): view_src [n_rays,3] = directions the viewdirs are taken from instead of rays_d (c2w_staticcam, :131-133);
 * near_rows / far_rows [n_rays] = per-ray bounds instead of the scalars (:106-107); depths [n_rays] = the COLMAP depth
 * column inserted in front of the viewdirs (:148-149).  ndc_near = the near plane of ndc_rays (render() passes 1).
 * Row = o(3) d(3) near far [depth] [viewdirs(3)]. */
int snr_pack_rays(const float* rays_o, const float* rays_d, const float* view_src, int64_t n_rays, int H, int W,
                  float focal, int ndc, float ndc_near, float near, float far, const float* near_rows,
                  const float* far_rows, const float* depths, int use_viewdirs, float* rays, int ray_ld,
                  snr_stream_t stream);

/* ---- positional encoding as a standalone op: replaces Embedder.embed (helpers:22-52) for callers of
 * get_embedder()[0]; x [n, n_cols] -> out [n, n_cols * (1 + 2 * multires)] = [x, sin(2^k x), cos(2^k x), k < multires].
 * (The MLP entry points fuse the encoding and never materialise it.) */
int snr_embed(const float* x, int64_t n, int n_cols, int multires, float* out, snr_stream_t stream);

/* ---- loss of one training step: replaces img2mse(rgb, target) [+ img2mse(rgb0, target)] and its autograd
 * (helpers:15, run_nerf.py:1482-1490).  a, b (may be NULL), target: n elements each.  loss[0] = the sum of the
 * two means, loss[1] = mean((a - target)^2) alone; grad_a / grad_b = d loss / d a, d b. */
int snr_mse_pair(const float* a, const float* b, const float* target, int64_t n, float* loss, float* grad_a,
                 float* grad_b, snr_stream_t stream);

/* ---- Adam on the flat parameter buffer: replaces torch.optim.Adam(lr, betas=(0.9,0.999))
 * (run_nerf.py:433-434, 1611-1612).  step is 1-based. grad_scale multiplies g first
 * (1/world_size for data-parallel sums). */
int snr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int step, float grad_scale, snr_stream_t stream);

/* Adam AND the re-pack of the weights in one launch, for up to two networks (the coarse and the fine MLP of one step):
 * replaces optimizer.step() (run_nerf.py:1611-1612) + the snr_mlp_pack the next forward would need.  The update is
 * snr_adam_step's, bit for bit; `packed` is rewritten in place wherever it holds parameters and must have been produced by
 * snr_mlp_pack (same cfg) once before — its zero padding is not written again.  state NULL: rate `lr` and bias corrections
 * from `step` (1-based); else lr / bc1 / bc2_sqrt are read from the device-side snr_step_state and lr / step are ignored.
 * Two items with the same cfg run as one launch. */
typedef struct snr_adam_pack_item {
  const snr_mlp_config* cfg;
  float* params;
  const float* grads;
  float* exp_avg;
  float* exp_avg_sq;
  void* packed;
} snr_adam_pack_item;
int snr_adam_pack_multi(const snr_adam_pack_item* items, int n_items, float lr, float beta1, float beta2, float eps, int step,
                        float grad_scale, const snr_step_state* state, snr_stream_t stream);

/* ---- diagnostics: per-kernel HIP-event timing (used by bench.py for the roofline line) ----
 * When enabled, every kernel launch of this library is bracketed by hipEvents on its stream;
 * snr_prof_read() waits for them, returns total elapsed ms and launch count per kernel id
 * (arrays of snr_prof_kernel_count() entries) and clears the log.  Process-global, off by default. */
int snr_prof_enable(int on);
/* re-read the SNR_* environment switches (they are read once, at the first call that needs them): for tests and A/B
 * scripts that change the environment inside one process.  Not to be called while launches are being enqueued. */
int snr_tunables_reload(void);
int snr_prof_kernel_count(void);
const char* snr_prof_kernel_name(int id);
int snr_prof_read(double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SPINNERF_HIP_H */
