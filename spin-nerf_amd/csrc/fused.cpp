// render_rays as one C-ABI call (include/spinnerf_hip.h: snr_render_rays_fused_*): the library itself enqueues the
// launch sequence of run_nerf.py:593-737 (+ the loss terms and autograd's head in training mode) on the caller's stream,
// carving every intermediate out of one workspace.  No kernels here — only the order of the entry points of
// render_ops.hip / mlp_fwd.hip / mlp_bwd.hip / hashgrid.hip, so that a training step costs the host two calls instead of
// a dozen allocations and launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spinnerf_hip.h"
#include "render_internal.h"

namespace {

constexpr int64_t kAlign = 256;
int64_t up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

int net_ok(const snr_net* n) {
  if (!n) return SNR_ERR_NULL;
  if (n->kind != SNR_NET_MLP && n->kind != SNR_NET_HASHGRID) return SNR_ERR_UNSUPPORTED;
  if (!n->packed) return SNR_ERR_NULL;
  if (n->kind == SNR_NET_HASHGRID && !n->params) return SNR_ERR_NULL;
  return SNR_OK;
}
int out_ch(const snr_net* n) { return n->kind == SNR_NET_MLP ? (n->mlp.use_viewdirs ? 4 : n->mlp.out_ch) : 4; }
bool wants_viewdirs(const snr_net* n) { return n->kind == SNR_NET_HASHGRID || n->mlp.use_viewdirs; }
int64_t act_bytes(const snr_net* n, int64_t m) {
  return n->kind == SNR_NET_MLP ? snr_mlp_act_bytes(&n->mlp, m) : snr_hashgrid_act_bytes(m);
}
int64_t bwd_bytes(const snr_net* n, int64_t m) {
  return n->kind == SNR_NET_MLP ? snr_mlp_bwd_ws_bytes(&n->mlp, m) : snr_hashgrid_bwd_ws_bytes(m);
}

int net_forward(const snr_net* n, const float* rays, int ld, const float* z, int64_t n_rays, int S, float* raw, void* act,
                snr_stream_t s) {
  const float* vd = wants_viewdirs(n) ? rays + (ld - 3) : nullptr;   // the last three columns of a packed row
  if (n->kind == SNR_NET_MLP)
    return snr_mlp_forward(&n->mlp, n->packed, nullptr, rays, ld, z, vd, ld, n_rays * S, S, raw, act, s);
  return snr_hashgrid_forward(n->params, n->packed, nullptr, rays, ld, z, vd, ld, n_rays * S, S, raw, act, s);
}
int net_backward(const snr_net* n, const float* rays, int ld, const float* z, int64_t n_rays, int S, const float* d_raw,
                 const void* act, void* ws, float* grad, int accumulate, snr_stream_t s) {
  if (n->kind == SNR_NET_MLP)
    return snr_mlp_backward(&n->mlp, n->packed, n->params, d_raw, n_rays * S, act, ws, grad, accumulate, s);
  return snr_hashgrid_backward(n->params, n->packed, nullptr, rays, ld, z, rays + (ld - 3), ld, d_raw, n_rays * S, S, act,
                               ws, grad, accumulate, s);
}

int layout(const snr_render_config* c, const snr_net* nc, const snr_net* nf, int64_t n, int train, snr_render_ws_layout* L) {
  if (!c || !L) return SNR_ERR_NULL;
  int st = net_ok(nc);
  if (st != SNR_OK) return st;
  if (nf && (st = net_ok(nf)) != SNR_OK) return st;
  if (n <= 0 || c->n_samples < 2 || c->n_importance < 0) return SNR_ERR_SHAPE;
  const snr_net* f = nf ? nf : nc;
  const int Nc = c->n_samples, Nf = c->n_importance, S = Nc + Nf;
  const int C0 = out_ch(nc), C1 = out_ch(f);
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t at = o; o += up(bytes); return at; };
  *L = snr_render_ws_layout{-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0};
  L->z_coarse = take(n * Nc * 4);
  L->raw0 = take(n * Nc * C0 * 4);
  L->weights0 = take(n * Nc * 4);
  L->depth0 = take(n * 4);
  if (Nf > 0) {
    L->z_vals = take(n * S * 4);
    L->raw = take(n * S * C1 * 4);
    L->weights = take(n * S * 4);
    L->z_samples = take(n * Nf * 4);
  }
  if (train) {
    const int64_t a0 = act_bytes(nc, n * Nc), b0 = bwd_bytes(nc, n * Nc);
    if (a0 <= 0 || b0 <= 0) return (int)(a0 <= 0 ? a0 : b0);
    L->d_raw0 = take(n * Nc * C0 * 4);
    L->act0 = take(a0);
    int64_t bw = b0;
    if (Nf > 0) {
      const int64_t a1 = act_bytes(f, n * S), b1 = bwd_bytes(f, n * S);
      if (a1 <= 0 || b1 <= 0) return (int)(a1 <= 0 ? a1 : b1);
      L->d_raw = take(n * S * C1 * 4);
      L->act = take(a1);
      if (b1 > bw) bw = b1;
    }
    L->bwd_ws = take(bw);   // the final pass's (or the only pass's) backward scratch
    if (Nf > 0) L->bwd_ws0 = take(b0);   // the coarse pass's own: both backward passes run as one launch sequence
  }
  L->total = o;
  return SNR_OK;
}

}  // namespace

extern "C" int snr_render_rays_fused_layout(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                            int64_t n_rays, int train, snr_render_ws_layout* out) {
  return layout(cfg, coarse, fine, n_rays, train, out);
}

namespace {
// terms == nullptr: inference.  (validated: ranges inside [0, n_rays), slots 0..3, targets present)
int forward_impl(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine, const float* rays, int ray_ld,
                 int64_t n_rays, const float* t_rand, const float* u, const float* noise0, const float* noise, uint64_t seed,
                 uint64_t offset, const uint64_t* offset_base, const snr_loss_terms* terms, void* ws, float* rgb_map,
                 float* disp_map, float* acc_map, float* depth_map, float* rgb0, float* disp0, float* acc0, float* z_std,
                 float* loss, snr_stream_t stream) {
  snr_render_ws_layout L;
  const int train = terms != nullptr;
  int st = layout(cfg, coarse, fine, n_rays, train, &L);
  if (st != SNR_OK) return st;
  if (!rays || !ws || !rgb_map || !disp_map || !acc_map || !depth_map) return SNR_ERR_NULL;
  const int Nc = cfg->n_samples, Nf = cfg->n_importance, S = Nc + Nf;
  if (Nf > 0 && (!rgb0 || !disp0 || !acc0 || !z_std)) return SNR_ERR_NULL;
  snr::LossSpec spec{};
  if (train) {
    if (!loss) return SNR_ERR_NULL;
    if (terms->n_terms < 1 || terms->n_terms > 4 || terms->guard_term >= terms->n_terms) return SNR_ERR_SHAPE;
    spec.n = terms->n_terms;
    for (int k = 0; k < spec.n; ++k) {
      const snr_loss_term& t = terms->term[k];
      if (!t.target) return SNR_ERR_NULL;
      if (t.first_ray < 0 || t.n_rays <= 0 || t.first_ray + t.n_rays > n_rays || t.count < t.n_rays || t.slot < 0 || t.slot > 3 ||
          t.slot_final > 3 || t.kind < SNR_LOSS_RGB || t.kind > SNR_LOSS_DISP)
        return SNR_ERR_SHAPE;
      const float inv = t.kind == SNR_LOSS_DISP ? 1.f / (float)t.count : 1.f / (3.f * (float)t.count);
      spec.t[k] = snr::LossTerm{t.first_ray, t.n_rays, t.kind, t.target, inv, t.slot, t.slot_final < 0 ? snr::kNoSlot : (int64_t)t.slot_final};
    }
    if (terms->guard_term >= 0 && terms->term[terms->guard_term].slot == 0) return SNR_ERR_SHAPE;
  }
  const snr_net* f = fine ? fine : coarse;
  if (ray_ld < 8 + ((wants_viewdirs(coarse) || wants_viewdirs(f)) ? 3 : 0)) return SNR_ERR_SHAPE;
  char* w = (char*)ws;
  auto F = [&](int64_t off) { return (float*)(w + off); };
  const int C0 = out_ch(coarse), C1 = out_ch(f);
  const int last0 = Nf == 0;   // the coarse pass is the final one
  auto guard = [&]() {         // run_nerf.py:1518-1521, behind the final pass
    if (!train || terms->guard_term < 0) return (int)SNR_OK;
    const snr_loss_term& t = terms->term[terms->guard_term];
    return snr::loss_guard_impl(loss, t.slot, t.first_ray, t.n_rays, F(L.d_raw0), (int64_t)Nc * C0, last0 ? nullptr : F(L.d_raw),
                                (int64_t)S * C1, stream);
  };

  // ---- coarse pass (run_nerf.py:646-692) ----
  if (cfg->flags & SNR_RENDER_Z_COARSE_READY) st = SNR_OK;   // snr_render_step_prepare wrote the stratified z_vals
  else if (cfg->perturb && !t_rand) st = snr::sample_coarse_rng_impl(rays, ray_ld, n_rays, Nc, cfg->lindisp, seed, offset + 1, offset_base, F(L.z_coarse), stream);
  else st = snr_sample_coarse(rays, ray_ld, n_rays, Nc, cfg->lindisp, cfg->perturb ? t_rand : nullptr, F(L.z_coarse), stream);
  if (st != SNR_OK) return st;
  st = net_forward(coarse, rays, ray_ld, F(L.z_coarse), n_rays, Nc, F(L.raw0), train ? w + L.act0 : nullptr, stream);
  if (st != SNR_OK) return st;
  float* m_rgb = last0 ? rgb_map : rgb0;
  float* m_disp = last0 ? disp_map : disp0;
  float* m_acc = last0 ? acc_map : acc0;
  float* m_depth = last0 ? depth_map : F(L.depth0);
  bool sampled = false;   // the hierarchical sampling ran inside the compositing launch
  if (train && !last0) {
    // compositing + loss + its backward of the coarse samples AND hierarchical sampling + sort from the weights, one kernel.
    // It declines shapes it does not cover (SNR_ERR_UNSUPPORTED: Nc > 64 or Nc < 3, unaligned raw rows, a sample union beyond its LDS): only then the two-kernel route
    // below takes over — any other status is a real error and is returned
    st = snr::composite_train_sample_impl(F(L.raw0), C0, F(L.z_coarse), rays, ray_ld, noise0, cfg->raw_noise_std, seed, offset + 2,
                                          offset_base, n_rays, Nc, cfg->white_bkgd, spec, m_rgb, m_disp, m_acc,
                                          m_depth, F(L.weights0), F(L.d_raw0), loss, cfg->perturb ? u : nullptr,
                                          cfg->perturb && !u, offset + 3, Nf, F(L.z_vals), F(L.z_samples), z_std, stream);
    if (st != SNR_OK && st != SNR_ERR_UNSUPPORTED) return st;
    sampled = st == SNR_OK;
  }
  if (sampled) {
  } else if (train) {
    st = snr::composite_train_impl(F(L.raw0), C0, F(L.z_coarse), rays, ray_ld, noise0, cfg->raw_noise_std, seed, offset + 2,
                                   offset_base, n_rays, Nc, cfg->white_bkgd, spec, last0, m_rgb, m_disp, m_acc, m_depth,
                                   F(L.weights0), F(L.d_raw0), loss, stream);
  } else {
    if (cfg->raw_noise_std > 0.f && !noise0) return SNR_ERR_UNSUPPORTED;   // inference renders without density noise
    st = snr_composite_forward(F(L.raw0), C0, F(L.z_coarse), rays, ray_ld, noise0, n_rays, Nc, cfg->white_bkgd, m_rgb, m_disp,
                               m_acc, m_depth, F(L.weights0), nullptr, stream);
  }
  if (st != SNR_OK) return st;
  if (last0) return guard();

  // ---- hierarchical sampling + fine pass (run_nerf.py:694-713) ----
  if (sampled) st = SNR_OK;
  else if (cfg->perturb && !u) st = snr::sample_fine_rng_impl(F(L.z_coarse), F(L.weights0), n_rays, Nc, Nf, seed, offset + 3, offset_base, F(L.z_vals), F(L.z_samples), z_std, stream);
  else st = snr_sample_fine(F(L.z_coarse), F(L.weights0), cfg->perturb ? u : nullptr, n_rays, Nc, Nf, F(L.z_vals), F(L.z_samples), z_std, stream);
  if (st != SNR_OK) return st;
  st = net_forward(f, rays, ray_ld, F(L.z_vals), n_rays, S, F(L.raw), train ? w + L.act : nullptr, stream);
  if (st != SNR_OK) return st;
  if (train) {
    st = snr::composite_train_impl(F(L.raw), C1, F(L.z_vals), rays, ray_ld, noise, cfg->raw_noise_std, seed, offset + 4, offset_base,
                                   n_rays, S, cfg->white_bkgd, spec, 1, rgb_map, disp_map, acc_map, depth_map, F(L.weights),
                                   F(L.d_raw), loss, stream);
    return st != SNR_OK ? st : guard();
  }
  if (cfg->raw_noise_std > 0.f && !noise) return SNR_ERR_UNSUPPORTED;
  return snr_composite_forward(F(L.raw), C1, F(L.z_vals), rays, ray_ld, noise, n_rays, S, cfg->white_bkgd, rgb_map, disp_map,
                               acc_map, depth_map, F(L.weights), nullptr, stream);
}
}  // namespace

extern "C" int snr_render_rays_fused_forward(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                             const float* rays, int ray_ld, int64_t n_rays, const float* t_rand,
                                             const float* u, const float* noise0, const float* noise, uint64_t seed,
                                             uint64_t offset, const uint64_t* offset_base, const float* target, int64_t n_rays_global, void* ws,
                                             float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* rgb0,
                                             float* disp0, float* acc0, float* z_std, float* loss, snr_stream_t stream) {
  if (!target)
    return forward_impl(cfg, coarse, fine, rays, ray_ld, n_rays, t_rand, u, noise0, noise, seed, offset, offset_base, nullptr, ws,
                        rgb_map, disp_map, acc_map, depth_map, rgb0, disp0, acc0, z_std, loss, stream);
  if (n_rays_global < n_rays) return SNR_ERR_SHAPE;
  // one term: mse(rgb, target) + mse(rgb0, target) over the global batch into loss[0], the final map's part also into loss[1]
  snr_loss_terms one{};
  one.n_terms = 1;
  one.term[0] = snr_loss_term{0, n_rays, SNR_LOSS_RGB, target, n_rays_global, 0, 1};
  one.guard_term = -1;
  return forward_impl(cfg, coarse, fine, rays, ray_ld, n_rays, t_rand, u, noise0, noise, seed, offset, offset_base, &one, ws,
                      rgb_map, disp_map, acc_map, depth_map, rgb0, disp0, acc0, z_std, loss, stream);
}

extern "C" int snr_render_rays_fused_forward_terms(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                                   const float* rays, int ray_ld, int64_t n_rays, const float* t_rand,
                                                   const float* u, const float* noise0, const float* noise, uint64_t seed,
                                                   uint64_t offset, const uint64_t* offset_base, const snr_loss_terms* terms, void* ws,
                                                   float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* rgb0,
                                                   float* disp0, float* acc0, float* z_std, float* loss, snr_stream_t stream) {
  if (!terms) return SNR_ERR_NULL;
  return forward_impl(cfg, coarse, fine, rays, ray_ld, n_rays, t_rand, u, noise0, noise, seed, offset, offset_base, terms, ws,
                      rgb_map, disp_map, acc_map, depth_map, rgb0, disp0, acc0, z_std, loss, stream);
}

extern "C" int snr_render_rays_fused_backward(const snr_render_config* cfg, const snr_net* coarse, const snr_net* fine,
                                              const float* rays, int ray_ld, int64_t n_rays, void* ws, float* grad_coarse,
                                              float* grad_fine, int accumulate, int passes, snr_stream_t stream) {
  snr_render_ws_layout L;
  int st = layout(cfg, coarse, fine, n_rays, 1, &L);
  if (st != SNR_OK) return st;
  if (!rays || !ws) return SNR_ERR_NULL;
  const int Nc = cfg->n_samples, Nf = cfg->n_importance, S = Nc + Nf;
  if ((passes & ~(SNR_PASS_COARSE | SNR_PASS_FINE)) || passes == 0) return SNR_ERR_SHAPE;
  const bool do_fine = (passes & SNR_PASS_FINE) && Nf > 0, do_coarse = passes & SNR_PASS_COARSE;
  if ((do_coarse || (do_fine && !fine)) && !grad_coarse) return SNR_ERR_NULL;
  if (do_fine && fine && !grad_fine) return SNR_ERR_NULL;
  if (coarse->kind == SNR_NET_MLP && coarse->mlp.use_viewdirs && !coarse->params) return SNR_ERR_NULL;
  char* w = (char*)ws;
  auto F = [&](int64_t off) { return (float*)(w + off); };
  char* ws0 = w + (Nf > 0 ? L.bwd_ws0 : L.bwd_ws);   // the coarse pass's backward scratch
  if (do_fine && do_coarse && fine && coarse->kind == SNR_NET_MLP && fine->kind == SNR_NET_MLP) {
    // both MLP backward passes as one launch sequence
    const snr_mlp_bwd_item items[2] = {
        {&fine->mlp, fine->packed, fine->params, F(L.d_raw), n_rays * S, w + L.act, w + L.bwd_ws, grad_fine, accumulate},
        {&coarse->mlp, coarse->packed, coarse->params, F(L.d_raw0), n_rays * Nc, w + L.act0, ws0, grad_coarse, accumulate}};
    return snr_mlp_backward_multi(items, 2, stream);
  }
  int acc_c = accumulate;
  if (do_fine) {   // the fine pass first: autograd's order, and (data-parallel) its all-reduce can start under the coarse pass
    const snr_net* f = fine ? fine : coarse;
    float* gf = fine ? grad_fine : grad_coarse;
    st = net_backward(f, rays, ray_ld, F(L.z_vals), n_rays, S, F(L.d_raw), w + L.act, w + L.bwd_ws, gf, accumulate, stream);
    if (st != SNR_OK) return st;
    if (!fine) acc_c = 1;
  }
  if (!do_coarse) return SNR_OK;
  return net_backward(coarse, rays, ray_ld, F(L.z_coarse), n_rays, Nc, F(L.d_raw0), w + L.act0, ws0, grad_coarse, acc_c,
                      stream);
}

extern "C" int snr_render_step_prepare(const snr_render_config* cfg, const float* rays_o, const float* rays_d, int64_t n_rays, int H,
                                       int W, float focal, int ndc, float near, float far, int use_viewdirs, float* rays, int ray_ld,
                                       const float* t_rand, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                                       float* z_coarse, float* loss, snr_stream_t stream) {
  if (!cfg) return SNR_ERR_NULL;
  if (cfg->n_samples < 2) return SNR_ERR_SHAPE;
  return snr::pack_rays_sample_impl(rays_o, rays_d, n_rays, H, W, focal, ndc, near, far, use_viewdirs, rays, ray_ld, cfg->n_samples,
                                    cfg->lindisp, cfg->perturb ? t_rand : nullptr, cfg->perturb && !t_rand, seed, offset + 1,
                                    offset_base, z_coarse, loss, loss ? ((cfg->flags & SNR_RENDER_LOSS4) ? 4 : 2) : 0, stream);
}
