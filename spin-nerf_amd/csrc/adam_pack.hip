// Adam + weight pack as ONE kernel, for up to two networks per launch (gfx950).
//
// Replaces, per optimisation step, torch.optim.Adam.step() on both MLPs (DS_NeRF/run_nerf.py:433-434, 1611-1612) AND the
// re-pack of their weights into MFMA fragment order (mlp_fwd.hip: mlp_pack_kernel) that the next forward needs: four
// launches (2 x adam_kernel, 2 x mlp_pack_kernel: 0.038 ms of a 1.1 ms step, each a latency-bound few microseconds) become
// one.  Built with -ffp-contract=off like render_ops.hip: the update must round exactly like adam_kernel there (the captured
// graph route still uses it, and the two routes are held bit-identical).
//
// Work unit = a PANEL: one 32-row tile of one weight matrix (the rows of one forward pack entry).  A workgroup
//   1. runs Adam on the panel's 32 x ld parameters (coalesced along the rows), writes p / m / v back and keeps the new
//      parameters in LDS;
//   2. writes every FORWARD fragment of that row tile (the chunk [entry][tile] of the blob: whole 1 KiB fragments);
//   3. writes every TRANSPOSED (dgrad) fragment whose k-slots are this panel's rows: for a 256-wide source the two
//      fragments 2 t, 2 t + 1 of each of the transposed entry's row tiles — again whole fragments, 16 bytes per lane.
// Every parameter belongs to exactly one panel (or to the bias block, handled by one more workgroup per network), every
// fragment of the blob that holds parameters is written by exactly one panel; the blob's padding (rows / slots beyond the
// valid ones, the block alignment gaps) is never touched: the blob must have been produced by snr_mlp_pack once before.
#include "snr_common.h"
#include "mlp_pack.h"

namespace snr {

constexpr int kMaxPanels = 96;
constexpr int kPanelCols = 320;  // columns kept per panel row (>= the widest weight matrix: 256 + 63)
constexpr int kPanelLd = 321;    // LDS row length in floats: odd, so that the 32 rows a forward fragment's lanes read (same
                                 // column, stride one row) fall into 32 different banks (a stride of 320 put them all into one)

struct AdamPackNet {
  float* p;
  const float* g;
  float* m;
  float* v;
  char* blob;
};
struct AdamPackArgs {
  PackTable T;                 // one table: the networks of a launch share their configuration
  int n_nets, n_panels;
  AdamPackNet net[2];
  float lr, b1, b2, eps, bc1, bc2_sqrt, gscale;
  const snr_step_state* st;    // non-null: rate and bias corrections of the step from device memory (captured graphs)
  unsigned char panel_entry[kMaxPanels], panel_tile[kMaxPanels];
  int orphan_begin, orphan_count;   // parameters no pack entry reads (views_linears.0 of a network without view directions): Adam only
};

struct AdamCoef { float lr, b1, b2, eps, bc1, bc2_sqrt, gscale; };

// the update of adam_kernel (render_ops.hip), expression for expression
__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamCoef& c) {
  const float gk = g * c.gscale;
  m = m * c.b1 + (1.f - c.b1) * gk;
  v = v * c.b2 + (1.f - c.b2) * gk * gk;
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  return p - (c.lr / c.bc1) * (m / denom);
}

template <int P> __device__ __forceinline__ int slot_of(int kind, int f, int g, int e, int L_pts, int L_dir) {
  if (kind == SRC_ENC_PTS) return enc_slot_feature<P>(f, g, e, L_pts);
  if (kind == SRC_ENC_DIR) return enc_slot_feature<P>(f, g, e, L_dir);
  if (kind == SRC_H) return h_slot_neuron<P>(f, g, e);
  return (P == kBF16) ? 8 * g + e : 2 * e + g;   // SRC_OUT: raw channel
}

constexpr int kApThreads = 1024, kApWaves = kApThreads / 64;   // 16 waves per panel: this kernel is latency (a few dependent
                                                                // load -> LDS -> store rounds); it needs waves, not work per wave
template <int P>
__global__ __launch_bounds__(kApThreads) void adam_pack_kernel(AdamPackArgs a) {
  using Frag = typename Mma<P>::Frag;
  constexpr int EPF = Prec<P>::EPF, FPT = Prec<P>::FPT;
  __shared__ float W[32 * kPanelLd];
  const int per_net = a.n_panels + 1;
  const AdamPackNet& N = a.net[blockIdx.x / per_net];
  const int pi = blockIdx.x % per_net;
  const int tid = threadIdx.x;
  AdamCoef c{a.lr, a.b1, a.b2, a.eps, a.bc1, a.bc2_sqrt, a.gscale};
  if (a.st) { c.lr = a.st->lr; c.bc1 = a.st->bc1; c.bc2_sqrt = a.st->bc2_sqrt; }
  const PackTable& T = a.T;
  const int total_frags = T.fwd_frags + T.bwd_frags;

  if (pi == a.n_panels) {
    // ---- the bias block: every bias parameter once; padding entries of the block stay zero ----
    float* bias = (float*)(N.blob + (int64_t)total_frags * 1024);
    // (as in the panels: every load first, then the arithmetic and the stores — one dependent load -> store round per
    //  batch of biases would serialise this workgroup)
    constexpr int NB = 3;    // >= bias_floats / kApThreads (2496 floats with view directions)
    int gi[NB];
    float pv[NB], gv[NB], mv[NB], vv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int idx = tid + kApThreads * k;
      gi[k] = -1;
      if (idx < T.bias_floats) {
        for (int b = 0; b < T.n_bias; ++b) {
          const int r = idx - T.b[b].dst;
          if (r >= 0 && r < T.b[b].count && r < T.b[b].n_valid) gi[k] = T.b[b].src + r;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const bool ok = gi[k] >= 0;
      pv[k] = ok ? N.p[gi[k]] : 0.f; gv[k] = ok ? N.g[gi[k]] : 0.f; mv[k] = ok ? N.m[gi[k]] : 0.f; vv[k] = ok ? N.v[gi[k]] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (gi[k] >= 0) {
        const float pn = adam_one(pv[k], gv[k], mv[k], vv[k], c);
        N.p[gi[k]] = pn; N.m[gi[k]] = mv[k]; N.v[gi[k]] = vv[k];
        bias[tid + kApThreads * k] = pn;
      }
    }
    for (int idx = tid; idx < a.orphan_count; idx += kApThreads) {
      const int gi = a.orphan_begin + idx;
      float m = N.m[gi], v = N.v[gi];
      const float pn = adam_one(N.p[gi], N.g[gi], m, v, c);
      N.p[gi] = pn; N.m[gi] = m; N.v[gi] = v;
    }
    return;
  }

  const PackEntry& E = T.e[a.panel_entry[pi]];
  const int tile = a.panel_tile[pi];
  const int n0 = 32 * tile;
  const int w_off = E.src[0].w_off, ld = E.src[0].ld;
  int nrows = E.rows_valid - n0;
  nrows = nrows > 32 ? 32 : nrows;

  // ---- 1. Adam on the panel, new parameters to LDS (rows beyond the valid ones: zero) ----
  // wave w takes rows 2 w, 2 w + 1, its lanes run along a row (coalesced); all loads of a row are issued before the first
  // store (the buffers may alias as far as the compiler knows: left in one loop, every iteration waits for its own loads)
  {
    const int lane_ = tid & 63, wave_ = tid >> 6;
    constexpr int NJ = kPanelCols / 64;
    // every load of this wave's rows first (2 rows x 5 column steps x 4 buffers in flight per lane), then the arithmetic and
    // the stores
    constexpr int RW = 32 / kApWaves;   // rows per wave
    float pv[RW][NJ], gv[RW][NJ], mv[RW][NJ], vv[RW][NJ];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int r = RW * wave_ + rr;
      const int64_t g0 = (int64_t)w_off + (int64_t)(n0 + r) * ld;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = lane_ + 64 * j;
        const bool ok = r < nrows && col < ld;
        pv[rr][j] = ok ? N.p[g0 + col] : 0.f;
        gv[rr][j] = ok ? N.g[g0 + col] : 0.f;
        mv[rr][j] = ok ? N.m[g0 + col] : 0.f;
        vv[rr][j] = ok ? N.v[g0 + col] : 0.f;
      }
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int r = RW * wave_ + rr;
      const int64_t g0 = (int64_t)w_off + (int64_t)(n0 + r) * ld;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = lane_ + 64 * j;
        const bool ok = r < nrows && col < ld;
        float pn = 0.f;
        if (ok) {
          pn = adam_one(pv[rr][j], gv[rr][j], mv[rr][j], vv[rr][j], c);
          N.p[g0 + col] = pn; N.m[g0 + col] = mv[rr][j]; N.v[g0 + col] = vv[rr][j];
        }
        W[r * kPanelLd + col] = pn;
      }
    }
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, g = lane >> 5;

  // ---- 2. the forward fragments of this row tile ----
  const int per_tile = E.src[0].ks + E.src[1].ks;
  for (int fl = wave; fl < per_tile; fl += kApWaves) {
    const int s = fl >= E.src[0].ks ? 1 : 0;
    const int f = s ? fl - E.src[0].ks : fl;
    const PackSrc& S = E.src[s];
    Frag out = Mma<P>::zero();
#pragma unroll
    for (int e = 0; e < EPF; ++e) {
      const int slot = slot_of<P>(S.kind, f, g, e, T.multires, T.multires_views);
      const float v = (i < nrows && slot >= 0) ? W[i * kPanelLd + S.col_off + slot] : 0.f;
      // (bf16 mode: the columns that multiply an encoding are fp16 — mlp_fwd.hip: mlp_pack_kernel, bit for bit)
      if (EncF16<P>::value && (S.kind == SRC_ENC_PTS || S.kind == SRC_ENC_DIR)) Mma<P>::set_f16(out, e, v);
      else Mma<P>::set(out, e, v);
    }
    const int64_t F = E.frag_begin + tile * per_tile + fl;
    *(Frag*)(N.blob + (F * 64 + lane) * 16) = out;
  }

  // ---- 3. the transposed fragments whose k-slots are this panel's rows ----
  for (int ei = 0; ei < T.n_entries; ++ei) {
    const PackEntry& ET = T.e[ei];
    if (!ET.transposed) continue;
    const int per_t = ET.src[0].ks + ET.src[1].ks;
    for (int s = 0; s < 2; ++s) {
      const PackSrc& S = ET.src[s];
      if (S.ks == 0 || S.w_off != w_off) continue;
      // fragments q of this source that hold the panel's rows: a 256- or 128-wide source keeps rows 32 t .. 32 t + 31 in
      // its fragments FPT t .. FPT t + FPT - 1; the raw-channel source is a single fragment (rows 0 .. n_valid - 1: tile 0)
      int q0, q1;
      if (S.kind == SRC_OUT) { q0 = 0; q1 = tile == 0 ? 1 : 0; }
      else { q0 = FPT * tile; q1 = q0 + FPT; if (q1 > S.ks) q1 = S.ks; }
      const int nq = q1 - q0;
      if (nq <= 0) continue;
      const int items = ET.n_tiles * nq;
      for (int it = wave; it < items; it += kApWaves) {
        const int kt = it / nq, q = q0 + it % nq;
        const int row = 32 * kt + i;   // input neuron (weight column S.col_off + row)
        Frag out = Mma<P>::zero();
#pragma unroll
        for (int e = 0; e < EPF; ++e) {
          const int slot = slot_of<P>(S.kind, q, g, e, T.multires, T.multires_views);
          const int n = slot - S.slot_off;   // weight row
          float v = 0.f;
          if (row < ET.rows_valid && n >= 0 && n < S.n_valid && n >= n0 && n < n0 + 32)
            v = W[(n - n0) * kPanelLd + S.col_off + row];
          Mma<P>::set(out, e, v);
        }
        const int64_t F = ET.frag_begin + kt * per_t + (s ? ET.src[0].ks : 0) + q;
        *(Frag*)(N.blob + (F * 64 + lane) * 16) = out;
      }
    }
  }
}

// beta^k by squaring, in double (render_ops.hip: powi — the same IEEE multiplications as the host side of snr_adam_step)
static double powi_host(double b, long long k) {
  double r = 1.0;
  while (k > 0) {
    if (k & 1) r *= b;
    b *= b;
    k >>= 1;
  }
  return r;
}

}  // namespace snr

using namespace snr;

static bool same_cfg(const snr_mlp_config* a, const snr_mlp_config* b) {
  return a->multires == b->multires && a->multires_views == b->multires_views && a->i_embed == b->i_embed &&
         a->use_viewdirs == b->use_viewdirs && a->out_ch == b->out_ch && a->precision == b->precision;
}

template <int P>
static int launch_adam_pack(const snr_adam_pack_item* items, int n, float lr, float b1, float b2, float eps, int step,
                            float gscale, const snr_step_state* st, hipStream_t s) {
  const snr_mlp_config* c = items[0].cfg;
  AdamPackArgs a{};
  a.T = make_pack_table<P>(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  int np = 0;
  for (int ei = 0; ei < a.T.n_entries; ++ei) {
    const PackEntry& E = a.T.e[ei];
    if (E.transposed) continue;
    if (E.src[0].ld > kPanelCols || (E.src[1].ks > 0 && E.src[1].w_off != E.src[0].w_off)) return SNR_ERR_UNSUPPORTED;
    for (int t = 0; t < E.n_tiles; ++t) {
      if (32 * t >= E.rows_valid) break;   // (a tile without valid rows holds no parameters)
      if (np >= kMaxPanels) return SNR_ERR_UNSUPPORTED;
      a.panel_entry[np] = (unsigned char)ei; a.panel_tile[np] = (unsigned char)t; ++np;
    }
  }
  a.n_panels = np; a.n_nets = n;
  if (a.T.bias_floats > 3 * kApThreads) return SNR_ERR_UNSUPPORTED;
  if (!c->use_viewdirs) {
    const ParamLayout L = make_param_layout(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
    a.orphan_begin = (int)L.w_views;
    a.orphan_count = (int)(L.b_views + kW / 2 - L.w_views);
  }
  for (int i = 0; i < n; ++i) a.net[i] = AdamPackNet{items[i].params, items[i].grads, items[i].exp_avg, items[i].exp_avg_sq, (char*)items[i].packed};
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.gscale = gscale; a.st = st;
  a.bc1 = (float)(1.0 - powi_host((double)b1, step));
  a.bc2_sqrt = (float)sqrt(1.0 - powi_host((double)b2, step));
  {
    ProfScope ps(K_ADAM_PACK, s);
    adam_pack_kernel<P><<<dim3((unsigned)(n * (np + 1))), dim3(kApThreads), 0, s>>>(a);
  }
  return launch_status();
}

extern "C" int snr_adam_pack_multi(const snr_adam_pack_item* items, int n_items, float lr, float beta1, float beta2, float eps,
                                   int step, float grad_scale, const snr_step_state* state, snr_stream_t stream) {
  SNR_CHECK_ARG(items, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_items >= 1 && n_items <= 2, SNR_ERR_SHAPE);
  SNR_CHECK_ARG(state || step >= 1, SNR_ERR_SHAPE);
  for (int i = 0; i < n_items; ++i) {
    const snr_adam_pack_item& it = items[i];
    SNR_CHECK_ARG(it.cfg && it.params && it.grads && it.exp_avg && it.exp_avg_sq && it.packed, SNR_ERR_NULL);
    if (it.cfg->precision != SNR_PREC_BF16 && it.cfg->precision != SNR_PREC_FP32) return SNR_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool together = n_items == 2 && same_cfg(items[0].cfg, items[1].cfg);
  for (int i = 0; i < n_items; i += together ? 2 : 1) {
    const int n = together ? 2 : 1;
    const int st = items[i].cfg->precision == SNR_PREC_BF16
                       ? launch_adam_pack<kBF16>(items + i, n, lr, beta1, beta2, eps, step, grad_scale, state, s)
                       : launch_adam_pack<kFP32>(items + i, n, lr, beta1, beta2, eps, step, grad_scale, state, s);
    if (st != SNR_OK) return st;
  }
  return SNR_OK;
}
