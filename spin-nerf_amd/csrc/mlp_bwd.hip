// Backward of the fused NeRF MLP for gfx950: d(loss)/d(params) from d(loss)/d(raw).
//
// Replaces torch autograd through NeRF.forward (DS_NeRF/run_nerf_helpers.py:104-127) as driven by
// loss.backward() in train() (DS_NeRF/run_nerf.py:1611).  No gradient is propagated into the
// encodings / sample positions (nothing upstream needs one).
//
// Three kernels:
//   1. dgrad chain  — same structure as the forward (ChainCfg waves x 32-sample tiles, activations in
//      registers, W^T blocks DMA'd through LDS): d z_{i-1}^T = relu'(h_{i-1}) * (W_i^T d z_i^T).
//      ReLU flags come from the 1 bit per activation the forward saved (asm loads with counted
//      completion); every d z_i is written (bf16 / fp32 fragments, asm streaming stores) for the
//      weight-gradient pass.
//   2. wgrad        — dW_i[n][k] = sum_s d z_i[s][n] h_{i-1}[s][k]: a contraction over SAMPLES, so
//      both operands need 8 consecutive samples per lane while the saved fragments hold 8
//      consecutive neurons per lane.  Tiles are DMA'd to LDS unchanged and transposed on the way
//      out with ds_read_b64_tr_b16 (bf16, inline asm) / per-lane ds_read_b32 (fp32).  Each workgroup owns
//      the whole (<= 256 x 256) output of one layer for a slice of the samples (split-K, exactly one
//      workgroup per CU); the kernel is HBM-bound (11.4 KB streamed per 1.19 MFLOP).
//   3. reduce       — sums the split-K partials and scatters them from fragment order to the
//      reference's [out, in] parameter layout (+ bias gradients).
#include <type_traits>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"
#include "mlp_wgrad.h"
#include "mlp_wgrad_pair.h"

namespace snr {

// ------------------------------------------------------------------------------------------
// 1. dgrad chain
// ------------------------------------------------------------------------------------------
struct DgradArgs {
  const char* blob_bwd;   // dgrad blocks (behind the forward section)
  int bwd_blocks;
  const float* d_raw;     // [n, out_ch]
  int64_t n_samples;
  int out_ch;
  const char* act;        // forward workspace (masks)
  char* ws;               // d z sections
  int save_even;          // 0: d z0, d z2, d z4, d z6 are not saved (rebuilt by the weight-gradient pass)
  unsigned* zero_words;   // progress words of the layer-pair launch that follows (mlp_wgrad_pair.h), cleared here; or null
  int n_zero;
  int block0, blocks;     // this network's workgroups of the launch: [block0, block0 + blocks), grid-striding over its tiles
};
// (round 4) one launch runs the chains of up to two networks (coarse + fine backward): a workgroup belongs to one network
// for its whole life (its weight stream is that network's)
struct DgradMulti { int n; DgradArgs net[kMaxReduceNets]; };

// SAVE_EVEN = false (bf16 with selective recompute): d z0, d z2, d z4, d z6 are not written, and d z0 is not even computed —
// nothing in this kernel consumes it, and the weight-gradient pass rebuilds it from d z1 (round 5: the last 256 x 256 stage of
// the chain, a ninth of its MFMAs and of its weight stream, used to be computed and dropped).
template <int P, bool VD, bool SAVE_EVEN>
__global__ __launch_bounds__((64 * ChainCfg<P, true>::WAVES)) void mlp_dgrad_kernel(DgradMulti mm) {
  const DgradArgs& a = mm.net[(mm.n > 1 && (int)blockIdx.x >= mm.net[1].block0) ? 1 : 0];
  using B = Blob<P>;
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int FPT = Prec<P>::FPT, EPF = Prec<P>::EPF, NJ = ChainCfg<P, true>::NJ, WAVES = ChainCfg<P, true>::WAVES;
  constexpr int KS_H = B::KS_H, KS_H9 = B::KS_H9;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sj = lane & 31, g = lane >> 5;

  Pipe<P, WAVES> pipe;
  pipe.init(smem, a.blob_bwd, a.bwd_blocks, wave, lane);

  const ActLayout<P> AL(a.n_samples, VD);
  const WsLayout<P> WL(a.n_samples, VD);
  const int64_t n_wg = AL.n_tiles / (WAVES * NJ);
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  using IH = std::integral_constant<int, KS_H>;
  using IH9 = std::integral_constant<int, KS_H9>;

  if ((int)blockIdx.x == a.block0 && a.zero_words && tid < a.n_zero) a.zero_words[tid] = 0u;
  for (int64_t wg = (int)blockIdx.x - a.block0; wg < n_wg; wg += a.blocks) {
    const int64_t tile0 = (wg * WAVES + wave) * NJ;

    // frags [n*nt/NT, n*(nt+1)/NT) of an n-frag section for every sample tile of this wave
    // (uniform base) + (lane offset) addressing, section bases derived at the point of use: mlp_fwd.hip
    const uint32_t lane_even = g * 16 + sj * 32, lane_odd = g * 16 + act_row<P>(sj, 1) * 32;
    auto ws_store_par = [&](auto PAR_, int k_sec, auto N_, const Frag* src, auto STRIDE_, int nt, auto NT_) {
      constexpr int n = decltype(N_)::value, stride = decltype(STRIDE_)::value, NT = decltype(NT_)::value;
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) {
        int64_t tl = WL.n_tiles;
        asm volatile("" : "+s"(tl));
        store_tile_slice<P, n, NT, decltype(PAR_)::value>(a.ws + (tl * k_sec + (tile0 + jt) * n) * 1024, src + jt * stride, nt, lane_even, lane_odd);
      }
    };
    // a saved d z section's even fragments leave from the stage that produces them, the odd ones from the next stage (mlp_device.h)
    constexpr bool kSplit = SNR_STORE_SPLIT && P == kBF16;
    using PALL = std::integral_constant<int, -1>;
    using PEVEN = std::integral_constant<int, 0>;
    using PODD = std::integral_constant<int, kSplit ? 1 : -1>;
    auto ws_store = [&](int k_sec, auto N_, const Frag* src, auto STRIDE_, int nt, auto NT_) {
      ws_store_par(PALL{}, k_sec, N_, src, STRIDE_, nt, NT_);
    };
    // ReLU flags of a stage, 16 B per lane.  Also asm (a load the compiler sees is waited for with vmcnt(0) at
    // its first use if anything else is pending).  Completion: loads retire in order, so once at most W
    // younger operations are outstanding and more than W DMA pieces were issued behind the load, it has
    // landed — masks_ready<W>() says so to the compiler.  Every load below sits at least one long stage
    // (>= 4 blocks = 8 pieces) ahead of its use, where W = 6 is what the block waits enforce anyway.
    auto mask_load = [&](int k_sec, u32x4* mk) {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) {
        int64_t tl = AL.n_tiles;
        asm volatile("" : "+s"(tl));
        const char* base = a.act + (tl * k_sec + tile0 + jt) * 1024;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(mk[jt]) : "v"(lane * 16u), "s"(base));   // s_nop: store16_stream
      }
    };
    auto masks_ready = [&](auto W_, u32x4* mk) {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(mk[jt]) : "n"(decltype(W_)::value));
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W6 = std::integral_constant<int, 6>;

    // ---- d raw -> OUT frag (bf16: k-slot 8g+e = channel; fp32: k-slot g of step e = channel 2e+g)
    Frag dout[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
      dout[jt] = M::zero();
      const int64_t m = (tile0 + jt) * 32 + sj;
      if (m < a.n_samples) {
        const float* dr = a.d_raw + m * a.out_ch;
#pragma unroll
        for (int e = 0; e < EPF; ++e) {
          const int ch = (P == kBF16) ? 8 * g + e : 2 * e + g;
          if (ch < a.out_ch) M::set(dout[jt], e, dr[ch]);
        }
      }
    }

    Frag hA[NJ][KS_H], hB[NJ][KS_H];
    u32x4 mk_cur[NJ], mk_next[NJ];
    auto roll_masks = [&](auto W_) {
      masks_ready(W_, mk_next);
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) mk_cur[jt] = mk_next[jt];
    };

    // generic stage: dst = mask * (W^T [sa|sb]); NT output tiles
    // k_out >= 0: the stage's output is a saved section of NOUT fragments per tile
    auto stage = [&](auto KA_, auto KB_, auto NT_, auto SA_, auto SB_, const Frag* sa, const Frag* sb, Frag* dst,
                     bool use_mask, auto&& pre, int k_out, auto NOUT_) {
      constexpr int KA = decltype(KA_)::value, KB = decltype(KB_)::value, NT = decltype(NT_)::value;
      constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value;
      pipe.template run_tiles<KA, KB, NT, NJ, SA, SB>(
          sa, sb, [&](int) { return zero16; },
          [&](int nt, int jt, f32x16 acc) {
            Frag* out = dst + jt * KS_H + nt * FPT;
            if constexpr (P == kBF16) {   // flag layout: mlp_device.h
              finish_dgrad_bf16(acc, out, use_mask, mk_cur[jt][nt >> 1], 8 * (nt & 1));
              if constexpr (kSplit) {
                if (k_out >= 0 && jt == NJ - 1) ws_store_par(PEVEN{}, k_out, NOUT_, dst, IH{}, nt, NT_);
              }
            } else {
              if (use_mask) {
                const unsigned bits = mk_cur[jt][nt >> 1] >> (16 * (nt & 1));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = ((bits >> r) & 1u) ? acc[r] : 0.f;
              }
              acc_to_frags<P>(acc, out);
            }
          },
          pre);
    };

    if constexpr (VD) {
      mask_load(AL.k_mask9(), mk_cur);
      mask_load(AL.k_mask(7), mk_next);
      masks_ready(W1{}, mk_cur);   // used right away: everything but the one younger load has to be back
      // d z9 = relu'(h9) * (W_rgb^T d rgb)                 -> hB[.][0..KS_H9)
      stage(I1{}, I0{}, I4{}, I1{}, I1{}, &dout[0], &dout[0], &hB[0][0], true,
            [&](int nt) { ws_store(WL.k_dout(), I1{}, &dout[0], I1{}, nt, I4{}); }, WL.k_dz9(), IH9{});
      // d feat = W_views[:, :256]^T d z9                    -> hA
      stage(IH9{}, I0{}, I8{}, IH{}, IH{}, &hB[0][0], &hB[0][0], &hA[0][0], false,
            [&](int nt) { ws_store_par(PODD{}, WL.k_dz9(), IH9{}, &hB[0][0], IH{}, nt, I8{}); }, -1, IH{});
      // d z7 = relu'(h7) * (W_feat^T d feat + W_alpha^T d alpha) -> hB
      roll_masks(W6{});
      mask_load(AL.k_mask(6), mk_next);
      stage(IH{}, I1{}, I8{}, IH{}, I1{}, &hA[0][0], &dout[0], &hB[0][0], true,
            [&](int) {}, WL.k_dz(7), IH{});   // d feat stays in registers: its weight gradient follows from d z9 (mlp_wgrad.h)
    } else {
      mask_load(AL.k_mask(7), mk_cur);
      mask_load(AL.k_mask(6), mk_next);
      masks_ready(W1{}, mk_cur);
      // d z7 = relu'(h7) * (W_out^T d raw)                  -> hB
      stage(I1{}, I0{}, I8{}, I1{}, I1{}, &dout[0], &dout[0], &hB[0][0], true,
            [&](int nt) { ws_store(WL.k_dout(), I1{}, &dout[0], I1{}, nt, I8{}); }, WL.k_dz(7), IH{});
      masks_ready(W1{}, mk_next);   // only one block (2 DMA pieces) was issued behind this load so far
    }
    // d z_{i-1} = relu'(h_{i-1}) * (W_i^T d z_i), i = 7..1 ; d z7 is in hB
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int i = 7 - 2 * it;  // consumes d z_i from hB
      roll_masks(W6{});
      mask_load(AL.k_mask(i - 2), mk_next);
      stage(IH{}, I0{}, I8{}, IH{}, IH{}, &hB[0][0], &hB[0][0], &hA[0][0], true,
            [&](int nt) { ws_store_par(PODD{}, WL.k_dz(i), IH{}, &hB[0][0], IH{}, nt, I8{}); }, SAVE_EVEN ? WL.k_dz(i - 1) : -1, IH{});
      roll_masks(W6{});
      if constexpr (SAVE_EVEN) mask_load(AL.k_mask(i - 3 >= 0 ? i - 3 : 0), mk_next);
      else if (i - 3 > 0) mask_load(AL.k_mask(i - 3), mk_next);   // (mask 0 belongs to the d z0 stage: no load is left in flight)
      stage(IH{}, I0{}, I8{}, IH{}, IH{}, &hA[0][0], &hA[0][0], &hB[0][0], true,
            [&](int nt) { if constexpr (SAVE_EVEN) ws_store_par(PODD{}, WL.k_dz(i - 1), IH{}, &hA[0][0], IH{}, nt, I8{}); }, WL.k_dz(i - 2), IH{});
    }
    if constexpr (SAVE_EVEN) {
      // i = 1: d z0 from d z1 (hB) -> hA
      roll_masks(W6{});
      stage(IH{}, I0{}, I8{}, IH{}, IH{}, &hB[0][0], &hB[0][0], &hA[0][0], true,
            [&](int nt) { ws_store_par(PODD{}, WL.k_dz(1), IH{}, &hB[0][0], IH{}, nt, I8{}); }, WL.k_dz(0), IH{});
      ws_store_par(PODD{}, WL.k_dz(0), IH{}, &hA[0][0], IH{}, 0, I1{});
    } else {
      ws_store_par(PODD{}, WL.k_dz(1), IH{}, &hB[0][0], IH{}, 0, I1{});   // the chain ends at d z1 (the stream wraps 4 blocks early)
    }
  }
  pipe.drain();
}


}  // namespace snr

int snr::wgrad_launch_bf16(const WgradArgs& w, int total_splits, float* grad, int accumulate, hipStream_t s) {
  constexpr int lds = WgradCfg<kBF16>::RING * 2 * Blob<kBF16>::KS_H * 1024;
  if (int e = ensure_dynamic_lds<&mlp_wgrad_kernel<kBF16>>(lds)) return e;
  {
    ProfScope ps(K_MLP_WGRAD, s);
    mlp_wgrad_kernel<kBF16><<<dim3((unsigned)total_splits), dim3(64 * kWgradWaves), lds, s>>>(w);
  }
  int st = launch_status();
  if (st != SNR_OK) return st;
  const int per_out = 4 * 256 * (256 / 4 + 1);   // 4 lanes per (row, 4-column group | bias) item
  {
    ProfScope ps(K_MLP_WGRAD_REDUCE, s);
    ReduceArgs ra{};
    if (!append_reduce(ra, w, 0, grad, accumulate)) return SNR_ERR_UNSUPPORTED;
    mlp_wgrad_reduce_kernel<kBF16><<<dim3((per_out + 255) / 256, (unsigned)ra.n_outs), dim3(256), 0, s>>>(ra);
  }
  return launch_status();
}

using namespace snr;

static int check_cfg_b(const snr_mlp_config* c) {
  // same rules as mlp_fwd.hip's check_cfg (kept local to this translation unit)
  if (!c) return SNR_ERR_NULL;
  if (c->precision != SNR_PREC_BF16 && c->precision != SNR_PREC_FP32) return SNR_ERR_UNSUPPORTED;
  if (c->i_embed != 0 && c->i_embed != -1) return SNR_ERR_UNSUPPORTED;
  if (c->multires < 0 || c->multires > kMaxMultires) return SNR_ERR_UNSUPPORTED;
  if (c->multires_views < (c->use_viewdirs ? 0 : -1) || c->multires_views > kMaxMultiresViews) return SNR_ERR_UNSUPPORTED;
  if (c->use_viewdirs) { if (c->out_ch != 4) return SNR_ERR_UNSUPPORTED; }
  else if (c->out_ch != 4 && c->out_ch != 5) return SNR_ERR_UNSUPPORTED;
  return SNR_OK;
}

template <int P> static int64_t ws_bytes(const snr_mlp_config* c, int64_t n) {
  const bool rc = P == kBF16 && recompute_enabled();
  int64_t pf;
  if (rc) {
    pf = wgall_part_bound();   // whatever launch the network takes part in (alone, or merged with another network's backward)
  } else {
    int ts;
    make_jobs<P>(c, n, &pf, &ts, false);
  }
  return WsLayout<P>(n, c->use_viewdirs).dz_bytes() + (pf + kPostFloats) * 4;
}

extern "C" int64_t snr_mlp_bwd_ws_bytes(const snr_mlp_config* c, int64_t n) {
  int st = check_cfg_b(c);
  if (st != SNR_OK) return st;
  if (n <= 0) return SNR_ERR_SHAPE;
  return c->precision == SNR_PREC_BF16 ? ws_bytes<kBF16>(c, n) : ws_bytes<kFP32>(c, n);
}

// one network's backward problem (include/spinnerf_hip.h: snr_mlp_bwd_item)
typedef snr_mlp_bwd_item BwdItem;

template <int P> static DgradArgs dgrad_args(const BwdItem& it, bool save_even) {
  const snr_mlp_config* c = it.cfg;
  const PackTable T = make_pack_table<P>(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  DgradArgs d{};
  d.blob_bwd = (const char*)it.packed + (int64_t)T.fwd_frags * 1024;
  d.bwd_blocks = T.bwd_frags / kBlockFrags - (save_even ? 0 : 8 * Blob<P>::KS_H / kBlockFrags);   // without the d z0 stage
  d.d_raw = it.d_raw; d.n_samples = it.n_samples; d.out_ch = c->out_ch;
  d.act = (const char*)it.act; d.ws = (char*)it.ws;
  d.save_even = save_even;
  return d;
}

// the chains of 1..2 networks in one launch; the workgroups are shared out in proportion to the networks' tiles
template <int P, bool VD, bool SAVE_EVEN>
static int launch_dgrad(DgradMulti& m, hipStream_t s) {
  constexpr int per_wg = ChainCfg<P, true>::WAVES * ChainCfg<P, true>::NJ;
  int64_t n_wg[kMaxReduceNets], total = 0;
  for (int i = 0; i < m.n; ++i) { n_wg[i] = padded_tiles<P>(m.net[i].n_samples) / per_wg; total += n_wg[i]; }
  const int64_t budget = tunables().chain_grid > 0 ? tunables().chain_grid : 1024;   // one workgroup per CU is resident; the rest grid-stride
  int block = 0;
  for (int i = 0; i < m.n; ++i) {
    int64_t g = total <= budget ? n_wg[i] : n_wg[i] * budget / total;
    if (g < 1) g = 1;
    m.net[i].block0 = block; m.net[i].blocks = (int)g; block += (int)g;
  }
  const int lds = kRingBytes;
  if (int e = ensure_dynamic_lds<&mlp_dgrad_kernel<P, VD, SAVE_EVEN>>(lds)) return e;
  {
    ProfScope ps(K_MLP_DGRAD, s);
    mlp_dgrad_kernel<P, VD, SAVE_EVEN><<<dim3((unsigned)block), dim3(64 * ChainCfg<P, true>::WAVES), lds, s>>>(m);
  }
  return launch_status();
}

// The reduce kernel stores every parameter a weight-gradient job produces; the only parameters no job
// produces are views_linears.0 of a network without view directions (unused, but part of the state dict).
static int clear_unused_grads(const BwdItem& it, hipStream_t s) {
  const snr_mlp_config* c = it.cfg;
  if (it.accumulate || c->use_viewdirs) return SNR_OK;
  const ParamLayout L = make_param_layout(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  hipError_t e = hipMemsetAsync(it.grad_params + L.w_views, 0, (size_t)((L.b_views + kW / 2) - L.w_views) * sizeof(float), s);
  return e == hipSuccess ? SNR_OK : (int)e;
}

static PostNet post_net(const BwdItem& it, const float* post) {
  const snr_mlp_config* c = it.cfg;
  const ParamLayout L = make_param_layout(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  PostNet pn{};
  pn.params = it.params; pn.post = post; pn.grad = it.grad_params;
  pn.w_views = (int)L.w_views; pn.ld_views = kW + L.in_dir; pn.w_feat = (int)L.w_feat; pn.b_feat = (int)L.b_feat;
  pn.accumulate = it.accumulate;
  return pn;
}

constexpr int kReducePerOut = 4 * 256 * (256 / 4 + 1);   // 4 lanes per item; items = rows x (4-column groups + the bias)

// ---- one network, every layer saved: the plain split-K pass (fp32 mode; bf16 with SNR_RECOMPUTE=0) ----
template <int P>
static int backward_plain(const BwdItem& it, hipStream_t s) {
  const snr_mlp_config* c = it.cfg;
  int st = clear_unused_grads(it, s);
  if (st != SNR_OK) return st;
  DgradMulti m{};
  m.n = 1;
  m.net[0] = dgrad_args<P>(it, true);
  st = c->use_viewdirs ? launch_dgrad<P, true, true>(m, s) : launch_dgrad<P, false, true>(m, s);
  if (st != SNR_OK) return st;
  int64_t pf; int total_splits;
  WgradArgs w = make_jobs<P>(c, it.n_samples, &pf, &total_splits, false);
  w.act = (const char*)it.act;
  w.ws = (const char*)it.ws;
  w.part = (float*)((char*)it.ws + WsLayout<P>(it.n_samples, c->use_viewdirs).dz_bytes());
  w.post = w.part + pf;
  constexpr int lds = WgradCfg<P>::RING * 2 * Blob<P>::KS_H * 1024;   // the largest ring of any job (mlp_wgrad.h)
  if (int e = ensure_dynamic_lds<&mlp_wgrad_kernel<P>>(lds)) return e;
  {
    ProfScope ps(K_MLP_WGRAD, s);
    mlp_wgrad_kernel<P><<<dim3((unsigned)total_splits), dim3(64 * kWgradWaves), lds, s>>>(w);
  }
  st = launch_status();
  if (st != SNR_OK) return st;
  {
    ProfScope ps(K_MLP_WGRAD_REDUCE, s);
    ReduceArgs ra{};
    if (!append_reduce(ra, w, 0, it.grad_params, it.accumulate)) return SNR_ERR_UNSUPPORTED;
    mlp_wgrad_reduce_kernel<P><<<dim3((kReducePerOut + 255) / 256, (unsigned)ra.n_outs), dim3(256), 0, s>>>(ra);
  }
  if (c->use_viewdirs) {
    ProfScope ps(K_WGRAD_POST, s);
    PostArgs pa{};
    pa.net[0] = post_net(it, w.post);
    wgrad_post_kernel<P><<<dim3(kPostTiles), dim3(256), 0, s>>>(pa);
  }
  return launch_status();
}

// ---- bf16 with selective recompute: ONE launch sequence for the backward passes of 1..2 networks ----
//   dgrad chains of every network | layer pairs + plain jobs of every network | reduce of every partial plane | the G products
static int backward_merged(const BwdItem* items, int n, hipStream_t s) {
  constexpr int P = kBF16;
  const bool vd = items[0].cfg->use_viewdirs;
  WgNet nets[kMaxReduceNets];
  for (int i = 0; i < n; ++i) {
    int st = clear_unused_grads(items[i], s);
    if (st != SNR_OK) return st;
    nets[i] = WgNet{};
    nets[i].c = items[i].cfg; nets[i].n_samples = items[i].n_samples;
  }
  WgAllPlan pl;
  int st = make_wgall_plan(nets, n, pl);
  if (st != SNR_OK) return st;
  DgradMulti m{};
  m.n = n;
  ReduceArgs ra{};
  PostArgs post{};
  for (int i = 0; i < n; ++i) {
    const BwdItem& it = items[i];
    const snr_mlp_config* c = it.cfg;
    const PackTable T = make_pack_table<P>(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
    float* part = (float*)((char*)it.ws + WsLayout<P>(it.n_samples, c->use_viewdirs).dz_bytes());
    float* g_block = part + nets[i].part_floats;     // G (+ s9) behind this launch's partial sums
    m.net[i] = dgrad_args<P>(it, false);
    m.net[i].zero_words = (unsigned*)(part + nets[i].sync_off);
    m.net[i].n_zero = kSyncWords;
    fill_wgall_pointers(pl, nets, i, (const char*)it.act, (const char*)it.ws, (const char*)it.packed,
                        (const float*)((const char*)it.packed + (int64_t)(T.fwd_frags + T.bwd_frags) * 1024), part);
    nets[i].plain.part = part; nets[i].plain.post = g_block;
    nets[i].red.part = part; nets[i].red.post = g_block;
    if (!append_reduce(ra, nets[i].plain, i, it.grad_params, it.accumulate)) return SNR_ERR_UNSUPPORTED;
    if (!append_reduce(ra, nets[i].red, i, it.grad_params, it.accumulate)) return SNR_ERR_UNSUPPORTED;
    post.net[i] = post_net(it, g_block);
  }
  st = vd ? launch_dgrad<P, true, false>(m, s) : launch_dgrad<P, false, false>(m, s);
  if (st != SNR_OK) return st;
  if (int e = ensure_dynamic_lds<&mlp_wgrad_pair_kernel>(kLdsBytes)) return e;
  {
    ProfScope ps(K_MLP_WGRAD_PAIR, s);
    mlp_wgrad_pair_kernel<<<dim3((unsigned)pl.grid), dim3(64 * kPairWaves), kLdsBytes, s>>>(pl.pa);
  }
  st = launch_status();
  if (st != SNR_OK) return st;
  {
    ProfScope ps(K_MLP_WGRAD_REDUCE, s);
    mlp_wgrad_reduce_kernel<P><<<dim3((kReducePerOut + 255) / 256, (unsigned)ra.n_outs), dim3(256), 0, s>>>(ra);
  }
  if (vd) {
    ProfScope ps(K_WGRAD_POST, s);
    wgrad_post_kernel<P><<<dim3((unsigned)(kPostTiles * n)), dim3(256), 0, s>>>(post);
  }
  return launch_status();
}

static int check_item(const BwdItem& it) {
  int st = check_cfg_b(it.cfg);
  if (st != SNR_OK) return st;
  SNR_CHECK_ARG(it.packed && it.d_raw && it.act && it.ws && it.grad_params && (it.params || !it.cfg->use_viewdirs), SNR_ERR_NULL);
  SNR_CHECK_ARG(it.n_samples > 0, SNR_ERR_SHAPE);
  return SNR_OK;
}

extern "C" int snr_mlp_backward_multi(const snr_mlp_bwd_item* items, int n_items, snr_stream_t stream) {
  SNR_CHECK_ARG(items, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_items >= 1 && n_items <= kMaxReduceNets, SNR_ERR_SHAPE);
  for (int i = 0; i < n_items; ++i) {
    int st = check_item(items[i]);
    if (st != SNR_OK) return st;
  }
  hipStream_t s = (hipStream_t)stream;
  // one launch sequence for all of them: bf16 networks with selective recompute that share the kernels' compile-time shape
  // (view directions or not) and are distinct problems (a buffer cannot be two networks' workspace at once)
  bool merged = recompute_enabled();
  for (int i = 0; i < n_items; ++i) {
    merged = merged && items[i].cfg->precision == SNR_PREC_BF16 && items[i].cfg->use_viewdirs == items[0].cfg->use_viewdirs;
    if (i > 0) merged = merged && tunables().merge_nets && items[i].ws != items[0].ws && items[i].act != items[0].act &&
                        items[i].grad_params != items[0].grad_params;
  }
  if (merged) return backward_merged(items, n_items, s);
  for (int i = 0; i < n_items; ++i) {
    const BwdItem& it = items[i];
    int st;
    if (it.cfg->precision == SNR_PREC_BF16) st = recompute_enabled() ? backward_merged(&it, 1, s) : backward_plain<kBF16>(it, s);
    else st = backward_plain<kFP32>(it, s);
    if (st != SNR_OK) return st;
  }
  return SNR_OK;
}

extern "C" int snr_mlp_backward(const snr_mlp_config* c, const void* packed, const float* params, const float* d_raw,
                                int64_t n, const void* act, void* ws, float* grad, int accumulate, snr_stream_t stream) {
  const snr_mlp_bwd_item it{c, packed, params, d_raw, n, act, ws, grad, accumulate};
  return snr_mlp_backward_multi(&it, 1, stream);
}

extern "C" int snr_tunables_reload(void) {
  tunables_storage() = read_tunables();
  return SNR_OK;
}

#ifdef SNR_TIMING
extern "C" int snr_debug_read(unsigned long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(snr::g_snr_dbg), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(snr::g_snr_dbg), z, sizeof(z));
  return (int)e;
}
#endif
