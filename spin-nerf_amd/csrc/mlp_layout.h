// Layout contract of the fused NeRF MLP kernels (gfx950).
//
// The network is the reference's NeRF (DS_NeRF/run_nerf_helpers.py:74-127), D=8, W=256, skip
// after layer 4.  The kernels evaluate every layer TRANSPOSED, out^T[neuron][sample] =
// W[neuron][k] * in^T[k][sample], with v_mfma_f32_32x32x16_bf16 (or _32x32x2_f32):
//   A operand = a 32-neuron x K weight fragment (streamed through LDS, shared by the workgroup),
//   B operand = the wave's own 32-sample activation fragment (registers),
//   C/D       = 32 neurons x 32 samples; lane l holds sample (l&31), rows
//               (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// Because a lane's C registers hold one sample and 16 neurons, they can be fed back as the next
// layer's B operand without any cross-lane movement, provided the next layer's weights are packed
// in the k-order the registers imply.  This header is that order.
//
// Vocabulary
//   tile   : 32 consecutive samples (one wave's unit of work)
//   frag   : 16 bytes per lane = one LDS read of A / one register group of B.
//            bf16: 8 elements = one K=16 MFMA step (lane half g supplies k-slots 8g..8g+7)
//            fp32: 4 elements = four K=2 MFMA steps (lane half g supplies k-slot g of each)
//   FPT    : frags per 32-neuron C tile (bf16 2, fp32 4); EPF elements per frag (8 / 4)
//   ACT    : activation tensors saved for backward, [tile][frag][sample j][2*EPF elements]
//            (lane (j,g) owns bytes [(frag*32 + j)*32 + g*16, +16) of its tile's layer block)
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define SNR_HD __host__ __device__ __forceinline__
#else
#define SNR_HD inline
#endif

namespace snr {

constexpr int kW = 256;            // netwidth (reference default, run_nerf.py config)
constexpr int kD = 8;              // netdepth
constexpr int kSkip = 4;           // skips=[4]  (run_nerf.py:391)
constexpr int kPeSlots = 64;       // padded positional-encoding width (3+6*10 = 63 -> 64)
constexpr int kDirSlots = 32;      // padded view-direction encoding width (3+6*4 = 27 -> 32)
constexpr int kMaxMultires = 10;
constexpr int kMaxMultiresViews = 4;
constexpr int kTileSamples = 32;

enum Precision { kBF16 = 0, kFP32 = 1 };

// fragments (1 KiB each) per DMA block of the packed weight stream; every stage starts on a block boundary
#ifndef SNR_BLOCK_FRAGS
#define SNR_BLOCK_FRAGS 32   // 32 KiB blocks: half the block entries (counted wait + barrier) of 16; same-call A/B:
                             // training forward -2.5 %, inference frame -1.8 %, dgrad unchanged (A/B builds: 16)
#endif
constexpr int kBlockFrags = SNR_BLOCK_FRAGS;

// Round 6: in bf16 mode the positional / directional encodings and the weight columns that multiply them are fp16 (11 mantissa
// bits for values in [-1, 1]) inside the same fragments, consumed by v_mfma_f32_32x32x16_f16 at the bf16 rate (snr_common.h:
// Mma<kBF16>::mma_f16).  SNR_ENC_F16=0 builds the round-5 all-bf16 arithmetic (A/B, tools/build_variant.py).
#ifndef SNR_ENC_F16
#define SNR_ENC_F16 1
#endif
template <int P> struct EncF16 { static constexpr bool value = SNR_ENC_F16 && P == kBF16; };

template <int P> struct Prec;
template <> struct Prec<kBF16> { static constexpr int EPF = 8, FPT = 2, ESZ = 2; };
template <> struct Prec<kFP32> { static constexpr int EPF = 4, FPT = 4, ESZ = 4; };

// Shape of a chained-kernel workgroup (forward, dgrad): WAVES waves, each owning NJ tiles of 32 samples.
//  * bf16 (inference forward, training forward, dgrad): 8 waves x 1 tile = two waves per SIMD in <= 256
//    architectural VGPRs each, no AGPRs (the epilogue reads MFMA results without v_accvgpr copies); the
//    encodings are re-derived where consumed instead of kept.  History of the A/B measurements on MI355X at
//    196 608 samples: before the LDS reads became inline asm the compiler drained the DMA queue at every block
//    (which looked like "each LDS-DMA instruction blocks its wave ~170 cycles") and 4 waves x 2 tiles was the
//    best training shape (0.373 / 0.286 ms forward / dgrad vs 0.384 / 0.331 at 4 x 1); with the asm reads
//    4 x 2 runs out of registers (139 spilled dwords) and 8 x 1 with packed epilogues is 0.33 / 0.28 ms;
//    inference 8 x 1: 0.21 ms = 1100 TFLOP/s (4 x 2: 113 spilled dwords, 0.281 ms).  Round 2, after the encoding's
//    loop invariants stopped occupying registers: 4 x 2 (512 registers, 27 spilled dwords) renders a frame in the
//    same 59.7 ms as 8 x 1 — halving the LDS fragment reads buys nothing, as tests/probes/mfma_feed.hip predicts.
//  * fp32 (parity path): 4 waves x 1 tile — its activations alone are 256 registers.
template <int P, bool TRAIN> struct ChainCfg { static constexpr int WAVES = 4, NJ = 1; };
#if defined(SNR_INFER_WAVES) && defined(SNR_INFER_NJ)   // A/B builds only
template <> struct ChainCfg<kBF16, false> { static constexpr int WAVES = SNR_INFER_WAVES, NJ = SNR_INFER_NJ; };
#else
template <> struct ChainCfg<kBF16, false> { static constexpr int WAVES = 8, NJ = 1; };
#endif
#if defined(SNR_TRAIN_WAVES) && defined(SNR_TRAIN_NJ)   // A/B builds only
template <> struct ChainCfg<kBF16, true> { static constexpr int WAVES = SNR_TRAIN_WAVES, NJ = SNR_TRAIN_NJ; };
#else
template <> struct ChainCfg<kBF16, true> { static constexpr int WAVES = 8, NJ = 1; };
#endif
// 32-sample tiles of the saved-activation sections: padded to whole training workgroups
template <int P> SNR_HD int64_t padded_tiles(int64_t n_samples) {
  constexpr int per_wg = ChainCfg<P, true>::WAVES * ChainCfg<P, true>::NJ;
  constexpr int wg = kTileSamples * per_wg;
  return (n_samples + wg - 1) / wg * per_wg;
}

// ---- k-slot -> neuron maps -------------------------------------------------------------

// Source = the previous layer's C tiles.  frag q, lane half g, element e -> true neuron index.
template <int P> SNR_HD int h_slot_neuron(int q, int g, int e) {
  const int t = q / Prec<P>::FPT;
  const int r = (q % Prec<P>::FPT) * Prec<P>::EPF + e;
  return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
}

// Source = an encoding computed in-kernel.  Features are laid out as (sin, cos) PAIRS so that one
// sincos serves both: lane half g of frag q owns pairs (2q+g)*(EPF/2) + (e>>1), element parity e&1
// selects sin/cos.  Pair p < 3L: frequency p/3, axis p%3.  Pair 3L = (x, y), pair 3L+1 = (z, pad).
// Returns the column in the reference's embedding order (helpers:22-52: x, then per frequency
// sin xyz, cos xyz) or -1 for padding.
template <int P> SNR_HD int enc_slot_feature(int q, int g, int e, int L) {
  const int p = (2 * q + g) * (Prec<P>::EPF / 2) + (e >> 1);
  const int which = e & 1;
  if (p < 3 * L) return 3 + 6 * (p / 3) + 3 * which + (p % 3);
  if (p == 3 * L) return which;             // x, y
  if (p == 3 * L + 1) return which ? -1 : 2;  // z, pad
  return -1;
}

// Hash-grid colour network (hashgrid.hip): its second input k-step is the sigma network's 16-row output tile as it sits
// in registers — slot (g, e) holds row (e&3) + 8(e>>2) + 4g; row 0 (sigma) is replaced by the constant-1 padding input
// (column 31 of the 32-wide first layer), row j by geo feature j-1 (column 15 + j).
SNR_HD int hg_inc_col(int g, int e) {
  const int row = (e & 3) + 8 * (e >> 2) + 4 * g;
  return row == 0 ? 31 : 15 + row;
}

// ---- flat parameter buffer (state-dict order of the reference module) -------------------------
// pts_linears.{0..7}.{weight,bias}, views_linears.0.{weight,bias}, then with viewdirs
// feature_linear, alpha_linear, rgb_linear; without: output_linear.  (helpers:86-102)
struct ParamLayout {
  int in_pts, in_dir, out_ch, use_viewdirs;
  int64_t w_pts[kD], b_pts[kD];
  int64_t w_views, b_views, w_feat, b_feat, w_alpha, b_alpha, w_rgb, b_rgb, w_out, b_out;
  int64_t total;
};

SNR_HD ParamLayout make_param_layout(int multires, int multires_views, int use_viewdirs, int out_ch,
                                     int i_embed_identity) {
  ParamLayout L{};
  L.in_pts = i_embed_identity ? 3 : 3 + 6 * multires;
  // multires_views < 0 = "views_linears.0 has no direction columns" (create_nerf without
  // use_viewdirs passes input_ch_views = 0, run_nerf.py:385-395); the layer is unused then but
  // still part of the state dict.
  L.in_dir = multires_views < 0 ? 0 : (i_embed_identity ? 3 : 3 + 6 * multires_views);
  L.out_ch = out_ch;
  L.use_viewdirs = use_viewdirs;
  int64_t o = 0;
  for (int i = 0; i < kD; ++i) {
    const int fin = (i == 0) ? L.in_pts : (i == kSkip + 1 ? kW + L.in_pts : kW);
    L.w_pts[i] = o; o += (int64_t)kW * fin;
    L.b_pts[i] = o; o += kW;
  }
  L.w_views = o; o += (int64_t)(kW / 2) * (L.in_dir + kW);
  L.b_views = o; o += kW / 2;
  L.w_feat = L.b_feat = L.w_alpha = L.b_alpha = L.w_rgb = L.b_rgb = L.w_out = L.b_out = -1;
  if (use_viewdirs) {
    L.w_feat = o; o += (int64_t)kW * kW;
    L.b_feat = o; o += kW;
    L.w_alpha = o; o += kW;
    L.b_alpha = o; o += 1;
    L.w_rgb = o; o += 3 * (kW / 2);
    L.b_rgb = o; o += 3;
  } else {
    L.w_out = o; o += (int64_t)out_ch * kW;
    L.b_out = o; o += out_ch;
  }
  L.total = o;
  return L;
}

// ---- packed blob --------------------------------------------------------------------------
// Forward chunks are stored in consumption order; a chunk = one 32-neuron output tile of one layer
// = KS frags x 1 KiB (frag f, lane l -> 16 bytes at (f*64 + l)*16).
//
// Fused layer list ("stages"):
//   0      pts0            src PE                  8 tiles  relu
//   1..4   pts1..4         src H                   8 tiles  relu
//   5      pts5            src PE | H              8 tiles  relu     (skip: input first, helpers:111)
//   6,7    pts6,7          src H                   8 tiles  relu
//   viewdirs:
//   8      feature+alpha   src H                   9 tiles  (tile 8 row 0 = alpha_linear)  no relu
//   9      views0          src FEAT | DIR          4 tiles  relu
//   10     rgb             src H9 (128)            1 tile   (rows 0..2)
//   no viewdirs:
//   8      output_linear   src H                   1 tile   (rows 0..out_ch-1)
template <int P> struct Blob {
  static constexpr int FPT = Prec<P>::FPT;
  static constexpr int KS_H = 8 * FPT;                  // frags of a 256-wide source
  static constexpr int KS_PE = kPeSlots / 32 * FPT;     // 64-wide
  static constexpr int KS_DIR = kDirSlots / 32 * FPT;   // 32-wide
  static constexpr int KS_H9 = 4 * FPT;                 // 128-wide
  static constexpr int KS_OUT = 1;                      // backward only: the 16-slot d_raw frag (bf16) ...
  static constexpr int MAX_CHUNK_FRAGS = KS_PE + KS_H;  // stage 5

  // forward section sizes in frags (KiB)
  static constexpr int F_S0 = 8 * KS_PE;
  static constexpr int F_SH = 8 * KS_H;
  static constexpr int F_S5 = 8 * (KS_PE + KS_H);
  static constexpr int F_S8V = 9 * KS_H;
  static constexpr int F_S9 = 4 * (KS_H + KS_DIR);
  static constexpr int F_S10 = KS_H9;
  static constexpr int F_S8N = KS_H;
  SNR_HD static int fwd_frags(int vd) { return F_S0 + 6 * F_SH + F_S5 + (vd ? F_S8V + F_S9 + F_S10 : F_S8N); }

  // bias block (fp32, true neuron order): 8 x 256, then vd: 288 (feat + alpha tile), 128, 32; novd: 32
  SNR_HD static int bias_floats(int vd) { return 8 * 256 + (vd ? 288 + 128 + 32 : 32); }
};

}  // namespace snr
