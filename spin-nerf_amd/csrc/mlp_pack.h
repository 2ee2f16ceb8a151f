// Host-side description of the packed weight blob and the saved-activation workspace of the fused
// NeRF MLP, plus the table that drives the (single-launch) pack kernel.
#pragma once
#include "mlp_layout.h"

namespace snr {

enum SrcKind { SRC_NONE = 0, SRC_ENC_PTS = 1, SRC_ENC_DIR = 2, SRC_H = 3, SRC_OUT = 4,
               SRC_NAT = 5,       // slot x = column x (hash-grid features)
               SRC_HG_INC = 6 };  // hash-grid colour-network input: 16 SH slots, then mlp_layout.h: hg_inc_col

// One source segment of a stage's contraction: `ks` frags whose k-slots map to weight columns
// (forward) or weight rows (transposed / dgrad).
struct PackSrc {
  int kind, ks;
  int w_off;      // offset of the weight matrix in the flat parameter buffer
  int ld;         // its row length
  int col_off;    // forward: first column of this source; dgrad: first column of the OUTPUT rows
  int n_valid;    // forward: #valid slot columns are implied by kind; dgrad: #valid weight rows
  int slot_off;   // dgrad: weight row = slot value - slot_off
};

struct PackEntry {
  int frag_begin;   // first frag (KiB) of this entry in the blob
  int n_tiles;      // 32-row tiles
  int transposed;   // 0 forward (rows = out neurons), 1 dgrad (rows = in neurons)
  int rows_valid;   // rows >= this are zero
  PackSrc src[2];
};

struct BiasEntry { int dst, count, src, n_valid; };

constexpr int kMaxPackEntries = 26;
constexpr int kMaxBiasEntries = 12;

struct PackTable {
  int n_entries, n_bias;
  int fwd_frags, bwd_frags, bias_floats;
  int multires, multires_views;
  PackEntry e[kMaxPackEntries];
  BiasEntry b[kMaxBiasEntries];
};

// bias block offsets (floats): stage s<8 -> 256*s; vd: feat 2048 (+256..287 alpha tile), views 2336, rgb 2464
// novd: out 2048
SNR_HD int bias_off_stage(int s) { return 256 * s; }
constexpr int kBiasFeat = 2048, kBiasAlphaTile = 2048 + 256, kBiasViews = 2048 + 288, kBiasRgb = 2048 + 288 + 128;
constexpr int kBiasOut = 2048;

template <int P>
inline PackTable make_pack_table(int multires, int multires_views, int use_viewdirs, int out_ch, int identity) {
  using B = Blob<P>;
  PackTable T{};
  const ParamLayout L = make_param_layout(multires, multires_views, use_viewdirs, out_ch, identity);
  T.multires = identity ? 0 : multires;
  T.multires_views = identity ? 0 : multires_views;
  int frag = 0, n = 0;
  constexpr int BF = kBlockFrags;  // every entry (= stage) starts on a DMA block boundary
  auto add = [&](int tiles, int transposed, int rows_valid, PackSrc a, PackSrc b) {
    frag = (frag + BF - 1) / BF * BF;
    PackEntry& E = T.e[n++];
    E.frag_begin = frag; E.n_tiles = tiles; E.transposed = transposed; E.rows_valid = rows_valid;
    E.src[0] = a; E.src[1] = b;
    frag += tiles * (a.ks + b.ks);
  };
  const PackSrc none{SRC_NONE, 0, 0, 0, 0, 0, 0};
  const int ip = L.in_pts;
  // ---------------- forward ----------------
  add(8, 0, kW, PackSrc{SRC_ENC_PTS, B::KS_PE, (int)L.w_pts[0], ip, 0, 0, 0}, none);
  for (int i = 1; i <= 4; ++i) add(8, 0, kW, PackSrc{SRC_H, B::KS_H, (int)L.w_pts[i], kW, 0, 0, 0}, none);
  add(8, 0, kW, PackSrc{SRC_ENC_PTS, B::KS_PE, (int)L.w_pts[5], kW + ip, 0, 0, 0},
      PackSrc{SRC_H, B::KS_H, (int)L.w_pts[5], kW + ip, ip, 0, 0});
  for (int i = 6; i <= 7; ++i) add(8, 0, kW, PackSrc{SRC_H, B::KS_H, (int)L.w_pts[i], kW, 0, 0, 0}, none);
  if (use_viewdirs) {
    add(8, 0, kW, PackSrc{SRC_H, B::KS_H, (int)L.w_feat, kW, 0, 0, 0}, none);
    add(1, 0, 1, PackSrc{SRC_H, B::KS_H, (int)L.w_alpha, kW, 0, 0, 0}, none);
    add(4, 0, kW / 2, PackSrc{SRC_H, B::KS_H, (int)L.w_views, kW + L.in_dir, 0, 0, 0},
        PackSrc{SRC_ENC_DIR, B::KS_DIR, (int)L.w_views, kW + L.in_dir, kW, 0, 0});
    add(1, 0, 3, PackSrc{SRC_H, B::KS_H9, (int)L.w_rgb, kW / 2, 0, 0, 0}, none);
  } else {
    add(1, 0, out_ch, PackSrc{SRC_H, B::KS_H, (int)L.w_out, kW, 0, 0, 0}, none);
  }
  frag = (frag + BF - 1) / BF * BF;
  T.fwd_frags = frag;
  // ---------------- dgrad (rows = input neurons, slots = output neurons) ----------------
  if (use_viewdirs) {
    // d h9 = W_rgb^T d rgb            (4 tiles of h9 rows; slots = out channels 0..2)
    add(4, 1, kW / 2, PackSrc{SRC_OUT, 1, (int)L.w_rgb, kW / 2, 0, 3, 0}, none);
    // d feat = W_views[:, :256]^T d z9
    add(8, 1, kW, PackSrc{SRC_H, B::KS_H9, (int)L.w_views, kW + L.in_dir, 0, kW / 2, 0}, none);
    // d h7 = W_feat^T d feat + W_alpha^T d alpha (channel 3 of the OUT frag)
    add(8, 1, kW, PackSrc{SRC_H, B::KS_H, (int)L.w_feat, kW, 0, kW, 0},
        PackSrc{SRC_OUT, 1, (int)L.w_alpha, kW, 0, 1, 3});
  } else {
    add(8, 1, kW, PackSrc{SRC_OUT, 1, (int)L.w_out, kW, 0, out_ch, 0}, none);
  }
  for (int i = 7; i >= 1; --i) {
    // d h_{i-1} = W_i[:, skip_cols:]^T d z_i
    const int ld = (i == kSkip + 1) ? kW + ip : kW;
    const int co = (i == kSkip + 1) ? ip : 0;
    add(8, 1, kW, PackSrc{SRC_H, B::KS_H, (int)L.w_pts[i], ld, co, kW, 0}, none);
  }
  frag = (frag + BF - 1) / BF * BF;
  T.bwd_frags = frag - T.fwd_frags;
  T.n_entries = n;
  // ---------------- biases ----------------
  int nb = 0;
  for (int i = 0; i < 8; ++i) T.b[nb++] = BiasEntry{bias_off_stage(i), 256, (int)L.b_pts[i], 256};
  if (use_viewdirs) {
    T.b[nb++] = BiasEntry{kBiasFeat, 256, (int)L.b_feat, 256};
    T.b[nb++] = BiasEntry{kBiasAlphaTile, 32, (int)L.b_alpha, 1};
    T.b[nb++] = BiasEntry{kBiasViews, 128, (int)L.b_views, 128};
    T.b[nb++] = BiasEntry{kBiasRgb, 32, (int)L.b_rgb, 3};
  } else {
    T.b[nb++] = BiasEntry{kBiasOut, 32, (int)L.b_out, out_ch};
  }
  T.n_bias = nb;
  T.bias_floats = B::bias_floats(use_viewdirs);
  return T;
}

// ---- saved activations (forward, training mode) and backward scratch ---------------------------
// Sections are [n_tiles][ks KiB]; masks are [n_tiles][1 KiB] (one u32x4 per lane: bit 16*(t&1)+r of
// word t>>1 = "C register r of tile t was > 0").
template <int P> struct ActLayout {
  using B = Blob<P>;
  int64_t n_tiles;
  int use_viewdirs;
  SNR_HD ActLayout(int64_t n_samples, int vd) : n_tiles(padded_tiles<P>(n_samples)), use_viewdirs(vd) {}
  // KiB per tile of each section
  // (the output of feature_linear is NOT saved: it is an affine function of h7, and the two weight gradients that would
  //  need it or its gradient follow from G = d z9^T h7 — mlp_wgrad.h, wgrad_post_kernel)
  SNR_HD int64_t kib_per_tile() const {
    return B::KS_PE + 8 * B::KS_H + 8 /*masks 0..7*/ + (use_viewdirs ? B::KS_DIR + B::KS_H9 + 1 : 0);
  }
  SNR_HD int64_t bytes() const { return n_tiles * kib_per_tile() * 1024; }
  // section starts, in KiB per tile (k_*) and in bytes (off_* = n_tiles * 1024 * k_*)
  SNR_HD int k_pe() const { return 0; }
  SNR_HD int k_h(int i) const { return B::KS_PE + i * B::KS_H; }
  SNR_HD int k_mask(int i) const { return B::KS_PE + 8 * B::KS_H + i; }  // i in 0..7
  SNR_HD int k_dir() const { return B::KS_PE + 8 * B::KS_H + 8; }
  SNR_HD int k_h9() const { return k_dir() + B::KS_DIR; }
  SNR_HD int k_mask9() const { return k_h9() + B::KS_H9; }
  SNR_HD int64_t off_pe() const { return 0; }
  SNR_HD int64_t off_h(int i) const { return n_tiles * 1024 * k_h(i); }
  SNR_HD int64_t off_mask(int i) const { return n_tiles * 1024 * k_mask(i); }
  SNR_HD int64_t off_dir() const { return n_tiles * 1024 * k_dir(); }
  SNR_HD int64_t off_h9() const { return n_tiles * 1024 * k_h9(); }
  SNR_HD int64_t off_mask9() const { return n_tiles * 1024 * k_mask9(); }
};

}  // namespace snr
