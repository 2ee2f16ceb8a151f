// Backward of the fused NeRF MLP for gfx950: d(loss)/d(params) from d(loss)/d(raw).
//
// Replaces torch autograd through NeRF.forward (DS_NeRF/run_nerf_helpers.py:104-127) as driven by
// loss.backward() in train() (DS_NeRF/run_nerf.py:1611).  No gradient is propagated into the
// encodings / sample positions (nothing upstream needs one).
//
// Three kernels:
//   1. dgrad chain  — same structure as the forward (4 waves x 32-sample tiles, activations in
//      registers, W^T chunks DMA'd through LDS): d z_{i-1}^T = relu'(h_{i-1}) * (W_i^T d z_i^T).
//      ReLU masks come from the 1-bit-per-activation masks the forward saved; every d z_i is
//      written (bf16 / fp32 fragments) for the weight-gradient pass.
//   2. wgrad        — dW_i[n][k] = sum_s d z_i[s][n] h_{i-1}[s][k]: a contraction over SAMPLES, so
//      both operands need 8 consecutive samples per lane while the saved fragments hold 8
//      consecutive neurons per lane.  Tiles are DMA'd to LDS unchanged and transposed on the way
//      out with ds_read_b64_tr_b16 (bf16) / per-lane ds_read_b32 (fp32).  Each workgroup owns the
//      whole (<= 256 x 256) output of one layer for a slice of the samples (split-K); the kernel is
//      HBM-bound (1 KB of saved activations per 131 KFLOP).
//   3. reduce       — sums the split-K partials and scatters them from fragment order to the
//      reference's [out, in] parameter layout (+ bias gradients).
#include <type_traits>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {

// ------------------------------------------------------------------------------------------
// backward scratch: d z sections ([n_tiles][ks KiB], like ActLayout) followed by wgrad partials
// ------------------------------------------------------------------------------------------
template <int P> struct WsLayout {
  using B = Blob<P>;
  int64_t n_tiles;
  int vd;
  SNR_HD WsLayout(int64_t n_samples, int vd_) : n_tiles(((n_samples + 127) / 128) * 4), vd(vd_) {}
  SNR_HD int64_t off_dout() const { return 0; }
  SNR_HD int64_t off_dz(int i) const { return n_tiles * 1024 * (1 + (int64_t)i * B::KS_H); }  // i in 0..7
  SNR_HD int64_t off_dfeat() const { return n_tiles * 1024 * (1 + 8 * (int64_t)B::KS_H); }
  SNR_HD int64_t off_dz9() const { return off_dfeat() + n_tiles * 1024 * B::KS_H; }
  SNR_HD int64_t dz_bytes() const {
    return n_tiles * 1024 * (1 + 8 * (int64_t)B::KS_H + (vd ? B::KS_H + B::KS_H9 : 0));
  }
};

// ------------------------------------------------------------------------------------------
// 1. dgrad chain
// ------------------------------------------------------------------------------------------
struct DgradArgs {
  const char* blob_bwd;   // dgrad chunks (behind the forward section)
  const float* d_raw;     // [n, out_ch]
  int64_t n_samples;
  int out_ch;
  const char* act;        // forward workspace (masks)
  char* ws;               // d z sections
};

template <int P, bool VD>
__global__ __launch_bounds__(256) void mlp_dgrad_kernel(DgradArgs a) {
  using B = Blob<P>;
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int FPT = Prec<P>::FPT, EPF = Prec<P>::EPF;
  constexpr int KS_H = B::KS_H, KS_H9 = B::KS_H9;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;

  constexpr int FIRST_KS = 1;  // both variants start with the OUT frag
  Pipe<P> pipe;
  pipe.slots = smem;
  pipe.gbase = pipe.gcur = a.blob_bwd;
  pipe.slot = 0; pipe.wave = wave; pipe.lane = lane;
  pipe.issue_into(0, FIRST_KS);

  const ActLayout<P> AL(a.n_samples, VD);
  const WsLayout<P> WL(a.n_samples, VD);
  const int64_t n_wg = AL.n_tiles / 4;
  auto nop = []() {};
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  for (int64_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
    const bool more = wg + gridDim.x < n_wg;
    const int64_t tile = wg * 4 + wave;
    const int64_t m = tile * 32 + j;
    const bool valid = m < a.n_samples;

    auto ws_store = [&](int64_t sec_off, int ks, const Frag* src, int n) {
      char* base = a.ws + sec_off + (tile * ks) * 1024 + g * 16;
#pragma unroll
      for (int f = 0; f < 64; ++f)
        if (f < n) *(Frag*)(base + f * 1024 + act_row<P>(j, f) * 32) = src[f];
    };
    auto mask_load = [&](int64_t sec_off) {
      return *(const u32x4*)(a.act + sec_off + tile * 1024 + lane * 16);
    };

    // ---- d raw -> OUT frag (bf16: k-slot 8g+e = channel; fp32: k-slot g of step e = channel 2e+g)
    Frag dout = M::zero();
    if (valid) {
      const float* dr = a.d_raw + m * a.out_ch;
#pragma unroll
      for (int e = 0; e < EPF; ++e) {
        const int ch = (P == kBF16) ? 8 * g + e : 2 * e + g;
        if (ch < a.out_ch) M::set(dout, e, dr[ch]);
      }
    }

    Frag hA[KS_H], hB[KS_H];
    u32x4 mk_cur, mk_next;

    // generic stage: dst = mask * (W^T [sa|sb]); NT output tiles
    auto stage = [&](auto KA_, auto KB_, auto NT_, const Frag* sa, const Frag* sb, Frag* dst, bool use_mask,
                     int next_ks, bool wrap_last, auto&& pre) {
      constexpr int KA = decltype(KA_)::value, KB = decltype(KB_)::value, NT = decltype(NT_)::value;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x16 acc = zero16;
        const bool last = nt == NT - 1;
        if (nt == 0) acc = pipe.template step<KA, KB>(acc, sa, sb, last ? next_ks : KA + KB, last && wrap_last, pre);
        else acc = pipe.template step<KA, KB>(acc, sa, sb, last ? next_ks : KA + KB, last && wrap_last, nop);
        if (use_mask) {
          const unsigned bits = mk_cur[nt >> 1] >> (16 * (nt & 1));
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = ((bits >> r) & 1u) ? acc[r] : 0.f;
        }
        acc_to_frags<P>(acc, dst + nt * FPT);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using IH = std::integral_constant<int, KS_H>;
    using IH9 = std::integral_constant<int, KS_H9>;

    if constexpr (VD) {
      mk_cur = mask_load(AL.off_mask9());
      mk_next = mask_load(AL.off_mask(7));
      // d z9 = relu'(h9) * (W_rgb^T d rgb)                 -> hB[0..KS_H9)
      stage(I1{}, I0{}, I4{}, &dout, &dout, hB, true, KS_H9, false, [&]() { ws_store(WL.off_dout(), 1, &dout, 1); });
      // d feat = W_views[:, :256]^T d z9                    -> hA
      stage(IH9{}, I0{}, I8{}, hB, hB, hA, false, KS_H + 1, false, [&]() { ws_store(WL.off_dz9(), KS_H9, hB, KS_H9); });
      // d z7 = relu'(h7) * (W_feat^T d feat + W_alpha^T d alpha) -> hB
      mk_cur = mk_next;
      mk_next = mask_load(AL.off_mask(6));
      stage(IH{}, I1{}, I8{}, hA, &dout, hB, true, KS_H, false, [&]() { ws_store(WL.off_dfeat(), KS_H, hA, KS_H); });
    } else {
      mk_cur = mask_load(AL.off_mask(7));
      mk_next = mask_load(AL.off_mask(6));
      // d z7 = relu'(h7) * (W_out^T d raw)                  -> hB
      stage(I1{}, I0{}, I8{}, &dout, &dout, hB, true, KS_H, false, [&]() { ws_store(WL.off_dout(), 1, &dout, 1); });
    }
    // d z_{i-1} = relu'(h_{i-1}) * (W_i^T d z_i), i = 7..1 ; d z7 is in hB
    for (int it = 0; it < 3; ++it) {
      const int i = 7 - 2 * it;  // consumes d z_i from hB
      mk_cur = mk_next;
      mk_next = mask_load(AL.off_mask(i - 2));
      stage(IH{}, I0{}, I8{}, hB, hB, hA, true, KS_H, false, [&]() { ws_store(WL.off_dz(i), KS_H, hB, KS_H); });
      mk_cur = mk_next;
      mk_next = mask_load(AL.off_mask(i - 3 >= 0 ? i - 3 : 0));
      stage(IH{}, I0{}, I8{}, hA, hA, hB, true, KS_H, false, [&]() { ws_store(WL.off_dz(i - 1), KS_H, hA, KS_H); });
    }
    // i = 1: d z0 from d z1 (hB) -> hA; prefetch wraps to the first chunk for the next tile
    mk_cur = mk_next;
    stage(IH{}, I0{}, I8{}, hB, hB, hA, true, more ? FIRST_KS : 0, more, [&]() { ws_store(WL.off_dz(1), KS_H, hB, KS_H); });
    ws_store(WL.off_dz(0), KS_H, hA, KS_H);
  }
}

// ------------------------------------------------------------------------------------------
// 2. wgrad (split-K over samples)
// ------------------------------------------------------------------------------------------
struct WgradJob {
  int64_t a_off, b_off;     // byte offsets of the [n_tiles][ks KiB] sections (A in ws, B in act)
  int a_ks, b_ks;           // KiB per tile
  int nta, ntb;             // 32-row / 32-column output tiles
  int a_kind, b_kind;       // SrcKind of the k-slot order (for the reduce scatter)
  int w_off, ld, col_off;   // destination weight matrix
  int row_off, rows_valid;  // OUT sources: weight row = channel - row_off
  int cols_valid;           // valid true columns of the B side
  int bias_off;             // destination bias or -1
  int split_begin, n_splits;
  int64_t part_off;         // float offset of this job's partials [n_splits][nta*32][ntb*32] (+ bias [n_splits][nta*32])
  int64_t bias_part_off;
};
constexpr int kMaxJobs = 16;
struct WgradArgs {
  int n_jobs;
  WgradJob job[kMaxJobs];
  const char* act;
  const char* ws;
  float* part;
  int64_t n_tiles;
  int L_pts, L_dir;
};

template <int P> struct WgradCfg;
template <> struct WgradCfg<kBF16> { static constexpr int TILES_PER_STEP = 2; };
template <> struct WgradCfg<kFP32> { static constexpr int TILES_PER_STEP = 1; };

template <int P>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(WgradArgs a) {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int TPS = WgradCfg<P>::TILES_PER_STEP;
  constexpr int KS_H = Blob<P>::KS_H;
  constexpr int TILE_BYTES_MAX = 2 * KS_H * 1024;
  constexpr int SLOT = TPS * TILE_BYTES_MAX;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int ji = 0;
  while (ji + 1 < a.n_jobs && a.job[ji + 1].split_begin <= (int)blockIdx.x) ++ji;
  const WgradJob& J = a.job[ji];
  const int split = blockIdx.x - J.split_begin;
  const int64_t n_steps = (a.n_tiles + TPS - 1) / TPS;
  const int64_t s0 = n_steps * split / J.n_splits, s1 = n_steps * (split + 1) / J.n_splits;
  const int tile_bytes = (J.a_ks + J.b_ks) * 1024;
  const int pieces = TPS * (J.a_ks + J.b_ks);

  auto issue = [&](int64_t step, int slot) {
    char* dst = smem + slot * SLOT;
    for (int p = wave; p < pieces; p += 4) {
      const int t = p / (J.a_ks + J.b_ks), blk = p - t * (J.a_ks + J.b_ks);
      int64_t tile = step * TPS + t;
      if (tile >= a.n_tiles) tile = a.n_tiles - 1;  // tail: duplicated tile, masked out below
      const char* src = blk < J.a_ks ? a.ws + J.a_off + (tile * J.a_ks + blk) * 1024
                                     : a.act + J.b_off + (tile * J.b_ks + (blk - J.a_ks)) * 1024;
      __builtin_amdgcn_global_load_lds(src + lane * 16, SNR_LDS(dst + p * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][8];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const int ta0 = 2 * wave;

  if (s0 < s1) issue(s0, 0);
  int slot = 0;
  for (int64_t step = s0; step < s1; ++step) {
    __syncthreads();
    if (step + 1 < s1) issue(step + 1, slot ^ 1);
    const char* sbase = smem + slot * SLOT;
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      if (step * TPS + t >= a.n_tiles) break;
      const char* tb_ = sbase + t * tile_bytes;
      if constexpr (P == kBF16) {
        // lane -> (group G, i'): G>>1 = sample half g, G&1 = which 16-neuron block of the 32-row tile
        const int G = lane >> 4, ip = lane & 15, gg = G >> 1, bh = G & 1, c = ip & 3, r = ip >> 2;
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        auto tr = [&](const char* p) {
          return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
        };
        auto frag_at = [&](const char* sec, int ks, int tI, int half) {
          int blk = 2 * tI + bh;
          if (blk >= ks) blk = ks - 1;
          // physical row of sample s in block blk is s ^ (4*(blk&1))  (act_row)
          const int s_lo = (16 * half + 8 * gg + r) ^ ((blk & 1) << 2);
          const int s_hi = (16 * half + 8 * gg + 4 + r) ^ ((blk & 1) << 2);
          const bf16x4 lo = tr(sec + blk * 1024 + s_lo * 32 + c * 8);
          const bf16x4 hi = tr(sec + blk * 1024 + s_hi * 32 + c * 8);
          return Frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          Frag fa[2];
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            fa[x] = frag_at(tb_, J.a_ks, ta0 + x, half);
            if (J.bias_off >= 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) bsum[x] += (float)fa[x][e];
            }
          }
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            if (y < J.ntb) {
              const Frag fb = frag_at(tb_ + J.a_ks * 1024, J.b_ks, y, half);
#pragma unroll
              for (int x = 0; x < 2; ++x)
                if (ta0 + x < J.nta) acc[x][y] = M::mma(fa[x], fb, acc[x][y]);
            }
          }
        }
      } else {
        // fp32: A[i = neuron][k = sample 2*ks2 + g], one float per lane; saved layout [q = neuron/8][sample][8]
        const int i = lane & 31, gg = lane >> 5;
        const float* fa_base = (const float*)tb_;
        const float* fb_base = (const float*)(tb_ + J.a_ks * 1024);
        auto elem = [&](const float* sec, int ks, int tI, int s) {
          int q = 4 * tI + (i >> 3);
          if (q >= ks) q = ks - 1;
          return sec[(q * 32 + s) * 8 + (i & 7)];
        };
#pragma unroll 4
        for (int ks2 = 0; ks2 < 16; ++ks2) {
          const int s = 2 * ks2 + gg;
          float fa[2];
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            fa[x] = elem(fa_base, J.a_ks, ta0 + x, s);
            if (J.bias_off >= 0) bsum[x] += fa[x];
          }
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            if (y < J.ntb) {
              const float fb = elem(fb_base, J.b_ks, y, s);
#pragma unroll
              for (int x = 0; x < 2; ++x)
                if (ta0 + x < J.nta)
                  acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x], fb, acc[x][y], 0, 0, 0);
            }
          }
        }
      }
    }
    slot ^= 1;
  }

  // partials: [split][nta*32][ntb*32] row-major
  const int NB = J.ntb * 32;
  float* part = a.part + J.part_off + (int64_t)split * J.nta * 32 * NB;
  const int jj = lane & 31, gq = lane >> 5;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    if (ta0 + x >= J.nta) continue;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      if (y >= J.ntb) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (ta0 + x) + (r & 3) + 8 * (r >> 2) + 4 * gq;
        part[(int64_t)row * NB + 32 * y + jj] = acc[x][y][r];
      }
    }
    if (J.bias_off >= 0) {
      float bs = bsum[x];
      if constexpr (P == kBF16) {
        // lanes l and l^32 hold the two sample halves of the same neuron
        bs += __shfl_xor(bs, 32, 64);
        if (lane < 32) {
          // lane -> neuron row of the tile: group bh = (lane>>4)&1, ip = lane&15  => row = 16*bh + ip = lane
          a.part[J.bias_part_off + (int64_t)split * J.nta * 32 + 32 * (ta0 + x) + lane] = bs;
        }
      } else {
        bs += __shfl_xor(bs, 32, 64);
        if (lane < 32) a.part[J.bias_part_off + (int64_t)split * J.nta * 32 + 32 * (ta0 + x) + lane] = bs;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// 3. reduce + scatter to the reference's parameter layout
// ------------------------------------------------------------------------------------------
template <int P> __device__ __forceinline__ int slot_true_index(int kind, int x, int L) {
  constexpr int EPF = Prec<P>::EPF;
  const int q = x / (2 * EPF), g = (x % (2 * EPF)) / EPF, e = x % EPF;
  if (kind == SRC_H) return h_slot_neuron<P>(q, g, e);
  if (kind == SRC_ENC_PTS || kind == SRC_ENC_DIR) return enc_slot_feature<P>(q, g, e, L);
  return (P == kBF16) ? 8 * g + e : 2 * e + g;  // SRC_OUT (single frag): raw channel
}

template <int P>
__global__ void mlp_wgrad_reduce_kernel(WgradArgs a, float* __restrict__ grad) {
  // blockIdx.y = job; threads cover (row a, col b) plus one extra column (b == NB) for the bias
  const WgradJob& J = a.job[blockIdx.y];
  const int NA = J.nta * 32, NB = J.ntb * 32;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)NA * (NB + 1)) return;
  const int ra = (int)(idx / (NB + 1)), cb = (int)(idx % (NB + 1));
  if (ra >= J.a_ks * 2 * Prec<P>::EPF) return;
  int n = slot_true_index<P>(J.a_kind, ra, 0) - (J.a_kind == SRC_OUT ? J.row_off : 0);
  if (n < 0 || n >= J.rows_valid) return;
  if (cb == NB) {
    if (J.bias_off < 0) return;
    float s = 0.f;
    for (int sp = 0; sp < J.n_splits; ++sp) s += a.part[J.bias_part_off + (int64_t)sp * NA + ra];
    grad[J.bias_off + n] += s;
    return;
  }
  if (cb >= J.b_ks * 2 * Prec<P>::EPF) return;
  const int k = slot_true_index<P>(J.b_kind, cb, J.b_kind == SRC_ENC_DIR ? a.L_dir : a.L_pts);
  if (k < 0 || k >= J.cols_valid) return;
  float s = 0.f;
  const float* p = a.part + J.part_off + (int64_t)ra * NB + cb;
  for (int sp = 0; sp < J.n_splits; ++sp) s += p[(int64_t)sp * NA * NB];
  grad[J.w_off + (int64_t)n * J.ld + J.col_off + k] += s;
}

// ------------------------------------------------------------------------------------------
// host: job list
// ------------------------------------------------------------------------------------------
template <int P>
static WgradArgs make_jobs(const snr_mlp_config* c, int64_t n_samples, int64_t* part_floats, int* total_splits) {
  using B = Blob<P>;
  const int vd = c->use_viewdirs;
  const ParamLayout L = make_param_layout(c->multires, c->multires_views, vd, c->out_ch, c->i_embed == -1);
  const ActLayout<P> AL(n_samples, vd);
  const WsLayout<P> WL(n_samples, vd);
  WgradArgs A{};
  A.n_tiles = AL.n_tiles;
  A.L_pts = c->i_embed == -1 ? 0 : c->multires;
  A.L_dir = c->i_embed == -1 ? 0 : c->multires_views;
  int n = 0;
  auto add = [&](int64_t a_off, int a_ks, int a_kind, int nta, int64_t b_off, int b_ks, int b_kind, int ntb,
                 int64_t w_off, int ld, int col_off, int row_off, int rows_valid, int cols_valid, int64_t bias_off) {
    WgradJob& J = A.job[n++];
    J.a_off = a_off; J.a_ks = a_ks; J.a_kind = a_kind; J.nta = nta;
    J.b_off = b_off; J.b_ks = b_ks; J.b_kind = b_kind; J.ntb = ntb;
    J.w_off = (int)w_off; J.ld = ld; J.col_off = col_off; J.row_off = row_off; J.rows_valid = rows_valid;
    J.cols_valid = cols_valid; J.bias_off = (int)bias_off;
  };
  const int ip = L.in_pts;
  add(WL.off_dz(0), B::KS_H, SRC_H, 8, AL.off_pe(), B::KS_PE, SRC_ENC_PTS, 2, L.w_pts[0], ip, 0, 0, kW, ip, L.b_pts[0]);
  for (int i = 1; i < 8; ++i) {
    if (i == kSkip + 1) {
      add(WL.off_dz(i), B::KS_H, SRC_H, 8, AL.off_pe(), B::KS_PE, SRC_ENC_PTS, 2, L.w_pts[i], kW + ip, 0, 0, kW, ip,
          L.b_pts[i]);
      add(WL.off_dz(i), B::KS_H, SRC_H, 8, AL.off_h(i - 1), B::KS_H, SRC_H, 8, L.w_pts[i], kW + ip, ip, 0, kW, kW, -1);
    } else {
      add(WL.off_dz(i), B::KS_H, SRC_H, 8, AL.off_h(i - 1), B::KS_H, SRC_H, 8, L.w_pts[i], kW, 0, 0, kW, kW, L.b_pts[i]);
    }
  }
  if (vd) {
    add(WL.off_dfeat(), B::KS_H, SRC_H, 8, AL.off_h(7), B::KS_H, SRC_H, 8, L.w_feat, kW, 0, 0, kW, kW, L.b_feat);
    add(WL.off_dout(), 1, SRC_OUT, 1, AL.off_h(7), B::KS_H, SRC_H, 8, L.w_alpha, kW, 0, 3, 1, kW, L.b_alpha);
    add(WL.off_dz9(), B::KS_H9, SRC_H, 4, AL.off_feat(), B::KS_H, SRC_H, 8, L.w_views, kW + L.in_dir, 0, 0, kW / 2, kW,
        L.b_views);
    if (L.in_dir > 0)
      add(WL.off_dz9(), B::KS_H9, SRC_H, 4, AL.off_dir(), B::KS_DIR, SRC_ENC_DIR, 1, L.w_views, kW + L.in_dir, kW, 0,
          kW / 2, L.in_dir, -1);
    add(WL.off_dout(), 1, SRC_OUT, 1, AL.off_h9(), B::KS_H9, SRC_H, 4, L.w_rgb, kW / 2, 0, 0, 3, kW / 2, L.b_rgb);
  } else {
    add(WL.off_dout(), 1, SRC_OUT, 1, AL.off_h(7), B::KS_H, SRC_H, 8, L.w_out, kW, 0, 0, c->out_ch, kW, L.b_out);
  }
  A.n_jobs = n;
  // split-K: the kernel is HBM-bound, so give each job workgroups in proportion to the bytes it streams
  const int64_t n_steps = (A.n_tiles + WgradCfg<P>::TILES_PER_STEP - 1) / WgradCfg<P>::TILES_PER_STEP;
  int64_t cost = 0;
  for (int i = 0; i < n; ++i) cost += A.job[i].a_ks + A.job[i].b_ks;
  const int target = 512;
  int sb = 0;
  int64_t po = 0;
  for (int i = 0; i < n; ++i) {
    WgradJob& J = A.job[i];
    int64_t s = ((int64_t)target * (J.a_ks + J.b_ks) + cost / 2) / cost;
    if (s < 1) s = 1;
    if (s > n_steps) s = n_steps;
    J.n_splits = (int)s; J.split_begin = sb; sb += (int)s;
    J.part_off = po; po += s * J.nta * 32 * J.ntb * 32;
    J.bias_part_off = po; po += s * J.nta * 32;
  }
  *part_floats = po;
  *total_splits = sb;
  return A;
}

}  // namespace snr

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
  int64_t pf; int ts;
  make_jobs<P>(c, n, &pf, &ts);
  return WsLayout<P>(n, c->use_viewdirs).dz_bytes() + pf * 4;
}

extern "C" int64_t snr_mlp_bwd_ws_bytes(const snr_mlp_config* c, int64_t n) {
  int st = check_cfg_b(c);
  if (st != SNR_OK) return st;
  if (n <= 0) return SNR_ERR_SHAPE;
  return c->precision == SNR_PREC_BF16 ? ws_bytes<kBF16>(c, n) : ws_bytes<kFP32>(c, n);
}

template <int P, bool VD>
static int launch_dgrad(const DgradArgs& a, int64_t n_wg, hipStream_t s) {
  const int lds = 2 * Pipe<P>::SLOT;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_dgrad_kernel<P, VD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t grid = n_wg < 1024 ? n_wg : 1024;
  mlp_dgrad_kernel<P, VD><<<dim3((unsigned)grid), dim3(256), lds, s>>>(a);
  return launch_status();
}

template <int P>
static int backward_impl(const snr_mlp_config* c, const void* packed, const float* d_raw, int64_t n, const void* act,
                         void* ws, float* grad, int accumulate, hipStream_t s) {
  const PackTable T = make_pack_table<P>(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  const ParamLayout L = make_param_layout(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)L.total * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  DgradArgs d{};
  d.blob_bwd = (const char*)packed + (int64_t)T.fwd_frags * 1024;
  d.d_raw = d_raw; d.n_samples = n; d.out_ch = c->out_ch;
  d.act = (const char*)act; d.ws = (char*)ws;
  const int64_t n_wg = (n + 127) / 128;
  int st = c->use_viewdirs ? launch_dgrad<P, true>(d, n_wg, s) : launch_dgrad<P, false>(d, n_wg, s);
  if (st != SNR_OK) return st;

  int64_t pf; int total_splits;
  WgradArgs w = make_jobs<P>(c, n, &pf, &total_splits);
  w.act = (const char*)act;
  w.ws = (const char*)ws;
  w.part = (float*)((char*)ws + WsLayout<P>(n, c->use_viewdirs).dz_bytes());
  constexpr int lds = 2 * WgradCfg<P>::TILES_PER_STEP * 2 * Blob<P>::KS_H * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_wgrad_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  mlp_wgrad_kernel<P><<<dim3((unsigned)total_splits), dim3(256), lds, s>>>(w);
  st = launch_status();
  if (st != SNR_OK) return st;
  const int per_job = 256 * 257;
  mlp_wgrad_reduce_kernel<P><<<dim3((per_job + 255) / 256, (unsigned)w.n_jobs), dim3(256), 0, s>>>(w, grad);
  return launch_status();
}

extern "C" int snr_mlp_backward(const snr_mlp_config* c, const void* packed, const float* d_raw, int64_t n,
                                const void* act, void* ws, float* grad, int accumulate, snr_stream_t stream) {
  int st = check_cfg_b(c);
  if (st != SNR_OK) return st;
  SNR_CHECK_ARG(packed && d_raw && act && ws && grad, SNR_ERR_NULL);
  SNR_CHECK_ARG(n > 0, SNR_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  return c->precision == SNR_PREC_BF16 ? backward_impl<kBF16>(c, packed, d_raw, n, act, ws, grad, accumulate, s)
                                       : backward_impl<kFP32>(c, packed, d_raw, n, act, ws, grad, accumulate, s);
}
