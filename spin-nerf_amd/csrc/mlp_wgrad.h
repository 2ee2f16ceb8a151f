// Weight-gradient pass of the fused NeRF MLP backward (gfx950): split-K kernel + reduce/scatter.
// Included by mlp_bwd.hip.  See the header comment there for the overall backward structure.
#pragma once
#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {

// backward scratch: d z sections ([n_tiles][ks KiB], like ActLayout) followed by wgrad partials
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

// One job = one (d z section) x (activation section) product = the gradient of one weight block.
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
  int64_t part_off;         // float offset of this job's partials [n_splits][nta*32][ntb*32]
  int64_t bias_part_off;    // ... and of its bias partials [n_splits][nta*32]
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

// Everything a workgroup needs from its job, copied to registers once (the job table lives in the
// kernarg segment; indexing it inside the hot loop costs a scalar load per use).
struct WgradLocal {
  const char* a_base;
  const char* b_base;
  float* part;
  float* bias_part;  // null = no bias
  int a_ks, b_ks, ntb_total;
  int64_t s0, s1, n_tiles;
};

// The whole life of one wave of a workgroup for a job whose output is NTB column tiles wide and
// of which this wave owns NX (0..2) row tiles starting at ta0.  All waves (any NX) run the same
// number of barriers and issue their share of the DMA.
template <int P, int NTB, int NX>
__device__ __forceinline__ void wgrad_run(const WgradLocal& L, char* smem, int wave, int lane, int ta0) {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int TPS = WgradCfg<P>::TILES_PER_STEP;
  constexpr int SLOT = TPS * 2 * Blob<P>::KS_H * 1024;
  const int a_ks = L.a_ks, b_ks = L.b_ks;
  const int per_tile = a_ks + b_ks;
  const int tile_bytes = per_tile * 1024;
  const int pieces = TPS * per_tile;
  const bool do_bias = L.bias_part != nullptr;

  auto issue = [&](int64_t step, int slot) {
    char* dst = smem + slot * SLOT;
    for (int p = wave; p < pieces; p += 4) {
      const int t = p >= per_tile ? 1 : 0;
      const int blk = p - t * per_tile;
      int64_t tile = step * TPS + t;
      if (tile >= L.n_tiles) tile = L.n_tiles - 1;  // tail: duplicated tile, skipped by the compute loop
      const char* src = blk < a_ks ? L.a_base + (tile * a_ks + blk) * 1024 : L.b_base + (tile * b_ks + (blk - a_ks)) * 1024;
      __builtin_amdgcn_global_load_lds(src + lane * 16, SNR_LDS(dst + p * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[NX > 0 ? NX : 1][NTB];
#pragma unroll
  for (int x = 0; x < (NX > 0 ? NX : 1); ++x)
#pragma unroll
    for (int y = 0; y < NTB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  // per-lane constants of the transposing read (bf16) / strided read (fp32)
  const int G = lane >> 4, ip = lane & 15, gg = G >> 1, bh = G & 1, c4 = ip & 3, r4 = ip >> 2;
  const int i32 = lane & 31, g32 = lane >> 5;

  if (L.s0 < L.s1) issue(L.s0, 0);
  int slot = 0;
  for (int64_t step = L.s0; step < L.s1; ++step) {
    __syncthreads();
    if (step + 1 < L.s1) issue(step + 1, slot ^ 1);
    if constexpr (NX > 0) {
      const char* sbase = smem + slot * SLOT;
#pragma unroll 1
      for (int t = 0; t < TPS; ++t) {
        if (step * TPS + t >= L.n_tiles) break;
        const char* secA = sbase + t * tile_bytes;
        const char* secB = secA + a_ks * 1024;
        if constexpr (P == kBF16) {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          // 16-lane group G reads a [4 samples][16 neurons] block and receives its column `ip`:
          // G>>1 = sample half (k-slots 8g..8g+7), G&1 = which 16-neuron block of the 32-row tile.
          auto frag_at = [&](const char* sec, int ks, int tI, int half) {
            int blk = 2 * tI + bh;
            if (blk >= ks) blk = ks - 1;
            const int sw = (blk & 1) << 2;   // act_row swizzle of odd blocks
            const char* base = sec + blk * 1024 + c4 * 8;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(base + ((16 * half + 8 * gg + r4) ^ sw) * 32));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(base + ((16 * half + 8 * gg + 4 + r4) ^ sw) * 32));
            return Frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          };
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {
            Frag fa[NX];
#pragma unroll
            for (int x = 0; x < NX; ++x) {
              fa[x] = frag_at(secA, a_ks, ta0 + x, half);
              if (do_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[x] += (float)fa[x][e];
              }
            }
#pragma unroll
            for (int y = 0; y < NTB; ++y) {
              const Frag fb = frag_at(secB, b_ks, y, half);
#pragma unroll
              for (int x = 0; x < NX; ++x) acc[x][y] = M::mma(fa[x], fb, acc[x][y]);
            }
          }
        } else {
          // fp32: A[i = neuron][k = sample 2*ks2 + g], one float per lane; saved layout [q = neuron/8][sample][8]
          const float* fa_base = (const float*)secA;
          const float* fb_base = (const float*)secB;
          auto elem = [&](const float* sec, int ks, int tI, int s) {
            int q = 4 * tI + (i32 >> 3);
            if (q >= ks) q = ks - 1;
            return sec[(q * 32 + s) * 8 + (i32 & 7)];
          };
#pragma unroll 4
          for (int ks2 = 0; ks2 < 16; ++ks2) {
            const int s = 2 * ks2 + g32;
            float fa[NX];
#pragma unroll
            for (int x = 0; x < NX; ++x) {
              fa[x] = elem(fa_base, a_ks, ta0 + x, s);
              if (do_bias) bsum[x] += fa[x];
            }
#pragma unroll
            for (int y = 0; y < NTB; ++y) {
              const float fb = elem(fb_base, b_ks, y, s);
#pragma unroll
              for (int x = 0; x < NX; ++x)
                acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x], fb, acc[x][y], 0, 0, 0);
            }
          }
        }
      }
    }
    slot ^= 1;
  }

  if constexpr (NX > 0) {
    // partials: [nta*32][ntb*32] row-major for this split
    const int NB = L.ntb_total * 32;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
#pragma unroll
      for (int y = 0; y < NTB; ++y) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * (ta0 + x) + (r & 3) + 8 * (r >> 2) + 4 * g32;
          L.part[(int64_t)row * NB + 32 * y + i32] = acc[x][y][r];
        }
      }
      if (do_bias) {
        // lanes l and l^32 hold the two sample halves of the same neuron row (row = lane & 31)
        const float bs = bsum[x] + __shfl_xor(bsum[x], 32, 64);
        if (lane < 32) L.bias_part[32 * (ta0 + x) + lane] = bs;
      }
    }
  }
}

template <int P>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TPS = WgradCfg<P>::TILES_PER_STEP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int ji = 0;
  while (ji + 1 < a.n_jobs && a.job[ji + 1].split_begin <= (int)blockIdx.x) ++ji;
  const WgradJob& J = a.job[ji];
  const int split = blockIdx.x - J.split_begin;
  const int64_t n_steps = (a.n_tiles + TPS - 1) / TPS;
  WgradLocal L;
  L.a_base = a.ws + J.a_off;
  L.b_base = a.act + J.b_off;
  L.a_ks = J.a_ks; L.b_ks = J.b_ks; L.ntb_total = J.ntb;
  L.n_tiles = a.n_tiles;
  L.s0 = n_steps * split / J.n_splits;
  L.s1 = n_steps * (split + 1) / J.n_splits;
  const int nta = J.nta, ntb = J.ntb;
  L.part = a.part + J.part_off + (int64_t)split * nta * 32 * ntb * 32;
  L.bias_part = J.bias_off >= 0 ? a.part + J.bias_part_off + (int64_t)split * nta * 32 : nullptr;
  const int ta0 = 2 * wave;
  const int nx = nta - ta0 >= 2 ? 2 : (nta - ta0 == 1 ? 1 : 0);

  // shapes that occur: 8x8 (256x256), 8x2 (x encodings), 4x8 / 4x1 (views layer), 1x8 / 1x4 (heads)
  if (nx == 2) {
    if (ntb == 8) wgrad_run<P, 8, 2>(L, smem, wave, lane, ta0);
    else if (ntb == 2) wgrad_run<P, 2, 2>(L, smem, wave, lane, ta0);
    else wgrad_run<P, 1, 2>(L, smem, wave, lane, ta0);
  } else if (nx == 1) {
    if (ntb == 8) wgrad_run<P, 8, 1>(L, smem, wave, lane, ta0);
    else wgrad_run<P, 4, 1>(L, smem, wave, lane, ta0);
  } else {
    wgrad_run<P, 1, 0>(L, smem, wave, lane, ta0);
  }
}

// ------------------------------------------------------------------------------------------
// reduce + scatter to the reference's parameter layout
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
  // blockIdx.y = job; threads cover (row, col) plus one extra column (col == NB) for the bias
  const WgradJob& J = a.job[blockIdx.y];
  const int NA = J.nta * 32, NB = J.ntb * 32;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)NA * (NB + 1)) return;
  const int ra = (int)(idx / (NB + 1)), cb = (int)(idx % (NB + 1));
  if (ra >= J.a_ks * 2 * Prec<P>::EPF) return;
  const int n = slot_true_index<P>(J.a_kind, ra, 0) - (J.a_kind == SRC_OUT ? J.row_off : 0);
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
  // split-K: the kernel streams saved activations once, so give each job workgroups in proportion
  // to the bytes it streams; ~2 workgroups per CU in total
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
