// Weight-gradient pass of the fused NeRF MLP backward (gfx950): split-K kernel + reduce/scatter.
// Included by mlp_bwd.hip.  See the header comment there for the overall backward structure.
#pragma once
#include <stdlib.h>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {

// backward scratch: d z sections ([n_tiles][ks KiB], like ActLayout) followed by wgrad partials
template <int P> struct WsLayout {
  using B = Blob<P>;
  int64_t n_tiles;
  int vd;
  SNR_HD WsLayout(int64_t n_samples, int vd_) : n_tiles(padded_tiles<P>(n_samples)), vd(vd_) {}
  SNR_HD int k_dout() const { return 0; }
  SNR_HD int k_dz(int i) const { return 1 + i * B::KS_H; }  // i in 0..7
  SNR_HD int k_dfeat() const { return 1 + 8 * B::KS_H; }
  SNR_HD int k_dz9() const { return k_dfeat() + B::KS_H; }
  SNR_HD int64_t off_dout() const { return 0; }
  SNR_HD int64_t off_dz(int i) const { return n_tiles * 1024 * k_dz(i); }
  SNR_HD int64_t off_dfeat() const { return n_tiles * 1024 * k_dfeat(); }
  SNR_HD int64_t off_dz9() const { return n_tiles * 1024 * k_dz9(); }
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

// LDS ring of whole tiles (A section | B section of 32 samples).  bf16: 5 slots x 32 KiB (all 160 KiB of
// the CU), 4 tiles in flight — 0.421 ms vs 0.443 ms with 4 slots / 3 in flight at 196 608 samples;
// fp32: 2 slots x 64 KiB, 1 in flight.
template <int P> struct WgradCfg;
template <> struct WgradCfg<kBF16> { static constexpr int RING = 5, DEPTH = 4; };
template <> struct WgradCfg<kFP32> { static constexpr int RING = 2, DEPTH = 1; };

// Everything a workgroup needs from its job, copied to registers once (the job table lives in the
// kernarg segment; indexing it inside the hot loop costs a scalar load per use).
struct WgradLocal {
  const char* a_base;
  const char* b_base;
  float* part;
  float* bias_part;  // null = no bias
  int a_ks, b_ks, ntb_total;
  int64_t t0, t1;    // tile range of this split
};

// 8 waves per workgroup = two per SIMD: while one wave sits in the vector-memory issue queue (DMA)
// or at a wait, its SIMD partner keeps the matrix pipe busy.  Each wave owns one 32-row tile of the
// output (128 accumulator registers) across all column tiles.
constexpr int kWgradWaves = 8;
#ifndef SNR_WGRAD_AUX
#define SNR_WGRAD_AUX 0
#endif

#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2
#define SNR_WGRAD_ISSUE(k) (void)0
#else
#define SNR_WGRAD_ISSUE(k) issue_piece(islot, (k))
#endif

// The whole life of one wave of a workgroup for a job whose output is NTB column tiles wide, of which
// this wave owns NX (0..1) row tiles starting at ta0, and whose 32-sample tile is NI DMA
// instructions per wave (short tiles re-load their last piece so every wave issues exactly NI: the
// counted wait needs an immediate).  All waves (any NX) run the same barriers and DMA.
// Everything that does not change from tile to tile (DMA source pointers, LDS offsets of the
// transposing reads) is computed once up front; the per-tile loop is waits, 2*NI... DMA issues,
// ds_read_b64_tr_b16 with immediate offsets, and MFMAs.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void tr_read(bf16x4& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void tr_wait(bf16x4& a, bf16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N < 15 ? N : 15));
}
template <int N> __device__ __forceinline__ void tr_wait(bf16x4& a, bf16x4& b, bf16x4& c, bf16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));
}

template <int P, int NTB, int NX, int NI>
__device__ __forceinline__ void wgrad_run(const WgradLocal& L, char* smem, int wave, int lane, int ta0, int ni) {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int R = WgradCfg<P>::RING, D = WgradCfg<P>::DEPTH;
  constexpr int SLOT = 2 * Blob<P>::KS_H * 1024;
  const int a_ks = L.a_ks, b_ks = L.b_ks;
  const int per_tile = a_ks + b_ks;
  const bool do_bias = L.bias_part != nullptr;

  // ---- DMA pieces of this wave: wave + 8k (clamped), source pointer advances one tile per tile ----
  const char* src[NI];
  int lds_off[NI], stride[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    int p = wave + kWgradWaves * k;
    if (p >= per_tile) p = per_tile - 1;
    const bool isA = p < a_ks;
    src[k] = (isA ? L.a_base + (L.t0 * a_ks + p) * 1024 : L.b_base + (L.t0 * b_ks + (p - a_ks)) * 1024) + lane * 16;
    stride[k] = (isA ? a_ks : b_ks) * 1024;
    lds_off[k] = p * 1024;
  }
  int64_t src_tile = L.t0;
  auto issue_piece = [&](int slot, int k) {
    // (non-temporal aux = 2 measured no faster: 0.76 vs 0.74 ms)
    // (s_nop: no LDS read may sit in the cycle in front of an LDS-DMA — mlp_device.h, Pipe::issue_one)
    if (k < ni) asm volatile("s_nop 0");
    if (k < ni) __builtin_amdgcn_global_load_lds(src[k], SNR_LDS(smem + slot * SLOT + lds_off[k]), 16, 0, SNR_WGRAD_AUX);
  };
  auto advance = [&]() {   // past the end the last tile is re-loaded: the instruction count stays uniform
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 4   // timing experiment: the stream re-reads one tile (L2 hits)
    if (false) {
#else
    if (src_tile + 1 < L.t1) {
#endif
      ++src_tile;
#pragma unroll
      for (int k = 0; k < NI; ++k) src[k] += stride[k];
    }
  };

  f32x16 acc[NX > 0 ? NX : 1][NTB];
#pragma unroll
  for (int x = 0; x < (NX > 0 ? NX : 1); ++x)
#pragma unroll
    for (int y = 0; y < NTB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float bsum = 0.f;

  // ---- tile-invariant LDS offsets of the reads ----
  // bf16: 16-lane group G reads a [4 samples][16 neurons] block and receives its column `ip`;
  // G>>1 = sample half gg (k-slots 8gg..8gg+7), G&1 = bh = which 16-neuron block of the 32-row tile;
  // odd blocks store sample row r at r^4 (act_row).  Read q (0/1) fetches samples 8gg+4q..+3, so for
  // block parity bh the physical row is 8gg + (4q ^ 4bh) + r4.  Offset within a tile section:
  //   (2*tI + bh)*1024 + row*32 + c4*8 + half*512 ; everything but tI and half is per-lane constant.
  const int G = lane >> 4, ip = lane & 15, gg = G >> 1, bh = G & 1, c4 = ip & 3, r4 = ip >> 2;
  const int bhA = a_ks == 1 ? 0 : bh;   // 16-wide OUT sections have one block: lanes of block 1 re-read block 0
  int offA[2], offB[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    offA[q] = (2 * ta0 + bhA) * 1024 + (8 * gg + ((4 * q) ^ (4 * bhA)) + r4) * 32 + c4 * 8;
    offB[q] = a_ks * 1024 + bh * 1024 + (8 * gg + ((4 * q) ^ (4 * bh)) + r4) * 32 + c4 * 8;
  }
  const int i32 = lane & 31, g32 = lane >> 5;

#if !(defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2)
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < NI; ++k) issue_piece(d % R, k);
    advance();
  }
#endif
  int slot = 0, islot = D % R;
  for (int64_t tile = L.t0; tile < L.t1; ++tile) {
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2   // timing experiment: compute on whatever is in LDS
    __builtin_amdgcn_s_barrier();
#else
    // this wave's pieces of `tile` have landed (DMA loads retire in order; nothing else is outstanding)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D == 1 ? 0 : NI * (D - 1)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and its reads of the previous tile are done
    __builtin_amdgcn_s_barrier();                           // same for everybody: slot of tile-1 is free
    asm volatile("" ::: "memory");
#endif
    int kq = 0;   // DMA pieces of tile + D issued so far
    if constexpr (NX > 0) {
      char* sbase = smem + slot * SLOT;
      if constexpr (P == kBF16) {
        // The transposing reads are inline asm with hand-counted completion (mlp_device.h, Pipe: reads the
        // compiler knows about are ordered behind every outstanding LDS-DMA load, i.e. vmcnt(0) per read).
        const uint32_t sb32 = lds_addr(smem) + slot * SLOT;
        const uint32_t aA0 = sb32 + offA[0], aA1 = sb32 + offA[1], aB0 = sb32 + offB[0], aB1 = sb32 + offB[1];
        static_for<0, 2>([&](auto H_) {
          constexpr int half = decltype(H_)::value;
          bf16x4 alo, ahi, blo[NTB], bhi[NTB];
          tr_read<half * 512>(alo, aA0);
          tr_read<half * 512>(ahi, aA1);
          static_for<0, NTB>([&](auto Y_) {
            constexpr int y = decltype(Y_)::value;
            tr_read<y * 2048 + half * 512>(blo[y], aB0);
            tr_read<y * 2048 + half * 512>(bhi[y], aB1);
          });
          Frag fa;
          static_for<0, NTB>([&](auto Y_) {
            constexpr int y = decltype(Y_)::value;
            // LDS reads retire in order: everything up to B pair y has landed once at most the
            // 2*(NTB-1-y) younger reads are outstanding
            if constexpr (y == 0) {
              tr_wait<2 * (NTB - 1)>(alo, ahi, blo[0], bhi[0]);
              fa = Frag{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
              if (do_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += (float)fa[e];
              }
            } else {
              tr_wait<2 * (NTB - 1 - y)>(blo[y], bhi[y]);
            }
            const Frag fb = Frag{blo[y][0], blo[y][1], blo[y][2], blo[y][3], bhi[y][0], bhi[y][1], bhi[y][2], bhi[y][3]};
            acc[0][y] = M::mma(fa, fb, acc[0][y]);
            if ((NTB < 8 || (y & 1)) && kq < NI) { SNR_WGRAD_ISSUE(kq); ++kq; }
          });
        });
      } else {
        // fp32: A[i = neuron][k = sample 2*ks2 + g], one float per lane; saved layout [q = neuron/8][sample][8]
        const float* fa_base = (const float*)sbase;
        const float* fb_base = (const float*)(sbase + a_ks * 1024);
        auto elem = [&](const float* sec, int ks, int tI, int sidx) {
          int q = 4 * tI + (i32 >> 3);
          if (q >= ks) q = ks - 1;
          return sec[(q * 32 + sidx) * 8 + (i32 & 7)];
        };
#pragma unroll 4
        for (int ks2 = 0; ks2 < 16; ++ks2) {
          const int sidx = 2 * ks2 + g32;
          const float fa = elem(fa_base, a_ks, ta0, sidx);
          if (do_bias) bsum += fa;
#pragma unroll
          for (int y = 0; y < NTB; ++y) {
            const float fb = elem(fb_base, b_ks, y, sidx);
            acc[0][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0][y], 0, 0, 0);
          }
          if (kq < NI) { SNR_WGRAD_ISSUE(kq); ++kq; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (k >= kq) { SNR_WGRAD_ISSUE(k); }
#if !(defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2)
    advance();
#endif
    islot = islot + 1 == R ? 0 : islot + 1;
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing re-loads

  if constexpr (NX > 0) {
    // partials: [nta*32][ntb*32] row-major for this split
    const int NB = L.ntb_total * 32;
#pragma unroll
    for (int y = 0; y < NTB; ++y) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * ta0 + (r & 3) + 8 * (r >> 2) + 4 * g32;
        L.part[(int64_t)row * NB + 32 * y + i32] = acc[0][y][r];
      }
    }
    if (do_bias) {
      // lanes l and l^32 hold the two sample halves of the same neuron row (row = lane & 31)
      const float bs = bsum + __shfl_xor(bsum, 32, 64);
      if (lane < 32) L.bias_part[32 * ta0 + lane] = bs;
    }
  }
}

template <int P>
__global__ __launch_bounds__(64 * kWgradWaves) void mlp_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int ji = 0;
  while (ji + 1 < a.n_jobs && a.job[ji + 1].split_begin <= (int)blockIdx.x) ++ji;
  const WgradJob& J = a.job[ji];
  const int split = blockIdx.x - J.split_begin;
  WgradLocal L;
  L.a_base = a.ws + J.a_off;
  L.b_base = a.act + J.b_off;
  L.a_ks = J.a_ks; L.b_ks = J.b_ks; L.ntb_total = J.ntb;
  L.t0 = a.n_tiles * split / J.n_splits;
  L.t1 = a.n_tiles * (split + 1) / J.n_splits;
  const int nta = J.nta, ntb = J.ntb;
  L.part = a.part + J.part_off + (int64_t)split * nta * 32 * ntb * 32;
  L.bias_part = J.bias_off >= 0 ? a.part + J.bias_part_off + (int64_t)split * nta * 32 : nullptr;
  const int ta0 = wave;
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 1   // timing experiment: DMA stream only
  const int nx = 0;
#else
  const int nx = ta0 < nta ? 1 : 0;
#endif

  // shapes that occur (rows x cols in 32-tiles): 8x8 (256x256), 8x2 (x encodings), 4x8 / 4x1 (views
  // layer), 1x8 / 1x4 (heads); ni = DMA instructions per wave per tile
  const int ni = (J.a_ks + J.b_ks + kWgradWaves - 1) / kWgradWaves;
#define SNR_RUN(NTB_, NX_, NI_) wgrad_run<P, NTB_, NX_, NI_>(L, smem, wave, lane, ta0, ni)
  if constexpr (P == kFP32) {   // one tile in flight: the wait immediate is 0 whatever ni is
    if (nx == 0) SNR_RUN(1, 0, 8);
    else if (ntb == 8) SNR_RUN(8, 1, 8);
    else if (ntb == 4) SNR_RUN(4, 1, 8);
    else if (ntb == 2) SNR_RUN(2, 1, 8);
    else SNR_RUN(1, 1, 8);
  } else if (ni == 4) {
    if (nx == 0) SNR_RUN(1, 0, 4); else SNR_RUN(8, 1, 4);
  } else if (ni == 3) {
    if (nx == 0) SNR_RUN(1, 0, 3);
    else if (ntb == 8) SNR_RUN(8, 1, 3);
    else SNR_RUN(2, 1, 3);
  } else {
    if (nx == 0) SNR_RUN(1, 0, 2);
    else if (ntb == 4) SNR_RUN(4, 1, 2);
    else SNR_RUN(1, 1, 2);
  }
#undef SNR_RUN
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
__global__ void mlp_wgrad_reduce_kernel(WgradArgs a, float* __restrict__ grad, int accumulate) {
  // Every parameter element a job produces is produced by exactly one (job, row, column): without
  // `accumulate` the result is stored, not added, and the gradient buffer needs no clearing first
  // (tests/test_gpu_kernels.py: test_mlp_backward_overwrites_every_element).
  // blockIdx.y = job; a thread covers (row, 4 consecutive partial-sum columns); one extra thread per row does
  // the bias.  16-byte loads, two independent accumulators per element.
  const WgradJob& J = a.job[blockIdx.y];
  const int NA = J.nta * 32, NB = J.ntb * 32, NQ = NB / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)NA * (NQ + 1)) return;
  const int ra = (int)(idx / (NQ + 1)), q = (int)(idx % (NQ + 1));
  if (ra >= J.a_ks * 2 * Prec<P>::EPF) return;
  const int n = slot_true_index<P>(J.a_kind, ra, 0) - (J.a_kind == SRC_OUT ? J.row_off : 0);
  if (n < 0 || n >= J.rows_valid) return;
  if (q == NQ) {
    if (J.bias_off < 0) return;
    float s = 0.f;
    for (int sp = 0; sp < J.n_splits; ++sp) s += a.part[J.bias_part_off + (int64_t)sp * NA + ra];
    grad[J.bias_off + n] = accumulate ? grad[J.bias_off + n] + s : s;
    return;
  }
  const int cb = 4 * q;
  if (cb >= J.b_ks * 2 * Prec<P>::EPF) return;
  const float* p = a.part + J.part_off + (int64_t)ra * NB + cb;
  const int64_t st = (int64_t)NA * NB;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int sp = 0;
  for (; sp + 2 <= J.n_splits; sp += 2) {
    s0 += *(const f32x4*)(p + (sp + 0) * st);
    s1 += *(const f32x4*)(p + (sp + 1) * st);
  }
  if (sp < J.n_splits) s0 += *(const f32x4*)(p + sp * st);
  const f32x4 s = s0 + s1;
  float* grow = grad + J.w_off + (int64_t)n * J.ld + J.col_off;
  const int L = J.b_kind == SRC_ENC_DIR ? a.L_dir : a.L_pts;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = slot_true_index<P>(J.b_kind, cb + e, L);
    if (k < 0 || k >= J.cols_valid) continue;
    grow[k] = accumulate ? grow[k] + s[e] : s[e];
  }
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
  // split-K: the kernel streams saved activations once and is bound by that stream, so every workgroup
  // gets the same number of bytes: job i receives target * bytes_i / bytes workgroups, apportioned by
  // largest remainder so that the total is exactly one workgroup per CU — measured on MI355X (bench
  // workload): 224-256 workgroups 0.57 ms wgrad + 0.04 ms reduce per step, 512 (two rounds, twice the
  // partial sums) 0.59 + 0.08, 128 0.84.  SNR_WGRAD_SPLITS overrides the total for experiments.
  const int64_t n_steps = A.n_tiles;
  int64_t cost = 0;
  for (int i = 0; i < n; ++i) cost += A.job[i].a_ks + A.job[i].b_ks;
  int target = 256;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      target = cus;
    else
      (void)hipGetLastError();   // no device (size queries on a CPU-only host): keep the default
  }
  if (const char* e = getenv("SNR_WGRAD_SPLITS")) target = atoi(e) > 0 ? atoi(e) : target;
  if (target < n) target = n;
  int64_t rem[kMaxJobs];
  int assigned = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t num = (int64_t)target * (A.job[i].a_ks + A.job[i].b_ks);
    int64_t s = num / cost;
    rem[i] = num % cost;
    if (s < 1) { s = 1; rem[i] = -1; }
    A.job[i].n_splits = (int)s;
    assigned += (int)s;
  }
  while (assigned < target) {   // hand the left-over workgroups to the largest remainders
    int best = 0;
    for (int i = 1; i < n; ++i) if (rem[i] > rem[best]) best = i;
    if (rem[best] < 0) break;
    ++A.job[best].n_splits; rem[best] = -1; ++assigned;
  }
  int sb = 0;
  int64_t po = 0;
  for (int i = 0; i < n; ++i) {
    WgradJob& J = A.job[i];
    int64_t s = J.n_splits;
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
