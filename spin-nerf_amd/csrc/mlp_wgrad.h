// Weight-gradient pass of the fused NeRF MLP backward (gfx950): split-K kernel + reduce/scatter + the
// small dense products that replace two streamed jobs.  Included by mlp_bwd.hip.  See the header comment
// there for the overall backward structure.
//
// Bytes are what this pass costs (it streams saved tensors once from HBM and is bound by that stream), so
// the job list is built to read as few of them as possible:
//   * jobs that share an operand are merged — one job's A side or B side may be the concatenation of two saved
//     sections (d z5 against [pe | h4]; [d z9 | d out] against [h7 | dir]), so the shared section is read once;
//   * feature_linear's output `feat` = W_f h7 + b_f has no activation, and neither has its gradient
//     d feat = W_v[:, :256]^T d z9.  With G = sum_s d z9[s] (x) h7[s] and s9 = sum_s d z9[s]
//         dW_views[:, :256] = sum_s d z9 (x) feat   = G W_f^T + s9 (x) b_f
//         dW_feat           = sum_s d feat (x) h7   = W_v[:, :256]^T G         db_feat = W_v[:, :256]^T s9
//     so neither `feat` nor `d feat` is written by the chain kernels or read here: one 128 x 256 job (G) and two
//     tiny dense products (wgrad_post_kernel) replace a 256 x 256 and a 128 x 256 job.
// Per sample (bf16, viewdirs) the pass reads 9.1 KB instead of 11.4 KB, and forward / dgrad write 0.5 KB less each.
#pragma once
#include <stdlib.h>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {

// backward scratch: d z sections ([n_tiles][ks KiB], like ActLayout) followed by wgrad partials and the G block
template <int P> struct WsLayout {
  using B = Blob<P>;
  int64_t n_tiles;
  int vd;
  SNR_HD WsLayout(int64_t n_samples, int vd_) : n_tiles(padded_tiles<P>(n_samples)), vd(vd_) {}
  SNR_HD int k_dout() const { return 0; }
  SNR_HD int k_dz(int i) const { return 1 + i * B::KS_H; }  // i in 0..7
  SNR_HD int k_dz9() const { return 1 + 8 * B::KS_H; }
  SNR_HD int64_t off_dout() const { return 0; }
  SNR_HD int64_t off_dz(int i) const { return n_tiles * 1024 * k_dz(i); }
  SNR_HD int64_t off_dz9() const { return n_tiles * 1024 * k_dz9(); }
  SNR_HD int64_t dz_bytes() const { return n_tiles * 1024 * (1 + 8 * (int64_t)B::KS_H + (vd ? B::KS_H9 : 0)); }
};

// G[128][256] (true neuron order) followed by s9[128]: floats behind the split-K partials
constexpr int kPostG = (kW / 2) * kW, kPostFloats = kPostG + kW / 2;

// One job = (one or two d z sections) x (one or two activation sections): a block of weight gradients.
struct WgradSec { int64_t off; int ks; };   // byte offset of the [n_tiles][ks KiB] section, KiB per tile
struct WgradJob {
  WgradSec a[2], b[2];      // A side (rows) lives in ws, B side (columns) in act; ks == 0 = unused
  int a_ks, b_ks;           // KiB per tile of each side (sum over its sections)
  int nta, ntb;             // 32-row / 32-column output tiles
  int split_begin, n_splits;
  int64_t part_off;         // float offset of this job's partials [n_splits][nta*32][ntb*32]
  int64_t bias_part_off;    // ... and of its row sums [n_splits][nta*32]
};
// One output = a rectangle of a job's partial sums and where it goes in the parameter gradient.
struct WgradOut {
  int job;
  int row0, rows, col0, cols;   // in k-slots of the job's A / B side; cols == 0: row sums (bias) only
  int a_kind, b_kind;           // SrcKind of the slot order inside the rectangle (relative to row0 / col0)
  int L;                        // multires of an encoding B side
  int w_off, ld, col_off;       // destination weight matrix (float offsets)
  int row_off, rows_valid;      // OUT sources: weight row = channel - row_off
  int cols_valid;               // valid true columns of the B side
  int bias_off;                 // destination bias or -1
  int to_scratch;               // 1: destination offsets are relative to the G block (stored, never accumulated)
};
constexpr int kMaxJobs = 12, kMaxOuts = 20;
struct WgradArgs {
  int n_jobs, n_outs;
  WgradJob job[kMaxJobs];
  WgradOut out[kMaxOuts];
  const char* act;
  const char* ws;
  float* part;
  float* post;      // G block (kPostFloats) or null
  int64_t n_tiles;
};

// LDS ring of whole tiles (A sections | B sections of 32 samples), sized by the job's tile: as many slots as fit
// into the CU's 160 KiB, at most 5 (bf16: 32 KiB tiles -> 5 slots, 4 tiles in flight — 0.421 ms vs 0.443 ms with 4 / 3
// at 196 608 samples; the 36 KiB tile of the merged skip-layer job -> 4 slots) / 2 (fp32: 64 KiB tiles, 1 in flight).
template <int P> struct WgradCfg;
template <> struct WgradCfg<kBF16> { static constexpr int RING = 5; };
template <> struct WgradCfg<kFP32> { static constexpr int RING = 2; };
constexpr int kLdsBytes = 160 * 1024;
template <int P> SNR_HD int wgrad_ring(int per_tile) {
  const int fit = kLdsBytes / (per_tile * 1024);
  return fit < WgradCfg<P>::RING ? fit : WgradCfg<P>::RING;
}

// Everything a workgroup needs from its job, copied to registers once (the job table lives in the
// kernarg segment; indexing it inside the hot loop costs a scalar load per use).
struct WgradLocal {   // (scalar members, no arrays: a select between two array elements becomes a run-time index
                      //  and drags the whole struct into scratch memory)
  const char *a0_base, *a1_base, *b0_base, *b1_base;
  int a0_ks, a1_ks, b0_ks, b1_ks;
  float* part;
  float* bias_part;
  int a_ks, b_ks, ntb_total;
  int64_t t0, t1;    // first tile of this split, number of its tiles
  int tstep;         // distance between consecutive tiles of this split
};

// 8 waves per workgroup = two per SIMD: while one wave sits in the vector-memory issue queue (DMA)
// or at a wait, its SIMD partner keeps the matrix pipe busy.  Each wave owns one 32-row tile of the
// output (up to 160 accumulator registers) across all column tiles.
constexpr int kWgradWaves = 8;
#ifndef SNR_WGRAD_AUX
#define SNR_WGRAD_AUX 0
#endif

#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2
#define SNR_WGRAD_ISSUE(k) (void)0
#else
#define SNR_WGRAD_ISSUE(k) issue_piece(islot, std::integral_constant<int, (k)>{})
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

template <int P, int NTB, int NX, int NI, int R>
__device__ __forceinline__ void wgrad_run(const WgradLocal& L, char* smem, int wave, int lane, int ta0, int ni) {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int D = R - 1;
  const int a_ks = L.a_ks, b_ks = L.b_ks;
  const int per_tile = a_ks + b_ks;
  const int SLOT = per_tile * 1024;

  // ---- DMA pieces of this wave: wave + 8k (clamped), source pointer advances one tile per tile ----
  // The section fields are copied into opaque scalars first: a `cond ? L.x : L.y` on the struct is turned into a load
  // from a selected ADDRESS before this function is inlined, which keeps the whole struct in scratch memory afterwards
  // (and every scratch load is a vector-memory load the compiler drains the DMA queue for).
  const char *a0b = L.a0_base, *a1b = L.a1_base, *b0b = L.b0_base, *b1b = L.b1_base;
  int a0k = L.a0_ks, a1k = L.a1_ks, b0k = L.b0_ks, b1k = L.b1_ks;
  asm volatile("" : "+s"(a0b), "+s"(a1b), "+s"(b0b), "+s"(b1b), "+s"(a0k), "+s"(a1k), "+s"(b0k), "+s"(b1k));
  const char* src[NI];
  int lds_off[NI], stride[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    int p = wave + kWgradWaves * k;
    if (p >= per_tile) p = per_tile - 1;
    // which section of which side piece p belongs to (A sections first, then B sections)
    const bool isA = p < a_ks;
    int q = isA ? p : p - a_ks;
    const int ks0 = isA ? a0k : b0k;
    const bool second = q >= ks0;
    if (second) q -= ks0;
    const char* base = isA ? (second ? a1b : a0b) : (second ? b1b : b0b);
    const int ks = isA ? (second ? a1k : a0k) : (second ? b1k : b0k);
    src[k] = base + (L.t0 * ks + q) * 1024 + lane * 16;
    stride[k] = ks * 1024 * L.tstep;
    lds_off[k] = p * 1024;
  }
  int64_t src_tile = 0;      // tiles issued so far (tile k of this workgroup = t0 + k * tstep)
  // Piece indices are compile-time constants everywhere: a run-time index into src[] / lds_off[] would put the arrays
  // into scratch memory, and every scratch load is a vector-memory load the compiler waits for with vmcnt(0) — i.e. it
  // would drain the DMA queue.
  auto issue_piece = [&](int slot, auto K_) {
    constexpr int k = decltype(K_)::value;
    if constexpr (k < NI) {
      // (non-temporal aux = 2 measured no faster: 0.76 vs 0.74 ms)
      // (s_nop: no LDS read may sit in the cycle in front of an LDS-DMA — mlp_device.h, Pipe::issue_one)
      if (k < ni) asm volatile("s_nop 0");
      if (k < ni) __builtin_amdgcn_global_load_lds(src[k], SNR_LDS(smem + slot * SLOT + lds_off[k]), 16, 0, SNR_WGRAD_AUX);
    }
  };
  auto advance = [&]() {   // past the end the last tile is re-loaded: the instruction count stays uniform
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 4   // timing experiment: the stream re-reads one tile (L2 hits)
    if (false) {
#else
    if (src_tile + 1 < L.t1) {   // t1 = number of tiles of this workgroup
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
  // a 16-wide OUT section closes the A side with a single block: lanes of block 1 re-read block 0 (rows 16..31 of that
  // output tile are duplicates nobody reads)
  const int bhA = (2 * ta0 + 1 < a_ks) ? bh : 0;
  int offA[2], offB[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    offA[q] = (2 * ta0 + bhA) * 1024 + (8 * gg + ((4 * q) ^ (4 * bhA)) + r4) * 32 + c4 * 8;
    offB[q] = a_ks * 1024 + bh * 1024 + (8 * gg + ((4 * q) ^ (4 * bh)) + r4) * 32 + c4 * 8;
  }
  const int i32 = lane & 31, g32 = lane >> 5;

#if !(defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2)
  for (int d = 0; d < D; ++d) {
    static_for<0, NI>([&](auto K_) { issue_piece(d % R, K_); });
    advance();
  }
#endif
  int slot = 0, islot = D % R;
  for (int64_t tile = 0; tile < L.t1; ++tile) {
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 2   // timing experiment: compute on whatever is in LDS
    __builtin_amdgcn_s_barrier();
#else
    // this wave's pieces of `tile` have landed (DMA loads retire in order; nothing else is outstanding)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D == 1 ? 0 : NI * (D - 1)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and its reads of the previous tile are done
    __builtin_amdgcn_s_barrier();                           // same for everybody: slot of tile-1 is free
    asm volatile("" ::: "memory");
#endif
    // DMA pieces of tile + D are dripped between the MFMAs: bf16 one behind every MFMA (every second one from 8 column
    // tiles on), fp32 one behind every k-step
    constexpr int kPerHalf = NTB < 8 ? NTB : NTB / 2;
    constexpr int kDripped = NX == 0 ? 0 : (P == kBF16 ? (2 * kPerHalf < NI ? 2 * kPerHalf : NI) : (16 < NI ? 16 : NI));
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
#pragma unroll
              for (int e = 0; e < 8; ++e) bsum += (float)fa[e];
            } else {
              tr_wait<2 * (NTB - 1 - y)>(blo[y], bhi[y]);
            }
            const Frag fb = Frag{blo[y][0], blo[y][1], blo[y][2], blo[y][3], bhi[y][0], bhi[y][1], bhi[y][2], bhi[y][3]};
            acc[0][y] = M::mma(fb, fa, acc[0][y]);   // transposed tile: a lane holds 16 k-columns of ONE output row (see the stores)
            if constexpr (NTB < 8 || (y & 1)) { SNR_WGRAD_ISSUE(half * kPerHalf + (NTB < 8 ? y : y / 2)); }
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
        static_for<0, 16>([&](auto KS_) {
          constexpr int ks2 = decltype(KS_)::value;
          const int sidx = 2 * ks2 + g32;
          const float fa = elem(fa_base, a_ks, ta0, sidx);
          bsum += fa;
#pragma unroll
          for (int y = 0; y < NTB; ++y) {
            const float fb = elem(fb_base, b_ks, y, sidx);
            acc[0][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[0][y], 0, 0, 0);
          }
          SNR_WGRAD_ISSUE(ks2);
        });
      }
    }
    static_for<kDripped, NI>([&](auto K_) { SNR_WGRAD_ISSUE(decltype(K_)::value); });
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
    // The MFMAs were issued with the operands swapped, so the accumulator tile is dW^T: lane (i32, g32) holds output row
    // 32 ta0 + i32 and, in registers 4k..4k+3, its columns 32 y + 8 k + 4 g32 + 0..3 — four neighbours of the row-major
    // partial plane, i.e. one 16-byte store instead of four 4-byte ones (the 160 scalar stores per wave were a visible
    // tail of this kernel).
    // bf16 mode writes the partial sums as bf16 (each plane in the first half of its fp32-sized slot): half the bytes of
    // the split-K round trip (62 -> 31 MB written here and read back by the reduce kernel, per launch); a partial sum is
    // the fp32 accumulation over this split's samples, rounded once — the same 2^-9 the operands already carry, summed
    // in fp32 over the splits by the reduce kernel.
    const int64_t pel = (int64_t)(32 * ta0 + i32) * NB + 4 * g32;
#pragma unroll
    for (int y = 0; y < NTB; ++y) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 v = {acc[0][y][4 * k], acc[0][y][4 * k + 1], acc[0][y][4 * k + 2], acc[0][y][4 * k + 3]};
        if constexpr (P == kBF16) {
          const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *(bf16x4*)((__bf16*)L.part + pel + 32 * y + 8 * k) = h;   // (plain store: 8-byte pieces, L2 merges them into sectors)
        } else {
          __builtin_nontemporal_store(v, (f32x4*)(L.part + pel + 32 * y + 8 * k));
        }
      }
    }
    // lanes l and l^32 hold the two sample halves of the same neuron row (row = lane & 31)
    const float bs = bsum + __shfl_xor(bsum, 32, 64);
    if (lane < 32) L.bias_part[32 * ta0 + lane] = bs;
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
  L.a0_base = a.ws + J.a[0].off; L.a0_ks = J.a[0].ks;
  L.a1_base = a.ws + J.a[1].off; L.a1_ks = J.a[1].ks;
  L.b0_base = a.act + J.b[0].off; L.b0_ks = J.b[0].ks;
  L.b1_base = a.act + J.b[1].off; L.b1_ks = J.b[1].ks;
  L.a_ks = J.a_ks; L.b_ks = J.b_ks; L.ntb_total = J.ntb;
#ifndef SNR_WGRAD_STRIDED
#define SNR_WGRAD_STRIDED 1
#endif
#if SNR_WGRAD_STRIDED
  // split s takes tiles s, s + n_splits, s + 2 n_splits ...: at any moment the workgroups of a job read neighbouring
  // tiles, so each section is swept once front to back by the whole chip (the access pattern of a plain streaming read)
  // instead of by n_splits separate sequential streams
  L.t0 = split; L.tstep = J.n_splits;
  L.t1 = (a.n_tiles - split + J.n_splits - 1) / J.n_splits;
  if (L.t1 <= 0) { L.t0 = 0; L.t1 = 0; }   // more splits than tiles: the pipeline prologue still reads one (valid) tile
#else
  L.t0 = a.n_tiles * split / J.n_splits; L.tstep = 1;
  L.t1 = a.n_tiles * (split + 1) / J.n_splits - L.t0;
#endif
  const int nta = J.nta, ntb = J.ntb;
  L.part = a.part + J.part_off + (int64_t)split * nta * 32 * ntb * 32;
  L.bias_part = a.part + J.bias_part_off + (int64_t)split * nta * 32;
  const int ta0 = wave;
#if defined(SNR_WGRAD_ABLATE) && SNR_WGRAD_ABLATE == 1   // timing experiment: DMA stream only
  const int nx = 0;
#else
  const int nx = ta0 < nta ? 1 : 0;
#endif

  // shapes that occur (rows x cols in 32-tiles; KiB per tile -> DMA instructions per wave, ring slots), bf16 viewdirs:
  //   8x2  (d z0 x pe; 20 -> 3, 5)      8x8 (d z_i x h_{i-1}; 32 -> 4, 5)     8x10 (d z5 x [pe | h4]; 36 -> 5, 4)
  //   5x9  ([d z9 | d out] x [h7 | dir]; 27 -> 4, 5)          1x4 (d out x h9; 9 -> 2, 5)
  //   without viewdirs: 1x8 (d out x h7; 17 -> 3, 5)
  const int per_tile = J.a_ks + J.b_ks;
  const int ni = (per_tile + kWgradWaves - 1) / kWgradWaves;
#define SNR_RUN(NTB_, NX_, NI_, R_) wgrad_run<P, NTB_, NX_, NI_, R_>(L, smem, wave, lane, ta0, ni)
  if constexpr (P == kFP32) {   // one tile in flight: the wait immediate is 0 whatever ni is
    if (nx == 0) SNR_RUN(1, 0, 8, 2);
    else if (ntb == 9) SNR_RUN(9, 1, 8, 2);
    else if (ntb == 8) SNR_RUN(8, 1, 8, 2);
    else if (ntb == 4) SNR_RUN(4, 1, 8, 2);
    else if (ntb == 2) SNR_RUN(2, 1, 8, 2);
    else SNR_RUN(1, 1, 8, 2);
  } else if (ni == 5) {
    if (nx == 0) SNR_RUN(1, 0, 5, 4); else SNR_RUN(10, 1, 5, 4);
  } else if (ni == 4) {
    if (nx == 0) SNR_RUN(1, 0, 4, 5);
    else if (ntb == 9) SNR_RUN(9, 1, 4, 5);
    else SNR_RUN(8, 1, 4, 5);
  } else if (ni == 3) {
    if (nx == 0) SNR_RUN(1, 0, 3, 5);
    else if (ntb == 8) SNR_RUN(8, 1, 3, 5);
    else SNR_RUN(2, 1, 3, 5);
  } else if (ni == 2) {
    if (nx == 0) SNR_RUN(1, 0, 2, 5);
    else if (ntb == 4) SNR_RUN(4, 1, 2, 5);
    else SNR_RUN(1, 1, 2, 5);
  } else {   // one DMA instruction per wave per tile: the small hash-grid jobs (1x1, 2x1, 1x2, 2x2 tiles of 5..8 KiB).
             // NI must equal the instructions really issued — the counted wait's immediate is NI * (tiles in flight)
    if (nx == 0) SNR_RUN(1, 0, 1, 5);
    else if (ntb == 2) SNR_RUN(2, 1, 1, 5);
    else SNR_RUN(1, 1, 1, 5);
  }
#undef SNR_RUN
}

// ------------------------------------------------------------------------------------------
// Round 4: the same split-K job as a FOUR-wave workgroup with up to 512 registers per lane — the shape of the layer-pair
// kernel (mlp_wgrad_pair.h), in whose launch the plain jobs now run on a few dedicated CUs: these jobs are bound by their
// DMA stream (a CU pulls ~33 GB/s through LDS-DMA whatever else the chip does: tests/probes/r04_cu_split.sh), the pair
// workgroups by the matrix pipe, so ~40 CUs stream the plain jobs' 0.46 GB while the other ~216 compute — instead of
// 256 CUs doing first one, then the other.
// With one wave per SIMD nothing hides a wave's stalls, so the waves specialise (first cut: all four issued DMA and
// computed, 1.85x the CU time of the 8-wave kernel — the DMA issue stalls and the LDS latency sat on the MFMA waves'
// critical path, and 5 x 9 tiles do not divide by four):
//   wave 3      LOADER: issues every DMA piece of a tile, waits for the oldest tile in flight, joins the barrier
//   waves 0..2  CONSUMERS: a third of the job's output tiles each — by rows (rows_mode: d z5 x pe, 8 x 2 tiles -> 3 + 3 + 2
//               row tiles) or by columns (5 x 9 -> 3 column tiles x 5 rows each) —; all transposing reads of a tile are
//               issued at once (both k-steps), then the MFMAs run back to back behind counted waits
// One barrier per tile: behind it tile t has landed (the loader waited) and everybody is done with tile t - 1, whose slot
// the loader refills.  Reads, MFMA operand order, partial-plane layout: exactly wgrad_run's.
// ------------------------------------------------------------------------------------------
constexpr int kPlainWaves = 4, kPlainConsumers = 3;
#ifndef SNR_PLAIN_AUX
#define SNR_PLAIN_AUX 0   // cache policy of the plain jobs' DMA stream (A/B builds: 2 = non-temporal)
#endif
SNR_HD int plain_third(int w, int n) { return (w * n + 2) / 3; }   // first tile of consumer w of n tiles: sizes differ by <= 1

// Y = tiles in flight behind the one being waited for (ring = Y + 2 slots).  Every wave issues DMA pieces — a wave's counted
// wait needs an immediate and vmcnt has 6 bits, so ONE wave cannot keep more than 63 KiB in flight, too little to cover the
// HBM latency at 33 GB/s —: the loader the first PL pieces of a tile, consumer w the pieces PL + w + 3 k, k < PC (clamped to
// the tile's last piece: every wave issues a fixed count), each at the top of its tile-loop body, where nothing of its own is
// waiting on the issue queue.  Y * PL and Y * PC <= 63.
template <int NR, int NC, int Y, int PL, int PC>
__device__ __forceinline__ void plain_run4(const WgradLocal& L, char* smem, int wave, int lane, int cols_mode, int nta, int ntb) {
  using M = Mma<kBF16>;
  using Frag = typename M::Frag;
  constexpr int R = Y + 2;
  static_assert(Y * PL <= 63 && Y * PC <= 63, "the counted waits need immediates below 64");
  const int a_ks = L.a_ks, b_ks = L.b_ks;
  const int per_tile = a_ks + b_ks;
  const int SLOT = per_tile * 1024;

  // where piece p of a tile comes from: (A sections first, then B sections)
  const char *a0b = L.a0_base, *a1b = L.a1_base, *b0b = L.b0_base, *b1b = L.b1_base;
  int a0k = L.a0_ks, a1k = L.a1_ks, b0k = L.b0_ks, b1k = L.b1_ks;
  asm volatile("" : "+s"(a0b), "+s"(a1b), "+s"(b0b), "+s"(b1b), "+s"(a0k), "+s"(a1k), "+s"(b0k), "+s"(b1k));
  // (computed in place, element by element with compile-time indices, like wgrad_run: handing an array element to a helper
  //  by reference leaves the arrays in scratch memory, and every scratch load is a vector-memory load that the compiler
  //  waits for with vmcnt(0) — it would drain the DMA queue)
#define SNR_PIECE_SRC(P_EXPR, SRC, STRIDE)                                          \
  do {                                                                              \
    const int p_ = (P_EXPR);                                                        \
    const bool isA_ = p_ < a_ks;                                                    \
    int q_ = isA_ ? p_ : p_ - a_ks;                                                 \
    const int ks0_ = isA_ ? a0k : b0k;                                              \
    const bool second_ = q_ >= ks0_;                                                \
    if (second_) q_ -= ks0_;                                                        \
    const char* base_ = isA_ ? (second_ ? a1b : a0b) : (second_ ? b1b : b0b);       \
    const int ks_ = isA_ ? (second_ ? a1k : a0k) : (second_ ? b1k : b0k);           \
    (SRC) = base_ + (L.t0 * ks_ + q_) * 1024 + lane * 16;                           \
    (STRIDE) = ks_ * 1024 * L.tstep;                                                \
  } while (0)

  if (wave == kPlainConsumers) {
    // ---------------- loader: pieces 0 .. PL-1 of every tile ----------------
    const char* src[PL];
    int stride[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) SNR_PIECE_SRC(k < per_tile ? k : per_tile - 1, src[k], stride[k]);
    int64_t issued = 0;
    auto issue_tile = [&](int slot) {
      static_for<0, PL>([&](auto K_) {
        constexpr int k = decltype(K_)::value;
        __builtin_amdgcn_global_load_lds(src[k], SNR_LDS(smem + slot * SLOT + (k < per_tile ? k : per_tile - 1) * 1024), 16, 0, SNR_PLAIN_AUX);
      });
      if (issued + 1 < L.t1) {   // past the end the last tile is loaded again: uniform counts
        ++issued;
#pragma unroll
        for (int k = 0; k < PL; ++k) src[k] += stride[k];
      }
    };
    for (int d = 0; d <= Y; ++d) issue_tile(d % R);
    int islot = (Y + 1) % R;
    for (int64_t tile = 0; tile < L.t1; ++tile) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * PL) : "memory");   // this wave's pieces of `tile` have landed (loads retire in order)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_tile(islot);
      islot = islot + 1 == R ? 0 : islot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing re-loads
    return;
  }

  // ---------------- consumers: pieces PL + wave + 3 k ----------------
  const char* src[PC];
  int stride[PC], lds_off[PC];
#pragma unroll
  for (int k = 0; k < PC; ++k) {
    int p = PL + wave + kPlainConsumers * k;
    if (p >= per_tile) p = per_tile - 1;
    SNR_PIECE_SRC(p, src[k], stride[k]);
    lds_off[k] = p * 1024;
  }
  int64_t issued = 0;
  auto issue_mine = [&](int slot) {
    static_for<0, PC>([&](auto K_) {
      constexpr int k = decltype(K_)::value;
      asm volatile("s_nop 0");   // no LDS read in the cycle in front of an LDS-DMA (mlp_device.h, Pipe::issue_one)
      __builtin_amdgcn_global_load_lds(src[k], SNR_LDS(smem + slot * SLOT + lds_off[k]), 16, 0, SNR_PLAIN_AUX);
    });
    if (issued + 1 < L.t1) {
      ++issued;
#pragma unroll
      for (int k = 0; k < PC; ++k) src[k] += stride[k];
    }
  };
  for (int d = 0; d <= Y; ++d) issue_mine(d % R);
  int islot = (Y + 1) % R;
#undef SNR_PIECE_SRC

  // ---------------- consumers ----------------
  // this wave's tiles (wave-uniform): row tiles r0 .. r0 + nr, column tiles c0 .. c0 + nc
  const int r0 = cols_mode ? 0 : plain_third(wave, nta), nr = (cols_mode ? nta : plain_third(wave + 1, nta)) - r0;
  const int c0 = cols_mode ? plain_third(wave, ntb) : 0, nc = (cols_mode ? plain_third(wave + 1, ntb) : ntb) - c0;
  const bool do_bias = !cols_mode || wave == 0;   // by columns every consumer holds every row: wave 0 sums them

  f32x16 acc[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) bsum[i] = 0.f;

  // tile-invariant LDS offsets of the transposing reads (wgrad_run); rows / columns this wave does not have re-read its first
  const int G = lane >> 4, ip = lane & 15, gg = G >> 1, bh = G & 1, c4 = ip & 3, r4 = ip >> 2;
  int offA[NR][2], offB[NC][2];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int rt = i < nr ? r0 + i : r0;
    const int bhA = (2 * rt + 1 < a_ks) ? bh : 0;   // a 16-wide OUT section closes the A side with a single block
#pragma unroll
    for (int q = 0; q < 2; ++q) offA[i][q] = (2 * rt + bhA) * 1024 + (8 * gg + ((4 * q) ^ (4 * bhA)) + r4) * 32 + c4 * 8;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int ct = j < nc ? c0 + j : c0;
#pragma unroll
    for (int q = 0; q < 2; ++q) offB[j][q] = a_ks * 1024 + (2 * ct + bh) * 1024 + (8 * gg + ((4 * q) ^ (4 * bh)) + r4) * 32 + c4 * 8;
  }
  const int i32 = lane & 31, g32 = lane >> 5;

  // Software pipeline over the k-steps (16 samples each, two per tile): while the MFMAs of one k-step run out of one operand
  // register set, the transposing reads of the NEXT k-step go to the other set, one or two behind every MFMA (an LDS read or
  // a DMA piece issued behind an MFMA runs in its shadow; issued in a block in front of the MFMAs — the first cut — they were
  // the workgroup's critical path: 18.8 CU-ms for the step's plain jobs against 14 of the 8-wave kernel).  A tile's body:
  //   MFMAs (k, step 0) from set 0 | reads (k, step 1) -> set 1 ; barrier k + 1 ; MFMAs (k, step 1) from set 1 | reads (k + 1, step 0) -> set 0
  // At the barrier this wave has READ all of tile k (its slot is refilled behind the barrier) and its own pieces of tile
  // k + 1 have landed; behind it tile k + 1 is complete for everybody.
  constexpr int NRDH = 2 * (NR + NC), NMMH = NR * NC;   // reads, MFMAs per k-step
  constexpr int NLEAD = NMMH > 6 ? NMMH - 4 : (NMMH > 1 ? NMMH / 2 : 1);   // the reads go out behind the first NLEAD MFMAs
  bf16x4 alo[2][NR], ahi[2][NR], blo[2][NC], bhi[2][NC];   // [register set = k-step][tile]
  int slot = 0;
  auto issue_read = [&](auto H_, auto R_, uint32_t sb32) {
    constexpr int half = decltype(H_)::value, r = decltype(R_)::value;
    if constexpr (r < 2 * NR) {
      constexpr int i = r / 2, q = r % 2;
      if constexpr (q == 0) tr_read<half * 512>(alo[half][i], sb32 + offA[i][0]);
      else tr_read<half * 512>(ahi[half][i], sb32 + offA[i][1]);
    } else {
      constexpr int jj = (r - 2 * NR) / 2, q = r % 2;
      if constexpr (q == 0) tr_read<half * 512>(blo[half][jj], sb32 + offB[jj][0]);
      else tr_read<half * 512>(bhi[half][jj], sb32 + offB[jj][1]);
    }
  };
  // No asm read stays in flight past the phase that issued it (for the compiler such a read is complete when issued: a copy
  // it places at the loop's back edge, or a live-range split, would meet the old register content — mlp_wgrad_pair.h): the
  // reads go out in front of the phase's last MFMAs and have landed by its end; the ties keep their registers allocated up to there
  auto wait_set = [&](auto H_) {
    constexpr int half = decltype(H_)::value;
    static_for<0, NR>([&](auto I_) { constexpr int i = decltype(I_)::value; tr_wait<0>(alo[half][i], ahi[half][i]); });
    static_for<0, NC>([&](auto J_) { constexpr int jj = decltype(J_)::value; tr_wait<0>(blo[half][jj], bhi[half][jj]); });
  };
  auto enter_tile = [&]() {   // wait for this wave's pieces of the next tile, barrier, refill the slot the barrier freed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y * PC) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_mine(islot);
    islot = islot + 1 == R ? 0 : islot + 1;
  };
  // MFMAs of k-step `half` (its operand set is complete), the reads of the other set from the tile at sb_rd behind them
  auto phase = [&](auto H_, uint32_t sb_rd) {
    constexpr int half = decltype(H_)::value;
    static_for<0, NMMH>([&](auto M_) {
      constexpr int m = decltype(M_)::value;
      constexpr int jj = m / NR, i = m % NR;
      // (no run-time guards: a tile this wave does not own is a duplicate of its first — computed and never stored —
      //  so that the phase stays one basic block)
      const Frag fa = Frag{alo[half][i][0], alo[half][i][1], alo[half][i][2], alo[half][i][3],
                           ahi[half][i][0], ahi[half][i][1], ahi[half][i][2], ahi[half][i][3]};
      const Frag fb = Frag{blo[half][jj][0], blo[half][jj][1], blo[half][jj][2], blo[half][jj][3],
                           bhi[half][jj][0], bhi[half][jj][1], bhi[half][jj][2], bhi[half][jj][3]};
      acc[i][jj] = M::mma(fb, fa, acc[i][jj]);
      if constexpr (jj == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[e];
      }
      if constexpr (m < NLEAD) {
        // (unconditional, also behind the last tile, where they fetch a stale slot nobody uses: a branch here would put a
        //  join — and the compiler's copies of the destination registers — between the reads and their wait)
        static_for<(m * NRDH + NLEAD - 1) / NLEAD, ((m + 1) * NRDH + NLEAD - 1) / NLEAD>([&](auto R_) {
          issue_read(std::integral_constant<int, 1 - half>{}, R_, sb_rd);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_set(std::integral_constant<int, 1 - half>{});
  };
  // prologue: tile 0 lands, the reads of its first k-step go to set 0
  enter_tile();
  static_for<0, NRDH>([&](auto R_) { issue_read(std::integral_constant<int, 0>{}, R_, lds_addr(smem)); });
  wait_set(std::integral_constant<int, 0>{});
  for (int64_t tile = 0; tile < L.t1; ++tile) {
    const bool more = tile + 1 < L.t1;
    const uint32_t sb_cur = lds_addr(smem) + slot * SLOT;
    phase(std::integral_constant<int, 0>{}, sb_cur);
    if (more) enter_tile();
    slot = slot + 1 == R ? 0 : slot + 1;
    phase(std::integral_constant<int, 1>{}, lds_addr(smem) + slot * SLOT);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing re-loads

  // partial plane [nta * 32][ntb * 32] of this split, bf16 in the first half of its fp32-sized slot (wgrad_run)
  const int NB = L.ntb_total * 32;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (i >= nr) continue;
    const int rt = r0 + i;
    const int64_t pel = (int64_t)(32 * rt + i32) * NB + 4 * g32;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (j >= nc) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bf16x4 h = {(__bf16)acc[i][j][4 * k], (__bf16)acc[i][j][4 * k + 1], (__bf16)acc[i][j][4 * k + 2],
                          (__bf16)acc[i][j][4 * k + 3]};
        *(bf16x4*)((__bf16*)L.part + pel + 32 * (c0 + j) + 8 * k) = h;
      }
    }
    if (do_bias) {
      const float bs = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      if (lane < 32) L.bias_part[32 * rt + lane] = bs;
    }
  }
}

// A plain job inside the layer-pair kernel's launch: the job, its network's buffers, and how its tiles go to waves.
struct PlainJob {
  WgradJob j;               // split_begin counts plain workgroups (0 = the launch's first plain workgroup)
  const char* act;          // the network's forward workspace
  const char* ws;           // ... and backward workspace (d z sections)
  float* part;              // ... and partial-sum buffer
  int64_t n_tiles;
  int cols_mode;            // plain_run4
};

__device__ __forceinline__ void plain_job_run(const PlainJob& Q, int split, char* smem, int wave, int lane) {
  const WgradJob& J = Q.j;
  WgradLocal L;
  L.a0_base = Q.ws + J.a[0].off; L.a0_ks = J.a[0].ks;
  L.a1_base = Q.ws + J.a[1].off; L.a1_ks = J.a[1].ks;
  L.b0_base = Q.act + J.b[0].off; L.b0_ks = J.b[0].ks;
  L.b1_base = Q.act + J.b[1].off; L.b1_ks = J.b[1].ks;
  L.a_ks = J.a_ks; L.b_ks = J.b_ks; L.ntb_total = J.ntb;
  L.t0 = split; L.tstep = J.n_splits;
  L.t1 = (Q.n_tiles - split + J.n_splits - 1) / J.n_splits;
  if (L.t1 <= 0) { L.t0 = 0; L.t1 = 0; }
  const int nta = J.nta, ntb = J.ntb;
  L.part = Q.part + J.part_off + (int64_t)split * nta * 32 * ntb * 32;
  L.bias_part = Q.part + J.bias_part_off + (int64_t)split * nta * 32;
  const int per_tile = J.a_ks + J.b_ks;
  // shapes that occur (bf16, recompute mode; rows x cols in 32-tiles; KiB = DMA pieces per tile):
  //   8x2  d z5 x pe (20), by rows                 5x9  [d z9 | d out] x [h7 | dir] (27), by columns
  //   1x4  d out x h9 (9), by columns              1x8  d out x h7 (17; no view directions), by columns
  // consumers hold at most NR x NC tiles; Y tiles in flight with (Y + 2) tiles inside the 160 KiB of LDS; PL pieces of a tile
  // issued by the loader, PC by each consumer (PL + 3 PC >= the tile's pieces)
  if (!Q.cols_mode) {
    plain_run4<3, 2, 4, 14, 2>(L, smem, wave, lane, 0, nta, ntb);      // 20 KiB tiles, ring of 6
  } else if (per_tile == 27) {
    plain_run4<5, 3, 3, 15, 4>(L, smem, wave, lane, 1, nta, ntb);      // ring of 5
  } else if (per_tile == 9) {
    plain_run4<1, 2, 7, 6, 1>(L, smem, wave, lane, 1, nta, ntb);       // ring of 9
  } else {
    plain_run4<1, 3, 4, 11, 2>(L, smem, wave, lane, 1, nta, ntb);      // 17 KiB tiles, ring of 6
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
  if (kind == SRC_NAT) return x;
  if (kind == SRC_HG_INC) return q == 0 ? 8 * g + e : hg_inc_col(g, e);   // (bf16 only)
  return (P == kBF16) ? 8 * g + e : 2 * e + g;  // SRC_OUT (single frag): raw channel
}

// One launch reduces the outputs of every job table of a backward pass (the plain pass's and the layer-pair kernel's): each
// output carries the fields of its job it needs.
// (round 4) one launch also covers the outputs of SEVERAL networks' backward passes (coarse + fine): an output names its
// network, whose partial-sum buffer, G block and gradient buffer the launch holds.
struct ReduceOut {
  WgradOut o;
  int nta, ntb, n_splits;
  int net;          // index into ReduceArgs::nets
  int64_t part_off, bias_part_off;
};
constexpr int kMaxReduceOuts = 32, kMaxReduceNets = 2;
struct ReduceNet {
  const float* part;
  float* post;      // G block or null
  float* grad;      // flat parameter gradient
  int accumulate;
};
struct ReduceArgs {
  int n_outs;
  ReduceOut out[kMaxReduceOuts];
  ReduceNet nets[kMaxReduceNets];
};
// false: the table is full (the caller returns an error: a dropped output would be a parameter gradient nobody writes)
inline bool append_reduce(ReduceArgs& r, const WgradArgs& w, int net, float* grad, int accumulate) {
  for (int i = 0; i < w.n_outs; ++i) {
    if (r.n_outs >= kMaxReduceOuts) return false;
    ReduceOut& R = r.out[r.n_outs++];
    const WgradJob& J = w.job[w.out[i].job];
    R.o = w.out[i];
    R.nta = J.nta; R.ntb = J.ntb; R.n_splits = J.n_splits; R.part_off = J.part_off; R.bias_part_off = J.bias_part_off;
    R.net = net;
  }
  r.nets[net] = ReduceNet{w.part, w.post, grad, accumulate};
  return true;
}

template <int P>
__global__ void mlp_wgrad_reduce_kernel(ReduceArgs a) {
  // Every parameter element is produced by exactly one (output, row, column): without `accumulate` the result is
  // stored, not added, and the gradient buffer needs no clearing first
  // (tests/test_gpu_kernels.py: test_mlp_backward_overwrites_every_element).
  // blockIdx.y = output.  A 16-lane group covers 16 consecutive work items (row, 4 consecutive partial-sum columns — one
  // extra item per row is the bias); the wave's four groups each sum every fourth split plane and are combined with two
  // cross-lane adds, which quadruples the loads in flight of this short, latency-bound kernel.  The summation order is
  // fixed: results are deterministic.
  const ReduceOut& J = a.out[blockIdx.y];
  const WgradOut& O = J.o;
  const ReduceNet& N = a.nets[J.net];
  float* __restrict__ grad = N.grad;
  const int accumulate = N.accumulate;
  const int NA = J.nta * 32, NB = J.ntb * 32, NQ = O.cols / 4;
  const int lane = threadIdx.x & 63, r = lane >> 4;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t idx = wave * 16 + (lane & 15);
  const int64_t items = (int64_t)O.rows * (NQ + 1);
  if (wave * 16 >= items) return;                       // wave-uniform: the cross-lane adds below see whole waves
  const bool live = idx < items;
  const int ra = live ? (int)(idx / (NQ + 1)) : 0, q = live ? (int)(idx % (NQ + 1)) : 0;
  const int n = slot_true_index<P>(O.a_kind, ra, 0) - (O.a_kind == SRC_OUT ? O.row_off : 0);
  const bool row_ok = live && n >= 0 && n < O.rows_valid;
  float* dst = O.to_scratch ? N.post : grad;
  const bool acc = accumulate && !O.to_scratch;
  const bool is_bias = q == NQ;
  const int cb = 4 * q;
  const int64_t st = (int64_t)NA * NB;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (row_ok && !is_bias) {
    // plane sp of this job starts at float offset part_off + sp * st; bf16 mode keeps bf16 elements in its first half
    const int64_t el = (int64_t)(O.row0 + ra) * NB + O.col0 + cb;
    auto plane = [&](int sp) -> f32x4 {
      const float* base = N.part + J.part_off + (int64_t)sp * st;
      if constexpr (P == kBF16) {
        const bf16x4 h = *(const bf16x4*)((const __bf16*)base + el);   // (plain load: the partials were just written; A/B −6 %)
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        return __builtin_nontemporal_load((const f32x4*)(base + el));
      }
    };
    int sp = r;
    for (; sp + 12 < J.n_splits; sp += 16) {
      const f32x4 v0 = plane(sp), v1 = plane(sp + 4), v2 = plane(sp + 8), v3 = plane(sp + 12);
      s0 += v0 + v2; s1 += v1 + v3;
    }
    for (; sp < J.n_splits; sp += 4) s0 += plane(sp);
  } else if (row_ok && O.bias_off >= 0) {
    for (int sp = r; sp < J.n_splits; sp += 4) s0[0] += N.part[J.bias_part_off + (int64_t)sp * NA + O.row0 + ra];
  }
  f32x4 s = s0 + s1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s[e] += __shfl_xor(s[e], 16);
    s[e] += __shfl_xor(s[e], 32);
  }
  if (!row_ok || r != 0) return;
  if (is_bias) {
    if (O.bias_off >= 0) dst[O.bias_off + n] = acc ? dst[O.bias_off + n] + s[0] : s[0];
    return;
  }
  float* grow = dst + O.w_off + (int64_t)n * O.ld + O.col_off;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = slot_true_index<P>(O.b_kind, cb + e, O.L);
    if (k < 0 || k >= O.cols_valid) continue;
    grow[k] = acc ? grow[k] + s[e] : s[e];
  }
}

// ------------------------------------------------------------------------------------------
// the two dense products behind G (header comment): exact fp32 MFMA, one workgroup of 4 waves per 32 x 32 output
// tile, each wave a quarter of the contraction, summed through LDS.
//   tiles  0..71  : dW_feat[j][k]        = sum_m Wv[m][j] G[m][k]                  (256 x 256, contraction 128), with s9
//                   as a 257th column of G: db_feat[j] = sum_m Wv[m][j] s9[m]
//   tiles 72..103 : dW_views[m][j < 256] = sum_k G[m][k] Wf[j][k] + s9[m] b_f[j]   (128 x 256, contraction 256)
// P == bf16 rounds the two weight matrices to bf16 first: the forward and dgrad kernels evaluated the layers with the
// rounded weights, and the products restate exactly those layers.
// ------------------------------------------------------------------------------------------
struct PostNet {
  const float* params;
  const float* post;   // G [128][256], s9 [128]
  float* grad;
  int w_views, ld_views, w_feat, b_feat;   // float offsets in params / grad (ld_views = 256 + in_dir)
  int accumulate;
};
constexpr int kPostTiles = 104;            // workgroups per network
struct PostArgs { PostNet net[kMaxReduceNets]; };   // blockIdx.x / kPostTiles = network

template <int P> __device__ __forceinline__ float wround(float x) {
  return P == kBF16 ? (float)(__bf16)x : x;
}

template <int P>
__global__ __launch_bounds__(256) void wgrad_post_kernel(PostArgs pa) {
  const PostNet& a = pa.net[blockIdx.x / kPostTiles];
  __shared__ float red[3][64][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const float* G = a.post;
  const float* s9 = a.post + kPostG;
  const float* Wv = a.params + a.w_views;
  const float* Wf = a.params + a.w_feat;
  const int t = blockIdx.x % kPostTiles;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int r0, c0;   // output tile origin
  if (t < 72) {
    // C[j][k] = sum_m A[j][m] B[m][k], A[j][m] = Wv[m][j0 + j], B[m][k] = G[m][k0 + k]; m = 32 * wave + 2 * s + h.
    // G is extended by one column tile whose first column is s9: C[j][256] = db_feat[j].
    r0 = 32 * (t / 9); c0 = 32 * (t % 9);
    const int m0 = 32 * wave;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = m0 + 2 * s + h;
      const float av = wround<P>(Wv[(int64_t)m * a.ld_views + r0 + i]);
      const float bv = c0 < kW ? G[m * kW + c0 + i] : (i == 0 ? s9[m] : 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  } else {
    // C[m][j] = sum_k A[m][k] B[k][j], A[m][k] = G[m0 + m][k], B[k][j] = Wf[j0 + j][k]; both contiguous in k:
    // a lane loads 4 consecutive k (16 B) and spends them on 4 MFMA steps — step e of a group uses k = base + 4 h + e
    // on both operands
    const int tt = t - 72;
    r0 = 32 * (tt >> 3); c0 = 32 * (tt & 7);
    const int k0 = 64 * wave;
    const float* ga = G + (int64_t)(r0 + i) * kW + k0 + 4 * h;
    const float* wb = Wf + (int64_t)(c0 + i) * kW + k0 + 4 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 av = *(const f32x4*)(ga + 8 * s);
      const f32x4 bv = *(const f32x4*)(wb + 8 * s);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wround<P>(bv[e]), acc, 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][lane][r] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r] + red[0][lane][r] + red[1][lane][r] + red[2][lane][r];
      const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h, col = c0 + i;
      float* d;
      float val = v;
      if (t < 72) {
        if (col >= kW) {
          if (col > kW) continue;
          d = a.grad + a.b_feat + row;
        } else {
          d = a.grad + a.w_feat + (int64_t)row * kW + col;
        }
      } else {
        val += s9[row] * a.params[a.b_feat + col];
        d = a.grad + a.w_views + (int64_t)row * a.ld_views + col;
      }
      *d = a.accumulate ? *d + val : val;
    }
  }
}

// split-K kernel + reduce of a prepared bf16 job list (defined in mlp_bwd.hip; also serves hashgrid.hip)
int wgrad_launch_bf16(const WgradArgs& w, int total_splits, float* grad, int accumulate, hipStream_t s);

// ------------------------------------------------------------------------------------------
// host: workgroups of a launch apportioned to jobs
// ------------------------------------------------------------------------------------------
// out[i] workgroups for job i of weight w[i], `total` in all (at least 1 and at most cap[i] each: a job cannot use more splits
// than it has tiles), chosen to minimise the LARGEST w[i] / out[i] — the launch ends with its slowest job, and with a few
// dozen workgroups over six jobs a proportional share rounded down leaves the smallest job 60 % over the mean (one
// workgroup where 1.6 were due: found as the whole launch waiting for the coarse network's d out x h9 job).  Greedy: every
// job starts with one workgroup, each further one goes to the job with the largest per-workgroup load.
inline void apportion(const int64_t* w, const int64_t* cap, int n, int total, int* out) {
  int assigned = 0;
  for (int i = 0; i < n; ++i) { out[i] = 1; ++assigned; }
  while (assigned < total) {
    int best = -1;
    for (int i = 0; i < n; ++i) {
      if (out[i] >= cap[i]) continue;
      // w[i] / out[i] > w[best] / out[best], in integers
      if (best < 0 || w[i] * out[best] > w[best] * out[i]) best = i;
    }
    if (best < 0) break;
    ++out[best]; ++assigned;
  }
}
// a job's place in the launch (its first workgroup) and in the partial-sum buffer
inline void place_job(WgradJob& J, int splits, int& split_begin, int64_t& part_off) {
  J.n_splits = splits; J.split_begin = split_begin; split_begin += splits;
  J.part_off = part_off; part_off += (int64_t)splits * J.nta * 32 * J.ntb * 32;
  J.bias_part_off = part_off; part_off += (int64_t)splits * J.nta * 32;
}

// ------------------------------------------------------------------------------------------
// host: job list
// ------------------------------------------------------------------------------------------
// recompute = the trunk layers are handled by the layer-pair kernel (mlp_wgrad_pair.h): what stays here is the skip layer's
// encoding columns and the layers above the trunk
template <int P>
static WgradArgs make_jobs(const snr_mlp_config* c, int64_t n_samples, int64_t* part_floats, int* total_splits,
                           bool recompute = false) {
  using B = Blob<P>;
  constexpr int SPF = 2 * Prec<P>::EPF;   // k-slots per frag
  const int vd = c->use_viewdirs;
  const ParamLayout L = make_param_layout(c->multires, c->multires_views, vd, c->out_ch, c->i_embed == -1);
  const ActLayout<P> AL(n_samples, vd);
  const WsLayout<P> WL(n_samples, vd);
  WgradArgs A{};
  A.n_tiles = AL.n_tiles;
  const int L_pts = c->i_embed == -1 ? 0 : c->multires, L_dir = c->i_embed == -1 ? 0 : c->multires_views;
  int n = 0, no = 0;
  const WgradSec none{0, 0};
  auto job = [&](WgradSec a0, WgradSec a1, WgradSec b0, WgradSec b1) {
    WgradJob& J = A.job[n];
    J.a[0] = a0; J.a[1] = a1; J.b[0] = b0; J.b[1] = b1;
    J.a_ks = a0.ks + a1.ks; J.b_ks = b0.ks + b1.ks;
    J.nta = (J.a_ks * SPF + 31) / 32; J.ntb = (J.b_ks * SPF + 31) / 32;
    return n++;
  };
  auto out = [&](int j, int row0, int rows, int a_kind, int col0, int cols, int b_kind, int Lenc, int64_t w_off, int ld,
                 int col_off, int row_off, int rows_valid, int cols_valid, int64_t bias_off, int to_scratch = 0) {
    WgradOut& O = A.out[no++];
    O.job = j; O.row0 = row0; O.rows = rows; O.a_kind = a_kind; O.col0 = col0; O.cols = cols; O.b_kind = b_kind;
    O.L = Lenc; O.w_off = (int)w_off; O.ld = ld; O.col_off = col_off; O.row_off = row_off; O.rows_valid = rows_valid;
    O.cols_valid = cols_valid; O.bias_off = (int)bias_off; O.to_scratch = to_scratch;
  };
  const int ip = L.in_pts;
  const int SH = B::KS_H * SPF, SPE = B::KS_PE * SPF, SDIR = B::KS_DIR * SPF, SH9 = B::KS_H9 * SPF;   // slots per section
  const WgradSec pe{AL.off_pe(), B::KS_PE};
  auto dz = [&](int i) { return WgradSec{WL.off_dz(i), B::KS_H}; };
  auto h = [&](int i) { return WgradSec{AL.off_h(i), B::KS_H}; };
  // merged skip-layer job: its 36 KiB bf16 tile still leaves a 4-slot ring; fp32 (72 KiB tiles, 9 DMA instructions per
  // wave) keeps the two jobs apart
  const bool merge5 = P == kBF16;
  if (recompute) {
    // d z5 x pe: the encoding columns of the skip layer (its bias gradient comes with the pair kernel's row sums of d z5)
    const int j = job(dz(kSkip + 1), none, pe, none);
    out(j, 0, SH, SRC_H, 0, SPE, SRC_ENC_PTS, L_pts, L.w_pts[kSkip + 1], kW + ip, 0, 0, kW, ip, -1);
  } else {
    const int j = job(dz(0), none, pe, none);
    out(j, 0, SH, SRC_H, 0, SPE, SRC_ENC_PTS, L_pts, L.w_pts[0], ip, 0, 0, kW, ip, L.b_pts[0]);
  }
  for (int i = 1; i < 8 && !recompute; ++i) {
    if (i == kSkip + 1) {
      if (merge5) {
        const int j = job(dz(i), none, pe, h(i - 1));
        out(j, 0, SH, SRC_H, 0, SPE, SRC_ENC_PTS, L_pts, L.w_pts[i], kW + ip, 0, 0, kW, ip, L.b_pts[i]);
        out(j, 0, SH, SRC_H, SPE, SH, SRC_H, 0, L.w_pts[i], kW + ip, ip, 0, kW, kW, -1);
      } else {
        int j = job(dz(i), none, pe, none);
        out(j, 0, SH, SRC_H, 0, SPE, SRC_ENC_PTS, L_pts, L.w_pts[i], kW + ip, 0, 0, kW, ip, L.b_pts[i]);
        j = job(dz(i), none, h(i - 1), none);
        out(j, 0, SH, SRC_H, 0, SH, SRC_H, 0, L.w_pts[i], kW + ip, ip, 0, kW, kW, -1);
      }
    } else {
      const int j = job(dz(i), none, h(i - 1), none);
      out(j, 0, SH, SRC_H, 0, SH, SRC_H, 0, L.w_pts[i], kW, 0, 0, kW, kW, L.b_pts[i]);
    }
  }
  const WgradSec dout{WL.off_dout(), 1};
  if (vd) {
    // [d z9 | d out] x [h7 | dir]: G (+ s9) to the scratch block, db_views, dW_views[:, 256:], dW_alpha (+ db_alpha);
    // the d out x dir corner is computed and ignored
    const WgradSec dz9{WL.off_dz9(), B::KS_H9}, dir{AL.off_dir(), B::KS_DIR}, h9{AL.off_h9(), B::KS_H9};
    int j = job(dz9, dout, h(7), dir);
    out(j, 0, SH9, SRC_H, 0, SH, SRC_H, 0, 0, kW, 0, 0, kW / 2, kW, kPostG, 1);
    out(j, 0, SH9, SRC_H, 0, 0, SRC_H, 0, 0, 0, 0, 0, kW / 2, 0, L.b_views);
    out(j, 0, SH9, SRC_H, SH, SDIR, SRC_ENC_DIR, L_dir, L.w_views, kW + L.in_dir, kW, 0, kW / 2, L.in_dir, -1);
    out(j, SH9, SPF, SRC_OUT, 0, SH, SRC_H, 0, L.w_alpha, kW, 0, 3, 1, kW, L.b_alpha);
    j = job(dout, none, h9, none);
    out(j, 0, SPF, SRC_OUT, 0, SH9, SRC_H, 0, L.w_rgb, kW / 2, 0, 0, 3, kW / 2, L.b_rgb);
  } else {
    const int j = job(dout, none, h(7), none);
    out(j, 0, SPF, SRC_OUT, 0, SH, SRC_H, 0, L.w_out, kW, 0, 0, c->out_ch, kW, L.b_out);
  }
  A.n_jobs = n; A.n_outs = no;
  if (part_floats == nullptr) return A;   // the caller assigns the splits (layer-pair launch: make_wgall_plan)
  // split-K: the kernel streams saved activations once and is bound by that stream, so every workgroup
  // gets the same number of bytes: job i receives target * bytes_i / bytes workgroups, apportioned by
  // largest remainder so that the total is exactly one workgroup per CU — measured on MI355X (bench
  // workload): 224-256 workgroups 0.57 ms wgrad + 0.04 ms reduce per step, 512 (two rounds, twice the
  // partial sums) 0.59 + 0.08, 128 0.84.  SNR_WGRAD_SPLITS overrides the total for experiments.
  int64_t w[kMaxJobs], cap[kMaxJobs];
  int splits[kMaxJobs];
  for (int i = 0; i < n; ++i) { w[i] = A.job[i].a_ks + A.job[i].b_ks; cap[i] = A.n_tiles; }
  int target = tunables().wgrad_splits > 0 ? tunables().wgrad_splits : cu_count();
  apportion(w, cap, n, target, splits);
  int sb = 0;
  int64_t po = 0;
  for (int i = 0; i < n; ++i) place_job(A.job[i], splits[i], sb, po);
  *part_floats = po;
  *total_splits = sb;
  return A;
}

}  // namespace snr
