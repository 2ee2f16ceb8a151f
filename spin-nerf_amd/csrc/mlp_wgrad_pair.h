// Weight gradients of the eight trunk layers with selective recompute (bf16 mode, gfx950).  Included by mlp_bwd.hip.
//
// The plain split-K pass (mlp_wgrad.h) reads d z_i and h_{i-1} for every layer: 1 KB per sample per layer, after the
// chain kernels wrote the same 1 KB.  Here the forward saves only the ODD hidden layers (h1, h3, h5, h7) and dgrad only
// the odd d z (d z1, d z3, d z5, d z7); the trunk is handled as four layer PAIRS (2k, 2k+1), k = 0..3, each by two kinds
// of workgroup that stream the SAME two saved tensors X = h_{2k-1} (k = 0: the positional encoding) and D = d z_{2k+1}:
//
//   kind A   rebuilds  h_{2k}   = relu(W_{2k} X + b_{2k})                 and accumulates  dW_{2k+1} = sum_s D (x) h_{2k}
//   kind B   rebuilds  d z_{2k} = relu'(h_{2k}) * (W_{2k+1}^T D)          and accumulates  dW_{2k}   = sum_s d z_{2k} (x) X
//            (relu' from the 1-bit flags the forward saves for every layer)
//
// A layer's 256 x 256 fp32 accumulator is half of a CU's register file, so the two kinds cannot share a workgroup; they
// share the HBM fetch instead: the workgroups b and b ^ 8 run on the same XCD (block b -> XCD b % 8) and sweep the same
// tiles at the same pace (kind A, whose body is shorter, paces itself on kind B's progress word: see the tile loop), so
// the second request of a tile is an L2 hit (tests/probes/l2_share.hip: two such workgroups are served 12 TB/s of
// requests out of 6 TB/s of HBM).  Per sample and layer pair: 1 KB written by the chain kernels
// and ~1 KB fetched here, against 4 KB + 4 KB without the recompute.  The price is twice the MFMAs of the plain pass —
// this kernel is bound by the matrix pipe, not by the stream — so its shape is chosen for MFMA issue:
//
//   * 4 waves x 512 registers, one wave per SIMD; wave w owns the output columns [64 w, 64 w + 64) as two 32-column
//     blocks: 256 accumulator registers, and the rebuild layer's weights for those columns (2 x 16 fragments = 128
//     registers, read once from the packed blob: they are exactly the forward / dgrad weight fragments of that
//     column block, mlp_pack.h) — the weight operand never touches LDS.
//   * The rebuild is evaluated TRANSPOSED, R[sample][column] = X[sample][:] W[column][:]: the saved tile is the A
//     operand as it lies in LDS (ds_read_b128, the chain kernels' fragment layout), the C tile comes out with lane =
//     column, registers = 16 samples, and — after bias / relu / flags and conversion — IS the B operand of the
//     accumulation over samples (the k-slot order of samples is free as long as the A side agrees: the transposing
//     reads of D / X fetch the sample quadruples the C layout implies).  No LDS round trip, no cross-lane movement.
//   * Every LDS read serves two MFMAs (the wave's two column blocks share the A operand).
//   * Software pipeline over tiles: body i accumulates tile i with the operand rebuilt during body i-1 while it
//     rebuilds tile i+1; all reads are inline asm with counted waits (mlp_device.h), DMA pieces of tile i+RING-1
//     are dripped between the MFMAs; one barrier per tile.
#pragma once
#include "mlp_wgrad.h"

namespace snr {

struct PairHalf {
  int64_t part_off;        // float offset of this kind's partial planes [n_splits][256 columns][32 NM rows] (bf16 in the
                           // first half of each fp32-sized plane, like the plain pass)
  int64_t bias_part_off;   // float offset of its bias partials [n_splits][256]
  int w_frag;              // first fragment of the rebuild layer's weight image inside the packed blob
  int bias_off;            // kind A: float offset of b_{2k} inside the blob's bias block
};
struct PairJob {
  int64_t x_off;           // saved X section in the forward workspace ([n_tiles][x_ks KiB])
  int64_t dz_off;          // saved d z_{2k+1} section in the backward workspace ([n_tiles][16 KiB])
  int64_t flag_off;        // ReLU flags of layer 2k in the forward workspace ([n_tiles][1 KiB])
  int x_ks;                // 16 (hidden layer) or 4 (positional encoding)
  int slot_begin, n_splits;   // pair slots [slot_begin, slot_begin + n_splits): slot p = workgroups (16 (p / 8) + p % 8) + {0, 8}
  PairHalf a, b;
  // (round 4) one launch serves the layer pairs of SEVERAL networks (coarse + fine): every job carries its network's buffers
  const char* act;         // forward workspace
  const char* ws;          // backward workspace
  const char* blob;        // packed weights (forward section first)
  const float* bias;       // bias block inside the blob
  float* part;             // partial-sum buffer
  unsigned* sync;          // progress words of this job's slots (pacing, see pair_run), zeroed by the backward's dgrad launch
  int64_t n_tiles;
};
constexpr int kMaxPairs = 8, kMaxPlain = 6;   // 4 layer pairs + 3 plain jobs per network, two networks per launch
struct PairArgs {
  int n_pairs, n_plain;
  PairJob job[kMaxPairs];
  PlainJob plain[kMaxPlain];  // the jobs of the plain split-K pass (mlp_wgrad.h), run by the launch's workgroups beyond the pair slots
  int n_slots;                // pair slots of the launch
  int n_plain_wgs;            // plain workgroups of the launch
#ifdef SNR_PAIR_DEBUG
  int only_kind, only_pair;   // timing experiments (SNR_PAIR_KIND / SNR_PAIR_PAIR; debug builds only): -1 = all
#endif
  int sync_period, sync_lead; // kind A looks every sync_period tiles and waits while it is more than sync_lead tiles ahead; 0 = off
};

constexpr int kPairWaves = 4;
#ifndef SNR_PAIR_NTOP_A
#define SNR_PAIR_NTOP_A 6
#endif
#ifndef SNR_PAIR_NTOP_B
#define SNR_PAIR_NTOP_B 4
#endif
#ifndef SNR_PAIR_SWZ
#define SNR_PAIR_SWZ 1
#endif
#ifndef SNR_PAIR_ABLATE
#define SNR_PAIR_ABLATE 0   // timing experiments (results are garbage): 1 no DMA in the tile loop, 2 no LDS waits, 4 no barrier,
#endif                      // 8 no finishing VALU work (relu / flags), 16 no rebuild MFMAs, 32 no accumulating MFMAs, 64 no operand reads, 128 no row sums

// ---- the static schedule of one tile-loop body --------------------------------------------------------------------
// A body works on two tiles at once: it ACCUMULATES tile i (operand P(i), rebuilt by the previous body) and REBUILDS
// tile i+1.  Its MFMAs come in operand STEPS of two (the wave's two column blocks share the A operand):
//   R-step q     one fragment of tile i+1's rebuild section (ds_read_b128)             R[c] += frag x W[c][q]
//   A-step (m,t) row tile m, k-step t of tile i's accumulation section (2 transposing reads)   acc[c][m] += frag x P[c][t]
// One wave per SIMD issues in order, so whatever sits between two MFMAs runs in the 32-cycle shadow of the first one
// and no further: tests/probes/mfma_agpr.hip measures 4 other instructions per MFMA as free, 5 as +20 %, 6 as +50 %.
// The ~230 non-MFMA instructions of a body are therefore PLACED, at most about four to a gap (nothing moves across an
// MFMA: sched_barrier), in this order:
//   head    all A-steps of k-step 0 (their reads were issued by the previous body; nothing in them needs the barrier,
//           which follows step 0).  In their gaps: the upper half of the previous rebuild (C registers 8..15 = k-step 1's
//           operand, whose registers the previous body used up last) is converted, one packed word per gap.
//   middle  all R-steps; in their gaps the DMA pieces of tile i+RING-1, the pointer updates, kind B's column sums.
//   tail    all A-steps of k-step 1; in their gaps the lower half of the new rebuild is converted into k-step 0's
//           operand registers (free since the head) — the next body starts on it at once.
// Every LDS read is issued LA steps ahead of its MFMAs — the head's at the top of the body: none is in flight across the loop's
// back edge (for the compiler an asm read is complete when issued).  Completion is counted:
// LDS operations retire in order, so the wait in front of an even step (it covers the odd step behind it too) allows
// exactly the reads issued since (PairProg::young_pair).
enum { EV_ISSUE = 0, EV_FLAGS, EV_STEP_A, EV_STEP_B, EV_CVT, EV_SYNC, EV_DMA, EV_SUM, EV_SUMA, EV_ADVANCE, EV_M0 };
template <int TYPE, int KX> struct PairCfg {
  static constexpr bool PB = TYPE == 1;
  static constexpr int KR = PB ? 16 : KX;        // fragments of the rebuild's contraction
  static constexpr int NM = PB ? KX / 2 : 8;     // 32-row tiles of the accumulated product
  static constexpr int XO = 0, DZO = KX * 1024, FO = (KX + 16) * 1024;   // tile sections inside a ring slot
  static constexpr int SLOT = (KX + 16 + (PB ? 1 : 0)) * 1024;
  static constexpr int FIT = kLdsBytes / SLOT;
  static constexpr int RING = FIT < 5 ? FIT : 5;
  static constexpr int NIX = KX / kPairWaves, NID = 16 / kPairWaves;     // 1 KiB DMA pieces per wave: X, D
  static constexpr int NI = NIX + NID + (PB ? 1 : 0);                    // ... + this wave's quarter of the flags
  static constexpr int RO = PB ? DZO : XO;       // section the rebuild reads (fragments as stored)
  static constexpr int AO = PB ? XO : DZO;       // section the accumulation reads (transposing reads)
  static constexpr int NR = KR, NA = 2 * NM, NS = NR + NA;               // operand steps of one body
  static constexpr int LA = 4;                   // operand steps the LDS reads run ahead of their MFMAs
  static constexpr int WIN = NS % 8 == 0 ? 8 : 5;   // operand register sets, step s uses set s % WIN (a divisor of NS, > LA;
                                                 // only the LA + 1 sets with reads in flight are live)
  static constexpr int LEADA = NM;               // head: the A-steps of k-step 0 (step 0 ahead of the barrier)
  static constexpr int TAILA = NM;               // tail: the A-steps of k-step 1
  static constexpr int SYNC_STEP = 0;            // the barrier follows this step
};

template <int TYPE, int KX> struct PairProg {
  using C = PairCfg<TYPE, KX>;
  static constexpr int NS = C::NS, NM = C::NM, NR = C::NR, LA = C::LA, MAXEV = 6 * C::NS + 96;
  int order[NS];   // >= 0: R-step q;  < 0: A-step ~e, e = t * NM + m
  int ip[NS];      // position (= "in front of step ip") at which the step's reads are issued; < 0: NS + ip of the previous body
  int kind[MAXEV], arg[MAXEV], nxt[MAXEV], n;
  constexpr void push(int k, int a, int b) { kind[n] = k; arg[n] = a; nxt[n] = b; ++n; }
  constexpr PairProg() : order{}, ip{}, kind{}, arg{}, nxt{}, n(0) {
    // ---- step order ----
    int s = 0, a0 = 0, a1 = 0, r = 0;
    for (; a0 < C::LEADA; ++a0) order[s++] = ~(0 * NM + a0);
    const int nA_mid = (NM - C::LEADA) + (NM - C::TAILA);
    int am = 0;
    while (r < NR || am < nA_mid) {
      const bool pick_r = r < NR && (am >= nA_mid || (int64_t)r * nA_mid <= (int64_t)am * NR);
      if (pick_r) order[s++] = r++;
      else {
        if (a0 < NM) order[s++] = ~(0 * NM + a0++);
        else order[s++] = ~(1 * NM + a1++);
        ++am;
      }
    }
    for (; a1 < NM; ++a1) order[s++] = ~(1 * NM + a1);
    // ---- issue positions ----
    for (int i = 0; i < NS; ++i) {
      ip[i] = i - LA;
      if (order[i] >= 0 && ip[i] < C::SYNC_STEP + 1) ip[i] = C::SYNC_STEP + 1;   // rebuild reads: behind the barrier
    }
    // ---- events: step p = MFMA a | gap 2p | MFMA b | gap 2p+1 ----
    // gap 2p holds the reads issued at position p; the odd gaps hold everything else.  The 8 conversion items of a half
    // (packed word d = j / 2 of column block c = j % 2) go one to an odd gap of the head (upper half) / tail (lower half)
    // when there are 8 steps, else evenly over all of its gaps.
    const int first_tail = NS - C::TAILA, G = 2 * NS;
    const int hi_gaps = 2 * C::LEADA - 1;            // gaps 0 .. hi_gaps-1 precede the first R-step's first MFMA
    const int lo_gap0 = 2 * first_tail + 1;          // first gap behind the tail's first step (two MFMAs behind the last R MFMA)
    const int lo_gaps = G - lo_gap0;
    const int dma_step0 = C::LEADA;                  // DMA pieces: behind MFMA b of the middle's steps
    const int adv_step0 = dma_step0 + C::NI;         // then the pointer updates (4 parts); kind B's column sums sit beside the first 8 pieces
    // Top of the body, in front of the first MFMA: the reads of the leading steps and of kind B's k-step-1 flag words, then
    // the first conversion items (they need no LDS data: the wait for the reads falls behind them).  No asm read is ever in
    // flight across the loop's back edge or at its exit: for the compiler such a read is complete when issued, and a copy it
    // places at a loop boundary (or a reuse of the register behind the loop) would meet the old register content.
    const int n_top = C::PB ? SNR_PAIR_NTOP_B : SNR_PAIR_NTOP_A;   // conversion micro-items at the top
    if (C::PB) { push(EV_FLAGS, 2, 0); push(EV_FLAGS, 3, 0); }
    for (int i = 0; i < NS; ++i) if (ip[i] < 0) push(EV_ISSUE, i, 0);
    for (int u = 0; u < n_top; ++u) push(EV_CVT, 16 * 1 + u, 0);
    for (int p = 0; p < NS; ++p) {
      push(EV_STEP_A, p, 0);
      for (int half = 0; half < 2; ++half) {         // gap 2p (half 0), then MFMA b, then gap 2p+1 (half 1)
        const int g = 2 * p + half;
        if (half == 0) {
          // (... second half — in front of this gap's reads: with WIN = LA + 1 the reads of step p + LA reuse the registers of step p - 1)
          if (p > 0 && order[p - 1] < 0 && !C::PB) push(EV_SUMA, 2 * (p - 1) + 1, 0);
          for (int i = 0; i < NS; ++i) if (ip[i] == p) push(EV_ISSUE, i, 0);
          // M0 of the DMA piece issued behind this step's second MFMA: written a gap ahead, the MFMA in between is the wait
          // state the hardware wants between a scalar write of M0 and the LDS-DMA that reads it
          if (p >= dma_step0 && p < adv_step0) push(EV_M0, p - dma_step0, 0);
        } else {
          push(EV_STEP_B, p, 0);
          if (p == C::SYNC_STEP) push(EV_SYNC, 0, 0);
          if (order[p] < 0 && !C::PB) push(EV_SUMA, 2 * p, 0);   // kind A: row sums of this A-step's operand (if the wave owns the rows),
          if (p >= dma_step0 && p < adv_step0) push(EV_DMA, p - dma_step0, 0);
          if (p >= adv_step0 && p < adv_step0 + 4) push(EV_ADVANCE, p - adv_step0, 0);
          if (C::PB && p >= dma_step0 && p < dma_step0 + 8) push(EV_SUM, p - dma_step0, 0);   // two words beside each DMA piece
          if (C::PB && p == first_tail - 1) { push(EV_FLAGS, 0, 1); push(EV_FLAGS, 1, 1); }   // next tile's k-step 0 flags: the tail's conversion
        }
        // 16 conversion micro-items per half of R: item j = 2 d + c (packed word d of column block c), part 0 (add / convert)
        // and part 1 (relu | flags) in different gaps, spread evenly
        if (g == G - 1 && order[NS - 1] < 0 && !C::PB) push(EV_SUMA, 2 * (NS - 1) + 1, 0);
        for (int u = 0; u < 16; ++u) {
          if (u >= n_top && (u - n_top) * hi_gaps / (16 - n_top) == g && g < hi_gaps) push(EV_CVT, 16 * 1 + u, 0);
          if (lo_gap0 + u * lo_gaps / 16 == g) push(EV_CVT, 16 * 0 + u, 0);
        }
      }
    }
  }
  constexpr int reads_of(int e) const {
    return kind[e] == EV_ISSUE ? (order[arg[e]] >= 0 ? 1 : 2) : (kind[e] == EV_FLAGS ? 2 : 0);
  }
  // LDS reads issued behind event `issue` and in front of event `use` (cyclically, when the issue belongs to the previous body)
  constexpr int between(int issue, int use) const {
    int c = 0;
    if (issue < use) { for (int j = issue + 1; j < use; ++j) c += reads_of(j); }
    else { for (int j = issue + 1; j < n; ++j) c += reads_of(j); for (int j = 0; j < use; ++j) c += reads_of(j); }
    return c;
  }
  constexpr int find(int k, int a) const { for (int j = 0; j < n; ++j) if (kind[j] == k && arg[j] == a) return j; return -1; }
  // e: the EV_STEP_A event of an even step; its wait also covers step + 1 (one s_waitcnt per two steps), whose reads are the younger
  constexpr int young_pair(int e) const {
    const int is = find(EV_ISSUE, arg[e] + 1);
    return is < 0 ? 15 : between(is, e);
  }
  constexpr int young_flags(int t, int e) const {   // e: the event that uses them
    const int is = find(EV_FLAGS, t);
    return is < 0 ? 15 : between(is, e);
  }
  constexpr int first_rstep() const { for (int i = 0; i < NS; ++i) if (order[i] >= 0) return i; return -1; }
  constexpr bool valid() const {
    if (NS % C::WIN || C::WIN <= LA || NS % 2) return false;
    for (int i = 0; i < NS; ++i) {
      if (ip[i] >= i - (i & 1) || (ip[i] < 0 && order[i] >= 0)) return false;      // issued before the pair's wait; R-steps never early
      if (order[i] >= 0 && i >= NS - C::TAILA) return false;
    }
    // k-step 0 is used up before the tail rewrites its operand; k-step 1 only after the head rewrote it
    for (int i = 0; i < NS; ++i) if (order[i] < 0) {
      const int t = (~order[i]) / NM;
      if (t == 0 && i >= NS - C::TAILA) return false;
      if (t == 1 && i <= C::SYNC_STEP) return false;
    }
    return n <= MAXEV && (C::PB ? C::LEADA + C::NI + 8 <= NS - C::TAILA + 1 : C::LEADA + C::NI + 4 <= NS) && first_rstep() == C::LEADA;
  }
};

template <int OFF> __device__ __forceinline__ void pair_read16(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void pair_read16u(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
#if SNR_PAIR_ABLATE & 2
#define SNR_PAIR_LGKM(N) 15
#else
#define SNR_PAIR_LGKM(N) ((N) < 15 ? (N) : 15)
#endif
template <typename T> __device__ __forceinline__ void pair_wait_tie(T& r) { asm volatile("" : "+v"(r)); }
// The wait and its registers.  A wait whose asm statement itself carries the covered registers ("+v") is padded by the
// compiler: gfx950's forwarding-hazard rule treats every inline asm that defines a register as a possible partial write,
// asks for one wait state before the next reader, and counts other asm statements as zero wait states — an `s_nop 0`
// behind every wait in front of its MFMA, 16 to 21 per body.  No tie at all is wrong: the destination of a read whose
// result the compiler sees no use for (body -1 issues the accumulation reads and skips their MFMAs) is handed to the next
// temporary while the read is in flight (garbage on the GPU; tools/check_lds_asm.py flags 121 places in that build).  So:
// the registers are tied in an empty statement IN FRONT of a compiler-visible s_waitcnt — they stay allocated up to the
// wait, the wait itself is the wait state the hazard rule wants, and the readers (asm volatile MFMAs, or VALU code behind
// the gap's sched_barrier) cannot move above it.  The checker proves the result on the listing at build time.
template <int N, typename... T> __device__ __forceinline__ void pair_wait(T&... regs) {
  (pair_wait_tie(regs), ...);
  __builtin_amdgcn_s_waitcnt(0xC07F | (SNR_PAIR_LGKM(N) << 8));   // vmcnt 63, expcnt 7: lgkmcnt(N) only
}

// The 256 accumulator registers are the whole accumulator file of a one-wave-per-SIMD kernel; with the rebuild's own
// 32 result registers the kernel holds more MFMA results than there are AGPRs, and left to itself the register
// allocator rotates accumulator tiles through arch VGPRs (v_accvgpr_read / mov / write around every MFMA).  Every MFMA
// is therefore inline asm: the accumulating ones with their tile tied in place in the accumulator file, the rebuild's
// with theirs in arch VGPRs (once a kernel has accumulator-file operands the compiler selects the accumulator form for
// every builtin MFMA, and there is no room left there).  Hazards the compiler no longer pads (cdna_hip_programming.md
// §5.7): a VALU-written operand needs 2 wait states in front of the MFMA (pair_operand_ready: s_nop 1 tied to the
// registers; the A operands come from LDS reads and their wait); an MFMA result needs 12 before a VALU read
// (pair_r_complete in front of the conversions; the accumulators are read after the tile loop, behind nops).
__device__ __forceinline__ void pair_mfma_acc(f32x16& acc, const bf16x8& a, const u32x4& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// FIRST: the chain's first MFMA takes the constant 0 as C — no initial value to materialise (a splat kept for that cost 16
// registers per column block; kind A adds the bias when the tile leaves the fp32 registers).  Its extra operands are the
// freshly converted k-step-1 operands: tying them here makes the conversion (the last reader of the previous R) precede
// the statement that re-defines R, so that no copy of R is needed.
__device__ __forceinline__ void pair_mfma_reb_first(f32x16& r, const bf16x8& a, const bf16x8& b, u32x4& p0, u32x4& p1) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, 0" : "=&v"(r), "+v"(p0), "+v"(p1) : "v"(a), "v"(b));
}
// kind A: C = the column's bias in all 16 registers (in the transposed tile a lane is one column of h_2k, so its bias is one
// value per lane): the addition rides in the MFMA instead of costing a v_pk_add_f32 — and the wait state gfx950 wants
// between that and the conversion that reads it — per packed word, 32 instructions a body
__device__ __forceinline__ void pair_mfma_reb_first(f32x16& r, const bf16x8& a, const bf16x8& b, u32x4& p0, u32x4& p1, const f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %5" : "=&v"(r), "+v"(p0), "+v"(p1) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void pair_mfma_reb(f32x16& r, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(r) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pair_r_complete(f32x16& r0, f32x16& r1) { asm volatile("s_nop 7\n\ts_nop 4" : "+v"(r0), "+v"(r1)); }
__device__ __forceinline__ void pair_operand_ready(u32x4& b0, u32x4& b1) { asm volatile("s_nop 1" : "+v"(b0), "+v"(b1)); }

// Kind B's finishing touches on one packed word (two samples of one neuron column): keep the low / high bf16 of x where bit
// `bit` of w0 / w1 (the two samples' flag words) is set.  Compiler-visible (v_bfe_i32 x2, v_perm_b32, v_and_b32): as inline
// asm the same four instructions needed copies of the flag words (they work in place) and a forwarding-hazard pad each;
// the schedule's sched_barriers keep the compiler from expanding a k-step's 16 words at once, which is what made asm
// necessary before the micro-items were pinned gap by gap.
__device__ __forceinline__ void pk_flags_extract(unsigned& w0, unsigned& w1, unsigned bit) {   // -> 0 / -1 each
  w0 = (unsigned)__builtin_amdgcn_sbfe((int)w0, bit, 1u);
  w1 = (unsigned)__builtin_amdgcn_sbfe((int)w1, bit, 1u);
}
__device__ __forceinline__ void pk_flags_apply(unsigned& x, unsigned& m0, unsigned m1) {   // low half by m0, high half by m1
  x &= __builtin_amdgcn_perm(m1, m0, 0x07060100u);   // bytes 3, 2 of m1 over bytes 1, 0 of m0
}
//   sum += low bf16 + high bf16 of x  (v_dot2c_f32_bf16 against (1, 1): one instruction per word)
__device__ __forceinline__ void pk_sum_bf16_2(float& sum, unsigned x0, unsigned x1) {
  asm("v_dot2c_f32_bf16 %0, %3, %1\n\tv_dot2c_f32_bf16 %0, %3, %2" : "+v"(sum) : "v"(x0), "v"(x1), "s"(0x3f803f80u));
}
// WV: the wave index as a compile-time constant (kind A: which row tiles' sums this wave keeps is decided without branches)
template <int TYPE, int KX, int WV>
__device__ __forceinline__ void pair_run(const PairArgs& a, const PairJob& J, int split, char* smem, int wave, int lane) {
  using C = PairCfg<TYPE, KX>;
  using Frag = bf16x8;
  constexpr bool PB = C::PB;
  constexpr int KR = C::KR, NM = C::NM, NI = C::NI, RING = C::RING, SLOT = C::SLOT, NS = C::NS, WIN = C::WIN;
  static constexpr PairProg<TYPE, KX> PG{};
  static_assert(PG.valid(), "inconsistent body schedule");
  const PairHalf& H = PB ? J.b : J.a;

  // tiles split, split + n_splits, ... (the same sequence in the kind-A and the kind-B workgroup of a slot)
  const int64_t tstep = J.n_splits;
  const int64_t t1 = (J.n_tiles - split + tstep - 1) / tstep;   // >= 1: the host never makes more splits than tiles

  // ---- DMA: this wave's pieces wave + 4 k of X and of D, and its quarter of the flag KiB -------------------------------
  const char* xp = J.act + J.x_off + ((int64_t)split * KX + wave) * 1024;
  const char* dp = J.ws + J.dz_off + ((int64_t)split * 16 + wave) * 1024;
  const char* fp = J.act + J.flag_off + (int64_t)split * 1024;
  const int64_t xs = (int64_t)KX * 1024 * tstep, dst_ = (int64_t)16 * 1024 * tstep, fs_ = (int64_t)1024 * tstep;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // flags: the forward stores one u32x4 per lane (sample s, half g) = [g][s][word]; the LDS image is [g][word][s], so that
  // the four samples a C register quad covers are one 16-byte read.  dword d = 64 wave + lane of the image:
  const int fd = 64 * wave + lane;
  const uint32_t f_lane = (uint32_t)((((fd >> 7) * 32 + (fd & 31)) * 4 + ((fd >> 5) & 3)) * 4);
  int64_t issued = 0;   // tiles issued so far
  bool go = false;
  // (Prologue) one statement per piece: M0 = LDS destination of the wave's 1 KiB (the hardware adds lane x 16), a wait state, the
  // load with a wave-uniform 64-bit base + a 32-bit lane offset (lane16 + 4096 k lives in four registers: no address
  // arithmetic per piece).  M0 is written in the statement that reads it (the compiler does not preserve it around asm);
  // the scalar add clobbers SCC, which the compiler may hold live across the statement (found as intermittent garbage).
  // (No LDS read may sit in the cycle in front of an LDS-DMA — mlp_device.h, Pipe::issue_one: the schedule places every
  //  piece behind a step's second MFMA, the prologue's behind nothing.)
  // The section the rebuild reads with ds_read_b128 (X in kind A, D in kind B) is stored with the two 16-byte halves of the
  // samples 16..31 of every fragment swapped: a b128 read is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
  // (+ 32), and with the halves as stored by the forward pass (chunk 2 s + half) a group touches only every other 16-byte
  // bank quad — twice: a 2-way conflict on every rebuild read.  The DMA does the swap on the way in (its lanes 32..63 fetch
  // the neighbouring chunk; the global side stays inside the same 1 KiB), the reading lanes undo it in their address.
  uint32_t voff[4], voffs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    voff[k] = lane16 + 4096u * k;
    voffs[k] = SNR_PAIR_SWZ ? ((uint32_t)(lane ^ (lane >> 5)) * 16u + 4096u * k) : voff[k];
  }
  const uint32_t lds_w = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
  // (the flag KiB: four bytes per lane, gathered into the [g][word][s] image; this wave's quarter starts 256 wave bytes in:
  //  lds_w holds 1024 wave)
  const uint32_t flag_w = C::FO - 768 * wave;
  auto issue_flags = [&](uint32_t sl) {
    asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %3" :: "v"(f_lane), "s"(sl), "s"(flag_w), "s"(fp) : "scc");
  };
  auto issue_piece = [&](int slot, auto K_) {
    constexpr int k = decltype(K_)::value;
    const uint32_t sl = lds_w + slot * SLOT;
#if SNR_PAIR_ABLATE & 1
    if (issued > 2 * RING) return;
#endif
    if constexpr (k < C::NIX) {
      asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3"
                   :: "v"(PB ? voff[k] : voffs[k]), "s"(sl), "n"(C::XO + 4096 * k), "s"(xp) : "scc");
    } else if constexpr (k < C::NIX + C::NID) {
      asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3"
                   :: "v"(PB ? voffs[k - C::NIX] : voff[k - C::NIX]), "s"(sl), "n"(C::DZO + 4096 * (k - C::NIX)), "s"(dp) : "scc");
    } else if constexpr (PB && k == C::NIX + C::NID) {
      issue_flags(sl);
    }
  };
  // In the tile loop the two halves of a piece are separate events (EV_M0 in the gap in front of the step's second MFMA,
  // the load behind it): the MFMA is the wait state, and the gap that takes the load holds one instruction, not three.
  auto flags_m0 = [&](uint32_t sl) { asm volatile("s_add_u32 m0, %0, %1" :: "s"(sl), "s"(flag_w) : "scc"); };
  auto flags_load = [&]() { asm volatile("global_load_lds_dword %0, %1" :: "v"(f_lane), "s"(fp)); };
  auto piece_m0 = [&](int slot, auto K_) {
    constexpr int k = decltype(K_)::value;
    const uint32_t sl = lds_w + slot * SLOT;
    if constexpr (k < C::NIX) asm volatile("s_add_u32 m0, %0, %1" :: "s"(sl), "n"(C::XO + 4096 * k) : "scc");
    else if constexpr (k < C::NIX + C::NID) asm volatile("s_add_u32 m0, %0, %1" :: "s"(sl), "n"(C::DZO + 4096 * (k - C::NIX)) : "scc");
    else if constexpr (PB && k == C::NIX + C::NID) flags_m0(sl);
  };
  auto piece_load = [&](auto K_) {
    constexpr int k = decltype(K_)::value;
#if SNR_PAIR_ABLATE & 1
    if (issued > 2 * RING) return;
#endif
    if constexpr (k < C::NIX) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(PB ? voff[k] : voffs[k]), "s"(xp));
    else if constexpr (k < C::NIX + C::NID) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(PB ? voffs[k - C::NIX] : voff[k - C::NIX]), "s"(dp));
    else if constexpr (PB && k == C::NIX + C::NID) flags_load();
  };
  auto advance = [&]() {   // past the end the last tile is loaded again: every body issues the same NI instructions
    if (issued + 1 < t1) { ++issued; xp += xs; dp += dst_; fp += fs_; }
  };

  // ---- weights of this wave's two column blocks (registers for the whole kernel) ---------------------------------------
  Frag W[2][KR];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < KR; ++q)
      W[c][q] = *(const Frag*)(J.blob + (((int64_t)H.w_frag + (2 * wave + c) * KR + q) * 64 + lane) * 16);
  // kind A: the bias of this lane's column of h_2k, 16 copies = the C operand of the rebuild chain's first MFMA (opaque to the
  // compiler, or it would re-materialise the copies in front of every use)
  f32x16 biasC[2];
  if constexpr (!PB) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float bj = J.bias[H.bias_off + 64 * wave + 32 * c + (lane & 31)];
#pragma unroll
      for (int r = 0; r < 16; ++r) biasC[c][r] = bj;
      asm volatile("" : "+v"(biasC[c]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Kind A of pair 0 rebuilds h0 = relu(W0 enc + b0) from the SAVED encoding, which is bf16; the forward's own fragments of W0
  // are fp16 (mlp_layout.h: EncF16): re-rounded to bf16 here, once per launch.  The rebuilt h0 then differs from the forward's
  // by the encoding's bf16 rounding (a bf16 ulp of h0 in a fraction of its elements) — it only meets d z1 in dW1.
  if constexpr (!PB && KR == 4 && EncF16<kBF16>::value) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < KR; ++q) W[c][q] = Mma<kBF16>::f16_to_bf16(W[c][q]);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < KR; ++q) asm volatile("" : "+v"(W[c][q]));

  // ---- per-lane LDS offsets ---------------------------------------------------------------------------------------------
  // rebuild A operand: lane (sample s = lane & 31, k-half gg) reads its 16 bytes of fragment q (row act_row(s, q))
  const int s32 = lane & 31, gg = lane >> 5;
  const uint32_t lds0 = lds_addr(smem);
  const int ggs = SNR_PAIR_SWZ ? (gg ^ (s32 >> 4)) : gg;   // (the swapped halves of samples 16..31, see the DMA offsets)
  const uint32_t rdE = C::RO + ggs * 16 + s32 * 32, rdO = C::RO + ggs * 16 + (s32 ^ 4) * 32;
  // accumulation A operand (transposing reads, mlp_wgrad.h): 16-lane group (gg, bh) receives neuron column ip of the
  // [4 samples][16 neurons] block of fragment 2 m + bh whose samples are 16 t + 8 q + 4 gg + 0..3 — the k-slot order the
  // rebuilt C tile implies: register 8 t + e of lane half gg holds sample 16 t + 8 (e >> 2) + 4 gg + (e & 3)
  const int ip = lane & 15, bh = (lane >> 4) & 1, c4 = ip & 3, r4 = ip >> 2;
  const uint32_t trA = C::AO + bh * 1024 + (((4 * gg + r4) ^ (4 * bh)) * 32) + c4 * 8;   // + m * 2048 + t * 512 + q * 256
  // flags of column n = 32 cb + i of layer 2k (chain tile nt = cb, lane half g_n, C register r_n): word nt >> 1, bit below
  uint32_t fl_addr[2] = {0, 0}, fl_bit[2] = {0, 0};
  if constexpr (PB) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int nt = 2 * wave + c, i = lane & 31;
      const int r_n = (i & 3) + 4 * (i >> 3), g_n = (i >> 2) & 1;
      fl_bit[c] = (r_n >> 1) + 8 * (nt & 1) + 16 * (r_n & 1);
      fl_addr[c] = C::FO + ((g_n * 4 + (nt >> 1)) * 32 + 4 * gg) * 4;   // + 32 j: samples 8 j + 4 gg + 0..3
    }
  }

  f32x16 acc[2][NM];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.f;
  float bsum[2] = {0.f, 0.f};   // kind A: row sums of D for row tiles 2 wave, 2 wave + 1; kind B: column sums of the rebuilt d z

  // ---- state carried from body to body ----
  f32x16 R[2];                   // rebuild of the next tile (fp32); its upper half is converted by the next body
  u32x4 P[2][2];                 // [column block][k-step]: the current tile's operand, packed bf16
  u32x4 fw[2][2];                // kind B: flag words of one k-step, [column block][quad]
  Frag rf[WIN];                  // operand registers in flight
  bf16x4 tl[WIN], th[WIN];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) R[c][r] = 0.f;
    P[c][0] = P[c][1] = fw[c][0] = fw[c][1] = u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < WIN; ++i) { rf[i] = Frag{0, 0, 0, 0, 0, 0, 0, 0}; tl[i] = th[i] = bf16x4{0, 0, 0, 0}; }

  int slot = RING - 1, slot_next = 0, islot = RING - 2;   // body -1: "tile -1" lives in the (empty) last slot

  // conversion item (h, d, c): C registers 8 h + 2 d, + 1 of column block c -> packed word d of P[c][h]; part 0 = bias add
  // (kind A) | convert + flag bits to masks (kind B), part 1 = convert + relu (kind A) | apply the masks (kind B)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  f32x2_t cv = {0.f, 0.f};
  auto convert_item = [&](auto H_, auto D_, auto C_, auto PART_) {
    constexpr int h = decltype(H_)::value, d = decltype(D_)::value, c = decltype(C_)::value, part = decltype(PART_)::value;
    if constexpr (!PB) {
      if constexpr (part == 0) {
        cv = f32x2_t{R[c][8 * h + 2 * d], R[c][8 * h + 2 * d + 1]};   // (bias: the chain's C operand)
      } else {
        unsigned x = __builtin_bit_cast(unsigned, __builtin_convertvector(cv, bf16x2_t));
#if !(SNR_PAIR_ABLATE & 8)
        x = pk_relu_bf16(x);
#endif
        P[c][h][d] = x;
      }
    } else {
      // quad (d >> 1) of this k-step's flag words, words 2 (d & 1), + 1 of it (consumed in place)
      unsigned w0 = fw[c][d >> 1][2 * (d & 1)], w1 = fw[c][d >> 1][2 * (d & 1) + 1];
      if constexpr (part == 0) {
        const f32x2_t v = f32x2_t{R[c][8 * h + 2 * d], R[c][8 * h + 2 * d + 1]};
        P[c][h][d] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#if !(SNR_PAIR_ABLATE & 8)
        pk_flags_extract(w0, w1, fl_bit[c]);
        fw[c][d >> 1][2 * (d & 1)] = w0; fw[c][d >> 1][2 * (d & 1) + 1] = w1;
#endif
      } else {
#if !(SNR_PAIR_ABLATE & 8)
        unsigned x = P[c][h][d];
        pk_flags_apply(x, w0, w1);
        P[c][h][d] = x;
#endif
      }
    }
    if constexpr (d == 3 && c == 1 && part == 1) pair_operand_ready(P[0][h], P[1][h]);
  };

  unsigned* sync_w = J.sync + split;   // this slot's progress word (pacing)
  unsigned post_word = 0, post_zero = 0;   // (VGPRs: the store's data and its zero offset)
  asm volatile("" : "+v"(post_zero));
  // One body.  FIRST = body -1: rebuilds tile 0, accumulates nothing.
  auto body = [&](auto FIRST_) {
    constexpr bool FIRST = decltype(FIRST_)::value;
    const uint32_t sA = lds0 + slot * SLOT, sN = lds0 + slot_next * SLOT;
    const uint32_t aE = sN + rdE, aO = sN + rdO, aT = sA + trA, aTn = sN + trA;
    static_for<0, PG.n>([&](auto I_) {
      constexpr int ei = decltype(I_)::value;
      constexpr int kind = PG.kind[ei], arg = PG.arg[ei], nxt = PG.nxt[ei];
      if constexpr (kind == EV_ISSUE && !(SNR_PAIR_ABLATE & 64)) {
        constexpr int o = PG.order[arg];
        if constexpr (o >= 0) {
          pair_read16<o * 1024>(rf[arg % WIN], (o & 1) ? aO : aE);
        } else {
          constexpr int e = ~o, t = e / NM, m = e % NM;
          tr_read<m * 2048 + t * 512>(tl[arg % WIN], nxt ? aTn : aT);
          tr_read<m * 2048 + t * 512 + 256>(th[arg % WIN], nxt ? aTn : aT);
        }
      } else if constexpr (kind == EV_FLAGS) {
        // flag words of k-step t = arg / 2, quad arg % 2 (C registers 8 t + 4 quad + 0..3), both column blocks: k-step 0's belong to
        // the NEXT tile (the rebuild this body finishes; its slot holds them since the barrier), k-step 1's to this body's tile
        constexpr int t = arg / 2, quad = arg % 2;
#pragma unroll
        for (int c = 0; c < 2; ++c) pair_read16u<64 * t + 32 * quad>(fw[c][quad], (nxt ? sN : sA) + fl_addr[c]);
      } else if constexpr (kind == EV_STEP_A || kind == EV_STEP_B) {
        constexpr int o = PG.order[arg];
        constexpr int c = kind == EV_STEP_B ? 1 : 0;   // the column block of this MFMA
        if constexpr (kind == EV_STEP_A && arg % 2 == 0) {   // one wait per two steps
          constexpr int young = PG.young_pair(ei);
          constexpr int o1 = PG.order[arg + 1];
          constexpr int w0 = arg % WIN, w1 = (arg + 1) % WIN;
          if constexpr (o >= 0 && o1 >= 0) pair_wait<young>(rf[w0], rf[w1]);
          else if constexpr (o >= 0) pair_wait<young>(rf[w0], tl[w1], th[w1]);
          else if constexpr (o1 >= 0) pair_wait<young>(rf[w1], tl[w0], th[w0]);
          else pair_wait<young>(tl[w0], th[w0], tl[w1], th[w1]);
        }
        if constexpr (o >= 0) {
          Frag& x = rf[arg % WIN];
          if constexpr (!(SNR_PAIR_ABLATE & 16) || o == 0) {
            if constexpr (o == 0 && PB) pair_mfma_reb_first(R[c], x, W[c][0], P[0][1], P[1][1]);
            else if constexpr (o == 0) pair_mfma_reb_first(R[c], x, W[c][0], P[0][1], P[1][1], biasC[c]);
            else pair_mfma_reb(R[c], x, W[c][o]);
          }
        } else {
          constexpr int e = ~o, t = e / NM, m = e % NM;
          bf16x4 &lo = tl[arg % WIN], &hi = th[arg % WIN];
          if constexpr (!FIRST) {
            // (tools/check_lds_asm.py verifies on the listing that no VALU instruction writes an MFMA operand in the two
            //  wait states in front of it: the compiler is free to assemble this tuple with moves)
            const Frag fa = Frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            // kind A multiplies the other way round — P as the A operand (rows = its columns j), the transposed read as B (columns =
            // d z slots): the tile comes out as dW_{2k+1}^T... i.e. [j][slot] with lane = slot, which makes its partial plane
            // [slot][j] row-major like the plain pass's (the reduce kernel scatters it with coalesced stores); the register
            // images of the two operands are the same either way
            if constexpr (!(SNR_PAIR_ABLATE & 32) || m == 0) {
              if constexpr (PB) pair_mfma_acc(acc[c][m], fa, P[c][t]);
              else pair_mfma_acc(acc[c][m], __builtin_bit_cast(Frag, P[c][t]), __builtin_bit_cast(u32x4, fa));
            }
          }
        }
        // nothing moves across an MFMA: what the schedule puts between two of them stays in that shadow (left alone, the
        // compiler gathers the conversions in front of the body's first MFMA)
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (kind == EV_SYNC) {
        // tile i+1 has landed (this wave's pieces: loads retire in order, RING - 3 younger tiles may be outstanding); behind
        // the barrier that holds for everybody, and everybody is done with tile i-1: its slot can be refilled
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * NI) : "memory");
#if !(SNR_PAIR_ABLATE & 4)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        // kind B's progress word (pacing, see the tile loop): right behind the wait, so that the store — vmcnt counts it, in
        // order with the DMA pieces — has a whole body to complete before the next counted wait looks at the queue
        // (unconditional: a branch here would cut the body into two scheduling regions)
        if constexpr (PB && !FIRST) asm volatile("global_store_dword %0, %1, %2" :: "v"(post_zero), "v"(post_word), "s"(sync_w));
      } else if constexpr (kind == EV_CVT) {
        constexpr int h = arg / 16, j = (arg % 16) / 2, part = arg % 2, d = j / 2, c = j % 2;
        static_assert(d < 4 && h < 2, "conversion item out of range");
        if constexpr (j == 0 && part == 0) {
          if constexpr (h == 0) pair_r_complete(R[0], R[1]);   // the rebuild's last MFMAs are only a step back
        }
        if constexpr (PB && c == 0 && d % 2 == 0 && part == 0) {   // the quad of flag words this item and the next three use
          constexpr int y = PG.young_flags(2 * h + d / 2, ei);
          pair_wait<y>(fw[0][d / 2], fw[1][d / 2]);
        }
        convert_item(std::integral_constant<int, h>{}, std::integral_constant<int, d>{}, std::integral_constant<int, c>{},
                     std::integral_constant<int, part>{});
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (kind == EV_M0) {
        piece_m0(islot, std::integral_constant<int, arg>{});
      } else if constexpr (kind == EV_DMA) {
        piece_load(std::integral_constant<int, arg>{});
      } else if constexpr (kind == EV_SUM) {
        if constexpr (!FIRST) {   // (k-step arg / 4, column block arg % 2, words 2 q, 2 q + 1 with q = arg / 2 % 2)
          const u32x4& w = P[arg % 2][arg / 4];
          pk_sum_bf16_2(bsum[arg % 2], w[2 * (arg / 2 % 2)], w[2 * (arg / 2 % 2) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (kind == EV_SUMA) {
        // kind A, bias gradient of layer 2k+1: the wave sums the rows of its two row tiles (the step's operand registers are
        // not reused before the reads of step + WIN are issued, LA - WIN steps from here)
        constexpr int st = arg / 2, part = arg % 2;
        constexpr int o = PG.order[st];
        constexpr int m = (~o) % NM;
        if constexpr (!FIRST && !(SNR_PAIR_ABLATE & 128) && m / 2 == WV) {
          bf16x4 &lo = tl[st % WIN], &hi = th[st % WIN];
          const u32x4 w = __builtin_bit_cast(u32x4, Frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
          pk_sum_bf16_2(bsum[m & 1], w[2 * part], w[2 * part + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (kind == EV_ADVANCE) {
        // all pieces of this body are issued: the pointers and the ring positions move on (the addresses of this body's
        // reads were formed at its top)
        if constexpr (arg == 0) { go = issued + 1 < t1; issued += go ? 1 : 0; }
        if constexpr (arg == 1) { xp += go ? xs : 0; dp += go ? dst_ : 0; }
        if constexpr (arg == 2) { fp += go ? fs_ : 0; islot = islot + 1 == RING ? 0 : islot + 1; }
        if constexpr (arg == 3) { slot = slot_next; slot_next = slot_next + 1 == RING ? 0 : slot_next + 1; }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };

  // ---- prologue: tiles 0 .. RING-3 in flight, body -1 rebuilds tile 0 (and issues tile RING-2) ----
  for (int d = 0; d < RING - 2; ++d) {
    static_for<0, NI>([&](auto K_) { issue_piece(d, K_); });
    advance();
  }
  body(std::true_type{});
  // Pacing.  The two workgroups of a slot (kinds A and B, same XCD) stream the same tiles, and the second one to ask finds them
  // in L2 — as long as it asks soon enough: the XCD's 4 MiB hold about seven tiles per slot.  Kind A's body is a fifth
  // shorter than kind B's, so left alone it runs away and every tile is fetched from HBM twice (FETCH_SIZE 0.78 GB per
  // launch instead of 0.55).  Kind B therefore posts its tile counter (one store per body, behind the body's barrier),
  // and kind A looks at it every sync_period tiles — a load and a full vmcnt(0), at the loop head where nothing else is in
  // flight that it could disturb — and sleeps while it is more than sync_lead tiles ahead.  Kind A has the time: the launch
  // ends when kind B does.  The wait is bounded (a kind-B workgroup that is not resident yet must not hang the launch):
  // after 256 looks without progress kind A stops looking for the rest of the launch.
#ifdef SNR_PAIR_DEBUG
  bool pacing = a.sync_period > 0 && a.only_kind < 0;
#else
  bool pacing = a.sync_period > 0;
#endif
  for (int tile = 0, nt = (int)t1; tile < nt; ++tile) {
    if constexpr (PB) {
      post_word = (unsigned)tile + 1u;   // tiles done when the store lands (behind this body's barrier: EV_SYNC); 0 = none yet
    } else if (pacing) {
      if (tile % a.sync_period == a.sync_period - 1) {
        for (int look = 0;; ++look) {
          unsigned w;
          asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(0u), "s"(sync_w) : "memory");
          w = __builtin_amdgcn_readfirstlane(w);
          const int theirs = (int)w - 1;
          if (tile - theirs <= a.sync_lead) break;
          if (look == 256) { pacing = false; break; }
          __builtin_amdgcn_s_sleep(32);
        }
      }
    }
    body(std::false_type{});
  }
  // trailing DMA loads; 12+ wait states from the last MFMA to the accumulator reads
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- partial sums, bf16, one 8-byte store per four registers ----
  //   kind B: plane [256 neurons n of layer 2k][32 NM X slots]: lane (n, half gg) holds slots 8 k + 4 gg + 0..3 of every row tile
  //   kind A: plane [256 d z_{2k+1} slots][256 columns j of h_2k]: lane (slot 32 m + lane & 31, half gg) holds columns
  //           64 wave + 32 c + 8 k + 4 gg + 0..3
  constexpr int NB = PB ? 32 * NM : 256;
  __bf16* plane = (__bf16*)(J.part + H.part_off + (int64_t)split * 256 * NB);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int64_t col = 64 * wave + 32 * c + (lane & 31);
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bf16x4 h = {(__bf16)acc[c][m][4 * k], (__bf16)acc[c][m][4 * k + 1], (__bf16)acc[c][m][4 * k + 2], (__bf16)acc[c][m][4 * k + 3]};
        if constexpr (PB) *(bf16x4*)(plane + col * NB + 32 * m + 8 * k + 4 * gg) = h;
        else *(bf16x4*)(plane + (int64_t)(32 * m + (lane & 31)) * NB + 64 * wave + 32 * c + 8 * k + 4 * gg) = h;
      }
  }
  float* bp = J.part + H.bias_part_off + (int64_t)split * 256;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float bs = bsum[c] + __shfl_xor(bsum[c], 32, 64);
    // kind A: row tile 2 wave + c of D (d z_{2k+1} slots); kind B: column block 2 wave + c (neurons of layer 2k)
    if (lane < 32) bp[64 * wave + 32 * c + lane] = bs;
  }
}

__global__ __launch_bounds__(64 * kPairWaves) void mlp_wgrad_pair_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Roles.  Blocks come in groups of 16: block 16 g + c (c < 8) is kind A, block 16 g + 8 + c kind B of pair slot 8 g + c —
  // the two kinds of a slot are the blocks b and b ^ 8, same XCD.  Every block whose slot number is not below n_slots is
  // a PLAIN workgroup (the split-K jobs of mlp_wgrad.h: stream-bound work on its own CUs beside the matrix-bound slots),
  // numbered in block order.
  const int b = blockIdx.x;
  const int grp = b >> 4, kind = (b >> 3) & 1, col = b & 7;
  const int p = grp * 8 + col;
  if (p >= a.n_slots) {
    const int full = a.n_slots >> 3, rem = a.n_slots & 7;
    const int q = grp == full ? kind * (8 - rem) + (col - rem) : (rem ? 2 * (8 - rem) : 0) + (b - 16 * (full + (rem ? 1 : 0)));
    if (q >= a.n_plain_wgs) return;
#ifdef SNR_PAIR_DEBUG
    if (a.only_kind == 3 || (a.only_kind >= 0 && a.only_kind < 2)) return;   // timing experiments: 3 = no plain workgroups, 2 = only them
#endif
    int qi = 0;
    while (qi + 1 < a.n_plain && a.plain[qi + 1].j.split_begin <= q) ++qi;
    plain_job_run(a.plain[qi], q - a.plain[qi].j.split_begin, smem, wave, lane);
    return;
  }
  int ji = 0;
  while (ji + 1 < a.n_pairs && a.job[ji + 1].slot_begin <= p) ++ji;
  const PairJob& J = a.job[ji];
  const int split = p - J.slot_begin;
  if (split >= J.n_splits) return;               // (slots beyond the last job)
#ifdef SNR_PAIR_DEBUG
  if (a.only_kind == 2 || (a.only_kind >= 0 && a.only_kind < 2 && kind != a.only_kind) || (a.only_pair >= 0 && ji != a.only_pair)) return;
#endif
  // kind A is compiled once per wave (its row sums), kind B once
#define SNR_PAIR_A(KX_) \
  do { \
    if (wave == 0) pair_run<0, KX_, 0>(a, J, split, smem, wave, lane); \
    else if (wave == 1) pair_run<0, KX_, 1>(a, J, split, smem, wave, lane); \
    else if (wave == 2) pair_run<0, KX_, 2>(a, J, split, smem, wave, lane); \
    else pair_run<0, KX_, 3>(a, J, split, smem, wave, lane); \
  } while (0)
#ifdef SNR_PAIR_ONLY   // register-allocation experiments: one instance per build
  if (SNR_PAIR_ONLY == 0) pair_run<0, 16, 1>(a, J, split, smem, wave, lane);
  if (SNR_PAIR_ONLY == 1) pair_run<1, 16, 0>(a, J, split, smem, wave, lane);
  if (SNR_PAIR_ONLY == 2) pair_run<0, 4, 1>(a, J, split, smem, wave, lane);
  if (SNR_PAIR_ONLY == 3) pair_run<1, 4, 0>(a, J, split, smem, wave, lane);
  return;
#endif
  if (J.x_ks == 16) {
    if (kind == 0) SNR_PAIR_A(16);
    else pair_run<1, 16, 0>(a, J, split, smem, wave, lane);
  } else {
    if (kind == 0) SNR_PAIR_A(4);
    else pair_run<1, 4, 0>(a, J, split, smem, wave, lane);
  }
#undef SNR_PAIR_A
}

}  // namespace snr

namespace snr {

// ------------------------------------------------------------------------------------------
// host: ONE weight-gradient launch for the backward passes of one or two networks (coarse + fine) — their layer pairs on
// the pair slots, their plain jobs on the remaining workgroups — and the reduce tables of its partial sums
// ------------------------------------------------------------------------------------------
struct WgNet {            // one network's part of the launch
  const snr_mlp_config* c;
  int64_t n_samples;
  // filled by make_wgall_plan
  WgradArgs plain;        // its plain jobs (job + output tables; offsets inside its partial-sum buffer)
  WgradArgs red;          // reduce / scatter table over its layer pairs' partial planes
  int64_t part_floats;    // floats of its partial-sum buffer: plain planes | pair planes | progress words
  int64_t sync_off;       // float offset of its progress words (kSyncWords of them)
};
constexpr int kSyncWords = 256;      // progress words reserved per network (>= its pair slots: n_slots is clamped to 128)
constexpr int kMaxPairSlots = 128;
constexpr int kMaxPlainWgs = 256;   // plain workgroups of the launch, at most
struct WgAllPlan {
  PairArgs pa;            // kernel arguments (pointers still to be filled in: fill_wgall_pointers)
  int grid;               // workgroups of the launch
  int pair_job0[kMaxReduceNets], plain_job0[kMaxReduceNets];   // first pair / plain job of each network inside `pa`
};

// workgroups of the launch: pair slots and plain workgroups (tunables: SNR_PAIR_SLOTS, SNR_PLAIN_WGS).  Default: of the
// device's CUs, 5 in 32 run plain jobs (256 CUs: 108 pair slots = 216 CUs + 40 plain CUs) — swept on MI355X,
// profiles/r04_wgall_tuning.txt.
inline void wgall_shape(int* n_slots, int* n_plain) {
  const int cus = cu_count();
  int slots = tunables().pair_slots > 0 ? tunables().pair_slots : (cus * 27 / 32) / 2;
  if (slots > kMaxPairSlots) slots = kMaxPairSlots;
  if (slots < kMaxPairs) slots = kMaxPairs;
  int plain = tunables().plain_wgs > 0 ? tunables().plain_wgs : cus - 2 * slots;
  if (plain < kMaxPlain) plain = kMaxPlain;
  if (plain > kMaxPlainWgs) plain = kMaxPlainWgs;
  *n_slots = slots; *n_plain = plain;
}

// upper bound of a network's partial-sum floats in ANY launch it may take part in (alone or merged with another network):
// every split of every job is at most the largest plane of its kind
// The bound holds for every launch shape wgall_shape can return — its CLAMPS, not the current tunables or the current
// device's CU count: a workspace sized before snr_tunables_reload() raised SNR_PAIR_SLOTS / SNR_PLAIN_WGS, or with another
// device current, stays large enough (120 MB per network instead of ~70, untouched unless used).
inline int64_t wgall_part_bound() {
  const int n_slots = kMaxPairSlots, n_plain = kMaxPlainWgs;
  const int64_t pair_split = 2 * (256 * 256 + 256);        // kind A + kind B planes of a hidden layer pair, with their bias rows
  const int64_t plain_split = 160 * 288 + 160;             // [d z9 | d out] x [h7 | dir]: 5 x 9 tiles
  return (int64_t)(n_slots + kMaxPairs) * pair_split + (int64_t)(n_plain + kMaxPlain) * plain_split + kSyncWords;
}

inline int make_wgall_plan(WgNet* nets, int n_nets, WgAllPlan& Pl) {
  using B = Blob<kBF16>;
  constexpr int SPF = 2 * Prec<kBF16>::EPF;
  Pl = WgAllPlan{};
  PairArgs& A = Pl.pa;
  int n_slots, n_plain;
  wgall_shape(&n_slots, &n_plain);
  const int w0 = tunables().pair_w0 > 0 ? tunables().pair_w0 : 80;   // cost of pair 0 relative to 100 of the others (swept: profiles/r03_pair_tuning.txt)

  // ---- the jobs of every network: plain jobs (no splits yet), layer pairs ----
  int64_t pw[kMaxPairs], pcap[kMaxPairs], qw[kMaxPlain], qcap[kMaxPlain];
  int np = 0, nq = 0;
  for (int i = 0; i < n_nets; ++i) {
    WgNet& N = nets[i];
    N.plain = make_jobs<kBF16>(N.c, N.n_samples, nullptr, nullptr, true);
    Pl.pair_job0[i] = np; Pl.plain_job0[i] = nq;
    if (N.plain.n_jobs > kMaxPlain - nq) return SNR_ERR_UNSUPPORTED;
    for (int j = 0; j < N.plain.n_jobs; ++j) {
      qw[nq] = (int64_t)(N.plain.job[j].a_ks + N.plain.job[j].b_ks) * N.plain.n_tiles;   // bytes streamed
      qcap[nq++] = N.plain.n_tiles;
    }
    for (int k = 0; k < 4; ++k) {
      pw[np] = (int64_t)(k == 0 ? w0 : 100) * N.plain.n_tiles;   // MFMAs: pair 0 works on the 64-wide encoding
      pcap[np++] = N.plain.n_tiles;
    }
  }
  int psplit[kMaxPairs], qsplit[kMaxPlain];
  apportion(pw, pcap, np, n_slots, psplit);
  apportion(qw, qcap, nq, n_plain, qsplit);

  // ---- placement: plain workgroups / pair slots in job order; every network's partial planes in its own buffer ----
  int slot = 0, qwg = 0;
  for (int i = 0; i < n_nets; ++i) {
    WgNet& N = nets[i];
    const snr_mlp_config* c = N.c;
    const int vd = c->use_viewdirs;
    const ParamLayout L = make_param_layout(c->multires, c->multires_views, vd, c->out_ch, c->i_embed == -1);
    const PackTable T = make_pack_table<kBF16>(c->multires, c->multires_views, vd, c->out_ch, c->i_embed == -1);
    const ActLayout<kBF16> AL(N.n_samples, vd);
    const WsLayout<kBF16> WL(N.n_samples, vd);
    const int L_pts = c->i_embed == -1 ? 0 : c->multires;
    const int ip = L.in_pts;
    int64_t po = 0;
    for (int j = 0; j < N.plain.n_jobs; ++j) {
      PlainJob& Q = A.plain[Pl.plain_job0[i] + j];
      place_job(N.plain.job[j], qsplit[Pl.plain_job0[i] + j], qwg, po);
      Q.j = N.plain.job[j];
      Q.n_tiles = N.plain.n_tiles;
      // by rows when the job has more than four row tiles (d z5 x pe), else by columns (mlp_wgrad.h: plain_run4)
      Q.cols_mode = !(Q.j.nta == 8 && Q.j.ntb == 2);
      // (the shapes plain_job_run is instantiated for: its loader's counted wait is an immediate per tile size)
      const int kib = Q.j.a_ks + Q.j.b_ks;
      if (Q.cols_mode ? !((kib == 27 && Q.j.nta == 5 && Q.j.ntb == 9) || (kib == 9 && Q.j.nta == 1 && Q.j.ntb == 4) ||
                          (kib == 17 && Q.j.nta == 1 && Q.j.ntb == 8))
                      : kib != 20)
        return SNR_ERR_UNSUPPORTED;
    }
    WgradArgs& R = N.red;
    R = WgradArgs{};
    R.n_tiles = AL.n_tiles;
    int nj = 0, no = 0;
    const int n_fwd_entries = 8 + (vd ? 4 : 1), n_top_bwd = vd ? 3 : 1;
    for (int k = 0; k < 4; ++k) {
      PairJob& J = A.job[Pl.pair_job0[i] + k];
      const int la = 2 * k, lb = 2 * k + 1;   // rebuilt / accumulated layers: kind A rebuilds h_{la}, accumulates dW_{lb}
      J.x_ks = k == 0 ? B::KS_PE : B::KS_H;
      J.x_off = k == 0 ? AL.off_pe() : AL.off_h(la - 1);
      J.dz_off = WL.off_dz(lb);
      J.flag_off = AL.off_mask(la);
      J.n_tiles = AL.n_tiles;
      const int64_t s = psplit[Pl.pair_job0[i] + k];
      J.slot_begin = slot; J.n_splits = (int)s; slot += (int)s;
      const int NMb = J.x_ks / 2;
      J.a.w_frag = T.e[la].frag_begin;
      J.a.bias_off = bias_off_stage(la);
      J.b.w_frag = T.e[n_fwd_entries + n_top_bwd + (7 - lb)].frag_begin;
      J.b.bias_off = 0;
      J.a.part_off = po; po += s * 256 * 256;
      J.a.bias_part_off = po; po += s * 256;
      J.b.part_off = po; po += s * 256 * 32 * NMb;
      J.b.bias_part_off = po; po += s * 256;
      // reduce table: kind A planes [row = d z_{lb} slot][column j of h_{la} (true order)] -> dW_{lb}[true(row)][j], row sums -> db_{lb}
      auto rjob = [&](int nta, int ntb, int64_t part_off, int64_t bias_part_off) {
        WgradJob& Q = R.job[nj];
        Q.nta = nta; Q.ntb = ntb; Q.n_splits = (int)s; Q.split_begin = 0; Q.part_off = part_off; Q.bias_part_off = bias_part_off;
        return nj++;
      };
      auto rout = [&](int j, int rows, int a_kind, int cols, int b_kind, int Lenc, int64_t w_off, int ld, int col_off,
                      int rows_valid, int cols_valid, int64_t bias_off) {
        WgradOut& O = R.out[no++];
        O.job = j; O.row0 = 0; O.rows = rows; O.a_kind = a_kind; O.col0 = 0; O.cols = cols; O.b_kind = b_kind; O.L = Lenc;
        O.w_off = (int)w_off; O.ld = ld; O.col_off = col_off; O.row_off = 0; O.rows_valid = rows_valid; O.cols_valid = cols_valid;
        O.bias_off = (int)bias_off; O.to_scratch = 0;
      };
      const int ld_b = lb == kSkip + 1 ? kW + ip : kW, co_b = lb == kSkip + 1 ? ip : 0;
      int j = rjob(8, 8, J.a.part_off, J.a.bias_part_off);
      rout(j, kW, SRC_H, kW, SRC_NAT, 0, L.w_pts[lb], ld_b, co_b, kW, kW, L.b_pts[lb]);
      // kind B planes [neuron n of layer la (true order)][row = X slot] -> dW_{la}[n][true(row)], column sums -> db_{la}
      j = rjob(8, NMb, J.b.part_off, J.b.bias_part_off);
      if (k == 0) rout(j, kW, SRC_NAT, 32 * NMb, SRC_ENC_PTS, L_pts, L.w_pts[la], ip, 0, kW, ip, L.b_pts[la]);
      else rout(j, kW, SRC_NAT, 32 * NMb, SRC_H, 0, L.w_pts[la], kW, 0, kW, kW, L.b_pts[la]);
    }
    R.n_jobs = nj; R.n_outs = no;
    N.sync_off = po; po += kSyncWords;
    N.part_floats = po;
  }
  A.n_pairs = np; A.n_plain = nq;
  A.n_slots = slot;            // (== n_slots unless a network has fewer tiles than its share of the slots)
  A.n_plain_wgs = qwg;
  // grid: whole groups of 16 for the slots; the plain workgroups take the non-slot blocks of the last group, then new blocks
  const int full = slot >> 3, rem = slot & 7;
  const int in_group = rem ? 2 * (8 - rem) : 0;
  Pl.grid = 16 * (full + (rem ? 1 : 0)) + (qwg > in_group ? qwg - in_group : 0);
#ifdef SNR_PAIR_DEBUG
  A.only_kind = tunables().only_kind; A.only_pair = tunables().only_pair;
#endif
  A.sync_period = tunables().pair_poll; A.sync_lead = tunables().pair_lead;
  return SNR_OK;
}

// pointers of network `i` into the kernel arguments (its jobs were placed by make_wgall_plan)
inline void fill_wgall_pointers(WgAllPlan& Pl, const WgNet* nets, int i, const char* act, const char* ws, const char* blob,
                                const float* bias, float* part) {
  const WgNet& N = nets[i];
  for (int k = 0; k < 4; ++k) {
    PairJob& J = Pl.pa.job[Pl.pair_job0[i] + k];
    J.act = act; J.ws = ws; J.blob = blob; J.bias = bias; J.part = part;
    J.sync = (unsigned*)(part + N.sync_off) + (J.slot_begin - Pl.pa.job[Pl.pair_job0[i]].slot_begin);
  }
  for (int j = 0; j < N.plain.n_jobs; ++j) {
    PlainJob& Q = Pl.pa.plain[Pl.plain_job0[i] + j];
    Q.act = act; Q.ws = ws; Q.part = part;
  }
}

}  // namespace snr
