// Device-side building blocks shared by the fused MLP forward and backward kernels (gfx950).
#pragma once
#include "snr_common.h"
#include "mlp_pack.h"
#include <type_traits>

namespace snr {

// ------------------------------------------------------------------------------------------
// weight-chunk pipeline
// ------------------------------------------------------------------------------------------
constexpr int kBiasLdsBytes = 12288;  // >= 2496 floats
#ifndef SNR_IN_PREFETCH
#define SNR_IN_PREFETCH 0   // A/B builds (round 5): 1 = the forward kernel fetches a pass's inputs through LDS one pass ahead, persistent
#endif                      // grid.  Measured: nothing (profiles/r05_fwd_prologue_ab.txt) — the loads' latency was never the cost


// a*b + c with the product rounded first (HIP's __fmul_rn/__fadd_rn are plain operators and would
// be contracted into one FMA)
__device__ __forceinline__ float mul_add_unfused(float a, float b, float c) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p + c;
}

// Row of sample j inside frag block f of a saved-activation tile.  bf16: odd blocks swap the two
// 4-sample groups of every 8 so that the transposing LDS reads of the weight-gradient kernel
// (two 16-lane groups reading blocks f and f+1, 1 KiB apart) fall on disjoint bank halves.
template <int P> __device__ __forceinline__ int act_row(int j, int f) {
  return P == kBF16 ? (j ^ ((f & 1) << 2)) : j;
}

// Weight stream.  The packed weights of one pass over the network are a cyclic sequence of
// 16 KiB blocks (kBlockFrags fragments of 1 KiB); every stage starts on a block boundary
// (mlp_pack.h pads).  Blocks are DMA'd global->LDS (global_load_lds_dwordx4, 16/WAVES per wave per block)
// into a ring of kRing slots, kDepth blocks ahead of the MFMAs that consume them.  Entering a new
// block costs one counted wait + one raw s_barrier:
//   s_waitcnt vmcnt(pieces*(kDepth-1))  this wave's pieces of the block have landed (DMA loads retire in
//                                  order; stores sharing the counter can only make the wait stricter)
//   s_barrier                      ... and so have everybody else's; everybody has at least started the
//                                  previous block, so the slot of the one before it (cur-2) can be
//                                  re-filled with block cur+kDepth while reads of cur-1 may still fly.
// No __syncthreads(): its fence would drain the whole prefetch queue (vmcnt(0)) every time.
#ifndef SNR_ABLATE
#define SNR_ABLATE 0   // bit mask of timing experiments (results are garbage): see Pipe
#endif
// kBlockFrags (mlp_layout.h) fragments per block; ring slots and blocks in flight are chosen so that the bytes in
// flight stay 64 KiB: 32-fragment blocks -> 4 slots, 2 ahead (128 KiB of LDS); 16-fragment blocks (A/B builds) -> 6 slots, 4 ahead
constexpr int kRing = kBlockFrags == 16 ? 6 : 4;
constexpr int kDepth = kRing - 2;    // the slot re-filled on entering block b is that of block b-2
constexpr int kRingBytes = kRing * kBlockFrags * 1024;

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

#ifdef SNR_TIMING   // instrumentation build: s_memtime around the block waits (tests/probes/fwd_ablate.py prints it)
static __device__ unsigned long long g_snr_dbg[8];   // wait, barrier, (unused), acquires, kernel, waves
#define SNR_T() __builtin_readcyclecounter()
#endif

template <int P, int WAVES_> struct Pipe {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  static constexpr int BF = kBlockFrags, BLOCK = kBlockFrags * 1024;
  static constexpr int WAVES = WAVES_, PIECES = kBlockFrags / WAVES_;
  char* ring;
  const char* gbase;
  int n_blocks;      // blocks in the cyclic stream
  int issue_blk;     // stream index of the next block to issue
  int issue_slot;    // ring slot it will land in
  int cur_slot;      // ring slot of the block being consumed
  int wave, lane;
  uint32_t lane_off;  // lane * 16
  uint32_t ring_base; // LDS byte address of this lane's 16 B in fragment 0 of slot 0
  uint32_t cur_base;  // ... of the slot being consumed
#ifdef SNR_TIMING
  unsigned long long t_wait = 0, t_bar = 0, n_acq = 0, t_start = 0;
  unsigned long long t_phase[3] = {0, 0, 0};   // (round 5) forward: pass prologue, the skip layer's encoding, the view-direction encoding
#endif
  int pend;          // DMA pieces of the block being issued that are still to be issued
  bool dma_on = true; // (SNR_ABLATE & 64)
  const char* pend_src;
  char* pend_dst;

  __device__ __forceinline__ void init(char* ring_, const char* gbase_, int n_blocks_, int wave_, int lane_) {
    ring = ring_; gbase = gbase_; n_blocks = n_blocks_; wave = wave_; lane = lane_;
    issue_blk = 0; issue_slot = 0; cur_slot = kRing - 1; pend = 0;
#ifdef SNR_TIMING
    t_start = SNR_T();
#endif
    lane_off = (uint32_t)lane_ * 16u;
    ring_base = lds_addr(ring_) + lane_off; cur_base = ring_base;
    for (int d = 0; d < kDepth; ++d) { begin_issue(); flush(); }
  }

  // DMA of the next block of the stream, PIECES instructions per wave.  Measured on MI355X: issued
  // as a burst right behind the barrier, the 16 DMA instructions of the 4 waves serialise on the CU's
  // address path and cost every wave ~690 cycles per block (vs 512 cycles of MFMA); dripped one at a
  // time between MFMAs (issue_one) they hide in the MFMA shadow.
  __device__ __forceinline__ void begin_issue() {
    pend_src = gbase + (int64_t)issue_blk * BLOCK + wave * 1024;   // wave-uniform: SGPR base + 32-bit lane offset
    pend_dst = ring + issue_slot * BLOCK + wave * 1024;
    pend = PIECES;
    issue_blk = issue_blk + 1 == n_blocks ? 0 : issue_blk + 1;
    issue_slot = issue_slot + 1 == kRing ? 0 : issue_slot + 1;
  }
  __device__ __forceinline__ void issue_one() {
#if SNR_ABLATE & 8   // timing experiment: no DMA
    pend = 0;
#endif
#if SNR_ABLATE & 64  // timing experiment (round 5): DMA during a workgroup's FIRST pass only — later passes multiply with the real
    if (!dma_on) pend = 0;   // (wrong-stage) weights the ring then holds, not with whatever an un-written LDS contains: the
#endif                       // "no DMA" build above measures zero operands at a higher clock, not the DMA (profiles/r05_chain_ab.txt)
    if (pend > 0) {
      // One wait state in front of the DMA.  Measured on MI355X: with an (asm) ds_read issued in the cycle
      // before it, the DMA occasionally went out with a corrupt global address (memory fault with the upper
      // address half zero, ~1 launch in 2 of the fp32 kernel; never with the s_nop).  The compiler separates
      // the two when it knows both instructions; it cannot see into the asm.
      asm volatile("s_nop 0");
#ifndef SNR_DMA_AUX
#define SNR_DMA_AUX 0   // A/B builds: cache policy bits of the weight stream's LDS-DMA (1 sc0, 2 nt, 16 sc1)
#endif
      __builtin_amdgcn_global_load_lds(pend_src + (size_t)lane_off, SNR_LDS(pend_dst), 16, 0, SNR_DMA_AUX);
      pend_src += WAVES * 1024;
      pend_dst += WAVES * 1024;
      --pend;
    }
  }
  __device__ __forceinline__ void flush() {
    while (pend > 0) issue_one();
  }

  __device__ __forceinline__ void acquire() {
#if SNR_ABLATE & 2   // timing experiment (racy): no wait, no barrier
    flush(); cur_slot = cur_slot + 1 == kRing ? 0 : cur_slot + 1; begin_issue(); return;
#endif
    flush();   // the counted wait below assumes every older block is completely issued
    // allowed outstanding = this wave's pieces of the kDepth-1 younger blocks
#ifdef SNR_TIMING
    const unsigned long long t0 = SNR_T();
#endif
#if SNR_ABLATE & 16   // timing experiment (racy): tolerate 24 more outstanding operations (stores)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (kDepth - 1) + 24) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (kDepth - 1)) : "memory");
#endif
#ifdef SNR_TIMING
    const unsigned long long t1 = SNR_T();
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef SNR_TIMING
    const unsigned long long t2 = SNR_T();
    t_wait += t1 - t0; t_bar += t2 - t1; ++n_acq;
#endif
    cur_slot = cur_slot + 1 == kRing ? 0 : cur_slot + 1;
    begin_issue();   // pieces follow from the MFMA loop (issue_one)
  }

  __device__ __forceinline__ void drain() {
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SNR_TIMING
    if (lane == 0) {
      atomicAdd(&g_snr_dbg[0], t_wait); atomicAdd(&g_snr_dbg[1], t_bar); atomicAdd(&g_snr_dbg[3], n_acq);
      atomicAdd(&g_snr_dbg[4], SNR_T() - t_start); atomicAdd(&g_snr_dbg[5], 1ull);
      atomicAdd(&g_snr_dbg[2], t_phase[0]); atomicAdd(&g_snr_dbg[6], t_phase[1]); atomicAdd(&g_snr_dbg[7], t_phase[2]);
    }
#endif
  }

  // ---- LDS reads the compiler must not see ----------------------------------------------------
  // The AMDGPU backend orders every LDS access it knows about behind ALL outstanding LDS-DMA loads
  // (s_waitcnt vmcnt(0) in front of each ds_read: it cannot tell ring slots apart), which would drain
  // the prefetch queue at every fragment.  The fragment and bias reads are therefore inline asm, and
  // their completion is tracked here: LDS operations retire in issue order, so "at most N younger
  // operations outstanding" (s_waitcnt lgkmcnt(N), tied to the destination registers so that the
  // consumer cannot be scheduled above it) is exact.  Scalar loads share the counter but can only
  // make the wait stricter.
  template <int OFF, class V> static __device__ __forceinline__ void lds_read16(V& dst, uint32_t addr) {
    static_assert(sizeof(V) == 16, "one ds_read_b128");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
  }
  template <int N> static __device__ __forceinline__ void lds_wait(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N < 15 ? N : 15));
  }
  template <int N> static __device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));
  }

  // NT output tiles of one stage for NJ sample tiles at once; each output tile is KA + KB MFMA groups
  // per sample tile against two register sources (tile j's sources start at sa + j*SA, sb + j*SB).
  //  * every weight fragment is read from LDS once and feeds NJ MFMAs (independent accumulators);
  //  * fragments come through a rolling window of G registers that runs ahead of the MFMAs across
  //    output-tile boundaries (the stage starts block-aligned, so block crossings are static);
  //  * init(nt) supplies the initial accumulator: either an f32x16 value, or the LDS byte address of
  //    this lane's 16 bias floats (4 x 16 B, 32 B apart) — those are read one output tile ahead;
  //    finish(nt, j, acc) is the epilogue; the epilogues of output tile nt-1 are placed behind the
  //    first MFMAs of tile nt so that their VALU work sits in the shadow of that tile's MFMA chain when there
  //    is one wave per SIMD (WAVES == 4); with two per SIMD the partner wave covers it and the epilogue runs
  //    straight after its tile (keeping the previous accumulator measured no gain and costs 16 registers);
  //  * pre(nt) issues tile nt's slice of the deferred global stores (the previous stage's output):
  //    spread over the tiles so that no burst of stores sits in front of the counted DMA waits.
  //  * A16 / B16: the first / second source's fragments (and their weight fragments) are fp16: that segment's MFMAs are
  //    v_mfma_f32_32x32x16_f16 into the same accumulator (the encodings of the bf16 mode, mlp_layout.h: EncF16).
  template <int KA, int KB, int NT, int NJ, int SA, int SB, bool A16 = false, bool B16 = false, class Init, class Finish, class Pre>
  __device__ __forceinline__ void run_tiles(const Frag* sa, const Frag* sb, Init&& init, Finish&& finish, Pre&& pre) {
    constexpr int K = KA + KB, NF = NT * K;
#ifndef SNR_WINDOW
#define SNR_WINDOW 4   // fragments in flight per wave; measured 4 / 6 / 8: 0.214 / 0.219 / 0.241 ms inference (6, 8 spill), dgrad flat
#endif
    constexpr int G0 = (P == kBF16) ? SNR_WINDOW : 4;
    constexpr int G = G0 < NF ? G0 : NF;
    constexpr bool OVERLAP = WAVES == 4;   // two waves per SIMD overlap each other; no need to hold two accumulators
#if SNR_ABLATE & 128   // timing experiment: no bias at all (zero-initialised accumulators)
    constexpr bool BIAS = false;
#else
    constexpr bool BIAS = !std::is_same_v<decltype(init(0)), f32x16>;
#endif
    Frag w[G];
    f32x4 bias[4];
    auto load = [&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr (i % BF == 0) {
        acquire();
        cur_base = ring_base + cur_slot * BLOCK;
      }
      lds_read16<(i % BF) * 1024>(w[i % G], cur_base);
    };
    // init(0) is the LDS address of the stage's first bias tile; tile nt follows 128 bytes per tile, as an instruction
    // immediate (one address register per stage — an address per tile kept twelve VGPRs live across the whole kernel)
    auto load_bias = [&](auto NT_) {
      if constexpr (BIAS) {
        constexpr int o = 128 * decltype(NT_)::value;
        const uint32_t ba = (uint32_t)init(0);
        lds_read16<o>(bias[0], ba); lds_read16<o + 32>(bias[1], ba);
        lds_read16<o + 64>(bias[2], ba); lds_read16<o + 96>(bias[3], ba);
      }
    };
    // tile t's bias reads are issued in step (t-1)*K + FB, behind that step's MFMAs and the epilogue of
    // tile t-2 (whose registers they can take) and ahead of that step's fragment read
    constexpr int FB = K > 1 ? 1 : 0;
    load_bias(std::integral_constant<int, 0>{});
    static_for<0, G>([&](auto I_) { load(I_); });
    f32x16 prev[NJ];
    static_for<0, NT>([&](auto NT_) {
      constexpr int nt = decltype(NT_)::value;
      f32x16 acc[NJ];
      if constexpr (BIAS) {
        // operations issued behind this tile's bias reads: the fragment reads of the steps since then
        constexpr int s_t = (nt - 1) * K + FB;
        constexpr int since = nt == 0 ? G : ((nt * K < NF - G ? nt * K : NF - G) - s_t);
        lds_wait<(since > 0 ? since : 0)>(bias[0], bias[1], bias[2], bias[3]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[j][4 * q + 0] = bias[q][0]; acc[j][4 * q + 1] = bias[q][1];
            acc[j][4 * q + 2] = bias[q][2]; acc[j][4 * q + 3] = bias[q][3];
          }
      } else {
#if SNR_ABLATE & 128
        const f32x16 b0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
        const f32x16 b0 = init(nt);
#endif
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = b0;
      }
#ifndef SNR_TWO_CHAIN
#define SNR_TWO_CHAIN 0   // A/B builds (round 4): 1 = the dgrad chains (no bias, bf16) accumulate even and odd k-steps of an output
#endif                    // tile in two independent accumulators (tests/probes/mfma_feed.hip: 82 % vs 75 % of the MFMA rate)
      constexpr bool TWO = SNR_TWO_CHAIN && !BIAS && P == kBF16 && K >= 4 && NJ == 1 && !A16 && !B16;
      f32x16 acc2;   // (TWO) the odd k-steps' chain; its first MFMA takes the constant 0 as C
      static_for<0, K>([&](auto F_) {
        constexpr int f = decltype(F_)::value;
        constexpr int i = nt * K + f;
        // younger LDS operations that may still be in flight: the rest of the window, plus the bias reads
        // issued since fragment i was
        constexpr int ahead = (G - 1 < NF - 1 - i) ? G - 1 : NF - 1 - i;
        constexpr int nb = !BIAS ? 0 : [] {
          int n = 0;
          for (int t = 1; t < NT; ++t)
            if ((t - 1) * K + FB >= i - G + 1 && (t - 1) * K + FB <= i - 1) ++n;
          return 4 * n;
        }();
        lds_wait<ahead + nb>(w[i % G]);
        if constexpr (TWO && (f & 1)) {
          const f32x16 z16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          const Frag& src = f < KA ? sa[f] : sb[f - KA];
          acc2 = M::mma(w[i % G], src, f == 1 ? z16 : acc2);
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if constexpr (f < KA) {
              if constexpr (A16) acc[j] = M::mma_f16(w[i % G], sa[j * SA + f], acc[j]);
              else acc[j] = M::mma(w[i % G], sa[j * SA + f], acc[j]);
            } else {
              if constexpr (B16) acc[j] = M::mma_f16(w[i % G], sb[j * SB + (f - KA)], acc[j]);
              else acc[j] = M::mma(w[i % G], sb[j * SB + (f - KA)], acc[j]);
            }
          }
        }
        if constexpr (f == 0) {
#if !(SNR_ABLATE & 4)   // timing experiment: no activation stores
          pre(nt);   // this tile's slice of the deferred global stores
#endif
          if constexpr (OVERLAP && nt > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) finish(nt - 1, j, prev[j]);
          }
        }
        if constexpr (f == FB && nt + 1 < NT) load_bias(std::integral_constant<int, nt + 1>{});
        if constexpr (i + G < NF) load(std::integral_constant<int, i + G>{});
#ifndef SNR_DMA_SPREAD
#define SNR_DMA_SPREAD 0   // A/B (round 5): 1 = this wave's PIECES pieces of a block evenly over the block's fragments (one per
#endif                     // BF / PIECES MFMAs) instead of one behind every other MFMA right after the block entry
        if constexpr (SNR_DMA_SPREAD && NJ == 1) {
          if constexpr (i % (BF / PIECES) == BF / PIECES - 1) issue_one();
        } else {
          if (NJ > 1 || (f & 1)) issue_one();   // one DMA piece per ~64 cycles of MFMA
        }
      });
      if constexpr (TWO) acc[0] = acc[0] + acc2;
      if constexpr (OVERLAP) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) prev[j] = acc[j];
      } else {
#if SNR_ABLATE & 1   // timing experiment: no epilogue
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[j]));
#else
#pragma unroll
        for (int j = 0; j < NJ; ++j) finish(nt, j, acc[j]);
#endif
      }
    });
    if constexpr (OVERLAP) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) finish(NT - 1, j, prev[j]);
    }
  }
};

// LDS byte address of this lane's slice of a 32-float bias tile (lane half g takes floats 4g..4g+3 of every 8)
__device__ __forceinline__ int bias_tile_addr(const float* bias_lds, int off, int g) {
  return (int)lds_addr(bias_lds + off + 4 * g);
}

template <int P, int I = 0>
__device__ __forceinline__ void acc_to_frags(const f32x16& acc, typename Mma<P>::Frag* dst) {
  if constexpr (I < Prec<P>::FPT) {
    dst[I] = Mma<P>::template from_acc<I>(acc);
    acc_to_frags<P, I + 1>(acc, dst);
  }
}

// ---- bf16 epilogues on packed pairs -----------------------------------------------------------
// The chained kernels issue from one or two waves per SIMD, one instruction per wave per 4 cycles: every
// epilogue instruction competes with the MFMAs for issue slots (measured: 8.5 VALU per MFMA before
// this, i.e. issue-bound).  So the bf16 epilogues work on the packed words after conversion:
//   relu      = v_pk_max_i16(x, 0)          (a negative bf16 is a negative int16; -0 -> 0)
//   flag      = v_pk_min_u16(x, 1)          (1 where the relu output is non-zero), gathered with v_lshl_or
//   dgrad     = v_pk_mul_lo_u16(x, flag)    (bit pattern times 0 / 1)
// Flag layout of one 32-neuron output tile inside a 32-bit word (two tiles per word, sh = 8 * (nt & 1)):
// packed word D = 0..7 of the tile (C registers 2D, 2D+1) -> bits D + sh and 16 + D + sh.
// The packed instructions are inline asm: written with vector types the optimiser turns min(x, 1) into a
// per-element compare + select and scalarises the whole epilogue (5 instructions per element, measured).
// The conversion itself stays with the compiler — it is the first reader of the MFMA result and the
// compiler has to see that to insert the wait states between an MFMA and a VALU read of its output.
__device__ __forceinline__ unsigned pk_relu_bf16(unsigned x) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ unsigned pk_nonzero_u16(unsigned x) {   // 1 per non-zero half
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(0x00010001u));
  return r;
}
__device__ __forceinline__ unsigned pk_mul_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

template <bool RELU, bool FLAGS>
__device__ __forceinline__ void finish_fwd_bf16(const f32x16& acc, bf16x8* dst, unsigned& word, int sh) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = acc[8 * h + e];
    u32x4 wv = __builtin_bit_cast(u32x4, __builtin_convertvector(t, bf16x8));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      unsigned x = wv[d];
      if constexpr (RELU) x = pk_relu_bf16(x);
      if constexpr (FLAGS) word |= pk_nonzero_u16(x) << (4 * h + d + sh);
      wv[d] = x;
    }
    dst[h] = __builtin_bit_cast(bf16x8, wv);
  }
}

__device__ __forceinline__ void finish_dgrad_bf16(const f32x16& acc, bf16x8* dst, bool use_mask, unsigned word, int sh) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = acc[8 * h + e];
    u32x4 wv = __builtin_bit_cast(u32x4, __builtin_convertvector(t, bf16x8));
    if (use_mask) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned x = wv[d];   // (bit_cast straight from a vector element reads element 0)
        wv[d] = pk_mul_u16(x, (word >> (4 * h + d + sh)) & 0x00010001u);
      }
    }
    dst[h] = __builtin_bit_cast(bf16x8, wv);
  }
}

// ---- saved-activation / d z stores ------------------------------------------------------------
// Streamed once and read by a later kernel -> non-temporal.  Inline asm with an SGPR base + 32-bit lane
// offset (the compiler kept a 64-bit vector pointer per store stream and advanced it with two VALU
// instructions per store), and — as important — invisible to the compiler's waitcnt bookkeeping: with
// loads and stores pending on the one gfx9 vmcnt it assumes out-of-order completion and drains the
// counter (vmcnt(0): every outstanding store AND the whole DMA queue) in front of each use of a loaded
// value.  Hidden stores only make its counted waits stricter.  The data registers are read when the
// store issues; the listing check covers the rest.
template <int OFF, class V>
__device__ __forceinline__ void store16_stream(const char* sbase, uint32_t voff, const V& v) {
  static_assert(sizeof(V) == 16 && OFF >= 0 && OFF < 4096, "one global_store_dwordx4, 12-bit offset");
  // s_nop: a store of more than 8 bytes still reads its data one cycle after it issues, and the compiler
  // (which would insert this wait state for a store it knows) may overwrite the registers right away
  // leading s_nop 4: the base may have been produced by a VALU instruction (v_readlane of a spilled SGPR,
  // v_readfirstlane), which a VMEM instruction may read only 5 wait states later; again the compiler
  // guarantees that only for memory instructions it knows.
#ifndef SNR_STORE_POLICY
#define SNR_STORE_POLICY "nt"   // A/B builds: "", "sc1", "sc0 sc1", "nt sc1" ...
#endif
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3 " SNR_STORE_POLICY "\n\ts_nop 0" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF));
}

// frags [n*nt/NT, n*(nt+1)/NT) of an n-frag section tile (layout [frag][32 samples][32 B], odd frags
// with the act_row swizzle): output tile nt's share of a stage's deferred stores.
// PAR = -1: every fragment; 0 / 1 (round 5): only the even / odd fragments, shared out over the NT tiles in the same way — a saved
// layer's even fragments leave from the stage that PRODUCES them (fragment 2 nt right behind output tile nt's epilogue), its
// odd fragments from the stage that consumes it, so that every stage issues one store per output tile instead of two per
// tile in every other stage (SNR_STORE_SPLIT; the store path of a CU takes ~57 cycles per 1 KiB store: 72 % busy during a
// storing stage, idle during the next).
#ifndef SNR_STORE_SPLIT
#define SNR_STORE_SPLIT 1
#endif
// CVT16: the fragments are fp16 (an encoding of the bf16 mode) and are saved re-rounded to bf16 — the conversion sits inside the
// store's own branch, so a fragment is converted by the output tile that stores it and no second copy of the encoding is live.
template <int P, int N, int NT, int PAR = -1, bool CVT16 = false, class Frag>
__device__ __forceinline__ void store_tile_slice(const char* tile_base, const Frag* src, int nt, uint32_t lane_even, uint32_t lane_odd) {
  constexpr int M = PAR < 0 ? N : (PAR == 0 ? (N + 1) / 2 : N / 2);   // fragments this call may store
  static_for<0, M>([&](auto I_) {
    constexpr int idx = decltype(I_)::value;
    constexpr int f = PAR < 0 ? idx : 2 * idx + PAR;
    // (spread over the tiles 0.288 ms dgrad / 0.358 forward; whole section at once 0.300 / 0.371; two halves 0.295 / 0.364)
#if defined(SNR_STORE_BURST) && SNR_STORE_BURST == 1     // A/B: the whole section behind the first output tile
    if (nt == 0)
#elif defined(SNR_STORE_BURST) && SNR_STORE_BURST == 2   // A/B: in two halves
    if ((nt == 0 && idx < M / 2) || (nt == NT / 2 && idx >= M / 2))
#else
    if (idx >= M * nt / NT && idx < M * (nt + 1) / NT)
#endif
    {
      if constexpr (CVT16) store16_stream<(f % 4) * 1024>(tile_base + (f / 4) * 4096, (f & 1) ? lane_odd : lane_even, Mma<P>::f16_to_bf16(src[f]));
      else store16_stream<(f % 4) * 1024>(tile_base + (f / 4) * 4096, (f & 1) ? lane_odd : lane_even, src[f]);
    }
  });
}

// sin/cos encoding of one 3-vector into KS frags (mlp_layout.h: enc_slot_feature)
// F16: the fragment holds fp16 values (bf16 mode, mlp_layout.h: EncF16) instead of the mode's own element type
template <int P, bool F16> __device__ __forceinline__ void enc_set(typename Mma<P>::Frag& f, int e, float x) {
  if constexpr (F16) Mma<P>::set_f16(f, e, x);
  else Mma<P>::set(f, e, x);
}
template <int P, int KS, bool F16 = false>
__device__ __forceinline__ void encode(float x, float y, float z, int L, int g, typename Mma<P>::Frag* out) {
  constexpr int HP = Prec<P>::EPF / 2;
  // Everything below that depends only on (g, L) — slot indices, axis selectors, 2^k — is invariant across the tile
  // loop, and the compiler used to hoist all of it out of the kernel's main loop: ~40 values and mask pairs held across
  // the MFMA chains, i.e. 220-280 SGPR spills and 12-16 VGPR spills to scratch in the chain kernels.  Recomputing them
  // per call costs a few dozen VALU instructions per 32-sample tile; the opaque copy of g pins them here.
  asm volatile("" : "+v"(g));
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    typename Mma<P>::Frag f = Mma<P>::zero();
#pragma unroll
    for (int pp = 0; pp < HP; ++pp) {
      const int p = (2 * q + g) * HP + pp;
      float s = 0.f, c = 0.f;
      if (p < 3 * L) {
        const int k = p / 3, ax = p - 3 * k;
        const float v = (ax == 0 ? x : (ax == 1 ? y : z)) * __builtin_ldexpf(1.0f, k);  // x * 2^k, exact
        if constexpr (P == kBF16) {
          // the result is rounded to bf16 (2^-9) right away: hardware sin/cos on the fractional number
          // of revolutions (abs error ~1e-4 rad at |v| ~ 5000) instead of the ~80-instruction libm path
          float rev = v * 0.15915494309189535f;
          rev = rev - __builtin_floorf(rev);
          s = __builtin_amdgcn_sinf(rev);
          c = __builtin_amdgcn_cosf(rev);
        } else {
          sincosf(v, &s, &c);
        }
      } else if (p == 3 * L) {
        s = x; c = y;
      } else if (p == 3 * L + 1) {
        s = z;
      }
      enc_set<P, F16>(f, 2 * pp, s);
      enc_set<P, F16>(f, 2 * pp + 1, c);
    }
    out[q] = f;
  }
}

// The same encoding with the number of frequencies known at compile time (bf16 only; the reference's configurations use
// multires = 10 / multires_views = 4: run_nerf.py:808-811) — round 5.  The generic version above derives, PER PAIR and at run
// time, which frequency and axis the lane's half g owns (p / 3, p % 3, 2^k by v_ldexp, the p < 3 L tests): ~28 instructions a
// pair, 450 per positional encoding, three encodings a pass — and the two waves of a SIMD run them at the same moment (the
// waves of a workgroup move through the weight stream in lockstep), so the matrix pipe idles meanwhile.  Here the two
// candidates of every pair (g = 0 / 1) are compile-time constants and the lane picks with one v_cndmask each: input
// coordinate, scale 2^k / 2 pi (ONE multiplication: scaling by a power of two is exact, so fl(x 2^k c) is the same number the
// generic version rounds to — the results are bit-identical, tests/test_gpu_kernels.py), then floor / sub / sin / cos.
template <int P, int KS, int LT, bool F16 = false>
__device__ __forceinline__ void encode_static(float x, float y, float z, int g, typename Mma<P>::Frag* out) {
  static_assert(P == kBF16, "hardware sin / cos: bf16 mode only");
  constexpr int HP = Prec<P>::EPF / 2;
  asm volatile("" : "+v"(g));   // (as above: nothing here may be hoisted out of the kernel's main loop and held across it)
  const bool hi = g != 0;
  static_for<0, KS>([&](auto Q_) {
    constexpr int q = decltype(Q_)::value;
    typename Mma<P>::Frag f = Mma<P>::zero();
    static_for<0, HP>([&](auto PP_) {
      constexpr int pp = decltype(PP_)::value;
      constexpr int p0 = (2 * q) * HP + pp, p1 = (2 * q + 1) * HP + pp;
      // 0: a (sin, cos) pair; 1: (x, y); 2: (z, pad); 3: padding
      constexpr int kind0 = p0 < 3 * LT ? 0 : (p0 == 3 * LT ? 1 : (p0 == 3 * LT + 1 ? 2 : 3));
      constexpr int kind1 = p1 < 3 * LT ? 0 : (p1 == 3 * LT ? 1 : (p1 == 3 * LT + 1 ? 2 : 3));
      float s = 0.f, c = 0.f;
      if constexpr (kind0 == 0 || kind1 == 0) {
        // (a lane whose own candidate is not a sin / cos pair computes the other half's: finite, and replaced below)
        constexpr int k0 = kind0 == 0 ? p0 / 3 : p1 / 3, ax0 = kind0 == 0 ? p0 % 3 : p1 % 3;
        constexpr int k1 = kind1 == 0 ? p1 / 3 : k0, ax1 = kind1 == 0 ? p1 % 3 : ax0;
        constexpr float C0 = 0.15915494309189535f * (float)(1 << k0), C1 = 0.15915494309189535f * (float)(1 << k1);
        const float in0 = ax0 == 0 ? x : (ax0 == 1 ? y : z), in1 = ax1 == 0 ? x : (ax1 == 1 ? y : z);
        const float in = ax0 == ax1 ? in0 : (hi ? in1 : in0);
        const float sc = k0 == k1 ? C0 : (hi ? C1 : C0);
        float rev = in * sc;
        rev = rev - __builtin_floorf(rev);
        s = __builtin_amdgcn_sinf(rev);
        c = __builtin_amdgcn_cosf(rev);
      }
      if constexpr (kind0 != 0) {
        const float s0 = kind0 == 1 ? x : (kind0 == 2 ? z : 0.f), c0 = kind0 == 1 ? y : 0.f;
        s = hi ? s : s0; c = hi ? c : c0;
      }
      if constexpr (kind1 != 0) {
        const float s1 = kind1 == 1 ? x : (kind1 == 2 ? z : 0.f), c1 = kind1 == 1 ? y : 0.f;
        s = hi ? s1 : s; c = hi ? c1 : c;
      }
      enc_set<P, F16>(f, 2 * pp, s);
      enc_set<P, F16>(f, 2 * pp + 1, c);
    });
    out[q] = f;
  });
}
// bf16 with the reference's frequency counts takes the static version (a wave-uniform branch); everything else the generic one
// (Lsel: L, or -1 to force the generic version — SNR_ENC_GENERIC, the bit-identity test)
template <int P, int KS, int LT, bool F16 = false>
__device__ __forceinline__ void encode_auto(float x, float y, float z, int Lsel, int L, int g, typename Mma<P>::Frag* out) {
#ifndef SNR_ENC_STATIC
#define SNR_ENC_STATIC 1   // A/B builds: 0 = the generic encoding everywhere
#endif
  if constexpr (P == kBF16 && SNR_ENC_STATIC) {
    if (Lsel == LT) { encode_static<P, KS, LT, F16>(x, y, z, g, out); return; }
  }
  encode<P, KS, F16>(x, y, z, L, g, out);
}

}  // namespace snr
