// Device-side building blocks shared by the fused MLP forward and backward kernels (gfx950).
#pragma once
#include "snr_common.h"
#include "mlp_pack.h"

namespace snr {

// ------------------------------------------------------------------------------------------
// weight-chunk pipeline
// ------------------------------------------------------------------------------------------
constexpr int kBiasLdsBytes = 12288;  // >= 2496 floats

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
#define SNR_ABLATE 0
#endif
constexpr int kBlockFrags = 16;
constexpr int kRing = 6;
constexpr int kDepth = 4;    // = kRing - 2: the slot re-filled on entering block b is that of block b-2
constexpr int kRingBytes = kRing * kBlockFrags * 1024;

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
  int pend;          // DMA pieces of the block being issued that are still to be issued
  const char* pend_src;
  char* pend_dst;

  __device__ __forceinline__ void init(char* ring_, const char* gbase_, int n_blocks_, int wave_, int lane_) {
    ring = ring_; gbase = gbase_; n_blocks = n_blocks_; wave = wave_; lane = lane_;
    issue_blk = 0; issue_slot = 0; cur_slot = kRing - 1; pend = 0;
    for (int d = 0; d < kDepth; ++d) { begin_issue(); flush(); }
  }

  // DMA of the next block of the stream, PIECES instructions per wave.  Measured on MI355X: issued
  // as a burst right behind the barrier, the 16 DMA instructions of the 4 waves serialise on the CU's
  // address path and cost every wave ~690 cycles per block (vs 512 cycles of MFMA); dripped one at a
  // time between MFMAs (issue_one) they hide in the MFMA shadow.
  __device__ __forceinline__ void begin_issue() {
    pend_src = gbase + (int64_t)issue_blk * BLOCK + wave * 1024 + lane * 16;
    pend_dst = ring + issue_slot * BLOCK + wave * 1024;
    pend = PIECES;
    issue_blk = issue_blk + 1 == n_blocks ? 0 : issue_blk + 1;
    issue_slot = issue_slot + 1 == kRing ? 0 : issue_slot + 1;
  }
  __device__ __forceinline__ void issue_one() {
    if (pend > 0) {
      __builtin_amdgcn_global_load_lds(pend_src, SNR_LDS(pend_dst), 16, 0, 0);
      pend_src += WAVES * 1024;
      pend_dst += WAVES * 1024;
      --pend;
    }
  }
  __device__ __forceinline__ void flush() {
    while (pend > 0) issue_one();
  }

  __device__ __forceinline__ void acquire() {
#if SNR_ABLATE >= 1 && SNR_ABLATE <= 3   // timing experiments only (results are garbage): no wait / barrier / DMA
    cur_slot = cur_slot + 1 == kRing ? 0 : cur_slot + 1;
    return;
#endif
    flush();   // the counted wait below assumes every older block is completely issued
    // allowed outstanding = this wave's pieces of the kDepth-1 younger blocks
    static_assert(PIECES * (kDepth - 1) == 6 || PIECES * (kDepth - 1) == 12, "add the immediate");
    if constexpr (PIECES * (kDepth - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur_slot = cur_slot + 1 == kRing ? 0 : cur_slot + 1;
    begin_issue();   // pieces follow from the MFMA loop (issue_one)
  }

  __device__ __forceinline__ void drain() { flush(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

  // NT output tiles of one stage for NJ sample tiles at once; each output tile is KA + KB MFMA groups
  // per sample tile against two register sources (tile j's sources start at sa + j*SA, sb + j*SB).
  //  * every weight fragment is read from LDS once and feeds NJ MFMAs (independent accumulators);
  //  * fragments come through a rolling window of G registers that runs ahead of the MFMAs across
  //    output-tile boundaries (the stage starts block-aligned, so block crossings are static);
  //  * init(nt) supplies the initial accumulator (bias), finish(nt, j, acc) the epilogue; the epilogues
  //    of output tile nt-1 are placed behind the first MFMAs of tile nt so that their VALU work sits
  //    in the shadow of that tile's MFMA chain (one wave per SIMD: nothing else would hide it);
  //  * pre(nt) issues tile nt's slice of the deferred global stores (the previous stage's output):
  //    spread over the tiles so that no burst of stores sits in front of the counted DMA waits.
  template <int KA, int KB, int NT, int NJ, int SA, int SB, class Init, class Finish, class Pre>
  __device__ __forceinline__ void run_tiles(const Frag* sa, const Frag* sb, Init&& init, Finish&& finish, Pre&& pre) {
    constexpr int K = KA + KB, NF = NT * K;
    constexpr int G = (P == kBF16 && NJ == 1 && WAVES == 4) ? 8 : 4;
    constexpr bool OVERLAP = WAVES == 4;   // two waves per SIMD overlap each other; no need to hold two accumulators
    Frag w[G];
    auto load = [&](int i) {
      if (i % BF == 0) acquire();
#if SNR_ABLATE >= 2   // no LDS reads either
      asm volatile("" : "+v"(w[i % G]));
#else
      w[i % G] = *(const Frag*)(ring + cur_slot * BLOCK + (i % BF) * 1024 + lane * 16);
#endif
    };
#if SNR_ABLATE >= 2
#pragma unroll
    for (int i = 0; i < G; ++i) w[i] = M::zero();
#endif
#pragma unroll
    for (int i = 0; i < (G < NF ? G : NF); ++i) load(i);
    f32x16 prev[NJ];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc[NJ];
      const f32x16 b0 = init(nt);
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = b0;
#pragma unroll
      for (int f = 0; f < K; ++f) {
        const int i = nt * K + f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[j] = M::mma(w[i % G], f < KA ? sa[j * SA + (f < KA ? f : 0)] : sb[j * SB + (f < KA ? 0 : f - KA)], acc[j]);
        if (i + G < NF) load(i + G);
        if (NJ > 1 || (f & 1)) issue_one();   // one DMA piece per ~64 cycles of MFMA
        if (f == 0) {
          pre(nt);   // this tile's slice of the deferred global stores
          if (OVERLAP && nt > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) finish(nt - 1, j, prev[j]);
          }
        }
      }
      if constexpr (OVERLAP) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) prev[j] = acc[j];
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) finish(nt, j, acc[j]);
      }
    }
    if constexpr (OVERLAP) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) finish(NT - 1, j, prev[j]);
    }
  }
};

__device__ __forceinline__ f32x16 bias_tile(const float* bias_lds, int off, int g) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *(const f32x4*)(bias_lds + off + 8 * q + 4 * g);
    acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
  }
  return acc;
}

template <int P, int I = 0>
__device__ __forceinline__ void acc_to_frags(const f32x16& acc, typename Mma<P>::Frag* dst) {
  if constexpr (I < Prec<P>::FPT) {
    dst[I] = Mma<P>::template from_acc<I>(acc);
    acc_to_frags<P, I + 1>(acc, dst);
  }
}

// sin/cos encoding of one 3-vector into KS frags (mlp_layout.h: enc_slot_feature)
template <int P, int KS>
__device__ __forceinline__ void encode(float x, float y, float z, int L, int g, typename Mma<P>::Frag* out) {
  constexpr int HP = Prec<P>::EPF / 2;
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
      Mma<P>::set(f, 2 * pp, s);
      Mma<P>::set(f, 2 * pp + 1, c);
    }
    out[q] = f;
  }
}

}  // namespace snr
