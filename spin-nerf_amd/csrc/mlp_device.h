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
constexpr int kBlockFrags = 16;
constexpr int kRing = 6;
constexpr int kDepth = 4;    // = kRing - 2: the slot re-filled on entering block b is that of block b-2
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
  int pend;          // DMA pieces of the block being issued that are still to be issued
  const char* pend_src;
  char* pend_dst;

  __device__ __forceinline__ void init(char* ring_, const char* gbase_, int n_blocks_, int wave_, int lane_) {
    ring = ring_; gbase = gbase_; n_blocks = n_blocks_; wave = wave_; lane = lane_;
    issue_blk = 0; issue_slot = 0; cur_slot = kRing - 1; pend = 0;
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
    if (pend > 0) {
      __builtin_amdgcn_global_load_lds(pend_src + (size_t)lane_off, SNR_LDS(pend_dst), 16, 0, 0);
      pend_src += WAVES * 1024;
      pend_dst += WAVES * 1024;
      --pend;
    }
  }
  __device__ __forceinline__ void flush() {
    while (pend > 0) issue_one();
  }

  __device__ __forceinline__ void acquire() {
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
  //    first MFMAs of tile nt so that their VALU work sits in the shadow of that tile's MFMA chain
  //    (one wave per SIMD: nothing else would hide it);
  //  * pre(nt) issues tile nt's slice of the deferred global stores (the previous stage's output):
  //    spread over the tiles so that no burst of stores sits in front of the counted DMA waits.
  template <int KA, int KB, int NT, int NJ, int SA, int SB, class Init, class Finish, class Pre>
  __device__ __forceinline__ void run_tiles(const Frag* sa, const Frag* sb, Init&& init, Finish&& finish, Pre&& pre) {
    constexpr int K = KA + KB, NF = NT * K;
    constexpr int G0 = (P == kBF16 && NJ == 1 && WAVES == 4) ? 8 : 4;
    constexpr int G = G0 < NF ? G0 : NF;
    constexpr bool OVERLAP = WAVES == 4;   // two waves per SIMD overlap each other; no need to hold two accumulators
    constexpr bool BIAS = !std::is_same_v<decltype(init(0)), f32x16>;
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
    auto load_bias = [&](int nt) {
      if constexpr (BIAS) {
        const uint32_t ba = (uint32_t)init(nt);
        lds_read16<0>(bias[0], ba); lds_read16<32>(bias[1], ba);
        lds_read16<64>(bias[2], ba); lds_read16<96>(bias[3], ba);
      }
    };
    // tile t's bias reads are issued in step (t-1)*K + FB, behind that step's MFMAs and the epilogue of
    // tile t-2 (whose registers they can take) and ahead of that step's fragment read
    constexpr int FB = K > 1 ? 1 : 0;
    load_bias(0);
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
        const f32x16 b0 = init(nt);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = b0;
      }
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
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (f < KA) acc[j] = M::mma(w[i % G], sa[j * SA + f], acc[j]);
          else acc[j] = M::mma(w[i % G], sb[j * SB + (f - KA)], acc[j]);
        }
        if constexpr (f == 0) {
          pre(nt);   // this tile's slice of the deferred global stores
          if constexpr (OVERLAP && nt > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) finish(nt - 1, j, prev[j]);
          }
        }
        if constexpr (f == FB && nt + 1 < NT) load_bias(nt + 1);
        if constexpr (i + G < NF) load(std::integral_constant<int, i + G>{});
        if (NJ > 1 || (f & 1)) issue_one();   // one DMA piece per ~64 cycles of MFMA
      });
      if constexpr (OVERLAP) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) prev[j] = acc[j];
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) finish(nt, j, acc[j]);
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
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

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
      if constexpr (RELU) x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), s16x2{0, 0}));
      if constexpr (FLAGS) {
        const unsigned fl = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), u16x2{1, 1}));
        word |= fl << (4 * h + d + sh);
      }
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
        const unsigned e = (word >> (4 * h + d + sh)) & 0x00010001u;
        const unsigned x = wv[d];   // (bit_cast straight from a vector element reads element 0)
        wv[d] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, e)));
      }
    }
    dst[h] = __builtin_bit_cast(bf16x8, wv);
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
