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
// (mlp_pack.h pads).  Blocks are DMA'd global->LDS (global_load_lds_dwordx4, 4 per wave per block)
// into a ring of kRing slots, kDepth blocks ahead of the MFMAs that consume them.  Entering a new
// block costs one counted wait + one raw s_barrier:
//   s_waitcnt vmcnt(4*(kDepth-1))  this wave's pieces of the block have landed (DMA loads retire in
//                                  order; stores sharing the counter can only make the wait stricter)
//   s_barrier                      ... and so have everybody else's; everybody has at least started the
//                                  previous block, so the slot of the one before it (cur-2) can be
//                                  re-filled with block cur+kDepth while reads of cur-1 may still fly.
// No __syncthreads(): its fence would drain the whole prefetch queue (vmcnt(0)) every time.
constexpr int kBlockFrags = 16;
constexpr int kRing = 8;
constexpr int kDepth = 6;   // = kRing - 2: the slot re-filled on entering block b is that of block b-2
constexpr int kRingBytes = kRing * kBlockFrags * 1024;

template <int P> struct Pipe {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  static constexpr int BF = kBlockFrags, BLOCK = kBlockFrags * 1024;
  char* ring;
  const char* gbase;
  int n_blocks;      // blocks in the cyclic stream
  int issue_blk;     // stream index of the next block to issue
  int issue_slot;    // ring slot it will land in
  int cur_slot;      // ring slot of the block being consumed
  int wave, lane;

  __device__ __forceinline__ void init(char* ring_, const char* gbase_, int n_blocks_, int wave_, int lane_) {
    ring = ring_; gbase = gbase_; n_blocks = n_blocks_; wave = wave_; lane = lane_;
    issue_blk = 0; issue_slot = 0; cur_slot = kRing - 1;
    for (int d = 0; d < kDepth; ++d) issue();
  }

  __device__ __forceinline__ void issue() {
    const char* src = gbase + (int64_t)issue_blk * BLOCK + wave * 1024 + lane * 16;
    char* dst = ring + issue_slot * BLOCK + wave * 1024;
#pragma unroll
    for (int p = 0; p < BF / 4; ++p)
      __builtin_amdgcn_global_load_lds(src + p * 4096, SNR_LDS(dst + p * 4096), 16, 0, 0);
    issue_blk = issue_blk + 1 == n_blocks ? 0 : issue_blk + 1;
    issue_slot = issue_slot + 1 == kRing ? 0 : issue_slot + 1;
  }

  __device__ __forceinline__ void acquire() {
    static_assert(kDepth == 6 && kBlockFrags == 16, "update the vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // 4 * (kDepth - 1)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur_slot = cur_slot + 1 == kRing ? 0 : cur_slot + 1;
    issue();
  }

  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

  // One output tile: KA + KB MFMA groups against the two register sources; `f0` is the position of
  // the tile's first fragment inside its (block-aligned) stage, so block crossings are static.
  // Weight fragments are read from LDS kAhead fragments ahead of the MFMA that consumes them.
  // `pre` (deferred global stores) runs once, before the tile's first fragment.
  template <int KA, int KB, class Pre>
  __device__ __forceinline__ f32x16 step(f32x16 acc, const Frag* sa, const Frag* sb, int f0, Pre&& pre) {
    constexpr int K = KA + KB;
    constexpr int G = (P == kBF16) ? 8 : 4;
    Frag w[K];
    auto load = [&](int f) {
      if ((f0 + f) % BF == 0) acquire();
      w[f] = *(const Frag*)(ring + cur_slot * BLOCK + ((f0 + f) % BF) * 1024 + lane * 16);
    };
    pre();
#pragma unroll
    for (int f = 0; f < (G < K ? G : K); ++f) load(f);
#pragma unroll
    for (int f = 0; f < K; ++f) {
      if (f + G < K) load(f + G);
      acc = M::mma(w[f], f < KA ? sa[f < KA ? f : 0] : sb[f < KA ? 0 : f - KA], acc);
    }
    return acc;
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
        sincosf(v, &s, &c);
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
