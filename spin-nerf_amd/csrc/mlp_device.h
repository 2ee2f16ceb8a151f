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

template <int P> struct Pipe {
  using M = Mma<P>;
  using Frag = typename M::Frag;
  static constexpr int SLOT = Blob<P>::MAX_CHUNK_FRAGS * 1024;
  char* slots;
  const char* gbase;
  const char* gcur;
  int slot, wave, lane;

  __device__ __forceinline__ void issue_into(int s, int nfrags) {
    char* dst = slots + s * SLOT;
    for (int p = wave; p < nfrags; p += 4)
      __builtin_amdgcn_global_load_lds(gcur + p * 1024 + lane * 16, SNR_LDS(dst + p * 1024), 16, 0, 0);
    gcur += nfrags * 1024;
  }

  // One chunk: wait for it (barrier also retires everyone's reads of the other slot), start the
  // DMA of the next chunk into the other slot, run `pre` (deferred global stores), then
  // KA + KB MFMA groups against the two register sources.
  template <int KA, int KB, class Pre>
  __device__ __forceinline__ f32x16 step(f32x16 acc, const Frag* sa, const Frag* sb, int next_frags, bool wrap,
                                         Pre&& pre) {
    __syncthreads();
    if (wrap) gcur = gbase;
    if (next_frags > 0) issue_into(slot ^ 1, next_frags);
    pre();
    const char* sp = slots + slot * SLOT + lane * 16;
#pragma unroll
    for (int f = 0; f < KA; ++f) acc = M::mma(*(const Frag*)(sp + f * 1024), sa[f], acc);
#pragma unroll
    for (int f = 0; f < KB; ++f) acc = M::mma(*(const Frag*)(sp + (KA + f) * 1024), sb[f], acc);
    slot ^= 1;
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
