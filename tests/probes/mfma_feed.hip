// Probe: what the MFMA pipe sustains in the chain kernels' inner pattern on MI355X — v_mfma_f32_32x32x16_bf16 with the A
// operand (weights) read from LDS per instruction (ds_read_b128) and the B operand (activations) in registers.
//   mode 0: 8 waves/CU, one accumulator chain per wave (16 dependent MFMAs per 32-row tile), A from LDS   [the kernels]
//   mode 1: as 0, A from registers (no LDS)                                                                [issue ceiling]
//   mode 2: 8 waves/CU, two independent accumulator chains per wave, A from LDS (no reuse)
//   mode 3: 4 waves/CU, two sample tiles per wave sharing every A fragment (NJ = 2), A from LDS
//   mode 4: 4 waves/CU, one chain per wave, A from LDS
// build: hipcc --offload-arch=gfx950 -O3 tests/probes/mfma_feed.hip -o mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include <utility>
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS reads the compiler cannot hoist or batch (the loop is iteration-invariant): asm read + counted wait, as in the kernels
template <int OFF> __device__ __forceinline__ void lds_read16(bf16x8& dst, uint32_t base) {   // base covers 64 KiB
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(OFF < 65536 ? base : base + 65536), "n"(OFF & 65535));
}
template <int N> __device__ __forceinline__ void lds_wait(bf16x8& x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait2(bf16x8& x, bf16x8& y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}

template <int MODE>
__global__ __launch_bounds__((MODE >= 3 ? 256 : 512)) void k(const bf16x8* __restrict__ wsrc, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  bf16x8* lds = (bf16x8*)lds_raw;                       // 128 fragments of 1 KiB = one 256x256 layer
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 128 * 64; i += blockDim.x) lds[i] = wsrc[i];
  __syncthreads();
  bf16x8 b[16], b2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    b[q] = wsrc[(q * 64 + lane) & 8191];
    b2[q] = wsrc[((q + 16) * 64 + lane) & 8191];
  }
  f32x16 tot = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int G = 4;                                  // fragments in flight ahead of the MFMAs (the kernels' window)
  constexpr int NT = (MODE == 2) ? 4 : 8;               // mode 2: two row tiles at a time
  bf16x8 w[G], w2[G];
  const uint32_t base = (uint32_t)(uintptr_t)lds_raw + lane * 16;   // LDS byte address (shared pointers are 32-bit offsets)
#pragma unroll
  for (int i = 0; i < G; ++i) {
    if (MODE != 1) lds_read16<0>(w[i], base + i * 1024);
    if (MODE == 2) lds_read16<0>(w2[i], base + (64 + i) * 1024);
  }
  for (int it = 0; it < iters; ++it) {
    static_for<0, NT>([&](auto NT_) {
      constexpr int nt = decltype(NT_)::value;
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;
      static_for<0, 16>([&](auto F_) {
        constexpr int f = decltype(F_)::value;
        constexpr int i = nt * 16 + f;
        // the read stream is cyclic (it wraps into the next iteration), so G - 1 younger reads are always in flight
        if constexpr (MODE == 2) lds_wait2<2 * (G - 1)>(w[i % G], w2[i % G]);
        else if constexpr (MODE != 1) lds_wait<G - 1>(w[i % G]);
        const bf16x8 a = MODE == 1 ? b2[f] : w[i % G];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[f], acc, 0, 0, 0);
        if constexpr (MODE == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[i % G], b[f], acc2, 0, 0, 0);
        if constexpr (MODE == 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2[f], acc2, 0, 0, 0);
        if constexpr (MODE != 1) {
          lds_read16<((i + G) % (NT * 16)) * 1024>(w[i % G], base);
          if constexpr (MODE == 2) lds_read16<(64 + (i + G) % (NT * 16)) * 1024>(w2[i % G], base);
        }
      });
      tot += acc + acc2;
      if constexpr (MODE == 1) asm volatile("" : "+v"(b2[nt]));   // keep the register operand from being hoisted into a constant
    });
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += tot[e];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE> void run(const bf16x8* w, float* out, const char* what) {
  const int waves = (MODE == 3 || MODE == 4) ? 4 : 8, iters = 400;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<MODE><<<256, 64 * waves, 131072>>>(w, out, 10);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  k<MODE><<<256, 64 * waves, 131072>>>(w, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double mfma_per_wave = (double)iters * 8 * 16 * (MODE == 3 ? 2 : 1);
  const double flops = mfma_per_wave * waves * 256 * 32768.0;
  printf("mode %d  %-70s %8.3f ms  %7.1f TFLOP/s  (%4.1f %% of 2500)\n", MODE, what, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}

int main() {
  bf16x8* w; hipMalloc(&w, 8192 * 16); hipMemset(w, 0, 8192 * 16);
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>(w, out, "8 waves, 1 chain/wave, A from LDS (the chain kernels' shape)");
  run<1>(w, out, "8 waves, 1 chain/wave, A from registers");
  run<2>(w, out, "8 waves, 2 independent chains/wave, A from LDS");
  run<3>(w, out, "4 waves, 2 sample tiles/wave sharing A (NJ=2), A from LDS");
  run<4>(w, out, "4 waves, 1 chain/wave, A from LDS");
  return 0;
}
