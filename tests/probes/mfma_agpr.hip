// Probe: MFMA issue rate of one wave per SIMD (4-wave workgroup, 512 registers) in the shape of the layer-pair
// weight-gradient kernel: 32 accumulating MFMAs on 16 accumulator-file tiles (inline asm, "+a") and 32 rebuild MFMAs on
// two arch-VGPR tiles per body, with F filler instructions (v_add_u32 on private registers / s_nop 0 / ds_read_b128)
// behind every MFMA.  Prints cycles per MFMA (s_memtime, 100 MHz constant clock -> converted with the measured wall time).
// build: hipcc --offload-arch=gfx950 -O3 tests/probes/mfma_agpr.hip -o mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int F>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  f32x16 acc[16];
  f32x16 R[2];
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) R[i][r] = 0.f;
  bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
  if (iters < 0) {   // random operands (a hash of lane and slot): what the clock does when the multipliers see real data
    iters = -iters;
    for (int i = 0; i < 8; ++i) {
      unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u;
      a[i] = (__bf16)(((h >> 8) & 0xffff) / 32768.f - 1.f);
      h *= 2246822519u;
      b[i] = (__bf16)(((h >> 8) & 0xffff) / 32768.f - 1.f);
    }
  }
  unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
  bf16x8 ld;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if ((s & 1) == 0 || MODE == 1) {
          if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(R[c]) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(s >> 1) * 2 % 16 + c]) : "v"(a), "v"(b));
        } else {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(R[c]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
          if (f % 4 == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
          if (f % 4 == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la));
          if (f % 4 == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
          if (f % 4 == 3) asm volatile("s_nop 0");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = x0 + x2 + (float)ld[0];
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  s += R[0][0] + R[1][0];
  if (s == 1234.5f) out[0] = s;
}

// G workgroups (one per CU): the same loop on a part of the chip shows what the clock does under load
template <int MODE, int F> void run(const char* name, float* out, int G = 256, bool random = false) {
  const int iters = random ? -2000 : 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, F><<<G, 256>>>(random ? -10 : 10, out);
  hipEventRecord(e0);
  k<MODE, F><<<G, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 / (2000 * 64.0);
  printf("%-44s %3d workgroups, fillers/MFMA %d: %.2f ns per MFMA = %.1f clk @2.4 GHz (%.0f TFLOP/s on those CUs)\n", random ? "alternating, random operands" : name, G, F,
         per * 1e9, per * 2.4e9, G * 4.0 * 32768 / per / 1e12);
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<0, 0>("acc (AGPR) / rebuild (VGPR) alternating", out);
  run<1, 0>("accumulator-file tiles only", out);
  run<2, 0>("two arch-VGPR chains only", out);
  run<0, 1>("alternating", out);
  run<0, 2>("alternating", out);
  run<0, 3>("alternating", out);
  run<0, 4>("alternating", out);
  run<0, 5>("alternating", out);
  run<0, 6>("alternating", out);
  run<1, 4>("accumulator-file tiles only", out);
  for (int G : {16, 32, 64, 128, 192, 256}) run<0, 0>("alternating", out, G);
  for (int G : {16, 32, 64, 128, 192, 256}) run<0, 4>("alternating", out, G);
  for (int G : {32, 128, 256}) run<0, 0>("", out, G, true);
  for (int G : {32, 128, 256}) run<0, 4>("", out, G, true);
  return 0;
}
