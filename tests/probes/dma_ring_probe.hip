// Probe: how fast can one workgroup (4 waves) stream 16 KiB blocks L2 -> LDS through the Pipe ring
// (counted vmcnt + raw barrier per block), with `reads` ds_read_b128 per wave per block and no MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../spin-nerf_amd/csrc/mlp_device.h"
using namespace snr;
__global__ __launch_bounds__(256) void probe(const char* blob, int n_blocks, int iters, int reads, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Pipe<kBF16> pipe;
  pipe.init(smem, blob, n_blocks, wave, lane);
  float s = 0.f;
  for (int i = 0; i < iters; ++i) {
    pipe.acquire();
    for (int r = 0; r < reads; ++r) {
      const bf16x8 w = *(const bf16x8*)(pipe.ring + pipe.cur_slot * Pipe<kBF16>::BLOCK + r * 1024 + lane * 16);
      s += (float)w[0];
    }
  }
  pipe.drain();
  if (s == 12345.f) out[0] = s;
}
int main() {
  const int n_blocks = 75;
  char* blob; float* out;
  hipMalloc(&blob, n_blocks * 16384); hipMemset(blob, 0, n_blocks * 16384); hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kRingBytes);
  for (int reads : {0, 4, 16}) {
    for (int grid : {256, 1024}) {
      const int iters = 75 * 8 * 256 / grid * (grid / 256);   // same per-WG work: 600 blocks
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      probe<<<grid, 256, kRingBytes>>>(blob, n_blocks, 600, reads, out);
      hipEventRecord(a);
      probe<<<grid, 256, kRingBytes>>>(blob, n_blocks, 600, reads, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double per_cu_blocks = 600.0 * grid / 256;
      printf("reads/wave/block=%2d grid=%4d: %.3f ms, %.1f ns per block per CU, %.1f GB/s per CU, %.2f TB/s chip\n", reads, grid, ms,
             ms * 1e6 / per_cu_blocks, 16384.0 * per_cu_blocks / (ms * 1e-3) / 1e9, 16384.0 * 600 * grid / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
