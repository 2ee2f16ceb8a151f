// Probe: do two workgroups of one XCD that stream the SAME tiles at the same time fetch them from HBM once?
// (The premise of the recompute weight-gradient pass: the two jobs of a layer pair read the same saved tensors.)
// 256 workgroups x 8 waves stream 32 KiB tiles HBM -> LDS with global_load_lds_dwordx4 through a 5-slot ring, 4 tiles
// in flight (the shape of mlp_wgrad_kernel), `mfma` MFMAs per wave per tile as compute ballast.  Modes:
//   0  every workgroup its own tiles (w, w + 256, ...)                       -> bytes fetched = bytes requested
//   1  workgroups w and w ^ 8 (same XCD: block b runs on XCD b % 8) share    -> fetched = requested / 2 if sharing works
//   2  workgroups w and w ^ 1 (different XCDs) share                          -> fetched = requested (no common L2)
//   3  as 1, but the partner runs `lag` tiles behind
// Run under rocprofv3 --pmc FETCH_SIZE for the HBM-side byte count (x2: gfx950 correction, MI355X_MICROARCH.md).
// build: hipcc --offload-arch=gfx950 -O3 tests/probes/l2_share.hip -o l2_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kTile = 32 * 1024, kRing = 5, kDepth = 4, kNI = 4;   // 4 DMA instructions per wave per tile

__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int64_t n_tiles, int tiles_per_wg, int mode, int lag,
                                         int mfma, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = blockIdx.x;
  int64_t first, step;
  int skew = 0;
  if (mode == 0) { first = w; step = gridDim.x; }
  else {
    const int bit = mode == 2 ? 1 : 8;
    const int lo = w & (bit - 1), hi = w / (2 * bit);
    first = hi * bit + lo; step = gridDim.x / 2;
    if (mode == 3 && (w & bit)) skew = lag;
  }
  auto tile_of = [&](int i) {
    int j = i - skew;
    if (j < 0) j = 0;
    return (first + (int64_t)j * step) % n_tiles;
  };
  auto issue = [&](int i, int slot) {
    const char* s = src + tile_of(i) * kTile + wave * 1024 + lane * 16;
#pragma unroll
    for (int p = 0; p < kNI; ++p)
      __builtin_amdgcn_global_load_lds(s + p * 8192, LDS(smem + slot * kTile + wave * 1024 + p * 8192), 16, 0, 0);
  };
  for (int d = 0; d < kDepth; ++d) issue(d, d);
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
  int slot = 0, islot = kDepth;
  for (int i = 0; i < tiles_per_wg; ++i) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kNI * (kDepth - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    for (int m = 0; m < mfma; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    issue(i + kDepth < tiles_per_wg ? i + kDepth : tiles_per_wg - 1, islot);
    islot = islot + 1 == kRing ? 0 : islot + 1;
    slot = slot + 1 == kRing ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 1234.5f) sink[0] = acc[0];
}

int main(int argc, char** argv) {
  const int64_t n_tiles = 65536;   // 2 GiB
  const int tiles_per_wg = 512;    // 16 MiB per workgroup, 4 GiB requested per launch
  char* src; float* sink;
  hipMalloc(&src, n_tiles * kTile); hipMemset(src, 0, n_tiles * kTile); hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kRing * kTile);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  for (int mfma : {0, 16, 32, 48})
    for (int mode = 0; mode < 4; ++mode) {
      if (only >= 0 && mode != only) continue;
      for (int lag : {2, 8}) {
        if (mode != 3 && lag != 2) continue;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(a);
          k<<<256, 512, kRing * kTile>>>(src, n_tiles, tiles_per_wg, mode, lag, mfma, sink);
          hipEventRecord(b); hipEventSynchronize(b);
          hipEventElapsedTime(&ms, a, b);
        }
        const double req = 256.0 * tiles_per_wg * kTile;
        printf("mode %d lag %d mfma/wave/tile %2d: %.3f ms  requested %.2f TB/s  (%.2f us per tile per CU)\n", mode, lag, mfma, ms,
               req / ms / 1e9, ms * 1e3 / tiles_per_wg);
      }
    }
  return 0;
}
