// Optional per-kernel timing with HIP events on the launch stream (diagnostics for bench.py's
// roofline line; off by default, zero cost when off).
#pragma once
#include <hip/hip_runtime.h>

namespace snr {

enum KernelId {
  K_MLP_PACK = 0, K_MLP_FWD, K_MLP_DGRAD, K_MLP_WGRAD, K_MLP_WGRAD_REDUCE, K_SAMPLE_COARSE, K_COMPOSITE_FWD,
  K_COMPOSITE_BWD, K_SAMPLE_FINE, K_MAKE_RAYS, K_ADAM, K_HG_PACK, K_HG_FWD, K_HG_BWD, K_MLP_WGRAD_PAIR,
  // (round 5) one id per kernel the training step launches, named like the kernel (bench.py's `kernels` keys = rocprofv3's names)
  K_COMPOSITE_TRAIN, K_COMPOSITE_TRAIN_REG, K_COMPOSITE_TRAIN_SAMPLE, K_PACK_RAYS_SAMPLE, K_PACK_RAYS, K_ADAM_PACK, K_WGRAD_POST,
  K_COUNT
};

void prof_begin(int id, hipStream_t s);
void prof_end(hipStream_t s);

struct ProfScope {
  hipStream_t s;
  ProfScope(int id, hipStream_t st) : s(st) { prof_begin(id, s); }
  ~ProfScope() { prof_end(s); }
};

}  // namespace snr
