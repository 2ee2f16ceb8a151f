// Shared device/host helpers for libspinnerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/spinnerf_hip.h"
#include "mlp_layout.h"
#include "prof.h"

namespace snr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SNR_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// 0 = ok, <0 = bad argument, >0 = hipError_t  (include/spinnerf_hip.h)
#define SNR_CHECK_ARG(cond, code) \
  do {                            \
    if (!(cond)) return (code);   \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SNR_OK : (int)e;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (kernel, device),
// from whichever thread launches first (forward and autograd threads both do).  Returns 0 or the hipError_t.
template <auto Kernel> inline int ensure_dynamic_lds(int bytes) {
  static std::atomic<uint64_t> done{0};   // one bit per device ordinal
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// ---- process-wide settings, read ONCE --------------------------------------------------------------------------------
// The SNR_* environment switches (A/B experiments) and the CU count of each device are read at the first call that needs
// them and cached — not on every launch.  snr_tunables_reload() (include/spinnerf_hip.h, diagnostics)
// re-reads the environment for tests and A/B scripts that change it inside one process.
struct Tunables {
  // bf16 training saves only the odd hidden layers / odd d z and rebuilds the even ones inside the weight-gradient pass
  // (mlp_wgrad_pair.h).  SNR_RECOMPUTE=0 selects the plain pass (every layer saved) for A/B measurements; the forward and
  // the backward of one step must see the same setting.
  bool recompute;
  int wgrad_splits;          // SNR_WGRAD_SPLITS: workgroups of the stand-alone plain split-K kernel (0 = one per CU)
  int pair_slots;            // SNR_PAIR_SLOTS: pair slots of the layer-pair launch (0 = default share of the CUs)
  int plain_wgs;             // SNR_PLAIN_WGS: plain workgroups inside the layer-pair launch (0 = the CUs the slots leave)
  int pair_w0;               // SNR_PAIR_W0: slot weight of pair 0 (others = 100)
  int pair_poll, pair_lead;  // SNR_PAIR_POLL / SNR_PAIR_LEAD: pacing of the two kinds of a slot
  int only_kind, only_pair;  // SNR_PAIR_KIND / SNR_PAIR_PAIR (debug builds of the pair kernel only)
  int merge_nets;            // SNR_MERGE_NETS=0: one backward launch sequence per network (A/B against the merged one)
  int chain_grid;            // SNR_CHAIN_GRID: workgroups of a chain-kernel launch (the rest grid-stride); 0 = default
  int enc_generic;           // SNR_ENC_GENERIC=1: the forward kernel's run-time positional encoding even where the compile-time one applies (tests)
};
inline Tunables read_tunables() {
  auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
  Tunables t{};
  const char* e = getenv("SNR_RECOMPUTE");
  t.recompute = !(e && e[0] == '0');
  t.wgrad_splits = geti("SNR_WGRAD_SPLITS", 0);
  t.pair_slots = geti("SNR_PAIR_SLOTS", 0);
  t.plain_wgs = geti("SNR_PLAIN_WGS", 0);
  t.pair_w0 = geti("SNR_PAIR_W0", 0);
  t.pair_poll = geti("SNR_PAIR_POLL", 16);
  t.pair_lead = geti("SNR_PAIR_LEAD", 2);
  t.only_kind = geti("SNR_PAIR_KIND", -1);
  t.only_pair = geti("SNR_PAIR_PAIR", -1);
  t.merge_nets = geti("SNR_MERGE_NETS", 1);
  t.enc_generic = geti("SNR_ENC_GENERIC", 0);
  t.chain_grid = geti("SNR_CHAIN_GRID", 0);
  return t;
}
inline Tunables& tunables_storage() { static Tunables t = read_tunables(); return t; }   // (thread-safe initialisation)
inline const Tunables& tunables() { return tunables_storage(); }
inline bool recompute_enabled() { return tunables().recompute; }

// CUs of the current device (cached per device ordinal; 256 when no device is present: size queries on a CPU-only host)
inline int cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  std::atomic<int>& c = cache[dev & 63];
  int n = c.load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  c.store(n, std::memory_order_relaxed);
  return n;
}

// ---- MFMA policies --------------------------------------------------------------------------
template <int P> struct Mma;

template <> struct Mma<kBF16> {
  using Frag = bf16x8;
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // frag h of a C tile = registers 8h..8h+7, rounded to bf16 (RNE, v_cvt_pk_bf16_f32)
  template <int H> static __device__ __forceinline__ Frag from_acc(const f32x16& v) {
    f32x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[8 * H + e];
    return __builtin_convertvector(t, Frag);
  }
  static __device__ __forceinline__ Frag zero() { return Frag{0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ void set(Frag& f, int e, float x) { f[e] = (__bf16)x; }
  static __device__ __forceinline__ float get(const Frag& f, int e) { return (float)f[e]; }
  // ---- fp16 segments (round 6) ----------------------------------------------------------------------------------------
  // The positional / directional ENCODINGS and the weight columns that multiply them travel as fp16 inside the same 16-byte
  // fragments: v_mfma_f32_32x32x16_f16 runs at the bf16 rate and accumulates into the same fp32 C tile, and sin / cos values
  // in [-1, 1] keep 11 mantissa bits instead of 8 (rounding error 2^-12 instead of 2^-9: the bf16 rounding of sin(2^9 x) was
  // most of the bf16 path's gradient error on semi-transparent rays, profiles/r05_bf16_grad_decomp.txt).  Hidden activations
  // stay bf16 (range); the weight-gradient pass consumes the encodings as bf16 (its other operand, d z, needs bf16's range).
  static __device__ __forceinline__ f32x16 mma_f16(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void set_f16(Frag& f, int e, float x) {
    f16x8 h = __builtin_bit_cast(f16x8, f);
    h[e] = (_Float16)x;
    f = __builtin_bit_cast(Frag, h);
  }
  // an fp16 fragment re-rounded to bf16 (what the saved-activation sections and the weight-gradient pass hold)
  static __device__ __forceinline__ Frag f16_to_bf16(Frag f) {
    return __builtin_convertvector(__builtin_convertvector(__builtin_bit_cast(f16x8, f), f32x8), Frag);
  }
};

template <> struct Mma<kFP32> {
  using Frag = f32x4;
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
  template <int H> static __device__ __forceinline__ Frag from_acc(const f32x16& v) {
    return Frag{v[4 * H], v[4 * H + 1], v[4 * H + 2], v[4 * H + 3]};
  }
  static __device__ __forceinline__ Frag zero() { return Frag{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ void set(Frag& f, int e, float x) { f[e] = x; }
  static __device__ __forceinline__ float get(const Frag& f, int e) { return f[e]; }
  // (the fp32 mode has no fp16 segments: the names exist so that the shared templates compile)
  static __device__ __forceinline__ f32x16 mma_f16(Frag a, Frag b, f32x16 c) { return mma(a, b, c); }
  static __device__ __forceinline__ void set_f16(Frag& f, int e, float x) { f[e] = x; }
  static __device__ __forceinline__ Frag f16_to_bf16(Frag f) { return f; }
};

}  // namespace snr
