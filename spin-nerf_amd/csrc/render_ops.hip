// Per-ray kernels of the volumetric renderer for gfx950: stratified sampling, alpha compositing
// (forward + backward), inverse-CDF hierarchical sampling + merge sort, ray generation, Adam.
//
// These are HBM/latency-bound fp32 kernels (<= 5 KB per ray against 0.3 GFLOP per ray in the MLP):
// one 64-lane wavefront per ray, coalesced row loads, wave-level scans via DPP shuffles, LDS only
// for the binary search table and the bitonic merge.  Compiled with -ffp-contract=off so the
// elementwise arithmetic rounds like the reference's separate torch ops.
#include "snr_common.h"
#include "render_internal.h"

namespace snr {

constexpr int kWave = 64;
constexpr int kRaysPerBlock = 4;  // 4 waves per workgroup, one ray each

__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const float o = __shfl_up(v, d, kWave);
    if (lane >= d) v *= o;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const float o = __shfl_up(v, d, kWave);
    if (lane >= d) v += o;
  }
  return v;
}
// inclusive suffix sum (lane i gets sum over lanes >= i)
__device__ __forceinline__ float wave_suffix_scan_add(float v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const float o = __shfl_down(v, d, kWave);
    if (lane + d < kWave) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// torch.linspace(0, 1, n)[i] in fp32: start + i*step below the midpoint, end - (n-1-i)*step above
__device__ __forceinline__ float linspace01(int i, int n) {
  if (n == 1) return 0.f;
  const float step = 1.0f / (float)(n - 1);
  return i < n / 2 ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// ------------------------------------------------------------------------------------------
// counter-based random numbers for the production path (the reference draws torch.rand / torch.randn,
// run_nerf.py:660-668, helpers:376-381; parity tests inject their draws instead): Philox4x32-10, element i of a
// call = counter (i, 0, offset_lo, offset_hi) under key (seed_lo, seed_hi) — no state, no generator launch, and a
// backward pass re-derives the forward's noise from the same (seed, offset) instead of reading it back from HBM.
// ------------------------------------------------------------------------------------------
// `base`: optional device counter added to the call offset — lets a captured HIP graph replay the same launch with
// fresh draws (the counter is advanced on the device, snr_step_state_advance)
struct Rng { uint32_t seed_lo, seed_hi, off_lo, off_hi; const unsigned long long* base; };
__device__ __forceinline__ void rng_offset(const Rng& g, uint32_t& lo, uint32_t& hi) {
  unsigned long long off = ((unsigned long long)g.off_hi << 32) | g.off_lo;
  if (g.base) off += *g.base;
  lo = (uint32_t)off; hi = (uint32_t)(off >> 32);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// U[0, 1) with 24 random bits (torch.rand's range)
__device__ __forceinline__ float rng_uniform(const Rng& g, int64_t i) {
  uint32_t olo, ohi;
  rng_offset(g, olo, ohi);
  uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), olo, ohi};
  philox4x32_10(c, g.seed_lo, g.seed_hi);
  return (float)(c[0] >> 8) * 5.9604644775390625e-8f;
}
// N(0, 1): Box-Muller on two words of one block
__device__ __forceinline__ float rng_normal(const Rng& g, int64_t i) {
  uint32_t olo, ohi;
  rng_offset(g, olo, ohi);
  uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), olo, ohi};
  philox4x32_10(c, g.seed_lo, g.seed_hi);
  const float u1 = (float)((c[0] >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
  const float u2 = (float)(c[1] >> 8) * 5.9604644775390625e-8f;
  return sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}

// ------------------------------------------------------------------------------------------
// stratified sampling (run_nerf.py:646-668)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float z_at(float near, float far, int i, int n, int lindisp) {
  const float t = linspace01(i, n);
  if (!lindisp) return near * (1.f - t) + far * t;
  return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
}

__global__ void sample_coarse_kernel(const float* __restrict__ rays, int ld, int64_t n_rays, int N, int lindisp,
                                     const float* __restrict__ t_rand, int use_rng, Rng rng,
                                     float* __restrict__ z_vals) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * N) return;
  const int64_t r = idx / N;
  const int i = (int)(idx - r * N);
  const float near = rays[r * ld + 6], far = rays[r * ld + 7];
  const float z = z_at(near, far, i, N, lindisp);
  if (!t_rand && !use_rng) { z_vals[idx] = z; return; }
  const float zl = i > 0 ? z_at(near, far, i - 1, N, lindisp) : z;
  const float zu = i < N - 1 ? z_at(near, far, i + 1, N, lindisp) : z;
  const float lower = i > 0 ? .5f * (z + zl) : z;       // run_nerf.py:656-658
  const float upper = i < N - 1 ? .5f * (zu + z) : z;
  z_vals[idx] = lower + (upper - lower) * (t_rand ? t_rand[idx] : rng_uniform(rng, idx));  // run_nerf.py:668
}

// ------------------------------------------------------------------------------------------
// alpha compositing (helpers:350-401); one wave per ray, S in chunks of 64 with a carried
// transmittance
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// density noise of sample e = ray * S + i: an array the caller drew (pre-scaled), the in-kernel generator, or none
struct NoiseSrc { const float* arr; int use_rng; Rng rng; float std; };
__device__ __forceinline__ float noise_at(const NoiseSrc& ns, int64_t e) {
  return ns.arr ? ns.arr[e] : (ns.use_rng ? rng_normal(ns.rng, e) * ns.std : 0.f);
}

// F.relu as torch evaluates it: a NaN density stays NaN (fmaxf would return 0 and hide a broken network behind "empty space")
__device__ __forceinline__ float relu_nan(float x) { return (x > 0.f || x != x) ? x : 0.f; }

struct CompMaps { float r, g, b, disp, acc, depth; };   // the same values in every lane

// forward of one ray by one wave: weights (and alpha) to memory, the maps returned
__device__ __forceinline__ CompMaps composite_fwd_ray(const float* __restrict__ raw, int C, const float* __restrict__ zr,
                                                      float dn, const NoiseSrc& ns, int64_t ray, int S, int white, int lane,
                                                      float* __restrict__ weights, float* __restrict__ alpha_out,
                                                      const float* __restrict__ alpha_in = nullptr) {
  float T = 1.f;  // transmittance entering this chunk
  float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
  for (int base = 0; base < S; base += kWave) {
    const int i = base + lane;
    const bool in = i < S;
    float w = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, z = 0.f, one_m = 1.f;
    if (in) {
      z = zr[i];
      float dist = (i + 1 < S) ? (zr[i + 1] - z) : 1e10f;  // helpers:366-367
      dist = dist * dn;                                     // helpers:369
      const float* rw = raw + (ray * S + i) * C;
      c0 = sigmoidf(rw[0]); c1 = sigmoidf(rw[1]); c2 = sigmoidf(rw[2]);
      const float s = rw[3] + noise_at(ns, ray * S + i);
      // alpha_in: the caller's own opacities (MVSeg's only_object post-processing) instead of helpers:364,382
      const float a = alpha_in ? alpha_in[ray * S + i] : 1.f - expf(-relu_nan(s) * dist);
      if (alpha_out) alpha_out[ray * S + i] = a;
      one_m = 1.f - a + 1e-10f;
      w = a;
    }
    const float incl = wave_incl_scan_mul(one_m, lane);
    float excl = __shfl_up(incl, 1, kWave);
    if (lane == 0) excl = 1.f;
    w = w * (T * excl);  // helpers:384
    T = T * __shfl(incl, kWave - 1, kWave);
    if (in) weights[ray * S + i] = w;
    sr += w * c0; sg += w * c1; sb += w * c2; sd += w * z; sa += w;
  }
  sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
  const float q = sd / sa;
  // torch.max(1e-10, q) propagates NaN (helpers:391)
  const float disp = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
  if (white) { sr += 1.f - sa; sg += 1.f - sa; sb += 1.f - sa; }  // helpers:394-395
  return CompMaps{sr, sg, sb, disp, sa, sd};
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    NoiseSrc ns, int64_t n_rays, int S, int white, float* __restrict__ rgb_map,
    float* __restrict__ disp_map, float* __restrict__ acc_map, float* __restrict__ depth_map,
    float* __restrict__ weights, float* __restrict__ alpha_out, const float* __restrict__ alpha_in) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float* rd = rays + ray * ld + 3;
  const float dn = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);  // torch.norm
  const CompMaps m = composite_fwd_ray(raw, C, z_vals + ray * S, dn, ns, ray, S, white, lane, weights, alpha_out, alpha_in);
  if (lane == 0) {
    rgb_map[3 * ray] = m.r; rgb_map[3 * ray + 1] = m.g; rgb_map[3 * ray + 2] = m.b;
    disp_map[ray] = m.disp; acc_map[ray] = m.acc; depth_map[ray] = m.depth;
  }
}

// Backward of one ray.  With G_i = dL/dw_i, dL/dalpha_i = G_i T_i - (sum_{j>i} G_j w_j) / (1 - alpha_i + 1e-10)
// (the cumprod backward torch uses when no factor is exactly 0), plus the direct g_alpha term.
__device__ __forceinline__ void composite_bwd_ray(const float* __restrict__ raw, int C, const float* __restrict__ zr,
                                                  float dn, const NoiseSrc& ns, int64_t ray, int S, int white, int detach,
                                                  float gr, float gg, float gb, float gD, float gA, float gP,
                                                  const float* __restrict__ g_w, const float* __restrict__ g_alpha,
                                                  float* __restrict__ d_raw, int lane,
                                                  const float* __restrict__ alpha_in = nullptr,
                                                  float* __restrict__ d_alpha_out = nullptr) {
  const int nchunks = (S + kWave - 1) / kWave;
  // pass 1: recompute w, accumulate acc/depth (needed for the disparity term)
  float T = 1.f, sd = 0.f, sa = 0.f;
  for (int base = 0; base < S; base += kWave) {
    const int i = base + lane;
    float one_m = 1.f, a = 0.f, z = 0.f;
    if (i < S) {
      z = zr[i];
      const float dist = ((i + 1 < S) ? (zr[i + 1] - z) : 1e10f) * dn;
      const float s = raw[(ray * S + i) * C + 3] + noise_at(ns, ray * S + i);
      a = alpha_in ? alpha_in[ray * S + i] : 1.f - expf(-relu_nan(s) * dist);
      one_m = 1.f - a + 1e-10f;
    }
    const float incl = wave_incl_scan_mul(one_m, lane);
    float excl = __shfl_up(incl, 1, kWave);
    if (lane == 0) excl = 1.f;
    const float w = a * (T * excl);
    T = T * __shfl(incl, kWave - 1, kWave);
    sd += w * z; sa += w;
  }
  sd = wave_sum(sd); sa = wave_sum(sa);
  const float q = sd / sa;
  // disp = 1/max(1e-10, q): d disp/dw_i = -(z_i - q)/acc / q^2 when q > 1e-10, else 0
  // A ray that hits nothing has acc = 0 and q = 0/0: its disparity is NaN and so is the gradient THROUGH the disparity —
  // as in the reference.  But without a gradient into the disparity (gP == 0: the map is not part of the loss, autograd
  // never visits that branch) the term must vanish exactly; 0 * NaN here used to put a NaN into d sigma of every such
  // ray that had one sample with a tiny positive density (relu' = 1, alpha = 0): noise-free training then died on its
  // first empty ray (tests/probes/hashgrid_nan_hunt.py, composite_nan.py).
  const bool use_disp = gP != 0.f;
  const float dq = (q > 1e-10f) ? -gP / (q * q) : ((q != q) ? q : 0.f);
  const float gwhite = white ? -(gr + gg + gb) : 0.f;

  // pass 2 (chunks in reverse): suffix sums of G_j w_j
  float suffix_next = 0.f;  // sum over samples of later chunks
  for (int ch = nchunks - 1; ch >= 0; --ch) {
    const int base = ch * kWave;
    // transmittance entering this chunk: recompute prefix product of earlier chunks
    float Tin = 1.f;
    for (int b2 = 0; b2 < base; b2 += kWave) {
      const int i2 = b2 + lane;
      const float z2 = zr[i2];
      const float dist2 = ((i2 + 1 < S) ? (zr[i2 + 1] - z2) : 1e10f) * dn;
      const float s2 = raw[(ray * S + i2) * C + 3] + noise_at(ns, ray * S + i2);
      const float a2 = alpha_in ? alpha_in[ray * S + i2] : 1.f - expf(-relu_nan(s2) * dist2);
      const float incl2 = wave_incl_scan_mul(1.f - a2 + 1e-10f, lane);
      Tin = Tin * __shfl(incl2, kWave - 1, kWave);
    }
    const int i = base + lane;
    const bool in = i < S;
    float one_m = 1.f, a = 0.f, z = 0.f, dist = 0.f, s = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (in) {
      z = zr[i];
      dist = ((i + 1 < S) ? (zr[i + 1] - z) : 1e10f) * dn;
      const float* rw = raw + (ray * S + i) * C;
      r0 = rw[0]; r1 = rw[1]; r2 = rw[2];
      s = rw[3] + noise_at(ns, ray * S + i);
      a = alpha_in ? alpha_in[ray * S + i] : 1.f - expf(-relu_nan(s) * dist);
      one_m = 1.f - a + 1e-10f;
    }
    const float incl = wave_incl_scan_mul(one_m, lane);
    float excl = __shfl_up(incl, 1, kWave);
    if (lane == 0) excl = 1.f;
    const float Ti = Tin * excl;
    const float w = a * Ti;
    const float c0 = sigmoidf(r0), c1 = sigmoidf(r1), c2 = sigmoidf(r2);
    float G = gD * z + gA + gwhite + (use_disp ? dq * (z - q) / sa : 0.f);
    if (!detach) G += gr * c0 + gg * c1 + gb * c2;
    if (g_w && in) G += g_w[ray * S + i];
    if (!in) G = 0.f;
    const float Gw = G * w;
    const float suf_incl = wave_suffix_scan_add(Gw, lane);
    const float suf_excl = suf_incl - Gw + suffix_next;  // sum_{j>i}
    suffix_next += __shfl(suf_incl, 0, kWave);
    if (in) {
      float dalpha = G * Ti - suf_excl / one_m;
      if (g_alpha) dalpha += g_alpha[ray * S + i];
      // d alpha / d s = dist * exp(-relu(s) dist) for s > 0; with the caller's own opacities the gradient stops at alpha
      const float ds = alpha_in ? 0.f : ((s > 0.f) ? dalpha * dist * expf(-s * dist) : 0.f);
      if (d_alpha_out) d_alpha_out[ray * S + i] = dalpha;
      float* o = d_raw + (ray * S + i) * C;
      o[0] = gr * w * c0 * (1.f - c0);
      o[1] = gg * w * c1 * (1.f - c1);
      o[2] = gb * w * c2 * (1.f - c2);
      o[3] = ds;
      for (int c = 4; c < C; ++c) o[c] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    NoiseSrc ns, int64_t n_rays, int S, int white, int detach, const float* __restrict__ g_rgb,
    const float* __restrict__ g_disp, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
    const float* __restrict__ g_w, const float* __restrict__ g_alpha, float* __restrict__ d_raw,
    const float* __restrict__ alpha_in, float* __restrict__ d_alpha_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  if (ray >= n_rays) return;
  const float* rd = rays + ray * ld + 3;
  const float dn = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
  const float gr = g_rgb ? g_rgb[3 * ray] : 0.f, gg = g_rgb ? g_rgb[3 * ray + 1] : 0.f,
              gb = g_rgb ? g_rgb[3 * ray + 2] : 0.f;
  const float gD = g_depth ? g_depth[ray] : 0.f, gA = g_acc ? g_acc[ray] : 0.f, gP = g_disp ? g_disp[ray] : 0.f;
  composite_bwd_ray(raw, C, z_vals + ray * S, dn, ns, ray, S, white, detach, gr, gg, gb, gD, gA, gP, g_w, g_alpha, d_raw,
                    lane, alpha_in, d_alpha_out);
}

// One kernel per network of the TRAINING step: compositing forward, the loss term mean((rgb - target)^2) of this
// network's colour map with its gradient (img2mse, helpers:15; run_nerf.py:1482-1490), and the compositing backward —
// render_rays' tail and autograd's head without the rgb / gradient round trip and two extra launches.  inv_count =
// 1 / (3 * rays of the GLOBAL batch); loss[slot] += this call's term (atomics, one per workgroup).
// the term a ray belongs to (-1: none — no loss, zero gradient), its target row and the loss gradient at the maps
struct RayLoss { int term; float gr, gg, gb, gP; int detach; float err; };
__device__ __forceinline__ int term_of(const LossSpec& sp, int64_t ray) {
  int ti = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < sp.n && ray >= sp.t[k].first && ray < sp.t[k].first + sp.t[k].n) ti = k;
  return ti;
}
// the ray's term and its target row, fetched at the TOP of a kernel: the loss needs them only behind the whole compositing
// forward, and a dependent global load there is a microsecond nothing else covers (one wave per ray, one wave per SIMD)
struct RayTarget { int term; float t0, t1, t2; };
__device__ __forceinline__ RayTarget ray_target(const LossSpec& sp, int64_t ray) {
  // (the ray, and with it the term, is the same in every lane of the wave: as a scalar the index selects the term's fields
  //  with scalar loads from the kernel arguments — indexed per lane the struct went to scratch memory)
  RayTarget T{__builtin_amdgcn_readfirstlane(term_of(sp, ray)), 0.f, 0.f, 0.f};
  if (T.term < 0) return T;
  const LossTerm& t = sp.t[T.term];
  const int64_t rel = ray - t.first;
  if (t.kind == 2) T.t0 = t.target[rel];
  else { T.t0 = t.target[3 * rel]; T.t1 = t.target[3 * rel + 1]; T.t2 = t.target[3 * rel + 2]; }
  return T;
}
// kind 0 / 1: mean((rgb - target)^2) (1: through detached weights, helpers:385); kind 2: mean((disp - target)^2)
__device__ __forceinline__ RayLoss ray_loss(const LossSpec& sp, const RayTarget& T, float r, float g, float b, float disp) {
  RayLoss L{T.term, 0.f, 0.f, 0.f, 0.f, 0, 0.f};
  if (L.term < 0) return L;
  const LossTerm& t = sp.t[L.term];
  if (t.kind == 2) {
    const float dd = disp - T.t0;
    L.err = dd * dd * t.inv_count;
    L.gP = 2.f * dd * t.inv_count;
  } else {
    const float dr = r - T.t0, dg = g - T.t1, db = b - T.t2;
    L.err = (dr * dr + dg * dg + db * db) * t.inv_count;
    L.gr = 2.f * dr * t.inv_count; L.gg = 2.f * dg * t.inv_count; L.gb = 2.f * db * t.inv_count;
    L.detach = t.kind == 1;
  }
  return L;
}
// per workgroup: the rays' (already scaled) errors into their terms' slots, one atomic per term present
__device__ __forceinline__ void add_losses(const LossSpec& sp, int final_pass, float err, int term, float* loss, float* sq, int* st) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sq[wv] = err; st[wv] = term; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // (no array indexed by a run-time term: that was a serial chain of scratch-memory read-modify-writes at the tail of every
    //  workgroup; the sums run over the waves in the same order as before)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float tk = 0.f;
      bool has = false;
#pragma unroll
      for (int w = 0; w < kRaysPerBlock; ++w)
        if (st[w] == k) { tk += sq[w]; has = true; }
      if (has) {
        atomicAdd(loss + sp.t[k].slot, tk);
        if (final_pass && sp.t[k].slot_final != kNoSlot) atomicAdd(loss + sp.t[k].slot_final, tk);
      }
    }
  }
}

__global__ __launch_bounds__(256) void composite_train_kernel(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    NoiseSrc ns, int64_t n_rays, int S, int white, LossSpec spec, int final_pass,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ depth_map, float* __restrict__ weights, float* __restrict__ d_raw, float* __restrict__ loss) {
  __shared__ float sq[kRaysPerBlock];
  __shared__ int st[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  float err = 0.f;
  int term = -1;
  if (ray < n_rays) {
    const float* rd = rays + ray * ld + 3;
    const float dn = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
    const RayTarget tgt = ray_target(spec, ray);
    const CompMaps m = composite_fwd_ray(raw, C, z_vals + ray * S, dn, ns, ray, S, white, lane, weights, nullptr);
    if (lane == 0) {
      rgb_map[3 * ray] = m.r; rgb_map[3 * ray + 1] = m.g; rgb_map[3 * ray + 2] = m.b;
      disp_map[ray] = m.disp; acc_map[ray] = m.acc; depth_map[ray] = m.depth;
    }
    const RayLoss L = ray_loss(spec, tgt, m.r, m.g, m.b, m.disp);
    err = L.err; term = L.term;
    composite_bwd_ray(raw, C, z_vals + ray * S, dn, ns, ray, S, white, L.detach, L.gr, L.gg, L.gb, 0.f, 0.f, L.gP, nullptr, nullptr,
                      d_raw, lane);
  }
  add_losses(spec, final_pass, err, term, loss, sq, st);
}

// The same kernel with the ray held in registers (S <= 64 * NCH): every global load of the ray is issued up front, the
// forward, the loss gradient and the backward then run out of registers.  The generic version above walks the ray three
// times (forward, the backward's accumulate pass, the backward's reverse pass with its prefix recomputation), each chunk
// waiting for its own loads — with one wave per SIMD that latency chain was the kernel (16.5 us for 1024 rays x 192
// samples).  Arithmetic and its order are those of composite_fwd_ray / composite_bwd_ray with g_disp = g_acc = g_depth = 0.
// one ray by one wave; returns the ray's squared colour error.  w_keep (LDS, may be null): the weights once more, for a
// hierarchical-sampling stage fused behind this one (composite_train_sample_kernel)
template <int NCH>
__device__ __forceinline__ float composite_train_ray(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    const NoiseSrc& ns, int64_t ray, int S, int white, const LossSpec& spec, int& term,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ depth_map, float* __restrict__ weights, float* __restrict__ d_raw, int lane, float* w_keep) {
  float e2 = 0.f;
  {
    const float* rd = rays + ray * ld + 3;
    const float* zr = z_vals + ray * S;
    const RayTarget tgt = ray_target(spec, ray);   // (with the ray's other loads)
    float z[NCH], zn[NCH], r0[NCH], r1[NCH], r2[NCH], s[NCH];
    bool in[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {          // all loads of the ray, independent of each other
      const int i = 64 * c + lane;
      in[c] = i < S;
      z[c] = zn[c] = r0[c] = r1[c] = r2[c] = s[c] = 0.f;
      if (in[c]) {
        z[c] = zr[i];
        zn[c] = (i + 1 < S) ? zr[i + 1] : 0.f;
        const float* rw = raw + (ray * S + i) * C;
        if (C == 4) {
          const f32x4 v = *(const f32x4*)rw;
          r0[c] = v[0]; r1[c] = v[1]; r2[c] = v[2]; s[c] = v[3];
        } else {
          r0[c] = rw[0]; r1[c] = rw[1]; r2[c] = rw[2]; s[c] = rw[3];
        }
      }
    }
    const float dn = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
    // ---- forward (composite_fwd_ray) ----
    float dist[NCH], c0[NCH], c1[NCH], c2[NCH], one_m[NCH], Ti[NCH], w[NCH];
    float T = 1.f, sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = 64 * c + lane;
      float a = 0.f;
      dist[c] = 0.f; c0[c] = c1[c] = c2[c] = 0.f; one_m[c] = 1.f;
      if (in[c]) {
        float d = (i + 1 < S) ? (zn[c] - z[c]) : 1e10f;
        dist[c] = d * dn;
        c0[c] = sigmoidf(r0[c]); c1[c] = sigmoidf(r1[c]); c2[c] = sigmoidf(r2[c]);
        s[c] = s[c] + noise_at(ns, ray * S + i);
        a = 1.f - expf(-relu_nan(s[c]) * dist[c]);
        one_m[c] = 1.f - a + 1e-10f;
      }
      const float incl = wave_incl_scan_mul(one_m[c], lane);
      float excl = __shfl_up(incl, 1, kWave);
      if (lane == 0) excl = 1.f;
      Ti[c] = T * excl;
      w[c] = a * Ti[c];
      T = T * __shfl(incl, kWave - 1, kWave);
      if (in[c]) { weights[ray * S + i] = w[c]; if (w_keep) w_keep[i] = w[c]; }
      sr += w[c] * c0[c]; sg += w[c] * c1[c]; sb += w[c] * c2[c]; sd += w[c] * z[c]; sa += w[c];
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    const float q = sd / sa;
    const float disp = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
    if (white) { sr += 1.f - sa; sg += 1.f - sa; sb += 1.f - sa; }
    if (lane == 0) {
      rgb_map[3 * ray] = sr; rgb_map[3 * ray + 1] = sg; rgb_map[3 * ray + 2] = sb;
      disp_map[ray] = disp; acc_map[ray] = sa; depth_map[ray] = sd;
    }
    // ---- loss gradient and backward (composite_bwd_ray with g_acc = g_depth = 0) ----
    const RayLoss L = ray_loss(spec, tgt, sr, sg, sb, disp);
    term = L.term;
    e2 = L.err;
    const float gr = L.gr, gg = L.gg, gb = L.gb;
    const int detach = L.detach;
    const float gwhite = white ? -(gr + gg + gb) : 0.f;
    // the disparity term of composite_bwd_ray: exactly absent without a gradient into the disparity
    const bool use_disp = L.gP != 0.f;
    const float dq = (q > 1e-10f) ? -L.gP / (q * q) : ((q != q) ? q : 0.f);
    float suffix_next = 0.f;
#pragma unroll
    for (int c = NCH - 1; c >= 0; --c) {
      const int i = 64 * c + lane;
      float G = 0.f * z[c] + 0.f + gwhite + (use_disp ? dq * (z[c] - q) / sa : 0.f);
      if (!detach) G += gr * c0[c] + gg * c1[c] + gb * c2[c];
      if (!in[c]) G = 0.f;
      const float Gw = G * w[c];
      const float suf_incl = wave_suffix_scan_add(Gw, lane);
      const float suf_excl = suf_incl - Gw + suffix_next;
      suffix_next += __shfl(suf_incl, 0, kWave);
      if (in[c]) {
        const float dalpha = G * Ti[c] - suf_excl / one_m[c];
        const float ds = (s[c] > 0.f) ? dalpha * dist[c] * expf(-s[c] * dist[c]) : 0.f;
        float* o = d_raw + (ray * S + i) * C;
        const float o0 = gr * w[c] * c0[c] * (1.f - c0[c]), o1 = gg * w[c] * c1[c] * (1.f - c1[c]),
                    o2 = gb * w[c] * c2[c] * (1.f - c2[c]);
        if (C == 4) {
          *(f32x4*)o = f32x4{o0, o1, o2, ds};
        } else {
          o[0] = o0; o[1] = o1; o[2] = o2; o[3] = ds;
          for (int k = 4; k < C; ++k) o[k] = 0.f;
        }
      }
    }
  }
  return e2;
}

template <int NCH>
__global__ __launch_bounds__(256) void composite_train_reg_kernel(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    NoiseSrc ns, int64_t n_rays, int S, int white, LossSpec spec, int final_pass,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ depth_map, float* __restrict__ weights, float* __restrict__ d_raw, float* __restrict__ loss) {
  __shared__ float sq[kRaysPerBlock];
  __shared__ int st[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  float err = 0.f;
  int term = -1;
  if (ray < n_rays)
    err = composite_train_ray<NCH>(raw, C, z_vals, rays, ld, ns, ray, S, white, spec, term, rgb_map, disp_map, acc_map,
                                   depth_map, weights, d_raw, lane, nullptr);
  add_losses(spec, final_pass, err, term, loss, sq, st);
}

// ------------------------------------------------------------------------------------------
// hierarchical sampling (helpers:304-347) + sort of the union (run_nerf.py:702) + z_std (:726)
// one wave per ray; LDS per wave: cdf[Nc-1], bins[Nc-1], sort buffer[pow2 >= Nc+Nf]
// ------------------------------------------------------------------------------------------
// one ray by one wave.  zc: the ray's coarse z (direct: its nb bin positions); wr: the ray's weights such that wr[i + 1] is
// the weight of bin i (global memory, or the LDS copy a fused compositing stage left); cdf: 2 * nb + npow2 floats of LDS
__device__ __forceinline__ void sample_fine_ray(const float* __restrict__ zc, const float* wr,
                                                const float* __restrict__ u_in, int64_t ray, int Nc, int Nf, int npow2,
                                                float* __restrict__ z_out, float* __restrict__ z_samples,
                                                float* __restrict__ z_std, int direct, int use_rng, const Rng& rng,
                                                float* cdf, int lane) {
  const int nb = Nc - 1;   // number of bins (midpoints), cdf has nb entries, pdf nb-1
  float* bins = cdf + nb;
  float* srt = bins + nb;

  // pdf = (w[1:-1] + 1e-5) / sum   (helpers:306-307; the slice is run_nerf.py:699)
  float tot = 0.f;
  for (int i = lane; i < nb - 1; i += kWave) tot += wr[i + 1] + 1e-5f;
  tot = wave_sum(tot);
  // cdf = [0, cumsum(pdf)]  (helpers:308-309)
  float carry = 0.f;
  for (int base = 0; base < nb - 1; base += kWave) {
    const int i = base + lane;
    const float p = (i < nb - 1) ? (wr[i + 1] + 1e-5f) / tot : 0.f;
    const float inc = wave_incl_scan_add(p, lane) + carry;
    if (i < nb - 1) cdf[i + 1] = inc;
    carry = __shfl(inc, kWave - 1, kWave);
  }
  if (lane == 0) cdf[0] = 0.f;
  for (int i = lane; i < nb; i += kWave) bins[i] = direct ? zc[i] : .5f * (zc[i + 1] + zc[i]);  // run_nerf.py:697
  if (!direct)
    for (int i = lane; i < Nc; i += kWave) srt[i] = zc[i];
  for (int i = Nc + Nf + lane; i < npow2; i += kWave) srt[i] = __builtin_inff();
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  float sum = 0.f;
  for (int k = lane; k < Nf; k += kWave) {
    const float u = u_in ? u_in[ray * Nf + k] : (use_rng ? rng_uniform(rng, ray * Nf + k) : linspace01(k, Nf));  // helpers:313-316
    // inds = searchsorted(cdf, u, right=True) = #{cdf <= u}
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;      // helpers:332
    const int above = lo < nb - 1 ? lo : nb - 1;    // helpers:333
    const float cb = cdf[below], ca = cdf[above];
    float denom = ca - cb;
    if (denom < 1e-5f) denom = 1.f;                 // helpers:343
    const float t = (u - cb) / denom;
    const float bb = bins[below];
    const float zs = bb + t * (bins[above] - bb);   // helpers:345
    srt[Nc + k] = zs;
    if (z_samples) z_samples[ray * Nf + k] = zs;
    sum += zs;
  }
  // z_std = population std of the new samples (run_nerf.py:726)
  sum = wave_sum(sum);
  const float mean = sum / (float)Nf;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float var = 0.f;
  for (int k = lane; k < Nf; k += kWave) { const float d = srt[Nc + k] - mean; var += d * d; }
  var = wave_sum(var);
  if (lane == 0 && z_std) z_std[ray] = sqrtf(var / (float)Nf);

  if (direct) return;
  if (npow2 == 256) {
    // (round 5) the standard union (64 + 128 -> 256 slots) is sorted in REGISTERS: four consecutive elements per lane, partners
    // at distance 1 / 2 inside the lane, further ones through a lane shuffle — the same compare-exchange network as the LDS loop
    // below (same values out), without its 36 rounds of LDS reads, writes and wave barriers
    f32x4 r;   // (the sort buffer starts 2 (Nc - 1) floats into the wave's LDS: not 16-byte aligned)
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = srt[4 * lane + c];
    auto cx = [](float own, float other, bool lower, bool up) {   // this element after a compare-exchange with its partner
      const float a = lower ? own : other, b = lower ? other : own;   // a: the pair's lower index
      const bool gt = a > b;
      return lower ? (up ? (gt ? b : a) : (gt ? a : b)) : (up ? (gt ? a : b) : (gt ? b : a));
    };
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        if (jj >= 4) {
          const bool lower = (lane & (jj >> 2)) == 0;
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = __shfl_xor(r[c], jj >> 2, kWave);
#pragma unroll
          for (int c = 0; c < 4; ++c) r[c] = cx(r[c], o[c], lower, ((4 * lane + c) & k) == 0);
        } else {
          f32x4 n;
#pragma unroll
          for (int c = 0; c < 4; ++c) n[c] = cx(r[c], r[c ^ jj], (c & jj) == 0, ((4 * lane + c) & k) == 0);
          r = n;
        }
      }
    }
    const int n_out = Nc + Nf;
    float* zo = z_out + ray * n_out;
    if ((n_out & 3) == 0) {
      if (4 * lane < n_out) *(f32x4*)(zo + 4 * lane) = r;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) if (4 * lane + c < n_out) zo[4 * lane + c] = r[c];
    }
    return;
  }
  // bitonic sort of srt[0..npow2) ascending (values only, run_nerf.py:702)
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = lane; t < npow2 / 2; t += kWave) {
        const int i = 2 * t - (t & (jj - 1));   // index with bit jj clear
        const int p = i + jj;
        const bool up = (i & k) == 0;
        const float a = srt[i], b = srt[p];
        if ((a > b) == up) { srt[i] = b; srt[p] = a; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (int i = lane; i < Nc + Nf; i += kWave) z_out[ray * (Nc + Nf) + i] = srt[i];
}

__global__ __launch_bounds__(256) void sample_fine_kernel(const float* __restrict__ z_coarse,
                                                          const float* __restrict__ weights,
                                                          const float* __restrict__ u_in, int64_t n_rays, int Nc, int Nf,
                                                          int npow2, float* __restrict__ z_out,
                                                          float* __restrict__ z_samples, float* __restrict__ z_std,
                                                          int direct, int use_rng, Rng rng) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  const int nb = Nc - 1;
  const int per_wave = 2 * nb + npow2;
  if (ray >= n_rays) return;  // whole wave exits together; barriers inside are wave-local
  // direct (snr_sample_pdf): z_coarse holds the nb bin positions themselves and weights the nb-1 bin weights;
  // no union / sort.  Otherwise bins are the midpoints of z_coarse and the pdf comes from weights[1:-1].
  const float* zc = z_coarse + ray * (direct ? nb : Nc);
  const float* wr = direct ? weights + ray * (nb - 1) - 1 : weights + ray * Nc;
  sample_fine_ray(zc, wr, u_in, ray, Nc, Nf, npow2, z_out, z_samples, z_std, direct, use_rng, rng, lds + wv * per_wave, lane);
}

// The coarse pass's tail and the fine pass's head of a TRAINING render_rays as one kernel (run_nerf.py:680-702 + the loss
// term and autograd's first step): compositing forward + loss + compositing backward of the coarse samples, then
// hierarchical sampling + sort from the weights just computed — they never leave the CU (LDS) between the two stages, and
// one launch (with its ramp: a wave per ray, 1024 rays) replaces two.  Nc <= 64.
__global__ __launch_bounds__(256) void composite_train_sample_kernel(
    const float* __restrict__ raw, int C, const float* __restrict__ z_vals, const float* __restrict__ rays, int ld,
    NoiseSrc ns, int64_t n_rays, int Nc, int white, LossSpec spec,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ depth_map, float* __restrict__ weights, float* __restrict__ d_raw, float* __restrict__ loss,
    const float* __restrict__ u_in, int Nf, int npow2, float* __restrict__ z_out, float* __restrict__ z_samples,
    float* __restrict__ z_std, int use_rng, Rng rng) {
  extern __shared__ float lds[];
  __shared__ float sq[kRaysPerBlock];
  __shared__ int st[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  const int nb = Nc - 1;
  const int per_wave = Nc + 2 * nb + npow2;
  float* w_keep = lds + wv * per_wave;
  float err = 0.f;
  int term = -1;
  if (ray < n_rays) {
    err = composite_train_ray<1>(raw, C, z_vals, rays, ld, ns, ray, Nc, white, spec, term, rgb_map, disp_map, acc_map,
                                 depth_map, weights, d_raw, lane, w_keep);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sample_fine_ray(z_vals + ray * Nc, w_keep, u_in, ray, Nc, Nf, npow2, z_out, z_samples, z_std, 0, use_rng, rng, w_keep + Nc, lane);
  }
  add_losses(spec, 0, err, term, loss, sq, st);
}

// the NaN guard of a loss term (run_nerf.py:1518-1521): see render_internal.h: loss_guard_impl
__global__ void loss_guard_kernel(float* __restrict__ loss, int slot, float* __restrict__ d0, int64_t n0, float* __restrict__ d1,
                                  int64_t n1) {
  const float v = loss[slot];
  const bool bad = v != v;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (bad) {
    if (d0 && i < n0) d0[i] = 0.f;
    if (d1 && i < n1) d1[i] = 0.f;
  }
  if (i == 0 && !bad) loss[0] += v;
}

// ------------------------------------------------------------------------------------------
// rays (helpers:249-260, 283-300; run_nerf.py:117-153)
// ------------------------------------------------------------------------------------------
struct Pose { float m[12]; };

// one packed row [o d near far (depth) (viewdirs)]: viewdirs = vsrc / |vsrc| taken BEFORE the NDC warp
// (run_nerf.py:128-135; vsrc = d unless c2w_staticcam swaps the camera), depth column per run_nerf.py:148-149
__device__ __forceinline__ void write_ray_row(float* o, float* d, const float* vsrc, int H, int W, float focal, int ndc,
                                              float ndc_near, float near, float far, const float* depth,
                                              int use_viewdirs, float* __restrict__ out) {
  int col = 8;
  if (depth) out[col++] = *depth;
  if (use_viewdirs) {
    const float n = sqrtf(vsrc[0] * vsrc[0] + vsrc[1] * vsrc[1] + vsrc[2] * vsrc[2]);
    out[col] = vsrc[0] / n; out[col + 1] = vsrc[1] / n; out[col + 2] = vsrc[2] / n;
  }
  if (ndc) {  // ndc_rays (helpers:283-300); render() calls it with near = 1 (run_nerf.py:140)
    const float nr = ndc_near;
    const float t = -(nr + o[2]) / d[2];
    const float ox = o[0] + t * d[0], oy = o[1] + t * d[1], oz = o[2] + t * d[2];
    const float sx = -1.f / ((float)W / (2.f * focal)), sy = -1.f / ((float)H / (2.f * focal));
    const float o0 = sx * ox / oz, o1 = sy * oy / oz, o2 = 1.f + 2.f * nr / oz;
    const float d0 = sx * (d[0] / d[2] - ox / oz), d1 = sy * (d[1] / d[2] - oy / oz), d2 = -2.f * nr / oz;
    o[0] = o0; o[1] = o1; o[2] = o2; d[0] = d0; d[1] = d1; d[2] = d2;
  }
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
  out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
  out[6] = near; out[7] = far;
}

__global__ void make_rays_kernel(int H, int W, float focal, Pose c2w, int i0, int j0, int h, int w, int ndc,
                                 float near, float far, int use_viewdirs, float* __restrict__ rays, int ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)h * w) return;
  const int row = i0 + (int)(idx / w), col = j0 + (int)(idx % w);
  // dirs = [(i - W/2)/f, -(j - H/2)/f, -1]; d = sum(dirs * c2w[:3,:3], -1); o = c2w[:3,3]
  const float dx = ((float)col - (float)W * .5f) / focal;
  const float dy = -((float)row - (float)H * .5f) / focal;
  const float dz = -1.f;
  float d[3], o[3];
  for (int r = 0; r < 3; ++r) {
    d[r] = dx * c2w.m[4 * r] + dy * c2w.m[4 * r + 1] + dz * c2w.m[4 * r + 2];
    o[r] = c2w.m[4 * r + 3];
  }
  const float v[3] = {d[0], d[1], d[2]};
  write_ray_row(o, d, v, H, W, focal, ndc, 1.f, near, far, nullptr, use_viewdirs, rays + idx * ld);
}

// rows from rays the caller already holds (render(rays=...), run_nerf.py:117-153): same row as above, plus the
// optional pieces of render(): viewing directions from a second camera (c2w_staticcam, :131-133), per-ray near / far
// (:106-107), the COLMAP depth column (:148-149)
__global__ void pack_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ view_src, int64_t n, int H, int W, float focal, int ndc,
                                 float ndc_near, float near, float far, const float* __restrict__ near_rows,
                                 const float* __restrict__ far_rows, const float* __restrict__ depths, int use_viewdirs,
                                 float* __restrict__ rays, int ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float o[3] = {rays_o[3 * idx], rays_o[3 * idx + 1], rays_o[3 * idx + 2]};
  float d[3] = {rays_d[3 * idx], rays_d[3 * idx + 1], rays_d[3 * idx + 2]};
  const float* vs = view_src ? view_src + 3 * idx : rays_d + 3 * idx;
  const float v[3] = {vs[0], vs[1], vs[2]};
  write_ray_row(o, d, v, H, W, focal, ndc, ndc_near, near_rows ? near_rows[idx] : near, far_rows ? far_rows[idx] : far,
                depths ? depths + idx : nullptr, use_viewdirs, rays + idx * ld);
}

// The head of a training render() as ONE kernel: the packed row of every ray (pack_rays_kernel's plain case), the stratified
// z_vals of its coarse samples (sample_coarse_kernel: run_nerf.py:646-668) and the zero fill of the step's loss accumulator —
// three launches of a few microseconds each before.  One thread per (ray, sample); the thread of sample 0 packs the row.
__global__ void pack_rays_sample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t n, int H,
                                        int W, float focal, int ndc, float ndc_near, float near, float far, int use_viewdirs,
                                        float* __restrict__ rays, int ld, int N, int lindisp,
                                        const float* __restrict__ t_rand, int use_rng, Rng rng, float* __restrict__ z_vals,
                                        float* __restrict__ zero, int n_zero) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n_zero) zero[idx] = 0.f;
  if (idx >= n * N) return;
  const int64_t r = idx / N;
  const int i = (int)(idx - r * N);
  if (i == 0) {
    float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float v[3] = {d[0], d[1], d[2]};
    write_ray_row(o, d, v, H, W, focal, ndc, ndc_near, near, far, nullptr, use_viewdirs, rays + r * ld);
  }
  const float z = z_at(near, far, i, N, lindisp);
  if (!t_rand && !use_rng) { z_vals[idx] = z; return; }
  const float zl = i > 0 ? z_at(near, far, i - 1, N, lindisp) : z;
  const float zu = i < N - 1 ? z_at(near, far, i + 1, N, lindisp) : z;
  const float lower = i > 0 ? .5f * (z + zl) : z;       // run_nerf.py:656-658
  const float upper = i < N - 1 ? .5f * (zu + z) : z;
  z_vals[idx] = lower + (upper - lower) * (t_rand ? t_rand[idx] : rng_uniform(rng, idx));  // run_nerf.py:668
}

// positional encoding as a standalone op (Embedder.embed, helpers:22-52): out[i] = [x, sin(2^0 x), cos(2^0 x), ...,
// sin(2^(L-1) x), cos(2^(L-1) x)], C input columns, C * (1 + 2L) output columns.  The MLP kernels fuse the encoding;
// this serves callers of get_embedder()[0] (run_nerf.py:383-388).
__global__ void embed_kernel(const float* __restrict__ x, int64_t n, int C, int L, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int od = C * (1 + 2 * L);
  if (idx >= n * od) return;
  const int64_t row = idx / od;
  const int c = (int)(idx - row * od);
  if (c < C) { out[idx] = x[row * C + c]; return; }
  const int k = (c - C) / (2 * C), r = (c - C) % (2 * C);
  const float v = x[row * C + (r % C)] * __builtin_ldexpf(1.0f, k);   // freq_bands = 2^k exactly (helpers:36-39)
  out[idx] = r < C ? sinf(v) : cosf(v);
}

// ------------------------------------------------------------------------------------------
// loss of the training step: mean((rgb - t)^2) [+ mean((rgb0 - t)^2)] and its gradients
// (img2mse, helpers:15; run_nerf.py:1482-1490).  One workgroup: 3 * N_rand elements.
// ------------------------------------------------------------------------------------------
__global__ void mse_pair_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ t,
                                int64_t n, float* __restrict__ loss, float* __restrict__ ga, float* __restrict__ gb) {
  __shared__ float red[2][16];
  float sa = 0.f, sb = 0.f;
  const float inv = 1.f / (float)n;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float da = a[i] - t[i];
    sa += da * da;
    ga[i] = 2.f * da * inv;
    if (b) {
      const float db = b[i] - t[i];
      sb += db * db;
      gb[i] = 2.f * db * inv;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sa += __shfl_down(sa, off, 64);
    sb += __shfl_down(sb, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wave] = sa; red[1][wave] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ta += red[0][w]; tb += red[1][w]; }
    loss[0] = ta * inv + tb * inv;
    loss[1] = ta * inv;   // the fine term alone (PSNR of the reference's log line)
  }
}

// ------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, run_nerf.py:433-434)
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 + 3 < n) {
    f32x4 P = *(f32x4*)(p + i0), G = *(const f32x4*)(g + i0), M = *(f32x4*)(m + i0), V = *(f32x4*)(v + i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] * gscale;
      M[k] = M[k] * b1 + (1.f - b1) * gk;
      V[k] = V[k] * b2 + (1.f - b2) * gk * gk;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - (lr / bc1) * (M[k] / denom);
    }
    *(f32x4*)(p + i0) = P; *(f32x4*)(m + i0) = M; *(f32x4*)(v + i0) = V;
  } else {
    for (int64_t i = i0; i < n; ++i) {
      const float gk = g[i] * gscale;
      m[i] = m[i] * b1 + (1.f - b1) * gk;
      v[i] = v[i] * b2 + (1.f - b2) * gk * gk;
      const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
      p[i] = p[i] - (lr / bc1) * (m[i] / denom);
    }
  }
}

// beta^k by squaring, in double: the same sequence of IEEE multiplications on the host (snr_adam_step) and on the device
// (step_state_advance_kernel), so that a replayed graph and the eager step use bit-identical bias corrections — libm's
// and the device library's pow() differ in the last bit now and then, and one ulp in a parameter is enough for bf16
// rounding + Adam to send two runs apart
__host__ __device__ inline double powi(double b, long long k) {
  double r = 1.0;
  while (k > 0) {
    if (k & 1) r *= b;
    b *= b;
    k >>= 1;
  }
  return r;
}

// Adam with the step's rate and bias corrections read from device memory (snr_step_state), for captured graphs
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, const snr_step_state* __restrict__ st, float b1, float b2,
                                float eps, float gscale) {
  const float lr = st->lr, bc1 = st->bc1, bc2_sqrt = st->bc2_sqrt;
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (int64_t i = i0; i < n && i < i0 + 4; ++i) {
    const float gk = g[i] * gscale;
    const float mk = m[i] * b1 + (1.f - b1) * gk;
    const float vk = v[i] * b2 + (1.f - b2) * gk * gk;
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    m[i] = mk; v[i] = vk;
    p[i] = p[i] - (lr / bc1) * (mk / denom);
  }
}

// what RenderTrainer.apply_gradients does on the host after a step (run_nerf.py:1616-1622, 1703), on the device
__global__ void step_state_advance_kernel(snr_step_state* st, double lrate, double decay_steps, double b1, double b2,
                                          unsigned long long n_offsets) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->offset_base += n_offsets;
  st->lr = (float)(lrate * pow(0.1, (double)st->global_step / decay_steps));
  st->global_step += 1;
  st->opt_step += 1;
  st->bc1 = (float)(1.0 - powi(b1, st->opt_step));
  st->bc2_sqrt = (float)sqrt(1.0 - powi(b2, st->opt_step));
}

}  // namespace snr

using namespace snr;

extern "C" int snr_abi_version(void) { return SNR_ABI_VERSION; }

extern "C" const char* snr_status_string(int st) {
  switch (st) {
    case SNR_OK: return "ok";
    case SNR_ERR_NULL: return "required pointer is NULL";
    case SNR_ERR_SHAPE: return "size out of supported range";
    case SNR_ERR_UNSUPPORTED: return "configuration not supported by the HIP path";
    default: return st > 0 ? hipGetErrorString((hipError_t)st) : "unknown error";
  }
}

extern "C" int snr_sample_coarse(const float* rays, int ld, int64_t n_rays, int N, int lindisp, const float* t_rand,
                                 float* z_vals, snr_stream_t stream) {
  SNR_CHECK_ARG(rays && z_vals, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && N > 0 && ld >= 8, SNR_ERR_SHAPE);
  const int64_t n = n_rays * N;
  {
    ProfScope ps(K_SAMPLE_COARSE, (hipStream_t)stream);
    sample_coarse_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        rays, ld, n_rays, N, lindisp, t_rand, 0, Rng{}, z_vals);
  }
  return launch_status();
}

extern "C" int snr_composite_forward(const float* raw, int C, const float* z, const float* rays, int ld,
                                     const float* noise, int64_t n_rays, int S, int white, float* rgb_map,
                                     float* disp_map, float* acc_map, float* depth_map, float* weights, float* alpha,
                                     snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && rgb_map && disp_map && acc_map && depth_map && weights, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && S > 0 && C >= 4 && ld >= 6, SNR_ERR_SHAPE);
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_COMPOSITE_FWD, (hipStream_t)stream);
    composite_fwd_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(raw, C, z, rays, ld, NoiseSrc{noise, 0, Rng{}, 0.f},
                                                                            n_rays, S, white, rgb_map, disp_map, acc_map,
                                                                            depth_map, weights, alpha, nullptr);
  }
  return launch_status();
}

extern "C" int snr_composite_backward(const float* raw, int C, const float* z, const float* rays, int ld,
                                      const float* noise, int64_t n_rays, int S, int white, int detach,
                                      const float* g_rgb, const float* g_disp, const float* g_acc,
                                      const float* g_depth, const float* g_w, const float* g_alpha, float* d_raw,
                                      snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && d_raw, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && S > 0 && C >= 4 && ld >= 6, SNR_ERR_SHAPE);
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_COMPOSITE_BWD, (hipStream_t)stream);
    composite_bwd_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(
        raw, C, z, rays, ld, NoiseSrc{noise, 0, Rng{}, 0.f}, n_rays, S, white, detach, g_rgb, g_disp, g_acc, g_depth, g_w,
        g_alpha, d_raw, nullptr, nullptr);
  }
  return launch_status();
}

extern "C" int snr_sample_fine(const float* z_coarse, const float* weights, const float* u, int64_t n_rays, int Nc,
                               int Nf, float* z_out, float* z_samples, float* z_std, snr_stream_t stream) {
  SNR_CHECK_ARG(z_coarse && weights && z_out, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && Nc >= 3 && Nf >= 1 && Nc + Nf <= 4096, SNR_ERR_SHAPE);
  int npow2 = 2;
  while (npow2 < Nc + Nf) npow2 <<= 1;
  const size_t lds = (size_t)kRaysPerBlock * (2 * (Nc - 1) + npow2) * sizeof(float);
  SNR_CHECK_ARG(lds <= 160 * 1024, SNR_ERR_SHAPE);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)sample_fine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_SAMPLE_FINE, (hipStream_t)stream);
    sample_fine_kernel<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(z_coarse, weights, u, n_rays, Nc, Nf, npow2,
                                                                            z_out, z_samples, z_std, 0, 0, Rng{});
  }
  return launch_status();
}

extern "C" int snr_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins,
                              int n_samples, float* samples, snr_stream_t stream) {
  SNR_CHECK_ARG(bins && weights && samples, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && n_bins >= 2 && n_samples > 0, SNR_ERR_SHAPE);
  const int Nc = n_bins + 1;
  int npow2 = 2;
  while (npow2 < Nc + n_samples) npow2 <<= 1;
  const size_t lds = (size_t)kRaysPerBlock * (2 * (Nc - 1) + npow2) * sizeof(float);
  SNR_CHECK_ARG(lds <= 160 * 1024, SNR_ERR_SHAPE);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)sample_fine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_SAMPLE_FINE, (hipStream_t)stream);
    sample_fine_kernel<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(bins, weights, u, n_rays, Nc, n_samples, npow2,
                                                                            nullptr, samples, nullptr, 1, 0, Rng{});
  }
  return launch_status();
}

// Compositing from opacities the caller computed (MVSeg's only_object path post-processes alpha before the
// transmittance product, MVSeg/DS_NeRF/run_nerf_helpers.py:383-397): colours from raw[..., :3], alpha [n_rays,S] as given.
extern "C" int snr_composite_alpha_forward(const float* raw, int C, const float* z, const float* rays, int ld,
                                           const float* alpha, int64_t n_rays, int S, int white, float* rgb_map,
                                           float* disp_map, float* acc_map, float* depth_map, float* weights,
                                           snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && alpha && rgb_map && disp_map && acc_map && depth_map && weights, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && S > 0 && C >= 4 && ld >= 6, SNR_ERR_SHAPE);
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_COMPOSITE_FWD, (hipStream_t)stream);
    composite_fwd_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(raw, C, z, rays, ld, NoiseSrc{nullptr, 0, Rng{}, 0.f},
                                                                            n_rays, S, white, rgb_map, disp_map, acc_map,
                                                                            depth_map, weights, nullptr, alpha);
  }
  return launch_status();
}

/* d_raw [n_rays,S,C]: colour channels only (channel 3 and up get 0); d_alpha [n_rays,S] = d loss / d alpha */
extern "C" int snr_composite_alpha_backward(const float* raw, int C, const float* z, const float* rays, int ld,
                                            const float* alpha, int64_t n_rays, int S, int white, int detach,
                                            const float* g_rgb, const float* g_disp, const float* g_acc,
                                            const float* g_depth, const float* g_w, float* d_raw, float* d_alpha,
                                            snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && alpha && d_raw && d_alpha, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && S > 0 && C >= 4 && ld >= 6, SNR_ERR_SHAPE);
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_COMPOSITE_BWD, (hipStream_t)stream);
    composite_bwd_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(
        raw, C, z, rays, ld, NoiseSrc{nullptr, 0, Rng{}, 0.f}, n_rays, S, white, detach, g_rgb, g_disp, g_acc, g_depth, g_w,
        nullptr, d_raw, alpha, d_alpha);
  }
  return launch_status();
}

static Rng make_rng(uint64_t seed, uint64_t offset, const uint64_t* base = nullptr) {
  return Rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
             (const unsigned long long*)base};
}

int snr::sample_coarse_rng_impl(const float* rays, int ld, int64_t n_rays, int N, int lindisp, uint64_t seed, uint64_t offset,
                                const uint64_t* base, float* z_vals, snr_stream_t stream) {
  SNR_CHECK_ARG(rays && z_vals, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && N > 0 && ld >= 8, SNR_ERR_SHAPE);
  const int64_t n = n_rays * N;
  {
    ProfScope ps(K_SAMPLE_COARSE, (hipStream_t)stream);
    sample_coarse_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        rays, ld, n_rays, N, lindisp, nullptr, 1, make_rng(seed, offset, base), z_vals);
  }
  return launch_status();
}
extern "C" int snr_sample_coarse_rng(const float* rays, int ld, int64_t n_rays, int N, int lindisp, uint64_t seed,
                                     uint64_t offset, float* z_vals, snr_stream_t stream) {
  return snr::sample_coarse_rng_impl(rays, ld, n_rays, N, lindisp, seed, offset, nullptr, z_vals, stream);
}

int snr::sample_fine_rng_impl(const float* z_coarse, const float* weights, int64_t n_rays, int Nc, int Nf, uint64_t seed,
                              uint64_t offset, const uint64_t* base, float* z_out, float* z_samples, float* z_std,
                              snr_stream_t stream) {
  SNR_CHECK_ARG(z_coarse && weights && z_out, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && Nc >= 3 && Nf >= 1 && Nc + Nf <= 4096, SNR_ERR_SHAPE);
  int npow2 = 2;
  while (npow2 < Nc + Nf) npow2 <<= 1;
  const size_t lds = (size_t)kRaysPerBlock * (2 * (Nc - 1) + npow2) * sizeof(float);
  SNR_CHECK_ARG(lds <= 160 * 1024, SNR_ERR_SHAPE);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)sample_fine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  {
    ProfScope ps(K_SAMPLE_FINE, (hipStream_t)stream);
    sample_fine_kernel<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(z_coarse, weights, nullptr, n_rays, Nc, Nf, npow2,
                                                                            z_out, z_samples, z_std, 0, 1,
                                                                            make_rng(seed, offset, base));
  }
  return launch_status();
}
extern "C" int snr_sample_fine_rng(const float* z_coarse, const float* weights, int64_t n_rays, int Nc, int Nf,
                                   uint64_t seed, uint64_t offset, float* z_out, float* z_samples, float* z_std,
                                   snr_stream_t stream) {
  return snr::sample_fine_rng_impl(z_coarse, weights, n_rays, Nc, Nf, seed, offset, nullptr, z_out, z_samples, z_std, stream);
}

int snr::composite_train_impl(const float* raw, int C, const float* z, const float* rays, int ld, const float* noise,
                              float noise_std, uint64_t seed, uint64_t offset, const uint64_t* base, int64_t n_rays, int S,
                              int white, const LossSpec& spec, int final_pass, float* rgb_map,
                              float* disp_map, float* acc_map, float* depth_map, float* weights, float* d_raw, float* loss,
                              snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && rgb_map && disp_map && acc_map && depth_map && weights && d_raw && loss, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && S > 0 && C >= 4 && ld >= 6 && spec.n >= 0 && spec.n <= 4, SNR_ERR_SHAPE);
  for (int k = 0; k < spec.n; ++k) SNR_CHECK_ARG(spec.t[k].target, SNR_ERR_NULL);
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const NoiseSrc ns{noise, (!noise && noise_std > 0.f) ? 1 : 0, make_rng(seed, offset, base), noise_std};
  {
    const int nch = (S + kWave - 1) / kWave;
    // 16-byte raw rows (C == 4) need 16-byte aligned bases for the vector accesses of the register-resident version
    const bool aligned = C != 4 || ((((uintptr_t)raw) | ((uintptr_t)d_raw)) & 15) == 0;
    ProfScope ps(nch <= 4 && aligned ? K_COMPOSITE_TRAIN_REG : K_COMPOSITE_TRAIN, (hipStream_t)stream);
#define SNR_CT_ARGS raw, C, z, rays, ld, ns, n_rays, S, white, spec, final_pass, rgb_map, disp_map, acc_map, depth_map, weights, d_raw, loss
    if (nch == 1 && aligned) composite_train_reg_kernel<1><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(SNR_CT_ARGS);
    else if (nch == 2 && aligned) composite_train_reg_kernel<2><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(SNR_CT_ARGS);
    else if (nch == 3 && aligned) composite_train_reg_kernel<3><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(SNR_CT_ARGS);
    else if (nch == 4 && aligned) composite_train_reg_kernel<4><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(SNR_CT_ARGS);
    else composite_train_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(SNR_CT_ARGS);
#undef SNR_CT_ARGS
  }
  return launch_status();
}
// composite_train of the coarse samples + hierarchical sampling from its weights, one kernel (Nc <= 64, 16-byte aligned raw rows)
int snr::composite_train_sample_impl(const float* raw, int C, const float* z, const float* rays, int ld, const float* noise,
                                     float noise_std, uint64_t seed, uint64_t offset, const uint64_t* base, int64_t n_rays, int Nc,
                                     int white, const LossSpec& spec, float* rgb_map,
                                     float* disp_map, float* acc_map, float* depth_map, float* weights, float* d_raw, float* loss,
                                     const float* u, int use_rng_u, uint64_t offset_u, int Nf, float* z_out, float* z_samples,
                                     float* z_std, snr_stream_t stream) {
  SNR_CHECK_ARG(raw && z && rays && rgb_map && disp_map && acc_map && depth_map && weights && d_raw && loss && z_out,
                SNR_ERR_NULL);
  // coverage limits first (the caller's two-kernel route takes these: ADVICE r05): one wave holds a ray's coarse samples, and the
  // register sort needs an interior bin
  if (Nc > kWave || Nc < 3) return SNR_ERR_UNSUPPORTED;
  SNR_CHECK_ARG(n_rays > 0 && Nf >= 1 && C >= 4 && ld >= 6 && spec.n >= 0 && spec.n <= 4, SNR_ERR_SHAPE);
  for (int k = 0; k < spec.n; ++k) SNR_CHECK_ARG(spec.t[k].target, SNR_ERR_NULL);
  if (C == 4 && ((((uintptr_t)raw) | ((uintptr_t)d_raw)) & 15) != 0) return SNR_ERR_UNSUPPORTED;
  int npow2 = 2;
  while (npow2 < Nc + Nf) npow2 <<= 1;
  const size_t lds = (size_t)kRaysPerBlock * (Nc + 2 * (Nc - 1) + npow2) * sizeof(float);
  if (lds > 48 * 1024) return SNR_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const NoiseSrc ns{noise, (!noise && noise_std > 0.f) ? 1 : 0, make_rng(seed, offset, base), noise_std};
  {
    ProfScope ps(K_COMPOSITE_TRAIN_SAMPLE, (hipStream_t)stream);
    composite_train_sample_kernel<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(
        raw, C, z, rays, ld, ns, n_rays, Nc, white, spec, rgb_map, disp_map, acc_map,
        depth_map, weights, d_raw, loss, u, Nf, npow2, z_out, z_samples, z_std, use_rng_u, make_rng(seed, offset_u, base));
  }
  return launch_status();
}

int snr::loss_guard_impl(float* loss, int slot, int64_t first_ray, int64_t n_rays, float* d_raw0, int64_t row0, float* d_raw,
                         int64_t row1, snr_stream_t stream) {
  SNR_CHECK_ARG(loss, SNR_ERR_NULL);
  SNR_CHECK_ARG(slot > 0 && slot < 4 && first_ray >= 0 && n_rays > 0, SNR_ERR_SHAPE);
  const int64_t n0 = d_raw0 ? n_rays * row0 : 0, n1 = d_raw ? n_rays * row1 : 0;
  const int64_t n = n0 > n1 ? n0 : n1;
  loss_guard_kernel<<<dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)), dim3(256), 0, (hipStream_t)stream>>>(
      loss, slot, d_raw0 ? d_raw0 + first_ray * row0 : nullptr, n0, d_raw ? d_raw + first_ray * row1 : nullptr, n1);
  return launch_status();
}

int snr::pack_rays_sample_impl(const float* rays_o, const float* rays_d, int64_t n_rays, int H, int W, float focal, int ndc,
                               float near, float far, int use_viewdirs, float* rays, int ld, int N, int lindisp,
                               const float* t_rand, int use_rng, uint64_t seed, uint64_t offset, const uint64_t* base, float* z_vals,
                               float* zero, int n_zero, snr_stream_t stream) {
  SNR_CHECK_ARG(rays_o && rays_d && rays && z_vals, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && H > 0 && W > 0 && N > 0 && ld >= 8 + (use_viewdirs ? 3 : 0) && n_zero >= 0 && n_zero <= 256, SNR_ERR_SHAPE);
  const int64_t n = n_rays * N;
  {
    ProfScope ps(K_PACK_RAYS_SAMPLE, (hipStream_t)stream);
    pack_rays_sample_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        rays_o, rays_d, n_rays, H, W, focal, ndc, 1.f, near, far, use_viewdirs, rays, ld, N, lindisp, t_rand, use_rng,
        make_rng(seed, offset, base), z_vals, zero, n_zero);
  }
  return launch_status();
}

extern "C" int snr_composite_train(const float* raw, int C, const float* z, const float* rays, int ld, const float* noise,
                                   float noise_std, uint64_t seed, uint64_t offset, int64_t n_rays, int S, int white,
                                   int detach, const float* target, int64_t n_rays_global, float* rgb_map,
                                   float* disp_map, float* acc_map, float* depth_map, float* weights, float* d_raw,
                                   float* loss, float* loss_also, snr_stream_t stream) {
  SNR_CHECK_ARG(target && loss, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && n_rays_global >= n_rays, SNR_ERR_SHAPE);
  // two independent accumulators (include/spinnerf_hip.h): the kernels add a term to loss[slot] and loss[slot_final], element
  // offsets from one pointer — here slot 0 of `loss` and the signed distance to `loss_also` (integer arithmetic on the addresses:
  // the two need not belong to one allocation)
  SNR_CHECK_ARG(((((uintptr_t)loss) | ((uintptr_t)loss_also)) & 3) == 0, SNR_ERR_SHAPE);
  snr::LossSpec sp = snr::plain_rgb_loss(target, n_rays, n_rays_global);
  sp.t[0].kind = detach ? 1 : 0;
  sp.t[0].slot = 0;
  sp.t[0].slot_final = loss_also ? ((int64_t)(intptr_t)loss_also - (int64_t)(intptr_t)loss) / 4 : snr::kNoSlot;
  return snr::composite_train_impl(raw, C, z, rays, ld, noise, noise_std, seed, offset, nullptr, n_rays, S, white, sp, 1, rgb_map,
                                   disp_map, acc_map, depth_map, weights, d_raw, loss, stream);
}

extern "C" int snr_make_rays(int H, int W, float focal, const float* c2w_host, int i0, int j0, int h, int w, int ndc,
                             float near, float far, int use_viewdirs, float* rays, int ld, snr_stream_t stream) {
  SNR_CHECK_ARG(c2w_host && rays, SNR_ERR_NULL);
  SNR_CHECK_ARG(H > 0 && W > 0 && h > 0 && w > 0 && i0 >= 0 && j0 >= 0 && ld >= (use_viewdirs ? 11 : 8),
                SNR_ERR_SHAPE);
  Pose p;
  for (int i = 0; i < 12; ++i) p.m[i] = c2w_host[i];
  const int64_t n = (int64_t)h * w;
  {
    ProfScope ps(K_MAKE_RAYS, (hipStream_t)stream);
    make_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        H, W, focal, p, i0, j0, h, w, ndc, near, far, use_viewdirs, rays, ld);
  }
  return launch_status();
}

extern "C" int snr_pack_rays(const float* rays_o, const float* rays_d, const float* view_src, int64_t n_rays, int H,
                             int W, float focal, int ndc, float ndc_near, float near, float far, const float* near_rows,
                             const float* far_rows, const float* depths, int use_viewdirs, float* rays, int ld,
                             snr_stream_t stream) {
  SNR_CHECK_ARG(rays_o && rays_d && rays, SNR_ERR_NULL);
  SNR_CHECK_ARG(n_rays > 0 && H > 0 && W > 0 && ld >= 8 + (depths ? 1 : 0) + (use_viewdirs ? 3 : 0), SNR_ERR_SHAPE);
  {
    ProfScope ps(K_PACK_RAYS, (hipStream_t)stream);
    pack_rays_kernel<<<dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        rays_o, rays_d, view_src, n_rays, H, W, focal, ndc, ndc_near, near, far, near_rows, far_rows, depths,
        use_viewdirs, rays, ld);
  }
  return launch_status();
}

extern "C" int snr_embed(const float* x, int64_t n, int n_cols, int multires, float* out, snr_stream_t stream) {
  SNR_CHECK_ARG(x && out, SNR_ERR_NULL);
  SNR_CHECK_ARG(n > 0 && n_cols > 0 && multires >= 0 && multires <= 24, SNR_ERR_SHAPE);
  const int64_t total = n * n_cols * (1 + 2 * multires);
  embed_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, n, n_cols, multires, out);
  return launch_status();
}

extern "C" int snr_mse_pair(const float* a, const float* b, const float* target, int64_t n, float* loss,
                            float* grad_a, float* grad_b, snr_stream_t stream) {
  SNR_CHECK_ARG(a && target && loss && grad_a && (!b || grad_b), SNR_ERR_NULL);
  SNR_CHECK_ARG(n > 0, SNR_ERR_SHAPE);
  mse_pair_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(a, b, target, n, loss, grad_a, grad_b);
  return launch_status();
}

extern "C" int snr_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float b1,
                             float b2, float eps, int step, float gscale, snr_stream_t stream) {
  SNR_CHECK_ARG(params && grads && m && v, SNR_ERR_NULL);
  SNR_CHECK_ARG(n > 0 && step >= 1, SNR_ERR_SHAPE);
  const float bc1 = (float)(1.0 - powi((double)b1, step));
  const float bc2s = (float)sqrt(1.0 - powi((double)b2, step));
  const int64_t threads = (n + 3) / 4;
  {
    ProfScope ps(K_ADAM, (hipStream_t)stream);
    adam_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        params, grads, m, v, n, lr, b1, b2, eps, bc1, bc2s, gscale);
  }
  return launch_status();
}

extern "C" int snr_adam_step_dev(float* params, const float* grads, float* m, float* v, int64_t n, const snr_step_state* state,
                                 float b1, float b2, float eps, float gscale, snr_stream_t stream) {
  SNR_CHECK_ARG(params && grads && m && v && state, SNR_ERR_NULL);
  SNR_CHECK_ARG(n > 0, SNR_ERR_SHAPE);
  const int64_t threads = (n + 3) / 4;
  {
    ProfScope ps(K_ADAM, (hipStream_t)stream);
    adam_dev_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(params, grads, m, v, n, state,
                                                                                                    b1, b2, eps, gscale);
  }
  return launch_status();
}

extern "C" int snr_step_state_advance(snr_step_state* state, double lrate, double decay_steps, float b1, float b2,
                                      uint64_t n_offsets, snr_stream_t stream) {
  SNR_CHECK_ARG(state, SNR_ERR_NULL);
  SNR_CHECK_ARG(decay_steps > 0.0, SNR_ERR_SHAPE);
  step_state_advance_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(state, lrate, decay_steps, (double)b1,
                                                                          (double)b2, (unsigned long long)n_offsets);
  return launch_status();
}
