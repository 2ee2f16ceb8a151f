// "chain2": the bf16 chain kernels (forward, dgrad) of the 8 x 256 NeRF MLP with the vector-memory work taken out of the MFMA
// waves (round 5; paper budget written before the code: profiles/r05_chain2_budget.md).
//
// Same arithmetic, same packed weights, same saved-tensor layouts as mlp_fwd.hip / mlp_bwd.hip (DS_NeRF/run_nerf_helpers.py:
// 104-127 evaluated transposed with v_mfma_f32_32x32x16_bf16, mlp_layout.h) — results are bit-identical to those kernels.
// What changes is who does what inside a workgroup:
//
//   12 waves x <= 168 registers, three per SIMD
//     waves 0..7   COMPUTE  one 32-sample tile each: MFMA chain, epilogues, LDS reads / writes.  No vector-memory instruction
//                           between the loads of the sample's coordinates and the store of its 16 bytes of raw output.
//     waves 8..11  HELPERS  loaders: every LDS-DMA piece of the weight stream (global_load_lds_dwordx4, L2 -> LDS ring);
//                           storers (training forward, dgrad): copy staged output tiles LDS -> HBM.
//
// HYPOTHESIS the structure was built to test: the shipped kernels' +13 ... +32 % (weight DMA) and +20 ... +48 % (saved-tensor
// stores) are time the ISSUING wave spends stuck at a vector-memory instruction with its MFMA chain stopped; a helper stuck
// there stops nobody.  OUTCOME (profiles/r05_chain_ab.txt): refuted.  Results are bit-identical, the kernels are 7-11 % slower:
// a DMA piece costs the same whether a helper issues it or the compute waves do — the weight stream is paid in CLOCK on this
// power-limited part (3 % in cycles, 10 % in effective shader clock with real-valued weights; section 5b of that file) and in
// LDS traffic every fragment read shares —, and the 168-register budget, the spill traffic and the progress words cost a further 8 %.  The code stays behind
// SNR_CHAIN2=1 (default 0) as the experiment's record and A/B switch; tests/test_gpu_chain2.py holds it bit-identical.
//
// Registers: a tile's input (64) and output (64) do not fit 168 next to accumulator, bias and fragment window, so the output
// of a layer is split: output tiles 0..NREG-1 stay in registers, tiles NREG..7 go to a lane-private LDS spill slot (2 x
// ds_write_b128) and come back as the next layer's input fragments (ds_read_b128).  A compute wave never has a vector-memory
// operation in flight inside the layer loop, so its LDS reads are plain C++ (the compiler's "drain vmcnt before an LDS read
// while an LDS-DMA is pending" never triggers there) and the compiler pipelines them.
//
// Weight ring: 4 slots of 16 fragments (16 KiB); block c of the workgroup's life lives in slot c mod 4 (a pass is padded to a
// multiple of 4 blocks, so the slot of every fragment read is an instruction immediate).  No workgroup barrier in the loop
// (first version: one raw s_barrier per block, 80 rendezvous of 12 waves per pass = 23 % of the compute waves' time parked,
// profiles/r05_chain_ab.txt): progress is published through LDS words,
//   landed[p]  += 1 by every loader of group p once its pieces of a block of parity p have landed (two loader groups, one per
//                 block parity: two blocks in flight; a loader waits vmcnt(0) — its only vector-memory operations — and
//                 publishes at once: nothing a compute wave waits for depends on a LATER block)
//   done[w]     = blocks compute wave w has completely read (an LDS write behind its last read of the block: LDS operations
//                 of one wave execute in order)
//   staged[w] / stored[w]   chunks (output tiles) wave w has put into its spill area for the storers / a storer has read back
// a compute wave looks at `landed` only when its cached copy does not already cover the block it is about to read, a loader
// refills the slot of block c-4 when every done[w] >= c-3.  Compute waves never wait for each other; they may drift apart by
// up to three blocks.
#pragma once
#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {
namespace c2 {

#ifndef SNR_C2_ABLATE
#define SNR_C2_ABLATE 0   // timing experiments (results garbage): 1 no weight DMA, 2 no spill traffic, 4 no epilogue, 8 storers drop
#endif                    // their chunks, 16 no helper waves at all (8 waves; nobody waits for a block)
constexpr int kCompute = 8;                 // compute waves = 32-sample tiles per pass
#if SNR_C2_ABLATE & 16
constexpr int kHelpers = 0;
#else
constexpr int kHelpers = 4;
#endif
constexpr int kWaves = kCompute + kHelpers;
constexpr int kBF = 16;                     // fragments per ring block
constexpr int kSlots = 4;
constexpr int kRing2Bytes = kSlots * kBF * 1024;
constexpr int kBiasBytes = 10240;           // >= 2496 floats
#ifndef SNR_C2_NREG
#define SNR_C2_NREG 3
#endif
constexpr int kNReg = SNR_C2_NREG;          // output tiles of a 8-tile layer that stay in registers
constexpr int kSpillTiles = 8 - kNReg;
constexpr int kSpillBytes = kSpillTiles * 2048;          // per compute wave
constexpr int kFlagBytes = 1024;                         // per compute wave (training: the layer's relu flags, staged for the storers)

// ---- the forward weight stream of a network with view directions, in ring blocks (mlp_pack.h: make_pack_table pads every
// entry to kBlockFrags = 32 fragments = 2 ring blocks) ------------------------------------------------------------------------
struct FwdMap {
  // first ring block and ring blocks (padding included) of the eleven stages
  static constexpr int B0 = 0, N0 = 2;                               // pts0: 8 tiles x 4 fragments
  static constexpr int B1 = 2, NH = 8;                               // pts1..4: 8 x 16 each
  static constexpr int B5 = B1 + 4 * NH, N5 = 10;                    // pts5: 8 x (4 + 16)
  static constexpr int B6 = B5 + N5;                                 // pts6, pts7
  static constexpr int B8 = B6 + 2 * NH, N8 = 10;                    // feature (8 tiles) + alpha tile (1) + 1 padding block
  static constexpr int B9 = B8 + N8, N9 = 6;                         // views: 4 x 18 = 72 fragments, padded to 96
  static constexpr int B10 = B9 + N9, N10 = 2;                       // rgb: 8 fragments, padded to 32
  static constexpr int kBlocks = B10 + N10;                          // 78
  static constexpr int kVirtual = (kSlots - kBlocks % kSlots) % kSlots;   // padding blocks that close a pass (nobody reads them): 2
  static constexpr int kPassBlocks = kBlocks + kVirtual;             // 80
};
static_assert(kBlockFrags == 32 && FwdMap::kBlocks == 78 && FwdMap::kPassBlocks % kSlots == 0, "chain2 assumes the 32-fragment stage padding of mlp_pack.h");

// ---- the dgrad weight stream (behind the forward section of the blob), view directions ---------------------------------------
// Only as far as d z1: with selective recompute the weight-gradient pass rebuilds d z0 itself (mlp_wgrad_pair.h), the round-2
// dgrad kernel computes and drops it.
struct BwdMap {
  static constexpr int B0 = 0, N0 = 2;                  // d z9 = relu'(h9) (W_rgb^T d rgb): 4 tiles x 1 fragment, padded to 32
  static constexpr int B1 = 2, N1 = 4;                  // d feat = W_views[:, :256]^T d z9: 8 tiles x 8 fragments
  static constexpr int B2 = 6, N2 = 10;                 // d z7: 8 tiles x (16 + 1) = 136 fragments, padded to 160
  static constexpr int B3 = 16, NH = 8;                 // d z6 ... d z1: six 256 x 256 stages
  static constexpr int kBlocks = B3 + 6 * NH;           // 64 of the stream's 72
  static constexpr int kStreamBlocks = B3 + 7 * NH;
  static constexpr int kPassBlocks = kBlocks;
  static constexpr int kChunks = 1 + 4 + 4 * 8;         // d out, d z9 (4 tiles), d z7 / 5 / 3 / 1 (8 tiles each)
};
static_assert(BwdMap::kPassBlocks % kSlots == 0, "a pass is a whole number of ring turns");

// ---- LDS progress words (inside the bias area's slack) ---------------------------------------------------------------
struct Flags {
  unsigned landed[2];         // blocks of each parity that have landed: += 1 per loader of the parity's group per block
  unsigned pad[14];
  unsigned done[kCompute];    // blocks completely read, per compute wave
  unsigned staged[kCompute];  // chunks (output tiles, ...) a compute wave has put into its spill area for the storers
  unsigned stored[kCompute];  // chunks a storer has read out of it again
};
typedef __attribute__((address_space(3))) volatile unsigned* LdsWord;   // (a generic pointer would make every access a flat_* instruction)
typedef __attribute__((address_space(3))) volatile Flags* LdsFlags;
constexpr int kFlagsOffset = kBiasBytes - (int)sizeof(Flags);
static_assert(kFlagsOffset >= 2496 * 4 && sizeof(Flags) == 160, "the progress words live behind the bias block");

// ---- helper waves: the weight stream -------------------------------------------------------------------------------------
// NL loaders in two groups; group p loads the blocks of parity p, each of its NL/2 loaders half of the block's 16 pieces
// (1 KiB each: SGPR base + lane * 16).  Per block: wait until every compute wave has left the slot's previous tenant
// (block c-4), issue, wait for the pieces (vmcnt(0): a loader's only vector-memory operations), publish.  So a block is
// visible as soon as it has landed, two blocks are in flight, and nothing a compute wave waits for depends on a LATER block.
// A pass is `pass_blocks` blocks; blocks beyond `real_blocks` are padding and re-load the head of the stream (nobody reads them).
template <int NL>
__device__ __forceinline__ void loader_run(char* ring, LdsFlags fl, const char* gbase, int real_blocks, int pass_blocks,
                                           int n_pass, int l, int lane) {
  static_assert(NL == 2 || NL == 4, "two groups of loaders");
  constexpr int PER = NL / 2;        // loaders per group
  constexpr int P = kBF / PER;       // pieces per block per loader
  const int grp = l / PER, li = l % PER;
  const int total = n_pass * pass_blocks;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  int j = grp;                       // position of block c in its pass (pass_blocks is even)
  for (int c = grp; c < total; c += 2) {
    if (c >= kSlots) {               // the slot still holds block c-4
      const unsigned need = (unsigned)(c - kSlots + 1);
#ifndef SNR_C2_LSLEEP
#define SNR_C2_LSLEEP 2   // x 64 cycles between two looks
#endif
      while (__builtin_amdgcn_ballot_w64(fl->done[lane & (kCompute - 1)] < need) != 0) __builtin_amdgcn_s_sleep(SNR_C2_LSLEEP);
      asm volatile("" ::: "memory");
    }
    const char* src = gbase + (int64_t)(j < real_blocks ? j : j - real_blocks) * (kBF * 1024) + li * 1024;
    char* dst = ring + (c & (kSlots - 1)) * (kBF * 1024) + li * 1024;
#pragma unroll
    for (int p = 0; p < P; ++p) {
#if !(SNR_C2_ABLATE & 1)
      asm volatile("s_nop 0");   // mlp_device.h: Pipe::issue_one
      __builtin_amdgcn_global_load_lds(src + (size_t)lane_off, SNR_LDS(dst), 16, 0, 0);
#endif
      src += PER * 1024; dst += PER * 1024;
    }
    j = j + 2 >= pass_blocks ? j + 2 - pass_blocks : j + 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned*)&fl->landed[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// ---- compute waves ---------------------------------------------------------------------------------------------------------
using Frag = bf16x8;

template <int NL> struct Cw {
  const char* ring_lane;   // ring + lane * 16: fragment (slot, pos) of this lane = ring_lane + slot * 16384 + pos * 1024
  char* spill_lane;        // this wave's spill area + lane * 16: tile slot s, half h at (2 s + h) * 1024
  const float* bias_lds;
  LdsFlags fl;
  LdsWord my_done;
  int g;
  unsigned seen[2];        // cached copies of fl->landed (wave-uniform)
  unsigned base;           // blocks of the passes this workgroup has finished (even)
  // staging for the storers (training forward, dgrad): a chunk = what one storer item copies to HBM (an output tile = 2
  // fragments, or a single fragment), written into a spill slot; chunks are numbered in staging order
  LdsWord my_staged, my_stored;
  unsigned n_staged, seen_stored;
  unsigned slot_chunk[kSpillTiles];   // number of the last chunk staged in each slot (0: none)

  // before anything is written into slot S: the storer must have read the chunk staged there
  template <int S> __device__ __forceinline__ void slot_free() {
    if (slot_chunk[S] > seen_stored) {
      for (;;) {
        seen_stored = (unsigned)__builtin_amdgcn_readfirstlane((int)*my_stored);
        if (seen_stored >= slot_chunk[S]) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    asm volatile("" ::: "memory");
  }
  // behind the LDS writes of a chunk in slot S (program order = LDS execution order)
  template <int S> __device__ __forceinline__ void post_chunk() {
    asm volatile("" ::: "memory");
    slot_chunk[S] = ++n_staged;
    *my_staged = n_staged;
  }

  // before the first read of block J of the pass
  template <int J> __device__ __forceinline__ void need_block() {
    const unsigned need = ((base + J) / 2 + 1) * (NL / 2);   // blocks of J's parity up to J, times the loaders of its group
#if SNR_C2_ABLATE & 16
    seen[J & 1] = need;
#endif
    if (seen[J & 1] < need) {
      for (;;) {
        seen[J & 1] = (unsigned)__builtin_amdgcn_readfirstlane((int)fl->landed[J & 1]);
        if (seen[J & 1] >= need) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    asm volatile("" ::: "memory");   // no ring read moves above the look
  }
  // behind the last read of block J-1 of the pass (program order = LDS execution order): J blocks of this pass are read
  template <int J> __device__ __forceinline__ void left_blocks() {
    asm volatile("" ::: "memory");
    *my_done = base + J;
  }

  template <int J, int POS> __device__ __forceinline__ Frag weight() const {
    return *(const Frag*)(ring_lane + (J & (kSlots - 1)) * (kBF * 1024) + POS * 1024);
  }
  template <int SLOT, int H> __device__ __forceinline__ void spill_write(const Frag& f) const {
#if SNR_C2_ABLATE & 2
    asm volatile("" ::"v"(f));
#else
    *(Frag*)(spill_lane + (2 * SLOT + H) * 1024) = f;
#endif
  }
  template <int SLOT, int H> __device__ __forceinline__ Frag spill_read() const {
#if SNR_C2_ABLATE & 2
    Frag f = Mma<kBF16>::zero();
    asm volatile("" : "+v"(f));
    return f;
#else
    return *(const Frag*)(spill_lane + (2 * SLOT + H) * 1024);
#endif
  }

  // fragment i of a stage that starts at block J0: look at `landed` when it opens a block, post `done` when it closes one
  template <int J0, int I> __device__ __forceinline__ Frag fetch() {
    if constexpr (I % kBF == 0) need_block<J0 + I / kBF>();
    const Frag w = weight<J0 + I / kBF, I % kBF>();
    if constexpr (I % kBF == kBF - 1) left_blocks<J0 + I / kBF + 1>();
    return w;
  }

  // One stage: NT output tiles, each KA + KB MFMAs against sa[0..KA) and sb[0..KB), starting at block J0 of the pass;
  // BLOCKS ring blocks belong to the stage (padding included).  epi(nt, acc) is the tile's epilogue.
  template <int J0, int BLOCKS, int KA, int KB, int NT, bool BIAS = true, class Epi>
  __device__ __forceinline__ void stage(const Frag* sa, const Frag* sb, int bias_off, Epi&& epi) {
    constexpr int K = KA + KB, NF = NT * K;
#ifndef SNR_C2_WINDOW
#define SNR_C2_WINDOW 4
#endif
    // fragments run through a window of G registers ahead of the MFMAs, across output tiles
    constexpr int G = SNR_C2_WINDOW < NF ? SNR_C2_WINDOW : NF;
    Frag w[G];
    static_for<0, G>([&](auto I_) { w[decltype(I_)::value] = fetch<J0, decltype(I_)::value>(); });
    // the bias tile (this lane's 16 rows) is read one output tile ahead and is the C operand of the tile's first MFMA
    auto load_bias = [&](int nt) {
      f32x16 r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (!BIAS) return r;   // dgrad: the constant 0 is the first MFMA's C operand
      const float* b = bias_lds + bias_off + 32 * nt + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(b + 8 * q);
        r[4 * q + 0] = v[0]; r[4 * q + 1] = v[1]; r[4 * q + 2] = v[2]; r[4 * q + 3] = v[3];
      }
      return r;
    };
    f32x16 acc, bn = load_bias(0);
    static_for<0, NT>([&](auto NT_) {
      constexpr int nt = decltype(NT_)::value;
      static_for<0, K>([&](auto F_) {
        constexpr int f = decltype(F_)::value;
        constexpr int i = nt * K + f;
        const Frag& src = [&]() -> const Frag& { if constexpr (f < KA) return sa[f]; else return sb[f - KA]; }();
        if constexpr (f == 0) acc = Mma<kBF16>::mma(w[i % G], src, bn);
        else acc = Mma<kBF16>::mma(w[i % G], src, acc);
        if constexpr (i + G < NF) w[i % G] = fetch<J0, i + G>();
        if constexpr (f == 1 && nt + 1 < NT) bn = load_bias(nt + 1);
#if defined(SNR_C2_SGB) && SNR_C2_SGB
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x2, SNR_C2_SGB, 0);     // a few VALU (the previous tile's epilogue)
#endif
      });
#if SNR_C2_ABLATE & 4
      asm volatile("" ::"v"(acc));
#else
      epi(NT_, acc);
#endif
    });
    if constexpr (NF % kBF != 0 || NF / kBF != BLOCKS) left_blocks<J0 + BLOCKS>();   // a partly used last block, padding blocks
  }
};

// ---- helper waves: the storers ---------------------------------------------------------------------------------------------
// A storer serves kCompute / 2 compute waves.  Its program is the static chunk list of a pass (Table::chunk(c): section,
// first fragment inside the section's tile, fragments, spill slot); per chunk and wave: wait until it is staged, read it
// (lane-linear, as the compute wave wrote it), post `stored` behind the reads, store it non-temporally in the saved-tensor
// layout (mlp_device.h: store_tile_slice — fragment f of a tile at f KiB, odd fragments with the act_row swizzle).
struct Chunk { int sec, f0, n, slot; };

template <class Table, class SecBase>
__device__ __forceinline__ void storer_run(const char* spill0, LdsFlags fl, int si, int lane, int n_pass, SecBase&& sec_base) {
  constexpr int PERW = kCompute / 2;
  const int sj = lane & 31, g = lane >> 5;
  const uint32_t lane_even = g * 16 + sj * 32, lane_odd = g * 16 + (sj ^ 4) * 32;   // act_row<kBF16>
  unsigned seen[PERW] = {};
  for (int pass = 0; pass < n_pass; ++pass) {
    static_for<0, Table::kChunks>([&](auto C_) {
      constexpr Chunk ch = Table::chunk(decltype(C_)::value);
      const unsigned need = (unsigned)(pass * Table::kChunks + decltype(C_)::value + 1);
      static_for<0, PERW>([&](auto W_) {
        constexpr int wi = decltype(W_)::value;
        const int w = si * PERW + wi;
        if (seen[wi] < need) {
          for (;;) {
            seen[wi] = (unsigned)__builtin_amdgcn_readfirstlane((int)fl->staged[w]);
            if (seen[wi] >= need) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        asm volatile("" ::: "memory");
        const char* src = spill0 + w * kSpillBytes + ch.slot * 2048 + lane * 16;
        Frag v[2];
        v[0] = *(const Frag*)src;
        if constexpr (ch.n == 2) v[1] = *(const Frag*)(src + 1024);
        asm volatile("" ::: "memory");
        fl->stored[w] = need;                        // (LDS executes a wave's operations in order: the reads are done)
        char* dst = sec_base(ch.sec, pass, w) + ch.f0 * 1024;
#if SNR_C2_ABLATE & 8   // timing experiment: the storers read the chunks and drop them
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(dst));
#else
        __builtin_nontemporal_store(v[0], (Frag*)(dst + ((ch.f0 & 1) ? lane_odd : lane_even)));
        if constexpr (ch.n == 2) __builtin_nontemporal_store(v[1], (Frag*)(dst + 1024 + (((ch.f0 + 1) & 1) ? lane_odd : lane_even)));
#endif
      });
    });
  }
}

}  // namespace c2
}  // namespace snr
