// Fused vocabulary projection + softmax + NLL, forward AND backward, for H = 512 on gfx950 -- no [T, V] matrix of
// logits or probabilities ever reaches HBM (decoders.py:191-193 + criterions.py:55-64 of the reference, and their
// backward).  Two flash-attention-shaped kernels, each two chained MFMA products per (32 rows x 32 columns) block:
//
//   vocab_fwd_dh_kernel   S^T = W h^T  ->  online softmax over the vocabulary  ->  O^T += W^T P^T      (per 128-row group)
//                         gives lse[t], the target logit, and U[t] = sum_v softmax[t, v] W[v] - W[y_t], the gradient of the
//                         NLL with respect to h[t] for a unit upstream gradient: dH needs no pass of its own in backward
//   vocab_dw_kernel       S = h W^T    ->  P = g (softmax - onehot)          ->  Z += P^T h            (per 128-column tile)
//                         gives dW and dbias
//
// Structure (both): one 256-thread workgroup per CU, ONE wave per SIMD with the whole 512-register file: a wave owns a
// 32 x 512 fp32 accumulator (256 registers) and keeps its stationary operand -- its 32 rows of h, or its 32 rows of W -- as
// 32 MFMA fragments in another 128 registers; the streamed operand (32 x 512 bf16 tiles of W, or of h) is staged
// global -> LDS with global_load_lds_dwordx4 into a 2-deep ring and is read twice: by rows (ds_read_b128, the first
// product) and by columns (ds_read_b64_tr_b16, the second product) from ONE image -- four 128-column panels of 256-byte
// rows with the XOR swizzle that serves both kinds of read.  The first product's accumulator IS the second product's
// operand (v_mfma_f32_32x32x16_bf16: the accumulator's row index is the next product's k index), so probabilities go from
// registers to registers.  Staged bytes per flop: 1/256 (the 128 x 128 tile GEMMs this replaces: 1/64).
//
// Work split: "stream-K" over the linear item space (unit-major: row group / column tile; item: one 32-wide slice of the
// streamed operand): workgroup w of G takes items [w N / G, (w+1) N / G), so every CU gets the same number of items
// whatever the live row count is (a device scalar: the launch geometry is static, HIP-graph capturable).  A workgroup's
// share of a unit is a SEGMENT; every segment leaves a partial result in a slab of the workspace and a second kernel
// folds a unit's segments in workgroup order (fixed order: bitwise reproducible, no atomics).
#include <type_traits>
#include <utility>
#include "common.h"
#include "dropout.h"
#include "../../include/pevae_hip.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* ltr_t;

constexpr int HD = 512;                 // hidden width the kernels are built for (16 column blocks of 32)
constexpr int NCB = HD / 32;            // 16 accumulator blocks per wave
constexpr int NKS = HD / 16;            // 32 k-steps of the first product
constexpr int SUB_BYTES = 32 * HD * 2;  // one sub-tile: 32 rows x 512 bf16 = 32 KiB, four panels of 8 KiB
constexpr int ITEM = 64;                // rows of the streamed operand per item: two sub-tiles
constexpr int TILE_BYTES = 2 * SUB_BYTES;
constexpr int AUX_BYTES = 1024;         // per-tile side data (bias slice, or lse / gscale / targets of 64 rows)
constexpr int BUF_BYTES = TILE_BYTES + AUX_BYTES;
constexpr int SMEM_BYTES = 2 * BUF_BYTES;
constexpr int UNIT = 128;               // rows (forward) / columns (dW) a workgroup's accumulators cover: 4 waves x 32
constexpr float NEG_BIG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 28.f;     // log2 units: a row's maximum may exceed the one in use by this much before O is rescaled

// Diagnostic build only (-DPV_VOCAB_STAMPS): every wave accumulates shader-clock cycles per section of its item loop
// (s_memtime; the stamp drains the LDS queue, so read the SHARES, never this build's run time) into a table of its own
// that no kernel reads; tools/vocab_stamps.py copies it out with pv_vocab_stamps_copy.
#ifdef PV_VOCAB_STAMPS
constexpr int NSTAMP = 8;
__device__ unsigned long long g_vstamps[2][256][4][NSTAMP];
#define PV_STAMP_DECL unsigned long long st_acc[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0; \
  { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define PV_STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
  st_acc[i] += t_ - st_last; st_last = t_; }
#define PV_STAMP_FLUSH(k) if (lane == 0 && blockIdx.x < 256) { _Pragma("unroll") for (int i_ = 0; i_ < NSTAMP; ++i_) g_vstamps[k][blockIdx.x][w][i_] = st_acc[i_]; }
#else
#define PV_STAMP_DECL
#define PV_STAMP(i)
#define PV_STAMP_FLUSH(k)
#endif

struct Segs {                            // a workgroup's item range and how units map to its slabs
  long beg, end;
};
__host__ __device__ __forceinline__ long seg_beg(long w, long N, int G) { return w * N / G; }

// Which stream-K range a workgroup takes.  Ranges are contiguous in the (unit, item) space, so range k starts at item
// phase(k) = seg_beg(k) mod `period` of its unit and walks the streamed operand cyclically from there.  Left in launch
// order, the 32 workgroups of an XCD start at 32 phases spread over the whole operand (7 MB of h, 20 MB of W) and every
// tile is fetched into that XCD's 4 MB L2 once per workgroup: FETCH_SIZE showed 1.13 GB per dW launch, 157 column tiles x
// 7.2 MB.  Here the ranges are sorted by phase and dealt out in runs of G/8 to the XCDs (workgroups with equal blockIdx.x
// % 8 share an XCD under round-robin dispatch): an XCD's workgroups then walk within period/8 items of each other and a
// tile is fetched once per XCD and pass.  A permutation only: the split, the slabs and the fold order are unchanged, and
// placement is speed, not correctness.  `scratch`: G + 1 ints of LDS, free again on return.
__device__ __forceinline__ int phase_sorted_range(long N, int G, int period, int* scratch) {
  if ((G & 7) || N < G || period <= 0) return blockIdx.x;
  const int tid = threadIdx.x, want = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  for (int t = tid; t < G; t += blockDim.x) scratch[t] = (int)(seg_beg(t, N, G) % period);
  __syncthreads();
  for (int t = tid; t < G; t += blockDim.x) {
    const int ph = scratch[t];
    int rank = 0;
    for (int u = 0; u < G; ++u) {
      const int pu = scratch[u];
      rank += (pu < ph || (pu == ph && u < t)) ? 1 : 0;
    }
    if (rank == want) scratch[G] = t;
  }
  __syncthreads();
  const int k = scratch[G];
  __syncthreads();
  return k;
}

// ---- the streamed tile: TWO 32-row sub-tiles (A, B) of 32 KiB, each four 128-column panels of 256-byte rows ---------------
// image of a sub-tile: off(panel, row, chunk) = panel * 8192 + row * 256 + 16 * (chunk ^ swz(row)), swz(row) = ((row & 3) << 2) | ((row >> 2) & 3)
// Staging: wave w fills panel w of both sub-tiles with 16 LDS-DMA pieces of 4 rows (buffer_load_dwordx4 ... lds; the whole
// byte offset sits in the VGPR operand -- the SGPR offset of a buffer instruction is NOT range-checked -- so rows past the
// matrix are outside the descriptor and arrive as zeros: no address is clamped) -- issued ONE piece at a time between the
// MFMAs of the item before (an LDS-DMA piece costs its wave 60-90 cycles of issue; back to back at the head of an item they
// were 15 % of the kernel).
// The buffer descriptor as four plain words (base, base high | stride 0, bytes, format word): the asm below wants it in
// SGPRs, and with an opaque descriptor type hipcc once kept a loop-invariant descriptor in VGPRs under scalar-register
// pressure (the assembler then rejects the instruction).  As words each one can be broadcast where it is used.
typedef unsigned int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, long bytes) {
  const unsigned long long b = (unsigned long long)p;
  return rsrc_t{(unsigned)b, (unsigned)(b >> 32) & 0xffffu, (unsigned)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000u};
}
// One LDS-DMA piece, issued from asm: hipcc must assume that an LDS-DMA builtin in flight aliases every later LDS read of the
// one dynamic LDS array and puts s_waitcnt vmcnt(0) in front of those reads -- each piece then cost its wave a full
// L2 round trip (~150 cycles, 30 % of the weight-gradient kernel by timing ablation).  From asm the compiler sees neither the
// load nor its LDS write; the kernels' own COUNTED waits (half_tile_landed) order the data.  M0 carries the wave-uniform LDS
// destination and is restored (the compiler does not expect it to change).  SIZE: 16 or 4 bytes per lane.
template <int SIZE>
__device__ __forceinline__ void lds_dma(rsrc_t src, int voff, const void* lds_dst) {
  // (wave-uniform by construction; the explicit broadcast keeps it in an SGPR where hipcc's divergence analysis gives up)
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lptr_t)lds_dst);
  src = rsrc_t{(unsigned)__builtin_amdgcn_readfirstlane(src.x), (unsigned)__builtin_amdgcn_readfirstlane(src.y),
               (unsigned)__builtin_amdgcn_readfirstlane(src.z), (unsigned)__builtin_amdgcn_readfirstlane(src.w)};
  unsigned keep;
  if constexpr (SIZE == 16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
}

struct StageLane {     // per lane: row part and chunk part of the source offset of a piece
  int rowpart;         // (lane >> 4) * ld_bytes + w * 256
  int chbase;          // (lane & 15) ^ ((lane >> 4) << 2)
};
__device__ __forceinline__ StageLane make_stage_lane(int lane, int w, int ld_bytes) {
  return StageLane{(lane >> 4) * ld_bytes + w * 256, (lane & 15) ^ ((lane >> 4) << 2)};
}
template <int I>       // piece I of 16: sub-tile I >> 3, rows 4 (I & 7) .. + 3; off_tile: byte offset of the tile's first row
__device__ __forceinline__ void stage_piece(rsrc_t src, const StageLane& sl, int ld_bytes, int off_tile, char* buf, int w) {
  int cb = sl.chbase, rp = sl.rowpart;
  asm volatile("" : "+v"(cb), "+v"(rp));     // (recomputed per piece: sixteen more loop-invariant registers would be spilled)
  const int voff = rp + ((cb ^ (I & 3)) << 4) + (off_tile + (32 * (I >> 3) + 4 * (I & 7)) * ld_bytes);
#ifdef PV_ABL_NOPIECE                    // timing ablation: the piece's address arithmetic without the instruction
  asm volatile("" :: "v"(voff));
#else
  lds_dma<16>(src, voff, buf + (I >> 3) * SUB_BYTES + w * 8192 + (I & 7) * 1024);
#endif
}

// ---- fragment reads.  A tile's LDS addresses are a per-lane base XOR a small constant (the swizzle) plus immediates:
//   row read, k-step ks (A operand of the first product; lane (r = lane & 31, hh = lane >> 5) gets row r, k = 16 ks + 8 hh ..):
//       (ks >> 3) * 8192 + [256 r | 16 (swz(r) ^ hh)] ^ 32 (ks & 7)
//   column read, column block c, k-step s, part u (operand of the second product: the 8 tile rows
//   16 s + 8 (j >> 2) + 4 hh + (j & 3), j = 0..7, of column 32 c + (lane & 31) -- the k order in which an accumulator of the
//   first product hands its rows to the next MFMA; lane: g = lane >> 4, hh = g >> 1, cg = g & 1, q = (lane & 15) >> 2, p = lane & 3):
//       (c >> 2) * 8192 + 4096 s + [256 (8 u + 4 hh + q) + 8 (p & 1) | 16 ((2 cg + (p >> 1)) ^ ((q << 2) | (2 u + hh)))] ^ 64 (c & 3)
// Held as registers, the 8 + 8 XOR variants are loop-invariant, and hipcc spilled them (13 scratch reloads at the head of
// every item).  The stream therefore walks the k-steps and the column blocks VARIANT-major -- four consecutive reads share
// one XOR constant and differ by immediates -- and derives the variant's address from the base when it changes (FragAddr;
// the empty asm keeps hipcc from hoisting the eight variants out of the loop again).
struct FragBase { int row, col0, col1; };      // the bracketed per-lane terms above, plus the tile's LDS offset
struct FragAddr { int row, col0, col1; };      // the current variant's addresses
__device__ __forceinline__ FragBase make_frag_base(int lane) {
  const int r32 = lane & 31, hh = lane >> 5, g = lane >> 4, cg = g & 1, q = (lane & 15) >> 2, p = lane & 3;
  const int lp = 2 * cg + (p >> 1);
  FragBase b;
  b.row = 256 * r32 + 16 * ((((r32 & 3) << 2) | ((r32 >> 2) & 3)) ^ hh);
  b.col0 = 256 * (4 * hh + q) + 8 * (p & 1) + 16 * (lp ^ ((q << 2) | hh));
  b.col1 = 256 * (8 + 4 * hh + q) + 8 * (p & 1) + 16 * (lp ^ ((q << 2) | (2 + hh)));
  return b;
}
// step N of an item (0..127) -> what its MFMA multiplies (the stream's layout: see "An item" below).  First product: k-step
// ks = 8 (n & 3) + (n >> 2) (variant-major); second product: k-step s = n >> 4, column block c = 4 (n & 3) + ((n & 15) >> 2)
template <int N> struct StepMap {
  static constexpr int phase = N >> 5, n = N & 31;
  static constexpr bool first = (phase & 1) == 0;
  static constexpr int sub = phase == 0 ? 0 : phase == 1 ? 1 : phase == 2 ? 1 : 0;   // the sub-tile whose LDS image is read
  static constexpr bool prev = phase == 1;                                         // ... of the tile BEFORE this item's
  static constexpr int ks = 8 * (n & 3) + (n >> 2);
  static constexpr int s = n >> 4, c = 4 * (n & 3) + ((n & 15) >> 2);
};
// dprev: byte distance from this item's LDS buffer to the other one (wave-uniform)
template <int N>
__device__ __forceinline__ bf16x8 item_frag(const char* lds, const FragBase& fb, int dprev, FragAddr& fa) {
  using M = StepMap<N>;
#ifdef PV_ABL_NOLDS                      // timing ablation: no fragment reads at all (results are garbage)
  return __builtin_bit_cast(bf16x8, uint4{(unsigned)fb.row, (unsigned)fb.col0, (unsigned)fb.col1, (unsigned)N});
#endif
  if constexpr (M::first) {
    if constexpr ((M::n & 3) == 0) {
      int t = fb.row;
      asm volatile("" : "+v"(t));
      fa.row = t ^ (32 * (M::n >> 2));
    }
    return __builtin_bit_cast(bf16x8, *(const uint4*)(lds + fa.row + M::sub * SUB_BYTES + (M::n & 3) * 8192));
  } else {
    if constexpr ((M::n & 3) == 0) {
      int t0 = fb.col0, t1 = fb.col1;
      asm volatile("" : "+v"(t0), "+v"(t1));
      fa.col0 = (t0 ^ (64 * ((M::n & 15) >> 2))) + (M::prev ? dprev : 0);
      fa.col1 = (t1 ^ (64 * ((M::n & 15) >> 2))) + (M::prev ? dprev : 0);
    }
    constexpr int imm = M::sub * SUB_BYTES + (M::n & 3) * 8192 + 4096 * M::s;
    const bf16x4 p0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(lds + fa.col0 + imm));
    const bf16x4 p1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(lds + fa.col1 + imm));
    return bf16x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
  }
}
// the second product of a segment's LAST sub-tile (B of the buffer at `fb`), which no later item carries: step n of 32
template <int n>
__device__ __forceinline__ bf16x8 tail_frag(const char* lds, const FragBase& fb, FragAddr& fa) {
  if constexpr ((n & 3) == 0) {
    int t0 = fb.col0, t1 = fb.col1;
    asm volatile("" : "+v"(t0), "+v"(t1));
    fa.col0 = t0 ^ (64 * ((n & 15) >> 2));
    fa.col1 = t1 ^ (64 * ((n & 15) >> 2));
  }
  constexpr int imm = SUB_BYTES + (n & 3) * 8192 + 4096 * (n >> 4);
  const bf16x4 p0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(lds + fa.col0 + imm));
  const bf16x4 p1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ltr_t)(lds + fa.col1 + imm));
  return bf16x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
}

// max over the two lanes (l, l + 32) that share a row of the 32x32 accumulator layout: one v_permlane32_swap, no LDS
__device__ __forceinline__ float max_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// The first product's accumulator must sit in VGPRs: the transform reads it with VALU instructions, and all 256 AGPRs belong
// to the second product's accumulators.  Left to hipcc (the builtin), it is placed in AGPRs, the second product's
// accumulators no longer fit and the kernel moves them in and out with thousands of v_accvgpr copies and spills; so this one
// MFMA is written as asm with VGPR operands.  Nothing inside asm is padded by the assembler: FIRST (the chain's head, whose C
// input a VALU or LDS instruction has just written) waits two states in front; mfma_v_done, behind the chain's tail, waits
// the 18 states a 16-pass MFMA needs before VALU instructions may read its result.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int HEAD>      // 0: X += a b;  1: the chain's head, X (just written) += a b;  2: the chain's head, X = a b
__device__ __forceinline__ void mfma_v(f32x16& X, const bf16x8& a, const bf16x8& b) {
  const u32x4 av = __builtin_bit_cast(u32x4, a), bv = __builtin_bit_cast(u32x4, b);
  if constexpr (HEAD == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(X) : "v"(av), "v"(bv));
  else if constexpr (HEAD == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(X) : "v"(av), "v"(bv));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(X) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void mfma_v_done(f32x16& X) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(X)); }
// The second product's accumulators, pinned to AGPRs the same way (under register pressure hipcc otherwise keeps some of the
// sixteen blocks in VGPRs, uses the freed AGPRs for LDS fragments and spills the stationary operand: 372 spilled registers in
// the weight-gradient kernel).  No wait states in front: the operands are LDS fragments (the compiler has waited for them) and
// v_cvt_pk results that are at least one whole step old -- a step's VALU fillers come AFTER its MFMA, and the next step opens
// with its LDS requests and the s_waitcnt, more than the two states a VALU-written operand needs.  (An s_nop 1 in front of every
// MFMA was 8 of the 32 issue cycles a step has: one wave per SIMD is bound by its own instruction issue.)
// mfma_a_done: before anything but an MFMA touches the accumulators.
__device__ __forceinline__ void mfma_a(f32x16& ACC, const bf16x8& a, const bf16x8& b) {
  const u32x4 av = __builtin_bit_cast(u32x4, a), bv = __builtin_bit_cast(u32x4, b);
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(av), "v"(bv));
}
__device__ __forceinline__ void mfma_a_done() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}
#define PV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// "this half-tile has landed": every wave waits for its own pieces of it -- all but the 8 pieces (of the NEXT half-tile) it has
// issued since -- and then the workgroup meets.  A raw s_barrier: __syncthreads() would add vmcnt(0) and drain the stream.
__device__ __forceinline__ void half_tile_landed() {
  PV_SCHED_FENCE();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PV_SCHED_FENCE();
}
// which first-product step q (0..31) of wave w issues staging piece q / 4: the four waves of a workgroup run in lockstep
// (one barrier per item) and share the CU's one vector-memory address unit, so their pieces go out at four different steps
// Staging schedule.  A CU takes in LDS-DMA pieces at about one per 37 cycles (45 GB/s), and a wave that issues a piece into
// a full queue stands still -- MFMAs included -- until there is room: 8 pieces per 32 steps from each of four waves ran at
// that limit and cost 28 % of the kernel (timing ablations).  So the 16 pieces of a tile go out one per 7 steps over the
// WHOLE item: during the 64 steps of sub-tile A the 8 pieces of the next tile's sub-tile A (needed 69+ steps later, at the
// next item's head), during sub-tile B's the 8 pieces of the next tile's sub-tile B (needed 58+ steps later, when the next
// item's stream reaches its first B fragment).  Two "half-tile landed" meetings per item, each behind a COUNTED wait.
// (tried: staggered by wave behind a wave-uniform branch per step -- 375 -> 415 us, the branches cost more than they saved)
#define PV_STAGE_AT(q) ((q) % 7 == 2 && (q) < 52)

// An item = 128 MFMA steps around one streamed tile (sub-tiles A, B), as ONE stream; the second product runs ONE sub-tile
// behind the first, so that each sub-tile's transform (VALU) sits in the shadow of the sub-tile-before's second product:
//   steps   0.. 31   X  = rows(A)[ks] x stat[ks]                           first product of A    | 8 pieces: next tile's A
//   (between)        decision for A (mask / target pick / row maximum / rare rescale)             the one section without MFMAs
//   steps  32.. 63   ACC[c] += cols(B of the tile BEFORE)[c, s] (x) P_B'   second product of B'  | transform X -> P_A
//   steps  64.. 95   X  = rows(B)[ks] x stat[ks]                           first product of B    | 8 pieces: next tile's B
//   (between)        decision for B
//   steps  96..127   ACC[c] += cols(A)[c, s] (x) P_A                       second product of A   | transform X -> P_B
// and the last sub-tile of a segment is multiplied in by a 32-step tail behind the item loop.  A segment starts with P_B' = 0
// (its first second product adds zeros from whatever finite tile the other buffer holds).  One accumulator X and two sets
// of probabilities are live (8 registers more than without the lag; both sub-tiles' X at once does not fit: hipcc spilled).
// The LDS fragment of step i is requested at step i - DEPTH (across the phase boundaries: an LDS round trip under load is
// several MFMAs long), and every step is fenced (sched_barrier) so that each MFMA keeps its few fillers -- one LDS request,
// now and then a piece of the next tile's staging, a slice of the transform -- in its own 32-cycle shadow: one wave per SIMD
// has nobody else to fill the matrix pipe's issue gaps.
#ifndef PV_DEPTH
#define PV_DEPTH 6
#endif
constexpr int DEPTH = PV_DEPTH;
constexpr int DECIDE_AT = 34;      // the step of a second product beside which the forward kernel decides about the maximum
// =====================================================================================================================
// forward + dH:  per row t   lse, target logit, O[t] = sum_v exp(s_tv - m) W[v]  (partial over a vocabulary range)
// =====================================================================================================================
struct FwdArgs {
  const bf16_t* h;      // [T][ldh]   compact rows (live first)
  const bf16_t* W;      // [V][ldw]
  const float* bias;    // [V]
  const int* target;    // [T] int32, compact order
  const int* live;      // device scalar: rows to compute (NULL: all T)
  int T, V, ldh, ldw, max_seg;
  float inv_temp;
  float* o_slab;        // [G * max_seg][HD][UNIT]
  float4* stats;        // [G * max_seg][UNIT]  (m, l, target logit, -)
};

__global__ __launch_bounds__(256, 1) void vocab_fwd_dh_kernel(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NVI = (a.V + ITEM - 1) / ITEM, NRG = (nlive + UNIT - 1) / UNIT;
  const long N = (long)NRG * NVI;
  const int rid = __builtin_amdgcn_readfirstlane(phase_sorted_range(N, G, NVI, (int*)smem));
  const long beg = seg_beg(rid, N, G), end = seg_beg(rid + 1, N, G);
  const FragBase fb0 = make_frag_base(lane);
  const int ldb = a.ldw * 2;
  const rsrc_t wsrc = make_rsrc(a.W, (long)a.V * ldb), bsrc = make_rsrc(a.bias, (long)a.V * 4);
  const StageLane sl = make_stage_lane(lane, w, ldb);
  const float c_l2 = a.inv_temp * LOG2E;
  int seg = 0, par = 0;
  PV_STAMP_DECL
  if (beg < end) {                               // the first tile of this workgroup's range (later ones ride in the stream)
    char* buf = smem;
    const int v0 = (int)(beg % NVI) * ITEM;
    // (in the stream's order: sub-tile A, side data, sub-tile B -- the counted waits rely on it)
    static_for<0, 8>([&](auto I) { stage_piece<decltype(I)::value>(wsrc, sl, ldb, v0 * ldb, buf, w); });
    if (w == 0) lds_dma<4>(bsrc, (v0 + lane) * 4, buf + TILE_BYTES);
    static_for<8, 16>([&](auto I) { stage_piece<decltype(I)::value>(wsrc, sl, ldb, v0 * ldb, buf, w); });
    // the first second product of a segment multiplies the OTHER buffer's sub-tile B by zeros: it must hold finite numbers
    for (int i = tid; i < SUB_BYTES / 16; i += 256) *(uint4*)(smem + BUF_BYTES + SUB_BYTES + 16 * i) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  for (long at = beg; at < end; ++seg) {
    const int rg = (int)(at / NVI), vi0 = (int)(at % NVI);
    const int nvi = (int)min((long)NVI - vi0, end - at);
    at += nvi;
    // ---- this wave's 32 rows of h as 32 B-operand fragments (lane: row r32, k = 16 ks + 8 hh ..) ----------------------
    const int t_mine = rg * UNIT + 32 * w + r32;
    const bf16_t* hrow = a.h + (long)min(t_mine, a.T - 1) * a.ldh + 8 * hh;
    bf16x8 hf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(hrow + 16 * ks));
    int y = a.target[min(t_mine, a.T - 1)];
    // The loads above must be waited for HERE: left alone, hipcc puts their vmcnt waits at the first uses, inside the item
    // loop, where they run every iteration -- and vmcnt also counts the stream's LDS-DMA pieces, which the compiler does not
    // know about: every item would drain the staging pipeline.  (An empty asm that reads the registers forces the wait.)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { u32x4 t_ = __builtin_bit_cast(u32x4, hf[ks]); asm volatile("" : "+v"(t_)); hf[ks] = __builtin_bit_cast(bf16x8, t_); }
    asm volatile("" : "+v"(y));
    f32x16 O[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) O[c][j] = 0.f;
    // running maximum (an INTEGER in log2 units: a rescale is then an exact power of two, also for probabilities that are
    // already rounded to bf16), the sum of exponentials at that maximum, the target logit (log2 units)
    float m_used = NEG_BIG, l_sum = 0.f, tl = 0.f;
    bf16x8 pfA[2], pfB[2];                        // probabilities of sub-tile A (this item) / B (the item before)
#pragma unroll
    for (int e = 0; e < 8; ++e) { pfB[0][e] = (__bf16)0.f; pfB[1][e] = (__bf16)0.f; }
    PV_STAMP(0)
    for (int k = 0; k < nvi; ++k, par ^= 1) {
      half_tile_landed();       // sub-tile A and the side data of this item's tile; the other buffer's sub-tile A is free
      PV_STAMP(1)
      const char* tile = smem + par * BUF_BYTES;
      char* nbuf = smem + (par ^ 1) * BUF_BYTES;
      const float* bias_t = (const float*)(tile + TILE_BYTES);
      const int v0 = (vi0 + k) * ITEM;
      // the tile the stream stages while it computes: the next of this segment, else the first of the next segment (vocabulary
      // row 0), else nothing (rows >= V are outside the descriptor: no fetch)
      const int nv0 = k + 1 < nvi ? v0 + ITEM : (at < end ? 0 : NVI * ITEM);
      const int noff = nv0 * ldb;
      const int dprev = par ? -BUF_BYTES : BUF_BYTES;
      bf16x8 f[8];
      f32x16 XA, XB;
      // the bias rides in as the initial accumulator: X = bias + W h^T needs no add pass (element j of a lane: row
      // 8 (j >> 2) + 4 hh + (j & 3) of the sub-tile)
      auto init_x = [&](f32x16& X, int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 b4 = *(const f32x4*)(bias_t + 32 * sub + 8 * g4 + 4 * hh);
#pragma unroll
          for (int i = 0; i < 4; ++i) X[4 * g4 + i] = b4[i];
        }
      };
      init_x(XA, 0);
      const FragBase fb{fb0.row + par * BUF_BYTES, fb0.col0 + par * BUF_BYTES, fb0.col1 + par * BUF_BYTES};
      FragAddr fa;
      static_for<0, DEPTH>([&](auto I) { f[decltype(I)::value & 7] = item_frag<decltype(I)::value>(smem, fb, dprev, fa); });
      PV_SCHED_FENCE();

      // one step of the stream: request the fragment of step i + DEPTH, run MFMA i, now and then issue a staging piece
      auto mfma_step = [&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        using M = StepMap<i>;
        if constexpr (i + DEPTH == 64) half_tile_landed();       // sub-tile B, ahead of the stream's first read of it
        if constexpr (i + DEPTH < 128) f[(i + DEPTH) & 7] = item_frag<i + DEPTH>(smem, fb, dprev, fa);
        if constexpr (M::phase == 0) mfma_v<(M::n == 0 ? 1 : 0)>(XA, f[i & 7], hf[M::ks]);
        else if constexpr (M::phase == 1) mfma_a(O[M::c], f[i & 7], pfB[M::s]);
        else if constexpr (M::phase == 2) mfma_v<(M::n == 0 ? 1 : 0)>(XB, f[i & 7], hf[M::ks]);
        else mfma_a(O[M::c], f[i & 7], pfA[M::s]);
#ifndef PV_ABL_NOSTAGE
        if constexpr (PV_STAGE_AT(i & 63)) stage_piece<8 * (i >> 6) + (i & 63) / 7>(wsrc, sl, ldb, noff, nbuf, w);
#endif
        if constexpr (i == 54) {                                 // the next tile's bias slice, behind its sub-tile A
          if (w == 0) lds_dma<4>(bsrc, (nv0 + lane) * 4, nbuf + TILE_BYTES);
        }
      };
      // element J of an accumulator -> probability at the maximum in use, its row sum, bf16
      auto exp_elem = [&](f32x16& X, bf16x8 (&pf)[2], auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const float p = __builtin_amdgcn_exp2f(fmaf(X[j], c_l2, -m_used));   // <= 2^RESCALE_THR; masked columns: 2^(-huge) = 0
        l_sum += p;
        pf[j >> 3][j & 7] = (__bf16)p;
      };
      // what stands between a sub-tile's first product and its transform: no MFMA runs beside this section
      auto decide = [&](f32x16& X, int vs, bf16x8 (&pending)[2], bool nothing_yet) __attribute__((always_inline)) {
        // (no wait for X here: two further MFMA steps -- 64+ cycles of the matrix pipe -- have been issued since its last one)
#ifdef PV_DBG_WAITX
        mfma_v_done(X);
#endif
        // ---- rare, wave-uniform: columns past V (the last tile) and the target pick (a row of this wave has its target here)
        if (__builtin_expect(vs + 32 > a.V || __any((unsigned)(y - vs) < 32u), 0)) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int v = vs + 8 * (j >> 2) + 4 * hh + (j & 3);
            X[j] = v < a.V ? X[j] : NEG_BIG;
            tl += v == y ? X[j] * c_l2 : 0.f;
          }
        }
        // ---- row maximum of the block (the two lane halves of a row share it), in log2 units -------------------------------
        float m0 = fmaxf(fmaxf(X[0], X[1]), X[2]), m1 = fmaxf(fmaxf(X[3], X[4]), X[5]);
        m0 = fmaxf(fmaxf(m0, X[6]), X[7]); m1 = fmaxf(fmaxf(m1, X[8]), X[9]);
        m0 = fmaxf(fmaxf(m0, X[10]), X[11]); m1 = fmaxf(fmaxf(m1, X[12]), X[13]);
        const float mx = max_xor32(fmaxf(fmaxf(fmaxf(m0, X[14]), X[15]), m1)) * c_l2;
        // ---- rare, wave-uniform: a row's maximum outgrew the one in use by more than RESCALE_THR -> bring O, l and the
        // probabilities that are exponentiated but not yet multiplied in (`pending`: the sub-tile before) to a new one.  The
        // maximum in use is an INTEGER, so the factor is an exact power of two, also for the bf16 probabilities.  The
        // accumulators live in AGPRs; written as plain C++ the 256 multiplies pull them into VGPRs and the whole kernel spills,
        // so each one is read / scaled / written back in place.  MFMA results need up to 18 wait states before
        // v_accvgpr_read may see them; the assembler pads nothing inside asm.
        if (__builtin_expect(!__all(mx - m_used <= RESCALE_THR), 0)) {
          const float m_new = ceilf(fmaxf(m_used, mx));
          const float alpha = __builtin_amdgcn_exp2f(m_used - m_new);
          if (!nothing_yet) {                                  // (the segment's first block: O, l and `pending` are zero)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                float tmp;
                asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 1\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1"
                             : "+a"(O[c][j]), "=&v"(tmp) : "v"(alpha));
              }
            asm volatile("s_nop 7" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int e = 0; e < 8; ++e) pending[s2][e] = (__bf16)((float)pending[s2][e] * alpha);
          }
          l_sum *= alpha;
          m_used = m_new;
        }
      };

      // ---- steps 0..31: first product of A ------------------------------------------------------------------------------------
      static_for<0, 32>([&](auto I) { mfma_step(I); PV_SCHED_FENCE(); });
      PV_STAMP(2)
      // ---- steps 32..63: second product of the sub-tile before.  Beside it: from its third step on (the first product's last
      // MFMA has left the pipe by then) the decision about A's maximum, then A's transform; B's bias into its accumulator near
      // the end.  The decision used to stand BETWEEN the two products, a stretch with no MFMA in flight; a rescale it asks for
      // (rare) waits for the second product's MFMAs in flight and scales O between two of its steps -- the products already
      // added are scaled with O, the ones to come use the scaled pending probabilities: the same sum.
      static_for<32, 64>([&](auto I) {
        constexpr int i = decltype(I)::value;
        mfma_step(I);
        if constexpr (i == DECIDE_AT) decide(XA, v0, pfB, k == 0);
        if constexpr (i > DECIDE_AT && (i & 1) == 1) exp_elem(XA, pfA, std::integral_constant<int, (i - DECIDE_AT - 1) / 2>{});
        if constexpr (i == 62) exp_elem(XA, pfA, std::integral_constant<int, 15>{});
        if constexpr (i == 63) init_x(XB, 1);
        PV_SCHED_FENCE();
      });
      PV_STAMP(4)
      // ---- steps 64..95: first product of B ------------------------------------------------------------------------------------
      static_for<64, 96>([&](auto I) { mfma_step(I); PV_SCHED_FENCE(); });
      PV_STAMP(2)
      // ---- steps 96..127: second product of A | decision about and transform of B -----------------------------------------------
      static_for<96, 128>([&](auto I) {
        constexpr int i = decltype(I)::value;
        mfma_step(I);
        if constexpr (i == 64 + DECIDE_AT) decide(XB, v0 + 32, pfA, false);
        if constexpr (i > 64 + DECIDE_AT && (i & 1) == 1) exp_elem(XB, pfB, std::integral_constant<int, (i - 64 - DECIDE_AT - 1) / 2>{});
        if constexpr (i == 126) exp_elem(XB, pfB, std::integral_constant<int, 15>{});
        PV_SCHED_FENCE();
      });
      PV_STAMP(4)
    }
    // ---- the segment's last sub-tile (B of the last item's buffer) has not been multiplied in yet ---------------------------
    {
      const int lpar = par ^ 1;
      const FragBase fbl{fb0.row + lpar * BUF_BYTES, fb0.col0 + lpar * BUF_BYTES, fb0.col1 + lpar * BUF_BYTES};
      FragAddr fa;
      bf16x8 f[8];
      static_for<0, DEPTH>([&](auto I) { f[decltype(I)::value & 7] = tail_frag<decltype(I)::value>(smem, fbl, fa); });
      static_for<0, 32>([&](auto I) {
        constexpr int n = decltype(I)::value;
        if constexpr (n + DEPTH < 32) f[(n + DEPTH) & 7] = tail_frag<n + DEPTH>(smem, fbl, fa);
        mfma_a(O[4 * (n & 3) + ((n & 15) >> 2)], f[n & 7], pfB[n >> 4]);
        PV_SCHED_FENCE();
      });
    }
    PV_STAMP(5)
    // ---- partial result of this segment: O^T as [hc][t] (128-byte runs), (m, l, target logit) per row -----------------
    mfma_a_done();
    const long slab = (long)rid * a.max_seg + seg;
    float* o = a.o_slab + slab * HD * UNIT + 32 * w + r32;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) o[(long)(32 * c + (j & 3) + 8 * (j >> 2) + 4 * hh) * UNIT] = O[c][j];
    l_sum += xor32(l_sum);
    tl += xor32(tl);
    if (hh == 0) a.stats[slab * UNIT + 32 * w + r32] = make_float4(m_used * LN2, l_sum, tl * LN2, 0.f);
    PV_STAMP(6)
  }
  PV_STAMP_FLUSH(0)
}

// fold a row group's segments: lse, nll, U[t][:] = sum_seg O_seg e^(m_seg - M) / L - W[y_t]
struct FwdCombineArgs {
  const float* o_slab;
  const float4* stats;
  const bf16_t* W;
  const int* target;
  const int* live;
  int T, V, ldw, max_seg, G;
  long ignore_index;
  float* lse;           // [T]
  float* nll;           // [T]
  float* U;             // [T][HD]
};

__global__ __launch_bounds__(256) void vocab_fwd_combine_kernel(FwdCombineArgs a) {
  __shared__ float tile[32][UNIT + 1];
  __shared__ float rowM[UNIT];
  __shared__ float invL[UNIT];
  const int rg = blockIdx.x, hc0 = blockIdx.y * 32, tid = threadIdx.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NVI = (a.V + ITEM - 1) / ITEM, NRG = (nlive + UNIT - 1) / UNIT;
  if (rg >= NRG) {                             // rows nobody computed (ignored tokens behind the live ones): zeros
    for (int i = tid; i < UNIT * 32; i += 256) {
      const int t = rg * UNIT + (i >> 5);
      if (t < a.T) a.U[(long)t * HD + hc0 + (i & 31)] = 0.f;
    }
    if (blockIdx.y == 0 && tid < UNIT && rg * UNIT + tid < a.T) {
      a.lse[rg * UNIT + tid] = 0.f;
      a.nll[rg * UNIT + tid] = 0.f;
    }
    return;
  }
  const long N = (long)NRG * NVI, x0 = (long)rg * NVI, x1 = x0 + NVI;
  // the workgroups whose item range meets [x0, x1) are consecutive: w0 .. w1 - 1 (some of them may be empty when there
  // are fewer items than workgroups); workgroup w keeps this unit in slab w * max_seg + (rg - first unit of w)
  int w0 = (int)(x0 * a.G / N);
  while (w0 > 0 && seg_beg(w0, N, a.G) > x0) --w0;
  while (seg_beg(w0 + 1, N, a.G) <= x0) ++w0;
  int w1 = w0 + 1;
  while (w1 < a.G && seg_beg(w1, N, a.G) < x1) ++w1;
  auto slab_of = [&](int w) -> long {
    const long b = seg_beg(w, N, a.G), e = seg_beg(w + 1, N, a.G);
    return (e <= x0 || e == b) ? -1 : (long)w * a.max_seg + (rg - (int)(b / NVI));
  };
  if (tid < UNIT) {
    float M = NEG_BIG;
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl >= 0) M = fmaxf(M, a.stats[sl * UNIT + tid].x);
    }
    float L = 0.f, tl = 0.f;
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl < 0) continue;
      const float4 st = a.stats[sl * UNIT + tid];
      L += st.y * __expf(st.x - M);
      tl += st.z;
    }
    rowM[tid] = M;
    invL[tid] = 1.f / L;
    const int t = rg * UNIT + tid;
    if (blockIdx.y == 0 && t < a.T) {
      const bool is_live = t < nlive && (long)a.target[t] != a.ignore_index;
      const float lse = M + logf(L);
      a.lse[t] = t < nlive ? lse : 0.f;
      a.nll[t] = is_live ? lse - tl : 0.f;
    }
  }
  __syncthreads();
  const int t = tid & (UNIT - 1), hsel = tid >> 7;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int w = w0; w < w1; ++w) {                  // segments in workgroup order: a fixed order of additions
    const long sl = slab_of(w);
    if (sl < 0) continue;
    const float sc = __expf(a.stats[sl * UNIT + t].x - rowM[t]);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += a.o_slab[(sl * HD + hc0 + hsel + 2 * i) * UNIT + t] * sc;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[hsel + 2 * i][t] = acc[i] * invL[t];
  __syncthreads();
  for (int i = tid; i < UNIT * 32; i += 256) {
    const int tl_ = i >> 5, c = i & 31, tg = rg * UNIT + tl_;
    if (tg >= a.T) continue;
    float u = 0.f;
    if (tg < nlive) {
      const int y = a.target[tg];
      const float wy = (y >= 0 && y < a.V && (long)y != a.ignore_index) ? bf2f(a.W[(long)y * a.ldw + hc0 + c]) : 0.f;
      u = ((long)y != a.ignore_index) ? tile[c][tl_] - wy : 0.f;
    }
    a.U[(long)tg * HD + hc0 + c] = u;
  }
}

// =====================================================================================================================
// dW / dbias:  per vocabulary column tile   Z[v][:] = sum_t P[t, v] h[t],  P = gs[t] (exp(s_tv - lse_t) - [v == y_t])
// =====================================================================================================================
struct DwArgs {
  const bf16_t* h;
  const bf16_t* W;
  const float* bias;
  const int* target;
  const float* lse;     // [T] compact
  const float* gs;      // [T] compact: dL/dnll, 0 for ignored rows (inv_temp is applied here)
  const int* live;
  int T, V, ldh, ldw, max_seg;
  float inv_temp;
  int v_base;           // vocabulary row of W[0] (a launch on a row range: the targets keep their global numbers)
  float* z_slab;        // [G * max_seg][UNIT][HD]
  float* b_slab;        // [G * max_seg][UNIT]
};

__global__ __launch_bounds__(256, 1) void vocab_dw_kernel(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NCT = (a.V + UNIT - 1) / UNIT, NRI = (nlive + ITEM - 1) / ITEM;
  const long N = (long)NCT * NRI;
  const int rid = __builtin_amdgcn_readfirstlane(phase_sorted_range(N, G, NRI, (int*)smem));
  const long beg = seg_beg(rid, N, G), end = seg_beg(rid + 1, N, G);
  const FragBase fb0 = make_frag_base(lane);
  const int ldb = a.ldh * 2;
  const rsrc_t hsrc = make_rsrc(a.h, (long)a.T * ldb);
  // side data of a tile: lse[64] | gs[64] | target[64], one 4-byte piece each (waves 0, 1, 2); rows >= T arrive as zeros
  const rsrc_t asrc = make_rsrc(w == 0 ? (const void*)a.lse : w == 1 ? (const void*)a.gs : (const void*)a.target, (long)a.T * 4);
  const StageLane sl = make_stage_lane(lane, w, ldb);
  const float c_l2 = a.inv_temp * LOG2E;
  int seg = 0, par = 0;
  PV_STAMP_DECL
  if (beg < end) {
    char* buf = smem;
    const int t0 = (int)(beg % NRI) * ITEM;
    static_for<0, 8>([&](auto I) { stage_piece<decltype(I)::value>(hsrc, sl, ldb, t0 * ldb, buf, w); });
    if (w < 3) lds_dma<4>(asrc, (t0 + lane) * 4, buf + TILE_BYTES + 256 * w);
    static_for<8, 16>([&](auto I) { stage_piece<decltype(I)::value>(hsrc, sl, ldb, t0 * ldb, buf, w); });
    // the first second product of a segment multiplies the OTHER buffer's sub-tile B by zeros: it must hold finite numbers
    for (int i = tid; i < SUB_BYTES / 16; i += 256) *(uint4*)(smem + BUF_BYTES + SUB_BYTES + 16 * i) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  for (long at = beg; at < end; ++seg) {
    const int ct = (int)(at / NRI), ri0 = (int)(at % NRI);
    const int nri = (int)min((long)NRI - ri0, end - at);
    at += nri;
    // ---- this wave's 32 rows of W as 32 B-operand fragments ------------------------------------------------------------
    const int vbase = ct * UNIT + 32 * w, v_mine = vbase + r32, v_tgt = v_mine + a.v_base;
    const bf16_t* wrow = a.W + (long)min(v_mine, a.V - 1) * a.ldw + 8 * hh;
    bf16x8 wf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(wrow + 16 * ks));
    float cb = a.bias[min(v_mine, a.V - 1)] * c_l2;
    // (waited for here, not at their first uses inside the item loop: see the forward kernel)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { u32x4 t_ = __builtin_bit_cast(u32x4, wf[ks]); asm volatile("" : "+v"(t_)); wf[ks] = __builtin_bit_cast(bf16x8, t_); }
    asm volatile("" : "+v"(cb));
    f32x16 Z[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) Z[c][j] = 0.f;
    float bsum = 0.f;
    bf16x8 pfA[2], pfB[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { pfB[0][e] = (__bf16)0.f; pfB[1][e] = (__bf16)0.f; }
    PV_STAMP(0)
    for (int k = 0; k < nri; ++k, par ^= 1) {
      half_tile_landed();
      PV_STAMP(1)
      const char* tile = smem + par * BUF_BYTES;
      char* nbuf = smem + (par ^ 1) * BUF_BYTES;
      float* lse_t = (float*)(tile + TILE_BYTES);
      float* gs_t = lse_t + 64;
      const int* tgt_t = (const int*)(tile + TILE_BYTES + 512);
      const int t0 = (ri0 + k) * ITEM;
      const int nt0 = k + 1 < nri ? t0 + ITEM : (at < end ? 0 : NRI * ITEM);
      const int noff = nt0 * ldb;
      const int dprev = par ? -BUF_BYTES : BUF_BYTES;
      // rows past the live count (the last item of a unit): their gs is zeroed in the tile's side data, so P = 0 exactly
      // (every wave stores the same zeros ahead of its own reads; 2^x is clamped to 1, so nothing is infinite)
      if (__builtin_expect(t0 + ITEM > nlive, 0)) {
        if (t0 + lane >= nlive) gs_t[lane] = 0.f;
      }
      bf16x8 f[8];
      f32x16 X;
      const FragBase fb{fb0.row + par * BUF_BYTES, fb0.col0 + par * BUF_BYTES, fb0.col1 + par * BUF_BYTES};
      FragAddr fa;
      static_for<0, DEPTH>([&](auto I) { f[decltype(I)::value & 7] = item_frag<decltype(I)::value>(smem, fb, dprev, fa); });
      PV_SCHED_FENCE();

      auto mfma_step = [&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        using M = StepMap<i>;
        if constexpr (i + DEPTH == 64) half_tile_landed();
        if constexpr (i + DEPTH < 128) f[(i + DEPTH) & 7] = item_frag<i + DEPTH>(smem, fb, dprev, fa);
        if constexpr (M::first) mfma_v<(M::n == 0 ? 2 : 0)>(X, f[i & 7], wf[M::ks]);
        else if constexpr (M::phase == 1) mfma_a(Z[M::c], pfB[M::s], f[i & 7]);
        else mfma_a(Z[M::c], pfA[M::s], f[i & 7]);
#ifndef PV_ABL_NOSTAGE
        if constexpr (PV_STAGE_AT(i & 63)) stage_piece<8 * (i >> 6) + (i & 63) / 7>(hsrc, sl, ldb, noff, nbuf, w);
#endif
        if constexpr (i == 54) {
          if (w < 3) lds_dma<4>(asrc, (nt0 + lane) * 4, nbuf + TILE_BYTES + 256 * w);
        }
      };
      // P = gs (softmax - onehot) for element J of a sub-tile's accumulator, in two halves (one per MFMA step): 2^z,
      // z = X c + cb - lse log2e, with the output clamped to [0, 1] by the instruction's own modifier (a dead row's stale lse
      // must not give infinity: its gs is 0); then q = p g - [y == v] g as one compare, one select and one fma.  Column sums of P
      // are the bias gradient.  inv_temp is applied by the reduce kernel.  Columns past V need no mask: nothing reads their Z rows.
      // The onehot is branch-free: with word frequencies as skewed as a vocabulary's, the waves that own the frequent tokens'
      // columns meet a target in every tile, and a slower variant for them would be the kernel's tail (stream-K gives every
      // workgroup the same number of items); two variants of the stream in one kernel also made hipcc spill 360 registers.
      f32x4 l4, g4;
      int4 y4;
      float p_half;
      auto elem_a = [&](int sub, auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
#ifndef PV_ABL_NOEXP
        if constexpr ((j & 3) == 0) {
          l4 = *(const f32x4*)(lse_t + 32 * sub + 2 * j + 4 * hh);
          g4 = *(const f32x4*)(gs_t + 32 * sub + 2 * j + 4 * hh);
          y4 = *(const int4*)(tgt_t + 32 * sub + 2 * j + 4 * hh);
        }
        const float z = fmaf(X[j], c_l2, fmaf(l4[j & 3], -LOG2E, cb));
        float pe;                                           // (asm operands cannot name a captured variable)
        asm("v_exp_f32_e64 %0, %1 clamp" : "=v"(pe) : "v"(z));
        p_half = pe;
#else
        p_half = X[j];
#endif
      };
      auto elem_b = [&](bf16x8 (&pf)[2], auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
#ifndef PV_ABL_NOEXP
        const int yy = (j & 3) == 0 ? y4.x : (j & 3) == 1 ? y4.y : (j & 3) == 2 ? y4.z : y4.w;
        const float g = g4[j & 3];
        const float q = fmaf(p_half, g, yy == v_tgt ? -g : 0.f);
#else
        const float q = p_half;
#endif
        bsum += q;
        pf[j >> 3][j & 7] = (__bf16)q;
      };
      // ---- steps 0..31: first product of A; 32..63: second product of the sub-tile before | transform of A --------------------
      // (the transform starts at the second product's third step: the first product's last MFMA has left the pipe by then, so
      // no wait with an idle matrix pipe stands between the two products; the two halves of an element alternate, the last two
      // steps take two halves each)
      static_for<0, 32>([&](auto I) { mfma_step(I); PV_SCHED_FENCE(); });
      PV_STAMP(2)
      static_for<32, 64>([&](auto I) {
        constexpr int i = decltype(I)::value, sl = i - 34;
        mfma_step(I);
        if constexpr (sl >= 0 && sl < 28) {
          if constexpr ((sl & 1) == 0) elem_a(0, std::integral_constant<int, sl / 2>{});
          else elem_b(pfA, std::integral_constant<int, sl / 2>{});
        } else if constexpr (sl >= 28) {
          elem_a(0, std::integral_constant<int, sl - 14>{});
          elem_b(pfA, std::integral_constant<int, sl - 14>{});
        }
        PV_SCHED_FENCE();
      });
      PV_STAMP(4)
      // ---- steps 64..95: first product of B; 96..127: second product of A | transform of B ------------------------------------
      static_for<64, 96>([&](auto I) { mfma_step(I); PV_SCHED_FENCE(); });
      PV_STAMP(2)
      static_for<96, 128>([&](auto I) {
        constexpr int i = decltype(I)::value, sl = i - 98;
        mfma_step(I);
        if constexpr (sl >= 0 && sl < 28) {
          if constexpr ((sl & 1) == 0) elem_a(1, std::integral_constant<int, sl / 2>{});
          else elem_b(pfB, std::integral_constant<int, sl / 2>{});
        } else if constexpr (sl >= 28) {
          elem_a(1, std::integral_constant<int, sl - 14>{});
          elem_b(pfB, std::integral_constant<int, sl - 14>{});
        }
        PV_SCHED_FENCE();
      });
      PV_STAMP(4)
    }
    // ---- the segment's last sub-tile (B of the last item's buffer) has not been multiplied in yet ---------------------------
    {
      const int lpar = par ^ 1;
      const FragBase fbl{fb0.row + lpar * BUF_BYTES, fb0.col0 + lpar * BUF_BYTES, fb0.col1 + lpar * BUF_BYTES};
      FragAddr fa;
      bf16x8 f[8];
      static_for<0, DEPTH>([&](auto I) { f[decltype(I)::value & 7] = tail_frag<decltype(I)::value>(smem, fbl, fa); });
      static_for<0, 32>([&](auto I) {
        constexpr int n = decltype(I)::value;
        if constexpr (n + DEPTH < 32) f[(n + DEPTH) & 7] = tail_frag<n + DEPTH>(smem, fbl, fa);
        mfma_a(Z[4 * (n & 3) + ((n & 15) >> 2)], pfB[n >> 4], f[n & 7]);
        PV_SCHED_FENCE();
      });
    }
    PV_STAMP(5)
    mfma_a_done();
    const long slab = (long)rid * a.max_seg + seg;
    float* z = a.z_slab + slab * UNIT * HD + r32;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) z[(long)(32 * w + (j & 3) + 8 * (j >> 2) + 4 * hh) * HD + 32 * c] = Z[c][j];
    bsum += xor32(bsum);
    if (hh == 0) a.b_slab[slab * UNIT + 32 * w + r32] = bsum;
    PV_STAMP(6)
  }
  PV_STAMP_FLUSH(1)
}

struct DwReduceArgs {
  const float* z_slab;
  const float* b_slab;
  const int* live;
  int T, V, max_seg, G, lddw, accumulate, accumulate_bias;
  float scale;          // inv_temp: d(logit)/d(W h + b)
  float* dW;            // [V][lddw]
  float* dbias;         // [V]
};

__global__ __launch_bounds__(256) void vocab_dw_reduce_kernel(DwReduceArgs a) {
  const int ct = blockIdx.x, tid = threadIdx.x;
  const int nlive = a.live ? min(a.T, *a.live) : a.T;
  const int NCT = (a.V + UNIT - 1) / UNIT, NRI = (nlive + ITEM - 1) / ITEM;
  const long N = (long)NCT * NRI, x0 = (long)ct * NRI, x1 = x0 + NRI;
  int w0 = 0, w1 = 0;
  if (N > 0) {
    w0 = (int)(x0 * a.G / N);
    while (w0 > 0 && seg_beg(w0, N, a.G) > x0) --w0;
    while (seg_beg(w0 + 1, N, a.G) <= x0) ++w0;
    w1 = w0 + 1;
    while (w1 < a.G && seg_beg(w1, N, a.G) < x1) ++w1;
  }
  auto slab_of = [&](int w) -> long {
    const long b = seg_beg(w, N, a.G), e = seg_beg(w + 1, N, a.G);
    return (e <= x0 || e == b) ? -1 : (long)w * a.max_seg + (ct - (int)(b / NRI));
  };
  // rows blockIdx.y * 16 .. + 16 of the tile, 4 columns per thread and step; segments in workgroup order
  for (int i = tid; i < 16 * (HD / 4); i += 256) {
    const int vl = blockIdx.y * 16 + i / (HD / 4), c4 = (i % (HD / 4)) * 4, v = ct * UNIT + vl;
    if (v >= a.V) continue;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w = w0; w < w1; ++w) {
      const long sl = slab_of(w);
      if (sl >= 0) acc += *(const f32x4*)(a.z_slab + (sl * UNIT + vl) * HD + c4);
    }
    f32x4* dst = (f32x4*)(a.dW + (long)v * a.lddw + c4);
    acc *= a.scale;
    *dst = a.accumulate ? *dst + acc : acc;
  }
  if (tid < 16) {
    const int vl = blockIdx.y * 16 + tid, v = ct * UNIT + vl;
    if (v < a.V) {
      float acc = 0.f;
      for (int w = w0; w < w1; ++w) {
        const long sl = slab_of(w);
        if (sl >= 0) acc += a.b_slab[sl * UNIT + vl];
      }
      acc *= a.scale;
      a.dbias[v] = a.accumulate_bias ? a.dbias[v] + acc : acc;
    }
  }
}

// dst[perm[new]][:] = new < count ? src[new][:] * scale[new] : 0
__global__ void scatter_rows_scaled_kernel(const float* __restrict__ src, const float* __restrict__ scale, float factor,
                                           const int64_t* __restrict__ perm, const int* __restrict__ count, int T, int cols4,
                                           float* __restrict__ dst, long ldd) {
  const long n = (long)T * cols4;
  const int live = count ? *count : T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols4), c = (int)(i % cols4) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < live) v = *(const f32x4*)(src + (long)row * cols4 * 4 + c) * (scale[row] * factor);
    *(f32x4*)(dst + (perm ? perm[row] : row) * ldd + c) = v;
  }
}

// the same with the backward of a dropout on the destination folded in: destination row r = i0 * n1p + i1 holds the
// logical elements i0 * m0 + i1 * m1 + c (pv_dropout_f32's index convention); rows with i1 >= n1 are padding
struct RowDrop {
  const uint8_t* mask;
  const RngState* rng;
  uint32_t site, thresh;
  float scale;
  int n1, n1p;
  long m0, m1;
};

__global__ void scatter_rows_scaled_drop_kernel(const float* __restrict__ src, const float* __restrict__ scale, float factor,
                                                const int64_t* __restrict__ perm, const int* __restrict__ count, int T, int cols4,
                                                float* __restrict__ dst, long ldd, RowDrop d) {
  const long n = (long)T * cols4;
  const int live = count ? *count : T;
  RngState st{0, 0};
  if (!d.mask) st = *d.rng;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols4), c = (int)(i % cols4) * 4;
    const long drow = perm ? perm[row] : row;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int i0 = (int)(drow / d.n1p), i1 = (int)(drow % d.n1p);
    if (row < live && i1 < d.n1) {
      const f32x4 x = *(const f32x4*)(src + (long)row * cols4 * 4 + c) * (scale[row] * factor);
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      drop4_at(d.mask, st, d.site, d.thresh, d.scale, i0 * d.m0 + i1 * d.m1 + c, v);
    }
    *(f32x4*)(dst + drow * ldd + c) = f32x4{v[0], v[1], v[2], v[3]};
  }
}

int grid_wgs() {
  static int g = 0;
  if (!g) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g = cus;
  }
  return g;
}

int max_seg_fwd(int T, int V, int G) {
  const long NVI = (V + ITEM - 1) / ITEM, NRG = (T + UNIT - 1) / UNIT;
  const long per = (NRG * NVI + G - 1) / G;
  return (int)(per / NVI) + 2;
}
int max_seg_dw(int V, int G) { return ((V + UNIT - 1) / UNIT + G - 1) / G + 1; }

int check_common(const void* h, int ldh, const void* W, int ldw, int T, int V, int H, const char* who) {
  PV_REQUIRE(h && W && T > 0 && V > 0, "%s: null / empty operand", who);
  PV_REQUIRE(H == HD, "%s: built for H = %d (got %d); use the tile-GEMM path", who, HD, H);
  PV_REQUIRE(ldh >= HD && ldw >= HD && ldh % 8 == 0 && ldw % 8 == 0, "%s: leading dimensions must be multiples of 8 and >= %d", who, HD);
  PV_REQUIRE(((uintptr_t)h & 15) == 0 && ((uintptr_t)W & 15) == 0, "%s: operands must be 16-byte aligned", who);
  PV_REQUIRE(((long)T + 2 * ITEM) * ldh * 2 < 0x7fffffffL && ((long)V + 2 * ITEM) * ldw * 2 < 0x7fffffffL,
             "%s: an operand of 2 GiB or more (buffer offsets are 32-bit)", who);
  return PV_OK;
}

template <class K>
int set_smem(K kernel) {
  PV_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  return PV_OK;
}

}  // namespace

extern "C" {

#ifdef PV_VOCAB_STAMPS
int pv_vocab_stamps_copy(void* host_dst, size_t bytes) {
  PV_HIP(hipDeviceSynchronize());
  PV_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_vstamps), bytes < sizeof(g_vstamps) ? bytes : sizeof(g_vstamps)));
  return PV_OK;
}
#endif

int pv_vocab_fused_supported(int H) { return H == HD ? 1 : 0; }

size_t pv_vocab_fused_fwd_ws_bytes(int T, int V) {
  const int G = grid_wgs();
  const size_t slabs = (size_t)G * max_seg_fwd(T, V, G);
  return slabs * ((size_t)HD * UNIT * sizeof(float) + UNIT * sizeof(float4));
}

int pv_vocab_fused_fwd(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, int T, int V,
                       int H, float inv_temp, int64_t ignore_index, void* workspace, float* lse, float* nll, float* U,
                       const int* live_rows, void* stream) {
  int rc = check_common(h, ldh, W, ldw, T, V, H, "pv_vocab_fused_fwd");
  if (rc) return rc;
  PV_REQUIRE(bias && target32 && workspace && lse && nll && U, "pv_vocab_fused_fwd: null argument");
  PV_REQUIRE(((uintptr_t)U & 15) == 0 && ((uintptr_t)workspace & 15) == 0, "pv_vocab_fused_fwd: U / workspace must be 16-byte aligned");
  static bool attr = false;
  if (!attr) {
    rc = set_smem(vocab_fwd_dh_kernel);
    if (rc) return rc;
    attr = true;
  }
  const int G = grid_wgs(), ms = max_seg_fwd(T, V, G);
  float* o_slab = (float*)workspace;
  float4* stats = (float4*)(o_slab + (size_t)G * ms * HD * UNIT);
  hipStream_t st = (hipStream_t)stream;
  FwdArgs a{(const bf16_t*)h, (const bf16_t*)W, bias, target32, live_rows, T, V, ldh, ldw, ms, inv_temp, o_slab, stats};
  hipLaunchKernelGGL(vocab_fwd_dh_kernel, dim3(G), dim3(256), SMEM_BYTES, st, a);
  PV_LAUNCH_CHECK("vocab_fwd_dh_kernel");
  FwdCombineArgs c{o_slab, stats, (const bf16_t*)W, target32, live_rows, T, V, ldw, ms, G, (long)ignore_index, lse, nll, U};
  hipLaunchKernelGGL(vocab_fwd_combine_kernel, dim3(cdiv(T, UNIT), HD / 32), dim3(256), 0, st, c);
  PV_LAUNCH_CHECK("vocab_fwd_combine_kernel");
  return PV_OK;
}

size_t pv_vocab_fused_dw_ws_bytes(int V) {
  const int G = grid_wgs();
  return (size_t)G * max_seg_dw(V, G) * ((size_t)UNIT * HD + UNIT) * sizeof(float);
}

int pv_vocab_fused_dw(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                      const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                      int accumulate, float* dbias, const int* live_rows, void* stream) {
  return pv_vocab_fused_dw_wgs(h, ldh, W, ldw, bias, target32, lse, gscale, T, V, H, inv_temp, workspace, dW, lddw, accumulate,
                               dbias, 0, live_rows, 0, stream);
}

int pv_vocab_fused_dw_wgs(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                          const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                          int accumulate, float* dbias, int accumulate_bias, const int* live_rows, int wgs, void* stream) {
  return pv_vocab_fused_dw_range(h, ldh, W, ldw, bias, target32, lse, gscale, T, V, H, inv_temp, 0, V, workspace, dW, lddw,
                                 accumulate, dbias, accumulate_bias, live_rows, wgs, stream);
}

int pv_vocab_fused_dw_range(const void* h, int ldh, const void* W_all, int ldw, const float* bias_all, const int* target32,
                            const float* lse, const float* gscale, int T, int V_all, int H, float inv_temp, int v0, int v1,
                            void* workspace, float* dW_all, int lddw, int accumulate, float* dbias_all, int accumulate_bias,
                            const int* live_rows, int wgs, void* stream) {
  PV_REQUIRE(W_all && bias_all && dW_all && dbias_all, "pv_vocab_fused_dw: null argument");
  PV_REQUIRE(0 <= v0 && v0 < v1 && v1 <= V_all && v0 % UNIT == 0, "pv_vocab_fused_dw_range: rows [%d, %d) of %d (v0 must be a multiple of %d)",
             v0, v1, V_all, UNIT);
  const int V = v1 - v0;
  const void* W = (const char*)W_all + (size_t)v0 * ldw * sizeof(bf16_t);
  const float* bias = bias_all + v0;
  float* dW = dW_all + (size_t)v0 * lddw;
  float* dbias = dbias_all + v0;
  int rc = check_common(h, ldh, W, ldw, T, V, H, "pv_vocab_fused_dw");
  if (rc) return rc;
  PV_REQUIRE(bias && target32 && lse && gscale && workspace && dW && dbias, "pv_vocab_fused_dw: null argument");
  PV_REQUIRE(lddw >= HD && lddw % 4 == 0 && ((uintptr_t)dW & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
             "pv_vocab_fused_dw: dW must be 16-byte aligned with lddw a multiple of 4 and >= %d", HD);
  static bool attr = false;
  if (!attr) {
    rc = set_smem(vocab_dw_kernel);
    if (rc) return rc;
    attr = true;
  }
  PV_REQUIRE(wgs >= 0, "pv_vocab_fused_dw_wgs: wgs = %d", wgs);
  const int G = wgs > 0 && wgs < grid_wgs() ? wgs : grid_wgs(), ms = max_seg_dw(V, G);
  PV_REQUIRE((size_t)G * ms * ((size_t)UNIT * HD + UNIT) * sizeof(float) <= pv_vocab_fused_dw_ws_bytes(V),
             "pv_vocab_fused_dw_wgs: %d workgroups need more than pv_vocab_fused_dw_ws_bytes", G);
  float* z_slab = (float*)workspace;
  float* b_slab = z_slab + (size_t)G * ms * UNIT * HD;
  hipStream_t st = (hipStream_t)stream;
  DwArgs a{(const bf16_t*)h, (const bf16_t*)W, bias, target32, lse, gscale, live_rows, T, V, ldh, ldw, ms, inv_temp, v0, z_slab, b_slab};
  hipLaunchKernelGGL(vocab_dw_kernel, dim3(G), dim3(256), SMEM_BYTES, st, a);
  PV_LAUNCH_CHECK("vocab_dw_kernel");
  DwReduceArgs r{z_slab, b_slab, live_rows, T, V, ms, G, lddw, accumulate, accumulate_bias, inv_temp, dW, dbias};
  hipLaunchKernelGGL(vocab_dw_reduce_kernel, dim3(cdiv(V, UNIT), UNIT / 16), dim3(256), 0, st, r);
  PV_LAUNCH_CHECK("vocab_dw_reduce_kernel");
  return PV_OK;
}

int pv_scatter_rows_scaled_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count, int T,
                               int cols, float* dst, long ldd, void* stream) {
  PV_REQUIRE(src && scale && dst && T > 0 && cols > 0 && cols % 4 == 0 && ldd >= cols && ldd % 4 == 0,
             "pv_scatter_rows_scaled_f32: bad argument (cols and ldd must be multiples of 4)");
  PV_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pv_scatter_rows_scaled_f32: buffers must be 16-byte aligned");
  const long n = (long)T * (cols / 4);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(scatter_rows_scaled_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream, src,
                     scale, factor, perm, count, T, cols / 4, dst, ldd);
  PV_LAUNCH_CHECK("scatter_rows_scaled_kernel");
  return PV_OK;
}

int pv_scatter_rows_scaled_drop_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count,
                                    int T, int cols, float* dst, long ldd, int n1, int n1p, long m0, long m1, float p,
                                    const void* mask, const void* rng_state, int site, void* stream) {
  PV_REQUIRE(src && scale && dst && T > 0 && cols > 0 && cols % 4 == 0 && ldd >= cols && ldd % 4 == 0,
             "pv_scatter_rows_scaled_drop_f32: bad argument (cols and ldd must be multiples of 4)");
  PV_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pv_scatter_rows_scaled_drop_f32: buffers must be 16-byte aligned");
  PV_REQUIRE(n1 > 0 && n1p >= n1 && T % n1p == 0, "pv_scatter_rows_scaled_drop_f32: T=%d rows are not whole groups of n1p=%d", T, n1p);
  PV_REQUIRE(p >= 0.f && p < 1.f && (mask || rng_state), "pv_scatter_rows_scaled_drop_f32: p in [0, 1) and a keep mask or a generator state");
  RowDrop d{(const uint8_t*)mask, (const RngState*)rng_state, (uint32_t)site, drop_threshold(p), 1.f / (1.f - p), n1, n1p, m0, m1};
  const long n = (long)T * (cols / 4);
  const long gx = (n + 255) / 256;
  hipLaunchKernelGGL(scatter_rows_scaled_drop_kernel, dim3((unsigned)(gx > 4096 ? 4096 : gx)), dim3(256), 0, (hipStream_t)stream,
                     src, scale, factor, perm, count, T, cols / 4, dst, ldd, d);
  PV_LAUNCH_CHECK("scatter_rows_scaled_drop_kernel");
  return PV_OK;
}

}  // extern "C"
