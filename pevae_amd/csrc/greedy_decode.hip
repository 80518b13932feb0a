// Persistent free-running greedy decode for gfx950: ONE launch walks all L tokens of a batch.
//
// Reference semantics: the GEN branch of DecoderRNN.forward with gen_type 'greedy' (models/enc2dec/decoders.py:
// 236-262): per token  embedding(y_{t-1}) -> [cat latent] -> one GRU step (base_modules.py:42, gate order r,z,n) ->
// project -> log_softmax (decoders.py:191-193) -> [mask_unk, :247-248] -> topk(1) (:252-254) -> next input.  No early
// exit at EOS (the reference decodes max_length tokens for every row and derives the lengths afterwards).
//
// Why one kernel: the token loop is a dependent chain (symbol -> input -> state -> logits -> symbol); as separate
// launches it costs ~25 launches and 70 us per token under graph replay.  Here nothing leaves the chip between tokens:
//
//   * the input projection of EVERY vocabulary entry is formed once per decode by one MFMA GEMM (the caller's
//     gx_table[v] = W_ih[:, :E] emb[v] + b_ih; with a latent input, gx_const[b] = W_ih[:, E:] latent[b]), so a token's
//     input projection is a row gather;
//   * workgroup r has up to two roles.  GRU role (g, s): batch rows [32g, 32g+32) x hidden units [16s, 16s+16), its
//     48 rows of W_hh resident in LDS, the fp32 state of its (row, unit) pairs in registers (one per thread); two waves
//     multiply (16 rows each, all three gates), all threads do the gate math.
//     Vocabulary role: columns [r NC, r NC + NC) of the projection, its NC rows of W_proj resident in LDS for the
//     whole decode (80 rows at V = 20 004 on 256 CUs): W_proj is read from HBM once per decode, not once per token;
//   * per token two chip-wide hand-offs: (A) every GRU workgroup has published its slice of h_t (bf16, 16-byte sc1
//     stores) -> vocabulary phase: each wave multiplies 32 batch rows by the resident columns (A fragments straight
//     from the published states, B fragments from LDS), reduces max / sum-exp / best allowed column over its columns
//     and publishes one 8-byte (best value, column) record per (row, workgroup); (B) every vocabulary workgroup has
//     published its records -> each GRU workgroup folds the records of its 32 rows (lowest column wins ties, as a
//     left-to-right scan would), which gives the emitted symbol = the next input row.  The (max, sum-exp) record that
//     only the symbol's log-probability needs is computed AFTER the hand-off, from the accumulators still in
//     registers, and all log-probabilities are folded once, after the last token, by every workgroup in parallel.
//   * hand-off = per-workgroup flags, not counters: a workgroup stores the token number into its own word after its
//     payload stores have drained (vmcnt(0), barrier); a waiting wave loads all flags with one coalesced sc1 load per
//     64 workgroups.  No read-modify-write anywhere, nothing to reset between tokens.  Every wait is wall-clock
//     bounded and reports through the error word; after a failed wait a workgroup stops waiting, so the grid drains.
//   * the published states are stored in MFMA A-fragment order -- [16-row tile][k-step][q4][row in tile][8 bf16], the
//     1 KB a wave needs for one MFMA is one contiguous block -- because a row-major image makes every lane of a
//     fragment load touch its own cache line (64 tag look-ups per instruction: measured 10 us per token for the 256 KB
//     of a vocabulary phase, from any cache level);
//   * every published buffer has a slot per token (states [(L+1)][Bpad][Hk], records [L][NV][Bpad]): an address is
//     written once and only read after the hand-off that covers it.
//
// The grid must be co-resident (1 workgroup per CU; the entry point checks the CU count).  Placement is speed only:
// GRU role (g, s) sits on workgroup 8 s + g so that under round-robin dispatch a row group's slices share an XCD.
#include "common.h"
#include "../../include/pevae_hip.h"
#include <stdlib.h>

namespace {

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s of the 100 MHz wall clock
constexpr int NT = 512;          // 8 waves
constexpr int GROWS = 32;        // batch rows per GRU row group
constexpr int UNITS = 16;        // hidden units per GRU slice
constexpr int CT = 5;            // 16-column tiles per vocabulary chunk
constexpr u32 NEVER = 0x7fffffffu;
constexpr int MAX_WGS = 256;
constexpr size_t SYNC_BYTES = 4096;   // err word, flag_a[256], flag_b[256]

struct DecPart {
  float m, s, best;      // max over all columns, sum exp(x - m), best value over the allowed columns
  int idx;               // its column (lowest on ties), -1 if none
};

struct DecodeArgs {
  const float* gx_table;   // [V][3H]
  const float* gx_const;   // optional [B][3H]
  const bf16_t* whh;       // [3H][Hk] bf16, columns >= H zero
  const float* bhh;        // [3H]
  const bf16_t* wp;        // [V][ldwp] bf16, columns >= H zero
  const float* bp;         // [V]
  const float* h0;         // optional [B][H]
  bf16_t* hb;              // [(L+1)] slots of Bpad x Hk in MFMA-fragment order (see below); slot 0 pre-filled
  DecPart* part;           // two images of [L][NV][Bpad] x 8 bytes: (best, column) records, then (max, sum-exp) records
  u32* err;                // 1 word
  u32* flag_a;             // [nwg]: GRU workgroup r has published h of token flag_a[r] - 1
  u32* flag_b;             // [nwg]: vocabulary workgroup r has published the records of token flag_b[r] - 1
  int64_t* seq;            // [B][L]
  float* logp;             // [B][L]
  float* hn;               // [B][H]
  int L, B, H, Hk, Bpad, V, G, S, NV, NC, nwg, sos, mask_col, ldwp;
  float inv_temp;
  unsigned hb_bytes, part_bytes, part_b;   // part_b: byte offset of the (max, sum-exp) records inside `part`
  // `sample` decoding (decoders.py:255-256 -> base_modules.py:54-59): the record's value is the scaled logit MINUS log q,
  // q ~ Exp(1) from Philox keyed by (seed, step, site0 + token, row, column) -- the draws of pv_vocab_sample --, and the
  // unperturbed logit of each workgroup's candidate goes to `xsel` [L][NV][Bpad] for the emitted symbol's log-probability
  const RngState* rng;     // NULL: greedy
  float* xsel;
  unsigned site0;
  int dbg;                 // timing ablation bits (diagnostic build only, env PV_DECODE_DBG; results are wrong when set)
};

__device__ __forceinline__ u32 ld_relaxed(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_relaxed(u32* p, u32 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// v_rcp_f32 (1 ulp), not the correctly rounded division sequence __frcp_rn expands to (10 instructions per reciprocal:
// a third of the gate phase of a recurrence step)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// The whole wave waits until every flag is >= want.  false: timed out or another workgroup already reported.
__device__ __forceinline__ bool wait_all_ge(const u32* flags, int n, u32 want, u32* err, int lane) {
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    u32 v = NEVER;
    for (int i = lane; i < n; i += 64) v = min(v, ld_relaxed(flags + i));
    if (__all(v >= want)) return true;
    if ((it & 15) == 15) {
      if (ld_relaxed(err) != 0) return false;
      if (wall_clock64() - t0 > SPIN_LIMIT) {
        if (lane == 0) st_relaxed(err, 1u);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// (m, s) <- (m, s) (+) (pm, ps); (best, idx) <- the better of the two, lowest column on ties
__device__ __forceinline__ void fold(float& m, float& s, float& best, int& idx, float pm, float ps, float pb, int pi) {
  const float mn = fmaxf(m, pm);
  s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (pm == -INFINITY ? 0.f : ps * __expf(pm - mn));
  m = mn;
  if (pi >= 0 && (pb > best || (pb == best && (idx < 0 || pi < idx)))) {
    best = pb;
    idx = pi;
  }
}

// Diagnostic builds only.  -DPV_DECODE_ABLATE: the DBG bits switch parts of the kernel off (tools/decode_ablate.py; the
// branches around the loads perturb the timing of everything else, so never together with the stamps).
// -DPV_DECODE_STAMPS: wave 0 accumulates wall-clock (100 MHz) time per segment of a token into
// words 576.. of the sync area (tools/decode_stamps.py).  Never compiled into the shipped library.
#ifdef PV_DECODE_ABLATE
#define DBG a.dbg
#else
#define DBG 0
#endif
#ifdef PV_DECODE_STAMPS
#define STAMP(i)                                                    \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
    const unsigned long long now__ = wall_clock64();                \
    stamp_acc[i] += (unsigned)(now__ - stamp_t);                    \
    stamp_t = now__;                                                \
  } while (0)
#else
#define STAMP(i)
#endif

// all-reduce over the 16 lanes of a DPP row: xor 1, xor 2 (quad_perm), half-row mirror, row mirror
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  return fmaxf(v, dpp_f<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  return v + dpp_f<0x140>(v);
}
__device__ __forceinline__ int row16_min_int(int v) {
  v = min(v, dpp_i<0xB1>(v));
  v = min(v, dpp_i<0x4E>(v));
  v = min(v, dpp_i<0x141>(v));
  return min(v, dpp_i<0x140>(v));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// the (m, s) half of fold(): only the emitted symbol's log-probability needs it
__device__ __forceinline__ void fold_ms(float& m, float& s, float pm, float ps) {
  const float mn = fmaxf(m, pm);
  s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (pm == -INFINITY ? 0.f : ps * __expf(pm - mn));
  m = mn;
}

// the (best, idx) half of fold(): all a workgroup needs to know the next input symbol
__device__ __forceinline__ void fold_best(float& best, int& idx, float pb, int pi) {
  if (pi >= 0 && (pb > best || (pb == best && (idx < 0 || pi < idx)))) {
    best = pb;
    idx = pi;
  }
}

template <int AUX>
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}

// NKS_C: 32-deep k-steps (Hk / 32) when known at compile time (16 = the Yelp / BASELINE decoder), 0 = run-time loop.
// PLAIN: the published states / records are read with plain loads (each address is written once per launch and read
// only after its hand-off, so a line can be shared through the XCD's L2) instead of sc1 loads.
template <int NKS_C, bool PLAIN>
__global__ __launch_bounds__(NT, 1) void greedy_decode_kernel(DecodeArgs a) {
  constexpr int LDAUX = PLAIN ? 0 : AUX_SC1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int r = blockIdx.x, g = r & 7, s = r >> 3;
  const bool gru = g < a.G && s < a.S, voc = r < a.NV;
  const int Hk = NKS_C ? NKS_C * 32 : a.Hk, NKS = Hk / 32, nch = Hk / 8, ldw = Hk * 2, H = a.H;
  // LDS images of the weights: 16-byte chunk c of row n sits at chunk c ^ (n & mask) (conflict-free ds_read_b128 of the
  // MFMA B fragments; a padded row stride is 2-way on the instruction's 16-lane groups).  Lane (c16, q4) reads chunk
  // ks*4 + q4 of a row with (row & 15) == c16: byte offset (ks >> 2) * 256 + sw[ks & 3] within the row
  int mask = 16;
  while (nch % mask) mask >>= 1;
  mask -= 1;
  int sw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sw[j] = ((j * 4 + q4) ^ (c16 & mask)) * 16;
  char* whl = smem;                                   // [48][ldw]: W_hh slice (gate-major, 16 units each)
  char* wpl = whl + (size_t)(gru ? 48 * ldw : 0);     // [NC][ldw]: W_proj columns
  float* gh = (float*)(wpl + (size_t)(voc ? a.NC * ldw : 0));   // [3][32][16] fp32: W_hh h per gate
  bf16_t* tile = (bf16_t*)(gh + 3 * GROWS * UNITS);   // [32][16] bf16: h_t of this slice
  int* tok = (int*)(tile + GROWS * UNITS);            // [32]: the group's current input symbols

  if (gru)
    for (int i = tid; i < 48 * nch; i += NT) {
      const int n = i / nch, ch = i % nch, gate = n / UNITS, unit = s * UNITS + n % UNITS;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (unit < H) v = *(const uint4*)(a.whh + (size_t)(gate * H + unit) * Hk + ch * 8);
      *(uint4*)(whl + (size_t)n * ldw + ((ch ^ (n & mask)) * 16)) = v;
    }
  if (voc)
    for (int i = tid; i < a.NC * nch; i += NT) {
      const int n = i / nch, ch = i % nch, col = r * a.NC + n;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (col < a.V) v = *(const uint4*)(a.wp + (size_t)col * a.ldwp + ch * 8);
      *(uint4*)(wpl + (size_t)n * ldw + ((ch ^ (n & mask)) * 16)) = v;
    }
  if (tid == 0) {
    if (!gru) st_relaxed(a.flag_a + r, NEVER);
    if (!voc) st_relaxed(a.flag_b + r, NEVER);
  }
  // GRU role: thread (row, unit) owns one state element for the whole decode
  const int trow = tid >> 4, tun = tid & 15, brow = g * GROWS + trow, unit = s * UNITS + tun;
  const bool live = gru && brow < a.B && unit < H;
  float hreg = 0.f, bh[3] = {0.f, 0.f, 0.f}, gxc[3] = {0.f, 0.f, 0.f};
  if (live) {
    if (a.h0) hreg = a.h0[(size_t)brow * H + unit];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
      bh[gt] = a.bhh[gt * H + unit];
      if (a.gx_const) gxc[gt] = a.gx_const[(size_t)brow * 3 * H + gt * H + unit];
    }
  }
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.hb, 0, a.hb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)a.part, 0, a.part_bytes, 0x00020000);
  const int NCT = a.NC / 16;
  bool dead = false;                                  // wave 0 only: a wait has failed, stop waiting (the grid drains)
  __syncthreads();
#ifdef PV_DECODE_STAMPS
  unsigned stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_t = wall_clock64();
#endif

  for (int t = 0; t <= a.L; ++t) {
    if (gru) {
      // ---- (1) the symbols emitted for token t-1 = the inputs of step t ------------------------------------------
      if (t == 0) {
        if (tid < GROWS) tok[tid] = a.sos;
      } else {
        if (w == 0 && !dead && !(DBG & 64)) {
          dead = !wait_all_ge(a.flag_b, a.nwg, (u32)t, a.err, lane);
        }
        STAMP(0);
        __syncthreads();
        // (best, column) records of token t-1, [workgroup][row], 8 bytes each: lane (row j = lane & 31, half = lane >> 5)
        // of wave w folds workgroups 2 (w + 8 i) + half, all loads in flight at once; halves, then waves, through LDS
        {
          const int j = lane & 31, half = lane >> 5;
          float best = -INFINITY;
          int idx = -1;
          u32x2 v[MAX_WGS / 16];
          const unsigned base = (unsigned)((((size_t)(t - 1) * a.NV) * a.Bpad + g * GROWS + j) * 8);
          const unsigned kstep = (unsigned)(a.Bpad * 8);
#pragma unroll
          for (int i = 0; i < MAX_WGS / 16; ++i) {
            const int k = 2 * (w + 8 * i) + half;
            if (k < a.NV && !(DBG & 16)) v[i] = __builtin_amdgcn_raw_buffer_load_b64(prs, base + k * kstep, 0, LDAUX);
          }
#pragma unroll
          for (int i = 0; i < MAX_WGS / 16; ++i) {
            const int k = 2 * (w + 8 * i) + half;
            if (k < a.NV && !(DBG & 16)) {
              const f32x2 f = __builtin_bit_cast(f32x2, v[i]);   // (bit_cast of a single vector ELEMENT miscompiles)
              fold_best(best, idx, f[0], (int)v[i][1]);
            }
          }
          fold_best(best, idx, __shfl_xor(best, 32), __shfl_xor(idx, 32));
          if (lane < 32) {
            const u32x2 o = {__builtin_bit_cast(u32, best), (u32)idx};
            *(u32x2*)(gh + (w * GROWS + j) * 2) = o;           // gh is free until the GRU step
          }
        }
        __syncthreads();
        if (tid < GROWS) {
          float best = -INFINITY;
          int idx = -1;
#pragma unroll
          for (int ww = 0; ww < NT / 64; ++ww) {
            const u32x2 o = *(const u32x2*)(gh + (ww * GROWS + tid) * 2);
            const f32x2 f = __builtin_bit_cast(f32x2, o);
            fold_best(best, idx, f[0], (int)o[1]);
          }
          const int row = g * GROWS + tid;
          tok[tid] = min(max(idx, 0), a.V - 1);
          if (s == 0 && row < a.B) a.seq[(size_t)row * a.L + t - 1] = idx;
        }
      }
      __syncthreads();
      STAMP(1);
      if (t == a.L) break;
      // ---- (2) one GRU step of the slice ------------------------------------------------------------------------
      float gx[3] = {0.f, 0.f, 0.f};
      if (live) {
        const float* gp = a.gx_table + (size_t)tok[trow] * 3 * H + unit;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) gx[gt] = gp[gt * H] + gxc[gt];
      }
      if (w < 2 && !(DBG & 32)) {                        // wave mt: 16 rows x 16 units, all three gates, full K
        const int mt = w;
        const unsigned abase = (unsigned)((size_t)t * a.Bpad * Hk * 2 + (size_t)(g * 2 + mt) * NKS * 1024 + lane * 16);
        const char* bp = whl + (size_t)c16 * ldw;
        f32x4 acc[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NKS_C > 0) {
          u32x4 av[NKS_C];
#pragma unroll
          for (int ks = 0; ks < NKS_C; ++ks) av[ks] = ld16<LDAUX>(hrs, abase + ks * 1024);
#pragma unroll
          for (int ks = 0; ks < NKS_C; ++ks) {
#pragma unroll
            for (int gt = 0; gt < 3; ++gt)
              acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, av[ks]),
                  __builtin_bit_cast(bf16x8, *(const uint4*)(bp + (size_t)gt * UNITS * ldw + (ks >> 2) * 256 + sw[ks & 3])), acc[gt], 0, 0, 0);
          }
        } else {
          for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 av = ld16<LDAUX>(hrs, abase + ks * 1024);
#pragma unroll
            for (int gt = 0; gt < 3; ++gt)
              acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, av),
                  __builtin_bit_cast(bf16x8, *(const uint4*)(bp + (size_t)gt * UNITS * ldw + (((ks * 4 + q4) ^ (c16 & mask)) * 16))), acc[gt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
          for (int j = 0; j < 4; ++j) gh[(gt * GROWS + mt * 16 + q4 * 4 + j) * UNITS + c16] = acc[gt][j];
      }
      __syncthreads();
      {
        const float hn_ = gh[(2 * GROWS + trow) * UNITS + tun] + bh[2];
        const float r_ = fast_sigmoid(gx[0] + gh[(0 * GROWS + trow) * UNITS + tun] + bh[0]);
        const float z_ = fast_sigmoid(gx[1] + gh[(1 * GROWS + trow) * UNITS + tun] + bh[1]);
        const float n_ = fast_tanh(gx[2] + r_ * hn_);
        hreg = live ? n_ + z_ * (hreg - n_) : 0.f;
        tile[trow * UNITS + tun] = f2bf(hreg);
      }
      __syncthreads();
      STAMP(2);
      if (tid < GROWS * 2) {                             // publish: 32 rows x 32 bytes, 16-byte write-through stores
        const int row = (tid >> 5) * 16 + (tid & 15), ch = (tid >> 4) & 1, k = s * UNITS + ch * 8;
        const u32x4 v = *(const u32x4*)(tile + row * UNITS + ch * 8);
        const unsigned off = (unsigned)((size_t)(t + 1) * a.Bpad * Hk * 2 + (size_t)((g * 2 + (tid >> 5)) * NKS + (k >> 5)) * 1024 +
                                        ((k >> 3) & 3) * 256 + (tid & 15) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, hrs, off, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 64) st_relaxed(a.flag_a + r, (u32)(t + 1));   // wave 1: wave 0's polls never queue behind this store
      STAMP(3);
    } else if (t == a.L) {
      break;
    }
    if (voc) {
      // ---- (3) h_{t+1} of every row x the resident columns -------------------------------------------------------
      if (w == 0 && !dead && !(DBG & 64)) {
        dead = !wait_all_ge(a.flag_a, a.nwg, (u32)(t + 1), a.err, lane);
      }
      STAMP(4);
      __syncthreads();
      // lane c16 = i (< 8) of each 16-lane row keeps the running record of the wave's row i (row tile i >> 2, accumulator
      // element i & 3): a few registers of state while the fragments are live.  The symbol needs only (best, column): that
      // record is published first; max / sum-exp (the log-probability's normaliser: 40 exponentials and twice the
      // reductions per lane) follow AFTER the hand-off when the share is one chunk, from the accumulators still in
      // registers, and are folded once, after the last token.
      float km = -INFINITY, ksum = 0.f, kb = -INFINITY, kx = 0.f;
      int ki = -1;
      RngState rst{0, 0};
      if (a.rng) rst = *a.rng;
      // which 32 rows this wave takes is rotated by the workgroup's index within its XCD
      const int wr = (w + (r >> 3)) % (a.Bpad / 32);
      const bool vwave = w * 32 < a.Bpad, defer = NCT <= CT;
      f32x4 acc[2][CT];
      float bv[CT];
      int colv[CT];
      auto sumexp = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int mt = i >> 2, j = i & 3;
          float x[CT], mx = -INFINITY, sc = 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            x[ct] = colv[ct] >= 0 ? (acc[mt][ct][j] + bv[ct]) * a.inv_temp : -INFINITY;
            mx = fmaxf(mx, x[ct]);
          }
          mx = row16_max(mx);
          if (mx > -INFINITY) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) sc += __expf(x[ct] - mx);
          }
          sc = row16_sum(sc);
          if (c16 == i) fold_ms(km, ksum, mx, sc);
        }
      };
      if (vwave) {
        const unsigned abase = (unsigned)((size_t)(t + 1) * a.Bpad * Hk * 2 + (size_t)(wr * 2) * NKS * 1024 + lane * 16);
        const unsigned astep = (unsigned)(NKS * 1024);       // second row tile
        u32x4 av[2][NKS_C > 0 ? NKS_C : 1];
        if constexpr (NKS_C > 0) {
          if (!(DBG & 1))
#pragma unroll
          for (int ks = 0; ks < NKS_C; ++ks) {
            av[0][ks] = ld16<LDAUX>(hrs, abase + ks * 1024);
            av[1][ks] = ld16<LDAUX>(hrs, abase + astep + ks * 1024);
          }
        }
#pragma unroll 1
        for (int c0 = 0; c0 < NCT; c0 += CT) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[0][ct] = acc[1][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
          // this lane's columns of the chunk and their bias (requested before the MFMAs)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            colv[ct] = r * a.NC + (c0 + ct) * 16 + c16;
            const bool ok = c0 + ct < NCT && colv[ct] < a.V;
            bv[ct] = a.bp[ok ? colv[ct] : 0];
            if (!ok) colv[ct] = -1;
          }
          // tiles past the share's end read the last tile again (results unused): no branches around the MFMAs
          int toff[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) toff[ct] = (min(c0 + ct, NCT - 1) * 16 + c16) * ldw;
          if constexpr (NKS_C > 0) {
            if (!(DBG & 2))
#pragma unroll
            for (int ks = 0; ks < NKS_C; ++ks) {
              uint4 fb[CT];
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) fb[ct] = *(const uint4*)(wpl + toff[ct] + (ks >> 2) * 256 + sw[ks & 3]);
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, fb[ct]);
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[0][ks]), b, acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[1][ks]), b, acc[1][ct], 0, 0, 0);
              }
            }
          } else {
            for (int ks = 0; ks < NKS; ++ks) {
              const bf16x8 a0 = __builtin_bit_cast(bf16x8, ld16<LDAUX>(hrs, abase + ks * 1024));
              const bf16x8 a1 = __builtin_bit_cast(bf16x8, ld16<LDAUX>(hrs, abase + astep + ks * 1024));
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *(const uint4*)(wpl + toff[ct] + (((ks * 4 + q4) ^ (c16 & mask)) * 16)));
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[1][ct], 0, 0, 0);
              }
            }
          }
          STAMP(5);
          // this lane's columns of the chunk (ascending): best allowed scaled logit and its column per row; the 16 lanes of
          // a row reduce with DPP (lowest column among equal values) and lane i keeps row i
          if (!(DBG & 4))
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int mt = i >> 2, j = i & 3;
            float bst = -INFINITY, bx = 0.f;
            int bi = -1;
            if (a.rng) {                                     // (wave-uniform) the multinomial draw: arg-max of x - log q
              const uint32_t row = (uint32_t)(wr * 32 + mt * 16 + q4 * 4 + j);
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const float x = (acc[mt][ct][j] + bv[ct]) * a.inv_temp;
                if (colv[ct] >= 0 && colv[ct] != a.mask_col) {
                  uint32_t rnd[4];
                  philox4x32_10((uint32_t)colv[ct], row, a.site0 + (uint32_t)t,
                                (uint32_t)rst.step ^ (uint32_t)(rst.step >> 32) * 0x9E3779B9u, (uint32_t)rst.seed,
                                (uint32_t)(rst.seed >> 32), rnd);
                  const float q = -__logf(((float)(rnd[0] >> 8) + 0.5f) * (1.0f / 16777216.0f));     // Exp(1), u in (0, 1)
                  const float xs = x - __logf(q);
                  if (xs > bst) {                            // ascending: the first maximum stays
                    bst = xs;
                    bx = x;
                    bi = colv[ct];
                  }
                }
              }
            } else {
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const float x = (acc[mt][ct][j] + bv[ct]) * a.inv_temp;
                if (colv[ct] >= 0 && colv[ct] != a.mask_col && x > bst) {   // ascending: the first maximum stays
                  bst = x;
                  bi = colv[ct];
                }
              }
            }
            const float brow = row16_max(bst);
            const int mine = bi;
            bi = row16_min_int((bst == brow && bi >= 0) ? bi : 0x7fffffff);
            float xw = 0.f;
            if (a.rng) xw = row16_max((mine == bi && bi != 0x7fffffff) ? bx : -INFINITY);   // the winner's own logit
            if (c16 == i) {
              const int cand = bi == 0x7fffffff ? -1 : bi;
              if (a.rng && cand >= 0 && (brow > kb || (brow == kb && (ki < 0 || cand < ki)))) kx = xw;   // fold_best's rule
              fold_best(kb, ki, brow, cand);
            }
          }
          if (!defer) sumexp();
        }
        STAMP(6);
        if (c16 < 8 && !(DBG & 8)) {
          const int row = wr * 32 + (c16 >> 2) * 16 + q4 * 4 + (c16 & 3);
          const u32x2 v = {__builtin_bit_cast(u32, kb), (u32)ki};
          __builtin_amdgcn_raw_buffer_store_b64(v, prs, (unsigned)((((size_t)t * a.NV + r) * a.Bpad + row) * 8), 0, AUX_SC1);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 64) st_relaxed(a.flag_b + r, (u32)(t + 1));
      if (vwave) {                                           // nobody waits for these
        if (defer) sumexp();
        if (c16 < 8) {
          const int row = wr * 32 + (c16 >> 2) * 16 + q4 * 4 + (c16 & 3);
          const u32x2 v = {__builtin_bit_cast(u32, km), __builtin_bit_cast(u32, ksum)};
          __builtin_amdgcn_raw_buffer_store_b64(v, prs, a.part_b + (unsigned)((((size_t)t * a.NV + r) * a.Bpad + row) * 8), 0,
                                                AUX_SC1);
          // (agent scope = write-through, like the records: the reader sits behind another XCD's L2)
          if (a.rng) __hip_atomic_store(a.xsel + ((size_t)t * a.NV + r) * a.Bpad + row, kx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      STAMP(7);
    }
  }
  if (live) a.hn[(size_t)brow * H + unit] = hreg;
  // ---- (4) log-probabilities of the emitted symbols: once every record of every token is there, unit (t, 32 rows) folds
  //      its 256 (best, column) and (max, sum-exp) records; units are dealt over all workgroups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (voc && tid == 64) st_relaxed(a.flag_b + r, (u32)(a.L + 1));
  if (w == 0 && !dead && !(DBG & 64)) dead = !wait_all_ge(a.flag_b, a.nwg, (u32)(a.L + 1), a.err, lane);
  __syncthreads();
  const int nrb = a.Bpad / GROWS;
  for (int u = r; u < a.L * nrb; u += a.nwg) {
    const int tt = u / nrb, rb = u % nrb, j = lane & 31, half = lane >> 5;
    float m = -INFINITY, sm = 0.f, best = -INFINITY;
    int idx = -1;
    u32x2 va[MAX_WGS / 16], vb[MAX_WGS / 16];
    const unsigned base = (unsigned)((((size_t)tt * a.NV) * a.Bpad + rb * GROWS + j) * 8);
    const unsigned kstep = (unsigned)(a.Bpad * 8);
#pragma unroll
    for (int i = 0; i < MAX_WGS / 16; ++i) {
      const int k = 2 * (w + 8 * i) + half;
      if (k < a.NV) {
        va[i] = __builtin_amdgcn_raw_buffer_load_b64(prs, base + k * kstep, 0, LDAUX);
        vb[i] = __builtin_amdgcn_raw_buffer_load_b64(prs, a.part_b + base + k * kstep, 0, LDAUX);
      }
    }
#pragma unroll
    for (int i = 0; i < MAX_WGS / 16; ++i) {
      const int k = 2 * (w + 8 * i) + half;
      if (k < a.NV) {
        const f32x2 fa = __builtin_bit_cast(f32x2, va[i]), fb = __builtin_bit_cast(f32x2, vb[i]);
        fold(m, sm, best, idx, fb[0], fb[1], fa[0], (int)va[i][1]);
      }
    }
    fold(m, sm, best, idx, __shfl_xor(m, 32), __shfl_xor(sm, 32), __shfl_xor(best, 32), __shfl_xor(idx, 32));
    if (lane < 32) {
      const u32x4 o = {__builtin_bit_cast(u32, m), __builtin_bit_cast(u32, sm), __builtin_bit_cast(u32, best), (u32)idx};
      *(u32x4*)(gh + (w * GROWS + j) * 4) = o;
    }
    __syncthreads();
    if (tid < GROWS) {
      m = -INFINITY, sm = 0.f, best = -INFINITY, idx = -1;
#pragma unroll
      for (int ww = 0; ww < NT / 64; ++ww) {
        const u32x4 o = *(const u32x4*)(gh + (ww * GROWS + tid) * 4);
        const f32x4 f = __builtin_bit_cast(f32x4, o);
        fold(m, sm, best, idx, f[0], f[1], f[2], (int)o[3]);
      }
      const int row = rb * GROWS + tid;
      if (row < a.B) {
        // sample: `best` is the perturbed score; the drawn symbol's own logit is in the x image of the workgroup that owns it
        const float xv = (a.rng && idx >= 0) ? __hip_atomic_load(a.xsel + ((size_t)tt * a.NV + idx / a.NC) * a.Bpad + row,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : best;
        a.logp[(size_t)row * a.L + tt] = xv - (m + __logf(sm));
      }
    }
    __syncthreads();
  }
#ifdef PV_DECODE_STAMPS
  if (tid == 0 && (r < 4 || r >= a.nwg - 4))
    for (int i = 0; i < 8; ++i) a.err[576 + (r < 4 ? r : 8 - (a.nwg - r)) * 8 + i] = stamp_acc[i];
#endif
}

// slot 0 of the published states = bf16(h0) (zero rows / columns of padding); flags and the error word cleared
__global__ void decode_init_kernel(const float* __restrict__ h0, int B, int H, int Hk, int Bpad, bf16_t* __restrict__ hb,
                                   u32* __restrict__ sync, int sync_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sync_words) sync[i] = 0;
  if (i < Bpad * Hk) {
    const int row = i / Hk, k = i % Hk;
    const size_t o = ((size_t)((row >> 4) * (Hk / 32) + (k >> 5)) * 4 + ((k >> 3) & 3)) * 128 + (row & 15) * 8 + (k & 7);
    hb[o] = f2bf(row < B && k < H && h0 ? h0[(size_t)row * H + k] : 0.f);
  }
}

int g_cus = 0;
int device_cus() {
  if (g_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    g_cus = n;
  }
  return g_cus;
}

struct DecodePlan {
  int Hk, Bpad, G, S, nwg, NV, NC;
  size_t lds, hb_bytes, part_bytes;
};

int plan_decode(int B, int H, int V, int L, int cus, DecodePlan* p) {
  PV_REQUIRE(B > 0 && H > 0 && V > 0 && L > 0, "pv_greedy_decode: B=%d H=%d V=%d L=%d must be positive", B, H, V, L);
  p->Hk = round_up(H, 32);
  p->Bpad = round_up(B, GROWS);
  p->G = p->Bpad / GROWS;
  p->S = cdiv(H, UNITS);
  const int wg_gru = 8 * (p->S - 1) + p->G;
  if (p->G > 8 || wg_gru > cus || cus > MAX_WGS) {
    pv_set_error("pv_greedy_decode: B=%d, H=%d need %d row groups (<= 8) and %d workgroups on a device of %d CUs (<= %d)", B,
                 H, p->G, wg_gru, cus, MAX_WGS);
    return PV_ERR_LIMIT;
  }
  p->nwg = wg_gru > cdiv(V, 16) ? wg_gru : (cdiv(V, 16) < cus ? cdiv(V, 16) : cus);
  p->NV = p->nwg < cdiv(V, 16) ? p->nwg : cdiv(V, 16);
  p->NC = round_up(cdiv(V, p->NV), 16);
  const size_t ldw = (size_t)p->Hk * 2;
  p->lds = (48 + (size_t)p->NC) * ldw + 3 * GROWS * UNITS * 4 + GROWS * UNITS * 2 + GROWS * 4;
  if (p->lds > 160 * 1024 - 64) {
    pv_set_error("pv_greedy_decode: V=%d over %d workgroups is %d resident columns of %zu bytes: %zu bytes of LDS > 160 KiB",
                 V, p->NV, p->NC, ldw, p->lds);
    return PV_ERR_LIMIT;
  }
  p->hb_bytes = (size_t)(L + 1) * p->Bpad * p->Hk * 2;
  p->part_bytes = (size_t)L * p->Bpad * p->NV * sizeof(DecPart);
  if (p->hb_bytes >= (1ull << 31) || p->part_bytes >= (1ull << 31)) {
    pv_set_error("pv_greedy_decode: L=%d is too long for 32-bit buffer offsets", L);
    return PV_ERR_LIMIT;
  }
  return PV_OK;
}

template <class K>
int launch_decode(K kernel, const DecodeArgs& a, size_t lds, hipStream_t st) {
  PV_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3(a.nwg), dim3(NT), lds, st, a);
  PV_LAUNCH_CHECK("greedy_decode_kernel");
  return PV_OK;
}

}  // namespace

extern "C" {

size_t pv_greedy_decode_ws_bytes(int B, int H, int V, int L) {
  if (B <= 0 || H <= 0 || V <= 0 || L <= 0) return 0;
  const size_t Bpad = round_up(B, GROWS), Hk = round_up(H, 32);
  return SYNC_BYTES + ((size_t)(L + 1) * Bpad * Hk * 2 + 255) / 256 * 256 + (size_t)L * Bpad * MAX_WGS * sizeof(DecPart);
}

size_t pv_sample_decode_ws_bytes(int B, int H, int V, int L) {
  if (B <= 0 || H <= 0 || V <= 0 || L <= 0) return 0;
  return (pv_greedy_decode_ws_bytes(B, H, V, L) + 255) / 256 * 256 + (size_t)L * round_up(B, GROWS) * MAX_WGS * sizeof(float);
}

static int decode_impl(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                       const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                       int sos_id, int mask_col, float inv_temp, const void* rng_state, int site0, void* workspace,
                       int64_t* seq, float* logp, float* hn, void* stream);

int pv_greedy_decode(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                     const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                     int sos_id, int mask_col, float inv_temp, void* workspace, int64_t* seq, float* logp, float* hn,
                     void* stream) {
  return decode_impl(gx_table, gx_const, whh_bf16, bhh, wproj_bf16, ldwp, bproj, h0, B, H, V, L, sos_id, mask_col, inv_temp,
                     nullptr, 0, workspace, seq, logp, hn, stream);
}

int pv_sample_decode(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                     const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                     int sos_id, int mask_col, float inv_temp, const void* rng_state, int site0, void* workspace,
                     int64_t* seq, float* logp, float* hn, void* stream) {
  PV_REQUIRE(rng_state, "pv_sample_decode: null generator state");
  return decode_impl(gx_table, gx_const, whh_bf16, bhh, wproj_bf16, ldwp, bproj, h0, B, H, V, L, sos_id, mask_col, inv_temp,
                     rng_state, site0, workspace, seq, logp, hn, stream);
}

static int decode_impl(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                       const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                       int sos_id, int mask_col, float inv_temp, const void* rng_state, int site0, void* workspace,
                       int64_t* seq, float* logp, float* hn, void* stream) {
  PV_REQUIRE(gx_table && whh_bf16 && bhh && wproj_bf16 && bproj && workspace && seq && logp && hn,
             "pv_greedy_decode: null argument");
  PV_REQUIRE(sos_id >= 0 && sos_id < V, "pv_greedy_decode: sos_id %d outside the vocabulary of %d", sos_id, V);
  const int cus = device_cus();
  PV_REQUIRE(cus > 0, "pv_greedy_decode: cannot read the device's CU count");
  DecodePlan p;
  int rc = plan_decode(B, H, V, L, cus, &p);
  if (rc != PV_OK) return rc;
  PV_REQUIRE(ldwp >= p.Hk && ldwp % 8 == 0, "pv_greedy_decode: ldwp=%d must be a multiple of 8 and >= %d", ldwp, p.Hk);
  PV_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)whh_bf16 & 15) == 0 && ((uintptr_t)wproj_bf16 & 15) == 0,
             "pv_greedy_decode: workspace must be 256-byte aligned, weights 16-byte aligned");
  char* ws = (char*)workspace;
  DecodeArgs a;
  a.gx_table = gx_table;
  a.gx_const = gx_const;
  a.whh = (const bf16_t*)whh_bf16;
  a.bhh = bhh;
  a.wp = (const bf16_t*)wproj_bf16;
  a.bp = bproj;
  a.h0 = h0;
  a.err = (u32*)ws;
  a.flag_a = (u32*)ws + 64;
  a.flag_b = (u32*)ws + 64 + MAX_WGS;
  a.hb = (bf16_t*)(ws + SYNC_BYTES);
  a.part = (DecPart*)(ws + SYNC_BYTES + (p.hb_bytes + 255) / 256 * 256);
  a.seq = seq;
  a.logp = logp;
  a.hn = hn;
  a.L = L, a.B = B, a.H = H, a.Hk = p.Hk, a.Bpad = p.Bpad, a.V = V, a.G = p.G, a.S = p.S, a.NV = p.NV, a.NC = p.NC;
  a.nwg = p.nwg, a.sos = sos_id, a.mask_col = mask_col, a.ldwp = ldwp, a.inv_temp = inv_temp;
  a.hb_bytes = (unsigned)p.hb_bytes, a.part_bytes = (unsigned)p.part_bytes, a.part_b = (unsigned)(p.part_bytes / 2);
  a.rng = (const RngState*)rng_state;
  a.site0 = (unsigned)site0;
  // (the x image of a sampling decode lies behind the greedy workspace: pv_sample_decode_ws_bytes)
  a.xsel = rng_state ? (float*)(ws + (pv_greedy_decode_ws_bytes(B, H, V, L) + 255) / 256 * 256) : nullptr;
  static const int dbg = getenv("PV_DECODE_DBG") ? atoi(getenv("PV_DECODE_DBG")) : 0;
  a.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  if (p.Hk != p.S * UNITS) PV_HIP(hipMemsetAsync(a.hb, 0, p.hb_bytes, st));   // padding columns no slice ever writes
  const int n0 = p.Bpad * p.Hk > (int)(SYNC_BYTES / 4) ? p.Bpad * p.Hk : (int)(SYNC_BYTES / 4);
  hipLaunchKernelGGL(decode_init_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, st, h0, B, H, p.Hk, p.Bpad, a.hb, (u32*)ws,
                     (int)(SYNC_BYTES / 4));
  PV_LAUNCH_CHECK("decode_init_kernel");
  static const int plain = getenv("PEVAE_DECODE_SC1_LOADS") ? 0 : 1;
  if (p.Hk == 512) return plain ? launch_decode(greedy_decode_kernel<16, true>, a, p.lds, st)
                                : launch_decode(greedy_decode_kernel<16, false>, a, p.lds, st);
  return plain ? launch_decode(greedy_decode_kernel<0, true>, a, p.lds, st)
               : launch_decode(greedy_decode_kernel<0, false>, a, p.lds, st);
}

int pv_greedy_decode_error(const void* workspace, int* flag) {
  PV_REQUIRE(workspace && flag, "pv_greedy_decode_error: null argument");
  PV_HIP(hipMemcpy(flag, workspace, sizeof(int), hipMemcpyDeviceToHost));
  return PV_OK;
}

}  // extern "C"
