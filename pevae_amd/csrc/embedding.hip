// Embedding lookup and its backward for gfx950: coalesced row gathers and an atomic-free,
// bitwise-reproducible sorted scatter-add.
//
// Reference: nn.Embedding gathers at models/dialog_models.py:758-765 (user / item / rating / word) and
// models/enc2dec/decoders.py:174 (decoder-side re-gather of the shared word table); the backward is
// torch's embedding_dense_backward (atomics on GPU -> run-to-run bit differences).  Here the backward
// sorts (row id, occurrence) pairs with a STABLE radix sort, then one wave per distinct table row sums
// that row's occurrences in occurrence order: no atomics, same bits every run, same order as a CPU
// index_add_ over the flattened ids.
//
// Occurrence r = i0*n1 + i1 reads ids[i0*s0 + i1*s1] and lives in row i0*n1p + i1 of the gathered matrix
// (n1p >= n1 leaves the GRU's batch-padding rows untouched), so the same kernels serve batch-major ([B,L] ids
// -> rows b*L+t) and time-major (rows t*Bpad+b, what the GRU wants) layouts.
#include "common.h"
#include "../../include/pevae_hip.h"
#include <hipcub/hipcub.hpp>

namespace {

// one wave per output row; 16-byte reads when D % 4 == 0
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, long num_rows, int D,
                                                      const int64_t* __restrict__ ids, int n, int n1, int n1p,
                                                      long s0, long s1, float* __restrict__ out, int ldo,
                                                      bf16_t* __restrict__ outb, int ldb, int* __restrict__ oob) {
  const int lane = threadIdx.x & 63;
  const int occ = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (occ >= n) return;
  int64_t id = ids[(long)(occ / n1) * s0 + (long)(occ % n1) * s1];
  const long r = (long)(occ / n1) * n1p + occ % n1;
  if (id < 0 || id >= num_rows) {          // the reference would raise an index error here
    if (lane == 0 && oob) atomicAdd(oob, 1);
    id = id < 0 ? 0 : num_rows - 1;
  }
  const float* src = table + id * D;
  for (int c = lane; c < D; c += 64) {
    const float v = src[c];
    if (out) out[(size_t)r * ldo + c] = v;
    if (outb) outb[(size_t)r * ldb + c] = f2bf(v);
  }
  if (outb)
    for (int c = D + lane; c < ldb; c += 64) outb[(size_t)r * ldb + c] = 0;   // K padding for the MFMA GEMM
}

__global__ void fill_keys_kernel(const int64_t* __restrict__ ids, int n, int n1, int n1p, long s0, long s1,
                                 long num_rows, int* __restrict__ keys, int* __restrict__ vals) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t id = ids[(long)(r / n1) * s0 + (long)(r % n1) * s1];
  keys[r] = (id < 0 || id >= num_rows) ? (int)num_rows : (int)id;   // out-of-range ids sort last, get no gradient
  vals[r] = (r / n1) * n1p + r % n1;                                // row of the gradient matrix
}

// Stable sort of the (row id, occurrence) pairs of one lookup: ONE workgroup, everything in LDS, keys made from the ids in the
// same launch (the hot path's lookups have n <= 16 384 occurrences and tables below 2^18 rows: one packed word
// key << 14 | occurrence per pair).  LSD radix sort, 4 bits a pass: thread t owns the E consecutive elements [tE, tE + E) of
// the current order, counts its digits (sixteen 8-bit counters in two registers), the counts -- laid out [digit][thread] -- get
// one exclusive scan over the whole workgroup, and each thread scatters its elements in order to the offsets of its own
// column: equal digits keep their order.  Element p lives at word p + p / 32, so that the threads' chunks start in different
// banks.  Elements past n are all-ones: they stay last in every pass and are never written out.
// 1 024 threads when the images fit the 160 KiB of LDS beside their count matrix (n <= 13 312: the step's 12 800-occurrence
// lookups), 512 otherwise: the scatter is a serial chain per thread, so threads are what shortens a pass.
constexpr int SORT_N_MAX = 16384, SORT_OCC_BITS = 14, SORT_KEY_BITS = 18;
__device__ __forceinline__ int sort_at(int p) { return p + (p >> 5); }
static inline size_t sort_lds_bytes(int n, int threads) {
  const int span = cdiv(n, threads) * threads;
  return (size_t)2 * (span + (span >> 5) + 1) * 4 + 16 * threads * 2 + 64;
}
template <int SORT_T>
__global__ __launch_bounds__(SORT_T) void sort_pairs_lds_kernel(const int64_t* __restrict__ ids, int n, int n1, int n1p, long s0,
                                                               long s1, long num_rows, int bits, int* __restrict__ skeys,
                                                               int* __restrict__ svals) {
  extern __shared__ unsigned sort_sm[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int E = (n + SORT_T - 1) / SORT_T, span = E * SORT_T, words = span + (span >> 5) + 1;
  unsigned* src = sort_sm;
  unsigned* dst = sort_sm + words;
  unsigned short* cnt = (unsigned short*)(sort_sm + 2 * words);
  unsigned* wsum = (unsigned*)(cnt + 16 * SORT_T);
  for (int p = t; p < span; p += SORT_T) {
    unsigned v = 0xffffffffu;
    if (p < n) {
      const int64_t id = ids[(long)(p / n1) * s0 + (long)(p % n1) * s1];
      const unsigned key = (id < 0 || id >= num_rows) ? (unsigned)num_rows : (unsigned)id;   // out of range: last, no gradient
      v = key << SORT_OCC_BITS | (unsigned)p;
    }
    src[sort_at(p)] = v;
  }
  __syncthreads();
  for (int shift = SORT_OCC_BITS; shift < SORT_OCC_BITS + bits; shift += 4) {
    unsigned long long c0 = 0, c1 = 0;
    for (int j = 0; j < E; ++j) {
      const unsigned d = (src[sort_at(t * E + j)] >> shift) & 15u;
      if (d < 8) c0 += 1ull << (8 * d);
      else c1 += 1ull << (8 * (d - 8));
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      cnt[d * SORT_T + t] = (unsigned short)((c0 >> (8 * d)) & 255u);
      cnt[(d + 8) * SORT_T + t] = (unsigned short)((c1 >> (8 * d)) & 255u);
    }
    __syncthreads();
    // exclusive scan of the 16 * SORT_T counts in [digit][thread] order: 16 consecutive entries per thread, then the threads
    unsigned loc[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      loc[k] = sum;
      sum += cnt[16 * t + k];
    }
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned base = incl - sum;
    for (int k = 0; k < w; ++k) base += wsum[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) cnt[16 * t + k] = (unsigned short)(base + loc[k]);
    __syncthreads();
    for (int j = 0; j < E; ++j) {
      const unsigned v = src[sort_at(t * E + j)];
      const unsigned d = (v >> shift) & 15u;
      const unsigned o = cnt[d * SORT_T + t];
      cnt[d * SORT_T + t] = (unsigned short)(o + 1);
      dst[sort_at((int)o)] = v;
    }
    __syncthreads();
    unsigned* tmp = src;
    src = dst;
    dst = tmp;
  }
  for (int p = t; p < n; p += SORT_T) {
    const unsigned v = src[sort_at(p)];
    const int r = (int)(v & ((1u << SORT_OCC_BITS) - 1));
    skeys[p] = (int)(v >> SORT_OCC_BITS);
    svals[p] = (r / n1) * n1p + r % n1;                                 // row of the gradient matrix
  }
}

// One wave per (sorted position that starts a piece, 64-column block).  A segment (all occurrences of one row id) is
// summed in occurrence order, so a row with fewer than SEG_CH occurrences gets the bits of a sequential CPU
// index_add_.  64 gradient rows are kept in flight per wave: the row ids of a batch are fetched with one coalesced load
// and broadcast with v_readlane, all 64 row loads are issued back to back, and only then added in order; the segment
// end is found with ballots over 64 keys at a time.  A Zipf-heavy word (~10^3 occurrences at the Yelp shape) would
// still be ~20 dependent batches of ~2.5 us, so long segments are cut into pieces: the head piece runs to the first
// position that is a multiple of SEG_CH and at least SEG_CH past the head, every following multiple of SEG_CH inside the
// segment starts another piece.  The head piece goes to the table row, the others to `part`, and
// segment_combine_kernel adds those to the row in position order: a fixed association, deterministic run to run.
constexpr int SEG_NB = 64;
constexpr int SEG_CH = 256;
__global__ __launch_bounds__(256) void segment_sum_kernel(const int* __restrict__ keys, const int* __restrict__ vals,
                                                           int n, const float* __restrict__ grad, int ldg, int D,
                                                           long num_rows, long padding_idx, int accumulate,
                                                           float* __restrict__ dtable, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int key = keys[i];
  if (key < 0 || key >= num_rows || key == padding_idx) return;
  const bool head = i == 0 || keys[i - 1] != key;
  const bool cut = !head && i % SEG_CH == 0 && keys[i - SEG_CH] == key;      // i >= SEG_CH: i is a non-zero multiple
  if (!head && !cut) return;
  const int stop = min(n, head ? (i + 2 * SEG_CH - 1) / SEG_CH * SEG_CH : i + SEG_CH);   // where the next piece would start
  int end = i + 1;
  while (end < stop) {                                     // wave-uniform: first position whose key differs
    const int p = end + lane;
    const bool same = p < stop && keys[p] == key;
    const unsigned long long m = __ballot(same);
    if (m == ~0ull) {
      end += 64;
      continue;
    }
    end += __builtin_ctzll(~m);
    break;
  }
  end = min(end, stop);
  const int c = blockIdx.y * 64 + lane;
  const bool cl = c < D;
  float* dst = head ? dtable + (size_t)key * D : part + (size_t)(i / SEG_CH) * D;
  float s = (head && accumulate && cl) ? dst[c] : 0.f;
  for (int j0 = i; j0 < end; j0 += SEG_NB) {
    const int cnt = min(SEG_NB, end - j0);
    const int myv = (lane < cnt) ? vals[j0 + lane] : 0;
    float r[SEG_NB];
#pragma unroll
    for (int k = 0; k < SEG_NB; ++k) {
      const int v = __builtin_amdgcn_readlane(myv, k);
      r[k] = (k < cnt && cl) ? grad[(size_t)v * ldg + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SEG_NB; ++k)
      if (k < cnt) s += r[k];
  }
  if (cl) dst[c] = s;
}

// One wave per (multiple q of SEG_CH, 64-column block): if q starts the FIRST cut piece of its segment, add that piece
// and the following ones to the table row, in position order.
__global__ __launch_bounds__(64) void segment_combine_kernel(const int* __restrict__ keys, int n, int D, long num_rows,
                                                             long padding_idx, const float* __restrict__ part,
                                                             float* __restrict__ dtable) {
  const int q = (blockIdx.x + 1) * SEG_CH;
  if (q >= n) return;
  const int key = keys[q];
  if (key < 0 || key >= num_rows || key == padding_idx) return;
  if (keys[q - SEG_CH] != key) return;                                   // q is not a cut
  if (q >= 2 * SEG_CH && keys[q - 2 * SEG_CH] == key) return;            // ... or not the first one of its segment
  const int c = blockIdx.y * 64 + threadIdx.x;
  if (c >= D) return;
  float* dst = dtable + (size_t)key * D;
  float s = dst[c];
  for (int p = q; p < n && keys[p] == key; p += SEG_CH) s += part[(size_t)(p / SEG_CH) * D + c];
  dst[c] = s;
}

// Data-parallel exchange of a table whose local gradient is non-zero only in the n rows looked up this step.
// pack: occurrence r sends its id (as a float: exact below 2^24) and the table's gradient row if r is the FIRST occurrence
// of that id, zeros otherwise, so that a receiver may add every occurrence.  One workgroup per occurrence.
__global__ __launch_bounds__(256) void sparse_pack_kernel(const float* __restrict__ grad, long num_rows, int D,
                                                           const int64_t* __restrict__ ids, int n, float* __restrict__ out_ids,
                                                           float* __restrict__ out_rows) {
  const int r = blockIdx.x;
  const int64_t id = ids[r];
  int dup = id < 0 || id >= num_rows;                        // an id outside the table sends zeros (and is dropped by the merge)
  for (int j = threadIdx.x; j < r; j += 256) dup |= ids[j] == id;
  dup = __syncthreads_or(dup);
  if (threadIdx.x == 0) out_ids[r] = (float)id;
  const float* src = grad + (size_t)(dup ? 0 : id) * D;
  for (int c = threadIdx.x; c < D; c += 256) out_rows[(size_t)r * D + c] = dup ? 0.f : src[c];
}

// merge: `gathered` holds every rank's block [ids n][rows n x D] at rank_stride floats from each other.  Entry e = (rank,
// occurrence); the workgroup of the first entry of an id sums that id's rows over all entries in entry order -- the same
// order on every rank, no atomics -- and OVERWRITES the table's gradient row (every locally touched row is among them).
__global__ __launch_bounds__(256) void sparse_merge_kernel(const float* __restrict__ gathered, int world, long rank_stride,
                                                            int n, int D, long num_rows, float* __restrict__ table) {
  extern __shared__ float ids_s[];                           // world * n ids, then the list of this id's entries
  __shared__ int wave_cnt[4];
  const int E = world * n, e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int* list = (int*)(ids_s + E);
  for (int j = tid; j < E; j += 256) ids_s[j] = gathered[(long)(j / n) * rank_stride + j % n];
  __syncthreads();
  const float idf = ids_s[e];
  int dup = 0;
  for (int j = tid; j < e; j += 256) dup |= ids_s[j] == idf;
  if (__syncthreads_or(dup)) return;                         // not the first entry of this id (workgroup-uniform)
  const long id = (long)idf;
  if (id < 0 || id >= num_rows) return;
  int total = 0;                                             // entries e .. E-1 with this id, compacted in entry order
  for (int base = e; base < E; base += 256) {
    const int j = base + tid;
    const bool m = j < E && ids_s[j] == idf;
    const unsigned long long bal = __ballot(m);
    if (lane == 0) wave_cnt[w] = __popcll(bal);
    __syncthreads();
    int off = total;
    for (int k = 0; k < w; ++k) off += wave_cnt[k];
    if (m) list[off + __popcll(bal & ((1ull << lane) - 1))] = j;
    total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  for (int c = tid; c < D; c += 256) {
    float acc = 0.f;
    for (int k = 0; k < total; ++k) {
      const int j = list[k];
      acc += gathered[(long)(j / n) * rank_stride + n + (long)(j % n) * D + c];
    }
    table[id * D + c] = acc;
  }
}

size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  hipcub::DeviceRadixSort::SortPairs((void*)nullptr, bytes, (const int*)nullptr, (int*)nullptr,
                                     (const int*)nullptr, (int*)nullptr, n);
  return bytes;
}

}  // namespace

extern "C" {

int pv_embedding_gather(const float* table, long num_rows, int D, const int64_t* ids, int n, int n1, int n1p,
                        long s0, long s1, float* out, int ldo, void* out_bf16, int ldb, int* oob_counter,
                        void* stream) {
  PV_REQUIRE(table && ids && (out || out_bf16), "pv_embedding_gather: null argument");
  PV_REQUIRE(n >= 0 && n1 > 0 && n1p >= n1 && D > 0 && num_rows > 0, "pv_embedding_gather: bad sizes");
  PV_REQUIRE((!out || ldo >= D) && (!out_bf16 || ldb >= D), "pv_embedding_gather: leading dimension < D");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL(gather_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, table, num_rows, D, ids,
                     n, n1, n1p, s0, s1, out, ldo, (bf16_t*)out_bf16, ldb, oob_counter);
  PV_LAUNCH_CHECK("gather_kernel");
  return PV_OK;
}

size_t pv_embedding_bwd_ws_bytes(int n, int D) {
  const size_t a = round_up(n, 64) * sizeof(int);
  return 4 * a + ((sort_temp_bytes(n) + 255) & ~(size_t)255) + 256 + ((size_t)(n / SEG_CH + 1) * D * sizeof(float));
}

// keys + stable sort of one lookup's (row id, occurrence) pairs into the workspace (sorted keys at 2a, sorted gradient-row
// numbers at 3a, a = round_up(n, 64) ints)
static int embedding_sort(const int64_t* ids, int n, int n1, int n1p, long s0, long s1, long num_rows, char* ws, hipStream_t st) {
  const size_t a = round_up(n, 64) * sizeof(int);
  int* keys = (int*)ws;
  int* vals = (int*)(ws + a);
  int* skeys = (int*)(ws + 2 * a);
  int* svals = (int*)(ws + 3 * a);
  void* temp = ws + 4 * a;
  size_t temp_bytes = sort_temp_bytes(n);
  int bits = 1;
  while ((1l << bits) <= num_rows && bits < 31) ++bits;
  static const int own_sort = getenv("PV_EMB_OWN_SORT") ? atoi(getenv("PV_EMB_OWN_SORT")) : 1;
  if (own_sort && n <= SORT_N_MAX && bits <= SORT_KEY_BITS) {
    // the hot path: keys and the stable sort in one single-workgroup launch
    // threads: one wave up to 512 occurrences (the user / item lookups of a batch: no cross-wave steps), 1 024 when the
    // images fit the LDS beside their count matrix (n <= 13 312), 512 otherwise
    static size_t seen[3] = {0, 0, 0};
    const int shape = n <= 512 ? 0 : sort_lds_bytes(n, 1024) <= 160 * 1024 ? 2 : 1;
    const int threads = shape == 0 ? 64 : shape == 2 ? 1024 : 512;
    const void* fn = shape == 0 ? (const void*)sort_pairs_lds_kernel<64>
                                : shape == 2 ? (const void*)sort_pairs_lds_kernel<1024> : (const void*)sort_pairs_lds_kernel<512>;
    const size_t lds = sort_lds_bytes(n, threads);
    if (lds > 64 * 1024 && lds > seen[shape]) {
      PV_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      seen[shape] = lds;
    }
    if (shape == 0)
      hipLaunchKernelGGL(sort_pairs_lds_kernel<64>, dim3(1), dim3(64), lds, st, ids, n, n1, n1p, s0, s1, num_rows, bits, skeys, svals);
    else if (shape == 2)
      hipLaunchKernelGGL(sort_pairs_lds_kernel<1024>, dim3(1), dim3(1024), lds, st, ids, n, n1, n1p, s0, s1, num_rows, bits, skeys, svals);
    else
      hipLaunchKernelGGL(sort_pairs_lds_kernel<512>, dim3(1), dim3(512), lds, st, ids, n, n1, n1p, s0, s1, num_rows, bits, skeys, svals);
    PV_LAUNCH_CHECK("sort_pairs_lds_kernel");
  } else {
    // lookups beyond 16 384 occurrences or tables of 2^18 rows and more: the library's device-wide sort
    hipLaunchKernelGGL(fill_keys_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ids, n, n1, n1p, s0, s1, num_rows, keys, vals);
    PV_LAUNCH_CHECK("fill_keys_kernel");
    PV_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, skeys, vals, svals, n, 0, bits, st));
  }
  return PV_OK;
}

int pv_embedding_sort(const int64_t* ids, int n, int n1, int n1p, long s0, long s1, long num_rows, int D, void* workspace,
                      size_t ws_bytes, void* stream) {
  PV_REQUIRE(ids && workspace, "pv_embedding_sort: null argument");
  PV_REQUIRE(n >= 0 && n1 > 0 && n1p >= n1 && D > 0, "pv_embedding_sort: bad sizes");
  PV_REQUIRE(ws_bytes >= pv_embedding_bwd_ws_bytes(n, D), "pv_embedding_sort: workspace too small");
  PV_REQUIRE(num_rows < (1l << 30), "pv_embedding_sort: table rows must fit int32 sort keys");
  if (n == 0) return PV_OK;
  return embedding_sort(ids, n, n1, n1p, s0, s1, num_rows, (char*)workspace, (hipStream_t)stream);
}

int pv_embedding_bwd_sorted(const float* grad_rows, int ldg, int n, long num_rows, int D, long padding_idx, int accumulate,
                            float* dtable, void* workspace, size_t ws_bytes, void* stream) {
  PV_REQUIRE(grad_rows && dtable && workspace, "pv_embedding_bwd_sorted: null argument");
  PV_REQUIRE(n >= 0 && D > 0 && ldg >= D, "pv_embedding_bwd_sorted: bad sizes");
  PV_REQUIRE(ws_bytes >= pv_embedding_bwd_ws_bytes(n, D), "pv_embedding_bwd_sorted: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) PV_HIP(hipMemsetAsync(dtable, 0, (size_t)num_rows * D * sizeof(float), st));
  if (n == 0) return PV_OK;
  const size_t a = round_up(n, 64) * sizeof(int);
  char* ws = (char*)workspace;
  const int* skeys = (const int*)(ws + 2 * a);
  const int* svals = (const int*)(ws + 3 * a);
  float* part = (float*)(ws + 4 * a + ((sort_temp_bytes(n) + 255) & ~(size_t)255) + 256);   // one row per multiple of SEG_CH
  hipLaunchKernelGGL(segment_sum_kernel, dim3(cdiv(n, 4), cdiv(D, 64)), dim3(256), 0, st, skeys, svals, n, grad_rows, ldg, D,
                     num_rows, padding_idx, accumulate, dtable, part);
  PV_LAUNCH_CHECK("segment_sum_kernel");
  if (n > SEG_CH) {
    hipLaunchKernelGGL(segment_combine_kernel, dim3((n - 1) / SEG_CH, cdiv(D, 64)), dim3(64), 0, st, skeys, n, D, num_rows,
                       padding_idx, part, dtable);
    PV_LAUNCH_CHECK("segment_combine_kernel");
  }
  return PV_OK;
}

int pv_embedding_bwd(const float* grad_rows, int ldg, const int64_t* ids, int n, int n1, int n1p, long s0, long s1,
                     long num_rows, int D, long padding_idx, int accumulate, float* dtable, void* workspace,
                     size_t ws_bytes, void* stream) {
  PV_REQUIRE(grad_rows && ids && dtable && workspace, "pv_embedding_bwd: null argument");
  PV_REQUIRE(n >= 0 && n1 > 0 && n1p >= n1 && D > 0 && ldg >= D, "pv_embedding_bwd: bad sizes");
  PV_REQUIRE(ws_bytes >= pv_embedding_bwd_ws_bytes(n, D), "pv_embedding_bwd: workspace too small");
  PV_REQUIRE(num_rows < (1l << 30), "pv_embedding_bwd: table rows must fit int32 sort keys");
  if (n > 0) {
    int rc = embedding_sort(ids, n, n1, n1p, s0, s1, num_rows, (char*)workspace, (hipStream_t)stream);
    if (rc) return rc;
  }
  return pv_embedding_bwd_sorted(grad_rows, ldg, n, num_rows, D, padding_idx, accumulate, dtable, workspace, ws_bytes, stream);
}

int pv_sparse_pack(const float* table_grad, long num_rows, int D, const int64_t* ids, int n, float* out_ids,
                   float* out_rows, void* stream) {
  PV_REQUIRE(table_grad && ids && out_ids && out_rows && n > 0 && D > 0, "pv_sparse_pack: bad argument");
  PV_REQUIRE(num_rows > 0 && num_rows <= (1l << 24), "pv_sparse_pack: row ids travel as fp32: at most 2^24 table rows");
  hipLaunchKernelGGL(sparse_pack_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table_grad, num_rows, D, ids, n, out_ids, out_rows);
  PV_LAUNCH_CHECK("sparse_pack_kernel");
  return PV_OK;
}

int pv_sparse_merge(const float* gathered, int world, long rank_stride, int n, int D, long num_rows, float* table_grad,
                    void* stream) {
  PV_REQUIRE(gathered && table_grad && world > 0 && n > 0 && D > 0 && rank_stride >= (long)n * (D + 1),
             "pv_sparse_merge: bad argument");
  const size_t lds = (size_t)world * n * (sizeof(float) + sizeof(int));
  PV_REQUIRE(lds <= 64 * 1024, "pv_sparse_merge: %d x %d ids exceed the 8192 the kernel stages", world, n);
  hipLaunchKernelGGL(sparse_merge_kernel, dim3(world * n), dim3(256), lds, (hipStream_t)stream, gathered, world, rank_stride, n,
                     D, num_rows, table_grad);
  PV_LAUNCH_CHECK("sparse_merge_kernel");
  return PV_OK;
}

}  // extern "C"
