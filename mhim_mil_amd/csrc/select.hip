// select.hip — hard-instance selection without leaving the device (replaces masking.py:9-88 and its
// .tolist() / Python set() / torch.tensor(list) round trip).
//
// One workgroup (1024 threads = 16 waves) per selection; all state in LDS:
//   1. 4-pass 8-bit radix select on the order-preserving uint32 image of the fp32 score -> threshold T
//      and the number of T-valued elements that belong to the top-k (per-wave private histograms);
//   2. one ordered sweep: elements above T are appended (any order), T-valued elements are taken lowest
//      index first (block prefix count) -> exactly k 64-bit keys (value image << 32 | ~index);
//   3. bitonic sort of the keys in LDS, descending => candidates ordered by (value desc, index asc): the
//      build's tie contract (the reference's torch.topk order is implementation-defined);
//   4. the caller's permutation picks n_sel of them (masking.py:66-71); flags[] marks the masked ids
//      (unioned with an earlier mask if given, masking.py:74-75);
//   5. ordered stream compaction of the unflagged ids -> kept ids ascending, then the masked ids.
// Integer work only: bit-exact against the oracle by construction.
#include "common.hpp"

namespace mhimx {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_WAVES = SEL_THREADS / 64;

MHIMX_DEV uint32_t mono32(float f, bool largest) {
  uint32_t b = __float_as_uint(f);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending fp32 order -> ascending uint32 order
  return largest ? b : ~b;                          // "largest key" == smallest value when !largest
}

// exclusive prefix count of `pred` over the 1024 threads in thread order; returns total via *total
MHIMX_DEV uint32_t block_prefix(bool pred, uint32_t* wave_tot /*[16] LDS*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(pred);
  const uint32_t in_wave = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SEL_WAVES; ++w) {
    const uint32_t c = wave_tot[w];
    if (w < wave) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + in_wave;
}

template <bool VOTE>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(
    const float* __restrict__ score_all, int64_t N, int k, int n_sel, int largest, const int64_t* __restrict__ perm,
    const int64_t* __restrict__ other, int64_t n_other, int64_t* __restrict__ mask_ids,
    int64_t* __restrict__ len_keep_dev, int64_t* __restrict__ topk_out, uint8_t* __restrict__ flags,
    float* __restrict__ vote, int P /* pow2 >= k */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);               // [P]
  uint32_t* hist = reinterpret_cast<uint32_t*>(keys + P);               // [16][256]
  uint32_t* wave_tot = hist + SEL_WAVES * 256;                          // [16]
  uint32_t* misc = wave_tot + SEL_WAVES;                                // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* score = score_all + (VOTE ? (int64_t)blockIdx.x * N : 0);
  const bool lg = largest != 0;

  // ---- 1. radix select ------------------------------------------------------------------------
  uint32_t prefix = 0, remaining = (uint32_t)k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < SEL_WAVES * 256; i += SEL_THREADS) hist[i] = 0;
    __syncthreads();
    const uint32_t hmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int64_t i = tid; i < N; i += SEL_THREADS) {
      const uint32_t key = mono32(score[i], lg);
      if ((key & hmask) == prefix) atomicAdd(&hist[wave * 256 + ((key >> shift) & 255u)], 1u);
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t c = 0;
#pragma unroll
      for (int w = 0; w < SEL_WAVES; ++w) c += hist[w * 256 + tid];
      hist[tid] = c;                       // bins of wave 0 now hold the totals
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t above = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (above + hist[d] >= remaining) break;
        above += hist[d];
      }
      misc[0] = (uint32_t)d;
      misc[1] = remaining - above;
    }
    __syncthreads();
    prefix |= misc[0] << shift;
    remaining = misc[1];
    __syncthreads();
  }
  const uint32_t T = prefix;                 // k-th largest key; `remaining` T-valued elements are in the top-k

  // ---- 2. gather exactly k keys ------------------------------------------------------------------
  if (tid == 0) misc[2] = 0;                 // append cursor
  __syncthreads();
  uint32_t eq_base = 0;
  for (int64_t c0 = 0; c0 < N; c0 += SEL_THREADS) {
    const int64_t i = c0 + tid;
    uint32_t key = 0;
    bool gt = false, eq = false;
    if (i < N) {
      key = mono32(score[i], lg);
      gt = key > T;
      eq = key == T;
    }
    uint32_t tot;
    const uint32_t rank = block_prefix(eq, wave_tot, &tot);
    const bool take = gt || (eq && (eq_base + rank) < remaining);
    if (take) {
      const uint32_t pos = atomicAdd(&misc[2], 1u);
      keys[pos] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    }
    eq_base += tot;
  }
  __syncthreads();

  if (VOTE) {
    for (int j = tid; j < k; j += SEL_THREADS) {
      const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(keys[j] & 0xFFFFFFFFull);
      atomicAdd(&vote[idx], 1.0f);           // integer-valued: exact in any order
    }
    return;
  }

  // ---- 3. bitonic sort, descending ---------------------------------------------------------------
  for (int j = k + tid; j < P; j += SEL_THREADS) keys[j] = 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += SEL_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  if (topk_out)
    for (int j = tid; j < k; j += SEL_THREADS) topk_out[j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(keys[j] & 0xFFFFFFFFull));

  // ---- 4. flags ------------------------------------------------------------------------------------
  for (int64_t i = tid; i < N; i += SEL_THREADS) flags[i] = 0;
  __syncthreads();
  const bool has_other = other != nullptr && n_other > 0;
  const int64_t len_keep_simple = N - n_sel;
  for (int j = tid; j < n_sel; j += SEL_THREADS) {
    const int64_t src = perm ? perm[j] : (int64_t)j;
    const int64_t idx = (int64_t)(0xFFFFFFFFu - (uint32_t)(keys[src] & 0xFFFFFFFFull));
    flags[idx] = 1;
    if (!has_other) mask_ids[len_keep_simple + j] = idx;     // masked ids keep candidate-order o perm
  }
  if (has_other)
    for (int64_t j = tid; j < n_other; j += SEL_THREADS) flags[other[j]] = 1;
  __threadfence_block();
  __syncthreads();

  // ---- 5. ordered compaction ---------------------------------------------------------------------
  uint32_t kept_base = 0;
  for (int64_t c0 = 0; c0 < N; c0 += SEL_THREADS) {
    const int64_t i = c0 + tid;
    const bool keep = (i < N) && flags[i] == 0;
    uint32_t tot;
    const uint32_t rank = block_prefix(keep, wave_tot, &tot);
    if (keep) mask_ids[kept_base + rank] = i;
    kept_base += tot;
  }
  if (has_other) {
    uint32_t m_base = 0;                     // union: masked ids sorted ascending (torch.unique, masking.py:75)
    for (int64_t c0 = 0; c0 < N; c0 += SEL_THREADS) {
      const int64_t i = c0 + tid;
      const bool msk = (i < N) && flags[i] != 0;
      uint32_t tot;
      const uint32_t rank = block_prefix(msk, wave_tot, &tot);
      if (msk) mask_ids[kept_base + m_base + rank] = i;
      m_base += tot;
    }
  }
  if (tid == 0 && len_keep_dev) *len_keep_dev = (int64_t)kept_base;
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

static size_t select_smem(int P) { return (size_t)P * 8 + (SEL_WAVES * 256 + SEL_WAVES + 8) * 4; }

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_select_ws_bytes(int64_t N) { return align_up(N, 256); }

extern "C" int mhimx_select_mask(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                                 const int64_t* perm, const int64_t* other, int64_t n_other, int64_t* mask_ids,
                                 int64_t* len_keep_dev, int64_t* topk_sorted, void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(score && mask_ids && ws, "select_mask: null args");
  MHIMX_CHECK_ARG(N > 0 && N <= (1ll << 24), "select_mask: N out of range");
  MHIMX_CHECK_ARG(k >= 1 && k <= N && k <= 16384, "select_mask: k=%lld out of range (1..min(N,16384))", (long long)k);
  MHIMX_CHECK_ARG(n_sel >= 0 && n_sel <= k, "select_mask: n_sel out of range");
  MHIMX_CHECK_ARG(ws_bytes >= N, "select_mask: workspace too small");
  const int P = next_pow2((int)k < 2 ? 2 : (int)k);
  const size_t smem = select_smem(P);
  static bool attr_set = false;
  if (!attr_set) {
    MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384)));
    MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384)));
    attr_set = true;
  }
  hipLaunchKernelGGL(select_kernel<false>, dim3(1), dim3(SEL_THREADS), smem, (hipStream_t)stream, score, N, (int)k, (int)n_sel,
                     largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, (uint8_t*)ws, (float*)nullptr, P);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_vote_scores(void* stream, const float* attn, int64_t H, int64_t N, int64_t k, int32_t largest,
                                 float* vote, void* ws, int64_t ws_bytes) {
  (void)ws; (void)ws_bytes;
  MHIMX_CHECK_ARG(attn && vote && H > 0 && N > 0 && N <= (1ll << 24), "vote_scores: bad args");
  MHIMX_CHECK_ARG(k >= 1 && k <= N && k <= 16384, "vote_scores: k out of range");
  MHIMX_HIP(hipMemsetAsync(vote, 0, (size_t)N * 4, (hipStream_t)stream));
  const int P = next_pow2((int)k < 2 ? 2 : (int)k);
  static bool attr_set = false;
  if (!attr_set) {
    MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384)));
    attr_set = true;
  }
  hipLaunchKernelGGL(select_kernel<true>, dim3((unsigned)H), dim3(SEL_THREADS), select_smem(P), (hipStream_t)stream, attn, N, (int)k,
                     0, largest, (const int64_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, (int64_t*)nullptr,
                     (int64_t*)nullptr, (int64_t*)nullptr, (uint8_t*)nullptr, vote, P);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
