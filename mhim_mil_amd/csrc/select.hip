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

#ifdef MHIMX_SEL_PROF                                           // phase stamps of select_small_kernel: tools/exp_select.py
__device__ unsigned long long sel_prof[32];
#define SEL_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) sel_prof[i] = wall_clock64(); } while (0)
#else
#define SEL_STAMP(i)
#endif

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
  // passes start at the highest bit in which the keys differ (see radix_select_regs): a score in [0.5, 1) has 9 common
  // leading bits, and a histogram pass over a constant digit is N LDS atomics on one address
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (int64_t i = tid; i < N; i += SEL_THREADS) {
    const uint32_t key = mono32(score[i], lg);
    mn = key < mn ? key : mn;
    mx = key > mx ? key : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = (uint32_t)__shfl_xor((int)mn, o, 64), c = (uint32_t)__shfl_xor((int)mx, o, 64);
    mn = a < mn ? a : mn;
    mx = c > mx ? c : mx;
  }
  if (lane == 0) { hist[wave] = mn; hist[SEL_WAVES + wave] = mx; }
  __syncthreads();
  for (int w = 0; w < SEL_WAVES; ++w) {
    const uint32_t a = hist[w], c = hist[SEL_WAVES + w];
    mn = a < mn ? a : mn;
    mx = c > mx ? c : mx;
  }
  __syncthreads();
  uint32_t prefix = mx, remaining = (uint32_t)k;
  if (mn != mx) {
    const int hb = 31 - __clz(mn ^ mx);
    uint32_t fixed_mask = hb >= 31 ? 0u : ~((2u << hb) - 1u);
    prefix = mx & fixed_mask;
    int shift = hb - 7 < 0 ? 0 : hb - 7;
    while (true) {
      for (int i = tid; i < SEL_WAVES * 256; i += SEL_THREADS) hist[i] = 0;
      __syncthreads();
      for (int64_t i = tid; i < N; i += SEL_THREADS) {
        const uint32_t key = mono32(score[i], lg);
        if ((key & fixed_mask) == prefix) atomicAdd(&hist[wave * 256 + ((key >> shift) & 255u)], 1u);
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
      fixed_mask |= 0xFFu << shift;
      __syncthreads();
      if (shift == 0) break;
      shift = shift - 8 < 0 ? 0 : shift - 8;
    }
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


// ------------------------------------------------------------------------------------------------
// Fast path for N <= 16384 (a bag of one slide at 20x is ~1e4 patches): every thread keeps its <= 16 keys
// in registers (scores are read from HBM exactly once), flags are an LDS bitmap, the k candidates are ordered
// by rank counting (k^2/1024 LDS compares per thread, no barriers) when k <= 2048.
// ------------------------------------------------------------------------------------------------
// Wave64 integer scans / reductions on the DPP network (a __shfl_* is a ds_bpermute: an LDS-crossbar round trip per step,
// and this kernel is one long dependent chain).  row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
MHIMX_DEV uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
MHIMX_DEV uint32_t wave_scan_incl(uint32_t v) {       // inclusive prefix sum in lane order
  v += dpp_u32<0x111, 0xf>(0u, v);
  v += dpp_u32<0x112, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);                    // inclusive inside each 16-lane row
  v += dpp_u32<0x142, 0xa>(0u, v);                    // rows 1,3 += last lane of rows 0,2
  v += dpp_u32<0x143, 0xc>(0u, v);                    // rows 2,3 += lane 31
  return v;
}
MHIMX_DEV uint32_t wave_min_u32(uint32_t v) {
  uint32_t t;
  t = dpp_u32<0xB1, 0xf>(v, v); v = t < v ? t : v;
  t = dpp_u32<0x4E, 0xf>(v, v); v = t < v ? t : v;
  t = dpp_u32<0x141, 0xf>(v, v); v = t < v ? t : v;
  t = dpp_u32<0x140, 0xf>(v, v); v = t < v ? t : v;
  t = dpp_u32<0x142, 0xa>(v, v); v = t < v ? t : v;
  t = dpp_u32<0x143, 0xc>(v, v); v = t < v ? t : v;
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
MHIMX_DEV uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

// sum of the per-wave totals below `wave` and of all 16 (four 16-byte LDS reads)
MHIMX_DEV void wave_tot_combine(const uint32_t* wave_tot, int wave, uint32_t* below, uint32_t* total) {
  uint32_t w[SEL_WAVES];
#pragma unroll
  for (int q = 0; q < SEL_WAVES / 4; ++q) {
    const uint4 v = reinterpret_cast<const uint4*>(wave_tot)[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  uint32_t b = 0, t = 0;
#pragma unroll
  for (int q = 0; q < SEL_WAVES; ++q) {
    b += q < wave ? w[q] : 0u;
    t += w[q];
  }
  *below = b;
  *total = t;
}

// exclusive prefix sum of one integer per thread over the 1024 threads (thread order); *total = block sum
MHIMX_DEV uint32_t block_scan_excl(uint32_t v, uint32_t* wave_tot /*[16] LDS, 16-byte aligned*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_scan_incl(v);
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base;
  wave_tot_combine(wave_tot, wave, &base, total);
  __syncthreads();
  return base + inc - v;
}

// radix select over register-resident keys: returns the k-th largest key T among the valid ones, in *remaining_out how
// many T-valued keys belong to the top-k and in *n_eq_out how many T-valued keys exist (remaining == n_eq: every tie is
// taken and the caller needs no tie ranking).
//   * Scores are low-entropy in their leading bits (a softmax-derived score in [0.5, 1) has 9 identical leading bits) and
//     a histogram pass over such a digit is 64 lanes x 10 keys of LDS atomics on ONE address: the passes start at the
//     highest bit in which the block's keys differ (min ^ max).  FULL32 (hashed keys) skips that pre-pass.
//   * One workgroup on one CU is a latency chain (a block barrier ~0.15 us, a pass ~7 of them): digits are 11 bits
//     (4 histogram copies of 2048 bins, wave w -> copy w & 3), and as soon as the threshold bin holds <= SEL_LIST keys
//     they are appended to an LDS list and the threshold is found by rank counting among them - for 1e4 continuous
//     scores that is ONE histogram pass + one list step instead of three or four passes.
// hist: [4][2048] LDS (also the list), wave_tot: [16], misc: [>=8] LDS.
constexpr int SEL_DIGIT = 11, SEL_BINS = 1 << SEL_DIGIT, SEL_COPIES = 4, SEL_LIST = 1024;

template <int KPT, bool FULL32>
MHIMX_DEV uint32_t radix_select_regs(const uint32_t (&key)[KPT], const bool (&valid)[KPT], uint32_t k, uint32_t* hist,
                                     uint32_t* wave_tot, uint32_t* misc, uint32_t* remaining_out, uint32_t* n_eq_out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint32_t prefix = 0u, fixed_mask = 0u;
  int hb = 31;
  if (!FULL32) {
    // ---- block min / max of the valid keys
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (valid[j]) { mn = key[j] < mn ? key[j] : mn; mx = key[j] > mx ? key[j] : mx; }
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane == 0) { hist[wave] = mn; hist[SEL_WAVES + wave] = mx; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SEL_WAVES / 4; ++q) {
      const uint4 a = reinterpret_cast<const uint4*>(hist)[q], c = reinterpret_cast<const uint4*>(hist + SEL_WAVES)[q];
      mn = min(min(mn, min(a.x, a.y)), min(a.z, a.w));
      mx = max(max(mx, max(c.x, c.y)), max(c.z, c.w));
    }
    __syncthreads();
    const uint32_t diff = mn ^ mx;
    if (diff == 0u) {                          // every key equal: all of them are T-valued
      uint32_t nv = 0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) nv += valid[j] ? 1u : 0u;
      uint32_t tot;
      block_scan_excl(nv, wave_tot, &tot);
      *remaining_out = k;
      *n_eq_out = tot;
      return mx;
    }
    hb = 31 - __clz(diff);                     // highest differing bit
    fixed_mask = hb >= 31 ? 0u : ~((2u << hb) - 1u);               // the common leading bits are decided
    prefix = mx & fixed_mask;
  }
  uint32_t remaining = k, n_eq = 0;
  int shift = hb - (SEL_DIGIT - 1) < 0 ? 0 : hb - (SEL_DIGIT - 1);
  uint32_t* copy = hist + (wave & (SEL_COPIES - 1)) * SEL_BINS;
  while (true) {
    {
      uint4* h4 = reinterpret_cast<uint4*>(hist);
      for (int i = tid; i < SEL_COPIES * SEL_BINS / 4; i += SEL_THREADS) h4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (valid[j] && (key[j] & fixed_mask) == prefix) atomicAdd(&copy[(key[j] >> shift) & (SEL_BINS - 1)], 1u);
    __syncthreads();
    // thread t owns bins 2t, 2t+1; population ABOVE them = reversed block scan of the per-thread sums
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int c = 0; c < SEL_COPIES; ++c) {
      const uint2 v = reinterpret_cast<const uint2*>(hist + c * SEL_BINS)[tid];
      c0 += v.x;
      c1 += v.y;
    }
    const uint32_t pre = wave_scan_incl(c0 + c1);                       // inclusive prefix in bin order
    if (lane == 63) wave_tot[wave] = pre;
    __syncthreads();
    uint32_t below, total;
    wave_tot_combine(wave_tot, wave, &below, &total);
    const uint32_t above = total - (below + pre);                       // population in the bins above 2t+1
    if (above < remaining && remaining <= above + c1) { misc[0] = 2u * tid + 1u; misc[1] = remaining - above; misc[3] = c1; }
    else if (above + c1 < remaining && remaining <= above + c1 + c0) { misc[0] = 2u * tid; misc[1] = remaining - above - c1; misc[3] = c0; }
    __syncthreads();
    prefix |= misc[0] << shift;
    remaining = misc[1];
    n_eq = misc[3];                            // keys in the threshold bin
    fixed_mask |= (uint32_t)(SEL_BINS - 1) << shift;
    if (shift == 0) break;                     // all bits decided: the bin IS the value T
    if (n_eq <= (uint32_t)SEL_LIST) {
      // ---- finish among the bin's keys: T = the remaining-th largest of the list
      if (tid == 0) misc[4] = 0;
      __syncthreads();                         // (also: everyone has read misc[0..3] and the histogram)
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if (valid[j] && (key[j] & fixed_mask) == prefix) hist[atomicAdd(&misc[4], 1u)] = key[j];
      __syncthreads();
      for (uint32_t j = tid; j < n_eq; j += SEL_THREADS) {
        const uint32_t mine = hist[j];
        uint32_t g = 0, e = 0;
        for (uint32_t q = 0; q < n_eq; ++q) {
          const uint32_t o = hist[q];
          g += o > mine ? 1u : 0u;
          e += o == mine ? 1u : 0u;
        }
        if (g < remaining && remaining <= g + e) { misc[5] = mine; misc[6] = remaining - g; misc[7] = e; }   // duplicates agree
      }
      __syncthreads();
      prefix = misc[5];
      remaining = misc[6];
      n_eq = misc[7];
      break;
    }
    __syncthreads();                           // misc / histogram are rewritten by the next pass
    shift = shift - SEL_DIGIT < 0 ? 0 : shift - SEL_DIGIT;
  }
  __syncthreads();
  *remaining_out = remaining;
  *n_eq_out = n_eq;
  return prefix;
}

// Thread t owns the CONTIGUOUS instances [t*KPT, (t+1)*KPT): every order-dependent step (ties lowest index first,
// ascending compaction) then costs ONE block scan of per-thread counts instead of one per 1024-instance chunk.
// LEAN: the production call (mhimx_select_rows without a mask-id list: device-drawn subsets, no injected permutation, no earlier
// mask, no top-k list) - the branches of every other form are compiled out.  The kernel runs ONCE per step on one workgroup: its
// 4 151 instructions (~25 KB) are fetched cold by 16 waves; the lean instantiation has 3 452.
template <int KPT, bool LEAN = false>
__global__ __launch_bounds__(SEL_THREADS) void select_small_kernel(
    const float* __restrict__ score, int N, int k, int n_sel, int largest, const int64_t* __restrict__ perm,
    const int64_t* __restrict__ other, int64_t n_other, int64_t* __restrict__ mask_ids, int64_t* __restrict__ len_keep_dev,
    int64_t* __restrict__ topk_out, int P, int use_rand, uint64_t rand_seed0, const uint64_t* __restrict__ tick, int merge_R,
    int64_t* __restrict__ rows_out,
    int64_t* __restrict__ rows_img /* optional (round 6): the same rows in the order of the projection's dPRE image when its producers write it -
                                      [rows that stay | 0 ... | rows to merge from position img_off | 0 ... to a multiple of 32] */,
    int img_off, BagBatch bb /* common.hpp: one workgroup per bag of an accumulation window */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (blockIdx.z) {
    MHIMX_BAG(score); MHIMX_BAG(perm); MHIMX_BAG(other); MHIMX_BAG(mask_ids); MHIMX_BAG(len_keep_dev); MHIMX_BAG(topk_out); MHIMX_BAG(rows_out);
    MHIMX_BAG(rows_img);
    rand_seed0 = bag_sel_seed(rand_seed0, bb);
  }
  const uint64_t rand_seed = eff_seed(rand_seed0, tick);
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);               // [P] gathered, [P] sorted / random keys
  uint64_t* sorted = keys + P;
  uint32_t* hist = reinterpret_cast<uint32_t*>(sorted + P);             // [4][2048]
  uint32_t* wave_tot = hist + SEL_COPIES * SEL_BINS;                    // [16]
  uint32_t* misc = wave_tot + SEL_WAVES;                                // [8]
  uint32_t* bitmap = misc + 8;                                          // [512] = 16384 bits
  uint32_t* rank = bitmap + 512;                                        // [P]
  uint16_t* klist = reinterpret_cast<uint16_t*>(rank + P);              // [N] kept rows in ascending order (the Merge draw indexes it)
  uint16_t* stage = reinterpret_cast<uint16_t*>(smem_raw);              // [N] row list staging (aliases keys / sorted / hist: last phase)
  const int tid = threadIdx.x;
  const bool lg = largest != 0;
  const int i0 = tid * KPT;

  SEL_STAMP(8);
  uint32_t key[KPT];
  bool valid[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    valid[j] = (i0 + j) < N;
    key[j] = valid[j] ? mono32(score[i0 + j], lg) : 0u;
  }
  SEL_STAMP(0);
  for (int i = tid; i < 512; i += SEL_THREADS) bitmap[i] = 0;
  if (tid == 0) misc[2] = 0;

  // ---- 1. threshold
  uint32_t remaining, n_eq;
  const uint32_t T = radix_select_regs<KPT, false>(key, valid, (uint32_t)k, hist, wave_tot, misc, &remaining, &n_eq);

  SEL_STAMP(1);
  // ---- 2. gather exactly k keys (ties: lowest index first)
  uint32_t eq_rank = 0;
  if (remaining != n_eq) {                 // only some of the T-valued keys belong to the top-k: rank them by index
    uint32_t neq = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) neq += (valid[j] && key[j] == T) ? 1u : 0u;
    uint32_t tot;
    eq_rank = block_scan_excl(neq, wave_tot, &tot);
  }
  // (one LDS atomic per WAVE reserves the wave's slots: k same-address returning atomics would serialise)
  bool take[KPT];
  uint32_t ntake = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    bool t = valid[j] && key[j] > T;
    if (valid[j] && key[j] == T) { t = remaining == n_eq || eq_rank < remaining; ++eq_rank; }
    take[j] = t;
    ntake += t ? 1u : 0u;
  }
  {
    // positions in THREAD order (one block scan): the device-drawn subset below picks candidates by their position in this list,
    // so the list must not depend on which wave reserves its slots first
    uint32_t tot;
    uint32_t pos = block_scan_excl(ntake, wave_tot, &tot);
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (take[j]) keys[pos++] = ((uint64_t)key[j] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(i0 + j));
  }
  __syncthreads();

  SEL_STAMP(2);
  // ---- 3. order the candidates: (value desc, index asc) == key desc
  // (the device-drawn random subset below depends on the candidate SET only: no ordering needed unless it is returned)
  const bool dev_rand = LEAN || (use_rand && !perm && n_sel < k);
  // The device-drawn subset ALWAYS picks by position in `keys` - the candidates in ascending row index (thread order) - whether or not the
  // value-ordered list is wanted as well: the same (seed, tick) masks the same rows with and without topk_out / LEAN.
  const uint64_t* cand = keys;
  if (LEAN || (dev_rand && !topk_out)) {
    // (no value-ordered list needed)
  } else if (k <= 2048) {
    for (int j = tid; j < k; j += SEL_THREADS) {
      const uint64_t mine = keys[j];
      int r = 0;
      for (int q = 0; q < k; ++q) r += keys[q] > mine ? 1 : 0;
      sorted[r] = mine;
    }
    __syncthreads();
  } else {
    for (int j = tid; j < P; j += SEL_THREADS) sorted[j] = j < k ? keys[j] : 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (P >> 1); t += SEL_THREADS) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const uint64_t a = sorted[lo], b = sorted[hi];
          if ((a < b) == desc) { sorted[lo] = b; sorted[hi] = a; }
        }
        __syncthreads();
      }
  }
  if (!LEAN && topk_out)
    for (int j = tid; j < k; j += SEL_THREADS) topk_out[j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(sorted[j] & 0xFFFFFFFFull));

  SEL_STAMP(3);
  // ---- 4. flags (LDS bitmap)
  const bool has_other = !LEAN && other != nullptr && n_other > 0;
  const int len_keep_simple = N - n_sel;
  if (dev_rand) {
    // masking.py:66-71 keeps a uniformly random n_sel-subset of the k candidates.  Drawn here without a host permutation and without
    // ranking random keys (k^2 / 1024 compares per thread were 3 us of this kernel): candidate pi(i), i < n_sel, of a keyed
    // pseudo-random permutation pi of the k list positions (common.hpp: feistel_small) - one short dependent chain per pick.
    // (both round keys depend on the WHOLE seed: with k1 a function of the high word alone, seeds that differ in the low word only - small
    // integers - shared it, and the 4-round network on a 6-bit domain masked one of 56 candidates 6.5 sigma off its share over 2000 seeds;
    // tools/feistel_sim.py, tests/test_round4_gpu.py::test_select_rows_draws_are_fair_on_small_lists)
    const uint32_t k0 = mix32((uint32_t)rand_seed ^ 0x9E3779B9u), k1 = mix32((uint32_t)(rand_seed >> 32) + 0x85EBCA6Bu + k0 * 0x632BE5ABu);
    const int bits = small_perm_bits((uint32_t)k);
    for (int i = tid; i < n_sel; i += SEL_THREADS) {
      const uint32_t j = feistel_small((uint32_t)i, (uint32_t)k, bits, k0, k1);
      const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(cand[j] & 0xFFFFFFFFull);
      atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
      if (!LEAN && !has_other && mask_ids) mask_ids[len_keep_simple + i] = (int64_t)idx;
    }
  } else if (!LEAN) {
    for (int j = tid; j < n_sel; j += SEL_THREADS) {
      const int64_t src = perm ? perm[j] : (int64_t)j;
      const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(sorted[src] & 0xFFFFFFFFull);
      atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
      if (!has_other && mask_ids) mask_ids[len_keep_simple + j] = (int64_t)idx;
    }
  }
  if (has_other)
    for (int64_t j = tid; j < n_other; j += SEL_THREADS) {
      const uint32_t idx = (uint32_t)other[j];
      atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
    }
  __syncthreads();

  SEL_STAMP(4);
  // ---- 5. ordered compaction: kept ids ascending (and, for a union with an earlier mask, masked ids ascending)
  bool kv[KPT];
  uint32_t nkeep = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = i0 + j;
    kv[j] = valid[j] && !((bitmap[i >> 5] >> (i & 31)) & 1u);
    nkeep += kv[j] ? 1u : 0u;
  }
  uint32_t kept_total;
  uint32_t kpos = block_scan_excl(nkeep, wave_tot, &kept_total);
  const uint32_t kpos0 = kpos;
  if (!LEAN && mask_ids) {
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (kv[j]) mask_ids[kpos++] = i0 + j;
    if (has_other) {
      uint32_t nm = 0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) nm += (valid[j] && !kv[j]) ? 1u : 0u;
      uint32_t mt;
      uint32_t mpos = kept_total + block_scan_excl(nm, wave_tot, &mt);
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if (valid[j] && !kv[j]) mask_ids[mpos++] = i0 + j;
    }
  }
  if (!LEAN && tid == 0 && len_keep_dev) *len_keep_dev = (int64_t)kept_total;
  SEL_STAMP(5);
  if (!LEAN && !rows_out) return;

  // ---- 6. Merge.masking (merge.py:158-176): a uniformly random R-subset of the kept rows is merged away.  Same device, the same
  // draw as step 4: the kept rows go to an LDS list in ascending order, row klist[pi2(i)], i < R, of a second keyed permutation (of the
  // Lrows list positions) is merged (flags in the bitmap, which step 5 has finished reading), then ordered compactions ->
  // rows_out = [kept rows that stay (ascending) | rows to merge (ascending)].  (A random key per kept row + a radix select of the R
  // largest was 4.1 us of this kernel.)  The pool is order independent, so the reference's random ORDER of the kept rows is not
  // reproduced (only fp summation order differs).
  const int Lrows = (int)kept_total;
  const int Lk = Lrows - merge_R;
  const bool partial = merge_R > 0 && merge_R < Lrows;
  if (partial) {
    uint32_t kp = kpos0;
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      if (kv[j]) klist[kp++] = (uint16_t)(i0 + j);
    for (int i = tid; i < 512; i += SEL_THREADS) bitmap[i] = 0;
    __syncthreads();
    const uint32_t q1 = mix32((uint32_t)rand_seed * 0x9E3779B1u + 0xC2B2AE35u), q0 = mix32(((uint32_t)(rand_seed >> 17) ^ 0x85EBCA6Bu) + q1 * 0x632BE5ABu);
    const int bits2 = small_perm_bits((uint32_t)Lrows);
    for (int i = tid; i < merge_R; i += SEL_THREADS) {
      const uint32_t row = klist[feistel_small((uint32_t)i, (uint32_t)Lrows, bits2, q0, q1)];
      atomicOr(&bitmap[row >> 5], 1u << (row & 31));
    }
    __syncthreads();
  }
  SEL_STAMP(6);
  bool mrg[KPT];
  uint32_t nstay = 0, nmrg = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = i0 + j;
    const bool m = kv[j] && (merge_R >= Lrows || (partial && ((bitmap[i >> 5] >> (i & 31)) & 1u)));
    mrg[j] = m;
    nmrg += m ? 1u : 0u;
    nstay += (kv[j] && !m) ? 1u : 0u;
  }
  // one scan for both lists (each total <= 16384 fits 16 bits).
  // use_rand == 2: rows to merge FIRST ([merge | stay]: lets the caller keep [stay rows | merged tokens] contiguous)
  uint32_t t2;
  const uint32_t packed = block_scan_excl(nstay | (nmrg << 16), wave_tot, &t2);
  uint32_t spos = (use_rand == 2 ? (uint32_t)(Lrows - Lk) : 0u) + (packed & 0xFFFFu);
  uint32_t mpos2 = (use_rand == 2 ? 0u : (uint32_t)Lk) + (packed >> 16);
  SEL_STAMP(10);
  // staged in LDS so the 8-byte row ids leave as full coalesced lines (a thread's own rows are 80 B apart from its
  // neighbour's: 10 scattered store instructions per wave otherwise)
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if (!kv[j]) continue;
    if (mrg[j]) stage[mpos2++] = (uint16_t)(i0 + j);
    else stage[spos++] = (uint16_t)(i0 + j);
  }
  __syncthreads();
  for (int i = tid; i < Lrows; i += SEL_THREADS) rows_out[i] = (int64_t)stage[i];
  if (rows_img) {
    const int mrg_n = Lrows - Lk, stay0 = use_rand == 2 ? mrg_n : 0, mrg0 = use_rand == 2 ? 0 : Lk;
    const int end = (img_off + mrg_n + 31) / 32 * 32;
    for (int p = tid; p < end; p += SEL_THREADS)
      rows_img[p] = p < Lk ? (int64_t)stage[stay0 + p] : ((p >= img_off && p - img_off < mrg_n) ? (int64_t)stage[mrg0 + p - img_off] : (int64_t)0);
  }
  SEL_STAMP(7);
}


// ------------------------------------------------------------------------------------------------
// Large bags (N > 16384: the c5 score vector has 200 000 entries and every rank of a sharded bag runs the same select): the
// same five steps spread over the chip.  One workgroup walking 200 000 keys several times is 0.6 ms; here every pass over the keys is
// a launch of up to 256 workgroups, the candidates' order included.
//   hist x3   three 11/11/10-bit digit histograms of the 32-bit key (per-workgroup LDS histogram, then global atomics); pass p
//             derives the digits fixed so far from the earlier histograms itself (every workgroup scans 2048 bins: no state kernel)
//   count     T = k-th largest key, `remaining` = how many T-valued keys belong to the top-k; per-workgroup counts of keys > T and == T
//   gather    exactly k 64-bit keys (value << 32 | ~index) at positions fixed by the counts: ties lowest index first
//   rank      place of every candidate in the (value desc, index asc) order by counting: k^2 compares over (k/1024) x (k/2048) workgroups
//   finish    top-k list, flags of the n_sel candidates the permutation picks (+ an earlier mask), the masked-id list
//   keep x2   per-workgroup counts of unflagged ids, then the ordered compaction (kept ids ascending, then - for a union - masked ids)
// Same outputs, same tie contract, bit for bit, as select_kernel<false>.
// ------------------------------------------------------------------------------------------------
constexpr int SELM_BINS = 2048, SELM_MAXG = 256;
// zero fill as a kernel of the same stream (a memset node of a captured graph is not ordered like a kernel node on every ROCm build)
__global__ void sel_zero_kernel(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static int sel_zero(hipStream_t st, void* p, int64_t bytes) {           // bytes % 16 == 0, p 16-byte aligned
  const int64_t n16 = bytes / 16;
  int64_t blocks = cdiv(n16, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sel_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p, n16);
  return 0;
}
struct SelMultiWs {
  uint32_t* hist;        // [3][2048], zeroed by the host's memset
  uint32_t* rank;        // [16384] rank of candidate j among the candidates (zeroed)
  uint32_t* selpos;      // [16384] selpos[r] = 1 + the place of sorted candidate r in the masked list, 0 = not selected (zeroed)
  uint32_t* blk_gt;      // [G]
  uint32_t* blk_eq;      // [G]
  uint32_t* blk_keep;    // [G]
  uint32_t* blk_mask;    // [G]
  uint32_t* state;       // [4]: T, remaining
  uint64_t* cand;        // [k]
  uint8_t* flags;        // [N], zeroed by the host's memset
};

// thread t owns bins 2t, 2t+1 of a 2048-bin histogram: find the bin in which the cumulative count FROM THE TOP reaches `remaining`
MHIMX_DEV void selm_find_bin(const uint32_t* __restrict__ h, uint32_t remaining, uint32_t* wave_tot, uint32_t* misc, uint32_t* bin_out,
                             uint32_t* rem_out) {
  const int tid = threadIdx.x;
  const uint2 v = reinterpret_cast<const uint2*>(h)[tid];
  uint32_t total;
  const uint32_t below = block_scan_excl(v.x + v.y, wave_tot, &total);
  const uint32_t above = total - (below + v.x + v.y);                 // population in the bins above 2t+1
  if (above < remaining && remaining <= above + v.y) { misc[0] = 2u * tid + 1u; misc[1] = remaining - above; }
  else if (above + v.y < remaining && remaining <= above + v.y + v.x) { misc[0] = 2u * tid; misc[1] = remaining - above - v.y; }
  __syncthreads();
  *bin_out = misc[0];
  *rem_out = misc[1];
  __syncthreads();
}

// digits fixed by the first `passes` histograms -> (prefix, remaining)
MHIMX_DEV void selm_prefix(const uint32_t* __restrict__ hist, int passes, uint32_t k, uint32_t* wave_tot, uint32_t* misc, uint32_t* prefix_out,
                           uint32_t* rem_out) {
  uint32_t prefix = 0, remaining = k, bin;
  if (passes >= 1) { selm_find_bin(hist, remaining, wave_tot, misc, &bin, &remaining); prefix = bin << 21; }
  if (passes >= 2) { selm_find_bin(hist + SELM_BINS, remaining, wave_tot, misc, &bin, &remaining); prefix |= bin << 10; }
  if (passes >= 3) { selm_find_bin(hist + 2 * SELM_BINS, remaining, wave_tot, misc, &bin, &remaining); prefix |= bin; }
  *prefix_out = prefix;
  *rem_out = remaining;
}

template <int PASS>
__global__ __launch_bounds__(SEL_THREADS) void selm_hist_kernel(const float* __restrict__ score, int64_t N, int largest, int k, SelMultiWs w) {
  __shared__ uint32_t lh[SELM_BINS];
  __shared__ __attribute__((aligned(16))) uint32_t wave_tot[SEL_WAVES];
  __shared__ uint32_t misc[8];
  const int tid = threadIdx.x;
  const bool lg = largest != 0;
  uint32_t prefix = 0, rem = 0;
  selm_prefix(w.hist, PASS, (uint32_t)k, wave_tot, misc, &prefix, &rem);
  const uint32_t fixed_mask = PASS == 0 ? 0u : (PASS == 1 ? 0xFFE00000u : 0xFFFFFC00u);
  const int shift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
  const uint32_t dmask = PASS == 2 ? 1023u : 2047u;
  for (int i = tid; i < SELM_BINS; i += SEL_THREADS) lh[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * SEL_THREADS + tid; i < N; i += (int64_t)gridDim.x * SEL_THREADS) {
    const uint32_t key = mono32(score[i], lg);
    if ((key & fixed_mask) == prefix) atomicAdd(&lh[(key >> shift) & dmask], 1u);
  }
  __syncthreads();
  uint32_t* gh = w.hist + PASS * SELM_BINS;
  for (int i = tid; i < SELM_BINS; i += SEL_THREADS)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// chunked walk: workgroup b owns instances [b*chunk, (b+1)*chunk)
__global__ __launch_bounds__(SEL_THREADS) void selm_count_kernel(const float* __restrict__ score, int64_t N, int largest, int k, int64_t chunk,
                                                                SelMultiWs w) {
  __shared__ __attribute__((aligned(16))) uint32_t wave_tot[SEL_WAVES];
  __shared__ uint32_t misc[8];
  const int tid = threadIdx.x;
  const bool lg = largest != 0;
  uint32_t T, remaining;
  selm_prefix(w.hist, 3, (uint32_t)k, wave_tot, misc, &T, &remaining);
  if (blockIdx.x == 0 && tid == 0) { w.state[0] = T; w.state[1] = remaining; }
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = b0 + chunk < N ? b0 + chunk : N;
  uint32_t gt = 0, eq = 0;
  for (int64_t i = b0 + tid; i < b1; i += SEL_THREADS) {
    const uint32_t key = mono32(score[i], lg);
    gt += key > T ? 1u : 0u;
    eq += key == T ? 1u : 0u;
  }
  uint32_t tg, te;
  block_scan_excl(gt, wave_tot, &tg);
  block_scan_excl(eq, wave_tot, &te);
  if (tid == 0) { w.blk_gt[blockIdx.x] = tg; w.blk_eq[blockIdx.x] = te; }
}

__global__ __launch_bounds__(SEL_THREADS) void selm_gather_kernel(const float* __restrict__ score, int64_t N, int largest, int64_t chunk,
                                                                 SelMultiWs w) {
  __shared__ __attribute__((aligned(16))) uint32_t wave_tot[SEL_WAVES];
  __shared__ uint32_t misc[8];
  const int tid = threadIdx.x, b = blockIdx.x;
  const bool lg = largest != 0;
  const uint32_t T = w.state[0], remaining = w.state[1];
  // positions: workgroup b' < b took gt[b'] + min(max(remaining - eq_before[b'], 0), eq[b']) candidates
  if (tid == 0) {
    uint32_t eqb = 0, pos = 0;
    for (int q = 0; q < b; ++q) {
      const uint32_t e = w.blk_eq[q];
      const uint32_t room = remaining > eqb ? remaining - eqb : 0u;
      pos += w.blk_gt[q] + (e < room ? e : room);
      eqb += e;
    }
    misc[0] = eqb;
    misc[1] = pos;
  }
  __syncthreads();
  uint32_t eq_base = misc[0], pos_base = misc[1];
  __syncthreads();
  const int64_t b0 = (int64_t)b * chunk, b1 = b0 + chunk < N ? b0 + chunk : N;
  for (int64_t c0 = b0; c0 < b1; c0 += SEL_THREADS) {
    const int64_t i = c0 + tid;
    uint32_t key = 0;
    bool gt = false, eq = false;
    if (i < b1) {
      key = mono32(score[i], lg);
      gt = key > T;
      eq = key == T;
    }
    uint32_t tot_eq, tot_take;
    const uint32_t erank = block_prefix(eq, wave_tot, &tot_eq);
    const bool take = gt || (eq && (eq_base + erank) < remaining);
    const uint32_t trank = block_prefix(take, wave_tot, &tot_take);
    if (take) w.cand[pos_base + trank] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    eq_base += tot_eq;
    pos_base += tot_take;
  }
}

// rank of every candidate by counting (the keys are distinct: value << 32 | ~index): workgroup (x, y) compares candidates
// [1024 x, 1024 x + 1024) against the SELM_SEG-key (256) segment y held in LDS (broadcast reads) - k^2 compares spread over the chip instead of a
// 105-step bitonic sort in one workgroup's LDS.  Integer atomics: the result does not depend on the order.
constexpr int SELM_SEG = 256;            // (2048: a c3 / c5 select ran its k^2 compares on 6 .. 70 workgroups of 16 waves, 42 .. 47 us)
__global__ __launch_bounds__(SEL_THREADS) void selm_rank_kernel(int k, int n_sel, const int64_t* __restrict__ perm, SelMultiWs w) {
  __shared__ uint64_t seg[SELM_SEG];
  const int tid = threadIdx.x;
  const int s0 = blockIdx.y * SELM_SEG;
  for (int i = tid; i < SELM_SEG; i += SEL_THREADS) seg[i] = (s0 + i) < k ? w.cand[s0 + i] : 0ull;       // (0 is below every key)
  if (blockIdx.y == 0) {                     // which sorted places the permutation picks (masking.py:66-71): a table for the finish launch
    for (int j = blockIdx.x * SEL_THREADS + tid; j < n_sel; j += gridDim.x * SEL_THREADS) w.selpos[perm ? perm[j] : (int64_t)j] = (uint32_t)j + 1u;
  }
  __syncthreads();
  const int j = blockIdx.x * SEL_THREADS + tid;
  if (j >= k) return;
  const uint64_t mine = w.cand[j];
  uint32_t r = 0;
#pragma unroll 8
  for (int q = 0; q < SELM_SEG; ++q) r += seg[q] > mine ? 1u : 0u;
  if (r) atomicAdd(&w.rank[j], r);
}

__global__ __launch_bounds__(SEL_THREADS) void selm_finish_kernel(int k, int n_sel, int64_t N, const int64_t* __restrict__ other, int64_t n_other,
                                                                 int64_t* __restrict__ mask_ids, int64_t* __restrict__ topk_out, SelMultiWs w) {
  const bool has_other = other != nullptr && n_other > 0;
  const int64_t len_keep_simple = N - n_sel;
  const int64_t t0 = (int64_t)blockIdx.x * SEL_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * SEL_THREADS;
  for (int64_t j = t0; j < k; j += stride) {
    const uint32_t r = w.rank[j];                             // place in the (value desc, index asc) order
    const int64_t idx = (int64_t)(0xFFFFFFFFu - (uint32_t)(w.cand[j] & 0xFFFFFFFFull));
    if (topk_out) topk_out[r] = idx;
    const uint32_t sp = w.selpos[r];
    if (sp) {
      w.flags[idx] = 1;
      if (!has_other) mask_ids[len_keep_simple + (sp - 1)] = idx;       // masked ids keep candidate-order o perm
    }
  }
  if (has_other)
    for (int64_t j = t0; j < n_other; j += stride) w.flags[other[j]] = 1;
}

__global__ __launch_bounds__(SEL_THREADS) void selm_keepcount_kernel(int64_t N, int64_t chunk, SelMultiWs w) {
  __shared__ __attribute__((aligned(16))) uint32_t wave_tot[SEL_WAVES];
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = b0 + chunk < N ? b0 + chunk : N;
  uint32_t keep = 0, n = 0;
  for (int64_t i = b0 + tid; i < b1; i += SEL_THREADS) { keep += w.flags[i] == 0 ? 1u : 0u; ++n; }
  uint32_t tk, tn;
  block_scan_excl(keep, wave_tot, &tk);
  block_scan_excl(n, wave_tot, &tn);
  if (tid == 0) { w.blk_keep[blockIdx.x] = tk; w.blk_mask[blockIdx.x] = tn - tk; }
}

__global__ __launch_bounds__(SEL_THREADS) void selm_compact_kernel(int64_t N, int64_t chunk, int has_other, int64_t* __restrict__ mask_ids,
                                                                  int64_t* __restrict__ len_keep_dev, SelMultiWs w) {
  __shared__ __attribute__((aligned(16))) uint32_t wave_tot[SEL_WAVES];
  __shared__ uint32_t misc[8];
  const int tid = threadIdx.x, b = blockIdx.x;
  if (tid == 0) {
    uint32_t kb = 0, mb = 0, kt = 0;
    for (int q = 0; q < (int)gridDim.x; ++q) {
      if (q < b) { kb += w.blk_keep[q]; mb += w.blk_mask[q]; }
      kt += w.blk_keep[q];
    }
    misc[0] = kb; misc[1] = mb; misc[2] = kt;
  }
  __syncthreads();
  uint32_t kept_base = misc[0], m_base = misc[1];
  const uint32_t kept_total = misc[2];
  __syncthreads();
  if (b == 0 && tid == 0 && len_keep_dev) *len_keep_dev = (int64_t)kept_total;
  const int64_t b0 = (int64_t)b * chunk, b1 = b0 + chunk < N ? b0 + chunk : N;
  for (int64_t c0 = b0; c0 < b1; c0 += SEL_THREADS) {
    const int64_t i = c0 + tid;
    const bool in = i < b1;
    const bool keep = in && w.flags[i] == 0;
    uint32_t tot;
    const uint32_t rank = block_prefix(keep, wave_tot, &tot);
    if (keep) mask_ids[kept_base + rank] = i;
    kept_base += tot;
    if (has_other) {                         // union: masked ids sorted ascending after the kept ones (torch.unique, masking.py:75)
      const bool msk = in && !keep;
      uint32_t mt;
      const uint32_t mr = block_prefix(msk, wave_tot, &mt);
      if (msk) mask_ids[kept_total + m_base + mr] = i;
      m_base += mt;
    }
  }
}

static size_t select_small_smem(int P) {          // keys+sorted, hist, wave_tot, misc, bitmap, rank, kept list (the 32 KB row staging aliases the head)
  return (size_t)2 * P * 8 + (size_t)(SEL_COPIES * SEL_BINS + SEL_WAVES + 8 + 512 + P) * 4 + 16384 * 2;
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

static size_t select_smem(int P) { return (size_t)P * 8 + (SEL_WAVES * 256 + SEL_WAVES + 8) * 4; }

}  // namespace mhimx

using namespace mhimx;

#ifdef MHIMX_SEL_PROF                                           // phase stamps of select_small_kernel: tools/exp_select.py
extern "C" int mhimx_sel_prof_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mhimx::sel_prof), 32 * 8);
}
#endif

// flags [N]; large bags also: 3 histograms, per-workgroup counts, the candidate list (select_multi)
static int64_t selm_extra_bytes() { return 3 * SELM_BINS * 4 + 2 * 16384 * 4 + 4 * SELM_MAXG * 4 + 256 + 16384 * 8; }
extern "C" int64_t mhimx_select_ws_bytes(int64_t N) { return align_up(N, 256) + (N > 16384 ? selm_extra_bytes() : 0); }

static int select_multi(hipStream_t st, const float* score, int64_t N, int k, int n_sel, int largest, const int64_t* perm, const int64_t* other,
                        int64_t n_other, int64_t* mask_ids, int64_t* len_keep_dev, int64_t* topk_sorted, void* ws, int P) {
  char* base = (char*)ws;
  SelMultiWs w;
  const int64_t nf = align_up(N, 256);
  w.flags = (uint8_t*)base;
  w.hist = (uint32_t*)(base + nf);
  w.rank = w.hist + 3 * SELM_BINS;
  w.selpos = w.rank + 16384;
  w.blk_gt = w.selpos + 16384;
  w.blk_eq = w.blk_gt + SELM_MAXG;
  w.blk_keep = w.blk_eq + SELM_MAXG;
  w.blk_mask = w.blk_keep + SELM_MAXG;
  w.state = w.blk_mask + SELM_MAXG;
  w.cand = (uint64_t*)((char*)w.state + 256);
  MHIMX_CHECK_ARG(aligned16(ws), "select_mask: workspace must be 16-byte aligned");
  sel_zero(st, base, nf + (3 * SELM_BINS + 2 * 16384) * 4);                 // flags, histograms, ranks, selection table
  int G = (int)cdiv(N, 4 * SEL_THREADS);
  if (G > SELM_MAXG) G = SELM_MAXG;
  if (G < 1) G = 1;
  const int64_t chunk = cdiv(N, G);
  hipLaunchKernelGGL(selm_hist_kernel<0>, dim3(G), dim3(SEL_THREADS), 0, st, score, N, largest, k, w);
  hipLaunchKernelGGL(selm_hist_kernel<1>, dim3(G), dim3(SEL_THREADS), 0, st, score, N, largest, k, w);
  hipLaunchKernelGGL(selm_hist_kernel<2>, dim3(G), dim3(SEL_THREADS), 0, st, score, N, largest, k, w);
  hipLaunchKernelGGL(selm_count_kernel, dim3(G), dim3(SEL_THREADS), 0, st, score, N, largest, k, chunk, w);
  hipLaunchKernelGGL(selm_gather_kernel, dim3(G), dim3(SEL_THREADS), 0, st, score, N, largest, chunk, w);
  (void)P;
  hipLaunchKernelGGL(selm_rank_kernel, dim3((unsigned)cdiv(k, SEL_THREADS), (unsigned)cdiv(k, SELM_SEG)), dim3(SEL_THREADS), 0, st, k, n_sel, perm, w);
  hipLaunchKernelGGL(selm_finish_kernel, dim3((unsigned)(cdiv(k, SEL_THREADS) < 16 ? 16 : cdiv(k, SEL_THREADS))), dim3(SEL_THREADS), 0, st, k, n_sel, N,
                     other, n_other, mask_ids, topk_sorted, w);
  hipLaunchKernelGGL(selm_keepcount_kernel, dim3(G), dim3(SEL_THREADS), 0, st, N, chunk, w);
  hipLaunchKernelGGL(selm_compact_kernel, dim3(G), dim3(SEL_THREADS), 0, st, N, chunk, (other != nullptr && n_other > 0) ? 1 : 0, mask_ids,
                     len_keep_dev, w);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

static int select_impl(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                       const int64_t* perm, const int64_t* other, int64_t n_other, int64_t* mask_ids,
                       int64_t* len_keep_dev, int64_t* topk_sorted, void* ws, int64_t ws_bytes, int g_use_rand,
                       uint64_t g_rand_seed, const uint64_t* g_tick, int g_merge_R, int64_t* g_rows_out, int64_t* g_rows_img = nullptr,
                       int g_img_off = 0) {
  MHIMX_CHECK_ARG(score && (mask_ids || g_rows_out) && ws, "select_mask: null args");
  MHIMX_CHECK_ARG(!g_rows_img || (g_rows_out && N <= 16384 && next_pow2((int)k < 2 ? 2 : (int)k) <= 4096),
                  "select_rows_img: the image-order list comes from the one-workgroup select (N <= 16384, k <= 4096)");
  MHIMX_CHECK_ARG(!g_rows_out || N <= 16384, "select_rows: fused row list needs N <= 16384 (use select_mask + compose_ids)");
  MHIMX_CHECK_ARG(N > 0 && N <= (1ll << 24), "select_mask: N out of range");
  MHIMX_CHECK_ARG(k >= 1 && k <= N && k <= 16384, "select_mask: k=%lld out of range (1..min(N,16384))", (long long)k);
  MHIMX_CHECK_ARG(n_sel >= 0 && n_sel <= k, "select_mask: n_sel out of range");
  MHIMX_CHECK_ARG(ws_bytes >= N, "select_mask: workspace too small");
  const int P = next_pow2((int)k < 2 ? 2 : (int)k);
  if (N <= 16384 && P <= 4096) {                       // registers + LDS fast path
    const size_t sm = select_small_smem(P);
#define MHIMX_SEL_SMALL(KPT)                                                                                                   \
    hipLaunchKernelGGL(select_small_kernel<KPT>, bgrid(1), dim3(SEL_THREADS), sm, (hipStream_t)stream, score, (int)N, (int)k,   \
                       (int)n_sel, largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, P, g_use_rand,          \
                       g_rand_seed, g_tick, g_merge_R, g_rows_out, g_rows_img, g_img_off, cur_batch())
    MHIMX_ONCE_PER_DEVICE(
        MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096)));
        MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096)));
        MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096))));
    const bool lean = g_use_rand && g_rows_out && !perm && !other && !mask_ids && !len_keep_dev && !topk_sorted && n_sel < k;
    if (lean) {
#define MHIMX_SEL_LEAN(KPT)                                                                                                    \
      hipLaunchKernelGGL((select_small_kernel<KPT, true>), bgrid(1), dim3(SEL_THREADS), sm, (hipStream_t)stream, score, (int)N, (int)k,   \
                         (int)n_sel, largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, P, g_use_rand,        \
                         g_rand_seed, g_tick, g_merge_R, g_rows_out, g_rows_img, g_img_off, cur_batch())
      MHIMX_ONCE_PER_DEVICE(
          MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096)));
          MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096)));
          MHIMX_HIP(hipFuncSetAttribute((const void*)select_small_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_small_smem(4096))));
      if (N <= 4096) MHIMX_SEL_LEAN(4);
      else if (N <= 10240) MHIMX_SEL_LEAN(10);
      else MHIMX_SEL_LEAN(16);
#undef MHIMX_SEL_LEAN
    } else if (N <= 4096) MHIMX_SEL_SMALL(4);
    else if (N <= 10240) MHIMX_SEL_SMALL(10);
    else MHIMX_SEL_SMALL(16);          // thread t owns instances [t*KPT, (t+1)*KPT)
#undef MHIMX_SEL_SMALL
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  MHIMX_CHECK_ARG(cur_batch().n == 0, "select: a bag-batched launch takes the one-workgroup select (N <= 16384, k <= 4096)");
  const size_t smem = select_smem(P);
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384))); MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384))));
  MHIMX_CHECK_ARG(!g_use_rand && !g_rows_out, "select: the random-subset forms need N <= 16384 and k <= 4096");
  if (N > 16384 && ws_bytes >= mhimx_select_ws_bytes(N) && mask_ids)
    return select_multi((hipStream_t)stream, score, N, (int)k, (int)n_sel, largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, ws, P);
  hipLaunchKernelGGL(select_kernel<false>, dim3(1), dim3(SEL_THREADS), smem, (hipStream_t)stream, score, N, (int)k, (int)n_sel,
                     largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, (uint8_t*)ws, (float*)nullptr, P);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_select_mask(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                                 const int64_t* perm, const int64_t* other, int64_t n_other, int64_t* mask_ids,
                                 int64_t* len_keep_dev, int64_t* topk_sorted, void* ws, int64_t ws_bytes) {
  return select_impl(stream, score, N, k, n_sel, largest, perm, other, n_other, mask_ids, len_keep_dev, topk_sorted, ws, ws_bytes,
                     0, 0, nullptr, 0, nullptr);
}

extern "C" int mhimx_select_rows(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                                 uint64_t rand_seed, const uint64_t* tick, int64_t merge_R, int64_t* rows_out, int64_t* mask_ids,
                                 void* ws, int64_t ws_bytes, int32_t merge_first) {
  MHIMX_CHECK_ARG(rows_out && merge_R >= 0 && merge_R <= N - n_sel, "select_rows: bad args");
  return select_impl(stream, score, N, k, n_sel, largest, nullptr, nullptr, 0, mask_ids, nullptr, nullptr, ws, ws_bytes,
                     merge_first ? 2 : 1, rand_seed, tick, (int)merge_R, rows_out);
}

extern "C" int mhimx_select_rows_img(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest, uint64_t rand_seed,
                                     const uint64_t* tick, int64_t merge_R, int64_t* rows_out, int64_t* rows_img, int64_t img_merge_off, void* ws,
                                     int64_t ws_bytes, int32_t merge_first) {
  MHIMX_CHECK_ARG(rows_out && rows_img && merge_R >= 0 && merge_R <= N - n_sel && img_merge_off >= N - n_sel - merge_R && img_merge_off % 32 == 0 &&
                      img_merge_off <= 32768,
                  "select_rows_img: bad args (the merged rows' image position is a multiple of 32 behind the rows that stay)");
  return select_impl(stream, score, N, k, n_sel, largest, nullptr, nullptr, 0, nullptr, nullptr, nullptr, ws, ws_bytes, merge_first ? 2 : 1, rand_seed, tick,
                     (int)merge_R, rows_out, rows_img, (int)img_merge_off);
}

extern "C" int mhimx_vote_scores(void* stream, const float* attn, int64_t H, int64_t N, int64_t k, int32_t largest,
                                 float* vote, void* ws, int64_t ws_bytes) {
  (void)ws; (void)ws_bytes;
  MHIMX_CHECK_ARG(attn && vote && H > 0 && N > 0 && N <= (1ll << 24), "vote_scores: bad args");
  MHIMX_CHECK_ARG(k >= 1 && k <= N && k <= 16384, "vote_scores: k out of range");
  {
    const int64_t nb = (N * 4 / 16) * 16;                                    // (the tail below 16 bytes: one more tiny launch)
    if (aligned16(vote) && nb) sel_zero((hipStream_t)stream, vote, nb);
    if (!aligned16(vote) || nb != N * 4) MHIMX_HIP(hipMemsetAsync((char*)vote + (aligned16(vote) ? nb : 0), 0, (size_t)(N * 4 - (aligned16(vote) ? nb : 0)), (hipStream_t)stream));
  }
  const int P = next_pow2((int)k < 2 ? 2 : (int)k);
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_smem(16384))));
  hipLaunchKernelGGL(select_kernel<true>, dim3((unsigned)H), dim3(SEL_THREADS), select_smem(P), (hipStream_t)stream, attn, N, (int)k,
                     0, largest, (const int64_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, (int64_t*)nullptr,
                     (int64_t*)nullptr, (int64_t*)nullptr, (uint8_t*)nullptr, vote, P);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
