// common.hpp — shared host/device helpers for libmhimx (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>

#include "../../include/mhimx.h"

namespace mhimx {

// ---------------------------------------------------------------- host: errors
void set_error(const std::string& s);
int fail(int code, const char* fmt, ...);
const std::string& last_error();

#define MHIMX_CHECK_ARG(cond, ...)                      \
  do {                                                  \
    if (!(cond)) return ::mhimx::fail(-1, __VA_ARGS__); \
  } while (0)

#define MHIMX_HIP(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return ::mhimx::fail((int)_e, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define MHIMX_LAUNCH_CHECK()                                                                  \
  do {                                                                                        \
    hipError_t _e = hipGetLastError();                                                        \
    if (_e != hipSuccess) return ::mhimx::fail((int)_e, "kernel launch: %s", hipGetErrorString(_e)); \
  } while (0)

// hipFuncSetAttribute is a per-DEVICE setting: the "done" flag of a call site is kept per device (a process may drive several
// GPUs), and the first call on each device must happen outside stream capture (the trainers run a warm-up step first).
struct DeviceOnce {
  bool done[64] = {};
  bool need(int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *dev_out = -1; return true; }
    *dev_out = dev;
    return !done[dev];
  }
};
#define MHIMX_ONCE_PER_DEVICE(...)                         \
  do {                                                     \
    static ::mhimx::DeviceOnce _once;                      \
    int _dev;                                              \
    if (_once.need(&_dev)) {                               \
      __VA_ARGS__;                                         \
      if (_dev >= 0) _once.done[_dev] = true;              \
    }                                                      \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }
// queue a final reduction on the caller's list (mhimx_reduce_flush runs it); false: no list / list full -> launch it now
inline bool defer_push(mhimx_reduce_list* l, const mhimx_reduce_job& j) {
  if (!l || l->n >= MHIMX_REDUCE_MAX) return false;
  l->j[l->n++] = j;
  return true;
}
inline mhimx_reduce_job reduce_job_parts(const float* parts, int64_t G, int64_t W, int64_t ld, float* out, int accumulate) {
  mhimx_reduce_job j = {};
  j.kind = 0; j.accumulate = accumulate; j.parts = parts; j.out = out; j.G = G; j.W = W; j.ld = ld;
  return j;
}
// A TALL set of slabs (>= 32 of them) with a contiguous output is summed as partial rows - 32 slabs at a time per column, the row-group form
// of reduce_parts_kernel - instead of one serial chain per element (thin_tn's 256 row chunks: 64 dependent rounds, 21 us for 1 024 sums).
// The rule lives HERE and in reduce_slabs_now, so that a sum has the same bits queued or not.
// (small outputs only: with 65 536 sums - the scorer's weight gradient, 64 slabs - the element-parallel form has enough threads and is faster)
inline bool reduce_slabs_as_parts(int64_t splits, int64_t K1, int64_t K2, int64_t ldo) { return splits >= 32 && ldo == K2 && K1 * K2 <= 16384; }
inline mhimx_reduce_job reduce_job_slabs(const float* ws, int64_t splits, int64_t K1, int64_t K2, int64_t ldo, float* out, int accumulate) {
  if (reduce_slabs_as_parts(splits, K1, K2, ldo)) return reduce_job_parts(ws, splits, K1 * K2, K1 * K2, out, accumulate);
  mhimx_reduce_job j = {};
  j.kind = 1; j.accumulate = accumulate; j.parts = ws; j.out = out; j.G = splits; j.K1 = K1; j.K2 = K2; j.ldo = ldo;
  return j;
}

// the two final-reduction launches when they are not queued (definitions: rows.hip, gemm.hip) — the one place each kernel is launched from
// other translation units
int reduce_parts_now(hipStream_t st, const float* part, int64_t G, int64_t W, int64_t ld, float* out, int accumulate);
int reduce_slabs_now(hipStream_t st, const float* ws, float* C, int64_t K1, int64_t K2, int64_t ldc, int splits, int accumulate,
                     int batch = 1, int64_t sC = 0);

// bump allocator over a caller-provided workspace (256-byte granules)
struct Arena {
  char* base;
  int64_t cap, off;
  Arena(void* p, int64_t bytes) : base((char*)p), cap(bytes), off(0) {}
  template <typename T>
  T* take(int64_t n) {
    int64_t b = align_up(n * (int64_t)sizeof(T), 256);
    T* r = (T*)(base + off);
    off += b;
    return r;
  }
  bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};

// ---------------------------------------------------------------- bag batching (round 6: mhimx_window_run, step.hip)
// The launches between the projection and the weight gradient of an accumulation window are issued ONCE for all its bags: gridDim.z = bags,
// blockIdx.z = the bag a workgroup works for.  The bags have the same shape and each owns a copy of the step's workspace at a fixed stride,
// so the host enqueues the window's middle exactly as it enqueues one bag's - with bag 0's pointers - and every kernel on that path starts
// by moving the pointers it was given to its own bag's copy: a pointer inside one of (up to three) address ranges moves by bag * stride of
// that range (the per-bag workspace), a pointer equal to the table key (bag 0's label) is replaced by the bag's table entry; any other
// pointer (parameters, prepared weight images, the dropout tick) is shared by the bags and stays.  Seeds: a bag's counter-hash streams differ from bag 0's by a per-bag additive constant (every seed the
// host derives is seed + constant).  A plain launch has n = 0 and gridDim.z = 1: blockIdx.z = 0 moves nothing.
constexpr int BAG_BATCH_MAX = MHIMX_WINDOW_MAX;
struct BagBatch {
  int32_t n, pad;
  uint64_t lo[3], span[3];
  int64_t stride[3];
  uint64_t dsel[BAG_BATCH_MAX], dmca[BAG_BATCH_MAX];       // seed of bag b = seed of bag 0 + d*[b] (the select's / Merge's streams)
  uint64_t tab_key, tab[BAG_BATCH_MAX];                    // a pointer EQUAL to tab_key (bag 0's label) stands for tab[b] in bag b's plane
};
// host: the batch the calling thread is enqueueing (none: n = 0), set by the window executor around the window's middle
const BagBatch& cur_batch();
void set_batch(const BagBatch* b);
inline dim3 bgrid(unsigned x, unsigned y = 1) { return dim3(x, y, cur_batch().n > 0 ? (unsigned)cur_batch().n : 1u); }

// ---------------------------------------------------------------- device helpers
#define MHIMX_DEV __device__ __forceinline__

template <typename T>
MHIMX_DEV T* bag_ptr(T* p, const BagBatch& bb) {
  const unsigned bag = blockIdx.z;
  if (bag == 0) return p;
  const uint64_t a = reinterpret_cast<uint64_t>(p);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (a - bb.lo[r] < bb.span[r]) return reinterpret_cast<T*>(a + (uint64_t)bag * (uint64_t)bb.stride[r]);
  if (a == bb.tab_key) return reinterpret_cast<T*>(bb.tab[bag]);
  return p;
}
#define MHIMX_BAG(p) p = ::mhimx::bag_ptr(p, bb)
MHIMX_DEV uint64_t bag_sel_seed(uint64_t s, const BagBatch& bb) { return blockIdx.z ? s + bb.dsel[blockIdx.z] : s; }
MHIMX_DEV uint64_t bag_mca_seed(uint64_t s, const BagBatch& bb) { return blockIdx.z ? s + bb.dmca[blockIdx.z] : s; }

// erf to fp32 rounding level (|error| <= 1.5e-7, Abramowitz & Stegun 7.1.26) in ~14 VALU operations instead of the ~40 of
// the library erff: the GELU epilogues are VALU-bound tails of their kernels.
MHIMX_DEV float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}

// tanh from one exp: (1 - e^{-2|x|}) / (1 + e^{-2|x|}), absolute error < 1e-7 (the library tanhf is ~40 branchy operations)
MHIMX_DEV float tanh_fast(float x) {
  const float t = __expf(-2.f * fabsf(x));
  return copysignf((1.f - t) * __frcp_rn(1.f + t), x);
}

MHIMX_DEV float act_fwd(float x, int act) {
  switch (act) {
    case MHIMX_ACT_RELU: return x > 0.f ? x : 0.f;
    case MHIMX_ACT_GELU: return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f));   // exact-erf GELU (erf to 1.5e-7)
    case MHIMX_ACT_TANH: return tanh_fast(x);
    default: return x;
  }
}

// derivative w.r.t. the pre-activation x (y = act(x) is passed for the cheap forms)
MHIMX_DEV float act_grad(float x, float y, int act) {
  switch (act) {
    case MHIMX_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case MHIMX_ACT_GELU: {
      const float phi = 0.39894228040143267794f * __expf(-0.5f * x * x);
      return 0.5f * (1.f + erf_fast(x * 0.70710678118654752440f)) + x * phi;
    }
    case MHIMX_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// y = act(x) and dy/dx in one evaluation (the erf is shared)
MHIMX_DEV void act_fwd_grad(float x, int act, float& y, float& g) {
  switch (act) {
    case MHIMX_ACT_RELU: y = x > 0.f ? x : 0.f; g = x > 0.f ? 1.f : 0.f; return;
    case MHIMX_ACT_GELU: {                                  // erf_fast inlined: its e^{-(x/sqrt2)^2} IS the Gaussian of phi(x)
      const float ax = fabsf(x) * 0.70710678118654752440f;
      const float ex = __expf(-ax * ax);
      const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
      float p = fmaf(1.061405429f, t, -1.453152027f);
      p = fmaf(p, t, 1.421413741f);
      p = fmaf(p, t, -0.284496736f);
      p = fmaf(p, t, 0.254829592f);
      const float cdf = 0.5f * (1.f + copysignf(1.f - p * t * ex, x));
      y = x * cdf;
      g = cdf + x * 0.39894228040143267794f * ex;
      return;
    }
    case MHIMX_ACT_TANH: y = tanh_fast(x); g = 1.f - y * y; return;
    default: y = x; g = 1.f; return;
  }
}

// Counter-based keep/drop decision: one 32-bit mix of (seed, row, col) per element.  The same
// (seed,row,col) gives the same bit in forward and backward, so no mask is ever stored.
MHIMX_DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Keyed pseudo-random PERMUTATION of 0 .. n-1, one element at a time: a 6-round Feistel network on the smallest even-bit domain
// 2^bits >= n (round function: one 32-bit mix of the half, the round and the key), cycle-walked into [0, n) (x = E(x) until x < n; fewer
// than 4 steps on average).  perm_bits(n) gives `bits`.
MHIMX_DEV uint64_t feistel_index(uint64_t j, uint64_t n, int bits, uint32_t k0, uint32_t k1) {
  const int h = bits >> 1;
  const uint32_t hm = (1u << h) - 1u;
  uint64_t x = j;
  do {
    uint32_t l = (uint32_t)(x >> h) & hm, r = (uint32_t)x & hm;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const uint32_t f = mix32(r * 0x9E3779B1u + (rd & 1 ? k1 : k0) + (uint32_t)rd * 0x7F4A7C15u) & hm;
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = ((uint64_t)l << h) | r;
  } while (x >= n);
  return x;
}
// The same for SMALL lists inside a kernel (n <= 2^16: the select's subset draws), tuned for a wave that waits for its slowest lane:
// the domain is the smallest power of two >= n with ANY bit count (halves of floor / ceil(bits / 2) bits: round r XORs one half with a
// function of the other, a bijection whatever the sizes) - fewer than 2 steps on average instead of up to 4, and the longest walk among
// the lanes of a wave halves with it; 32-bit arithmetic, two multiplies per round, 4 rounds.
MHIMX_DEV uint32_t feistel_small(uint32_t j, uint32_t n, int bits, uint32_t k0, uint32_t k1) {
  const int rb = bits >> 1, lb = bits - rb;
  const uint32_t rm = (1u << rb) - 1u, lm = (1u << lb) - 1u;
  uint32_t x = j;
  do {
    uint32_t l = x >> rb, r = x & rm;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      uint32_t f = ((rd & 1) ? l : r) * 0x9E3779B1u + ((rd & 1) ? k1 : k0) + (uint32_t)rd * 0x7F4A7C15u;
      f ^= f >> 15; f *= 0x846ca68bU; f ^= f >> 13;
      if (rd & 1) r ^= f & rm;
      else l ^= f & lm;
    }
    x = (l << rb) | r;
  } while (x >= n);
  return x;
}
MHIMX_DEV int small_perm_bits(uint32_t n) {
  int bits = 2;
  while ((1u << bits) < n) ++bits;
  return bits;
}
MHIMX_DEV int perm_bits(uint64_t n) {
  int bits = 2;
  while (((uint64_t)1 << bits) < n) bits += 2;
  return bits;
}
// two-level form: a per-(seed,row) key (two mixes, amortised over the columns a thread handles) and ONE mix per element
MHIMX_DEV uint32_t drop_row_key(uint64_t seed, uint64_t row) {
  const uint32_t r = mix32((uint32_t)row * 0x9E3779B1u ^ (uint32_t)seed);
  return mix32(r + (uint32_t)(seed >> 32) + (uint32_t)(row >> 32) * 0x85EBCA77u);
}
MHIMX_DEV bool drop_keep_k(uint32_t row_key, uint32_t col, float p) {
  const uint32_t h = mix32(row_key + col * 0x85EBCA77u);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;        // keep with probability 1-p
}
MHIMX_DEV bool drop_keep(uint64_t seed, uint64_t row, uint32_t col, float p) {
  return drop_keep_k(drop_row_key(seed, row), col, p);
}

// Wave64 all-reduce on the VALU's DPP network (no LDS round trips: __shfl_xor lowers to ds_bpermute, ~6 dependent
// LDS-crossbar hops per reduction; this is 6 DPP moves + one v_readlane).  DPP controls: quad_perm [1,0,3,2] = 0xB1,
// [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
MHIMX_DEV float dpp_mov(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// dropout seed of this launch: the by-value seed plus an optional device-resident step counter, so a captured
// hipGraph (kernel arguments frozen at capture) still draws a fresh mask on every replay
MHIMX_DEV uint64_t eff_seed(uint64_t seed, const uint64_t* tick) {
  return tick ? seed + tick[0] * 0x9E3779B97F4A7C15ull : seed;
}

MHIMX_DEV float wave_sum(float v) {
  v += dpp_mov<0xB1, 0xf>(0.f, v);
  v += dpp_mov<0x4E, 0xf>(0.f, v);
  v += dpp_mov<0x141, 0xf>(0.f, v);
  v += dpp_mov<0x140, 0xf>(0.f, v);          // every lane of a 16-lane row now holds its row sum
  v += dpp_mov<0x142, 0xa>(0.f, v);          // rows 1,3 += row 0,2
  v += dpp_mov<0x143, 0xc>(0.f, v);          // rows 2,3 += row 1 (= rows 0+1): lane 63 = total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
MHIMX_DEV float wave_max(float v) {
  const float ninf = -__builtin_inff();
  v = fmaxf(v, dpp_mov<0xB1, 0xf>(ninf, v));
  v = fmaxf(v, dpp_mov<0x4E, 0xf>(ninf, v));
  v = fmaxf(v, dpp_mov<0x141, 0xf>(ninf, v));
  v = fmaxf(v, dpp_mov<0x140, 0xf>(ninf, v));
  v = fmaxf(v, dpp_mov<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_mov<0x143, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace mhimx
