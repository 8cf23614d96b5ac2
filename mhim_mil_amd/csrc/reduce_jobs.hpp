// reduce_jobs.hpp — the body of the deferred final reductions (mhimx_reduce_list: sums of per-block column partials, sums of split-K slabs)
// as a device function, so that the same arithmetic runs as the launch of its own (reduce_batch_kernel, rows.hip) and as trailing
// workgroups of the projection's weight-gradient launch (bag_wgrad_ws_kernel, wgrad.hip: round 4 - the reductions that are final before
// that launch starts ride in the CUs its last round of tiles leaves idle instead of taking a launch on the step's serial chain).
// Same summation order for any block size and any number of blocks: queued, ridden or launched at once, the bits are the same.
#pragma once
#include "common.hpp"

namespace mhimx {

struct ReduceTable { mhimx_reduce_job j[MHIMX_REDUCE_MAX]; int first[MHIMX_REDUCE_MAX + 1]; int n; };

// blocks a job wants at `threads` threads per block (every job loop is grid-stride: any count is correct)
inline int reduce_job_blocks(const mhimx_reduce_job& j, int threads) {
  int64_t nb = j.kind == 0 ? cdiv(j.W, 32) : cdiv(j.K1 * j.K2, threads);
  return (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));
}
inline int reduce_table_fill(ReduceTable& t, const mhimx_reduce_job* jobs, int n, int threads) {
  int first = 0;
  t.n = n;
  for (int i = 0; i < n; ++i) {
    t.j[i] = jobs[i];
    t.first[i] = first;
    first += reduce_job_blocks(jobs[i], threads);
  }
  t.first[n] = first;
  return first;
}

// block bx (of t.first[t.n]) of the table; red: [32][33] floats of LDS.  THREADS = 32 columns x THREADS / 32 row groups; a kind-0 sum always
// goes through 32 row-group partials (a thread of a smaller block owns several), so its bits do not depend on the block size.
template <int THREADS, bool BATCHED = false>
MHIMX_DEV void reduce_jobs_block(const ReduceTable& t, int bx, float (*red)[33], const BagBatch& bb /* BATCHED: a bag-batched launch, common.hpp */) {
  static_assert(THREADS % 32 == 0 && THREADS <= 1024 && 32 % (THREADS / 32) == 0, "32 columns x a divisor of 32 row groups");
  constexpr int NRG = THREADS / 32;
  int jb = 0;
  while (jb + 1 < t.n && bx >= t.first[jb + 1]) ++jb;
  const mhimx_reduce_job J = t.j[jb];
  // (a bag-batched launch: this bag's partials and outputs.  Local copies of the two pointers - writing into a copy of the table entry
  // sent the whole by-value table through scratch memory: 520 us for a 15 us launch)
  const float* __restrict__ parts = J.parts;
  float* __restrict__ outp = J.out;
  if constexpr (BATCHED) {
    if (blockIdx.z) { parts = bag_ptr(parts, bb); outp = bag_ptr(outp, bb); }
  }
  const int blk = bx - t.first[jb], nblk = t.first[jb + 1] - t.first[jb];
  if (J.kind == 0) {
    const int c = threadIdx.x & 31, rg0 = threadIdx.x >> 5;
    for (int64_t j0 = (int64_t)blk * 32; j0 < J.W; j0 += (int64_t)nblk * 32) {
      const int64_t j = j0 + c;
#pragma unroll
      for (int rg = rg0; rg < 32; rg += NRG) {
        float acc = 0.f;
        if (j < J.W) {
#pragma unroll 8
          for (int64_t b = rg; b < J.G; b += 32) acc += parts[b * J.ld + j];
        }
        red[rg][c] = acc;
      }
      __syncthreads();
      if (rg0 == 0 && j < J.W) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) v += red[q][c];
        outp[j] = J.accumulate ? outp[j] + v : v;
      }
      __syncthreads();
    }
  } else {
    const int64_t n = J.K1 * J.K2;
    for (int64_t idx = (int64_t)blk * THREADS + threadIdx.x; idx < n; idx += (int64_t)nblk * THREADS) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int64_t z = 0;
      for (; z + 8 <= J.G; z += 8) {                   // (eight slabs in flight; the same sums in the same order as four at a time)
        const float p0 = parts[(z + 0) * n + idx], p1 = parts[(z + 1) * n + idx], p2 = parts[(z + 2) * n + idx],
                    p3 = parts[(z + 3) * n + idx], p4 = parts[(z + 4) * n + idx], p5 = parts[(z + 5) * n + idx],
                    p6 = parts[(z + 6) * n + idx], p7 = parts[(z + 7) * n + idx];
        s0 += p0; s1 += p1; s2 += p2; s3 += p3;
        s0 += p4; s1 += p5; s2 += p6; s3 += p7;
      }
      for (; z + 4 <= J.G; z += 4) {
        s0 += parts[(z + 0) * n + idx];
        s1 += parts[(z + 1) * n + idx];
        s2 += parts[(z + 2) * n + idx];
        s3 += parts[(z + 3) * n + idx];
      }
      for (; z < J.G; ++z) s0 += parts[z * n + idx];
      const float v = (s0 + s1) + (s2 + s3);
      float* p = outp + (idx / J.K2) * J.ldo + (idx % J.K2);
      *p = J.accumulate ? *p + v : v;
    }
  }
}

template <int THREADS>
MHIMX_DEV void reduce_jobs_block(const ReduceTable& t, int bx, float (*red)[33]) {
  reduce_jobs_block<THREADS, false>(t, bx, red, BagBatch{});
}

}  // namespace mhimx
