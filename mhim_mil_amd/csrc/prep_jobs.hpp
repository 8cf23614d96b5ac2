// prep_jobs.hpp — the parameter-only preparation jobs of a train step (mhimx_prep_batch: weight images, transposes, the Merge's query side)
// as a device function: the body of prep_batch_kernel (gemm_dma.hip) and of the RIDER blocks of the teacher's one-pass scorer launch
// (scorer_fused.hip, round 4: every job the teacher's forward does not need - the student's scorer images, the transposes of the
// backward, the Merge preparation's chain - runs in the workgroup slots that launch leaves free instead of in front of the projection).
#pragma once
#include "mma_tile.hpp"
#include "mca2_prep.hpp"

namespace mhimx {

typedef float pj_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 pj_b8 __attribute__((ext_vector_type(8)));
// x = hi + lo in bf16 (the split of gemm_dma.hip's Frag<BF16X3>: the same roundings)
MHIMX_DEV void pj_split(const float (&x)[8], pj_b8& hi, pj_b8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}
MHIMX_DEV void pj_split2(const pj_f4& a, const pj_f4& b, pj_b8& hi, pj_b8& lo) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  pj_split(x, hi, lo);
}

constexpr int PREP_MERGE_MAX = 8;
struct PrepJobs { mhimx_prep_job j[MHIMX_PREP_MAX]; int first[MHIMX_PREP_MAX + 1]; int n; Merge2PrepArgs m2; int64_t m2_shift[PREP_MERGE_MAX]; };
constexpr int PREP_LDS_FLOATS = 6 * M2_E + 6 * 64;             // the largest job's LDS (kind 6; kind 0 takes 32 x 33)
static_assert(PREP_LDS_FLOATS >= 32 * 33, "the transpose tile fits the job LDS");

// block `block` (of pj.first[pj.n]) of the job table; 256 threads; lds: PREP_LDS_FLOATS floats, 16-byte aligned
MHIMX_DEV void prep_job_block(const PrepJobs& pj, int block, float* lds) {
  // 1-D grid: job q owns blocks [first[q], first[q+1]) - sized per job (the bag's paired-plane image wants thousands of
  // workgroups, a weight transpose a few dozen; a rectangular grid would launch tens of thousands of empty blocks)
  int q = 0;
  while (q + 1 < pj.n && block >= pj.first[q + 1]) ++q;
  const mhimx_prep_job jb = pj.j[q];
  const int bid = block - pj.first[q], nblk = pj.first[q + 1] - pj.first[q];
  const int64_t R = jb.R, C = jb.C;
  if (jb.kind == 0) {
    float (*tile)[33] = reinterpret_cast<float(*)[33]>(lds);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                   // 32 x 8
    const int64_t tiles_c = (C + 31) / 32, ntiles = ((R + 31) / 32) * tiles_c;
    for (int64_t t = bid; t < ntiles; t += nblk) {
      const int64_t r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
      for (int i = ty; i < 32; i += 8) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? jb.in[r * C + c] : 0.f;
      }
      __syncthreads();
      for (int i = ty; i < 32; i += 8) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (r < R && c < C) jb.out[c * R + r] = tile[tx][i];
      }
      __syncthreads();
    }
  } else if (jb.kind == 1) {
    const int64_t K8 = C / 8, n = R * K8;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
      const pj_f4 a = *reinterpret_cast<const pj_f4*>(jb.in + i * 8);
      const pj_f4 b = *reinterpret_cast<const pj_f4*>(jb.in + i * 8 + 4);
      pj_b8 hi, lo;
      pj_split2(a, b, hi, lo);
      pj_f4* o = reinterpret_cast<pj_f4*>(jb.out + i * 8);
      o[0] = __builtin_bit_cast(pj_f4, hi);
      o[1] = __builtin_bit_cast(pj_f4, lo);
    }
  } else if (jb.kind == 2) {
    const int64_t n = R * C;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) jb.out[i] = jb.in[i];
  } else if (jb.kind == 3) {
    if (bid == 0 && threadIdx.x == 0) *reinterpret_cast<uint64_t*>(jb.out) += 1;
  } else if (jb.kind == 4) {
    // B-operand fragment image for v_mfma_f32_32x32x16_bf16: item (nt, ks, lane) holds the 8 hi | 8 lo bf16 of
    // in[32 nt + (lane & 31)][16 ks + 8 (lane >> 5) .. + 8]: a wave's fragment load is 2 KB contiguous
    const int64_t KS = C / 16, n = R * C / 8;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
      const int64_t lane = i & 63, ks = (i >> 6) % KS, nt = (i >> 6) / KS;
      const float* src = jb.in + (32 * nt + (lane & 31)) * C + 16 * ks + 8 * (lane >> 5);
      const pj_f4 a = *reinterpret_cast<const pj_f4*>(src);
      const pj_f4 b = *reinterpret_cast<const pj_f4*>(src + 4);
      pj_b8 hi, lo;
      pj_split2(a, b, hi, lo);
      pj_f4* o = reinterpret_cast<pj_f4*>(jb.out + i * 8);
      o[0] = __builtin_bit_cast(pj_f4, hi);
      o[1] = __builtin_bit_cast(pj_f4, lo);
    }
  } else if (jb.kind == 5) {
    // the kind-4 image of in^T ([C, R]) made straight from in[R,C]: item (nt, ks, lane) holds in[16 ks + 8 (lane >> 5) + u][32 nt + (lane & 31)],
    // u < 8 (a job of the same launch cannot read the transpose another job is still writing)
    const int64_t KS = R / 16, n = R * C / 8;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
      const int64_t lane = i & 63, ks = (i >> 6) % KS, nt = (i >> 6) / KS;
      const float* src = jb.in + (16 * ks + 8 * (lane >> 5)) * C + 32 * nt + (lane & 31);
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = src[u * C];
      pj_b8 hi, lo;
      pj_split(x, hi, lo);
      pj_f4* o = reinterpret_cast<pj_f4*>(jb.out + i * 8);
      o[0] = __builtin_bit_cast(pj_f4, hi);
      o[1] = __builtin_bit_cast(pj_f4, lo);
    }
  } else if (jb.kind == 7) {
    // paired planes of in^T [C, R]: item (g8, m) holds in[8 g8 + u][m], u < 8 (adjacent threads = adjacent columns m: coalesced reads)
    const int64_t R8 = R / 8, n = R8 * C;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
      const int64_t g8 = i / C, m = i % C;
      const float* src = jb.in + 8 * g8 * C + m;
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = src[u * C];
      pj_b8 hi, lo;
      pj_split(x, hi, lo);
      pj_f4* o = reinterpret_cast<pj_f4*>(jb.out + (m * R8 + g8) * 8);
      o[0] = __builtin_bit_cast(pj_f4, hi);
      o[1] = __builtin_bit_cast(pj_f4, lo);
    }
  } else if (jb.kind == 8) {
    // B-operand fragment image for v_mfma_f32_16x16x32_bf16 of in[R, C] with the rows padded to a multiple of 16 (zeros): item (nb, ks, lane)
    // holds the 8 hi | 8 lo bf16 of in[16 nb + (lane & 15)][32 ks + 8 (lane >> 4) .. + 8] - a wave's fragment load is 2 KB contiguous
    // (what mhimx_proj_score.wa16 takes: the teacher's scorer inside the projection's epilogue, bag_project_ws.hip)
    const int64_t KS = C / 32, NB = (R + 15) / 16, n = NB * KS * 64;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
      const int64_t lane = i & 63, ks = (i >> 6) % KS, nb = (i >> 6) / KS;
      const int64_t row = 16 * nb + (lane & 15);
      pj_f4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
      if (row < R) {
        const float* src = jb.in + row * C + 32 * ks + 8 * (lane >> 4);
        a = *reinterpret_cast<const pj_f4*>(src);
        b = *reinterpret_cast<const pj_f4*>(src + 4);
      }
      pj_b8 hi, lo;
      pj_split2(a, b, hi, lo);
      pj_f4* o = reinterpret_cast<pj_f4*>(jb.out + i * 8);
      o[0] = __builtin_bit_cast(pj_f4, hi);
      o[1] = __builtin_bit_cast(pj_f4, lo);
    }
  } else if (jb.kind == 9) {
    // the bag as the weight-gradient product's D-side operand image (wgrad.hip, bag_wgrad_dma_kernel): tile (ks, cb) = rows 32 ks .. + 32,
    // columns 256 cb .. + 256; thread -> (row octet, column quad q): 8 rows x 16 B in flight, then the 8 k-consecutive values of each of
    // its four columns as 16 B of bf16 hi and 16 B of lo - a wave's stores are 1 KiB contiguous (column 4 q + j lives in slot 64 j + q)
    const int64_t nJ = C / 256, ntiles = ((R + 31) / 32) * nJ;
    const int koct = threadIdx.x >> 6, q = threadIdx.x & 63;
    for (int64_t t = bid; t < ntiles; t += nblk) {
      const int64_t ks = t / nJ, cb = t % nJ, r0 = ks * 32 + koct * 8;
      const float* src = jb.in + r0 * C + cb * 256 + 4 * q;
      pj_f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r0 + u < R ? *reinterpret_cast<const pj_f4*>(src + u * C) : pj_f4{0.f, 0.f, 0.f, 0.f};
      char* tile = reinterpret_cast<char*>(jb.out) + t * 32768 + ((koct * 2) * 256 + q) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x[8] = {v[0][j], v[1][j], v[2][j], v[3][j], v[4][j], v[5][j], v[6][j], v[7][j]};
        pj_b8 hi, lo;
        pj_split(x, hi, lo);
        *reinterpret_cast<pj_f4*>(tile + j * 1024) = __builtin_bit_cast(pj_f4, hi);
        *reinterpret_cast<pj_f4*>(tile + j * 1024 + 4096) = __builtin_bit_cast(pj_f4, lo);
      }
    }
  } else if (jb.kind == 10) {
    // ((int64_t*)out)[i] = R + i, i < C: the constant tail of a step's row list (the k merged-token rows behind a bag's N feature rows)
    int64_t* o = reinterpret_cast<int64_t*>(jb.out);
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < C; i += (int64_t)nblk * 256) o[i] = R + i;
  } else if (jb.kind == 6) {
    Merge2Ws w = pj.m2.w;
    const int64_t sh = pj.m2_shift[jb.R];             // (only the fields the preparation writes are shifted)
    w.gq += sh; w.gmean += sh; w.grstd += sh; w.Q += sh; w.aq += sh; w.aqf += sh; w.gtf_aq += sh; w.gate += sh;
    merge2_prep_body(bid, lds, pj.m2.q_param, pj.m2.ln_w, pj.m2.ln_b, pj.m2.wq, pj.m2.wkv, pj.m2.k, pj.m2.scale, w);
  }
}

int prep_jobs_fill(const mhimx_prep_job* jobs, int n, PrepJobs* out);      // gemm_dma.hip (host): validated table, returns its block count
int prep_batch(hipStream_t st, const mhimx_prep_job* jobs, int n);         // the launch of its own

}  // namespace mhimx
