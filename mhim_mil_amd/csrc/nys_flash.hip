// nys_flash.hip — the Nystrom attention block without its n x m matrices (nystrom_attention.py:111-136).
//
// The reference materialises attn1 = softmax(q k~^T) [n, 256] and attn3 = softmax(q~ k^T) [256, n] per head (411 MB each at n = 50 000)
// and the composed path here did the same, forward and backward.  These kernels stream the tokens instead:
//   forward   a3v = softmax_n(q~ k^T) v      : ny_a3v_fwd (online softmax over token chunks, per-chunk partials) + ny_a3v_merge
//             out = softmax_m(q k~^T) w2     : ny_out_fwd  (w2 = pinv(attn2) a3v, [256, 64] per head), writes the row log-sum-exp
//   backward  of out  : ny_out_bwd_q (dq, and delta = rowsum(P dP)) ; ny_out_bwd_l (dk~, dw2: reductions over tokens) + ny_reduce
//             of a3v  : ny_a3v_bwd_t (dk, dv)                       ; ny_a3v_bwd_l (dq~: reduction over tokens)       + ny_reduce
//   and the cls token's attention row  r = (attn1[cls] pinv) attn3  (nystrom:143-150) : ny_cls_attn.
// Every score tile is recomputed from q / k (K = 64: cheap) with the saved log-sum-exps; nothing of size n x 256 touches HBM.
//
// One scheme for all of them.  A workgroup = (head, token chunk), 8 waves (two per SIMD: one wave's exp / split VALU phase runs under
// the other's MFMAs); wave w OWNS landmarks 32w .. 32w+31 and keeps the landmark-side operands of its products in REGISTERS for the
// whole chunk as ready bf16 hi/lo MFMA fragments ("LM" fragments: rows = landmarks, k = the 64 head dims; "LT" fragments: rows = head
// dims, k = its 32 landmarks = one MFMA k-step).  Token tiles (64 tokens x 64 dims of q / k /
// v / dout) are split to bf16 hi/lo ONCE by the loading threads and staged in LDS as fragment images: "RM" (index = token, k = dims)
// and "TR" (index = dim, k = tokens), 1 KiB per fragment = 64 lanes x 16 B contiguous (conflict-free ds_read_b128).
// v_mfma_f32_16x16x32_bf16: A/B lane = (index lane & 15, k-octet lane >> 4), C lane = (col lane & 15, rows 4 (lane >> 4) + i).
//   "token-column" kernels  (acc col = token, rows = landmarks):  S^T = LM(kl) x RM(q);  the accumulator is directly the B operand
//       (index = token, k = landmarks) of the products that contract over landmarks (out, dq, dk, dv);
//   "landmark-column" kernels (acc col = landmark, rows = tokens): S = RM(q) x LM(kl);  the accumulator is the B operand (index =
//       landmark, k = tokens) of the products that contract over tokens (a3v, dk~, dw2, dq~) against TR images.
// The k order inside a 32-deep MFMA step is free as long as both operands agree: an accumulator pair (blocks 2s, 2s+1) gives lane
// k-octet kg the eight k {16(2s) + 4kg + i, 16(2s+1) + 4kg + i}; the LT fragments and TR images are laid out in exactly that order.
// Products are 3-term bf16 (hi hi + hi lo + lo hi, ~2^-16), as everywhere on this path.
// Sums across the four waves (softmax statistics, partial outputs) go through LDS in a FIXED order: results are run-to-run identical.
// Measured and not kept (round 3): two image buffers with ONE barrier per tile and the two waves of a SIMD staging the next tile at
// opposite ends of the iteration (one wave's split / store work under the other's products): a3v forward 108.3 -> 105.9 us per entry.
#include <stdlib.h>
#include <string.h>

#include "nys_args.hpp"

namespace mhimx {

constexpr int NY_THREADS = 512, NY_NW = 8, NY_LB = 2;

MHIMX_DEV float ny_kgsum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
MHIMX_DEV float ny_kgmax(float v) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); return v; }

MHIMX_DEV void ny_split8(const float (&v)[8], f32x4& hi_o, f32x4& lo_o) {
  bf8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 h = (__bf16)v[q];
    hi[q] = h;
    lo[q] = (__bf16)(v[q] - (float)h);
  }
  hi_o = __builtin_bit_cast(f32x4, hi);
  lo_o = __builtin_bit_cast(f32x4, lo);
}
MHIMX_DEV void ny_split44(const f32x4& a, const f32x4& b, f32x4& hi, f32x4& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  ny_split8(v, hi, lo);
}

// ---- token tile: thread = (token pair p = tid >> 4, dim quad g = tid & 15); r[0] = token 2p dims 4g..4g+3, r[1] = token 2p+1
MHIMX_DEV void ny_load(const float* base, int64_t ld, int64_t t0, int tid, f32x4 (&r)[2]) {
  const float* p = base + (t0 + 2 * (tid >> 4)) * ld + 4 * (tid & 15);
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + ld);
}
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
MHIMX_DEV void ny_split4(const f32x4& a, bf4& hi, bf4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const __bf16 h = (__bf16)a[q];
    hi[q] = h;
    lo[q] = (__bf16)(a[q] - (float)h);
  }
}
// Token tile [64 tokens, 64 dims] in LDS: ONE row-major pair of bf16 planes per matrix (hi at +0, lo at +8 KiB; 128-byte rows, the
// 16-byte chunk index XORed with row & 7) serves BOTH fragment forms - "RM" (index = token, k = 8 consecutive dims: a 16-byte read of
// the token's row) and "TR" (index = dim, k = 8 tokens in accumulator-pair order: two ds_read_b64_tr_b16, see nys_flash_tok.hip for
// the lane mapping of that instruction).  The first form kept two ready-made fragment images per matrix (RM and TR): the tile was
// split twice and the TR image written with eight 4-byte stores per thread (56 % of the LDS-active cycles were bank conflicts before
// a slot swizzle, profiles/r03_pmc_lds_c3.md); here a thread writes its 4 dims of a token as ONE 8-byte store per plane (the 16 lanes of
// a store group = one token's row: conflict-free), and all three reads are conflict-free under the XOR.
typedef __attribute__((address_space(3))) char* ny_lds;
typedef __bf16 ny_bf4 __attribute__((ext_vector_type(4)));
MHIMX_DEV void ny_store_pl(char* pl, int tid, const f32x4 (&r)[2]) {
  const int g = tid & 15, t = 2 * (tid >> 4);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    bf4 hi, lo;
    ny_split4(r[e], hi, lo);
    const int row = t + e;
    char* p = pl + row * 128 + (((g >> 1) ^ (row & 7)) << 4) + (g & 1) * 8;
    *reinterpret_cast<bf4*>(p) = hi;
    *reinterpret_cast<bf4*>(p + 8192) = lo;
  }
}
MHIMX_DEV f32x4 ny_frag_rm(const char* pl, int tb, int ks, int hl, int lane) {
  const int c = lane & 15, kg = lane >> 4;
  return *reinterpret_cast<const f32x4*>(pl + hl * 8192 + (16 * tb + c) * 128 + (((4 * ks + kg) ^ (c & 7)) << 4));
}
// TR fragment (db, ts): lane (dim c, kg) holds tokens 32 ts + {4 kg + i, 16 + 4 kg + i} of dim 16 db + c
MHIMX_DEV f32x4 ny_frag_tr(const char* pl, int db, int ts, int hl, int lane) {
  typedef __attribute__((address_space(3))) ny_bf4* P;
  const int c = lane & 15, kg = lane >> 4;
  const int r = 4 * kg + (c >> 2);
  const ny_lds a = (ny_lds)pl + hl * 8192 + (32 * ts + r) * 128 + (((2 * db + ((c & 3) >> 1)) ^ (r & 7)) << 4) + (c & 1) * 8;
  const ny_bf4 x = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((P)a);
  const ny_bf4 y = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((P)(a + 2048));
  bf8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = x[j]; o[4 + j] = y[j]; }
  return __builtin_bit_cast(f32x4, o);
}

// ---- landmark-side fragments (global fp32 -> registers), M = [256, 64] of this head with row pitch ldm, the wave's landmarks at lm0
struct NyLM { f32x4 h[NY_LB][2], l[NY_LB][2]; };   // [lb][ks] : lane (c, kg) holds M[lm0 + 16 lb + c][32 ks + 8 kg .. +7]
struct NyLT { f32x4 h[4], l[4]; };                 // [db]     : lane (c, kg) holds M[lm0 + 16 blk + 4 kg + i][16 db + c] at slot 4 blk + i
MHIMX_DEV void ny_lm_frags(const float* M, int64_t ldm, int lm0, int lane, NyLM& f) {
  const int c = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int lb = 0; lb < NY_LB; ++lb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* p = M + (int64_t)(lm0 + 16 * lb + c) * ldm + 32 * ks + 8 * kg;
      ny_split44(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), f.h[lb][ks], f.l[lb][ks]);
    }
}
MHIMX_DEV void ny_lt_frags(const float* M, int64_t ldm, int lm0, int lane, NyLT& f) {
  const int c = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = M[(int64_t)(lm0 + 16 * (j >> 2) + 4 * kg + (j & 3)) * ldm + 16 * db + c];
    ny_split8(v, f.h[db], f.l[db]);
  }
}

// c[j] += a x b[j], j < NB, 3 bf16 terms, term-major (NB independent MFMAs between two on the same accumulator)
template <int NB>
MHIMX_DEV void ny_mma_a(const f32x4& ah, const f32x4& al, const f32x4 (&bh)[NB], const f32x4 (&bl)[NB], f32x4 (&c)[NB]) {
#pragma unroll
  for (int j = 0; j < NB; ++j) c[j] = mt_mfma(al, bh[j], c[j]);
#pragma unroll
  for (int j = 0; j < NB; ++j) c[j] = mt_mfma(ah, bl[j], c[j]);
#pragma unroll
  for (int j = 0; j < NB; ++j) c[j] = mt_mfma(ah, bh[j], c[j]);
}

MHIMX_DEV void ny_chunk(const NyArgs& g, int ch, int& t_begin, int& t_end) {
  const int64_t tiles = g.T / NY_TT;
  t_begin = (int)(tiles * ch / g.nch);
  t_end = (int)(tiles * (ch + 1) / g.nch);
}

#define NY_ZERO(a, n1, n2)                \
  _Pragma("unroll") for (int _i = 0; _i < n1; ++_i) _Pragma("unroll") for (int _j = 0; _j < n2; ++_j) a[_i][_j] = f32x4{0.f, 0.f, 0.f, 0.f}
#define NY_IDS                                                                                              \
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;                \
  const int h = blockIdx.y, ch = blockIdx.x, lm0 = 16 * NY_LB * w;                                          \
  int t_begin, t_end;                                                                                       \
  ny_chunk(g, ch, t_begin, t_end);                                                                          \
  (void)c; (void)kg

// ===========================================================================================================================
// forward 1: a3v partials.  landmark-column: S[tb][lb] = RM(k) x LM(q~); online softmax per landmark (= per lane column);
// o^T[db][lb] += TR(v) x P.      part[h][ch] = { o [256][64] | m [256] | l [256] }
// ===========================================================================================================================
__global__ __launch_bounds__(NY_THREADS) void ny_a3v_fwd_kernel(NyArgs g) {
  __shared__ __attribute__((aligned(16))) char sm[2 * NY_IMG];
  NY_IDS;
  NyLM qf;
  ny_lm_frags(g.ql + h * NY_D, g.ldl, lm0, lane, qf);
  f32x4 o[4][NY_LB];
  NY_ZERO(o, 4, NY_LB);
  float m[NY_LB], l[NY_LB];
#pragma unroll
  for (int lb = 0; lb < NY_LB; ++lb) { m[lb] = -__builtin_inff(); l[lb] = 0.f; }
  const float* kb = g.k + h * NY_D;
  const float* vb = g.v + h * NY_D;
  f32x4 rk[2], rv[2];
  ny_load(kb, g.ld, (int64_t)t_begin * NY_TT, tid, rk);
  ny_load(vb, g.ld, (int64_t)t_begin * NY_TT, tid, rv);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    ny_store_pl(sm, tid, rk);
    ny_store_pl(sm + NY_IMG, tid, rv);
    __syncthreads();
    if (t + 1 < t_end) {
      ny_load(kb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rk);
      ny_load(vb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rv);
    }
    f32x4 s[4][NY_LB];                                        // [tb][lb]
    NY_ZERO(s, 4, NY_LB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 bh[NY_LB], bl[NY_LB];
#pragma unroll
      for (int lb = 0; lb < NY_LB; ++lb) { bh[lb] = qf.h[lb][ks]; bl[lb] = qf.l[lb][ks]; }
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) ny_mma_a<NY_LB>(ny_frag_rm(sm, tb, ks, 0, lane), ny_frag_rm(sm, tb, ks, 1, lane), bh, bl, s[tb]);
    }
#pragma unroll
    for (int lb = 0; lb < NY_LB; ++lb) {
      float mx = s[0][lb][0];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[tb][lb][i]);
      mx = ny_kgmax(mx);
      const float mn = fmaxf(m[lb], mx);
      const float alpha = NY_EXP2((m[lb] - mn) * g.sl2e);
      m[lb] = mn;
      const float off = mn * g.sl2e;
      float sum = 0.f;
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = NY_EXP2(fmaf(s[tb][lb][i], g.sl2e, -off));
          s[tb][lb][i] = p;
          sum += p;
        }
      l[lb] = l[lb] * alpha + ny_kgsum(sum);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db][lb] *= alpha;
    }
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) {
      f32x4 ph[NY_LB], pl[NY_LB];
#pragma unroll
      for (int lb = 0; lb < NY_LB; ++lb) ny_split44(s[2 * ts][lb], s[2 * ts + 1][lb], ph[lb], pl[lb]);
#pragma unroll
      for (int db = 0; db < 4; ++db)
        ny_mma_a<NY_LB>(ny_frag_tr(sm + NY_IMG, db, ts, 0, lane), ny_frag_tr(sm + NY_IMG, db, ts, 1, lane), ph, pl, o[db]);
    }
  }
  float* pp = g.part + ((int64_t)h * g.nch + ch) * (NY_PART + 2 * NY_M);
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int lb = 0; lb < NY_LB; ++lb) *reinterpret_cast<f32x4*>(pp + (lm0 + 16 * lb + c) * NY_D + 16 * db + 4 * kg) = o[db][lb];
  if (kg == 0) {
#pragma unroll
    for (int lb = 0; lb < NY_LB; ++lb) {
      pp[NY_PART + lm0 + 16 * lb + c] = m[lb];
      pp[NY_PART + NY_M + lm0 + 16 * lb + c] = l[lb];
    }
  }
}

// a3v[h][lm][d] = sum_ch o e^{(m_ch - M) scale} / L,  lse3[h][lm] = M sl2e + log2 L  (base-2 log-sum-exp of the scaled scores).
// 64 outputs x 4 chunk quarters per workgroup: the maximum over ALL chunks first (LDS), then every quarter's weighted sums against
// it, added in a fixed order.
__global__ __launch_bounds__(256) void ny_a3v_merge_kernel(const float* part, int nch, float sl2e, float* a3v, float* lse3) {
  __shared__ f32x4 sacc[3][64];
  __shared__ float smax[4][64], sl[3][64];
  const int q = threadIdx.x >> 6, t = threadIdx.x & 63, idx = blockIdx.x * 64 + t;            // (h, lm, d4)
  const int d4 = idx & 15, lm = (idx >> 4) & 255, h = idx >> 12;
  const float* pp = part + (int64_t)h * nch * (NY_PART + 2 * NY_M);
  const int c0 = nch * q / 4, c1 = nch * (q + 1) / 4;
  float M = -__builtin_inff();
  for (int ch = c0; ch < c1; ++ch) M = fmaxf(M, pp[(int64_t)ch * (NY_PART + 2 * NY_M) + NY_PART + lm]);
  smax[q][t] = M;
  __syncthreads();
  M = fmaxf(fmaxf(smax[0][t], smax[1][t]), fmaxf(smax[2][t], smax[3][t]));
  float L = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ch = c0; ch < c1; ++ch) {
    const float* pc = pp + (int64_t)ch * (NY_PART + 2 * NY_M);
    const float e = NY_EXP2((pc[NY_PART + lm] - M) * sl2e);
    L += pc[NY_PART + NY_M + lm] * e;
    acc += *reinterpret_cast<const f32x4*>(pc + lm * NY_D + 4 * d4) * e;
  }
  if (q) { sacc[q - 1][t] = acc; sl[q - 1][t] = L; }
  __syncthreads();
  if (q == 0) {
    acc = ((acc + sacc[0][t]) + sacc[1][t]) + sacc[2][t];
    L = ((L + sl[0][t]) + sl[1][t]) + sl[2][t];
    const float inv = 1.f / L;
    *reinterpret_cast<f32x4*>(a3v + ((int64_t)h * NY_M + lm) * NY_D + 4 * d4) = acc * inv;
    if (d4 == 0) lse3[h * NY_M + lm] = M * sl2e + log2f(L);
  }
}

// ===========================================================================================================================
// backward of out, landmark side: dk~ and dw2 partials.  landmark-column, 32-token steps:
// S = RM(q) x LM(k~), dP = RM(dout) x LM(w2), P, dS with the row's lse1 / delta;  dw2^T += TR(dout) x P,  dk~^T += TR(q) x dS.
// ===========================================================================================================================
__global__ __launch_bounds__(NY_THREADS) void ny_out_bwd_l_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  float* rowst = reinterpret_cast<float*>(sm + 2 * NY_IMG);    // lse1[64] | delta[64]
  NY_IDS;
  NyLM kf, wf;
  ny_lm_frags(g.kl + h * NY_D, g.ldl, lm0, lane, kf);
  ny_lm_frags(g.w2 + (int64_t)h * NY_PART, NY_D, lm0, lane, wf);
  f32x4 dkl[4][NY_LB], dw2[4][NY_LB];                          // [db][lb]
  NY_ZERO(dkl, 4, NY_LB);
  NY_ZERO(dw2, 4, NY_LB);
  const float* qb = g.q + h * NY_D;
  const float* gb = g.dout + h * NY_D;
  f32x4 rq[2], rg[2];
  float rst = 0.f;
  auto load_st = [&](int t) {
    if (tid < 64) rst = g.lse1[(int64_t)h * g.T + (int64_t)t * NY_TT + tid];
    else if (tid < 128) rst = g.delta_i[(int64_t)h * g.T + (int64_t)t * NY_TT + tid - 64];
  };
  ny_load(qb, g.ld, (int64_t)t_begin * NY_TT, tid, rq);
  ny_load(gb, g.ldd, (int64_t)t_begin * NY_TT, tid, rg);
  load_st(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    ny_store_pl(sm, tid, rq);
    ny_store_pl(sm + NY_IMG, tid, rg);
    if (tid < 128) rowst[tid] = rst;
    __syncthreads();
    if (t + 1 < t_end) {
      ny_load(qb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rq);
      ny_load(gb, g.ldd, (int64_t)(t + 1) * NY_TT, tid, rg);
      load_st(t + 1);
    }
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) {
      f32x4 s[2][NY_LB], dp[2][NY_LB];                        // [tb2][lb]
      NY_ZERO(s, 2, NY_LB);
      NY_ZERO(dp, 2, NY_LB);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 bh[NY_LB], bl[NY_LB], eh[NY_LB], el[NY_LB];
#pragma unroll
        for (int lb = 0; lb < NY_LB; ++lb) { bh[lb] = kf.h[lb][ks]; bl[lb] = kf.l[lb][ks]; eh[lb] = wf.h[lb][ks]; el[lb] = wf.l[lb][ks]; }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          ny_mma_a<NY_LB>(ny_frag_rm(sm, 2 * ts + tb, ks, 0, lane), ny_frag_rm(sm, 2 * ts + tb, ks, 1, lane), bh, bl, s[tb]);
          ny_mma_a<NY_LB>(ny_frag_rm(sm + NY_IMG, 2 * ts + tb, ks, 0, lane), ny_frag_rm(sm + NY_IMG, 2 * ts + tb, ks, 1, lane), eh, el, dp[tb]);
        }
      }
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(rowst + 32 * ts + 16 * tb + 4 * kg);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(rowst + 64 + 32 * ts + 16 * tb + 4 * kg);
#pragma unroll
        for (int lb = 0; lb < NY_LB; ++lb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = NY_EXP2(fmaf(s[tb][lb][i], g.sl2e, -ls[i]));
            s[tb][lb][i] = p;
            dp[tb][lb][i] = g.scale * p * (dp[tb][lb][i] - dl[i]);
          }
      }
      f32x4 ph[NY_LB], pl[NY_LB], sh[NY_LB], sl[NY_LB];
#pragma unroll
      for (int lb = 0; lb < NY_LB; ++lb) {
        ny_split44(s[0][lb], s[1][lb], ph[lb], pl[lb]);
        ny_split44(dp[0][lb], dp[1][lb], sh[lb], sl[lb]);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        ny_mma_a<NY_LB>(ny_frag_tr(sm + NY_IMG, db, ts, 0, lane), ny_frag_tr(sm + NY_IMG, db, ts, 1, lane), ph, pl, dw2[db]);
        ny_mma_a<NY_LB>(ny_frag_tr(sm, db, ts, 0, lane), ny_frag_tr(sm, db, ts, 1, lane), sh, sl, dkl[db]);
      }
    }
  }
  float* p1 = g.part + ((int64_t)h * g.nch + ch) * NY_PART;
  float* p2 = g.part2 + ((int64_t)h * g.nch + ch) * NY_PART;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int lb = 0; lb < NY_LB; ++lb) {
      *reinterpret_cast<f32x4*>(p1 + (lm0 + 16 * lb + c) * NY_D + 16 * db + 4 * kg) = dkl[db][lb];
      *reinterpret_cast<f32x4*>(p2 + (lm0 + 16 * lb + c) * NY_D + 16 * db + 4 * kg) = dw2[db][lb];
    }
}

// out[h][lm][d] (row pitch ldo, head pitch hs) = sum_ch part[h][ch][lm][d].  A workgroup = 64 outputs (16 bytes each) x 4 chunk
// quarters, summed through LDS in a fixed order (128 workgroups of 32-deep serial sums were 11 us on half the CUs: latency-bound).
// TWO reductions per launch (blockIdx.y; the second may be absent).
struct NyReduce { const float* part; float* out; int64_t ldo, hs; };
__global__ __launch_bounds__(256) void ny_reduce_kernel(NyReduce r0, NyReduce r1, int nch) {
  __shared__ f32x4 sm[3][64];
  const NyReduce r = blockIdx.y == 0 ? r0 : r1;
  const int q = threadIdx.x >> 6, idx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int d4 = idx & 15, lm = (idx >> 4) & 255, h = idx >> 12;
  const float* pp = r.part + (int64_t)h * nch * NY_PART + lm * NY_D + 4 * d4;
  const int c0 = nch * q / 4, c1 = nch * (q + 1) / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ch = c0; ch < c1; ++ch) acc += *reinterpret_cast<const f32x4*>(pp + (int64_t)ch * NY_PART);
  if (q) sm[q - 1][threadIdx.x & 63] = acc;
  __syncthreads();
  if (q == 0) {
    acc = ((acc + sm[0][threadIdx.x]) + sm[1][threadIdx.x]) + sm[2][threadIdx.x];
    *reinterpret_cast<f32x4*>(r.out + (int64_t)h * r.hs + (int64_t)lm * r.ldo + 4 * d4) = acc;
  }
}
static void ny_reduce(hipStream_t st, int nch, const NyReduce& a, const NyReduce* b = nullptr) {
  hipLaunchKernelGGL(ny_reduce_kernel, dim3(NY_H * NY_M * 16 / 64, b ? 2 : 1), dim3(256), 0, st, a, b ? *b : a, nch);
}

// delta3[h][lm] = sum_d a3v da3v
__global__ __launch_bounds__(256) void ny_delta3_kernel(const float* a3v, const float* da3v, float* delta3) {
  const int idx = blockIdx.x * 256 + threadIdx.x;            // (h, lm)
  const float* a = a3v + (int64_t)idx * NY_D;
  const float* b = da3v + (int64_t)idx * NY_D;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < NY_D; d += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a + d), y = *reinterpret_cast<const f32x4*>(b + d);
    s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  }
  delta3[idx] = s;
}

// ===========================================================================================================================
// backward of a3v, landmark side: dq~ partials.  landmark-column:  S3 = RM(k) x LM(q~), dP3 = RM(v) x LM(da3v),
// dS = scale P (dP - delta3) with the lane's landmark statistics;  dq~^T += TR(k) x dS.
// ===========================================================================================================================
__global__ __launch_bounds__(NY_THREADS) void ny_a3v_bwd_l_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  NY_IDS;
  NyLM qf, af;
  ny_lm_frags(g.ql + h * NY_D, g.ldl, lm0, lane, qf);
  ny_lm_frags(g.da3v + (int64_t)h * NY_PART, NY_D, lm0, lane, af);
  float ls[NY_LB], dl[NY_LB];
#pragma unroll
  for (int lb = 0; lb < NY_LB; ++lb) {
    ls[lb] = g.lse3[h * NY_M + lm0 + 16 * lb + c];
    dl[lb] = g.delta3[h * NY_M + lm0 + 16 * lb + c];
  }
  f32x4 dq[4][NY_LB];                                          // [db][lb]
  NY_ZERO(dq, 4, NY_LB);
  const float* kb = g.k + h * NY_D;
  const float* vb = g.v + h * NY_D;
  f32x4 rk[2], rv[2];
  ny_load(kb, g.ld, (int64_t)t_begin * NY_TT, tid, rk);
  ny_load(vb, g.ld, (int64_t)t_begin * NY_TT, tid, rv);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    ny_store_pl(sm, tid, rk);
    ny_store_pl(sm + NY_IMG, tid, rv);
    __syncthreads();
    if (t + 1 < t_end) {
      ny_load(kb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rk);
      ny_load(vb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rv);
    }
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) {
      f32x4 s[2][NY_LB], dp[2][NY_LB];                        // [tb2][lb]
      NY_ZERO(s, 2, NY_LB);
      NY_ZERO(dp, 2, NY_LB);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 bh[NY_LB], bl[NY_LB], eh[NY_LB], el[NY_LB];
#pragma unroll
        for (int lb = 0; lb < NY_LB; ++lb) { bh[lb] = qf.h[lb][ks]; bl[lb] = qf.l[lb][ks]; eh[lb] = af.h[lb][ks]; el[lb] = af.l[lb][ks]; }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          ny_mma_a<NY_LB>(ny_frag_rm(sm, 2 * ts + tb, ks, 0, lane), ny_frag_rm(sm, 2 * ts + tb, ks, 1, lane), bh, bl, s[tb]);
          ny_mma_a<NY_LB>(ny_frag_rm(sm + NY_IMG, 2 * ts + tb, ks, 0, lane), ny_frag_rm(sm + NY_IMG, 2 * ts + tb, ks, 1, lane), eh, el, dp[tb]);
        }
      }
      f32x4 sh[NY_LB], sl[NY_LB];
#pragma unroll
      for (int lb = 0; lb < NY_LB; ++lb) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            dp[tb][lb][i] = g.scale * NY_EXP2(fmaf(s[tb][lb][i], g.sl2e, -ls[lb])) * (dp[tb][lb][i] - dl[lb]);
        ny_split44(dp[0][lb], dp[1][lb], sh[lb], sl[lb]);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db)
        ny_mma_a<NY_LB>(ny_frag_tr(sm, db, ts, 0, lane), ny_frag_tr(sm, db, ts, 1, lane), sh, sl, dq[db]);
    }
  }
  float* p1 = g.part + ((int64_t)h * g.nch + ch) * NY_PART;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int lb = 0; lb < NY_LB; ++lb) *reinterpret_cast<f32x4*>(p1 + (lm0 + 16 * lb + c) * NY_D + 16 * db + 4 * kg) = dq[db][lb];
}

// ===========================================================================================================================
// forward 2, token-owning form: out = softmax_m(q k~^T) w2 with NO barrier in the loop.  The landmark-side operands of BOTH products
// sit in LDS as ready fragments for the whole chunk (k~: [16 lb][2 ks] LM fragments, w2: [4 db][8 s] LT fragments: 64 KiB each);
// a wave owns 32 tokens and ALL 256 landmarks: S^T[16 lb][2 tb] in registers, the softmax over the landmarks is the lane's own rows
// plus its three k-octet lanes, o^T[4 db][2 tb] = LT(w2) x P^T needs no cross-wave sum.  q fragments come straight from global
// memory (a lane's 8 consecutive head dims of its token: two 16-byte loads per k-step).  8 independent waves per workgroup.
// ===========================================================================================================================
constexpr int NY_SM_OUT8 = 2 * 65536;
__global__ __launch_bounds__(NY_THREADS) void ny_out_fwd_tok8_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  {
    const float* kl = g.kl + h * NY_D;
    const float* w2 = g.w2 + (int64_t)h * NY_PART;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int unit = tid + NY_THREADS * u, fl = unit & 63, f = unit >> 6;           // fragment f (0..31), lane fl
      const int fc = fl & 15, fk = fl >> 4;
      {                                                                               // k~ LM fragment f = lb * 2 + ks
        const int lb = f >> 1, ks = f & 1;
        const float* p = kl + (int64_t)(16 * lb + fc) * g.ldl + 32 * ks + 8 * fk;
        f32x4 hi, lo;
        ny_split44(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), hi, lo);
        *reinterpret_cast<f32x4*>(sm + (f * 2) * 1024 + fl * 16) = hi;
        *reinterpret_cast<f32x4*>(sm + (f * 2 + 1) * 1024 + fl * 16) = lo;
      }
      {                                                                               // w2 LT fragment f = db * 8 + s
        const int db = f >> 3, sx = f & 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w2[(int64_t)(32 * sx + 16 * (j >> 2) + 4 * fk + (j & 3)) * NY_D + 16 * db + fc];
        f32x4 hi, lo;
        ny_split8(v, hi, lo);
        *reinterpret_cast<f32x4*>(sm + 65536 + (f * 2) * 1024 + fl * 16) = hi;
        *reinterpret_cast<f32x4*>(sm + 65536 + (f * 2 + 1) * 1024 + fl * 16) = lo;
      }
    }
  }
  __syncthreads();
  const float* qb = g.q + h * NY_D;
  f32x4 raw[2][2][2];                                          // [tb][ks][half]: the NEXT group's q rows, in flight under the softmax / PV
  auto ld_q = [&](int64_t tk) {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float* p = qb + (tk + 16 * tb + c) * g.ld + 32 * ks + 8 * kg;
        raw[tb][ks][0] = *reinterpret_cast<const f32x4*>(p);
        raw[tb][ks][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
  };
  const int64_t grp0 = (int64_t)t_begin * 2 + w, grp_end = (int64_t)t_end * 2;
  if (grp0 < grp_end) ld_q(grp0 * 32);
  for (int64_t grp = grp0; grp < grp_end; grp += NY_NW) {
    const int64_t tk0 = grp * 32;
    f32x4 qh[2][2], ql[2][2];                                  // [ks][tb]
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ny_split44(raw[tb][ks][0], raw[tb][ks][1], qh[ks][tb], ql[ks][tb]);
    f32x4 s[16][2];                                            // [lb][tb]
    NY_ZERO(s, 16, 2);
    // fragments of landmark block lb + 1 are requested before the MFMAs of block lb; the scheduling barrier keeps the compiler from
    // hoisting all 64 fragment reads to the top (256 VGPRs of fragments: 1.8 KB of scratch per lane without it)
    f32x4 fa[4], fb[4];
    auto ld_k = [&](int lb, f32x4 (&f)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const f32x4*>(sm + (lb * 4 + q) * 1024 + lane * 16);      // ks0 hi, ks0 lo, ks1 hi, ks1 lo
    };
    ld_k(0, fa);
#pragma unroll
    for (int lb = 0; lb < 16; lb += 2) {
      ld_k(lb + 1, fb);
      ny_mma_a<2>(fa[0], fa[1], qh[0], ql[0], s[lb]);
      ny_mma_a<2>(fa[2], fa[3], qh[1], ql[1], s[lb]);
      __builtin_amdgcn_sched_barrier(0);
      if (lb + 2 < 16) ld_k(lb + 2, fa);
      ny_mma_a<2>(fb[0], fb[1], qh[0], ql[0], s[lb + 1]);
      ny_mma_a<2>(fb[2], fb[3], qh[1], ql[1], s[lb + 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (grp + NY_NW < grp_end) ld_q((grp + NY_NW) * 32);
    __builtin_amdgcn_sched_barrier(0);
    float inv[2], lse[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      float mx = s[0][tb][0];
#pragma unroll
      for (int lb = 0; lb < 16; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[lb][tb][i]);
      mx = ny_kgmax(mx);
      const float off = mx * g.sl2e;
      float sum = 0.f;
#pragma unroll
      for (int lb = 0; lb < 16; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = NY_EXP2(fmaf(s[lb][tb][i], g.sl2e, -off));
          s[lb][tb][i] = p;
          sum += p;
        }
      sum = ny_kgsum(sum);
      inv[tb] = 1.f / sum;
      lse[tb] = off + log2f(sum);
    }
    f32x4 o[4][2];                                             // [db][tb]
    NY_ZERO(o, 4, 2);
    auto ld_w = [&](int sx, f32x4 (&f)[8]) {                     // the four d blocks of landmark step sx: hi, lo each
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f[2 * db] = *reinterpret_cast<const f32x4*>(sm + 65536 + ((db * 8 + sx) * 2) * 1024 + lane * 16);
        f[2 * db + 1] = *reinterpret_cast<const f32x4*>(sm + 65536 + ((db * 8 + sx) * 2 + 1) * 1024 + lane * 16);
      }
    };
    f32x4 wa[8], wb[8], old[4][2];
    NY_ZERO(old, 4, 2);
    ld_w(0, wa);
#pragma unroll
    for (int sx = 0; sx < 8; sx += 2) {
      f32x4 ph[2], pl[2];
      ld_w(sx + 1, wb);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) ny_split44(s[2 * sx][tb], s[2 * sx + 1][tb], ph[tb], pl[tb]);
#pragma unroll
      for (int db = 0; db < 4; ++db) ny_mma_a<2>(wa[2 * db], wa[2 * db + 1], ph, pl, o[db]);
      __builtin_amdgcn_sched_barrier(0);
      if (sx + 2 < 8) ld_w(sx + 2, wa);
      if (sx == 4 && g.accumulate) {                            // out += : what the buffer holds, requested under the last PV steps (half the
#pragma unroll                                                  // score registers are free by now); a read at the store was +10 us per launch
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int db = 0; db < 4; ++db)
            old[db][tb] = *reinterpret_cast<const f32x4*>(g.out + (tk0 + 16 * tb + c) * g.ldo + h * NY_D + 16 * db + 4 * kg);
      }
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) ny_split44(s[2 * sx + 2][tb], s[2 * sx + 3][tb], ph[tb], pl[tb]);
#pragma unroll
      for (int db = 0; db < 4; ++db) ny_mma_a<2>(wb[2 * db], wb[2 * db + 1], ph, pl, o[db]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const int64_t row = tk0 + 16 * tb + c;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        float* op = g.out + row * g.ldo + h * NY_D + 16 * db + 4 * kg;
        f32x4 v;                                               // (rounded product, then the add: out += is bit for bit plain + old)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float pr = o[db][tb][i] * inv[tb];
          asm volatile("" : "+v"(pr));                         // (no fused multiply-add across the two roundings)
          v[i] = pr + old[db][tb][i];
        }
        *reinterpret_cast<f32x4*>(op) = v;
      }
      if (kg == 0 && g.lse1_o) g.lse1_o[(int64_t)h * g.T + row] = lse[tb];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
constexpr int NY_SM_BWD_L = 2 * NY_IMG + 128 * 4;
constexpr int NY_SM_A3_L = 2 * NY_IMG;

static int ny_base(const mhimx_nys* a, NyArgs& g, const char* who) {
  MHIMX_CHECK_ARG(a && a->T >= NY_TT && a->T % NY_TT == 0, "%s: T must be a positive multiple of 64", who);
  MHIMX_CHECK_ARG(a->ld % 4 == 0 && a->ldl % 4 == 0, "%s: row pitches must be multiples of 4 floats", who);
  MHIMX_CHECK_ARG(a->ws && a->ws_floats >= mhimx_nys_ws_floats(a->T), "%s: workspace too small", who);
  memset(&g, 0, sizeof(g));
  g.q = a->q; g.k = a->k; g.v = a->v; g.ld = a->ld; g.T = a->T; g.ql = a->ql; g.kl = a->kl; g.ldl = a->ldl;
  g.scale = a->scale;
  g.sl2e = a->scale * 1.4426950408889634f;
  const int64_t tiles = a->T / NY_TT;
  // one workgroup per CU: 64 chunks (two per CU; the landmark-column kernels would fit) measured SLOWER - a3v forward + merge 146 vs
  // 125 us, out backward 365 vs 341 us: twice the partials to write and merge, and these kernels are not latency-bound
  static const int lch = getenv("MHIMX_NYS_LMCH") ? atoi(getenv("MHIMX_NYS_LMCH")) : NY_MAXCH;   // (experiments; <= NY_MAXCH)
  const int per = lch < 1 ? 1 : (lch > NY_MAXCH ? NY_MAXCH : lch);
  g.nch = (int)(tiles < per ? tiles : per);
  return 0;
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_nys_ws_floats(int64_t T) {
  (void)T;
  return (int64_t)NY_H * NY_MAXCH * (2 * NY_PART + 2 * NY_M) + NY_H * NY_M;
}

extern "C" int mhimx_nys_a3v_fwd(void* stream, const mhimx_nys* a, float* a3v, float* lse3) {
  NyArgs g;
  if (int e = ny_base(a, g, "nys_a3v_fwd")) return e;
  MHIMX_CHECK_ARG(a->k && a->v && a->ql && a3v && lse3 && aligned16(a->k) && aligned16(a->v) && aligned16(a->ql), "nys_a3v_fwd: null / unaligned operands");
  g.part = a->ws;
  hipLaunchKernelGGL(ny_a3v_fwd_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), 0, (hipStream_t)stream, g);
  hipLaunchKernelGGL(ny_a3v_merge_kernel, dim3(NY_H * NY_M * 16 / 64), dim3(256), 0, (hipStream_t)stream, a->ws, g.nch, g.sl2e, a3v, lse3);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_nys_out_fwd(void* stream, const mhimx_nys* a, const float* w2, float* out, int64_t ldo, float* lse1, int32_t accumulate) {
  NyArgs g;
  if (int e = ny_base(a, g, "nys_out_fwd")) return e;
  MHIMX_CHECK_ARG(a->q && a->kl && w2 && out && aligned16(a->q) && aligned16(a->kl) && aligned16(out) && ldo % 4 == 0, "nys_out_fwd: null / unaligned operands");
  g.w2 = w2; g.out = out; g.ldo = ldo; g.lse1_o = lse1; g.accumulate = accumulate;
  static const bool v1 = getenv("MHIMX_NYS_OUT_V1") != nullptr;         // (experiments: the landmark-split form with cross-wave sums)
  if (v1) return nytok_out_fwd((hipStream_t)stream, g);
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_out_fwd_tok8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_OUT8)));
  const int64_t tiles = g.T / NY_TT;
  g.nch = (int)(tiles < NY_TOKCH ? tiles : NY_TOKCH);                    // 128 KB of LDS: one workgroup per CU
  hipLaunchKernelGGL(ny_out_fwd_tok8_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_OUT8, (hipStream_t)stream, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_nys_out_bwd(void* stream, const mhimx_nys* a, const float* w2, const float* dout, int64_t ldd, const float* lse1,
                                 float* delta1, float* dq, int64_t lddq, float* dkl, int64_t lddl, float* dw2) {
  NyArgs g;
  if (int e = ny_base(a, g, "nys_out_bwd")) return e;
  MHIMX_CHECK_ARG(a->q && a->kl && w2 && dout && lse1 && delta1 && dq && dkl && dw2, "nys_out_bwd: null operands");
  MHIMX_CHECK_ARG(aligned16(a->q) && aligned16(dout) && aligned16(dq) && aligned16(dkl) && ldd % 4 == 0 && lddq % 4 == 0 && lddl % 4 == 0,
                  "nys_out_bwd: unaligned operands");
  g.w2 = w2; g.dout = dout; g.ldd = ldd; g.lse1 = lse1; g.delta = delta1; g.delta_i = delta1; g.out = dq; g.ldo = lddq;
  g.part = a->ws; g.part2 = a->ws + (int64_t)NY_H * NY_MAXCH * NY_PART;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_out_bwd_l_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_BWD_L)));
  hipStream_t st = (hipStream_t)stream;
  if (int e = nytok_out_bwd_q(st, g)) return e;
  hipLaunchKernelGGL(ny_out_bwd_l_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_BWD_L, st, g);
  const NyReduce ra = {g.part, dkl, lddl, (int64_t)NY_D}, rb = {g.part2, dw2, (int64_t)NY_D, (int64_t)NY_PART};
  ny_reduce(st, g.nch, ra, &rb);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_nys_a3v_bwd(void* stream, const mhimx_nys* a, const float* a3v, const float* da3v, const float* lse3, float* dk,
                                 float* dv, int64_t lddk, int32_t accumulate_dv, float* dql, int64_t lddl) {
  NyArgs g;
  if (int e = ny_base(a, g, "nys_a3v_bwd")) return e;
  MHIMX_CHECK_ARG(a->k && a->v && a->ql && a3v && da3v && lse3 && dk && dv && dql, "nys_a3v_bwd: null operands");
  MHIMX_CHECK_ARG(aligned16(a->k) && aligned16(a->v) && aligned16(dk) && aligned16(dv) && aligned16(dql) && lddk % 4 == 0 && lddl % 4 == 0,
                  "nys_a3v_bwd: unaligned operands");
  float* delta3 = a->ws + (int64_t)NY_H * NY_MAXCH * (2 * NY_PART + 2 * NY_M);
  g.da3v = da3v; g.lse3 = lse3; g.delta3 = delta3; g.out = dk; g.ldo = lddk; g.out2 = dv; g.ldo2 = lddk; g.accumulate = accumulate_dv;
  g.part = a->ws;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_a3v_bwd_l_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_A3_L)));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ny_delta3_kernel, dim3(NY_H * NY_M / 256), dim3(256), 0, st, a3v, da3v, delta3);
  if (int e = nytok_a3v_bwd_t(st, g, 0)) return e;
  hipLaunchKernelGGL(ny_a3v_bwd_l_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_A3_L, st, g);
  ny_reduce(st, g.nch, NyReduce{g.part, dql, lddl, (int64_t)NY_D});
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_nys_cls_attn(void* stream, const mhimx_nys* a, const float* lse3, const float* u, float* r) {
  NyArgs g;
  if (int e = ny_base(a, g, "nys_cls_attn")) return e;
  MHIMX_CHECK_ARG(a->k && a->ql && lse3 && u && r && aligned16(a->k) && aligned16(a->ql), "nys_cls_attn: null / unaligned operands");
  g.lse3 = lse3; g.u = u; g.out = r;
  return nytok_a3v_bwd_t((hipStream_t)stream, g, 1);
}
