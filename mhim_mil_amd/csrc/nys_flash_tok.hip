// nys_flash_tok.hip - the TOKEN-COLUMN kernels of the streamed Nystrom attention (see nys_flash.hip for the scheme): out forward, its
// token-side backward (dq), the token-side backward of a3v (dk, dv) and the cls row.  Their products contract over the landmarks, so
// the waves' partial outputs are summed through LDS: 4 waves x 64 landmarks (1 wave per SIMD, 3 barriers per 32-token half) measured
// faster here than 8 x 32 (twice the partials to write and read per output: 196 vs 140 us for the out forward at T = 50 176), the
// opposite of the landmark-column kernels, which have no cross-wave step at all.
#include "nys_args.hpp"

namespace mhimx {
namespace nytok {

constexpr int NY_THREADS = 256;
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

// ---- token tile: thread = (token pair p = tid >> 3, dim octet g = tid & 7); r[0..1] = token 2p dims 8g..8g+7, r[2..3] = token 2p+1
MHIMX_DEV void ny_load(const float* base, int64_t ld, int64_t t0, int tid, f32x4 (&r)[4]) {
  const float* p = base + (t0 + 2 * (tid >> 3)) * ld + 8 * (tid & 7);
  r[0] = *reinterpret_cast<const f32x4*>(p);
  r[1] = *reinterpret_cast<const f32x4*>(p + 4);
  r[2] = *reinterpret_cast<const f32x4*>(p + ld);
  r[3] = *reinterpret_cast<const f32x4*>(p + ld + 4);
}
// RM image: fragment (tb, ks, hl) at ((tb*2 + ks)*2 + hl) KiB, lane (token & 15, dim octet & 3) x 16 B
MHIMX_DEV void ny_store_rm(char* img, int tid, const f32x4 (&r)[4]) {
  const int g = tid & 7, t = 2 * (tid >> 3);
  f32x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    ny_split44(r[2 * e], r[2 * e + 1], hi, lo);
    char* p = img + ((((t + e) >> 4) * 2 + (g >> 2)) * 2) * 1024 + ((g & 3) * 16 + (((t + e) & 15) ^ g)) * 16;   // ^ (octet + 4 ks): nys_flash.hip
    *reinterpret_cast<f32x4*>(p) = hi;
    *reinterpret_cast<f32x4*>(p + 1024) = lo;
  }
}
MHIMX_DEV f32x4 ny_frag(const char* img, int blk, int step, int hl, int lane) {      // RM images only (the swizzle of ny_store_rm)
  return *reinterpret_cast<const f32x4*>(img + (((blk * 2 + step) * 2 + hl) * 64 + (lane ^ ((lane >> 4) + 4 * step))) * 16);
}

// ---- landmark-side fragments (global fp32 -> registers), M = [256, 64] of this head with row pitch ldm, the wave's landmarks at lm0
struct NyFrag { f32x4 h[4][2], l[4][2]; };
// LM: [lb][ks] : lane (c, kg) holds M[lm0 + 16 lb + c][32 ks + 8 kg .. +7]
MHIMX_DEV void ny_lm_frags(const float* M, int64_t ldm, int lm0, int lane, NyFrag& f) {
  const int c = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int lb = 0; lb < 4; ++lb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* p = M + (int64_t)(lm0 + 16 * lb + c) * ldm + 32 * ks + 8 * kg;
      ny_split44(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), f.h[lb][ks], f.l[lb][ks]);
    }
}
// LT: [db][s] : lane (c, kg) holds M[lm0 + 32 s + 16 blk + 4 kg + i][16 db + c] at slot 4 blk + i
MHIMX_DEV void ny_lt_frags(const float* M, int64_t ldm, int lm0, int lane, NyFrag& f) {
  const int c = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = M[(int64_t)(lm0 + 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3)) * ldm + 16 * db + c];
      ny_split8(v, f.h[db][s], f.l[db][s]);
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
MHIMX_DEV void ny_mma1(const f32x4& ah, const f32x4& al, const f32x4& bh, const f32x4& bl, f32x4& c) {
  c = mt_mfma(al, bh, c);
  c = mt_mfma(ah, bl, c);
  c = mt_mfma(ah, bh, c);
}

MHIMX_DEV void ny_chunk(const NyArgs& g, int ch, int& t_begin, int& t_end) {
  const int64_t tiles = g.T / NY_TT;
  t_begin = (int)(tiles * ch / g.nch);
  t_end = (int)(tiles * (ch + 1) / g.nch);
}

// ---- cross-wave sums of a [64 d][32 token] output: wave w's partial at part[w][token][NY_P68] (an accumulator's four values are
// four consecutive d of one token: one 16-byte store; the reader takes 16 bytes per wave).  Conflict-free both ways: a store phase
// is 8 lanes = 8 consecutive tokens (68 floats apart: 4 banks), a load phase 8 lanes = the 8 d-octets of one token.  The scalar
// [w][d][36] form (64 + 64 four-byte LDS operations per thread and matrix, the reads four-way conflicted) was 18 % of these kernels.
constexpr int NY_P68 = 68;
constexpr int NY_PART_F = 4 * 32 * NY_P68;                    // floats of one matrix's partials
MHIMX_DEV void ny_part_put(float* part, int w, int c, int kg, const f32x4 (&o)[4][2]) {
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) *reinterpret_cast<f32x4*>(part + (w * 32 + 16 * tb + c) * NY_P68 + 16 * db + 4 * kg) = o[db][tb];
}
MHIMX_DEV f32x4 ny_part_sum(const float* part, int tk, int d) {
  const float* p = part + tk * NY_P68 + d;
  return ((*reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + 32 * NY_P68)) + *reinterpret_cast<const f32x4*>(p + 64 * NY_P68)) +
         *reinterpret_cast<const f32x4*>(p + 96 * NY_P68);
}

#define NYT_ZERO(a, n1, n2)                \
  _Pragma("unroll") for (int _i = 0; _i < n1; ++_i) _Pragma("unroll") for (int _j = 0; _j < n2; ++_j) a[_i][_j] = f32x4{0.f, 0.f, 0.f, 0.f}

// ===========================================================================================================================
// forward 2: out = softmax_m(q k~^T) w2.   token-column: S^T[lb][tb] = LM(k~) x RM(q); softmax over all 256 landmarks = over the
// lane's rows, its k-octet lanes and the four waves (LDS); o^T[db][tb] = LT(w2) x P^T, summed over the waves through LDS.
// ===========================================================================================================================
__global__ __launch_bounds__(NY_THREADS) void ny_out_fwd_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  float* red = reinterpret_cast<float*>(sm + NY_IMG);          // [2][4][64]
  float* part = red + 512;                                     // [4][64 d][68]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x, lm0 = 64 * w;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  NyFrag kf, wt;
  ny_lm_frags(g.kl + h * NY_D, g.ldl, lm0, lane, kf);
  ny_lt_frags(g.w2 + (int64_t)h * NY_PART, NY_D, lm0, lane, wt);
  const float* qb = g.q + h * NY_D;
  f32x4 rq[4];
  ny_load(qb, g.ld, (int64_t)t_begin * NY_TT, tid, rq);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    ny_store_rm(sm, tid, rq);
    __syncthreads();
    if (t + 1 < t_end) ny_load(qb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rq);
    f32x4 s[4][4];                                            // [lb][tb]
    NYT_ZERO(s, 4, 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 bh[4], bl[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) { bh[tb] = ny_frag(sm, tb, ks, 0, lane); bl[tb] = ny_frag(sm, tb, ks, 1, lane); }
#pragma unroll
      for (int lb = 0; lb < 4; ++lb) ny_mma_a<4>(kf.h[lb][ks], kf.l[lb][ks], bh, bl, s[lb]);
    }
    float mx[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      float v = s[0][tb][0];
#pragma unroll
      for (int lb = 0; lb < 4; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i) v = fmaxf(v, s[lb][tb][i]);
      v = ny_kgmax(v);
      if (kg == 0) red[w * 64 + 16 * tb + c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      const int tk = 16 * tb + c;
      mx[tb] = fmaxf(fmaxf(red[tk], red[64 + tk]), fmaxf(red[128 + tk], red[192 + tk]));
      float sum = 0.f;
#pragma unroll
      for (int lb = 0; lb < 4; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = NY_EXP2((s[lb][tb][i] - mx[tb]) * g.sl2e);
          s[lb][tb][i] = p;
          sum += p;
        }
      sum = ny_kgsum(sum);
      if (kg == 0) red[256 + w * 64 + tk] = sum;
    }
    f32x4 o[4][4];                                            // [db][tb]
    NYT_ZERO(o, 4, 4);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      f32x4 ph[4], pl[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) ny_split44(s[2 * sx][tb], s[2 * sx + 1][tb], ph[tb], pl[tb]);
#pragma unroll
      for (int db = 0; db < 4; ++db) ny_mma_a<4>(wt.h[db][sx], wt.l[db][sx], ph, pl, o[db]);
    }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(w * 64 + 16 * db + 4 * kg + i) * NY_P68 + 16 * tb + c] = o[db][tb][i];
    __syncthreads();
    {
      const int tk = tid >> 2, qd = tid & 3;
      const float L = ((red[256 + tk] + red[320 + tk]) + red[384 + tk]) + red[448 + tk];
      const float inv = 1.f / L;
      const int64_t row = (int64_t)t * NY_TT + tk;
      float* op = g.out + row * g.ldo + h * NY_D + 16 * qd;
#pragma unroll
      for (int x4 = 0; x4 < 4; ++x4) {
        f32x4 v;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int d = 16 * qd + 4 * x4 + x;
          v[x] = (((part[d * NY_P68 + tk] + part[(64 + d) * NY_P68 + tk]) + part[(128 + d) * NY_P68 + tk]) + part[(192 + d) * NY_P68 + tk]) * inv;
        }
        if (g.accumulate) v += *reinterpret_cast<const f32x4*>(op + 4 * x4);
        *reinterpret_cast<f32x4*>(op + 4 * x4) = v;
      }
      if (qd == 0 && g.lse1_o) {
        const float M = fmaxf(fmaxf(red[tk], red[64 + tk]), fmaxf(red[128 + tk], red[192 + tk]));
        g.lse1_o[(int64_t)h * g.T + row] = M * g.sl2e + log2f(L);
      }
    }
  }
}

// ===========================================================================================================================
// backward of out, token side: dq (and delta = sum_m P dP, saved for the landmark-side kernel).  token-column, 32-token halves:
// S^T = LM(k~) x RM(q), dP^T = LM(w2) x RM(dout), P = exp2(S sl2e - lse1), dS = scale P (dP - delta), dq^T = LT(k~) x dS^T.
// ===========================================================================================================================
__global__ __launch_bounds__(NY_THREADS) void ny_out_bwd_q_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  float* red = reinterpret_cast<float*>(sm + 2 * NY_IMG);      // [4][32]
  float* part = red + 128;                                     // [4][32 tok][68]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x, lm0 = 64 * w;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  NyFrag kf, wf, kt;
  ny_lm_frags(g.kl + h * NY_D, g.ldl, lm0, lane, kf);
  ny_lm_frags(g.w2 + (int64_t)h * NY_PART, NY_D, lm0, lane, wf);
  ny_lt_frags(g.kl + h * NY_D, g.ldl, lm0, lane, kt);
  const float* qb = g.q + h * NY_D;
  const float* gb = g.dout + h * NY_D;
  const float* lse = g.lse1 + (int64_t)h * g.T;
  f32x4 rq[4], rg[4];
  ny_load(qb, g.ld, (int64_t)t_begin * NY_TT, tid, rq);
  ny_load(gb, g.ldd, (int64_t)t_begin * NY_TT, tid, rg);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();
    ny_store_rm(sm, tid, rq);
    ny_store_rm(sm + NY_IMG, tid, rg);
    __syncthreads();
    if (t + 1 < t_end) {
      ny_load(qb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rq);
      ny_load(gb, g.ldd, (int64_t)(t + 1) * NY_TT, tid, rg);
    }
    for (int hf = 0; hf < 2; ++hf) {
      f32x4 s[4][2], dp[4][2];                                // [lb][tb2]
      NYT_ZERO(s, 4, 2);
      NYT_ZERO(dp, 4, 2);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 bh[2], bl[2], eh[2], el[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          bh[tb] = ny_frag(sm, 2 * hf + tb, ks, 0, lane); bl[tb] = ny_frag(sm, 2 * hf + tb, ks, 1, lane);
          eh[tb] = ny_frag(sm + NY_IMG, 2 * hf + tb, ks, 0, lane); el[tb] = ny_frag(sm + NY_IMG, 2 * hf + tb, ks, 1, lane);
        }
#pragma unroll
        for (int lb = 0; lb < 4; ++lb) {
          ny_mma_a<2>(kf.h[lb][ks], kf.l[lb][ks], bh, bl, s[lb]);
          ny_mma_a<2>(wf.h[lb][ks], wf.l[lb][ks], eh, el, dp[lb]);
        }
      }
      const int64_t tok0 = (int64_t)t * NY_TT + 32 * hf;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const float ls = lse[tok0 + 16 * tb + c];
        float dl = 0.f;
#pragma unroll
        for (int lb = 0; lb < 4; ++lb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = NY_EXP2(s[lb][tb][i] * g.sl2e - ls);
            s[lb][tb][i] = p;
            dl += p * dp[lb][tb][i];
          }
        dl = ny_kgsum(dl);
        if (kg == 0) red[w * 32 + 16 * tb + c] = dl;
      }
      __syncthreads();
      f32x4 o[4][2];                                          // [db][tb2]
      NYT_ZERO(o, 4, 2);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const int tk = 16 * tb + c;
        const float dl = ((red[tk] + red[32 + tk]) + red[64 + tk]) + red[96 + tk];
        if (w == 0 && kg == 0) g.delta[(int64_t)h * g.T + tok0 + tk] = dl;
#pragma unroll
        for (int lb = 0; lb < 4; ++lb)
#pragma unroll
          for (int i = 0; i < 4; ++i) s[lb][tb][i] = g.scale * s[lb][tb][i] * (dp[lb][tb][i] - dl);
      }
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        f32x4 ph[2], pl[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) ny_split44(s[2 * sx][tb], s[2 * sx + 1][tb], ph[tb], pl[tb]);
#pragma unroll
        for (int db = 0; db < 4; ++db) ny_mma_a<2>(kt.h[db][sx], kt.l[db][sx], ph, pl, o[db]);
      }
      ny_part_put(part, w, c, kg, o);
      __syncthreads();
      {
        const int tk = tid >> 3, d0 = 8 * (tid & 7);
        float* op = g.out + (tok0 + tk) * g.ldo + h * NY_D + d0;
#pragma unroll
        for (int x4 = 0; x4 < 2; ++x4) *reinterpret_cast<f32x4*>(op + 4 * x4) = ny_part_sum(part, tk, d0 + 4 * x4);
      }
    }
  }
}

// ===========================================================================================================================
// backward of a3v, token side: dk and dv.  token-column, 32-token halves:  S3^T = LM(q~) x RM(k), dP3^T = LM(da3v) x RM(v),
// P = exp2(S sl2e - lse3[lm]), dS = scale P (dP - delta3[lm]);  dv^T = LT(da3v) x P^T,  dk^T = LT(q~) x dS^T.
// MODE 1 (the cls row, no gradients): r[token] = sum_lm u[lm] P[lm, token].
// ===========================================================================================================================
template <int MODE>
__global__ __launch_bounds__(NY_THREADS) void ny_a3v_bwd_t_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  float* lmst = reinterpret_cast<float*>(sm + 2 * NY_IMG);     // lse3[256] | delta3 or u [256]
  float* part = lmst + 512;                                    // MODE 0: [4][32 tok][68] ; MODE 1: red [4][64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x, lm0 = 64 * w;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  lmst[tid] = g.lse3[h * NY_M + tid];
  lmst[256 + tid] = MODE == 0 ? g.delta3[h * NY_M + tid] : g.u[h * NY_M + tid];
  NyFrag qf;
  ny_lm_frags(g.ql + h * NY_D, g.ldl, lm0, lane, qf);
  const float* kb = g.k + h * NY_D;
  const float* vb = g.v + h * NY_D;
  if constexpr (MODE == 1) {
    f32x4 rk[4];
    ny_load(kb, g.ld, (int64_t)t_begin * NY_TT, tid, rk);
    for (int t = t_begin; t < t_end; ++t) {
      __syncthreads();
      ny_store_rm(sm, tid, rk);
      __syncthreads();
      if (t + 1 < t_end) ny_load(kb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rk);
      f32x4 s[4][4];                                          // [lb][tb]
      NYT_ZERO(s, 4, 4);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 bh[4], bl[4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) { bh[tb] = ny_frag(sm, tb, ks, 0, lane); bl[tb] = ny_frag(sm, tb, ks, 1, lane); }
#pragma unroll
        for (int lb = 0; lb < 4; ++lb) ny_mma_a<4>(qf.h[lb][ks], qf.l[lb][ks], bh, bl, s[lb]);
      }
      float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int lb = 0; lb < 4; ++lb) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lmst + lm0 + 16 * lb + 4 * kg);
        const f32x4 uu = *reinterpret_cast<const f32x4*>(lmst + 256 + lm0 + 16 * lb + 4 * kg);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int i = 0; i < 4; ++i) r[tb] += uu[i] * NY_EXP2(s[lb][tb][i] * g.sl2e - ls[i]);
      }
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const float v = ny_kgsum(r[tb]);
        if (kg == 0) part[w * 64 + 16 * tb + c] = v;
      }
      __syncthreads();
      if (tid < 64) g.out[(int64_t)h * g.T + (int64_t)t * NY_TT + tid] = ((part[tid] + part[64 + tid]) + part[128 + tid]) + part[192 + tid];
    }
    return;
  } else {
    NyFrag af, qt, at;
    ny_lm_frags(g.da3v + (int64_t)h * NY_PART, NY_D, lm0, lane, af);
    ny_lt_frags(g.ql + h * NY_D, g.ldl, lm0, lane, qt);
    ny_lt_frags(g.da3v + (int64_t)h * NY_PART, NY_D, lm0, lane, at);
    f32x4 rk[4], rv[4];
    ny_load(kb, g.ld, (int64_t)t_begin * NY_TT, tid, rk);
    ny_load(vb, g.ld, (int64_t)t_begin * NY_TT, tid, rv);
    for (int t = t_begin; t < t_end; ++t) {
      __syncthreads();
      ny_store_rm(sm, tid, rk);
      ny_store_rm(sm + NY_IMG, tid, rv);
      __syncthreads();
      if (t + 1 < t_end) {
        ny_load(kb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rk);
        ny_load(vb, g.ld, (int64_t)(t + 1) * NY_TT, tid, rv);
      }
      for (int hf = 0; hf < 2; ++hf) {
        f32x4 s[4][2], dp[4][2];                              // [lb][tb2]
        NYT_ZERO(s, 4, 2);
        NYT_ZERO(dp, 4, 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f32x4 bh[2], bl[2], eh[2], el[2];
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) {
            bh[tb] = ny_frag(sm, 2 * hf + tb, ks, 0, lane); bl[tb] = ny_frag(sm, 2 * hf + tb, ks, 1, lane);
            eh[tb] = ny_frag(sm + NY_IMG, 2 * hf + tb, ks, 0, lane); el[tb] = ny_frag(sm + NY_IMG, 2 * hf + tb, ks, 1, lane);
          }
#pragma unroll
          for (int lb = 0; lb < 4; ++lb) {
            ny_mma_a<2>(qf.h[lb][ks], qf.l[lb][ks], bh, bl, s[lb]);
            ny_mma_a<2>(af.h[lb][ks], af.l[lb][ks], eh, el, dp[lb]);
          }
        }
#pragma unroll
        for (int lb = 0; lb < 4; ++lb) {
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lmst + lm0 + 16 * lb + 4 * kg);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(lmst + 256 + lm0 + 16 * lb + 4 * kg);
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float p = NY_EXP2(s[lb][tb][i] * g.sl2e - ls[i]);
              s[lb][tb][i] = p;
              dp[lb][tb][i] = g.scale * p * (dp[lb][tb][i] - dl[i]);
            }
        }
        f32x4 ov[4][2], ok[4][2];                               // [db][tb2]
        NYT_ZERO(ov, 4, 2);
        NYT_ZERO(ok, 4, 2);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          f32x4 ph[2], pl[2];
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) ny_split44(s[2 * sx][tb], s[2 * sx + 1][tb], ph[tb], pl[tb]);
#pragma unroll
          for (int db = 0; db < 4; ++db) ny_mma_a<2>(at.h[db][sx], at.l[db][sx], ph, pl, ov[db]);
        }
        __syncthreads();                                      // the previous half's partials have been read
        ny_part_put(part, w, c, kg, ov);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          f32x4 sh[2], sl[2];
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) ny_split44(dp[2 * sx][tb], dp[2 * sx + 1][tb], sh[tb], sl[tb]);
#pragma unroll
          for (int db = 0; db < 4; ++db) ny_mma_a<2>(qt.h[db][sx], qt.l[db][sx], sh, sl, ok[db]);
        }
        const int tk = tid >> 3, d0 = 8 * (tid & 7);
        const int64_t row = (int64_t)t * NY_TT + 32 * hf + tk;
        float* okp = g.out + row * g.ldo + h * NY_D + d0;
        float* ovp = g.out2 + row * g.ldo2 + h * NY_D + d0;
        __syncthreads();
#pragma unroll
        for (int x4 = 0; x4 < 2; ++x4) {
          f32x4 b = ny_part_sum(part, tk, d0 + 4 * x4);
          if (g.accumulate) b += *reinterpret_cast<const f32x4*>(ovp + 4 * x4);
          *reinterpret_cast<f32x4*>(ovp + 4 * x4) = b;
        }
        // dk through the SAME partials (its products were issued above, under the dv exchange)
        __syncthreads();
        ny_part_put(part, w, c, kg, ok);
        __syncthreads();
#pragma unroll
        for (int x4 = 0; x4 < 2; ++x4) *reinterpret_cast<f32x4*>(okp + 4 * x4) = ny_part_sum(part, tk, d0 + 4 * x4);
      }
    }
  }
}

// ===========================================================================================================================
// Token-owning form of the a3v token-side backward (dk, dv): NO barrier and no cross-wave sum in the loop.  The two landmark-side
// matrices (q~ and da3v, [256, 64] of this head) sit in LDS ONCE, row-major, as bf16 hi / lo planes (128-byte rows, 16-byte chunk
// index XOR (row & 7)): an LM fragment (landmark rows, 8 consecutive head dims per lane) is a 16-byte read of a row, an LT fragment
// (head-dim rows, 8 landmarks per lane in accumulator-pair order) is TWO ds_read_b64_tr_b16 of the SAME planes - in a 16-lane group
// lane 4 r + q supplies the address of (landmark r, dims 4 q .. 4 q + 3) and lane l receives dim l of landmarks 0..3 (probed:
// tools/micro/tr_probe.hip).  Both reads are conflict-free under the XOR (b128: 16-lane groups {0-3,12-15,20-27}.. hit 16 different
// 4-bank spans; tr: 32 lanes = 8 rows x 32 bytes, even rows banks 0-31, odd rows 32-63, chunk pair db ^ (r >> 1)).  Four fragment
// images of the landmark-split form (256 KB) would not fit; the two planes pairs are 128 KB.
// A wave owns 16 tokens and ALL 256 landmarks, streamed 32 at a time: S^T, dP^T [2 lb] -> P, dS -> dv^T, dk^T [4 db]
// accumulate in registers.  k / v fragments come straight from global memory (8 consecutive head dims of the lane's token).
// 281 -> 156 us at T = 50 176 (the out backward's twin below: 199 -> 103 us).  Measured and dropped: starting waves 4..7 (the second
// wave of each SIMD) 8..48 x 64 cycles late so that the pair does not run its matrix and VALU phases in lockstep: +-1 %.
// ===========================================================================================================================
constexpr int NY8_THREADS = 512, NY8_NW = 8;
constexpr int NY_PLANE = NY_M * 128;                           // one bf16 plane of a [256, 64] matrix
constexpr int NY_SM_A3_T8 = 4 * NY_PLANE + 2 * NY_M * 4;
typedef __bf16 ny_bf4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* ny_lds;

MHIMX_DEV void ny_plane_fill(char* hi, char* lo, const float* M, int64_t ldm, int tid) {
#pragma unroll
  for (int u = 0; u < NY_M * 8 / NY8_THREADS; ++u) {
    const int unit = tid + NY8_THREADS * u, row = unit >> 3, oct = unit & 7;
    const float* p = M + (int64_t)row * ldm + 8 * oct;
    f32x4 h, l;
    ny_split44(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), h, l);
    const int off = row * 128 + ((oct ^ (row & 7)) << 4);
    *reinterpret_cast<f32x4*>(hi + off) = h;
    *reinterpret_cast<f32x4*>(lo + off) = l;
  }
}
// LT fragment: landmarks 32 sx + {4 kg + i, 16 + 4 kg + i}, head dim 16 db + c; `a` = the lane's address for (sx 0, blk 0, this db)
MHIMX_DEV f32x4 ny_plane_lt(ny_lds a, int sx) {
  typedef __attribute__((address_space(3))) ny_bf4* P;
  const ny_bf4 x = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((P)(a + sx * 4096));
  const ny_bf4 y = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((P)(a + sx * 4096 + 2048));
  bf8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = x[j]; r[4 + j] = y[j]; }
  return __builtin_bit_cast(f32x4, r);
}

__global__ __launch_bounds__(NY8_THREADS) void ny_a3v_bwd_t8_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* q_hi = sm;
  char* q_lo = sm + NY_PLANE;
  char* a_hi = sm + 2 * NY_PLANE;
  char* a_lo = sm + 3 * NY_PLANE;
  float* lmst = reinterpret_cast<float*>(sm + 4 * NY_PLANE);   // lse3[256] | delta3[256]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  ny_plane_fill(q_hi, q_lo, g.ql + h * NY_D, g.ldl, tid);
  ny_plane_fill(a_hi, a_lo, g.da3v + (int64_t)h * NY_PART, NY_D, tid);
  if (tid < NY_M) {
    lmst[tid] = g.lse3[h * NY_M + tid];
    lmst[NY_M + tid] = g.delta3[h * NY_M + tid];
  }
  __syncthreads();
  // per-lane offsets inside a plane
  // (the XOR touches the chunk bits only: ks flips bit 6 of the offset, db bits 5-6 - one register each, an immediate per use)
  const int lm_off0 = c * 128 + ((kg ^ (c & 7)) << 4);         // LM fragment (lb 0, ks 0); ks 1: ^ 64
  const int lt_r = 4 * kg + (c >> 2);
  const int lt_off0 = lt_r * 128 + ((((c & 3) >> 1) ^ (lt_r & 7)) << 4) + (c & 1) * 8;   // LT fragment (sx 0, blk 0, db 0); db: ^ (db << 5)
  const ny_lds lq_hi = (ny_lds)q_hi, lq_lo = (ny_lds)q_lo, la_hi = (ny_lds)a_hi, la_lo = (ny_lds)a_lo;
  const float* kb = g.k + h * NY_D;
  const float* vb = g.v + h * NY_D;
  // 16 tokens per wave; the LDS reads of a phase are issued one phase ahead: the LT fragments of step sx before its S / dP products,
  // the LM fragments of step sx + 1 before its exp / split work (step 8 = step 0 of the wave's next group: the landmark side does not
  // depend on the tokens).  With the reads next to their use the waves sat in s_waitcnt for 47 % of their cycles (SQ_WAIT_ANY).
  const int64_t grp_end = (int64_t)t_end * 4;
  f32x4 lmf[16];                                               // [l2][ks][q~ hi, q~ lo, da hi, da lo]
  auto ld_lm = [&](int sx) {
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = (2 * sx + l2) * 2048 + (lm_off0 ^ (ks << 6));
        lmf[(l2 * 2 + ks) * 4 + 0] = *reinterpret_cast<const f32x4*>(q_hi + o);
        lmf[(l2 * 2 + ks) * 4 + 1] = *reinterpret_cast<const f32x4*>(q_lo + o);
        lmf[(l2 * 2 + ks) * 4 + 2] = *reinterpret_cast<const f32x4*>(a_hi + o);
        lmf[(l2 * 2 + ks) * 4 + 3] = *reinterpret_cast<const f32x4*>(a_lo + o);
      }
  };
  ld_lm(0);
  // kv: [k ks0 | k ks1 | v ks0 | v ks1] x (hi, lo), or the two raw 16-byte halves before the split.  (Requesting the next group's
  // k / v into these registers as soon as the group's last S / dP products have issued measured SLOWER: 343 vs 336 us per entry.)
  f32x4 kv[8];
  auto ld_kv = [&](int64_t grp) {
    const int64_t o = (grp * 16 + c) * g.ld + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kv[2 * ks] = *reinterpret_cast<const f32x4*>(kb + o + 32 * ks);
      kv[2 * ks + 1] = *reinterpret_cast<const f32x4*>(kb + o + 32 * ks + 4);
      kv[4 + 2 * ks] = *reinterpret_cast<const f32x4*>(vb + o + 32 * ks);
      kv[5 + 2 * ks] = *reinterpret_cast<const f32x4*>(vb + o + 32 * ks + 4);
    }
  };
  const int64_t grp0 = (int64_t)t_begin * 4 + w;
  if (grp0 < grp_end) ld_kv(grp0);
  for (int64_t grp = grp0; grp < grp_end; grp += NY8_NW) {
    const int64_t tok = grp * 16 + c;
    if (grp != grp0) ld_kv(grp);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      f32x4 hi, lo;
      ny_split44(kv[2 * x], kv[2 * x + 1], hi, lo);
      kv[2 * x] = hi;
      kv[2 * x + 1] = lo;
    }
    f32x4 ov[4], ok[4];                                        // [db]
#pragma unroll
    for (int db = 0; db < 4; ++db) { ov[db] = f32x4{0.f, 0.f, 0.f, 0.f}; ok[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma nounroll
    for (int sx = 0; sx < 8; ++sx) {
      f32x4 ath[4], atl[4], qth[4], qtl[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        ath[db] = ny_plane_lt(la_hi + (lt_off0 ^ (db << 5)), sx); atl[db] = ny_plane_lt(la_lo + (lt_off0 ^ (db << 5)), sx);
        qth[db] = ny_plane_lt(lq_hi + (lt_off0 ^ (db << 5)), sx); qtl[db] = ny_plane_lt(lq_lo + (lt_off0 ^ (db << 5)), sx);
      }
      f32x4 ls[2], dl[2];                                      // (LDS reads return in order: these must not sit behind the LM prefetch)
#pragma unroll
      for (int l2 = 0; l2 < 2; ++l2) {
        ls[l2] = *reinterpret_cast<const f32x4*>(lmst + 32 * sx + 16 * l2 + 4 * kg);
        dl[l2] = *reinterpret_cast<const f32x4*>(lmst + NY_M + 32 * sx + 16 * l2 + 4 * kg);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 s[2], dp[2];                                       // [lb & 1]
#pragma unroll
      for (int l2 = 0; l2 < 2; ++l2) { s[l2] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[l2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                         // term-major over the four (matrix, landmark block) accumulators
#pragma unroll
        for (int l2 = 0; l2 < 2; ++l2) {
          s[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 1], kv[2 * ks], s[l2]);
          dp[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 3], kv[4 + 2 * ks], dp[l2]);
        }
#pragma unroll
        for (int l2 = 0; l2 < 2; ++l2) {
          s[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 0], kv[2 * ks + 1], s[l2]);
          dp[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 2], kv[5 + 2 * ks], dp[l2]);
        }
#pragma unroll
        for (int l2 = 0; l2 < 2; ++l2) {
          s[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 0], kv[2 * ks], s[l2]);
          dp[l2] = mt_mfma(lmf[(l2 * 2 + ks) * 4 + 2], kv[4 + 2 * ks], dp[l2]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ld_lm((sx + 1) & 7);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pr = NY_EXP2(s[l2][i] * g.sl2e - ls[l2][i]);
          s[l2][i] = pr;
          dp[l2][i] = g.scale * pr * (dp[l2][i] - dl[l2][i]);
        }
      f32x4 ph, pl, sh, sl;
      ny_split44(s[0], s[1], ph, pl);
      ny_split44(dp[0], dp[1], sh, sl);
#pragma unroll
      for (int db = 0; db < 4; ++db) { ov[db] = mt_mfma(atl[db], ph, ov[db]); ok[db] = mt_mfma(qtl[db], sh, ok[db]); }
#pragma unroll
      for (int db = 0; db < 4; ++db) { ov[db] = mt_mfma(ath[db], pl, ov[db]); ok[db] = mt_mfma(qth[db], sl, ok[db]); }
#pragma unroll
      for (int db = 0; db < 4; ++db) { ov[db] = mt_mfma(ath[db], ph, ov[db]); ok[db] = mt_mfma(qth[db], sh, ok[db]); }
    }
    float* okp = g.out + tok * g.ldo + h * NY_D + 4 * kg;
    float* ovp = g.out2 + tok * g.ldo2 + h * NY_D + 4 * kg;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      *reinterpret_cast<f32x4*>(okp + 16 * db) = ok[db];
      f32x4 bv = ov[db];
      if (g.accumulate) bv += *reinterpret_cast<const f32x4*>(ovp + 16 * db);
      *reinterpret_cast<f32x4*>(ovp + 16 * db) = bv;
    }
  }
}

// ===========================================================================================================================
// Token-owning form of the out token-side backward (dq, delta): k~ (LM for S, LT for dq) and w2 (LM for dP) as row-major planes;
// a wave owns 16 tokens and holds S^T and dP^T of ALL 256 landmarks ([16 lb] accumulators each), because delta = sum_m P dP of a
// token must be complete before any dS: the softmax row is the lane's own accumulators + its three k-octet lanes, no cross-wave
// step, no barrier.  q / dout fragments straight from global memory.
// ===========================================================================================================================
__global__ __launch_bounds__(NY8_THREADS) void ny_out_bwd_q8_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* k_hi = sm;
  char* k_lo = sm + NY_PLANE;
  char* w_hi = sm + 2 * NY_PLANE;
  char* w_lo = sm + 3 * NY_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  ny_plane_fill(k_hi, k_lo, g.kl + h * NY_D, g.ldl, tid);
  ny_plane_fill(w_hi, w_lo, g.w2 + (int64_t)h * NY_PART, NY_D, tid);
  __syncthreads();
  int lm_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lm_off[ks] = c * 128 + (((4 * ks + kg) ^ (c & 7)) << 4);
  int lt_off[4];
  {
    const int r = 4 * kg + (c >> 2), r7 = r & 7;
#pragma unroll
    for (int db = 0; db < 4; ++db) lt_off[db] = r * 128 + (((2 * db + ((c & 3) >> 1)) ^ r7) << 4) + (c & 1) * 8;
  }
  const ny_lds lk_hi = (ny_lds)k_hi, lk_lo = (ny_lds)k_lo;
  const float* qb = g.q + h * NY_D;
  const float* gb = g.dout + h * NY_D;
  const float* lse = g.lse1 + (int64_t)h * g.T;
  const int64_t grp_end = (int64_t)t_end * 4;
  for (int64_t grp = (int64_t)t_begin * 4 + w; grp < grp_end; grp += NY8_NW) {
    const int64_t tok = grp * 16 + c;
    f32x4 qh[2], ql[2], gh[2], gl[2];                          // [ks]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* pq = qb + tok * g.ld + 32 * ks + 8 * kg;
      const float* pg = gb + tok * g.ldd + 32 * ks + 8 * kg;
      ny_split44(*reinterpret_cast<const f32x4*>(pq), *reinterpret_cast<const f32x4*>(pq + 4), qh[ks], ql[ks]);
      ny_split44(*reinterpret_cast<const f32x4*>(pg), *reinterpret_cast<const f32x4*>(pg + 4), gh[ks], gl[ks]);
    }
    const float ls = lse[tok];
    f32x4 s[16], dp[16];                                       // [lb]
#pragma unroll
    for (int lb = 0; lb < 16; ++lb) { s[lb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[lb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // fragments of block lb + 1 are requested before the products of block lb (the scheduling barriers keep the 128 reads from
    // being hoisted to the top)
    f32x4 fa[8], fb[8];                                        // k~ (ks0 hi, lo, ks1 hi, lo), w2 (the same)
    auto ld_lm = [&](int lb, f32x4 (&f)[8]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = lb * 2048 + lm_off[ks];
        f[2 * ks] = *reinterpret_cast<const f32x4*>(k_hi + o);
        f[2 * ks + 1] = *reinterpret_cast<const f32x4*>(k_lo + o);
        f[4 + 2 * ks] = *reinterpret_cast<const f32x4*>(w_hi + o);
        f[5 + 2 * ks] = *reinterpret_cast<const f32x4*>(w_lo + o);
      }
    };
    auto mm = [&](int lb, const f32x4 (&f)[8]) {                // term-major over the four (matrix, ks) products
      s[lb] = mt_mfma(f[1], qh[0], s[lb]);
      dp[lb] = mt_mfma(f[5], gh[0], dp[lb]);
      s[lb] = mt_mfma(f[0], ql[0], s[lb]);
      dp[lb] = mt_mfma(f[4], gl[0], dp[lb]);
      s[lb] = mt_mfma(f[0], qh[0], s[lb]);
      dp[lb] = mt_mfma(f[4], gh[0], dp[lb]);
      s[lb] = mt_mfma(f[3], qh[1], s[lb]);
      dp[lb] = mt_mfma(f[7], gh[1], dp[lb]);
      s[lb] = mt_mfma(f[2], ql[1], s[lb]);
      dp[lb] = mt_mfma(f[6], gl[1], dp[lb]);
      s[lb] = mt_mfma(f[2], qh[1], s[lb]);
      dp[lb] = mt_mfma(f[6], gh[1], dp[lb]);
    };
    ld_lm(0, fa);
#pragma unroll
    for (int lb = 0; lb < 16; lb += 2) {
      ld_lm(lb + 1, fb);
      mm(lb, fa);
      __builtin_amdgcn_sched_barrier(0);
      if (lb + 2 < 16) ld_lm(lb + 2, fa);
      mm(lb + 1, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    float dl = 0.f;
#pragma unroll
    for (int lb = 0; lb < 16; ++lb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pr = NY_EXP2(s[lb][i] * g.sl2e - ls);
        s[lb][i] = pr;
        dl += pr * dp[lb][i];
      }
    dl = ny_kgsum(dl);
    if (kg == 0) g.delta[(int64_t)h * g.T + tok] = dl;
    f32x4 o[4];                                                // [db]
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sx = 0; sx < 8; ++sx) {
      f32x4 dh, dlo, th[4], tl[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) { th[db] = ny_plane_lt(lk_hi + lt_off[db], sx); tl[db] = ny_plane_lt(lk_lo + lt_off[db], sx); }
#pragma unroll
      for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[2 * sx + l2][i] = g.scale * s[2 * sx + l2][i] * (dp[2 * sx + l2][i] - dl);
      ny_split44(s[2 * sx], s[2 * sx + 1], dh, dlo);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = mt_mfma(tl[db], dh, o[db]);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = mt_mfma(th[db], dlo, o[db]);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = mt_mfma(th[db], dh, o[db]);
      __builtin_amdgcn_sched_barrier(0);
    }
    float* op = g.out + tok * g.ldo + h * NY_D + 4 * kg;
#pragma unroll
    for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(op + 16 * db) = o[db];
  }
}

// ===========================================================================================================================
// The cls row r[token] = sum_lm u[lm] softmax_n(q~ k^T)[lm, token] (nystrom:143-150) in the token-owning form: q~ as row-major planes,
// a wave owns 16 tokens and all 256 landmarks (S^T[16 lb] in registers), no barrier in the loop.
// ===========================================================================================================================
__global__ __launch_bounds__(NY8_THREADS) void ny_cls_row8_kernel(NyArgs g) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* q_hi = sm;
  char* q_lo = sm + NY_PLANE;
  float* lmst = reinterpret_cast<float*>(sm + 2 * NY_PLANE);   // lse3[256] | u[256]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, kg = lane >> 4;
  const int h = blockIdx.y, ch = blockIdx.x;
  int t_begin, t_end;
  ny_chunk(g, ch, t_begin, t_end);
  ny_plane_fill(q_hi, q_lo, g.ql + h * NY_D, g.ldl, tid);
  if (tid < NY_M) {
    lmst[tid] = g.lse3[h * NY_M + tid];
    lmst[NY_M + tid] = g.u[h * NY_M + tid];
  }
  __syncthreads();
  int lm_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lm_off[ks] = c * 128 + (((4 * ks + kg) ^ (c & 7)) << 4);
  const float* kb = g.k + h * NY_D;
  const int64_t grp_end = (int64_t)t_end * 4;
  for (int64_t grp = (int64_t)t_begin * 4 + w; grp < grp_end; grp += NY8_NW) {
    const int64_t tok = grp * 16 + c;
    f32x4 kh[2], kl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* pk = kb + tok * g.ld + 32 * ks + 8 * kg;
      ny_split44(*reinterpret_cast<const f32x4*>(pk), *reinterpret_cast<const f32x4*>(pk + 4), kh[ks], kl[ks]);
    }
    float r = 0.f;
    f32x4 fa[4], fb[4];                                        // (ks0 hi, lo, ks1 hi, lo) of one landmark block
    auto ld_lm = [&](int lb, f32x4 (&f)[4]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = lb * 2048 + lm_off[ks];
        f[2 * ks] = *reinterpret_cast<const f32x4*>(q_hi + o);
        f[2 * ks + 1] = *reinterpret_cast<const f32x4*>(q_lo + o);
      }
    };
    auto blk = [&](int lb, const f32x4 (&f)[4]) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      s = mt_mfma(f[1], kh[0], s);
      s = mt_mfma(f[0], kl[0], s);
      s = mt_mfma(f[0], kh[0], s);
      s = mt_mfma(f[3], kh[1], s);
      s = mt_mfma(f[2], kl[1], s);
      s = mt_mfma(f[2], kh[1], s);
      const f32x4 ls = *reinterpret_cast<const f32x4*>(lmst + 16 * lb + 4 * kg);
      const f32x4 uu = *reinterpret_cast<const f32x4*>(lmst + NY_M + 16 * lb + 4 * kg);
#pragma unroll
      for (int i = 0; i < 4; ++i) r += uu[i] * NY_EXP2(s[i] * g.sl2e - ls[i]);
    };
    ld_lm(0, fa);
#pragma unroll
    for (int lb = 0; lb < 16; lb += 2) {
      ld_lm(lb + 1, fb);
      blk(lb, fa);
      __builtin_amdgcn_sched_barrier(0);
      if (lb + 2 < 16) ld_lm(lb + 2, fa);
      blk(lb + 1, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    r = ny_kgsum(r);
    if (kg == 0) g.out[(int64_t)h * g.T + tok] = r;
  }
}

constexpr int NY_SM_OUT_FWD = NY_IMG + 512 * 4 + 4 * 64 * NY_P68 * 4;
constexpr int NY_SM_BWD_Q = 2 * NY_IMG + 128 * 4 + NY_PART_F * 4;
constexpr int NY_SM_A3_T = 2 * NY_IMG + 512 * 4 + NY_PART_F * 4;
constexpr int NY_SM_CLS = 2 * NY_IMG + 512 * 4 + 256 * 4;

}  // namespace nytok

int nytok_out_fwd(hipStream_t st, const NyArgs& g) {
  using namespace nytok;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_out_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_OUT_FWD)));
  hipLaunchKernelGGL(ny_out_fwd_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_OUT_FWD, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// The two backward kernels write per-token outputs only (no per-chunk partials), so their chunking is free.  One workgroup per CU
// (the landmark fragments fill 256 VGPRs + 155 / 205 AGPRs: one wave per SIMD): 64 chunks measured 9.48 vs 9.43 ms per c3 step.
static NyArgs ny_tok_chunks(const NyArgs& g0) {
  NyArgs g = g0;
  static const int per = getenv("MHIMX_NYS_TOKCH") ? atoi(getenv("MHIMX_NYS_TOKCH")) : NY_TOKCH;
  const int64_t tiles = g.T / NY_TT;
  g.nch = (int)(tiles < per ? tiles : per);
  return g;
}
int nytok_out_bwd_q(hipStream_t st, const NyArgs& g0) {
  using namespace nytok;
  const NyArgs g = ny_tok_chunks(g0);
  static const bool v1 = getenv("MHIMX_NYS_BWD_Q_V1") != nullptr;       // (experiments: the landmark-split form with cross-wave sums)
  if (!v1) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_out_bwd_q8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NY_PLANE)));
    hipLaunchKernelGGL(ny_out_bwd_q8_kernel, dim3(g.nch, NY_H), dim3(NY8_THREADS), 4 * NY_PLANE, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_out_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_BWD_Q)));
  hipLaunchKernelGGL(ny_out_bwd_q_kernel, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_BWD_Q, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
int nytok_a3v_bwd_t(hipStream_t st, const NyArgs& g0, int mode) {
  using namespace nytok;
  const NyArgs g = mode == 0 ? ny_tok_chunks(g0) : g0;
  static const bool v1 = getenv("MHIMX_NYS_BWD_T_V1") != nullptr;       // (experiments: the landmark-split form with cross-wave sums)
  if (mode == 0 && !v1) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_a3v_bwd_t8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_A3_T8)));
    hipLaunchKernelGGL(ny_a3v_bwd_t8_kernel, dim3(g.nch, NY_H), dim3(NY8_THREADS), NY_SM_A3_T8, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_a3v_bwd_t_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_A3_T));
                        MHIMX_HIP(hipFuncSetAttribute((const void*)ny_a3v_bwd_t_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NY_SM_CLS)));
  if (mode == 1 && !v1) {
    constexpr int sm_cls = 2 * NY_PLANE + 2 * NY_M * 4;
    NyArgs gc = g0;                                            // 66 KB of LDS, 84 VGPRs: two workgroups per CU
    static const int cls_ch = getenv("MHIMX_NYS_CLSCH") ? atoi(getenv("MHIMX_NYS_CLSCH")) : 64;
    const int64_t tiles = gc.T / NY_TT;
    gc.nch = (int)(tiles < cls_ch ? tiles : cls_ch);
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)ny_cls_row8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, sm_cls)));
    hipLaunchKernelGGL(ny_cls_row8_kernel, dim3(gc.nch, NY_H), dim3(NY8_THREADS), sm_cls, st, gc);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (mode == 0) hipLaunchKernelGGL(ny_a3v_bwd_t_kernel<0>, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_A3_T, st, g);
  else hipLaunchKernelGGL(ny_a3v_bwd_t_kernel<1>, dim3(g.nch, NY_H), dim3(NY_THREADS), NY_SM_CLS, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx
