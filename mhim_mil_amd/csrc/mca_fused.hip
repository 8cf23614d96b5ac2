// mca_fused.hip — Merge's cross attention forward over the key rows as ONE kernel per (32-row tile, pair of heads)
// (mhim_modules/merge.py:8-65 MCA.forward restated: kv = to_kv(LN(x)), dots = q k^T / sqrt(d), attn = dropout(softmax_n(dots)),
// out = attn v).  Replaces, for the 512-wide / 8-head / 64-dim configuration, the K/V projection GEMM + its split-K slab
// reduce + the row pass of mca.hip (three launches, K/V written and read back once more):
//   1. the tile's 32 normalised rows land in LDS once (64.5 KB);
//   2. wave w projects them onto 64 columns of Wkv on the matrix cores (bf16x3; K of head a, K of head b, V of head a, V of
//      head b), B fragments from the prep-time fragment image of Wkv; the tile goes to LDS and to HBM (the backward reads K/V);
//   3. dots for the k queries (8 lanes per row: 2 heads x 4 quarter dot products, DPP quad reductions), stored for the backward;
//   4. per (head, query): max / probabilities / sum over the tile's rows, then the probability-weighted V sums - one
//      log-sum-exp partial per tile, merged by mca_fwd_final_kernel exactly like the row pass's partials (31 instead of 247).
#include <math.h>

#include "common.hpp"

namespace mhimx {

typedef float mf_f32x16 __attribute__((ext_vector_type(16)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 mf_b8 __attribute__((ext_vector_type(8)));

#ifdef MHIMX_MF_PROF
__device__ unsigned long long mf_prof[16];
#define MF_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 50) mf_prof[i] = wall_clock64(); } while (0)
extern "C" int mhimx_mf_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mf_prof), 16 * 8); }
#else
#define MF_STAMP(i)
#endif
constexpr int MF_ROWS = 32, MF_E = 512, MF_I = 512, MF_LD = MF_E + 4, MF_KVLD = 256 + 4, MF_THREADS = 256, MF_MAXK = 16;
constexpr size_t MF_SMEM = (size_t)(MF_ROWS * MF_LD + MF_ROWS * MF_KVLD + MF_MAXK * 128 + 3 * MF_ROWS * 2 * MF_MAXK + 4 * MF_MAXK) * sizeof(float);

MHIMX_DEV void mf_split(const mf_f4& a, const mf_f4& b, mf_b8& hi, mf_b8& lo) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}

__global__ __launch_bounds__(MF_THREADS) void mca_fused_fwd_kernel(
    const float* __restrict__ xn, int64_t R, const float* __restrict__ wkv_frag, const float* __restrict__ Q, int kq, int heads,
    float scale, float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick, float* __restrict__ KV, float* __restrict__ dots,
    float* __restrict__ pm, float* __restrict__ pl, float* __restrict__ po) {
  extern __shared__ __attribute__((aligned(16))) float mf_sm[];
  float* Xs = mf_sm;                                   // [32][516] LN(x) rows
  float* KVs = Xs + MF_ROWS * MF_LD;                   // [32][260]: K head a | K head b | V head a | V head b
  float* qs = KVs + MF_ROWS * MF_KVLD;                 // [kq][128] scaled queries of the two heads
  float* ds = qs + MF_MAXK * 128;                      // [32][2][16] dots
  float* ps = ds + MF_ROWS * 2 * MF_MAXK;              // [32][2][16] e^{d - m}
  float* pk = ps + MF_ROWS * 2 * MF_MAXK;              // [32][2][16] ... times the dropout keep factor
  float* ms = pk + MF_ROWS * 2 * MF_MAXK;              // [2][16] max, [2][16] sum
  const uint64_t seed = eff_seed(seed0, tick);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kg = lane >> 5;
  const int tile = blockIdx.x >> 2, hg = blockIdx.x & 3;
  const int64_t row0 = (int64_t)tile * MF_ROWS;
  const int inner = heads * 64;                         // = 512
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;

  MF_STAMP(0);
  // ---- 1. rows and queries -> LDS
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = tid + MF_THREADS * i, r = f >> 7, c4 = f & 127;
    const int64_t n = row0 + r;
    mf_f4 v = reinterpret_cast<const mf_f4*>(xn + (n < R ? n : R - 1) * MF_E)[c4];
    if (n >= R) v = mf_f4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<mf_f4*>(Xs + r * MF_LD + 4 * c4) = v;
  }
  for (int i = tid; i < kq * 128; i += MF_THREADS) {
    const int qi = i >> 7, c = i & 127;
    qs[i] = Q[(int64_t)qi * inner + (2 * hg) * 64 + c] * scale;
  }
  __syncthreads();
  MF_STAMP(1);
  // ---- 2. K / V tile on the matrix cores: wave 0/1 -> K of heads 2hg, 2hg+1; wave 2/3 -> V of the same heads
  {
    const int h = 2 * hg + (wave & 1);
    const int col_g = (wave >> 1) * inner + h * 64;     // first of this wave's 64 columns of Wkv's output
    const float* fptr = wkv_frag + ((int64_t)(col_g / 32) * (MF_E / 16) * 64 + lane) * 8;      // + (nt * 32 + ks) * 512 floats
    const float* aptr = Xs + r32 * MF_LD + 8 * kg;
    mf_f32x16 acc[2], acc2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[nt][i] = 0.f; acc2[nt][i] = 0.f; }
    constexpr int PF = 4, KS = MF_E / 16;
    mf_f4 bh_[PF][2], bl_[PF][2];
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bh_[q][nt] = *reinterpret_cast<const mf_f4*>(fptr + (nt * KS + q) * 512);
        bl_[q][nt] = *reinterpret_cast<const mf_f4*>(fptr + (nt * KS + q) * 512 + 4);
      }
    // software pipeline: the next k-step's A fragment is read from LDS and split to bf16 hi/lo (VALU) while this k-step's six
    // MFMAs are in flight - one wave per SIMD has no other wave to hide that latency behind
    mf_b8 ah, al;
    mf_split(*reinterpret_cast<const mf_f4*>(aptr), *reinterpret_cast<const mf_f4*>(aptr + 4), ah, al);
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const int k1 = ks + 1 < KS ? ks + 1 : ks;
      const mf_f4 na0 = *reinterpret_cast<const mf_f4*>(aptr + 16 * k1), na1 = *reinterpret_cast<const mf_f4*>(aptr + 16 * k1 + 4);
      const int kn = ks + PF < KS ? ks + PF : ks;
      // term-major issue order: two MFMAs into the same accumulator are at least two others apart (a dependent MFMA issued
      // back to back waits out the first one's full latency)
      mf_b8 bh[2], bl[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bh[nt] = __builtin_bit_cast(mf_b8, bh_[ks % PF][nt]);
        bl[nt] = __builtin_bit_cast(mf_b8, bl_[ks % PF][nt]);
        bh_[ks % PF][nt] = *reinterpret_cast<const mf_f4*>(fptr + (nt * KS + kn) * 512);
        bl_[ks % PF][nt] = *reinterpret_cast<const mf_f4*>(fptr + (nt * KS + kn) * 512 + 4);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt], acc2[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt], acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt], acc2[nt], 0, 0, 0);
      mf_b8 nh, nl;
      mf_split(na0, na1, nh, nl);
      ah = nh;
      al = nl;
    }
    MF_STAMP(2);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i >> 2) + 4 * kg + (i & 3);
        const float v = acc[nt][i] + acc2[nt][i];
        KVs[row * MF_KVLD + 64 * wave + 32 * nt + r32] = v;
        const int64_t n = row0 + row;
        if (n < R) KV[n * (2 * inner) + col_g + 32 * nt + r32] = v;
      }
  }
  __syncthreads();
  MF_STAMP(3);
  // ---- 3. dots: 8 lanes per row = 2 heads x 4 quarters of the 64-dim dot product
  {
    const int row = tid >> 3, hs = (tid >> 2) & 1, qt = tid & 3;
    const float* kr = KVs + row * MF_KVLD + 64 * hs + 16 * qt;
    mf_f4 kv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = reinterpret_cast<const mf_f4*>(kr)[j];
    const int64_t n = row0 + row;
    for (int i = 0; i < kq; ++i) {
      const mf_f4* qv = reinterpret_cast<const mf_f4*>(qs + i * 128 + 64 * hs + 16 * qt);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const mf_f4 q4 = qv[j];
        d += kv[j][0] * q4[0] + kv[j][1] * q4[1] + kv[j][2] * q4[2] + kv[j][3] * q4[3];
      }
      d += dpp_mov<0xB1, 0xf>(0.f, d);                  // lanes ^1
      d += dpp_mov<0x4E, 0xf>(0.f, d);                  // lanes ^2
      if (qt == 0) {
        ds[(row * 2 + hs) * MF_MAXK + i] = d;
        if (n < R) dots[((int64_t)(2 * hg + hs) * kq + i) * R + n] = d;
      }
    }
  }
  __syncthreads();
  MF_STAMP(4);
  // ---- 4. per (head, query): a half-wave owns one (head, query) pair - lane = row: max, e^{d - max} (x dropout keep), sum by
  // 32-lane butterflies; no serial row loops, no extra LDS round trips
  const int nrows = (int)((R - row0) < MF_ROWS ? (R - row0) : MF_ROWS);
  for (int combo = tid >> 5; combo < 2 * kq; combo += MF_THREADS / 32) {
    const int hs = combo / kq, i = combo % kq, r = tid & 31;
    const float d = r < nrows ? ds[(r * 2 + hs) * MF_MAXK + i] : -INFINITY;
    float m = d;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float p = r < nrows ? __expf(d - m) : 0.f;
    float ksf = 1.f;
    if (drop_p > 0.f) ksf = drop_keep(seed, (uint64_t)((2 * hg + hs) * kq + i), (uint32_t)(row0 + r), drop_p) ? keep_scale : 0.f;
    pk[(r * 2 + hs) * MF_MAXK + i] = p * ksf;
    float l = p;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if (r == 0) { ms[hs * MF_MAXK + i] = m; ms[2 * MF_MAXK + hs * MF_MAXK + i] = l; }
  }
  __syncthreads();
  MF_STAMP(5);
  // probability-weighted V sums: thread = (head, column, half of the rows); V[r, c] is read once per row for all the queries
  {
    const int hs = tid >> 7, rh = (tid >> 6) & 1, c = tid & 63;
    float o[MF_MAXK];
#pragma unroll
    for (int i = 0; i < MF_MAXK; ++i) o[i] = 0.f;
#pragma unroll 4
    for (int rr = 0; rr < MF_ROWS / 2; ++rr) {
      const int r = rh * (MF_ROWS / 2) + rr;
      const float v = KVs[r * MF_KVLD + 128 + 64 * hs + c];
      const mf_f4* pr = reinterpret_cast<const mf_f4*>(pk + (r * 2 + hs) * MF_MAXK);
#pragma unroll
      for (int q = 0; q < MF_MAXK / 4; ++q) {
        if (4 * q < kq) {
          const mf_f4 p4 = pr[q];
          o[4 * q] += p4[0] * v; o[4 * q + 1] += p4[1] * v; o[4 * q + 2] += p4[2] * v; o[4 * q + 3] += p4[3] * v;
        }
      }
    }
    float* osum = Xs;                                  // [2 halves][2 heads][16][64] (the row tile is dead by now)
#pragma unroll
    for (int i = 0; i < MF_MAXK; ++i)
      if (i < kq) osum[((rh * 2 + hs) * MF_MAXK + i) * 64 + c] = o[i];
  }
  __syncthreads();
  if (tid < 128) {
    const int hs = tid >> 6, c = tid & 63;
    const int h = 2 * hg + hs;
    const float* osum = Xs;
    for (int i = 0; i < kq; ++i) {
      const int64_t slot = (int64_t)tile * heads * kq + h * kq + i;
      if (c == 0) { pm[slot] = ms[hs * MF_MAXK + i]; pl[slot] = ms[2 * MF_MAXK + hs * MF_MAXK + i]; }
      po[slot * 64 + c] = osum[((0 * 2 + hs) * MF_MAXK + i) * 64 + c] + osum[((1 * 2 + hs) * MF_MAXK + i) * 64 + c];
    }
  }
  MF_STAMP(6);
}

bool mca_fused_ok(int64_t E, int64_t heads, int64_t dh, int64_t k, const float* wkv_frag, const float* xn, const float* KV) {
  return E == MF_E && heads == 8 && dh == 64 && k >= 1 && k <= MF_MAXK && wkv_frag && aligned16(wkv_frag) && aligned16(xn) && aligned16(KV);
}

// returns the number of partial blocks written to pm / pl / po (= row tiles), < 0 on error
int mca_fused_fwd(hipStream_t st, const float* xn, int64_t R, const float* wkv_frag, const float* Q, int kq, int heads, float scale,
                  float drop_p, uint64_t seed, const uint64_t* tick, float* KV, float* dots, float* pm, float* pl, float* po) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)mca_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MF_SMEM)));
  const int tiles = (int)cdiv(R, MF_ROWS);
  hipLaunchKernelGGL(mca_fused_fwd_kernel, dim3((unsigned)(tiles * 4)), dim3(MF_THREADS), MF_SMEM, st, xn, R, wkv_frag, Q, kq, heads, scale,
                     drop_p, seed, tick, KV, dots, pm, pl, po);
  MHIMX_LAUNCH_CHECK();
  return tiles;
}

}  // namespace mhimx
