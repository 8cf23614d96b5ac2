// scorer_fused.hip — the ABMIL scorer + softmax-pool forward of one token matrix in ONE pass over it
// (modules/mhim.py:14-63 DAttention.forward restated: a = act(T Wa^T + ba), s = a.wc + bc, z = softmax(s) T).
//
// The three-kernel form (split-K scorer GEMM -> slab reduce -> row pass) reads T twice and pays three launch floors on the
// teacher -> select -> student critical path.  Here one workgroup owns 32 token rows:
//   1. the 32 x 512 fp32 rows land in LDS once (64.5 KB, rows padded by 16 B so the MFMA operand reads spread over banks);
//   2. U = T Wa^T (32 x 128, K = 512) on the matrix cores in the bf16x3 form (hi*hi + hi*lo + lo*hi, ~fp32 accuracy: the scores
//      feed the top-k); wave w owns output columns [32w, 32w+32) - v_mfma_f32_32x32x16_bf16, A from LDS, B (Wa, 256 KB, L2
//      resident for every workgroup) straight from global memory one k-step ahead;
//   3. epilogue in registers: + ba, pre-activation stored for the backward, act, * wc, 32-lane DPP row sums -> s;
//   4. the per-class projections cproj = T wp^T (pseudo score) and the log-sum-exp partial (m, l, sum_r e^{s_r - m} T[r,:])
//      come from the LDS copy of the rows - no second read of T.
// Partials are merged by pool_finalize_kernel (rows.hip) in a fixed order: deterministic.
#include <math.h>

#include "common.hpp"
#include "mca2_rows.hpp"
#include "prep_jobs.hpp"

namespace mhimx {

typedef float sf_f32x16 __attribute__((ext_vector_type(16)));
typedef float sf_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 sf_b8 __attribute__((ext_vector_type(8)));

#ifdef MHIMX_SF_PROF
__device__ unsigned long long sf_prof[16];
#define SF_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 7) sf_prof[i] = wall_clock64(); } while (0)
#else
#define SF_STAMP(i)
#endif

constexpr int SF_ROWS = 32, SF_E = 512, SF_A = 128, SF_LD = SF_E + 4, SF_THREADS = 256;
constexpr int SF_MAXC = 4;                  // class projections staged in LDS (more classes: the row-pass form)
constexpr size_t SF_SMEM = (size_t)(SF_ROWS * SF_LD + 4 * SF_ROWS + 2 * SF_ROWS + SF_MAXC * SF_E + 2 * SF_ROWS) * sizeof(float);

MHIMX_DEV void sf_split(const sf_f4& a, const sf_f4& b, sf_b8& hi, sf_b8& lo) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}

// sum over the 32 lanes that share (lane >> 5); valid in lanes 16..31 and 48..63
MHIMX_DEV float sf_sum32(float v) {
  v += dpp_mov<0xB1, 0xf>(0.f, v);
  v += dpp_mov<0x4E, 0xf>(0.f, v);
  v += dpp_mov<0x141, 0xf>(0.f, v);
  v += dpp_mov<0x140, 0xf>(0.f, v);          // every lane of a 16-lane row holds its row sum
  v += dpp_mov<0x142, 0xa>(0.f, v);          // rows 1,3 += rows 0,2
  return v;
}

template <bool FRAG>
__global__ __launch_bounds__(SF_THREADS, 2) void scorer_fused_kernel(
    const float* __restrict__ T, int64_t M, const float* __restrict__ wa, const float* __restrict__ wa_frag,
    const float* __restrict__ ba, int act,
    const float* __restrict__ wc, const float* __restrict__ bc, const float* __restrict__ wp, int C, float* __restrict__ u_pre,
    float* __restrict__ s_out, float* __restrict__ cproj, float* __restrict__ pm, float* __restrict__ pl, float* __restrict__ pz,
    int tiles, const int64_t* __restrict__ rows /* optional gather: token n is T[rows[n]] */,
    const uint8_t* __restrict__ excl /* optional, by source row: the row does not take part (score -inf) */,
    int n_main /* workgroups of the scorer itself; the blocks behind them are riders */, PrepJobs rider,
    int n_front /* round 5: the FIRST n_front blocks are the row tiles of a Merge forward (mca2_rows.hpp) - the student's scorer over the rows
                   that stay and the Merge over the rows to merge are independent until the tokens exist, and the Merge heads the longer chain */,
    M2RowsFwd mf, BagBatch bb /* common.hpp: blockIdx.z = the bag of an accumulation window this workgroup works for */) {
  extern __shared__ __attribute__((aligned(16))) float sf_sm[];
  if (blockIdx.z) {
    MHIMX_BAG(T); MHIMX_BAG(u_pre); MHIMX_BAG(s_out); MHIMX_BAG(cproj); MHIMX_BAG(pm); MHIMX_BAG(pl); MHIMX_BAG(pz); MHIMX_BAG(rows); MHIMX_BAG(excl);
    MHIMX_BAG(wa_frag);
    bag_move(mf, bb);
  }
  if ((int)blockIdx.x < n_front) {
    if ((int)blockIdx.x < mf.w.T)
      merge2_rows_fwd_body<16>((int)blockIdx.x, sf_sm, mf.X, mf.xrows, mf.R, mf.ln_w, mf.ln_b, mf.J, mf.drop_p, mf.seed0, mf.tick, mf.w);
    return;
  }
  const int bid = (int)blockIdx.x - n_front;
  if (bid >= n_main) {
    // parameter-only preparation jobs of the step riding in this launch's free workgroup slots (prep_jobs.hpp; the table is a by-value
    // kernel argument: only these blocks read it)
    prep_job_block(rider, bid - n_main, sf_sm);
    return;
  }
  float* Hs = sf_sm;                          // [32][516]
  float* sred = Hs + SF_ROWS * SF_LD;         // [4][32] per-wave partial scores
  float* srow = sred + 4 * SF_ROWS;           // [32] scores
  float* prow = srow + SF_ROWS;               // [32] e^{s - m}
  float* wps = prow + SF_ROWS;                // [C][512] predictor rows
  int64_t* ridx = reinterpret_cast<int64_t*>(wps + SF_MAXC * SF_E);   // [32] source rows of the tile (gathered form)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kg = lane >> 5;
  const int n_col = 32 * wave + r32;
  const float bias_c = bc ? bc[0] : 0.f;
  const float bn = ba ? ba[n_col] : 0.f, wn = wc[n_col];
  const float* bptr = wa + (int64_t)n_col * SF_E + 8 * kg;
  const float* fptr = wa_frag ? wa_frag + ((int64_t)wave * (SF_E / 16) * 64 + lane) * 8 : nullptr;   // + ks * 512 floats
  // (the same address as a uniform base + a 32-bit lane offset: the k-step term stays in scalar registers, one VGPR for all 32 k-steps)
  const unsigned foff = ((unsigned)wave * (SF_E / 16) * 64 + (unsigned)lane) * 8;
  const float* aptr = Hs + r32 * SF_LD + 8 * kg;
  if (wp)
    for (int i = tid; i < C * SF_E / 4; i += SF_THREADS) reinterpret_cast<sf_f4*>(wps)[i] = reinterpret_cast<const sf_f4*>(wp)[i];

  float m_run = -INFINITY, l_run = 0.f, z0 = 0.f, z1 = 0.f;
  for (int tile = bid; tile < tiles; tile += n_main) {
    const int64_t row0 = (int64_t)tile * SF_ROWS;
    SF_STAMP(0);
    sf_f32x16 acc, acc2, acc3;                  // one accumulator per bf16x3 term: an MFMA into the accumulator of the previous one waits out its latency
    if constexpr (FRAG) {
    // ---- 1 + 2. rows -> LDS in four 128-column chunks, the U product of chunk c on the matrix cores while chunks c+1, c+2 are on their way.
    // Before, the whole tile was loaded (an HBM burst of every workgroup of the launch at once, ~5.5 us), THEN multiplied (~6.5 us, of which
    // the MFMAs are 1.3: the k loop's back-edge made the compiler drain the fragment prefetch every fourth k-step).  Here every load is
    // inline asm with hand-counted waits (vector loads return in order): straight-line code, the weight fragments (prep kind 4: two
    // coalesced 16-byte loads per k-step) stay FOUR k-steps ahead, tile chunk c+2 is requested when chunk c starts.  (The k-step term of a
    // fragment address goes into the VGPR offset and the base stays ONE scalar pair; 32 scalar pointers get spilled to VGPR lanes, and a
    // v_readlane right in front of an asm VMEM instruction that uses the SGPR is a hazard the compiler cannot see: hence also the s_nop.)
    // Load j of a thread: columns [128 (j >> 2) + 32 (j & 3), + 32) of row tid >> 3 (8 lanes x 16 B = one 128-byte line per row).
    if (rows) {
      if (tid < SF_ROWS) { const int64_t n = row0 + tid; ridx[tid] = rows[n < M ? n : M - 1]; }
      __syncthreads();
    }
    const int lr = tid >> 3, ls = tid & 7;
    const int64_t ln = row0 + lr;
    const float* lsrc = T + (rows ? ridx[lr] : (ln < M ? ln : M - 1)) * SF_E + 4 * ls;      // (clamped row: every load is issued)
    float* ldst = Hs + lr * SF_LD + 4 * ls;
    const unsigned fo = foff * 4;                                                           // byte offset of this lane in a fragment block
    sf_f4 av[16], bh_[4], bl_[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
#define SF_LOAD_A(c)                                                                                                                          \
    asm volatile("global_load_dwordx4 %0, %4, off offset:%5\n\tglobal_load_dwordx4 %1, %4, off offset:%6\n\t"                                  \
                 "global_load_dwordx4 %2, %4, off offset:%7\n\tglobal_load_dwordx4 %3, %4, off offset:%8"                                      \
                 : "=&v"(av[4 * (c)]), "=&v"(av[4 * (c) + 1]), "=&v"(av[4 * (c) + 2]), "=&v"(av[4 * (c) + 3])                                  \
                 : "v"(lsrc), "i"(512 * (c)), "i"(512 * (c) + 128), "i"(512 * (c) + 256), "i"(512 * (c) + 384) : "memory")
#define SF_LOAD_B(ks)                                                                                                                         \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"                                      \
                 : "=&v"(bh_[(ks) & 3]), "=&v"(bl_[(ks) & 3]) : "v"(fo + 2048u * (ks)), "s"(wa_frag) : "memory")
#define SF_WAIT_B(n, ks) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(bh_[(ks) & 3]), "+v"(bl_[(ks) & 3]) : : "memory")
#define SF_WAIT_A(n, c)                                                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(av[4 * (c)]), "+v"(av[4 * (c) + 1]), "+v"(av[4 * (c) + 2]), "+v"(av[4 * (c) + 3]) : : "memory")
    // one k-step: wait until its fragments are in (n = loads issued after them), copy them out of the ring, refill the slot, multiply
#define SF_KSTEP(ks, n, refill)                                                                                                               \
    {                                                                                                                                        \
      SF_WAIT_B(n, ks);                                                                                                                      \
      const sf_b8 bh = __builtin_bit_cast(sf_b8, bh_[(ks) & 3]), bl = __builtin_bit_cast(sf_b8, bl_[(ks) & 3]);                             \
      if (refill) SF_LOAD_B((ks) + 4 < SF_E / 16 ? (ks) + 4 : (ks));                                                                         \
      const sf_f4 a0 = *reinterpret_cast<const sf_f4*>(aptr + 16 * (ks)), a1 = *reinterpret_cast<const sf_f4*>(aptr + 16 * (ks) + 4);       \
      sf_b8 ah, al;                                                                                                                          \
      sf_split(a0, a1, ah, al);                                                                                                              \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);                                                                 \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);                                                                   \
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc3, 0, 0, 0);                                                                 \
    }
#define SF_STORE_A(c)                                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                                            \
        *reinterpret_cast<sf_f4*>(ldst + 128 * (c) + 32 * q) = ln < M ? av[4 * (c) + q] : sf_f4{0.f, 0.f, 0.f, 0.f};
    // issue order: A0 A1 B0..B3 | chunk 0: A2, B4..B11 | chunk 1: A3, B12..B19 | chunk 2: B20..B27 | chunk 3: B28..B31
    SF_LOAD_A(0);
    SF_LOAD_A(1);
    SF_LOAD_B(0); SF_LOAD_B(1); SF_LOAD_B(2); SF_LOAD_B(3);
    SF_WAIT_A(12, 0);                            // behind chunk 0: A1 (4) + B0..B3 (8)
    SF_STORE_A(0);
    __syncthreads();
    SF_STAMP(1);
    SF_LOAD_A(2);
    // k-steps 0..3: behind B(ks) are B(ks+1..3) + A2 (4) + the refills B(4..ks+3) = 6 + 4; k-steps 4..7: the three younger fragment pairs
    SF_KSTEP(0, 10, true) SF_KSTEP(1, 10, true) SF_KSTEP(2, 10, true) SF_KSTEP(3, 10, true)
    SF_KSTEP(4, 6, true) SF_KSTEP(5, 6, true) SF_KSTEP(6, 6, true) SF_KSTEP(7, 6, true)
    SF_WAIT_A(8, 1);                             // (A1 is older than fragments already waited for; 8 = the four pairs in flight)
    SF_STORE_A(1);
    __syncthreads();
    SF_LOAD_A(3);
    SF_KSTEP(8, 10, true) SF_KSTEP(9, 10, true) SF_KSTEP(10, 10, true) SF_KSTEP(11, 10, true)
    SF_KSTEP(12, 6, true) SF_KSTEP(13, 6, true) SF_KSTEP(14, 6, true) SF_KSTEP(15, 6, true)
    SF_WAIT_A(8, 2);
    SF_STORE_A(2);
    __syncthreads();
    SF_KSTEP(16, 6, true) SF_KSTEP(17, 6, true) SF_KSTEP(18, 6, true) SF_KSTEP(19, 6, true)
    SF_KSTEP(20, 6, true) SF_KSTEP(21, 6, true) SF_KSTEP(22, 6, true) SF_KSTEP(23, 6, true)
    SF_WAIT_A(8, 3);
    SF_STORE_A(3);
    __syncthreads();
    SF_KSTEP(24, 6, true) SF_KSTEP(25, 6, true) SF_KSTEP(26, 6, true) SF_KSTEP(27, 6, true)
    SF_KSTEP(28, 6, false) SF_KSTEP(29, 4, false) SF_KSTEP(30, 2, false) SF_KSTEP(31, 0, false)
#undef SF_LOAD_A
#undef SF_LOAD_B
#undef SF_WAIT_A
#undef SF_WAIT_B
#undef SF_KSTEP
#undef SF_STORE_A
    } else {
    // ---- 1. rows -> LDS (gathered form: the tile's 32 row indices first, ONE coalesced load instead of 16 dependent ones per thread)
    if (rows) {
      if (tid < SF_ROWS) { const int64_t n = row0 + tid; ridx[tid] = rows[n < M ? n : M - 1]; }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = tid + SF_THREADS * i, r = f >> 7, c4 = f & 127;
      const int64_t n = row0 + r;
      const int64_t nc = n < M ? n : M - 1;                                                // clamped: all 16 loads in flight
      sf_f4 v = reinterpret_cast<const sf_f4*>(T + (rows ? ridx[r] : nc) * SF_E)[c4];
      if (n >= M) v = sf_f4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<sf_f4*>(Hs + r * SF_LD + 4 * c4) = v;
    }
    __syncthreads();
    SF_STAMP(1);
    // ---- 2. U tile on the matrix cores
    // one accumulator per bf16x3 term: an MFMA into the accumulator of the previous one waits out its full latency
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    {
      sf_f4 b0 = *reinterpret_cast<const sf_f4*>(bptr), b1 = *reinterpret_cast<const sf_f4*>(bptr + 4);
#pragma unroll 4
      for (int ks = 0; ks < SF_E / 16; ++ks) {
        const int kn = ks + 1 < SF_E / 16 ? ks + 1 : ks;
        const sf_f4 nb0 = *reinterpret_cast<const sf_f4*>(bptr + 16 * kn), nb1 = *reinterpret_cast<const sf_f4*>(bptr + 16 * kn + 4);
        const sf_f4 a0 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks), a1 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks + 4);
        sf_b8 ah, al, bh, bl;
        sf_split(a0, a1, ah, al);
        sf_split(b0, b1, bh, bl);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc3, 0, 0, 0);
        b0 = nb0;
        b1 = nb1;
      }
    }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc2[i] + acc3[i];
    SF_STAMP(2);
    // ---- 3. epilogue: acc[i] = U[row = 8*(i>>2) + 4*kg + (i&3)][n_col]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * (i >> 2) + 4 * kg + (i & 3);
      const int64_t n = row0 + row;
      const float u = acc[i] + bn;
      if (n < M && u_pre) u_pre[n * SF_A + n_col] = u;
      const float v = sf_sum32(wn * act_fwd(u, act));
      if (r32 == 31) sred[wave * SF_ROWS + row] = v;
    }
    __syncthreads();
    if (tid < SF_ROWS) {
      const int64_t n = row0 + tid;
      float s = (sred[tid] + sred[SF_ROWS + tid]) + (sred[2 * SF_ROWS + tid] + sred[3 * SF_ROWS + tid]) + bias_c;
      if (n >= M || (excl && excl[rows ? ridx[tid] : n])) s = -INFINITY;
      if (n < M) s_out[n] = s;
      srow[tid] = s;
    }
    __syncthreads();
    SF_STAMP(3);
    // ---- 4. log-sum-exp partial over the tile (running over this workgroup's tiles)
    float mt = -INFINITY;
#pragma unroll
    for (int q = 0; q < SF_ROWS / 4; ++q) {
      const sf_f4 v = reinterpret_cast<const sf_f4*>(srow)[q];
      mt = fmaxf(fmaxf(mt, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    const float m_new = fmaxf(m_run, mt);
    const float scale = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
    if (tid < SF_ROWS) prow[tid] = srow[tid] == -INFINITY ? 0.f : __expf(srow[tid] - m_new);   // (a tile of excluded rows only: m_new = -inf)
    __syncthreads();
    float lsum = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < SF_ROWS / 4; ++q) {
      const sf_f4 p = reinterpret_cast<const sf_f4*>(prow)[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* hr = Hs + (4 * q + j) * SF_LD;
        lsum += p[j];
        a0 += p[j] * hr[tid];
        a1 += p[j] * hr[tid + SF_THREADS];
      }
    }
    l_run = l_run * scale + lsum;
    z0 = z0 * scale + a0;
    z1 = z1 * scale + a1;
    m_run = m_new;
    SF_STAMP(4);
    // ---- class projections of the rows (pseudo score): 8 lanes per row; lane `seg` takes the 16-byte groups seg, seg+8, ...
    // (neighbouring lanes read neighbouring groups: no bank conflicts)
    if (wp) {
      const int r = tid >> 3, seg = tid & 7;
      const int64_t n = row0 + r;
      const sf_f4* hr = reinterpret_cast<const sf_f4*>(Hs + r * SF_LD) + seg;
      for (int c = 0; c < C; ++c) {
        const sf_f4* w = reinterpret_cast<const sf_f4*>(wps + c * SF_E) + seg;
        float d = 0.f;
#pragma unroll 8
        for (int q = 0; q < 16; ++q) {
          const sf_f4 hv = hr[8 * q], ww = w[8 * q];
          d += hv[0] * ww[0] + hv[1] * ww[1] + hv[2] * ww[2] + hv[3] * ww[3];
        }
        d += dpp_mov<0xB1, 0xf>(0.f, d);                      // lanes ^1
        d += dpp_mov<0x4E, 0xf>(0.f, d);                      // lanes ^2
        d += dpp_mov<0x141, 0xf>(0.f, d);                     // row_half_mirror: the other quad of the 8
        if (seg == 0 && n < M) cproj[n * C + c] = d;
      }
    }
    __syncthreads();                           // the next tile overwrites Hs / srow
    SF_STAMP(5);
  }
  if (tid == 0) { pm[bid] = m_run; pl[bid] = l_run; }
  pz[(int64_t)bid * SF_E + tid] = z0;
  pz[(int64_t)bid * SF_E + tid + SF_THREADS] = z1;
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward of the same block in one pass over the rows (replaces the row pass + the dT GEMM of rows.hip):
//   attn_n = e^{s_n - max} / L,  ds_n = attn_n (T_n.g_z - z.g_z),  du[n,a] = ds_n wc[a] act'(u_pre[n,a])   (stored: d_wa = du^T T)
//   dT[n,:] = du[n,:] Wa + attn_n g_z                              (32 x 512 tile, K = 128, on the matrix cores)
//   per-workgroup partials of d_wc[a] = sum_n ds_n act(u_pre[n,a]) and d_bc = sum_n ds_n
// T is read once, straight into registers for the row dots (8 lanes per row); du goes to LDS as the A operand; Wa^T comes as
// the prep-time fragment image (or is split on the fly).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int SB_LD = SF_A + 4;
constexpr size_t SB_SMEM = (size_t)(SF_ROWS * SB_LD + SF_E + 3 * SF_ROWS + 2 * SF_A + 8 + 2 * SF_ROWS) * sizeof(float);
// Round 6 - the rows' share of the projection's dPRE image written HERE (mhimx_pool_grad.img; wgrad.hip's operand format) instead of dT going
// to memory in fp32 for rows_dpre_image to read back: dPRE[n, e] = dT[n, e] * dact16[row(n), e] for the first img_rows tokens of the list (the
// bag rows that stay; the merged tokens behind them keep their fp32 gradient rows: the Merge backward reads them), every other row of the
// launch's 32-row tiles a zero row.  Tile t IS k-step t of the image: one contiguous 16 KiB block [k-octet 4][hi|lo][column slot 128][8 bf16]
// per 128 columns.  The accumulator holds rows 8 o + 4 (lane >> 5) + m of a column: v_permlane32_swap pairs the half-waves so that a lane
// ends with the 8 consecutive rows of one k-octet, which it splits into bf16 hi / lo once and stores as two 16-byte units.  The tile's
// d out / d pre rows come through LDS (32 KiB, direct DMA at the top of the tile); the column sums of the tile (the bias gradient's
// partials) leave as one row of img_part.
constexpr int SB_DLD = 1040;                         // bytes per staged d out / d pre row (1024 + 16: the two half-waves read rows 4 apart)
constexpr size_t SB_SMEM_IMG = SB_SMEM + (size_t)SF_ROWS * SB_DLD;
// a's lanes 32..63 <-> b's lanes 0..31 (wgrad.hip's wg_swap: inline asm - the builtin of this compiler drops its second result)
MHIMX_DEV void sb_swap(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

__global__ __launch_bounds__(SF_THREADS, 2) void scorer_fused_bwd_kernel(
    const float* __restrict__ T, int64_t M, const float* __restrict__ u_pre, const float* __restrict__ s_in,
    const float* __restrict__ stats, const float* __restrict__ g_z, const float* __restrict__ z, const float* __restrict__ wc, int act,
    const float* __restrict__ wat, const float* __restrict__ wat_frag, float* __restrict__ du, float* __restrict__ dT,
    float* __restrict__ dwc_part, float* __restrict__ dbc_part, int tiles,
    const int64_t* __restrict__ rows /* optional: token n is T[rows[n]] and its gradient goes to dT[rows[n]] */,
    int n_main /* workgroups of the backward itself; the blocks behind them: a Merge backward's first stage, riding */, Merge2Side pre,
    int64_t gate_row0 /* >= 0: the rows gate_row0 .. of dT are that stage's dz - stored write-through and announced on pre.w.gate[1] */,
    char* __restrict__ img /* optional: the dPRE image takes the first img_rows tokens' gradient (see SB_SMEM_IMG) */,
    const _Float16* __restrict__ dact, float* __restrict__ img_part, int64_t img_rows, BagBatch bb) {
  extern __shared__ __attribute__((aligned(16))) float sb_sm[];
  if (blockIdx.z) {
    MHIMX_BAG(T); MHIMX_BAG(u_pre); MHIMX_BAG(s_in); MHIMX_BAG(stats); MHIMX_BAG(g_z); MHIMX_BAG(z); MHIMX_BAG(du); MHIMX_BAG(dT); MHIMX_BAG(dwc_part);
    MHIMX_BAG(dbc_part); MHIMX_BAG(rows); MHIMX_BAG(wat); MHIMX_BAG(wat_frag); MHIMX_BAG(img); MHIMX_BAG(dact); MHIMX_BAG(img_part);
    bag_move(pre, bb);
  }
  if ((int)blockIdx.x >= n_main) {
    // (the LAST blocks of the grid: every producer of dz is resident or done when one of these starts; they request their weights, then
    // wait for pre.k announced rows)
    merge2_bwd_pre_body((int)blockIdx.x - n_main, sb_sm, pre.dz, pre.wo_t, pre.wkv, pre.k, pre.drop_p, pre.oseed, pre.tick, pre.d_bo, pre.accumulate,
                        pre.w, pre.rep, pre.w.gate + 1, (unsigned)pre.k);
    return;
  }
  const bool gated = gate_row0 >= 0;
  float* Ds = sb_sm;                           // [32][132] du tile (A operand)
  float* gzs = Ds + SF_ROWS * SB_LD;           // [512]
  float* an_s = gzs + SF_E;                    // [32] attn
  float* gs_s = an_s + SF_ROWS;                // [32] ds
  float* red = gs_s + SF_ROWS;                 // [32] scratch: c0 partials (4 used)
  float* dwc_s = red + SF_ROWS;                // [2][128]
  int64_t* ridx = reinterpret_cast<int64_t*>(dwc_s + 2 * SF_A + 8);   // [32] source / destination rows of the tile (gathered form)
  char* dact_s = reinterpret_cast<char*>(sb_sm) + SB_SMEM;            // [32][SB_DLD] (img only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kg = lane >> 5;
  // g_z -> LDS, c0 = z . g_z
  float c0p = 0.f;
  for (int e = tid; e < SF_E; e += SF_THREADS) {
    const float g = g_z[e];
    gzs[e] = g;
    c0p += g * z[e];
  }
  c0p = wave_sum(c0p);
  if (lane == 0) red[wave] = c0p;
  __syncthreads();
  const float c0 = (red[0] + red[1]) + (red[2] + red[3]);
  const float mx = stats[0], invL = 1.f / stats[1];
  const int a_col = tid & (SF_A - 1), half = tid >> 7;           // du / d_wc column of this thread
  const float wca = wc[a_col];
  float dwc_run = 0.f, dbc_run = 0.f;

  for (int tile = blockIdx.x; tile < tiles; tile += n_main) {
    const int64_t row0 = (int64_t)tile * SF_ROWS;
    if (rows) {
      if (tid < SF_ROWS) { const int64_t n = row0 + tid; ridx[tid] = rows[n < M ? n : M - 1]; }
      __syncthreads();
    }
    if (img) {
      // the tile's d out / d pre rows -> LDS by direct DMA (one 1 KiB row per wave instruction), consumed by the epilogue
#pragma unroll
      for (int q = 0; q < SF_ROWS / 4; ++q) {
        const int r = wave + 4 * q;
        const int64_t n = row0 + r;
        const int64_t src = rows ? ridx[r] : (n < M ? n : M - 1);
        __builtin_amdgcn_global_load_lds((gptr_f)(reinterpret_cast<const char*>(dact + src * SF_E) + lane * 16), (lptr_f)(dact_s + r * SB_DLD), 16, 0, 0);
      }
    }
    // ---- 1. row dots T_n . g_z: 8 lanes per row, lane `seg` takes the 16-byte groups seg, seg+8, ...
    float up[SF_ROWS / 2];
    {
      const int r = tid >> 3, seg = tid & 7;
      const int64_t n = row0 + r;
      const int64_t nc = n < M ? n : M - 1;
      const sf_f4* tr = reinterpret_cast<const sf_f4*>(T + (rows ? ridx[r] : nc) * SF_E) + seg;
      const sf_f4* gr = reinterpret_cast<const sf_f4*>(gzs) + seg;
      sf_f4 tv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) tv[q] = tr[8 * q];
      // the scorer pre-activations of step 2, requested now: their latency passes under the row dots
#pragma unroll
      for (int k = 0; k < SF_ROWS / 2; ++k) {
        const int64_t nu = row0 + half + 2 * k;
        up[k] = u_pre[(nu < M ? nu : M - 1) * SF_A + a_col];
      }
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const sf_f4 g = gr[8 * q];
        d += tv[q][0] * g[0] + tv[q][1] * g[1] + tv[q][2] * g[2] + tv[q][3] * g[3];
      }
      d += dpp_mov<0xB1, 0xf>(0.f, d);
      d += dpp_mov<0x4E, 0xf>(0.f, d);
      d += dpp_mov<0x141, 0xf>(0.f, d);
      if (seg == 0) {
        const float an = n < M ? __expf(s_in[n] - mx) * invL : 0.f;
        an_s[r] = an;
        gs_s[r] = an * (d - c0);
      }
    }
    __syncthreads();
    // ---- 2. du tile (global + LDS), d_wc / d_bc partials
#pragma unroll
    for (int k = 0; k < SF_ROWS / 2; ++k) {
      const int r = half + 2 * k;
      const int64_t n = row0 + r;
      float d = 0.f;
      if (n < M) {
        const float u = up[k];
        float ya, ga;
        act_fwd_grad(u, act, ya, ga);
        const float ds = gs_s[r];
        d = ds * wca * ga;
        du[n * SF_A + a_col] = d;
        dwc_run += ds * ya;
      }
      Ds[r * SB_LD + a_col] = d;
    }
    if (tid == 0) {
      float b = 0.f;
      for (int r = 0; r < SF_ROWS; ++r) b += gs_s[r];
      dbc_run += b;
    }
    __syncthreads();
    // ---- 3. dT tile = du Wa (+ attn g_z): wave w owns columns [128 w, 128 w + 128)
    sf_f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    const float* aptr = Ds + r32 * SB_LD + 8 * kg;
    if (wat_frag) {
      const float* fptr = wat_frag + ((int64_t)(4 * wave) * (SF_A / 16) * 64 + lane) * 8;     // + (nt * 8 + ks) * 512 floats
      sf_f4 bh_[4], bl_[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bh_[nt] = *reinterpret_cast<const sf_f4*>(fptr + (nt * 8) * 512);
        bl_[nt] = *reinterpret_cast<const sf_f4*>(fptr + (nt * 8) * 512 + 4);
      }
#pragma unroll 1
      for (int ks = 0; ks < SF_A / 16; ++ks) {
        const sf_f4 a0 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks), a1 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks + 4);
        sf_b8 ah, al;
        sf_split(a0, a1, ah, al);
        sf_f4 nh[4], nl[4];
        const int kn = ks + 1 < SF_A / 16 ? ks + 1 : ks;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          nh[nt] = *reinterpret_cast<const sf_f4*>(fptr + (nt * 8 + kn) * 512);
          nl[nt] = *reinterpret_cast<const sf_f4*>(fptr + (nt * 8 + kn) * 512 + 4);
        }
        // term-major: the three MFMAs into acc[nt] are three others apart (no dependent back-to-back issue)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(sf_b8, bh_[nt]), acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(sf_b8, bl_[nt]), acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(sf_b8, bh_[nt]), acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          bh_[nt] = nh[nt];
          bl_[nt] = nl[nt];
        }
      }
    } else {
#pragma unroll 1
      for (int ks = 0; ks < SF_A / 16; ++ks) {
        const sf_f4 a0 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks), a1 = *reinterpret_cast<const sf_f4*>(aptr + 16 * ks + 4);
        sf_b8 ah, al;
        sf_split(a0, a1, ah, al);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float* bp = wat + (int64_t)(128 * wave + 32 * nt + r32) * SF_A + 16 * ks + 8 * kg;
          sf_b8 bh, bl;
          sf_split(*reinterpret_cast<const sf_f4*>(bp), *reinterpret_cast<const sf_f4*>(bp + 4), bh, bl);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
        }
      }
    }
    if (img) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMA rows have landed ...
      __syncthreads();                                           // ... and everyone's
      const bool has_tail = row0 + SF_ROWS > img_rows;           // (the list's last tiles: the merged tokens' rows keep their fp32 gradient)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int e = 128 * wave + 32 * nt + r32;
        const float ge = gzs[e];
        float val[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 8 * (i >> 2) + 4 * kg + (i & 3);
          const int64_t n = row0 + row;
          const float v = acc[nt][i] + an_s[row] * ge;
          if (has_tail && n >= img_rows && n < M) {
            const int64_t dr = rows ? ridx[row] : n;
            if (gated && dr >= gate_row0) __hip_atomic_store(dT + dr * SF_E + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dT[dr * SF_E + e] = v;
          }
          const float d = (float)*reinterpret_cast<const _Float16*>(dact_s + row * SB_DLD + 2 * e);
          val[i] = n < img_rows ? v * d : 0.f;
        }
        float cs = 0.f;
        char* unit = img + ((int64_t)tile * (SF_E / 128) + wave) * 16384 + ((8 * nt + (r32 >> 2)) * 16 + (r32 & 3) * 512);
#pragma unroll
        for (int o = 0; o < 4; o += 2) {
          float kv[8];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float a = val[4 * o + m], b = val[4 * (o + 1) + m];
            sb_swap(a, b);                    // lanes 0..31: rows 8 o + m, 8 o + 4 + m;  lanes 32..63: rows 8 (o + 1) + m, 8 (o + 1) + 4 + m
            kv[m] = a;
            kv[4 + m] = b;
          }
          sf_b8 hi, lo;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            cs += kv[q];
            const __bf16 h = (__bf16)kv[q];
            hi[q] = h;
            lo[q] = (__bf16)(kv[q] - (float)h);
          }
          char* u = unit + (o + kg) * 4096;                        // k-octet o + kg: 2 planes x 128 slots x 16 B
          *reinterpret_cast<sf_f4*>(u) = __builtin_bit_cast(sf_f4, hi);
          *reinterpret_cast<sf_f4*>(u + 2048) = __builtin_bit_cast(sf_f4, lo);
        }
        if (img_part) {
          const float tot = cs + __shfl_xor(cs, 32);             // the other half-wave holds the other two k-octets of the column
          if (kg == 0) img_part[(int64_t)tile * SF_E + e] = tot;
        }
      }
    } else {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int e = 128 * wave + 32 * nt + r32;
      const float ge = gzs[e];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i >> 2) + 4 * kg + (i & 3);
        const int64_t n = row0 + row;
        if (n < M) {
          const int64_t dr = rows ? ridx[row] : n;
          const float v = acc[nt][i] + an_s[row] * ge;
          // (a row of the riding stage's dz: past the caches - its readers sit on other XCDs of this same launch)
          if (gated && dr >= gate_row0) __hip_atomic_store(dT + dr * SF_E + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else dT[dr * SF_E + e] = v;
        }
      }
    }
    }
    if (gated) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores are acknowledged before the barrier
    __syncthreads();                           // Ds / an_s / gs_s are rewritten by the next tile
    if (gated && wave == 0) {
      // the tile's rows of dz, counted once per row: their number is added to the gate (the riders wait for pre.k rows in all)
      const int64_t n = row0 + lane;
      const bool mine = lane < SF_ROWS && n < M && (rows ? ridx[lane < SF_ROWS ? lane : 0] : n) >= gate_row0;
      const unsigned cnt = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
      if (lane == 0 && cnt) __hip_atomic_fetch_add(pre.w.gate + 1, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  dwc_s[half * SF_A + a_col] = dwc_run;
  __syncthreads();
  if (tid < SF_A) dwc_part[(int64_t)blockIdx.x * SF_A + tid] = dwc_s[tid] + dwc_s[SF_A + tid];
  if (tid == 0) dbc_part[blockIdx.x] = dbc_run;
}

#ifdef MHIMX_SF_PROF
extern "C" int mhimx_sf_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sf_prof), 16 * 8); }
#endif

bool scorer_fused_ok(int64_t E, int64_t A, int gated, int prec, const float* T, const float* wa, const float* wp, int64_t C) {
  return E == SF_E && A == SF_A && !gated && prec != MHIMX_PREC_F32 && aligned16(T) && aligned16(wa) && (!wp || (aligned16(wp) && C <= SF_MAXC));
}

// one launch per token segment; returns the number of partials written to pm/pl/pz (<= max_parts), < 0 on error
int scorer_fused_fwd(hipStream_t st, const float* T, int64_t M, const float* wa, const float* wa_frag, const float* ba, int act, const float* wc,
                     const float* bc, const float* wp, int C, float* u_pre, float* s_out, float* cproj, float* pm, float* pl,
                     float* pz, int max_parts, const int64_t* rows, const uint8_t* excl, const mhimx_prep_job* ride_jobs, int n_ride_jobs,
                     const void* merge_rows /* optional M2RowsFwd: its row tiles ride at the front of the launch */) {
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)scorer_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_SMEM)));
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)scorer_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_SMEM)));
  const int tiles = (int)cdiv(M, SF_ROWS);
  const int grid = tiles < max_parts ? tiles : max_parts;
  static_assert(SF_SMEM >= PREP_LDS_FLOATS * sizeof(float), "the riders' LDS is the scorer's");
  static_assert(SF_SMEM >= m2_fwd_smem(16), "the Merge row tiles' LDS is the scorer's");
  static_assert(SF_THREADS == 256 && SF_THREADS == M2_THREADS, "the preparation jobs and the Merge row tiles are written for 256 threads");
  PrepJobs rider = {};
  int ride_blocks = 0;
  MHIMX_CHECK_ARG(!(ride_jobs && n_ride_jobs > 0 && cur_batch().n > 0), "scorer: preparation jobs do not ride in a bag-batched launch");
  if (ride_jobs && n_ride_jobs > 0) {
    ride_blocks = prep_jobs_fill(ride_jobs, n_ride_jobs, &rider);
    if (ride_blocks < 0) return ride_blocks;
  }
  M2RowsFwd mf = {};
  int n_front = 0;
  if (merge_rows) {
    mf = *reinterpret_cast<const M2RowsFwd*>(merge_rows);
    n_front = mf.w.T;
  }
  if (wa_frag)
    hipLaunchKernelGGL(scorer_fused_kernel<true>, bgrid(n_front + grid + ride_blocks), dim3(SF_THREADS), SF_SMEM, st, T, M, wa, wa_frag, ba, act, wc, bc, wp, C, u_pre,
                       s_out, cproj, pm, pl, pz, tiles, rows, excl, grid, rider, n_front, mf, cur_batch());
  else
    hipLaunchKernelGGL(scorer_fused_kernel<false>, bgrid(n_front + grid + ride_blocks), dim3(SF_THREADS), SF_SMEM, st, T, M, wa, wa_frag, ba, act, wc, bc, wp, C, u_pre,
                       s_out, cproj, pm, pl, pz, tiles, rows, excl, grid, rider, n_front, mf, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return grid;
}

// returns the number of d_wc / d_bc partial rows written (<= max_parts), < 0 on error
int scorer_fused_bwd(hipStream_t st, const float* T, int64_t M, const float* u_pre, const float* s_in, const float* stats,
                     const float* g_z, const float* z, const float* wc, int act, const float* wa_t, const float* wa_t_frag, float* du,
                     float* dT, float* dwc_part, float* dbc_part, int max_parts, const int64_t* rows, const void* pre_side, int64_t gate_row0,
                     void* img, const void* img_dact, float* img_part, int64_t img_rows) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)scorer_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB_SMEM_IMG)));
  MHIMX_CHECK_ARG(!img || (img_dact && aligned16(img) && aligned16(img_dact) && img_rows >= 0 && img_rows <= M),
                  "scorer backward: the dPRE image needs the d out / d pre rows, 16-byte aligned buffers and img_rows <= M");
  const int tiles = (int)cdiv(M, SF_ROWS);
  const int grid = tiles < max_parts ? tiles : max_parts;
  static_assert(SB_SMEM >= M2_BWD_PRE_LDS * sizeof(float), "the riding Merge stage's LDS is the backward's");
  static_assert(SF_THREADS == M2_THREADS, "the riding Merge stage is written for 256 threads");
  Merge2Side pre = {};
  int ride = 0;
  if (pre_side) {
    pre = *reinterpret_cast<const Merge2Side*>(pre_side);
    ride = M2_BWD_PRE_BLOCKS;
  } else {
    gate_row0 = -1;
  }
  hipLaunchKernelGGL(scorer_fused_bwd_kernel, bgrid(grid + ride), dim3(SF_THREADS), img ? SB_SMEM_IMG : SB_SMEM, st, T, M, u_pre, s_in, stats, g_z, z, wc, act, wa_t,
                     wa_t_frag, du, dT, dwc_part, dbc_part, tiles, rows, grid, pre, gate_row0, (char*)img, (const _Float16*)img_dact, img_part, img_rows, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return grid;
}

}  // namespace mhimx
