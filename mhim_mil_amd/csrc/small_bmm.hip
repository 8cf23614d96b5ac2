// small_bmm.hip — batches of SMALL square-ish products in one launch:  C_b = ident * I + alpha * op(A_b, B_b)  (+ C_b)
//
// The Nystrom pseudo-inverse (nystrom_attention.py:12-27, six iterations of four 256 x 256 x 256 products per head, forward and
// backward: ~200 launches per TransMIL train step) ran on the generic 128 x 128-tile kernels: 8 heads x 4 tiles = 32 workgroups on 256
// CUs, 18-57 us per launch.  Here a workgroup owns a 64 x 64 tile (8 heads x 16 tiles = 128 workgroups), 4 waves x one
// v_mfma_f32_32x32x16_bf16 block each, 3-term bf16 (hi*hi + hi*lo + lo*hi, ~2^-16), K in steps of 32 with the next step's global loads
// in flight; both operand tiles sit in LDS k-contiguous ([row][32 k], pitch 36), whatever their layout in memory - the operand that
// is not k-contiguous in memory is transposed by its LDS stores - so all three modes share the fragment path.  The affine epilogue
// (ident * I + alpha * product) fuses the "a I - M" steps of the iteration.   mode 0: A[M,K] B[N,K]^T, 1: A[M,K] B[K,N], 2: A[K,M]^T B[K,N].
#include "common.hpp"

namespace mhimx {

typedef float sb_f4 __attribute__((ext_vector_type(4)));
typedef float sb_f16 __attribute__((ext_vector_type(16)));
typedef __bf16 sb_b8 __attribute__((ext_vector_type(8)));
constexpr int SB_T = 64, SB_K = 32, SB_PITCH = 36, SB_THREADS = 256;

struct SmallBmm {
  const float* A; const float* B; float* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int M, N, K;
  float alpha, ident;
  int accumulate;
  float* C2;              // optional second output of the same product: C2 = ident2 * I + alpha2 * op(A, B)  (whole-panel kernel only)
  float alpha2, ident2;
};

MHIMX_DEV void sb_split(const sb_f4& a, const sb_f4& b, sb_b8& hi, sb_b8& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 h = (__bf16)v[q];
    hi[q] = h;
    lo[q] = (__bf16)(v[q] - (float)h);
  }
}

// TA / TB: the operand is stored [K, rows] in memory (rows contiguous) and is transposed on its way into LDS
template <bool TA, bool TB>
__global__ __launch_bounds__(SB_THREADS) void small_bmm_kernel(SmallBmm g) {
  __shared__ __attribute__((aligned(16))) float As[SB_T * SB_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[SB_T * SB_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * SB_T, n0 = (int64_t)blockIdx.x * SB_T;
  const float* A = g.A + (int64_t)blockIdx.z * g.sA;
  const float* B = g.B + (int64_t)blockIdx.z * g.sB;
  float* C = g.C + (int64_t)blockIdx.z * g.sC;

  sb_f4 ra[2], rb[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j;
      if (TA) ra[j] = *reinterpret_cast<const sb_f4*>(A + (int64_t)(k0 + (f >> 4)) * g.lda + m0 + (f & 15) * 4);
      else ra[j] = *reinterpret_cast<const sb_f4*>(A + (m0 + (f >> 3)) * g.lda + k0 + (f & 7) * 4);
      if (TB) rb[j] = *reinterpret_cast<const sb_f4*>(B + (int64_t)(k0 + (f >> 4)) * g.ldb + n0 + (f & 15) * 4);
      else rb[j] = *reinterpret_cast<const sb_f4*>(B + (n0 + (f >> 3)) * g.ldb + k0 + (f & 7) * 4);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j;
      if (TA) {
#pragma unroll
        for (int c = 0; c < 4; ++c) As[((f & 15) * 4 + c) * SB_PITCH + (f >> 4)] = ra[j][c];
      } else {
        *reinterpret_cast<sb_f4*>(As + (f >> 3) * SB_PITCH + (f & 7) * 4) = ra[j];
      }
      if (TB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[((f & 15) * 4 + c) * SB_PITCH + (f >> 4)] = rb[j][c];
      } else {
        *reinterpret_cast<sb_f4*>(Bs + (f >> 3) * SB_PITCH + (f & 7) * 4) = rb[j];
      }
    }
  };

  sb_f16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = As + (wm * 32 + r) * SB_PITCH + 8 * kh;
  const float* bp = Bs + (wn * 32 + r) * SB_PITCH + 8 * kh;
  load(0);
  for (int k0 = 0; k0 < g.K; k0 += SB_K) {
    __syncthreads();                                       // the previous step's fragment reads are over
    store();
    __syncthreads();
    if (k0 + SB_K < g.K) load(k0 + SB_K);                  // in flight under the MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sb_b8 ah, al, bh, bl;
      sb_split(*reinterpret_cast<const sb_f4*>(ap + 16 * s), *reinterpret_cast<const sb_f4*>(ap + 16 * s + 4), ah, al);
      sb_split(*reinterpret_cast<const sb_f4*>(bp + 16 * s), *reinterpret_cast<const sb_f4*>(bp + 16 * s + 4), bh, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  const int64_t n = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    float v = g.alpha * acc[e];
    if (m == n) v += g.ident;
    float* p = C + m * g.ldc + n;
    if (g.accumulate) v += *p;
    *p = v;
  }
}

// The same product with the WHOLE K panel of both operands in LDS (K <= 256: the 256 x 256 x 256 products of the pseudo-inverse):
// the eight k-steps of small_bmm_kernel are eight global-load -> barrier -> store -> barrier rounds; here the panel is staged behind
// ONE barrier and the 8 x 6 MFMAs per wave run with no barrier between (8.5 -> 7.4 us per 8-head product; forcing all 32 loads of a
// thread in flight before the first store, or two accumulator chains, measured slower: 8.8 us).
constexpr int SBF_KMAX = 256, SBF_PITCH = SBF_KMAX + 4;
template <bool TA, bool TB>
MHIMX_DEV void small_bmm_full_body(const SmallBmm& g, int z, float* sbf) {
  float* As = sbf;
  float* Bs = sbf + SB_T * SBF_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * SB_T, n0 = (int64_t)blockIdx.x * SB_T;
  const float* A = g.A + (int64_t)z * g.sA;
  const float* B = g.B + (int64_t)z * g.sB;
  float* C = g.C + (int64_t)z * g.sC;
  // K == SBF_KMAX (checked by the launcher): all 32 loads of the thread are in flight together, then the LDS stores
  constexpr int NKS = SBF_KMAX / SB_K;
  sb_f4 ra[NKS][2], rb[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j, k0 = ks * SB_K;
      if (TA) ra[ks][j] = *reinterpret_cast<const sb_f4*>(A + (int64_t)(k0 + (f >> 4)) * g.lda + m0 + (f & 15) * 4);
      else ra[ks][j] = *reinterpret_cast<const sb_f4*>(A + (m0 + (f >> 3)) * g.lda + k0 + (f & 7) * 4);
      if (TB) rb[ks][j] = *reinterpret_cast<const sb_f4*>(B + (int64_t)(k0 + (f >> 4)) * g.ldb + n0 + (f & 15) * 4);
      else rb[ks][j] = *reinterpret_cast<const sb_f4*>(B + (n0 + (f >> 3)) * g.ldb + k0 + (f & 7) * 4);
    }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j, k0 = ks * SB_K;
      if (TA) {
#pragma unroll
        for (int c = 0; c < 4; ++c) As[((f & 15) * 4 + c) * SBF_PITCH + k0 + (f >> 4)] = ra[ks][j][c];
      } else {
        *reinterpret_cast<sb_f4*>(As + (f >> 3) * SBF_PITCH + k0 + (f & 7) * 4) = ra[ks][j];
      }
      if (TB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[((f & 15) * 4 + c) * SBF_PITCH + k0 + (f >> 4)] = rb[ks][j][c];
      } else {
        *reinterpret_cast<sb_f4*>(Bs + (f >> 3) * SBF_PITCH + k0 + (f & 7) * 4) = rb[ks][j];
      }
    }
  __syncthreads();
  sb_f16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = As + (wm * 32 + r) * SBF_PITCH + 8 * kh;
  const float* bp = Bs + (wn * 32 + r) * SBF_PITCH + 8 * kh;
  for (int k0 = 0; k0 < SBF_KMAX; k0 += 16) {
    sb_b8 ah, al, bh, bl;
    sb_split(*reinterpret_cast<const sb_f4*>(ap + k0), *reinterpret_cast<const sb_f4*>(ap + k0 + 4), ah, al);
    sb_split(*reinterpret_cast<const sb_f4*>(bp + k0), *reinterpret_cast<const sb_f4*>(bp + k0 + 4), bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  const int64_t n = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    float v = g.alpha * acc[e];
    if (m == n) v += g.ident;
    float* p = C + m * g.ldc + n;
    if (g.accumulate) v += *p;
    *p = v;
    if (g.C2) g.C2[(int64_t)z * g.sC + m * g.ldc + n] = g.alpha2 * acc[e] + (m == n ? g.ident2 : 0.f);
  }
}
template <bool TA, bool TB>
__global__ __launch_bounds__(SB_THREADS) void small_bmm_full_kernel(SmallBmm g) {
  extern __shared__ __attribute__((aligned(16))) float sbf[];
  small_bmm_full_body<TA, TB>(g, (int)blockIdx.z, sbf);
}
// TWO independent batches of products in one launch (same shapes, any two modes): blockIdx.z < batch -> the first.  The backward of a
// pseudo-inverse iteration is four pairs of independent 256^3 products (e.g. dzp = dz t3^T and dt3 = zp^T dz): 9 launches become 5.
MHIMX_DEV void small_bmm_full_any(const SmallBmm& g, int mode, int z, float* sbf) {
  if (mode == 0) small_bmm_full_body<false, false>(g, z, sbf);
  else if (mode == 1) small_bmm_full_body<false, true>(g, z, sbf);
  else small_bmm_full_body<true, true>(g, z, sbf);
}
__global__ __launch_bounds__(SB_THREADS) void small_bmm_pair_kernel(SmallBmm g0, int mode0, SmallBmm g1, int mode1, int batch) {
  extern __shared__ __attribute__((aligned(16))) float sbf[];
  if ((int)blockIdx.z < batch) small_bmm_full_any(g0, mode0, (int)blockIdx.z, sbf);
  else small_bmm_full_any(g1, mode1, (int)blockIdx.z - batch, sbf);
}

// ---------------------------------------------------------------------------------------------------------------------------
// A CHAIN of dependent batched 256 x 256 x 256 products in ONE launch (mhimx_bmm_chain): the Nystrom pseudo-inverse iteration is 24 such
// products per layer pass, each reading the previous one's output (nystrom_attention.py:21-25).  As launches a product took 7.4 us, of
// which (stamped, round 3): 1.0 us operand loads, 1.8 us LDS staging (transposing stores), 3.1 us the 16 dependent k-steps - and 2.5 us of
// THOSE are the bf16 hi/lo splits, every operand element split again by each of the two waves that use it and again by every product
// that reads the matrix.  Here
//   * a matrix lives in memory as its SPLIT IMAGE - bf16 hi plane + bf16 lo plane, [256][256] each, in the orientation its reader wants
//     ("N": rows k-contiguous as stored, the A operand; "T": the transpose, the B operand) - written once by the step that produces it
//     (4096 splits per workgroup instead of 65536); fp32 copies are written only where something outside the chain reads them;
//   * both operand panels ([64 rows][256 k] x 2 planes) go global -> LDS by DMA (no registers, no LDS stores), rows XOR-swizzled by the
//     per-lane GLOBAL address (LDS stays lane-linear: chunk c of row r sits at chunk c ^ (r & 31)), so the fragment reads are b128 and
//     conflict-free with an unpadded 512-byte pitch;
//   * 16 waves: wave w owns the 16 x 16 block (w >> 2, w & 3) of the tile over the WHOLE k - 24 MFMAs (16x16x32), no VALU between, no
//     partial sums (k quarters of 32 x 32 blocks meeting through LDS read half the fragments but paid 2 300 cycles for the meeting:
//     168 -> 162 us per forward chain);
//   * 128 persistent workgroups (one 64 x 64 tile of one of the 8 heads each) walk the chain; only the 16 workgroups of a head depend on
//     each other, so the hand-off is a per-head arrival counter, not a grid barrier:
//       producer: everything leaves as 16-byte WRITE-THROUGH stores (sc0 sc1), s_waitcnt vmcnt(0), workgroup barrier, one relaxed
//                 agent-scope atomic add;
//       consumer: one lane polls the counter (relaxed agent-scope loads, s_sleep between), workgroup barrier, then DMA-loads its panels
//                 with sc0 sc1 (they bypass this CU's L1 and this XCD's L2 and see what the producers wrote through) - the {sc0 sc1 stores
//                 and loads on both sides} form of MI355X_MICROARCH.md: no L2 write-back, no L1 invalidate, placement-independent.
// Workgroup b serves head b % 8.  The last arrival of a head's last step zeroes its counter: the counters are all-zero again when the
// launch ends (hipGraph replays re-use them).  tools/exp_chain.py: values, time, the phase stamps of one step (-DCH_PROF=<step>), a stress run.
// MEASURED and not kept (round 3, same box, forward / forward + backward of one layer as hipGraph replays, 166 / 426 us here):
//   * plain stores (acknowledged by the XCD's L2) instead of write-through ones when the head's workgroups verify at run time (HW_REG_XCC_ID
//     table) that they share an XCD: 166 / 424 us, bit-identical over 300 repetitions under load - the store acknowledge is not the bound;
//   * an arrival atomic without return value (no round trip for lane 0), s_sleep 0 / 4 in the poll: +-1 %.
// One stage of the k-quarter form, stamped (shader clocks, forward): poll 900, panels by DMA 3700 (128 KiB per workgroup at the ~85 GB/s a
// block reads written-through lines at), products 1900 (LDS-read bound: 256 KiB of fragments), quarters -> tiles 2300, stores + acknowledge
// 2400, arrival 1300.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CH_MAX = 38, CH_THREADS = 1024, CH_SPIN_MAX = 1 << 22;
struct ChainStep {             // (104 bytes: 38 of them + the header must fit the 4 KiB of kernel arguments)
  const void* A; const void* B; float* C; void* PN; void* PT; void* PN2; void* PT2; const float* D; const float* D2;
  float alpha, ident, alpha2, ident2, dscale, d2scale;
  int kind;
};
static_assert(sizeof(ChainStep) * CH_MAX + 32 <= 4096, "the chain table is a by-value kernel argument");
struct Chain { ChainStep st[CH_MAX]; int n, groups; unsigned* ctr; };
typedef unsigned sb_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) void* ch_gptr;
typedef __attribute__((address_space(3))) void* ch_lptr;
constexpr int CH_PLANE = SBF_KMAX * SBF_KMAX * 2;               // bytes of one bf16 plane of one head
constexpr int CH_PANEL = SB_T * SBF_KMAX * 2;                   // bytes of one plane of one 64-row panel in LDS (32 KiB)

__global__ __launch_bounds__(CH_THREADS) void bmm_chain_kernel(Chain c) {
  extern __shared__ __attribute__((aligned(16))) char chs[];   // [A hi | A lo | B hi | B lo] panels; after the products: the output tiles
  float* T1 = reinterpret_cast<float*>(chs + 2 * CH_PANEL);     // output tiles [64][68] fp32
  float* T2 = T1 + SB_T * 68;
  float* TT = T2 + SB_T * 68;                                   // transposes of the first / second output
  float* TT2 = TT + SB_T * 68;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Which tile a workgroup computes is NOT its block index: the workgroups of head b % 8 draw TICKETS (stage-major: ticket t = stage *
  // per_stage + group * 16 + tile) from the head's ticket counter.  A ticket waits only for tickets below it, and every ticket below it
  // is held by a workgroup that has started - so the chain completes with ANY number of resident workgroups (other kernels or processes
  // holding CUs, a grid larger than the free CUs): co-residency buys speed, never correctness.  With all of them resident every workgroup
  // draws one ticket per stage; the next ticket is drawn while the current one is computed (no exposed round trip).
  const int head = (int)blockIdx.x & 7;
  constexpr unsigned MAT_BYTES = SBF_KMAX * SBF_KMAX * 4;
  constexpr int POL = 1 | 16;                                   // sc0 sc1
  __shared__ int dead;
  __shared__ unsigned next_ticket;
  const int64_t hoff = (int64_t)head * SBF_KMAX * SBF_KMAX;     // elements of an fp32 matrix; a split image has the same byte size
  unsigned* ctr = c.ctr + head * 64;                            // arrivals of the head: a 128-byte line of its own (256 workgroups polling and
  unsigned* tkt = ctr + 32;                                     // adding on ONE line serialised at its L2 bank: +4 us per stage); tickets: the next line
  unsigned* lft = ctr + 48;
  const unsigned per_stage = 16u * (unsigned)c.groups, total = per_stage * (unsigned)c.n;
  if (tid == 0) {
    dead = 0;
    next_ticket = __hip_atomic_fetch_add(tkt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  unsigned ticket = __builtin_amdgcn_readfirstlane(next_ticket);        // wave-uniform BY CONSTRUCTION: say so (descriptors live in SGPRs)
#ifdef CH_PROF
  unsigned long long tsv[12];
#define CH_STAMP(i) do { if (s == CH_PROF) tsv[i] = __builtin_readcyclecounter(); } while (0)
#else
#define CH_STAMP(i) do {} while (0)
#endif
  while (ticket < total) {
    const int s = (int)(ticket / per_stage), slot = (int)(ticket % per_stage);
    const int group = slot >> 4, tile = slot & 15;
    const int m0 = (tile >> 2) * SB_T, n0 = (tile & 3) * SB_T;
    const ChainStep st = c.st[s * c.groups + group];
    CH_STAMP(0);
    unsigned nt = 0;
    if (tid == 0) {
      if (s > 0) {                                              // every tile of this head's previous stage is written
        const unsigned want = per_stage * (unsigned)s;
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < CH_SPIN_MAX) __builtin_amdgcn_s_sleep(1);
        if (spins >= CH_SPIN_MAX) {                             // cannot happen by the ticket order; a backstop against a hung GPU all the same:
          dead = 1;                                             // counters[512] != 0 tells the host, the outputs are invalid
          __hip_atomic_store(c.ctr + 512, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      nt = __hip_atomic_fetch_add(tkt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next ticket: its round trip flies under this one's work
    }
    __syncthreads();
    if (dead) return;
    CH_STAMP(1);
    if (st.kind < 0) {                                          // nothing to do for this group at this stage: only arrive
    } else if (st.kind == 0) {
      // 128 DMA instructions of 1 KiB (two 512-byte rows of one plane) per workgroup, 8 per wave
      const char* ga = reinterpret_cast<const char*>(st.A) + hoff * 4 + (int64_t)m0 * 512;
      const char* gb = reinterpret_cast<const char*>(st.B) + hoff * 4 + (int64_t)n0 * 512;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = wave * 8 + i, opnd = q >> 6, plane = (q >> 5) & 1, rp = q & 31;
        const int row = 2 * rp + (lane >> 5), chunk = (lane & 31) ^ (row & 31);
        const char* src = (opnd ? gb : ga) + plane * CH_PLANE + row * 512 + chunk * 16;
        __builtin_amdgcn_global_load_lds((ch_gptr)src, (ch_lptr)(chs + (opnd * 2 + plane) * CH_PANEL + rp * 1024), 16, 0, POL);
      }
      // every wave one 16 x 16 block over the WHOLE k: no partial sums, no reduce (twice the fragment reads: 512 KiB per tile)
      const int bm = wave >> 2, bn = wave & 3, c16 = lane & 15, q4 = lane >> 4;
      sb_f4 a4 = sb_f4{0.f, 0.f, 0.f, 0.f};
      if (st.D) {                                               // the accumulator starts at (dscale D + d2scale D2) / alpha
        const float ia = st.dscale / st.alpha;
        __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(st.D + hoff), 0, MAT_BYTES, 0x00027000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a4[i] = ia * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                           rD, (unsigned)(((m0 + 16 * bm + 4 * q4 + i) * SBF_KMAX + n0 + 16 * bn + c16) * 4), 0, POL));
        if (st.D2) {
          const float ib = st.d2scale / st.alpha;
          __amdgpu_buffer_rsrc_t rD2 = __builtin_amdgcn_make_buffer_rsrc((void*)(st.D2 + hoff), 0, MAT_BYTES, 0x00027000);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            a4[i] += ib * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                              rD2, (unsigned)(((m0 + 16 * bm + 4 * q4 + i) * SBF_KMAX + n0 + 16 * bn + c16) * 4), 0, POL));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my rows have landed in LDS
      CH_STAMP(2);
      __syncthreads();
      CH_STAMP(3);
      {
        const int ra = 16 * bm + c16, rb = 16 * bn + c16;
        const char* pa = chs + ra * 512;
        const char* pb = chs + 2 * CH_PANEL + rb * 512;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int chunk = 4 * ks + q4;
          const int oa = ((chunk ^ (ra & 31)) * 16), ob = ((chunk ^ (rb & 31)) * 16);
          const sb_b8 ah = *reinterpret_cast<const sb_b8*>(pa + oa), al = *reinterpret_cast<const sb_b8*>(pa + CH_PANEL + oa);
          const sb_b8 bh = *reinterpret_cast<const sb_b8*>(pb + ob), bl = *reinterpret_cast<const sb_b8*>(pb + CH_PANEL + ob);
          a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, a4, 0, 0, 0);
          a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, a4, 0, 0, 0);
          a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, a4, 0, 0, 0);
        }
      }
      CH_STAMP(4);
      __syncthreads();                                          // the panels are consumed: the tiles go through the same LDS
      CH_STAMP(8);
      CH_STAMP(9);
      {
        const int ml = 16 * bm + 4 * q4, nl = 16 * bn + c16;
        sb_f4 t, t2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool diag = (m0 + ml + j) == (n0 + nl);
          const float v1 = st.alpha * a4[j] + (diag ? st.ident : 0.f);
          const float v2 = st.alpha2 * a4[j] + (diag ? st.ident2 : 0.f);
          T1[(ml + j) * 68 + nl] = v1;
          T2[(ml + j) * 68 + nl] = v2;
          t[j] = v1;
          t2[j] = v2;
        }
        *reinterpret_cast<sb_f4*>(TT + nl * 68 + ml) = t;
        *reinterpret_cast<sb_f4*>(TT2 + nl * 68 + ml) = t2;
      }
      CH_STAMP(10);
    } else {                                                    // kind 1: split the fp32 matrix alpha * A (+ D) into its images, no product
      const int row = tid >> 4, c4 = (tid & 15) * 4;
      const unsigned off = (unsigned)(((m0 + row) * SBF_KMAX + n0 + c4) * 4);
      __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const float*>(st.A) + hoff), 0, MAT_BYTES, 0x00027000);
      sb_f4 v = st.alpha * __builtin_bit_cast(sb_f4, __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, POL));
      if (st.D) {
        __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(st.D + hoff), 0, MAT_BYTES, 0x00027000);
        v += __builtin_bit_cast(sb_f4, __builtin_amdgcn_raw_buffer_load_b128(rD, off, 0, POL));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        T1[row * 68 + c4 + q] = v[q];
        TT[(c4 + q) * 68 + row] = v[q];
      }
    }
    __syncthreads();
    CH_STAMP(5);
    {
      const int row = tid >> 4, c4 = (tid & 15) * 4;
      const unsigned off = (unsigned)(((m0 + row) * SBF_KMAX + n0 + c4) * 4);
      if (st.C) {
        __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(st.C + hoff), 0, MAT_BYTES, 0x00027000);
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const sb_u4*>(T1 + row * 68 + c4), rC, off, 0, POL);
      }
      // the split images: threads 0-511 the N image(s), 512-1023 the T image(s), 8 elements each
      const int t = tid & 511, prow = t >> 3, c8 = (t & 7) * 8;
      const bool second = tid >= 512;
      const unsigned po = (unsigned)((((second ? n0 : m0) + prow) * SBF_KMAX + (second ? m0 : n0) + c8) * 2);
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        void* img = o ? (second ? st.PT2 : st.PN2) : (second ? st.PT : st.PN);
        if (img) {
          const float* src = (o ? (second ? TT2 : T2) : (second ? TT : T1)) + prow * 68 + c8;
          sb_b8 hi, lo;
          sb_split(*reinterpret_cast<const sb_f4*>(src), *reinterpret_cast<const sb_f4*>(src + 4), hi, lo);
          __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(img) + hoff * 4), 0, MAT_BYTES, 0x00027000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sb_u4, hi), rP, po, 0, POL);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sb_u4, lo), rP, po + CH_PLANE, 0, POL);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my write-through stores have left (inline asm: the compiler must not drop it)
    if (tid == 0) next_ticket = nt;
    __syncthreads();                                            // ... and everybody's; the LDS tiles are free again
    CH_STAMP(6);
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CH_STAMP(7);
    ticket = __builtin_amdgcn_readfirstlane(next_ticket);       // (written before this iteration's last barrier; rewritten after the next one's first)
  }
  // leaving: the head's LAST workgroup to leave puts its three counters back to zero for the next launch (nobody of this launch reads them
  // after that: every other workgroup of the head has drawn its final, out-of-range ticket already)
  if (tid == 0) {
    const unsigned left = __hip_atomic_fetch_add(lft, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == per_stage - 1u) {
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tkt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(lft, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef CH_PROF
  if (tid == 0 && blockIdx.x < 8) for (int i = 0; i < 12; ++i) reinterpret_cast<unsigned long long*>(c.ctr + 576)[blockIdx.x * 12 + i] = tsv[i];
#endif
}

bool small_bmm_ok(int mode, const mhimx_gemm_nt_args& g, int batch, int64_t sA, int64_t sB, int64_t sC) {
  if (g.prec == MHIMX_PREC_F32 || g.rows || g.bias || g.M % SB_T || g.N % SB_T || g.K % SB_K) return false;
  if (g.M > 512 || g.N > 512 || g.K > 1024 || batch > 65535) return false;
  if ((g.M / SB_T) * (g.N / SB_T) * batch < 32) return false;                 // too few workgroups to be worth it
  return g.lda % 4 == 0 && g.ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 && aligned16(g.A) && aligned16(g.B) && mode >= 0 && mode <= 2;
}

int small_bmm2(hipStream_t st, int mode, const mhimx_gemm_nt_args& a, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident,
               float* c2, float alpha2, float ident2) {
  SmallBmm g;
  g.A = a.A; g.B = a.B; g.C = a.C; g.lda = a.lda; g.ldb = a.ldb; g.ldc = a.ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K; g.alpha = alpha; g.ident = ident; g.accumulate = a.accumulate;
  g.C2 = c2; g.alpha2 = alpha2; g.ident2 = ident2;
  if (c2 && g.K != SBF_KMAX) return fail(-1, "bmm_affine2: the second output needs the whole-panel kernel (K = 256)");
  dim3 grid((unsigned)(a.N / SB_T), (unsigned)(a.M / SB_T), (unsigned)batch);
  if (g.K == SBF_KMAX) {
    constexpr int SM = 2 * SB_T * SBF_PITCH * 4;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM)));
    if (mode == 0) hipLaunchKernelGGL((small_bmm_full_kernel<false, false>), grid, dim3(SB_THREADS), SM, st, g);
    else if (mode == 1) hipLaunchKernelGGL((small_bmm_full_kernel<false, true>), grid, dim3(SB_THREADS), SM, st, g);
    else hipLaunchKernelGGL((small_bmm_full_kernel<true, true>), grid, dim3(SB_THREADS), SM, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (mode == 0) hipLaunchKernelGGL((small_bmm_kernel<false, false>), grid, dim3(SB_THREADS), 0, st, g);
  else if (mode == 1) hipLaunchKernelGGL((small_bmm_kernel<false, true>), grid, dim3(SB_THREADS), 0, st, g);
  else hipLaunchKernelGGL((small_bmm_kernel<true, true>), grid, dim3(SB_THREADS), 0, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int small_bmm(hipStream_t st, int mode, const mhimx_gemm_nt_args& a, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident) {
  return small_bmm2(st, mode, a, batch, sA, sB, sC, alpha, ident, nullptr, 0.f, 0.f);
}

static SmallBmm sb_args(const mhimx_gemm_nt_args& a, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident) {
  SmallBmm g;
  g.A = a.A; g.B = a.B; g.C = a.C; g.lda = a.lda; g.ldb = a.ldb; g.ldc = a.ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K; g.alpha = alpha; g.ident = ident; g.accumulate = a.accumulate;
  g.C2 = nullptr; g.alpha2 = g.ident2 = 0.f;
  return g;
}

}  // namespace mhimx

extern "C" int mhimx_bmm_affine_pair(void* stream, int32_t mode0, const mhimx_gemm_nt_args* a0, float alpha0, float ident0, int32_t mode1,
                                     const mhimx_gemm_nt_args* a1, float alpha1, float ident1, int32_t batch, int64_t stride) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a0 && a1 && a0->A && a0->B && a0->C && a1->A && a1->B && a1->C && batch >= 1, "bmm_affine_pair: null args");
  MHIMX_CHECK_ARG(a0->M == SBF_KMAX && a0->N == SBF_KMAX && a0->K == SBF_KMAX && a1->M == SBF_KMAX && a1->N == SBF_KMAX && a1->K == SBF_KMAX,
                  "bmm_affine_pair: 256 x 256 x 256 products only");
  MHIMX_CHECK_ARG(a0->C != a1->C, "bmm_affine_pair: the two products must write different outputs");
  MHIMX_CHECK_ARG(small_bmm_ok(mode0, *a0, 2 * batch, stride, stride, stride) && small_bmm_ok(mode1, *a1, 2 * batch, stride, stride, stride),
                  "bmm_affine_pair: 16-byte aligned contiguous operands, not the f32 mode");
  constexpr int SM = 2 * SB_T * SBF_PITCH * 4;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM)));
  hipLaunchKernelGGL(small_bmm_pair_kernel, dim3(SBF_KMAX / SB_T, SBF_KMAX / SB_T, (unsigned)(2 * batch)), dim3(SB_THREADS), SM, (hipStream_t)stream,
                     sb_args(*a0, stride, stride, stride, alpha0, ident0), (int)mode0, sb_args(*a1, stride, stride, stride, alpha1, ident1), (int)mode1,
                     (int)batch);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_bmm_affine(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA, int64_t strideB,
                                int64_t strideC, float alpha, float ident) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a && a->A && a->B && a->C && batch >= 1, "bmm_affine: null args");
  MHIMX_CHECK_ARG(ident == 0.f || a->M == a->N, "bmm_affine: ident * I needs square outputs");
  MHIMX_CHECK_ARG(small_bmm_ok(mode, *a, batch, strideA, strideB, strideC),
                  "bmm_affine: M, N multiples of 64 (<= 512), K a multiple of 32 (<= 1024), 16-byte aligned operands, not the f32 mode");
  return small_bmm((hipStream_t)stream, mode, *a, batch, strideA, strideB, strideC, alpha, ident);
}

extern "C" int mhimx_bmm_affine2(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA, int64_t strideB,
                                 int64_t strideC, float alpha, float ident, float* C2, float alpha2, float ident2) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a && a->A && a->B && a->C && C2 && batch >= 1 && a->M == a->N, "bmm_affine2: null args / non-square output");
  MHIMX_CHECK_ARG(small_bmm_ok(mode, *a, batch, strideA, strideB, strideC),
                  "bmm_affine2: M, N multiples of 64 (<= 512), K = 256, 16-byte aligned operands, not the f32 mode");
  return small_bmm2((hipStream_t)stream, mode, *a, batch, strideA, strideB, strideC, alpha, ident, C2, alpha2, ident2);
}

extern "C" int mhimx_bmm_chain(void* stream, const mhimx_bmm_step* steps, int32_t stages, int32_t groups, uint32_t* counters) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(steps && counters && stages >= 1 && (groups == 1 || groups == 2) && stages * groups <= CH_MAX,
                  "bmm_chain: stages x groups (1 or 2) <= %d steps, 513 zeroed counters", CH_MAX);
  Chain c;
  c.n = stages;
  c.groups = groups;
  c.ctr = counters;
  for (int i = 0; i < stages * groups; ++i) {
    const mhimx_bmm_step& t = steps[i];
    MHIMX_CHECK_ARG(t.kind >= -1 && t.kind <= 1, "bmm_chain: step %d: kind -1 (idle), 0 (product of two split images) or 1 (split an fp32 matrix)", i);
    if (t.kind >= 0) {
      const bool second = t.PN2 || t.PT2;
      MHIMX_CHECK_ARG(t.A && (t.kind == 1 || t.B) && (t.C || t.PN || t.PT || second), "bmm_chain: step %d: missing operand / no output", i);
      MHIMX_CHECK_ARG(t.kind == 0 || !(t.C || second), "bmm_chain: step %d: a split step (kind 1) has the outputs PN / PT only", i);
      MHIMX_CHECK_ARG(!t.D || t.kind == 1 || (!second && t.alpha != 0.f), "bmm_chain: step %d: an addend goes with a single-output product, alpha != 0", i);
      MHIMX_CHECK_ARG(!t.D2 || (t.D && t.kind == 0), "bmm_chain: step %d: a second addend goes with a first one, on a product", i);
      const void* all[9] = {t.A, t.B, t.C, t.PN, t.PT, t.PN2, t.PT2, t.D, t.D2};
      for (const void* o : all) MHIMX_CHECK_ARG(aligned16(o), "bmm_chain: step %d: operands must be 16-byte aligned", i);
      const void* outs[5] = {t.C, t.PN, t.PT, t.PN2, t.PT2};
      for (const void* o : outs)
        MHIMX_CHECK_ARG(!o || (o != t.A && o != t.B && o != t.D2 && (o != t.D || o == t.C)), "bmm_chain: step %d: a step may not overwrite its own operands", i);
    }
    c.st[i] = ChainStep{t.A, t.B, t.C, t.PN, t.PT, t.PN2, t.PT2, t.D, t.D2, t.alpha, t.ident, t.alpha2, t.ident2, t.dscale != 0.f ? t.dscale : 1.f,
                        t.d2scale != 0.f ? t.d2scale : 1.f, t.kind};
  }
  constexpr int SM = 2 * CH_PANEL + 4 * SB_T * 68 * 4;          // the partials over the A panels, four output tiles over (and past) the B panels
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bmm_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM)));
  hipLaunchKernelGGL(bmm_chain_kernel, dim3(128 * groups), dim3(CH_THREADS), SM, (hipStream_t)stream, c);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
