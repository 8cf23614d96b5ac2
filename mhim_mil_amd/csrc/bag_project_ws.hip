// bag_project_ws.hip — the bag projection of bag_project.hip (teacher + student, ONE pass over the raw fp32 bag, 3-term bf16) with
// SPECIALISED waves and a ping-pong of the two consumer waves of every SIMD:
//
//   8 consumer waves (2 (M) x 4 (N) of 80 x 64 outputs, as in bag_project.hip) touch no global memory in the k loop: a k-step is a
//     LOAD phase (the 18 fragment reads of the tile) and a COMPUTE phase (its 60 MFMAs, every operand in registers).  Waves w and w + 4
//     share a SIMD and run half a k-step apart: while one issues MFMAs its partner reads its fragments, so the matrix pipe of every SIMD
//     always has a wave feeding it and the LDS reads run under it;
//   4 producer waves (one per SIMD) own the whole global -> LDS stream: the raw fp32 rows of X through registers (split to bf16 hi / lo on
//     the way, the paired 128-byte row image), the paired weight planes by LDS-DMA.  Their waits on memory stall nobody's MFMA issue.
//
// Why (round 3 measurements on bag_project.hip): in its lock-step loop all eight waves read LDS, then all eight
// issue MFMAs (MFMA-only loop 27 us, everything but the MFMAs 33 us, together 55 us); a ping-pong of UNspecialised waves hid the
// fragment reads (no-global-traffic loop 35 us) but its load phase - reads + split + stores + four DMA issues (~60-100 cycles each) + the
// wait on the landing data - was 1.8x the MFMA phase and set the pace (71 us, no gain).  The global stream alone needs ~26 us of the
// launch (1.66 MB per CU at ~28 B/clk/CU): it has to run beside the MFMAs, not between them.
//
// Slots (a k-step s is two slots, a workgroup barrier after each; k-step s's tile lives in stage (s + 2) % 3 of the 156 KB ring - the
// offset of 2 since round 5: the epilogue's staging rows take the low 83.5 KB, and the NEXT output tile's first stages are requested into
// stage 2 while they are read; the stage indices below are relative to that offset):
//   slot 2s   : group 0 loads tile s    | group 1 computes tile s-1 | producers split + store A(s+1) (stage (s+1) % 3: tile s-2, long read),
//                                                                     request A(s+3)
//   slot 2s+1 : group 0 computes tile s | group 1 loads tile s      | producers issue the DMA of B(s+2) (stage (s+2) % 3 = (s-1) % 3: both
//                                                                     groups are past tile s-1), then wait until A(s+2) is in registers and
//                                                                     B(s+1) has landed
//   Tile s+1 is complete at the end of slot 2s+1; its first reader is group 0 in slot 2s+2.  Requests run two k-steps ahead (A) / one and
//   a half (B).
//
// Stamped with -DPW_PROF (tools/exp_proj_prof.py; c2 shape, shader cycles of workgroup 0): entry -> loop 5.2 k, k loop 69.4 k (32 k-steps;
// 2 x 60 MFMAs x 16 cycles = 61.4 k is the matrix pipe's own time), epilogue 27-29 k.  What the stamps changed:
//   * the producer waves are the YOUNGEST of the workgroup and VALU issue is arbitrated by priority, then age: at priority 0 their split +
//     store slots took 805 / 684 cycles of a ~1000-cycle slot and set the pace (78.7 k cycles for the loop); s_setprio 1 for the k loop
//     (back to 0 for the epilogue, which shares rows evenly) brought the loop to 69.4 k: 65.5 -> 61.4 us per launch.  Priority 2 / 3, or
//     priority 1 for the younger consumer group as well: no further gain.
//   * the epilogue, MEASURED and not kept: transposed accumulators (MFMA operands exchanged: four consecutive output COLUMNS per lane) with
//     bias / dropout / activation / 16-byte stores straight from the eight consumer waves' registers - no LDS staging, no barriers:
//     epilogue 38.9 k cycles (eight waves instead of twelve carry the arithmetic; a store instruction touches 16 rows x 64 B instead of one
//     1 KiB row), 65.6 vs 62.2 us; the same accumulators staged with 16-byte LDS stores (20 instead of 80 per lane): 64.2 vs 62.6 us.
//     Without its arithmetic AND its global stores the launch is 5 us shorter: the epilogue is the drain of 51 MB into HBM by 252 workgroups
//     that all reach it together, not instruction time.
//
// Round 5 (profiles/r05_store_path.md; the comments at the sites): PERSISTENT workgroups for launches of more than 256 output tiles, the next
// tile's first stages requested by LDS-DMA from inside the epilogue (first tile peeled: a straight-line listing for tools/asm_lint.py);
// the workgroups with the shorter tile list start late, each by its own fraction of the spare tile time (no lock-step store bursts);
// MODE 3 = PLAIN products (no activation / dropout / residual / d out-d pre): accumulators stored from the consumers' registers while the
// producers run the next tile's direct prologue.
#include "mma_tile.hpp"

namespace mhimx {

constexpr int WBM = 160, WBN = 256, WBK = 32;
constexpr int W_CONS = 8, W_PROD = 4, WTHREADS = 64 * (W_CONS + W_PROD);                           // 768 threads: three waves per SIMD
constexpr int WA_BYTES = WBM * 128, WB_BYTES = WBN * 128, WSTAGE = WA_BYTES + WB_BYTES, WNST = 3;     // 20 KiB + 32 KiB, x 3 = 156 KiB
constexpr int W_CUS = 256;                                                                          // MI355X: one persistent workgroup per CU
constexpr int WTP = WBN + 4;                                                                         // epilogue tile pitch (floats)

typedef __bf16 pw_bf4 __attribute__((ext_vector_type(4)));
typedef _Float16 pw_h4 __attribute__((ext_vector_type(4)));

MHIMX_DEV uint32_t pw_pair_hash(uint32_t row_key, uint32_t pair) { return mix32(row_key + pair * 0x85EBCA77u); }   // (bag_project.hip's stream)

// The bags of ONE launch (mhimx_bag_project_multi: the bags of an accumulation window share weights, shapes and the dropout law; each has its
// own rows, outputs and dropout seeds).  Row tiles are numbered bag-major, a bag's count rounded up to a multiple of 8 (XCD mapping).
constexpr int W_MAX_BAGS = 8;
struct ProjBags {
  int n_bags, tiles_per_bag;
  int stagger;                      // shader cycles the workgroups with the SHORTER tile list wait before their first tile (0: none)
  int stagger_spread;               // 1: each of them waits its own fraction of 2 x stagger (a whole tile's time) instead
  const float* X[W_MAX_BAGS];
  float* H[W_MAX_BAGS][MHIMX_PROJ_MAX_HEADS];
  void* dact[W_MAX_BAGS][MHIMX_PROJ_MAX_HEADS];
  uint64_t seed[W_MAX_BAGS][MHIMX_PROJ_MAX_HEADS];
};

// ---------------------------------------------------------------------------------------------------------------------------------
// The teacher's scorer in the epilogue (mhimx_proj_score, round 4).  Called by all twelve waves of a MODEL-0 workgroup after the k loop
// with the 160 x 256 pre-activations in the consumers' accumulators (wave (wm, wn): rows 80 wm .., columns 64 wn ..).
//   per 80-row half:  h = dropout(act(acc + bias)) in the owners' registers (it stays there: the pool needs it again), its bf16 hi / lo
//                     image to LDS [8 k-steps][80 rows][128 B] (the k loop's A-tile layout), then u_part [80, 144] = h [Wa ; Wp]_half^T
//                     on the matrix cores - wave w owns the 16-column block w of the 9 (wave 8, a producer, the class projections) -
//                     written through to the exchange scratch;
//   pair gate:        one arrival per workgroup, wait for the partner (block index +- 8: the same XCD, dispatched next to this one);
//   both alike:       u = u_part(columns 0..255) + u_part(columns 256..511) in that order, s = wc . act(u), class projections, the tile's
//                     max / sum, and sum_r e^{s_r - max} h_r over the OWN 256 columns from the registers.
// Nothing of the [N, 512] feature rows is written (unless H.H is given).
// ---------------------------------------------------------------------------------------------------------------------------------
struct ProjScore { const float *wa16, *wc; int act, C; float *s, *cproj, *pm, *pl, *pz, *xch; uint32_t* gate; };
constexpr int PS_UCOLS = 144;                       // 128 scorer columns + 16 (class projections, padded)
constexpr unsigned PS_GATE_SPINS = 1u << 22;

MHIMX_DEV float ps_row16_sum(float v) {              // sum over the 16 lanes of a DPP row, in every lane of the row
  v += dpp_mov<0xB1, 0xf>(0.f, v);
  v += dpp_mov<0x4E, 0xf>(0.f, v);
  v += dpp_mov<0x141, 0xf>(0.f, v);
  v += dpp_mov<0x140, 0xf>(0.f, v);
  return v;
}

template <bool WRITE_H>
MHIMX_DEV void pw_scored_epilogue(const mhimx_bag_project_args& g, const mhimx_proj_head& H, const ProjScore& sc, f32x4 (&acc)[NRA][NRB],
                                  char* smem, int m_tile, int64_t m0, int64_t n0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < W_CONS;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int cl = lane & 15, rq = lane >> 4;
  const int colhalf = (int)(n0 / WBN);
  const int E = (int)g.E;
  char* img = smem;                                                      // [8][80][128 B] = 80 KiB
  uint32_t* rkeys = reinterpret_cast<uint32_t*>(smem + 81920);           // [160]
  float* sred = reinterpret_cast<float*>(smem + 81920 + 1024);           // [8][160]
  float* srow = sred + 8 * 160;                                          // [160]
  float* prow = srow + 160;                                              // [160]
  float* zred = prow + 160;                                              // [2][256]
  float* red = zred + 512;                                               // [16]
  const bool hashed = H.drop_p > 0.f;                                    // (no injected masks here: the host refuses them)
  const uint64_t dseed = hashed ? eff_seed(H.drop_seed, g.drop_tick) : 0;
  const uint32_t thr16 = (uint32_t)(H.drop_p * 65536.f + 0.5f);
  const float inv_keep = 65536.f / (float)(65536u - thr16);
  __syncthreads();                                                       // the k loop's fragment reads are over
  if (hashed && tid < 160) rkeys[tid] = drop_row_key(dseed, (uint64_t)(m0 + tid));
  const int64_t xbase = ((int64_t)m_tile * 2) * 160 * PS_UCOLS;          // this row tile's two blocks of the exchange scratch
  float* xmine = sc.xch + xbase + (int64_t)colhalf * 160 * PS_UCOLS;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                                                     // row keys visible / the previous half's image reads are over
    if (consumer && wm == half) {
      // (row-major over (i, e), the four column blocks inside: one row key and one swizzle per row, and a scheduling fence per row - left
      // to itself the compiler interleaves all 80 elements' activations and spills 300 registers beside the 80 accumulators)
      float bias4[NRB];
#pragma unroll
      for (int j = 0; j < NRB; ++j) bias4[j] = H.bias ? H.bias[n0 + wn * 64 + j * 16 + cl] : 0.f;
      const int kin0 = cl, g80 = cl >> 3, idx = cl & 7;                  // column block j: k-step wn * 2 + (j >> 1), slot group (j & 1) * 2 + g80
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rl = i * 16 + rq * 4 + e;                            // row inside the half
          const int64_t m = m0 + half * 80 + rl;
          const int sw = mt_swz(rq * 4 + e);                             // (mt_swz depends on row & 15 only)
          const uint32_t rk = hashed ? rkeys[half * 80 + rl] : 0u;
          char* rowp = img + rl * 128 + idx * 2;
#pragma unroll
          for (int j = 0; j < NRB; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + cl;
            float v = act_fwd(acc[i][j][e] + bias4[j], g.act);
            float ks = 1.f;
            if (hashed) {
              const uint32_t hsh = pw_pair_hash(rk, (uint32_t)(n >> 1));
              ks = ((n & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= thr16 ? inv_keep : 0.f;
            }
            v *= ks;
            acc[i][j][e] = v;
            if constexpr (WRITE_H)                                       // (tests only - its own instantiation: the 64-bit addresses cost ~60
              if (m < g.N) H.H[m * H.ldh + n] = v;                       //  registers beside the 80 accumulators; production passes no buffer)
            const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
            const int g8 = (j & 1) * 2 + g80;
            char* base = rowp + (wn * 2 + (j >> 1)) * 10240;
            *reinterpret_cast<__bf16*>(base + (((2 * g8) ^ sw) << 4)) = hi;
            *reinterpret_cast<__bf16*>(base + (((2 * g8 + 1) ^ sw) << 4)) = lo;
          }
          __builtin_amdgcn_sched_barrier(0);
          (void)kin0;
        }
    }
    __syncthreads();
    if (wave < 8 || (wave == 8 && sc.C > 0)) {
      const int ab = wave;
      f32x4 u[NRA];
#pragma unroll
      for (int i = 0; i < NRA; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* bimg = sc.wa16 + ((int64_t)(ab * (E / 32) + (int)(n0 >> 5)) * 64 + lane) * 8;      // + 512 floats per k-step
      const int r16 = lane & 15, kg = lane >> 4, sw = mt_swz(r16);
      const char* fa_hi = img + r16 * 128 + (((2 * kg) ^ sw) << 4);
      const char* fa_lo = img + r16 * 128 + (((2 * kg + 1) ^ sw) << 4);
      f32x4 bh = *reinterpret_cast<const f32x4*>(bimg), bl = *reinterpret_cast<const f32x4*>(bimg + 4);
#pragma unroll 1
      for (int kk = 0; kk < 8; ++kk) {
        const int kn = kk + 1 < 8 ? kk + 1 : kk;
        const f32x4 nbh = *reinterpret_cast<const f32x4*>(bimg + kn * 512), nbl = *reinterpret_cast<const f32x4*>(bimg + kn * 512 + 4);
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
          const f32x4 ah = *reinterpret_cast<const f32x4*>(fa_hi + kk * 10240 + i * 2048);
          const f32x4 al = *reinterpret_cast<const f32x4*>(fa_lo + kk * 10240 + i * 2048);
          u[i] = mt_mfma(al, bh, u[i]);
          u[i] = mt_mfma(ah, bl, u[i]);
          u[i] = mt_mfma(ah, bh, u[i]);
        }
        bh = nbh;
        bl = nbl;
      }
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __hip_atomic_store(xmine + (int64_t)(half * 80 + i * 16 + rq * 4 + e) * PS_UCOLS + ab * 16 + cl, u[i][e], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- pair gate (counters only count up: an even value = nobody of the current launch has arrived)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(sc.gate + m_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old | 1u) + 1u;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(sc.gate + m_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < PS_GATE_SPINS)
      __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  // ---- scores: wave w sums its 16 scorer columns of every row; the class projections are columns 128.. of wave 8
  const float* x0 = sc.xch + xbase;
  const float* x1 = x0 + 160 * PS_UCOLS;
  if (wave < 8) {
    const float wca = sc.wc[wave * 16 + cl];
#pragma unroll 2
    for (int i2 = 0; i2 < 10; ++i2)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i2 * 16 + rq * 4 + e;
        const int64_t o = (int64_t)row * PS_UCOLS + wave * 16 + cl;
        const float uu = __hip_atomic_load(x0 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                         __hip_atomic_load(x1 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float t = ps_row16_sum(wca * act_fwd(uu, sc.act));
        if (cl == 0) sred[wave * 160 + row] = t;
      }
  } else if (wave == 8 && sc.C > 0 && colhalf == 0) {
    for (int i2 = 0; i2 < 10; ++i2)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = i2 * 16 + rq * 4 + e;
        const int64_t m = m0 + row;
        const int64_t o = (int64_t)row * PS_UCOLS + 128 + cl;
        if (cl < sc.C && m < g.N)
          sc.cproj[m * sc.C + cl] = __hip_atomic_load(x0 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                                    __hip_atomic_load(x1 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  __syncthreads();
  if (tid < 160) {
    float sv = ((sred[tid] + sred[160 + tid]) + (sred[320 + tid] + sred[480 + tid])) + ((sred[640 + tid] + sred[800 + tid]) + (sred[960 + tid] + sred[1120 + tid]));
    const int64_t m = m0 + tid;
    if (m >= g.N) sv = -INFINITY;
    else if (colhalf == 0) sc.s[m] = sv;
    srow[tid] = sv;
  }
  __syncthreads();
  if (wave == 0) {
    const float a = srow[lane], b = srow[lane + 64], c = lane < 32 ? srow[lane + 128] : -INFINITY;
    const float mt = wave_max(fmaxf(fmaxf(a, b), c));
    if (lane == 0) red[0] = mt;
  }
  __syncthreads();
  const float mt = red[0];
  if (tid < 160) prow[tid] = srow[tid] == -INFINITY ? 0.f : __expf(srow[tid] - mt);
  __syncthreads();
  if (wave == 0) {
    const float l = wave_sum((prow[lane] + prow[lane + 64]) + (lane < 32 ? prow[lane + 128] : 0.f));
    if (lane == 0 && colhalf == 0) { sc.pm[m_tile] = mt; sc.pl[m_tile] = l; }
  }
  if (consumer) {
#pragma unroll
    for (int j = 0; j < NRB; ++j) {
      float zc = 0.f;
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) zc += prow[wm * 80 + i * 16 + rq * 4 + e] * acc[i][j][e];
      zc += __shfl_xor(zc, 16);
      zc += __shfl_xor(zc, 32);
      if (rq == 0) zred[wm * 256 + wn * 64 + j * 16 + cl] = zc;
    }
  }
  __syncthreads();
  if (tid < 256) sc.pz[(int64_t)m_tile * E + n0 + tid] = zred[tid] + zred[256 + tid];
}

// MODE 0: feature rows only; 1: model 0 scored in the epilogue, its rows not written; 2: scored AND written (tests); 3: PLAIN products (no
// activation / dropout / residual rows / d out-d pre on any head: to_qkv, the backward's dX products) - the consumer waves store their
// accumulators straight from registers (64-byte segments: 68-71 B/ns for one CU against 139 for 16-byte stores, both far above a CU's share
// of the chip's 5.4-7.3 TB/s, tools/micro/store_rate.hip) while the producers are already on the next tile's prologue: no staging rows, no
// epilogue barriers, every tile takes the direct prologue.
template <int MODE>
__global__ __launch_bounds__(WTHREADS) void bag_project_ws_kernel(mhimx_bag_project_args g, ProjBags pb, ProjScore sc) {
  constexpr bool PLAIN = MODE == 3;
  constexpr int SCORED = PLAIN ? 0 : MODE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nN = (int)(g.n_heads * g.E / WBN), nM = (int)((g.N + WBM - 1) / WBM);
  const int tiles_per_head = (int)(g.E / WBN);
  const int nk = (int)(g.D / WBK);
  const bool producer = wave >= W_CONS;
  const int wm = (wave >> 2) & 1, wn = wave & 3;                            // consumer waves: M half (= ping-pong group), N quarter
  // PERSISTENT workgroups (round 5): virtual block v = blockIdx.x + j gridDim.x (the grid is a multiple of 8, so a workgroup keeps its
  // XCD); the column tiles of one row tile share an XCD (X rows via its L2) as before.  A launch's workgroups all take the same time, so
  // its cohorts run in lock-step and a new tile's first cold loads ran into the store burst of every other CU's epilogue
  // (profiles/r05_store_path.md: 8.7-10.3 k cycles entry -> loop in the third cohort against 4.8 k in the first): now the producers request
  // the NEXT tile's first stages right after the barrier that ends the k loop - B(0) by DMA into ring stage 2, which the epilogue's staging
  // tile (the low 83.5 KB) leaves alone; A(0), A(1), A(2) into their idle registers - and the k loop of the next tile starts one barrier
  // after the last row of this one.  Tile s of EVERY k loop lives in ring stage (s + 2) % 3 for that.
  const int total = nN * pb.tiles_per_bag * pb.n_bags;
  const int G = (int)gridDim.x;
  auto decode = [&](int v, int& bag, int& m_tile, int& n_tile) {
    const int xcd = v & 7, sidx = v >> 3;
    const int m_all = (sidx / nN) * 8 + xcd;
    n_tile = sidx % nN;
    bag = m_all / pb.tiles_per_bag;
    m_tile = m_all - bag * pb.tiles_per_bag;
  };
  auto next_tile = [&](int v) {                                             // the next virtual block of this workgroup that holds rows
    for (v += G; v < total; v += G) {
      int b_, m_, n_;
      decode(v, b_, m_, n_);
      if (m_ < nM) return v;
    }
    return total;
  };
  int vb = (int)blockIdx.x;
  {
    int b_, m_, n_;
    decode(vb, b_, m_, n_);
    if (m_ >= nM) vb = next_tile(vb);
  }
  if (vb >= total) return;
  // The workgroups of a launch all take the same time per tile, so their epilogues - each a burst at the chip's store ceiling - stay in
  // lock-step through every cohort (profiles/r05_store_path.md).  When the tiles do not divide evenly, the workgroups with one tile fewer
  // have a tile's time to spare: they start half a tile late, and the launch's stores leave as two smaller bursts in antiphase - for free.
  // Measured same-box: c3 (7-8 cohorts per product) 8.34 -> 8.22-8.26 ms for offsets of 20-45 k cycles; c5 (one product of 19.5 cohorts)
  // 3.85 -> 3.88-4.00 ms: over many cohorts the workgroups drift apart by themselves and the offset only costs - so up to 10 cohorts.
  if (pb.stagger > 0 && total > 2 * G && total <= 10 * G) {
    int mine = 0, longest = 0;
    for (int v = vb; v < total; v = next_tile(v)) ++mine;
    {
      int v0 = 0, b_, m_, n_;
      decode(v0, b_, m_, n_);
      if (m_ >= nM) v0 = next_tile(v0);
      for (; v0 < total; v0 = next_tile(v0)) ++longest;                      // (workgroup 0 holds a longest list)
    }
    if (mine < longest) {
      // spread over the whole spare tile time (a 97-stride permutation of the block index: neighbours - one XCD's CUs - far apart)
      const int64_t wait = pb.stagger_spread ? ((int64_t)pb.stagger * 2 * (int)((blockIdx.x * 97u) & 255u)) >> 8 : (int64_t)pb.stagger;
      const uint64_t t0 = __builtin_readcyclecounter();
      while ((int64_t)(__builtin_readcyclecounter() - t0) < wait) __builtin_amdgcn_s_sleep(16);
    }
  }

  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

#ifdef PW_PROF                                                  // cycles per phase, summed over the k loop (tools/exp_proj_prof.py)
#ifndef PW_PROF_BLOCK
#define PW_PROF_BLOCK 0
#endif
#ifndef PW_PROF_TILE
#define PW_PROF_TILE 0
#endif
  uint64_t pf_t0 = 0, pf_t1 = 0, pf_t2 = 0;
  uint64_t pf_rt0 = 0;                                            // the 100 MHz constant clock at the same instants (round 6: the shader clock a launch holds)
  uint32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t pf_t = 0;
  int pf_iter = 0;
#define PW_MARK(i) { const uint64_t now_ = __builtin_readcyclecounter(); pf[i] += (uint32_t)(now_ - pf_t); pf_t = now_; }
#define PW_START() { pf_t = __builtin_readcyclecounter(); }
#else
#define PW_MARK(i)
#define PW_START()
#endif

  // ---------------------------------------------------------------- producers' state (256 threads own the global -> LDS stream)
  // (a lean instruction stream matters: a producer wave shares its SIMD's issue with two consumer waves - the first version spent ~250
  // instructions per k-step here, mostly 64-bit address arithmetic and unpacked conversions, and its slot outlasted the 60-MFMA phase)
#ifndef PW_PROD_PRIO
#define PW_PROD_PRIO 1
#endif
  const int pt = tid - 64 * W_CONS, pw = wave - W_CONS;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  // A: 160 rows x 8 sixteen-byte units per k-step = 1280 units, five per thread: unit u = pt + 256 j -> row u >> 3, slot u & 7
  unsigned aoff[5], a_hi[5], a_lo[5];
  // B: 32 DMA pieces of 1 KiB (8 rows x 128 B) per k-step, eight per producer wave: piece P = 8 pw + q covers rows 8 P .. 8 P + 7; the
  // SOURCE slot is swizzled by the row (mt_swz depends on row & 15 = 8 (q & 1) + lane / 8).  One per-lane byte offset per piece against
  // ONE uniform base that advances 128 B per k-step.
  unsigned bvoff[8];
  const float* px = nullptr;                                                // the tile's bag
  const char* bbase = nullptr;                                              // the tile's weight rows, k-step 0
  struct ARegs { f32x4 v[5]; };
  if (producer) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int u = pt + 256 * j, row = u >> 3, slot = u & 7;
      const int sw = mt_swz(row), kg2 = (slot >> 1) * 2;
      a_hi[j] = lds0 + (unsigned)(row * 128 + ((kg2 ^ sw) << 4) + (slot & 1) * 8);
      a_lo[j] = lds0 + (unsigned)(row * 128 + (((kg2 + 1) ^ sw) << 4) + (slot & 1) * 8);
    }
    const int rl = lane >> 3, sl = lane & 7;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      bvoff[q] = (unsigned)((((pw * 64 + q * 8 + rl) * g.D) + (sl ^ mt_swz(8 * (q & 1) + rl)) * 4) * 4);
  }
  auto p_setup = [&](int v) {                                               // the tile's row offsets and weight rows
    int bag, m_tile, n_tile;
    decode(v, bag, m_tile, n_tile);
    px = pb.X[bag];
    const int64_t m0 = (int64_t)m_tile * WBM;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int u = pt + 256 * j, row = u >> 3, slot = u & 7;
      int64_t m = m0 + row;
      if (m >= g.N) m = g.N - 1;                                            // clamped rows feed accumulators that are never stored
      aoff[j] = (unsigned)((m * g.ldx + slot * 4) * 4);
    }
    const float* wp = (n_tile / tiles_per_head == 1) ? g.head[1].wp : g.head[0].wp;
    bbase = reinterpret_cast<const char*>(wp + (int64_t)(n_tile % tiles_per_head) * WBN * g.D);     // + 128 B per k-step
  };
  auto issue_b = [&](const char* bk, unsigned stage_off, int q0) {          // pieces q0 .. q0+3
    const unsigned sb = lds0 + stage_off + WA_BYTES + pw * 8192;
#pragma unroll
    for (int q = q0; q < q0 + 4; ++q)
      __builtin_amdgcn_global_load_lds((gptr_f)(bk + bvoff[q]), (lptr_f)(uintptr_t)(sb + q * 1024), 16, 0, 0);
  };
  auto uni = [](const float* p) {                                           // (a loop-carried pointer: the compiler cannot see that it is uniform)
    const uint64_t u = (uint64_t)(uintptr_t)p;
    return reinterpret_cast<const float*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u)));
  };
  auto load_a3 = [&](const float* xk, ARegs& r) {                           // units 0..2
    asm volatile("global_load_dwordx4 %0, %3, %6\n\tglobal_load_dwordx4 %1, %4, %6\n\tglobal_load_dwordx4 %2, %5, %6"
                 : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]) : "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "s"(xk) : "memory");
  };
  auto load_a2 = [&](const float* xk, ARegs& r) {                           // units 3..4
    asm volatile("global_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4"
                 : "=&v"(r.v[3]), "=&v"(r.v[4]) : "v"(aoff[3]), "v"(aoff[4]), "s"(xk) : "memory");
  };
  typedef float pw_f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pw_b2 __attribute__((ext_vector_type(2)));
  typedef unsigned pw_u2 __attribute__((ext_vector_type(2)));
  auto split_store = [&](const f32x4& v, unsigned hi_addr, unsigned lo_addr) {        // one v_cvt_pk_bf16_f32 per PAIR, one ds_write_b64 per plane
    pw_u2 hi, lo;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float a = v[2 * p], b = v[2 * p + 1];
      const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(pw_f2{a, b}, pw_b2));
      hi[p] = h;
      lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(pw_f2{a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u)}, pw_b2));
    }
    *reinterpret_cast<__attribute__((address_space(3))) pw_u2*>((uintptr_t)hi_addr) = hi;
    *reinterpret_cast<__attribute__((address_space(3))) pw_u2*>((uintptr_t)lo_addr) = lo;
  };
#define PW_NAME5(r) "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4])
  // the NEXT tile's first stages, requested from inside this tile's epilogue - by LDS-DMA only, into the 72 KB the staging rows leave free: the
  // producers' registers must not hold requests in flight across compiler-scheduled code (a first version kept A(0..2) in registers: the
  // compiler moved them between physical registers around the loops BEFORE the wait, and spilled one).  B(0) -> its place in ring stage 2;
  // the raw fp32 rows of A(0) -> the A region of stage 2 ([row][128 B]: unit u of the producers' mapping is bytes 16 u .. 16 u + 15, so a wave's
  // DMA instruction j covers its 8 rows 8 pw + 32 j ..), converted IN PLACE at the top of the next tile (the 8 units of a row belong to 8
  // consecutive lanes of one wave: all of them have read before any writes); the raw rows of A(1) -> the gap between the staging rows and
  // stage 2, read into `ra` there.
  constexpr unsigned RAW1_OFF = 82 * 1024;                                  // staging rows + row keys end at 83 520
  static_assert(80 * WTP * 4 + 320 <= RAW1_OFF && RAW1_OFF + WA_BYTES <= 2 * WSTAGE, "raw A(1) sits between the staging rows and ring stage 2");
  auto prefetch = [&]() {
    issue_b(bbase, 2 * WSTAGE, 0);
    issue_b(bbase, 2 * WSTAGE, 4);
    const char* xb0 = reinterpret_cast<const char*>(px);
    const char* xb1 = xb0 + (nk > 1 ? 128 : 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const unsigned dst = lds0 + (unsigned)((pw * 8 + 32 * j) * 128);
      __builtin_amdgcn_global_load_lds((gptr_f)(xb0 + aoff[j]), (lptr_f)(uintptr_t)(dst + 2 * WSTAGE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_f)(xb1 + aoff[j]), (lptr_f)(uintptr_t)(dst + RAW1_OFF), 16, 0, 0);
    }
  };
  struct TileCtx { int m_tile, hd, vb_next; int64_t m0, n0; mhimx_proj_head H; };
  auto tile_ctx = [&](int v) {
    int bag, m_tile, n_tile;
    decode(v, bag, m_tile, n_tile);
    TileCtx t;
    t.m_tile = m_tile;
    t.vb_next = next_tile(v);
    t.m0 = (int64_t)m_tile * WBM;
    t.hd = n_tile / tiles_per_head;
    t.n0 = (int64_t)(n_tile % tiles_per_head) * WBN;
    t.H = g.head[0];
    if (t.hd == 1) t.H = g.head[1];
    t.H.H = pb.H[bag][t.hd];
    t.H.dact = pb.dact[bag][t.hd];
    t.H.drop_seed = pb.seed[bag][t.hd];
    return t;
  };
  // ================================================================= epilogue: all twelve waves, two 80-row halves through LDS
  // (a lambda instantiated once per ROLE: the tile loops below are separate per role, so that the producers' loop-carried registers - row
  // offsets, the three A register sets - and the consumers' accumulators are never live together: in one loop they spilled 576 bytes)
  const int lane_ = lane, tid_ = tid;
  auto epilogue = [&](auto is_prod, f32x4 (&acc)[NRA][NRB], const TileCtx& t) {
  constexpr bool producer = decltype(is_prod)::value;
  // (laundered lane / thread ids: everything the epilogue derives from them - column offsets, 64-bit row addresses, bias - is the same for
  // every tile, so the compiler hoisted it out of the tile loop, kept it live across the k loop beside 152 accumulator and fragment
  // registers and spilled it around the loop: 60 bytes of scratch per lane, +7 MB of traffic each way per c2 launch under the counters)
  int lane = lane_, tid = tid_;
  asm volatile("" : "+v"(lane), "+v"(tid));
  const int64_t m0 = t.m0, n0 = t.n0;
  const mhimx_proj_head& H = t.H;
  const int vb_next = t.vb_next;
  float* tile = reinterpret_cast<float*>(smem);
  uint32_t* rkeys = reinterpret_cast<uint32_t*>(smem + 80 * WTP * 4);
  const int c4 = lane * 4, r0w = wave;                                      // this thread's 4 columns are fixed; rows wave, wave + 12, ..
  const int64_t n = n0 + c4;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (H.bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(H.bias + n); bias[0] = b[0]; bias[1] = b[1]; bias[2] = b[2]; bias[3] = b[3]; }
  const bool hashed = H.drop_p > 0.f && !H.drop_mask;
  const uint64_t dseed = hashed ? eff_seed(H.drop_seed, g.drop_tick) : 0;
  const uint32_t thr16 = (uint32_t)(H.drop_p * 65536.f + 0.5f);
  const float inv_keep = H.drop_mask ? 1.f / (1.f - H.drop_p) : 65536.f / (float)(65536u - thr16);
  _Float16* dact = reinterpret_cast<_Float16*>(H.dact);
#ifdef PW_PROF
  uint32_t pe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  pf_t = __builtin_readcyclecounter();
#define PE_MARK(i) { const uint64_t now_ = __builtin_readcyclecounter(); pe[i] += (uint32_t)(now_ - pf_t); pf_t = now_; }
#else
#define PE_MARK(i)
#endif
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    // residual rows (the TransLayer's y = x + ...): a wave's seven rows of the half are read one after another - with the read next to
    // its add every row waited a memory latency (the resid launches of c3 ran ~160 us against ~81 us for the same product without).
    // A ring of three rows, requested before the accumulators go through LDS and refilled three rows ahead.
    f32x4 rq0 = {0.f, 0.f, 0.f, 0.f}, rq1 = rq0, rq2 = rq0;
    auto ld_res = [&](int r) {
      const int64_t m = m0 + half * 80 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < 80 && m < g.N) v = *reinterpret_cast<const f32x4*>(H.resid + m * H.ldr + n);
      return v;
    };
    if (H.resid) { rq0 = ld_res(r0w); rq1 = ld_res(r0w + (W_CONS + W_PROD)); rq2 = ld_res(r0w + 2 * (W_CONS + W_PROD)); }
    __syncthreads();                                          // fragment reads / the previous half's tile reads are over
    if constexpr (producer)
      if (half == 0 && vb_next < total) {                     // every stage is free now: the next tile's first requests leave before the rows
        p_setup(vb_next);
        prefetch();
      }
    PE_MARK(0);
    if constexpr (!producer) if (wm == half) {
      const int cl = lane & 15, rq = lane >> 4;
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) tile[(i * 16 + rq * 4 + e) * WTP + wn * 64 + j * 16 + cl] = acc[i][j][e];
    }
    if (hashed && tid < 80) rkeys[tid] = drop_row_key(dseed, (uint64_t)(m0 + half * 80 + tid));
    PE_MARK(1);
    __syncthreads();
    PE_MARK(2);
#pragma unroll 1
    for (int r = r0w; r < 80; r += W_CONS + W_PROD) {
      const int64_t m = m0 + half * 80 + r;
      if (m >= g.N) break;
      const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * WTP + c4);
      float v[4] = {a[0] + bias[0], a[1] + bias[1], a[2] + bias[2], a[3] + bias[3]};
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (H.drop_mask) {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(H.drop_mask + m * g.E + n);
        ks[0] = mk.x ? inv_keep : 0.f; ks[1] = mk.y ? inv_keep : 0.f; ks[2] = mk.z ? inv_keep : 0.f; ks[3] = mk.w ? inv_keep : 0.f;
      }
#ifndef PW_EPI_NOHASH
      else if (hashed) {
        const uint32_t rk = rkeys[r];
        const uint32_t h0 = pw_pair_hash(rk, (uint32_t)(n >> 1)), h1 = pw_pair_hash(rk, (uint32_t)(n >> 1) + 1u);
        ks[0] = (h0 & 0xffffu) >= thr16 ? inv_keep : 0.f;
        ks[1] = (h0 >> 16) >= thr16 ? inv_keep : 0.f;
        ks[2] = (h1 & 0xffffu) >= thr16 ? inv_keep : 0.f;
        ks[3] = (h1 >> 16) >= thr16 ? inv_keep : 0.f;
      }
#endif
#ifdef PW_EPI_NOACT
      if (dact) {
        pw_h4 d;
        for (int q = 0; q < 4; ++q) { d[q] = (_Float16)(v[q] * ks[q]); v[q] = v[q] * ks[q]; }
#ifdef MHIMX_PROJ_WT2
        st_b8_wt(dact + m * g.E + n, __builtin_bit_cast(f32x2_wt, d));
#else
        *reinterpret_cast<pw_h4*>(dact + m * g.E + n) = d;
#endif
      } else
#endif
      if (dact) {
        pw_h4 d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float y, gq;
          act_fwd_grad(v[q], g.act, y, gq);                   // d out / d pre for the backward (shares the erf)
          v[q] = y * ks[q];
          d[q] = (_Float16)(gq * ks[q]);
        }
#ifdef MHIMX_PROJ_WT2
        st_b8_wt(dact + m * g.E + n, __builtin_bit_cast(f32x2_wt, d));
#else
        *reinterpret_cast<pw_h4*>(dact + m * g.E + n) = d;
#endif
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], g.act) * ks[q];
      }
      if (H.resid) {
        const f32x4 rr = rq0;
        rq0 = rq1; rq1 = rq2;
        rq2 = ld_res(r + 3 * (W_CONS + W_PROD));
        v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
      }
#if defined(MHIMX_PROJ_WT) || defined(MHIMX_PROJ_WT2)
      st_f4_wt(H.H + m * H.ldh + n, f32x4{v[0], v[1], v[2], v[3]});
#else
      *reinterpret_cast<f32x4*>(H.H + m * H.ldh + n) = f32x4{v[0], v[1], v[2], v[3]};
#endif
    }
    PE_MARK(3);
  }
#ifdef PW_PROF
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t pf_t3a = __builtin_readcyclecounter();
    __syncthreads();
    if (blockIdx.x == PW_PROF_BLOCK && pf_iter == PW_PROF_TILE && lane == 0) {
      const uint64_t pf_t3 = __builtin_readcyclecounter();
      for (int i = 0; i < 8; ++i) H.H[(m0 + wave) * H.ldh + i] = (float)(PW_PROF == 2 ? pe[i] : pf[i]);
      H.H[(m0 + wave) * H.ldh + 8] = (float)(pf_t1 - pf_t0);      // tile top -> main loop
      H.H[(m0 + wave) * H.ldh + 9] = (float)(pf_t2 - pf_t1);      // main loop
      H.H[(m0 + wave) * H.ldh + 10] = (float)(pf_t3 - pf_t2);     // epilogue
      H.H[(m0 + wave) * H.ldh + 11] = (float)(pf_t3a - pf_t);     // the wave's last stores leave
      H.H[(m0 + wave) * H.ldh + 12] = (float)m0;
      H.H[(m0 + wave) * H.ldh + 13] = (float)(wall_clock64() - pf_rt0);   // tile top -> here, in 10 ns ticks
      H.H[(m0 + wave) * H.ldh + 14] = (float)(pf_t3 - pf_t0);             // the same span in shader cycles
    }
    ++pf_iter;
  }
#endif
  };

  if (producer) {
    // ================================================================= producers
    // (the launch's first tile is peeled off the loop: its prologue requests everything itself, a later tile's finds its first stages in LDS -
    // as two branches of one loop body the listing was no straight line any more and tools/asm_lint.py could not follow the registers in flight)
    ARegs ra, rb;                                                           // ra: odd tiles, rb: even tiles
    auto tile_body = [&](const TileCtx& t, const float* pxu) -> bool {      // the k loop and the epilogue; true: another tile follows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // A(0) is in LDS
    slot_end();                                                             // ---- tile 0 complete
    // running state of k-step s: st1 = byte offset of the stage of tile s+1 (A(s+1) is stored there), st2 = of tile s+2 (B(s+2) lands there)
    unsigned st1 = 0, st2 = WSTAGE;
    const char* bcur = bbase + 2 * 128;                                     // -> B(s+2)
    const char* const bdummy = bbase;      // past the last tile the same COUNT of requests re-reads k-step 0 (uniform vmcnt; a stage nobody reads)
    // Both slots of a k-step carry half of the work: units 0..2 / 3..4 of A(s+1) (split + stores) and of the A(s+3) request, pieces 0..3 /
    // 4..7 of B(s+2).  After the odd slot: wait until only the 13 youngest requests are in flight (this k-step's A(s+3) [5] and B(s+2) [8]):
    // A(s+2) - requested a k-step ago - is in its registers and B(s+1) has landed.
    auto kstep = [&](int s, ARegs& r, ARegs& r_next) {                       // r: A(s+1), arrived; r_next: A(s+2), in flight
      const bool st_ok = s + 1 < nk, b_ok = s + 2 < nk;
      const float* xn = pxu + (int64_t)(s + 3 < nk ? s + 3 : nk - 1) * WBK;
      const char* bk = b_ok ? bcur : bdummy;
      // ---- slot 2s
      PW_START();
      if (st_ok) {
        split_store(r.v[0], a_hi[0] + st1, a_lo[0] + st1);
        split_store(r.v[1], a_hi[1] + st1, a_lo[1] + st1);
        split_store(r.v[2], a_hi[2] + st1, a_lo[2] + st1);
      }
      __builtin_amdgcn_sched_barrier(0);
      PW_MARK(0);
      load_a3(xn, r);                                                       // (issued after the split read the registers)
      issue_b(bk, st2, 0);
      PW_MARK(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PW_MARK(2);
      slot_end();
      PW_MARK(3);
      // ---- slot 2s+1
      if (st_ok) {
        split_store(r.v[3], a_hi[3] + st1, a_lo[3] + st1);
        split_store(r.v[4], a_hi[4] + st1, a_lo[4] + st1);
      }
      __builtin_amdgcn_sched_barrier(0);
      PW_MARK(4);
      load_a2(xn, r);
      issue_b(bk, st2, 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PW_MARK(5);
      asm volatile("s_waitcnt vmcnt(13)" : PW_NAME5(r_next) : : "memory");
      PW_MARK(6);
      slot_end();
      PW_MARK(7);
      st1 = st2;
      st2 = st2 == 2 * WSTAGE ? 0u : st2 + WSTAGE;
      bcur += 128;
    };
#ifdef PW_PROF
    pf_t1 = __builtin_readcyclecounter();
#endif
    int s = 0;
#pragma unroll 1
    for (; s + 1 < nk; s += 2) {
      kstep(s, ra, rb);
      kstep(s + 1, rb, ra);
    }
    if (s < nk) kstep(s, ra, rb);
    slot_end();                                                             // slot 2 nk: group 1's last compute phase
    asm volatile("s_waitcnt vmcnt(0)" : PW_NAME5(ra), PW_NAME5(rb) : : "memory");
    if constexpr (PLAIN) return t.vb_next < total;                          // (the consumers store their registers themselves)
    if (PW_PROD_PRIO) __builtin_amdgcn_s_setprio(0);                        // the epilogue shares its rows evenly among all twelve waves
#ifdef PW_PROF
    pf_t2 = __builtin_readcyclecounter();
#endif
    f32x4 acc0[NRA][NRB];                                                   // (the producers hold no outputs; the scored epilogue reads zeros)
    if constexpr (SCORED) {
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t.hd == 0) {                                      // model 0: scorer + pool partials instead of the feature rows
        pw_scored_epilogue<SCORED == 2>(g, t.H, sc, acc0, smem, t.m_tile, t.m0, t.n0);   // (scored launches are not persistent: one tile per workgroup)
        return false;
      }
    }
    epilogue(std::true_type{}, acc0, t);
    if (t.vb_next >= total) return false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next tile's first stages have landed (and this tile's stores are out)
    __syncthreads();                                          // the staging rows are read: B(1) of the next tile may land in stage 0
    return true;
    };
    p_setup(vb);
    TileCtx t = tile_ctx(vb);
#ifdef PW_PROF
    pf_t0 = __builtin_readcyclecounter(); pf_rt0 = wall_clock64();
#endif
    if (PW_PROD_PRIO) __builtin_amdgcn_s_setprio(PW_PROD_PRIO);
    const float* pxu = uni(px);
    auto prologue_direct = [&](const float* pxu_) {
      // B(0), B(1) requested, A(0) -> stage 2 (the compiler's wait in front of the conversion drains the DMA pieces too), A(1), A(2)
      // requested into the two register sets; A(1) waited for.  (The launch's first tile; every tile of a PLAIN launch.)
      issue_b(bbase, 2 * WSTAGE, 0);
      issue_b(bbase, 2 * WSTAGE, 4);
      issue_b(bbase + (nk > 1 ? 128 : 0), 0, 0);
      issue_b(bbase + (nk > 1 ? 128 : 0), 0, 4);
      {
        const char* xb = reinterpret_cast<const char*>(px);
#pragma unroll
        for (int j = 0; j < 5; ++j) split_store(*reinterpret_cast<const f32x4*>(xb + aoff[j]), a_hi[j] + 2 * WSTAGE, a_lo[j] + 2 * WSTAGE);
      }
      const float* x1 = pxu_ + (nk > 1 ? 1 : 0) * WBK;
      const float* x2 = pxu_ + (nk > 2 ? 2 : nk - 1) * WBK;
      load_a3(x1, ra); load_a2(x1, ra);
      load_a3(x2, rb); load_a2(x2, rb);
      asm volatile("s_waitcnt vmcnt(5)" : PW_NAME5(ra) : : "memory");        // A(1) is here
    };
    prologue_direct(pxu);
    bool more = tile_body(t, pxu);
    if constexpr (SCORED) more = false;                       // (scored launches: one tile per workgroup, the loop is not instantiated)
#pragma unroll 1
    while (more) {
      vb = t.vb_next;
      t = tile_ctx(vb);
#ifdef PW_PROF
      pf_t0 = __builtin_readcyclecounter(); pf_rt0 = wall_clock64();
      for (int i = 0; i < 8; ++i) pf[i] = 0;
#endif
      if (PW_PROD_PRIO) __builtin_amdgcn_s_setprio(PW_PROD_PRIO);
      if constexpr (PLAIN) {                                  // nothing was requested ahead: the consumers' stores cover this prologue
        p_setup(vb);
        pxu = uni(px);
        prologue_direct(pxu);
        more = tile_body(t, pxu);
        continue;
      }
      pxu = uni(px);
      // a later tile: B(0) and the raw rows of A(0), A(1) landed during the previous epilogue (its closing wait and barrier).  A(1) -> ra,
      // A(0) converted in place, A(2) requested, B(1) -> stage 0 (which held the staging rows until that barrier)
      ARegs r0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        r0.v[j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(lds0 + 2 * WSTAGE + (unsigned)(pt + 256 * j) * 16));
        ra.v[j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(lds0 + RAW1_OFF + (unsigned)(pt + 256 * j) * 16));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : PW_NAME5(r0), PW_NAME5(ra) : : "memory");
      const float* x2 = pxu + (nk > 2 ? 2 : nk - 1) * WBK;
      load_a3(x2, rb); load_a2(x2, rb);
#pragma unroll
      for (int j = 0; j < 5; ++j) split_store(r0.v[j], a_hi[j] + 2 * WSTAGE, a_lo[j] + 2 * WSTAGE);
      issue_b(bbase + (nk > 1 ? 128 : 0), 0, 0);
      issue_b(bbase + (nk > 1 ? 128 : 0), 0, 4);
      more = tile_body(t, pxu);
    }
  } else {
#pragma unroll 1
    for (;;) {
    const TileCtx t = tile_ctx(vb);
    f32x4 acc[NRA][NRB];
#pragma unroll
    for (int i = 0; i < NRA; ++i)
#pragma unroll
      for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef PW_PROF
    pf_t0 = __builtin_readcyclecounter(); pf_rt0 = wall_clock64();
    for (int i = 0; i < 8; ++i) pf[i] = 0;
#endif
    // ================================================================= consumers: fragment reads and MFMAs only
    const int r16 = lane & 15, kg = lane >> 4;
    const int ra_ = wm * 80 + r16, rb_ = wn * 64 + r16;
    const unsigned fa_hi = lds0 + ra_ * 128 + (((2 * kg) ^ mt_swz(ra_)) << 4);
    const unsigned fa_lo = lds0 + ra_ * 128 + (((2 * kg + 1) ^ mt_swz(ra_)) << 4);
    const unsigned fb_hi = lds0 + WA_BYTES + rb_ * 128 + (((2 * kg) ^ mt_swz(rb_)) << 4);
    const unsigned fb_lo = lds0 + WA_BYTES + rb_ * 128 + (((2 * kg + 1) ^ mt_swz(rb_)) << 4);
    f32x4 x[NFR];
    auto load_phase = [&](unsigned so) {
      MT_READ9(x, 5, 10, fa_lo + so, fb_hi + so);
      MT_READ9(x, 0, 14, fa_hi + so, fb_lo + so);
      MT_WAIT9(0, x, 5, 10);
      MT_WAIT9(0, x, 0, 14);
    };
    auto compute_phase = [&]() {
#ifdef PW_NOMMA
      return;
#endif
      mt_term(x, 5, 10, acc);                                               // lo*hi
      mt_term(x, 0, 14, acc);                                               // hi*lo
      mt_term(x, 0, 10, acc);                                               // hi*hi
    };
    slot_end();                                                             // ---- tile 0 complete (producers' prologue)
#ifdef PW_PROF
    pf_t1 = __builtin_readcyclecounter();
#endif
#ifdef PW_G1_PRIO
    if (wm != 0) __builtin_amdgcn_s_setprio(PW_G1_PRIO);
#endif
    const bool late = wm != 0;                                              // group 1 runs the same loop one slot later
    if (late) slot_end();
    unsigned so = 2 * WSTAGE;                                               // tile s lives in ring stage (s + 2) % 3
#pragma unroll 1
    for (int s = 0; s < nk; ++s) {
      PW_START();
      load_phase(so);      PW_MARK(0);   slot_end();   PW_MARK(1);
      compute_phase();     PW_MARK(2);   slot_end();   PW_MARK(3);
      so = so == 2 * WSTAGE ? 0u : so + WSTAGE;
    }
    if (!late) slot_end();
#ifdef PW_G1_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef PW_PROF
    pf_t2 = __builtin_readcyclecounter();
#endif
    if constexpr (SCORED) {
      if (t.hd == 0) {
        pw_scored_epilogue<SCORED == 2>(g, t.H, sc, acc, smem, t.m_tile, t.m0, t.n0);
        return;
      }
    }
    if constexpr (PLAIN) {
      // registers -> global: lane (cl, rq) holds rows rq * 4 + e of the 16-row blocks i and column cl of the 16-column blocks j; one store
      // instruction = 4 rows x 64 B.  No barrier: the producers' next prologue is running, the ring is theirs.
      const int cl = lane & 15, rq = lane >> 4;
      const int64_t ncol = t.n0 + wn * 64 + cl;
      float bias4[NRB];
#pragma unroll
      for (int j = 0; j < NRB; ++j) bias4[j] = t.H.bias ? t.H.bias[ncol + j * 16] : 0.f;
      float* orow = t.H.H + (t.m0 + wm * 80 + rq * 4) * t.H.ldh + ncol;
      const int64_t mrow = t.m0 + wm * 80 + rq * 4;
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (mrow + i * 16 + e < g.N) {
            float* o = orow + (int64_t)(i * 16 + e) * t.H.ldh;
#pragma unroll
            for (int j = 0; j < NRB; ++j) o[j * 16] = acc[i][j][e] + bias4[j];
          }
        }
      if (t.vb_next >= total) break;
      vb = t.vb_next;
      continue;
    }
    epilogue(std::false_type{}, acc, t);
    if constexpr (SCORED) break;
    if (t.vb_next >= total) break;
    __syncthreads();
    vb = t.vb_next;
    }
  }
#undef PW_NAME5
}

// out = x * keep / (1 - p) with the PROJECTION kernels' dropout stream (one 32-bit mix per pair of columns, 16-bit thresholds: the
// epilogue above): a backward re-applies the forward's mask to the incoming gradient, no mask is stored (mhimx_dropout_apply_proj).
__global__ __launch_bounds__(256) void proj_dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M, int E, float p,
                                                                uint64_t seed0, const uint64_t* __restrict__ tick) {
  const uint64_t seed = eff_seed(seed0, tick);
  const uint32_t thr16 = (uint32_t)(p * 65536.f + 0.5f);
  const float inv_keep = 65536.f / (float)(65536u - thr16);
  const int e4 = E >> 2;
  const int64_t n4 = M * e4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / e4;
    const int n = (int)(i - m * e4) * 4;
    const uint32_t rk = drop_row_key(seed, (uint64_t)m);
    const uint32_t h0 = pw_pair_hash(rk, (uint32_t)(n >> 1)), h1 = pw_pair_hash(rk, (uint32_t)(n >> 1) + 1u);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + m * E + n);
    v[0] = (h0 & 0xffffu) >= thr16 ? v[0] * inv_keep : 0.f;
    v[1] = (h0 >> 16) >= thr16 ? v[1] * inv_keep : 0.f;
    v[2] = (h1 & 0xffffu) >= thr16 ? v[2] * inv_keep : 0.f;
    v[3] = (h1 >> 16) >= thr16 ? v[3] * inv_keep : 0.f;
    *reinterpret_cast<f32x4*>(out + m * E + n) = v;
  }
}

int bag_project_ws(hipStream_t st, const mhimx_bag_project_args* bags, int n_bags) {
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_project_ws_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WNST * WSTAGE));
                        MHIMX_HIP(hipFuncSetAttribute((const void*)bag_project_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WNST * WSTAGE));
                        MHIMX_HIP(hipFuncSetAttribute((const void*)bag_project_ws_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WNST * WSTAGE));
                        MHIMX_HIP(hipFuncSetAttribute((const void*)bag_project_ws_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WNST * WSTAGE)));
  const mhimx_bag_project_args& g = bags[0];
  ProjScore sc = {};
  if (g.score0) {
    const mhimx_proj_score& q = *g.score0;
    MHIMX_CHECK_ARG(n_bags == 1 && g.E == 512 && q.wa16 && q.wc && q.s && q.pm && q.pl && q.pz && q.xch && q.gate && q.C >= 0 && q.C <= 16 &&
                        (q.C == 0 || q.cproj) && aligned16(q.wa16) && !g.head[0].resid && !g.head[0].dact && !g.head[0].drop_mask,
                    "bag_project: the scored model 0 needs E = 512, one bag, every output of mhimx_proj_score, no residual rows / dact / injected mask");
    static_assert(81920 + 1024 + (8 * 160 + 160 + 160 + 512 + 16) * 4 <= WNST * WSTAGE, "the scored epilogue's LDS fits the ring");
    sc = ProjScore{q.wa16, q.wc, q.act, q.C, q.s, q.cproj, q.pm, q.pl, q.pz, q.xch, q.gate};
  }
  const int nN = (int)(g.n_heads * g.E / WBN), nM = (int)cdiv(g.N, WBM);
  ProjBags pb = {};
  pb.n_bags = n_bags;
  pb.tiles_per_bag = (int)(8 * cdiv(nM, 8));
  {
    // half of a steady-state tile's life in shader cycles: k-step ~2.5 k, prologue + epilogue ~22 k (profiles/r05_store_path.md);
    // MHIMX_PROJ_STAGGER=0 switches the offset off, another value replaces the estimate
    static const int stagger_env = [] { const char* e = getenv("MHIMX_PROJ_STAGGER"); return e ? atoi(e) : -1; }();
    pb.stagger = stagger_env >= 0 ? stagger_env : (int)((g.D / WBK) * 1250 + 11000);
    static const int spread_env = [] { const char* e = getenv("MHIMX_PROJ_SPREAD"); return e ? atoi(e) : 1; }();
    pb.stagger_spread = spread_env;         // (c3 same-box: 8.523 -> 8.479 ms against the one half-tile offset, which itself measured 0 to -1 %)
  }
  for (int b = 0; b < n_bags; ++b) {
    pb.X[b] = bags[b].X;
    for (int h = 0; h < g.n_heads; ++h) {
      pb.H[b][h] = bags[b].head[h].H;
      pb.dact[b][h] = bags[b].head[h].dact;
      pb.seed[b][h] = bags[b].head[h].drop_seed;
    }
  }
  // persistent workgroups (one per CU; the grid stays a multiple of 8: XCD mapping) unless the launch is scored (its epilogue pairs workgroups
  // through a gate: every tile needs its partner resident) or MHIMX_PROJ_PERSIST=0 asks for one workgroup per tile
  static const bool persist = [] { const char* e = getenv("MHIMX_PROJ_PERSIST"); return !(e && e[0] == '0'); }();
  unsigned nblocks = (unsigned)(nN * pb.tiles_per_bag * n_bags);
  if (persist && !g.score0 && nblocks > W_CUS) nblocks = W_CUS;
  dim3 grid(nblocks);
  bool plain = !g.score0 && g.act == MHIMX_ACT_NONE;
  for (int b = 0; b < n_bags && plain; ++b)
    for (int h = 0; h < g.n_heads; ++h) {
      const mhimx_proj_head& q = bags[b].head[h];
      if (q.drop_p > 0.f || q.drop_mask || q.resid || q.dact) plain = false;
    }
  static const bool plain_ok = [] { const char* e = getenv("MHIMX_PROJ_PLAIN"); return !(e && e[0] == '0'); }();
  if (plain && plain_ok) hipLaunchKernelGGL(bag_project_ws_kernel<3>, grid, dim3(WTHREADS), WNST * WSTAGE, st, g, pb, sc);
  else if (g.score0 && g.head[0].H) hipLaunchKernelGGL(bag_project_ws_kernel<2>, grid, dim3(WTHREADS), WNST * WSTAGE, st, g, pb, sc);
  else if (g.score0) hipLaunchKernelGGL(bag_project_ws_kernel<1>, grid, dim3(WTHREADS), WNST * WSTAGE, st, g, pb, sc);
  else hipLaunchKernelGGL(bag_project_ws_kernel<0>, grid, dim3(WTHREADS), WNST * WSTAGE, st, g, pb, sc);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx

extern "C" int64_t mhimx_proj_score_parts(int64_t N) { return (N + mhimx::WBM - 1) / mhimx::WBM; }
extern "C" int64_t mhimx_proj_score_xch_floats(int64_t N) { return mhimx_proj_score_parts(N) * 2 * 160 * mhimx::PS_UCOLS; }

extern "C" int mhimx_dropout_apply_proj(void* stream, const float* x, float* out, int64_t M, int64_t E, float p, uint64_t seed, const uint64_t* tick) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(x && out && M >= 0 && E > 0 && E % 4 == 0 && p > 0.f && p < 1.f && aligned16(x) && aligned16(out), "dropout_apply_proj: bad args");
  if (M == 0) return 0;
  const int64_t blocks = cdiv(M * (E / 4), 256);
  hipLaunchKernelGGL(proj_dropout_apply_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, out, M, (int)E, p,
                     seed, tick);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
