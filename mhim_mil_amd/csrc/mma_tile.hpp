// mma_tile.hpp — the 80 x 64 per-wave bf16x3 tile step shared by the paired-plane kernels (bag_project.hip, wgrad.hip).
//
// LDS image of an operand tile: [rows][128 B] = one 32-deep k-step; per 8 consecutive k, 16 B of bf16 hi then 16 B of bf16 lo
// (mhimx_pair_planes), the 16-byte slot index XOR-ed with (row >> 1) & 7.  A lane of a 16x16x32 MFMA reads row (lane & 15) of a
// 16-row block and k-group (lane >> 4): slot 2kg is its hi fragment, slot 2kg+1 its lo fragment.  Blocks are 2 KiB apart and the
// swizzle does not depend on the block index, so the blocks of one operand are immediate offsets from ONE per-lane address.
#pragma once
#include "common.hpp"

namespace mhimx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// a 16-byte store that goes THROUGH the XCD's L2 to memory as it is issued (sc1: the line is not kept dirty in L2) - for outputs the next
// LAUNCH reads: a kernel ends with a write-back of every dirty L2 line (MI355X_MICROARCH.md "boundary", "publish-large").
// MEASURED, round 6 (side builds -DMHIMX_PROJ_WT2 / -DMHIMX_IMG_WT / -DMHIMX_SLAB_WT: the projection's feature rows + d out / d pre, the dPRE
// image, the weight gradient's split-K slabs; parity tests green): the producers get shorter (projection 65.0 -> 63.1 us, image 13.7 -> 12.2,
// weight gradient 46.2 -> 45.5) and their consumers longer by the same (teacher scorer 21.1 -> 23.4, Adam 18.1 -> 19.2): one step on the
// timeline 308.7 -> 309.5 us under the profiler, 0.3084-0.3117 -> 0.3051-0.3054 ms in `tools/ab.sh` (inside the box's run-to-run spread).
// The write-back a launch boundary pays is not saved, it moves.  Not enabled.
// (the s_nop: a VMEM store of more than 8 bytes must not be followed at once by a VALU write of its data registers - the compiler's hazard
// recogniser does not look into asm, and without the wait states the next store's address arithmetic overwrote the data: NaN images)
__device__ __forceinline__ void st_f4_wt(float* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory"); }
typedef float f32x2_wt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_b8_wt(void* p, const f32x2_wt& v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_f;
typedef __attribute__((address_space(3))) void* lptr_f;

// Bank swizzle of the 16-byte slot index inside a 128-byte row.  A ds_read_b128 is served in four groups of 16 lanes and the groups are
// NOT the four quarters of the wave: they are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}
// (MI355X_MICROARCH.md, LDS).  With lane = 16 kg + r a group therefore holds rows 0-3 and 12-15 of one k-group and rows 4-11 of the
// NEIGHBOURING k-group (slot index ^ 2).  The 16 lanes must hit 16 different 16-byte slots of the 256-byte bank row, i.e. the 8 even
// (and the 8 odd) rows need 8 different swizzled slots: (row >> 1) & 7 alone makes every inner row collide with an outer one (2-way
// conflicts on every fragment read); flipping bit 1 for the inner rows restores a bijection.
MHIMX_DEV int mt_swz(int row) { return ((row >> 1) & 7) ^ (((((row & 15) + 4) >> 3) & 1) << 1); }

constexpr int NRA = 5, NRB = 4, NFR = 2 * (NRA + NRB);      // fragments of one k-step: x[0..4] A hi, x[5..9] A lo, x[10..13] B hi, x[14..17] B lo

// five 16-row blocks of operand A into x[oa..oa+4] and four of operand B into x[ob..ob+3]
#define MT_READ9(x, oa, ob, a, b)                                                                                      \
  asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:2048\n\tds_read_b128 %2, %9 offset:4096\n\t"         \
               "ds_read_b128 %3, %9 offset:6144\n\tds_read_b128 %4, %9 offset:8192\n\t"                                \
               "ds_read_b128 %5, %10\n\tds_read_b128 %6, %10 offset:2048\n\tds_read_b128 %7, %10 offset:4096\n\t"      \
               "ds_read_b128 %8, %10 offset:6144"                                                                      \
               : "=&v"(x[oa]), "=&v"(x[oa + 1]), "=&v"(x[oa + 2]), "=&v"(x[oa + 3]), "=&v"(x[oa + 4]), "=&v"(x[ob]),       \
                 "=&v"(x[ob + 1]), "=&v"(x[ob + 2]), "=&v"(x[ob + 3])                                                    \
               : "v"(a), "v"(b)                                                                                        \
               : "memory")
// wait until at most `n` LDS operations are outstanding; names the 9 registers the following MFMAs read
#define MT_WAIT9(n, x, oa, ob)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                             \
               : "+v"(x[oa]), "+v"(x[oa + 1]), "+v"(x[oa + 2]), "+v"(x[oa + 3]), "+v"(x[oa + 4]), "+v"(x[ob]),             \
                 "+v"(x[ob + 1]), "+v"(x[ob + 2]), "+v"(x[ob + 3])                                                       \
               :                                                                                                       \
               : "memory")

MHIMX_DEV f32x4 mt_mfma(const f32x4& a, const f32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// 20 MFMAs of one term: A blocks x[oa..oa+4] times B blocks x[ob..ob+3]
MHIMX_DEV void mt_term(const f32x4 (&x)[NFR], int oa, int ob, f32x4 (&acc)[NRA][NRB]) {
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = mt_mfma(x[oa + i], x[ob + j], acc[i][j]);
}

// The 60 MFMAs of one k-step after MT_READ9(x, 5, 10, A lo, B hi); MT_READ9(x, 0, 14, A hi, B lo) were the last two LDS operations
// issued: the lo*hi term starts as soon as the first nine fragments are here.  Term-major: 20 independent MFMAs between two that
// touch the same accumulator.
MHIMX_DEV void mt_mma_3term(f32x4 (&x)[NFR], f32x4 (&acc)[NRA][NRB]) {
  MT_WAIT9(9, x, 5, 10);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = mt_mfma(x[5 + i], x[10 + j], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);                          // (the second wait must not be hoisted over the first term)
  MT_WAIT9(0, x, 0, 14);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = mt_mfma(x[i], x[14 + j], acc[i][j]);
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = mt_mfma(x[i], x[10 + j], acc[i][j]);
}

}  // namespace mhimx
