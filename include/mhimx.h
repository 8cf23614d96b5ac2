/* mhimx.h — C-ABI of libmhimx.so: the MI355X-native (gfx950) MHIM aggregation path.
 *
 * The reference (DearCaat/MHIM-MIL) is pure Python/PyTorch: it has no FFI seam for
 * this path.  The boundary is therefore NEW; each entry point
 * below cites the reference function(s) it replaces (file:line into the reference
 * tree).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless the name says otherwise
 *    (ids/rows/perm are int64 like torch index tensors; all must be 16-byte aligned
 *    where they are matrices);
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - every function only ENQUEUES work on `stream` (no host sync, no allocation): the
 *    caller owns all buffers including the workspace;
 *  - return value: 0 ok; <0 argument/shape error; >0 a hipError_t.  The message is
 *    retrievable with mhimx_last_error() (thread-local);
 *  - nothing throws across this boundary.
 */
#ifndef MHIMX_H
#define MHIMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an entry point or a struct changes shape; the ctypes binding (mhim_mil_amd/_lib.py ABI_VERSION) refuses any other value */
#define MHIMX_VERSION 610

/* activations (feature act: mhim.py:71-74 relu|gelu|none; scorer act: baseline.py:17-22 gelu|relu|tanh|none) */
enum { MHIMX_ACT_NONE = 0, MHIMX_ACT_RELU = 1, MHIMX_ACT_GELU = 2, MHIMX_ACT_TANH = 3 };

/* matrix-core precision of a GEMM:
 *  F32   : v_mfma_f32_32x32x2_f32, exact fp32 (bitwise an fmaf chain), 1/16 of the 16-bit rate
 *  F16S  : A rounded to one fp16 term, B (the weight) split into fp16 hi+lo: 2 MFMAs / tile-step
 *  BF16X3: A and B both split into bf16 hi+lo, 3 MFMAs (hi*hi + hi*lo + lo*hi): fp32 range, ~2^-16 */
enum { MHIMX_PREC_F32 = 0, MHIMX_PREC_F16S = 1, MHIMX_PREC_BF16X3 = 2 };

const char* mhimx_last_error(void);
int mhimx_version(void);

/* ------------------------------------------------------------------------------------------
 * Building blocks (exposed so each kernel is parity-tested on its own)
 * ---------------------------------------------------------------------------------------- */

/* C[m,n] = epilogue( sum_k A[rows?rows[m]:m, k] * B[n,k] )      (torch nn.Linear: x @ W^T)
 * epilogue: v = acc; if bias v += bias[n]; if rowv v += rowv[m]*colv[n]; if pre pre[m,n] = v;
 *           v = act(v); if drop_p>0 v = keep(seed, row_id, n) ? v/(1-drop_p) : 0  (or drop_mask u8 [M,N]);
 *           if accumulate v += C[m,n]; C[m,n] = v.
 * K % 4 == 0; A,B,C 16-byte aligned with lda/ldb/ldc % 4 == 0.
 * replaces: every nn.Linear on the path, e.g. mhim.py:69 (feature), baseline.py:14 (scorer), merge.py:35-39. */
typedef struct {
  const float* A; int64_t lda; const int64_t* rows;      /* optional gather of A rows             */
  const float* B; int64_t ldb;                           /* [N,K] row-major (a Linear weight)     */
  float* C; int64_t ldc;
  int64_t M, N, K;
  const float* bias;                                     /* [N] or NULL                           */
  const float* rowv; const float* colv;                  /* optional rank-1 term                  */
  float* pre; int64_t ldpre;                             /* optional pre-activation copy          */
  int32_t act;                                           /* MHIMX_ACT_*                           */
  float drop_p; uint64_t drop_seed; const uint8_t* drop_mask;   /* dropout: hashed RNG or injected keep-mask */
  int32_t accumulate;                                    /* C += ...                              */
  int32_t prec;                                          /* MHIMX_PREC_*                          */
  const uint64_t* drop_tick;                             /* optional device step counter mixed into drop_seed (graph replay) */
  int32_t paired;                                        /* 1: A and B are paired-plane images made by mhimx_pair_planes      */
  float* ws; int64_t ws_floats;                          /* optional scratch: lets a GEMM with few output tiles and a long K   */
                                                         /* split its reduction over up to ws_floats/(M*N) slabs (epilogue-free calls) */
  float* dact; int64_t lddact;                           /* optional: d(output)/d(pre-activation) = act'(pre) * keep/(1-p) per element, */
                                                         /* so the backward through act+dropout is one multiply (mhimx_mul_colsum).      */
                                                         /* Only the paired-plane projection kernel writes it (else the call fails).      */
} mhimx_gemm_nt_args;
int mhimx_gemm_nt(void* stream, const mhimx_gemm_nt_args* a);

/* The bag projection of up to TWO models in ONE pass over the raw fp32 bag (teacher and student share the bag: the reference's
 * student computes the feature on all N rows before it masks, modules/mhim.py:335-336, and its teacher does the same on the same
 * bag, mhim.py:186):   H_g[N,E] = dropout_g( act( X[N,D] W_g[E,D]^T + b_g ) ),  g < n_heads.
 * X is plain fp32 (split to bf16 hi/lo on its way into LDS); each W_g is the paired-plane image of the weight (mhimx_pair_planes /
 * prep kind 1).  3-term bf16 (~2^-16).  dact (optional, fp16 [N,E]): d out / d pre = act'(pre) * keep/(1-p), so the backward through
 * activation + dropout is one multiply (mhimx_rows_dpre).  Dropout: counter hash of (seed + *drop_tick, row, column pair), or an
 * injected keep-mask u8 [N,E].  D % 32 == 0, E % 256 == 0.
 * replaces: mhim.py:69-76 (self.feature) applied at mhim.py:186 (teacher) and :336 (student). */
#define MHIMX_PROJ_MAX_HEADS 2
typedef struct {
  const float* wp;                    /* paired-plane image of the weight [E,D]                            */
  const float* bias;                  /* [E] or NULL                                                       */
  float* H; int64_t ldh;              /* output rows [N, ldh >= E]                                         */
  void* dact;                         /* optional fp16 [N,E]                                               */
  float drop_p; uint64_t drop_seed; const uint8_t* drop_mask;
  const float* resid; int64_t ldr;    /* optional fp32 [N, ldr >= E], added AFTER activation and dropout: H = resid + dropout(act(..)) -  */
                                      /* the residual connection around a TransLayer's to_out (baseline.py:215); not H itself; dact ignores it */
} mhimx_proj_head;
/* The teacher's scorer INSIDE the projection (round 4; modules/mhim.py:193-205 forward_teacher: feature -> DAttention -> pseudo score):
 * with this block model 0's feature rows never reach HBM.  A workgroup of model 0 holds 160 rows x 256 of the 512 feature columns after the
 * k loop; in its epilogue it applies bias / activation / dropout in registers, multiplies its half rows with the matching half of
 * [Wa ; Wp] on the matrix cores (3-term bf16), swaps the [160, 144] partial products with the workgroup that owns the other 256 columns
 * (same XCD, neighbouring block index: a pair gate in `gate`), and both finish alike: s = wc . act(u), the class projections, and the
 * log-sum-exp pool partial (max, sum, sum_r e^{s_r - max} h_r over its own 256 columns).  mhimx_pool_finalize merges the G partials.
 * head[0].H may then be NULL (non-NULL: the rows are written as well - tests).  E = 512, scorer width 128, no scorer bias, C <= 16. */
typedef struct {
  const float* wa16;                  /* prep kind-8 image of [Wa (128 rows) ; Wp (C rows)] stacked: 9 blocks of 16 rows, K = E         */
  const float* wc;                    /* [128] second scorer layer                                                                    */
  int32_t act;                        /* scorer activation (MHIMX_ACT_*)                                                              */
  int32_t C;                          /* classes whose projections h . Wp_c are wanted (0: none)                                      */
  float* s;                           /* [N] scorer outputs                                                                           */
  float* cproj;                       /* [N, C]                                                                                       */
  float* pm; float* pl; float* pz;    /* pool partials of the G = mhimx_proj_score_parts(N) row tiles: [G], [G], [G, E]               */
  float* xch;                         /* exchange scratch, mhimx_proj_score_xch_floats(N) floats                                      */
  uint32_t* gate;                     /* [G] pair counters: zero ONCE (they only count up, two per launch)                             */
} mhimx_proj_score;
/* merges G pool partials (pm, pl, pz [G, E]) into stats = {max, sum}, z [E] and - with pscore - the pseudo score of the M1 instances
 * (mhimx_pseudo_score's arithmetic): the last launch of mhimx_abmil_pool_fwd, for partials the caller holds (mhimx_proj_score's) */
int mhimx_pool_finalize(void* stream, const float* pm, const float* pl, const float* pz, int64_t G, int64_t E, float* stats, float* z,
                        const float* s, const float* cproj, const float* bp, int64_t C, int64_t M1, float* pscore);
int64_t mhimx_proj_score_parts(int64_t N);
int64_t mhimx_proj_score_xch_floats(int64_t N);
typedef struct {
  const float* X; int64_t ldx;        /* the bag [N, D] fp32                                               */
  int64_t N, D, E;
  int32_t act;                        /* MHIMX_ACT_*                                                       */
  int32_t n_heads;
  mhimx_proj_head head[MHIMX_PROJ_MAX_HEADS];
  const uint64_t* drop_tick;          /* optional device step counter mixed into every drop_seed           */
  const mhimx_proj_score* score0;     /* optional: model 0's scorer + pool partials in the epilogue (above) */
} mhimx_bag_project_args;
int mhimx_bag_project(void* stream, const mhimx_bag_project_args* a);
/* The projections of the n_bags <= 8 bags of an accumulation window (they share both models' weights: base_engine.py:100-119 steps the
 * optimiser once per window) in ONE launch: row tiles of the next bag start while the last ones of a bag drain their outputs.  The bags share
 * N, D, E, ldx, act, n_heads, drop_tick and per model wp, bias, ldh, drop_p; each brings its X, H, dact and drop_seed (no masks, no
 * residual rows). */
int mhimx_bag_project_multi(void* stream, const mhimx_bag_project_args* bags, int32_t n_bags);

/* C[m,n] (+)= alpha * sum_k A[m,k] * B[k,n]   with B row-major [K,N] (ldb = its row pitch).  Only A, lda, rows, B, ldb,
 * C, ldc, M, N, K, accumulate and prec of the argument block are read.  Both operands may be activations (attention
 * products, the Nystrom pseudo-inverse iterations, dX = dY . W).   replaces: torch.matmul / `@` on the path. */
int mhimx_gemm_nn(void* stream, const mhimx_gemm_nt_args* a, float alpha, int32_t splits, float* ws /* splits*M*N floats when splits > 1 */);
/* `batch` GEMMs of one shape in ONE launch (attention heads, the Nystrom pseudo-inverse iterations): operand b of
 * A / B / C lives at base + b*stride (floats).  mode 0: C_b = A_b B_b^T;  1: C_b = alpha A_b B_b;
 * 2: C_b = A_b^T B_b with A_b [K,M], B_b [K,N].  No gather / epilogue fields.  splits > 1 (modes 1, 2) splits the
 * reduction; ws >= batch*splits*M*N floats.   replaces: batched torch.matmul / einsum on [b h n d] tensors. */
int mhimx_gemm_batched(void* stream, int32_t mode, const mhimx_gemm_nt_args* args, int32_t batch, int64_t strideA,
                       int64_t strideB, int64_t strideC, float alpha, int32_t splits, float* ws);
/* Batches of SMALL products with an affine epilogue, one launch:  C_b (+)= ident * I + alpha * op(A_b, B_b)   (modes as above).
 * M, N multiples of 64 (<= 512), K a multiple of 32 (<= 1024), 16-byte aligned operands, 3-term bf16.  The Nystrom pseudo-inverse
 * iteration z <- 0.25 z (13 I - az (15 I - az (7 I - az)))  (nystrom_attention.py:21-25) is four of these per step; mhimx_gemm_batched
 * takes the same kernel for such shapes.   replaces: torch.matmul + the scalar * eye arithmetic around it. */
int mhimx_bmm_affine(void* stream, int32_t mode, const mhimx_gemm_nt_args* args, int32_t batch, int64_t strideA, int64_t strideB,
                     int64_t strideC, float alpha, float ident);
/* the same product with TWO outputs: C = ident I + alpha P and C2 = ident2 I + alpha2 P (K = 256; C2 laid out like C): `xz = x @ z` and
 * `7 I - xz` of nystrom_attention.py:23-25 in one launch */
int mhimx_bmm_affine2(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA, int64_t strideB,
                      int64_t strideC, float alpha, float ident, float* C2, float alpha2, float ident2);
/* A CHAIN of dependent steps over batches of 8 contiguous [256, 256] matrices in ONE launch - the six iterations of the pseudo-inverse
 * (nystrom_attention.py:21-25) are 24 products forward, 48 + 6 sums backward.  Operands are SPLIT IMAGES: per matrix, a bf16 hi plane
 * then a bf16 lo plane, [256][256] each (hi = bf16(x), lo = bf16(x - hi); 128 KiB + 128 KiB = the bytes of the fp32 matrix, heads
 * contiguous), of the matrix itself ("N") or of its transpose ("T").
 *   kind 0:  P = A B  with A = an N image of A, B = a T image of B (3-term bf16, fp32 accumulate);
 *            X1 = ident I + alpha P (+ dscale D + d2scale D2: fp32 matrices - D may be C itself; they enter as the accumulator's start value
 *            (dscale D + d2scale D2) / alpha, exact for |alpha| a power of two; a scale of 0 means 1; not together with a second output);
 *            X2 = ident2 I + alpha2 P;   any of: C = X1 (fp32), PN / PT = N / T image of X1, PN2 / PT2 = N / T image of X2.
 *   kind 1:  no product: PN / PT = images of alpha * A (+ D), A and D fp32 matrices (how a chain takes its inputs in, or sums two).
 *   kind -1: nothing (an idle slot of a two-group stage).
 * The table is steps[stage * groups + group]: the steps of one stage are independent of each other and may read whatever earlier STAGES
 * wrote.  128 x groups persistent workgroups of 1024 threads (groups = 1 or 2); the tiles of a head hand over through a per-head arrival
 * counter (write-through stores, cache-bypassing loads: no grid barrier).  Tiles are drawn as TICKETS in stage order, so a launch
 * completes whatever part of its grid is resident (other work on the GPU costs time, never the result).
 * counters: 513 uint32, ZERO before the first launch (head h: arrivals at [64 h], tickets at [64 h + 32], leave count at [64 h + 48] -
 * each head's hot words on 128-byte lines of their own); all zero again when a launch ends; [512] != 0 afterwards means a workgroup
 * gave up waiting (a backstop: the ticket order excludes it) and the outputs are invalid.  stages * groups <= 38.  A step must not
 * overwrite its own operands.
 * Round 5: the scaled addends dscale D + d2scale D2 let the pseudo-inverse iteration z' = 1/4 z (13 I - M (15 I - M (7 I - M))), M = a2 z
 * (nystrom_attention.py:21-25), also run in its EXPANDED form - three dependent levels instead of Horner's four: M = a2 z;  P = z M  ||
 * N' = -15 I + 7 M - M M;  z' = 1/4 P N' + 13/4 z (the backward: four two-product stages per iteration instead of five).  The host side
 * (nystrom.py) keeps Horner's four levels as the default and the expanded form behind MHIMX_PINV_LEVELS=3: measured equal
 * (profiles/r05_pinv_levels.md).
 *   replaces: the same torch.matmul chain as mhimx_bmm_affine, 7.4 us per product (13.7 us per pair) as launches. */
typedef struct {
  const void* A; const void* B; float* C; void* PN; void* PT; void* PN2; void* PT2; const float* D; const float* D2;
  float alpha, ident, alpha2, ident2, dscale, d2scale;
  int32_t kind;
} mhimx_bmm_step;
int mhimx_bmm_chain(void* stream, const mhimx_bmm_step* steps, int32_t stages, int32_t groups, uint32_t* counters);
/* two INDEPENDENT batches of 256 x 256 x 256 products in one launch (contiguous [batch, 256, 256] operands, stride = 65536): the
 * backward of a pseudo-inverse iteration is four such pairs (nystrom_attention.py:21-26 under autograd). */
int mhimx_bmm_affine_pair(void* stream, int32_t mode0, const mhimx_gemm_nt_args* a0, float alpha0, float ident0, int32_t mode1,
                          const mhimx_gemm_nt_args* a1, float alpha1, float ident1, int32_t batch, int64_t stride);

/* Deferred final reductions.  Weight / bias gradients are consumed only by the optimizer, so the last stage of their
 * two-stage reductions (summing split-GEMM slabs, summing per-block column partials) need not run where it is produced:
 * a producer given a list appends a job instead of launching its reduce kernel, and mhimx_reduce_flush runs every pending
 * job in ONE launch (each tiny launch is ~5 us on the step's serial chain; a train step has six of these).  The partial
 * buffers (workspaces passed to the producers) must stay untouched until the flush.
 *   kind 0: out[j] (+)= sum_{b<G} parts[b*ld + j], j < W        (per-block column partials)
 *   kind 1: out[i*ldo + j] (+)= sum_{z<G} parts[z*K1*K2 + i*K2 + j]   (split-GEMM slabs) */
typedef struct {
  int32_t kind; int32_t accumulate;
  const float* parts; float* out;
  int64_t G, W, ld, K1, K2, ldo;
} mhimx_reduce_job;
#define MHIMX_REDUCE_MAX 16
/* Deferred side work.  The tail of the Merge backward (merging its pooled-row partials, rank-k weight gradients: three small dependent
 * launches) feeds nothing but the optimiser either.  With a list, mhimx_merge_bwd parks it here (an opaque argument block) and the later
 * launches of the same backward that take the list give its stages a ride as extra workgroups: stage 1 in mhimx_rows_dpre, stage 2 in the
 * projection's weight-gradient mhimx_gemm_tn, stage 3 in mhimx_reduce_flush; mhimx_reduce_flush first launches whatever got no ride.
 * pending: 0 = nothing parked, else the next stage to run. */
#define MHIMX_SIDE_BYTES 512
typedef struct { int32_t pending; int32_t reserved; unsigned char blob[MHIMX_SIDE_BYTES]; } mhimx_side_work;
/* A parked GEMM: with a list, mhimx_abmil_pool_bwd does not launch its scorer-weight gradient GEMM (d_wa = du^T T, ~12 us, needed by the
 * optimiser only) but parks its arguments here; the Merge backward that follows (mhimx_merge_bwd with the same list) launches it with
 * its own first, parameter-only stage riding along as extra workgroups - one launch instead of two on the serial chain; without a Merge
 * backward mhimx_reduce_flush launches it.  blob = a mhimx_gemm_tn_args. */
typedef struct { int32_t pending; int32_t reserved /* > 0: size the product as if reserved - 1 Merge row tiles shared its launch (the
                                                        chain form's slab count - mhimx_step_run's DAG form keeps the chain's bits) */;
                 unsigned char blob[128]; } mhimx_parked_gemm;
/* pre (round 5): the FIRST stage of a Merge backward (parameters x dz, where dz = the merged tokens' gradient rows that the pool backward
 * produces) parked by mhimx_merge_bwd_park BEFORE mhimx_abmil_pool_bwd: the pool backward's one-pass rows launch gives it a ride behind a
 * gate on the row tile(s) that hold dz, and mhimx_merge_bwd then finds it done (pending: 0 nothing, 1 parked, 2 it rode). */
typedef struct { mhimx_reduce_job j[MHIMX_REDUCE_MAX]; int32_t n; mhimx_side_work side; mhimx_parked_gemm parked; mhimx_side_work pre; } mhimx_reduce_list;
int mhimx_reduce_flush(void* stream, mhimx_reduce_list* list);      /* no-op when list->n == 0; list->n = 0 on return */

/* C[i,j] = sum_m A[m,i] * B[rows?rows[m]:m, j]   (weight gradients dW = dY^T X), reduction split over
 * `splits` slabs: ws must hold splits*K1*K2 floats when splits>1 (deterministic two-stage reduction).
 * accumulate: C += result.   replaces: autograd of every nn.Linear weight on the path. */
typedef struct {
  const float* A; int64_t lda;                           /* [M,K1]                                */
  const float* B; int64_t ldb; const int64_t* rows;      /* [*,K2], optional row gather           */
  float* C; int64_t ldc;                                 /* [K1,K2]                               */
  int64_t M, K1, K2;
  int32_t splits; float* ws;
  int32_t accumulate;
  int32_t prec;
  int64_t ws_floats;                                     /* capacity of ws; the library may raise `splits` up to it   */
  mhimx_reduce_list* defer;                              /* optional: queue the slab reduction instead of launching it */
} mhimx_gemm_tn_args;
int mhimx_gemm_tn(void* stream, const mhimx_gemm_tn_args* a);

/* Paired planes: out[m, k] ("floats", same shape and pitch as x[M,K]) holds, for every 8 consecutive k of a row, 8 bf16
 * hi values followed by 8 bf16 lo values (x = hi + lo to ~2^-16).  A GEMM whose operands are both in this form
 * (args.paired = 1, MHIMX_PREC_BF16X3) moves 16-bit MFMA fragments HBM -> LDS -> matrix core with no conversion in its
 * inner loop; the bag X is paired once per step and shared by the teacher's and the student's projection. */
int mhimx_pair_planes(void* stream, const float* x, int64_t ldx, int64_t M, int64_t K, float* out);

/* The parameter-only preparation of one train step as ONE launch (each tiny kernel costs ~5 us of dispatch latency on the
 * step's serial chain): up to MHIMX_PREP_MAX jobs of kind 0 = transpose in[R,C] -> out[C,R], 1 = paired planes of in[R,C],
 * 2 = copy R*C floats, 3 = *(uint64_t*)out += 1 (the device-resident dropout / Adam step counters),
 * 4 = matrix-core fragment image of in[R,C] (R % 32 == 0, C % 16 == 0; same size as in): for every 32-row block nt and
 *     16-column step ks, 64 consecutive 32-byte items (item l: 8 bf16 hi | 8 bf16 lo of in[32 nt + l % 32][16 ks + 8 (l / 32) ..])
 *     - what mhimx_scorer.wa_frag takes,
 * 5 = the kind-4 image of the TRANSPOSE in^T [C,R] made straight from in[R,C] (C % 32 == 0, R % 16 == 0) - what
 *     mhimx_pool_grad.wa_t_frag takes (jobs of one launch run concurrently: one cannot read another's output),
 * 6 = the parameter-only part of a Merge forward (mhimx_merge_fwd with .prepared = 1 then skips it): `in` is the HOST address of the
 *     mhimx_merge block (read while enqueueing), out = the Merge workspace (device), R = rows to merge, C = workspace bytes.  Up to 8
 *     per launch when they share their parameters (the bags of an accumulation window: one workspace each),
 * 7 = paired planes of the TRANSPOSE in^T [C,R] made straight from in[R,C] (R % 8 == 0): the weight image of a data-gradient product
 *     dX = dY W on the projection kernel (the TransMIL layers' to_qkv / to_out, baseline.py:213-218),
 * 8 = the fragment image of in[R,C] for 16-row blocks and 32-deep steps (rows padded to a multiple of 16 with zeros, C % 32 == 0; out:
 *     ceil(R / 16) * 16 * C floats): item (nb, ks, lane) = 8 hi | 8 lo bf16 of in[16 nb + lane % 16][32 ks + 8 (lane / 16) ..] - what
 *     mhimx_proj_score.wa16 takes,
 * 9 = the bag in[R,C] (contiguous rows, C % 256 == 0) as the D-side operand image of the weight-gradient product (mhimx_bag_wgrad_args.ximg;
 *     out: ceil(R / 32) * 32 * C floats): per 32-row k-step ks and 256-column block cb one 32 KiB tile (ks * C / 256 + cb) laid out
 *     [row octet 4][hi | lo][column slot 256][8 bf16], column c of the block in slot (c % 4) * 64 + c / 4; rows past R are zero.  Not a
 *     parameter-only job (it reads the bag) but one that depends on nothing else: it rides in a forward launch that leaves the chip idle.
 * 10 = ((int64_t*)out)[i] = R + i for i < C (in unused): the constant tail of a step's row list - the ids N .. N + k - 1 of the k merged-token
 *     rows behind a bag's N feature rows (mhimx_step_run writes it with the step's first launch).
 */
#define MHIMX_PREP_MAX 32
typedef struct { int32_t kind; const float* in; float* out; int64_t R, C; } mhimx_prep_job;
int mhimx_prep_batch(void* stream, const mhimx_prep_job* jobs, int32_t n);

/* out[c,r] = in[r,c]  (weights are transposed once per step so that dX = dY W is also an NT GEMM) */
int mhimx_transpose(void* stream, const float* in, float* out, int64_t R, int64_t C);

/* ------------------------------------------------------------------------------------------
 * ABMIL scorer + softmax pool on a token matrix T[M,E]          
 * replaces: mhim_modules/baseline.py:31-41 (Attention.forward), :72-86 (AttentionGated.forward),
 *           :97-110 (DAttention.forward);  modules/abmil.py:111-143, :203-251 (standalone, with biases)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t E;              /* token width (mlp_dim, 512)                                            */
  int64_t A;              /* scorer hidden width (128; 384 for modules/abmil.AttentionGated)       */
  int32_t act;            /* scorer activation                                                      */
  int32_t gated;
  int32_t prec;
  const float* wa; const float* ba;     /* [A,E], [A]|NULL   attention.0 / attention_a.0            */
  const float* wb; const float* bb;     /* gated: [A,E], [A]|NULL   attention_b.0                   */
  const float* wc; const float* bc;     /* [A], [1]|NULL     attention.2 / attention_c              */
  const float* wa_frag;                 /* optional: prep kind-4 image of wa (made once per step with the other parameter
                                           preparation); NULL: the fused scorer splits wa to bf16 hi/lo on the fly   */
  float gate_drop_p;                    /* gated form only: dropout inside the scorer (modules/abmil.py:96-98: nn.Dropout(0.25) after the
                                           tanh and after the sigmoid branch, training mode); 0 = none                */
  uint64_t gate_drop_seed;              /* counter-based mask of (seed, row, column j) for the tanh branch, (seed, row, A + j) for the gate:
                                           the masks of mhimx_dropout_apply on an [M, 2A] matrix                      */
} mhimx_scorer;

/* Forward over up to two token segments (segment 1 = feature rows, segment 2 = merged tokens).
 *  s[M]      raw scores (baseline.py:32 before softmax; what no_norm=True returns)
 *  stats[2]  = {max_n s, sum_n exp(s-max)}
 *  z[E]      = sum_n softmax(s)_n T[n,:]
 *  u_pre     [M, A*(1+gated)] scorer pre-activations (kept for backward), may be NULL only if ws holds it
 *  cproj     optional [M,C] = T Wp^T (class projections for the pseudo score), needs wp [C,E]
 *  pscore    optional [M1]: the pseudo score of the segment-1 instances (mhimx_pseudo_score's arithmetic and bits, with the
 *            predictor bias bp) written by the pool's own finalize launch - one launch less on the teacher's chain; needs cproj
 *  ws        workspace, mhimx_abmil_pool_ws_bytes(M, E, A, gated) bytes                           */
typedef struct {
  const float* T1; int64_t M1; const float* T2; int64_t M2;
  float* s; float* stats; float* z;
  float* u_pre;
  const float* wp; int64_t C; float* cproj;
  void* ws; int64_t ws_bytes;
  const float* bp; float* pscore;
  const int64_t* rows1;               /* optional gather (M2 == 0): token n of segment 1 is T1[rows1[n]] and, in the backward, its gradient
                                         goes to dT1[rows1[n]] - the student reads the kept rows and the merged tokens straight out of the
                                         bag's feature buffer [N + k, E] (masking.py:107 mask_fn and merge.py:190-194 without any copy).
                                         One-pass scorer shapes only (E = 512, A = 128, plain form). */
  const uint8_t* excl;                /* optional, indexed by SOURCE row (rows1[n] or n) of segment 1: non-zero = the row does not take part
                                         (its score is written as -inf: softmax weight 0, zero gradient in the backward).  Lets a shard of an
                                         instance-sharded bag run the pool over ALL its rows with fixed launch shapes (no data-dependent
                                         counts, no host sync).  One-pass scorer shapes only. */
  const mhimx_prep_job* ride_jobs;    /* optional: n_ride_jobs parameter-only preparation jobs (mhimx_prep_batch's) that run as extra workgroups of the
                                         one-pass scorer launch of this forward - in the workgroup slots it leaves free - instead of in a launch
                                         of their own in front of the step: whatever the teacher's forward does not read itself (the student's
                                         images, the backward's transposes, the Merge preparation's chain).  Their outputs are complete when
                                         this call's launches are.  Other scorer shapes: a launch of their own inside this call. */
  int32_t n_ride_jobs;
  /* (round 5) The forward in two calls around the launches that PRODUCE the last `tail_tokens` tokens of segment 1 - the student's pool over
   * [rows that stay | merged tokens] (mhim.py:351-366) where the tokens come out of Merge, which itself only needs the rows to merge:
   *   phase 1  the one-pass scorer launch over the first M1 - tail_tokens tokens (their partials stay in ws); with ride_merge, the row tiles of
   *            that Merge forward run as the launch's FIRST workgroups (ride_X / ride_R / ride_ws: mhimx_merge_fwd's X, R, ws) - the scorer's
   *            tiles and the Merge rows pass are independent, and the latter heads the longer chain;
   *   (caller)  mhimx_merge_fwd with .rows_done = 1: partial merge, O, to_out -> the tokens;
   *   phase 2  the finalize launch, which first scores the tail tokens itself (u_pre, s written like the scorer's) and merges them with the
   *            partials: stats, z.
   * phase 0 (default): the whole forward in one call, as always.  One-pass scorer shapes, M2 == 0, no pscore / cproj, tail_tokens <= 6. */
  int32_t phase; int32_t tail_tokens;
  const void* ride_merge;             /* const mhimx_merge* (prepared = 1, projection-free form, R <= 4096) or NULL */
  const float* ride_X; int64_t ride_R; void* ride_ws; int64_t ride_ws_bytes;
  int32_t rode_merge;                 /* OUT (phase 1): 1 if the Merge rows pass rode (then call mhimx_merge_fwd with rows_done = 1), else 0 */
  const float* tail_wa_t;             /* phase 2, optional: transpose of sc->wa [E, A] (prep kind 0; what the backward takes as wa_t): the tail
                                         tokens' scorer product then reads the weight coalesced, without cross-lane reductions */
  int64_t tail_row0;                  /* phase 2: >= 0: the tail tokens are the rows tail_row0 .. tail_row0 + tail_tokens - 1 of T1 (what rows1's
                                         last entries hold - the caller's word, saves the finalize launch a dependent load); < 0: read rows1 */
  int32_t no_backward;                /* (round 6) 1: no backward follows (a teacher's forward): the one-pass scorer does not store the
                                         pre-activations u_pre [M, A] - 5 MB per 10 000 rows nobody reads */
} mhimx_pool_io;
int64_t mhimx_abmil_pool_ws_bytes(int64_t M, int64_t E, int64_t A, int32_t gated);
int mhimx_abmil_pool_fwd(void* stream, const mhimx_scorer* sc, mhimx_pool_io* io);      /* (io->rode_merge is written in phase 1) */

/* Backward of the pool given g_z[E] (= dLoss/dz).  Produces dT1,dT2 (overwritten), and the scorer
 * weight gradients (overwritten unless accumulate).  wa_t = transpose(wa) [E,A] (and wb_t). */
typedef struct {
  const float* g_z;
  float* dT1; float* dT2;
  float* d_wa; float* d_ba; float* d_wb; float* d_bb; float* d_wc; float* d_bc;
  const float* wa_t; const float* wb_t;
  int32_t accumulate;
  int32_t splits;
  mhimx_reduce_list* defer;           /* optional: queue the d_wa / d_wc / d_bc final reductions                 */
  const float* wa_t_frag;             /* optional: prep kind-4 image of wa_t [E,A] for the one-pass backward      */
  /* (round 6) optional, one-pass backward with M2 == 0: the gradient of the first img_rows tokens of the list leaves as THEIR PART OF THE
   * PROJECTION'S dPRE IMAGE (mhimx_rows_dpre_image's format, the operand of mhimx_bag_wgrad) instead of as fp32 rows of dT1:
   * dPRE[n,:] = dT[n,:] * img_dact[row(n),:] (row(n) = rows1[n], or n), token n = image row n, every other row of the launch's 32-row tiles
   * a zero row (the image needs ceil(M1 / 32) * 32 rows); img_part (optional) [ceil(M1 / 32), E]: per-tile column sums of dPRE - the
   * bias gradient's partials.  Tokens img_rows .. M1-1 (a Merge's tokens) keep their dT1 rows.  replaces: the dH rows' round trip through
   * memory between this backward and mhimx_rows_dpre_image (mhim.py:69-76's activation / dropout backward). */
  void* img; const void* img_dact; float* img_part; int64_t img_rows;
} mhimx_pool_grad;
int mhimx_abmil_pool_bwd(void* stream, const mhimx_scorer* sc, const mhimx_pool_io* io, const mhimx_pool_grad* g);

/* attn[n] = exp(s[n]-stats[0])/stats[1]   (baseline.py:35, what return_attn gives) */
int mhimx_softmax_from_stats(void* stream, const float* s, const float* stats, float* attn, int64_t M);

/* Instance-sharded bag (BASELINE config c5): merge W partial pools parts[W][2+E] = (max_w, L_w, z_w[E]) — one per rank,
 * all-gathered — into the bag's (stats[2] = {max, L}, z[E]).  Fixed summation order; an empty shard sends L_w = 0.
 * replaces: the single softmax over all N instances of baseline.py:35-41 when the rows live on several GPUs. */
int mhimx_lse_merge(void* stream, const float* parts, int64_t W, int64_t E, float* stats, float* z);

/* score[n] = max_c softmax_c( attn_n * cproj[n,c] + bp0 )
 * replaces: mhim_modules/scoring.py:37-58 (get_pseudo_score), incl. the class-0 bias quirk (:54).
 * s == NULL: attn_n = 1, i.e. score[n] = max_c softmax_c(cproj[n,c] + bp0), the tail of get_pseudo_score_trans (scoring.py:27-33). */
int mhimx_pseudo_score(void* stream, const float* s, const float* stats, const float* cproj, const float* bp,
                       float* score, float* attn_out, int64_t M, int64_t C);

/* ------------------------------------------------------------------------------------------
 * Nystrom / TransMIL encoder primitives (forward and backward)          
 * The encoder is composed from the GEMMs above plus these streaming kernels; q,k,v stay packed as the
 * [n_pad, 3*inner] output of to_qkv and every per-head matrix is addressed by (pointer offset, row pitch).
 * ---------------------------------------------------------------------------------------- */
/* LayerNorm over the last dim (eps 1e-5, biased variance) and its backward (dx may be NULL; d_w, d_b (+)= if accumulate).
 * ws (bwd): 2*512*E floats.   replaces: nn.LayerNorm at baseline.py:199,222 / merge.py:96 and its autograd. */
int mhimx_layernorm_fwd(void* stream, const float* x, int64_t M, int64_t E, const float* w, const float* b, float* y, float* mean,
                        float* rstd);
int mhimx_layernorm_bwd(void* stream, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                        const float* rstd, float* dx, float* d_w, float* d_b, int32_t accumulate, float* ws);
/* the same with dx = (LayerNorm backward) + resid: the gradient of y = x + f(LayerNorm(x)) (baseline.py:213-218) in one pass instead
 * of a LayerNorm backward and an addition of the residual branch's gradient */
int mhimx_layernorm_bwd_res(void* stream, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                            const float* rstd, const float* resid, float* dx, float* d_w, float* d_b, int32_t accumulate, float* ws);
/* out = x * keep/(1-p) with the counter-based mask of (seed + *tick, row, col): re-applies a forward dropout to a gradient */
int mhimx_dropout_apply(void* stream, const float* x, float* out, int64_t M, int64_t E, float p, uint64_t seed, const uint64_t* tick);
/* the same for the dropout stream of mhimx_bag_project's epilogue (one mix per PAIR of columns, 16-bit thresholds): the gradient of a
 * projection that ran with drop_p > 0 and no stored mask.  E % 4 == 0, 0 < p < 1. */
int mhimx_dropout_apply_proj(void* stream, const float* x, float* out, int64_t M, int64_t E, float p, uint64_t seed, const uint64_t* tick);
/* y[r,:] = softmax(alpha * x[r,:]) over the last dim of x[R,L]; bwd: dx = alpha*y*(dy - sum(y*dy)).
 * replaces: `.softmax(dim=-1)` nystrom_attention.py:130 and its autograd. */
int mhimx_softmax_rows(void* stream, const float* x, float* y, int64_t R, int64_t L, float alpha);
int mhimx_softmax_rows_bwd(void* stream, const float* y, const float* dy, float* dx, int64_t R, int64_t L, float alpha);
/* out[j,c] = mean over the l consecutive rows j*l..(j+1)*l-1 of x[T, C] (row pitch ldx).  nystrom_attention.py:93-109 */
int mhimx_landmark_mean(void* stream, const float* x, int64_t ldx, int64_t T, int64_t l, int64_t C, float* out);
int mhimx_landmark_mean_bwd(void* stream, const float* dout, int64_t T, int64_t l, int64_t C, float* dx, int64_t ldx, int32_t accumulate);
/* y = a*I + b*x on [B,n,n] (the 7I-, 15I-, 13I- terms of nystrom_attention.py:25);  y = alpha*x + beta*y element-wise */
int mhimx_affine_ident(void* stream, const float* x, float* y, int64_t B, int64_t n, float a, float b);
int mhimx_axpby(void* stream, const float* x, float* y, int64_t n, float alpha, float beta);
/* z0 = a^T / (max row-abs-sum * max col-abs-sum) with GLOBAL maxima over the B matrices (nystrom_attention.py:15-18).
 * stats[4] = {c, r, argmax row, argmax col}; ws: 2*B*n floats (fwd), 256 floats (bwd).  bwd includes the gradient that
 * flows through the two maxima (torch autograd differentiates them). */
int mhimx_pinv_init(void* stream, const float* a, int64_t B, int64_t n, float* z, float* stats, float* ws);
int mhimx_pinv_init_bwd(void* stream, const float* dz, const float* z0, const float* stats, int64_t B, int64_t n, float* da, float* ws);
/* out[t,c] (+)= sum_tau w[c/dh, tau] * v[t+tau-KS/2, c] (zero outside [0,T)); flip=1 applies the transposed stencil
 * (the gradient w.r.t. v).  nystrom_attention.py:59-63,135-136 (Conv2d(heads,heads,(33,1),groups=heads)). */
int mhimx_resconv(void* stream, const float* v, int64_t ldv, const float* w, int64_t KS, int64_t dh, int64_t T, int64_t C, float* out,
                  int64_t ldo, int32_t accumulate, int32_t flip);
int64_t mhimx_resconv_dw_ws_floats(int64_t T, int64_t C, int64_t dh, int64_t KS);
int mhimx_resconv_dw(void* stream, const float* dout, int64_t ldo, const float* v, int64_t ldv, int64_t KS, int64_t dh, int64_t T,
                     int64_t C, float* dw, float* ws);
/* The Nystrom attention block WITHOUT its n x m matrices (nystrom_attention.py:111-136): attn1 = softmax(q k~^T) and attn3 =
 * softmax(q~ k^T) are never written; token tiles are streamed, scores recomputed from q / k with saved log-sum-exps (csrc/nys_flash.hip).
 * 8 heads x 64 dims, 256 landmarks.  q, k, v: head h = 64 columns at +64h of a row of pitch ld (the packed to_qkv output); ql, kl: the
 * landmark means [256, .] (row pitch ldl, head h at +64h); T tokens (multiple of 64); scale multiplies every score (nystrom:83).
 * ws: >= mhimx_nys_ws_floats(T) floats.  Log-sum-exps are base 2 of the scaled scores: P = 2^(s * scale * log2 e - lse). */
typedef struct mhimx_nys {
  const float* q; const float* k; const float* v;
  int64_t ld, T;
  const float* ql; const float* kl;
  int64_t ldl;
  float scale;
  float* ws;
  int64_t ws_floats;
} mhimx_nys;
int64_t mhimx_nys_ws_floats(int64_t T);
/* a3v[8,256,64] = softmax_n(scale q~ k^T) v  (nystrom:116,131 + the attn3 @ v of :133), lse3[8,256] */
int mhimx_nys_a3v_fwd(void* stream, const mhimx_nys* a, float* a3v, float* lse3);
/* out[T, 64h..] (row pitch ldo) = softmax_m(scale q k~^T) w2, w2[8,256,64] = pinv(attn2) a3v  (nystrom:114,129,133); lse1[8,T].
 * accumulate != 0: out += (the residual convolution of v, nystrom:135-136, written there first: one pass over out less). */
int mhimx_nys_out_fwd(void* stream, const mhimx_nys* a, const float* w2, float* out, int64_t ldo, float* lse1, int32_t accumulate);
/* its backward: dq (into [T, 64h..], pitch lddq), dkl (landmark layout, pitch lddl; the S1 term only), dw2[8,256,64];
 * delta1[8,T] is scratch (rowsum(P dP)). */
int mhimx_nys_out_bwd(void* stream, const mhimx_nys* a, const float* w2, const float* dout, int64_t ldd, const float* lse1,
                      float* delta1, float* dq, int64_t lddq, float* dkl, int64_t lddl, float* dw2);
/* backward of a3v: dk (=), dv (= or += when accumulate_dv), both pitch lddk, dql (landmark layout, pitch lddl; the S3 term only) */
int mhimx_nys_a3v_bwd(void* stream, const mhimx_nys* a, const float* a3v, const float* da3v, const float* lse3, float* dk, float* dv,
                      int64_t lddk, int32_t accumulate_dv, float* dql, int64_t lddl);
/* r[8,T] = u attn3 with u[8,256] = attn1[cls] pinv: the cls token's attention row (nystrom:143-150) */
int mhimx_nys_cls_attn(void* stream, const mhimx_nys* a, const float* lse3, const float* u, float* r);
/* BatchNorm1d over the M instances of one bag (mil_norm='bn': modules/abmil.py:167-169,206-225, transmil.py:79-81,112-115).
 * fwd, training: mean / rstd / var_out [C] are OUTPUTS (biased variance; the host updates the running statistics from mean and var_out);
 *      eval: mean = running mean (in), rstd = 1/sqrt(running var + eps) (in).  y = (x - mean) rstd w + b.
 * bwd: dw = sum dy xhat, db = sum dy, dx (training: through the batch statistics; eval: dy w rstd).  ws: mhimx_bn_ws_floats(M, C). */
int64_t mhimx_bn_ws_floats(int64_t M, int64_t C);
int mhimx_bn_fwd(void* stream, const float* x, int64_t M, int64_t C, const float* w, const float* b, float eps, int32_t training, float* mean,
                 float* rstd, float* var_out, float* y, float* ws);
int mhimx_bn_bwd(void* stream, const float* dy, const float* x, int64_t M, int64_t C, const float* w, const float* mean, const float* rstd,
                 int32_t training, float* dx, float* dw, float* db, float* ws);
/* out = x + SINCOS(pos): the parameter-free 2-d sin-cos position embedding of modules/abmil.DAttention(pos='sincos')
 * (emb_position.py:5-83); pos_xy[N,2] = the patch grid coordinates (x, y) of every row. */
int mhimx_sincos_add(void* stream, const float* x, const int64_t* pos_xy, int64_t N, int64_t C, float* out);
/* PPEG (emb_position.py:85-120).  combine: wc[C,49] = w7 + pad(w5) + pad(w3) + identity, bc = b7+b5+b3;
 * fwd: y[N,C] = depth-wise 7x7 stencil of the wrap-padded H x H token grid; bwd: dx, dwc [C,49], dbc [C]. */
int mhimx_ppeg_combine(void* stream, const float* w7, const float* w5, const float* w3, const float* b7, const float* b5,
                       const float* b3, int64_t C, float* wc, float* bc);
int mhimx_ppeg_fwd(void* stream, const float* x, int64_t N, int64_t C, const float* wc, const float* bc, float* y,
                   int64_t grid /* 0: emb_position.PPEG's own grid rule; > 0: an explicit grid x grid layout (transmil.py:57-64) */);
int64_t mhimx_ppeg_bwd_ws_floats(int64_t N, int64_t C);
int mhimx_ppeg_bwd(void* stream, const float* dy, const float* x, int64_t N, int64_t C, const float* wc, float* dx, float* dwc, float* dbc,
                   float* ws, int64_t grid);
/* PPEG on a band of the token grid (the sequence-parallel encoder: every rank holds its own tokens' grid rows + three halo rows either
 * side, emb_position.py:92-120).  side = mhimx_ppeg_side(N, &wrapN): the grid is side x side, cells [N, wrapN) repeat the first tokens
 * (emb_position.py:100-103), cells past wrapN are zero.  The buffers hold the CELLS [cell0, cell0 + ncell) (cell0 % side == 0), wrap cells
 * filled by the caller's exchange:
 *   band_fwd   y[cell - out0] for cell in [out0, out1)  from xb                       (out1 <= N)
 *   band_bwd   dx[cell - out0] for cell in [out0, out1) from dyb (dy of the band's cells; zero for cells >= N) - out1 may reach wrapN: the
 *              wrap cells' gradient belongs to the first tokens (the caller sends it there); dwc [C,49] / dbc [C] = this band's partial sums
 *              over its first n_dw produced cells (the ones that are outputs of the forward).  ws: mhimx_ppeg_band_bwd_ws_floats(out1 - out0, C). */
typedef struct { int64_t H, cell0, ncell, out0, out1; } mhimx_ppeg_band;
int64_t mhimx_ppeg_side(int64_t N, int64_t* wrapN);
int mhimx_ppeg_band_fwd(void* stream, const float* xb, const mhimx_ppeg_band* band, int64_t C, const float* wc, const float* bc, float* y);
int64_t mhimx_ppeg_band_bwd_ws_floats(int64_t n_out, int64_t C);
int mhimx_ppeg_band_bwd(void* stream, const float* dyb, const float* xb, const mhimx_ppeg_band* band, int64_t n_dw, int64_t C, const float* wc,
                        float* dx, float* dwc, float* dbc, float* ws);
/* out[t,c] = v[t,c] * a[c/dh, t]   (scoring.py:25: per-head value x attention, heads interleaved as (h d)) */
int mhimx_scale_heads(void* stream, const float* v, int64_t ldv, const float* a, int64_t lda, int64_t dh, int64_t T, int64_t C, float* out);

/* ------------------------------------------------------------------------------------------
 * Hard-instance select                                          
 * replaces: mhim_modules/masking.py:9-88 (select_mask_fn, 2-D scores) incl. the Python set()
 *           round trip (:77-80), and its composition in modules/mhim.py:109-179 (get_mask).
 * Tie contract (the reference's torch.topk order is implementation-defined): candidates are ordered by
 * (value desc [asc if !largest], index asc).  Kept ids are emitted ascending.
 *  score[N]; k = ceil(ps*ratio) candidates; the first n_sel entries of perm[k] (int64, a permutation of
 *  0..k-1; NULL = identity) index the sorted candidate list (masking.py:66-71);
 *  other[n_other] (int64, may be NULL) = ids masked by an earlier call: result is the sorted union (:74-75);
 *  mask_ids[N] int64 = kept ascending ++ masked;  len_keep is returned through *len_keep_dev (device int64)
 *  and equals N - |masked| (host-computable when other == NULL: N - n_sel).
 *  N <= 2^24, k <= 16384 (LDS-resident sort).  ws: mhimx_select_ws_bytes(N) bytes. */
int64_t mhimx_select_ws_bytes(int64_t N);
int mhimx_select_mask(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                      const int64_t* perm, const int64_t* other, int64_t n_other,
                      int64_t* mask_ids, int64_t* len_keep_dev, int64_t* topk_sorted /* [k] optional */,
                      void* ws, int64_t ws_bytes);

/* Production form of the HAM mask + Merge split, all on the device and without host-drawn permutations:
 *  candidates = top-k of score (tie contract above); a uniformly random n_sel-subset of them is masked (masking.py:66-71);
 *  of the L = N - n_sel kept rows a uniformly random merge_R-subset is set aside for merging (merge.py:158-176).
 *  rows_out[L] int64 = [kept rows that stay, ascending (L - merge_R) | rows to merge, ascending (merge_R)];
 *  mask_ids[N] optional (kept ascending ++ masked).  Random keys: counter hash of (rand_seed + *tick, row).
 *  Same distribution of SETS as the reference; the reference's random ORDER of kept rows is irrelevant to the pool.
 *  N <= 16384, k <= 4096. */
int mhimx_select_rows(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest,
                      uint64_t rand_seed, const uint64_t* tick, int64_t merge_R, int64_t* rows_out, int64_t* mask_ids,
                      void* ws, int64_t ws_bytes, int32_t merge_first /* 1: rows_out = [rows to merge | rows that stay] */);

/* vote[n] = number of heads whose top-k contains n (masking.py:49-57, msa_fusion='vote'); attn [H,N] */
/* mhimx_select_rows + (round 6) the kept rows a second time, in the order of the projection's dPRE image when the backward's kernels write
 * it themselves (mhimx_pool_grad.img): rows_img = [rows that stay (Lk) | 0 ... | rows to merge (merge_R) from position img_merge_off | 0 ... up
 * to a multiple of 32] - img_merge_off a multiple of 32, >= Lk (+ the merged tokens the scorer's last tile also holds); the zeros stand for
 * zero rows of the image (any valid row id).  rows_img: int64 [ceil((img_merge_off + merge_R) / 32) * 32].  N <= 16384, k <= 4096. */
int mhimx_select_rows_img(void* stream, const float* score, int64_t N, int64_t k, int64_t n_sel, int32_t largest, uint64_t rand_seed,
                          const uint64_t* tick, int64_t merge_R, int64_t* rows_out, int64_t* rows_img, int64_t img_merge_off, void* ws,
                          int64_t ws_bytes, int32_t merge_first);
int mhimx_vote_scores(void* stream, const float* attn, int64_t H, int64_t N, int64_t k, int32_t largest,
                      float* vote, void* ws, int64_t ws_bytes);

/* out[i] = a[b[i]]  (index composition: ids_keep[ids_shuffle]) */
int mhimx_compose_ids(void* stream, const int64_t* a, const int64_t* b, int64_t* out, int64_t n);
/* out[j] = pi(j) (src NULL) or src[pi(j)], pi = a keyed pseudo-random permutation of 0 .. n-1 (6-round Feistel network on 2^b >= n,
 * cycle-walked; key = seed + *tick): the draws the reference takes from torch.randperm - the random subset of the top-k
 * (masking.py:67) and Merge's random split of the kept rows (merge.py:165-170) - as one element-wise launch, reproducible from (seed, tick)
 * alone (every rank of a sharded bag computes the same list).  Every element lands in a prefix of length m with probability m / n. */
int mhimx_random_perm(void* stream, int64_t n, uint64_t seed, const uint64_t* tick, const int64_t* src, int64_t* out);

/* ------------------------------------------------------------------------------------------
 * Merge / MCA                                                   
 * replaces: mhim_modules/merge.py:43-65 (MCA.forward), :131-144 (Merge.merge), :127-129 (EMA).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t E; int64_t k;               /* token width, number of global queries                       */
  int64_t heads; int64_t dim_head;    /* 8 x 64                                                      */
  const float* q_param;               /* merge.global_q_mm [k,E]                                     */
  const float* ln_w; const float* ln_b;
  const float* wkv; const float* wq; const float* wo; const float* bo;
  const float* wkv_t; const float* wq_t; const float* wo_t;   /* transposes, backward only           */
  float mm;                           /* EMA momentum g_q_mm                                         */
  float drop_p; uint64_t drop_seed;   /* MCA dropout (merge.py:33,40), hashed RNG; 0 = off           */
  int32_t prec;
  const uint64_t* drop_tick;          /* optional device step counter mixed into drop_seed           */
  const float* wkv_frag;              /* optional: prep kind-4 image of wkv [2*heads*dim_head, E]: with it (E = 512, 8 x 64) the K/V
                                         projection, its dots and the softmax partials are ONE kernel per row tile       */
  const int64_t* x_rows;              /* optional gather: row n of the block to merge is X[x_rows[n]] (forward) and its gradient goes to
                                         dX[x_rows[n]] (backward): merge.py:158-176 masking without a row copy         */
  int32_t prepared;                   /* 1: the parameter-only part (LN(q), Q, the score vectors) is already in the workspace, written by a
                                         kind-6 job of mhimx_prep_batch on the same stream, for the same weights, R and workspace */
  int64_t own_lo, own_n;              /* an instance-sharded bag (one bag's rows split over the GPUs, BASELINE c5): x_rows holds BAG row ids, this
                                         shard owns [own_lo, own_lo + own_n) and its X / dX hold only those rows (row id - own_lo).  Rows of the
                                         list that are another shard's take no part here.  Forward: mhimx_merge_fwd_part on every shard, an
                                         all-gather of the blocks, mhimx_merge_fwd_finish on every shard.  Backward: mhimx_merge_bwd - dX for the
                                         own rows, parameter gradients as PARTIAL sums over the shards (the caller's gradient all-reduce adds
                                         them).  own_n == 0: one process holds every row.  Projection-free form only (E = 512, 8 x 64, k <= 6). */
  float rep;                          /* with own_n > 0: weight of the gradient terms every shard computes identically (d_bo, d_wo, the V half of
                                         d_wkv) - 1 on one shard, 0 on the others, so that the sum over the shards counts them once */
  int32_t rows_done;                  /* 1: the rows pass of this forward already ran as riding workgroups of the caller's previous launch on this
                                         workspace (mhimx_pool_io.ride_merge): mhimx_merge_fwd starts at the merge of the tile partials */
} mhimx_merge;
int64_t mhimx_merge_ws_bytes(int64_t R, int64_t E, int64_t k, int64_t heads, int64_t dim_head);
/* X[R,E] rows to merge -> z[k,E]; q_new[k,E] = mm*q + (1-mm)*z if update_q; ws keeps what backward needs. */
int mhimx_merge_fwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new,
                    int32_t update_q, void* ws, int64_t ws_bytes);
typedef struct {
  float* d_ln_w; float* d_ln_b; float* d_wkv; float* d_wq; float* d_wo; float* d_bo;
  int32_t accumulate; int32_t splits;
  mhimx_reduce_list* defer;           /* optional: queue the d_ln_w / d_ln_b / d_wkv final reductions            */
} mhimx_merge_grad;
/* dz[k,E] -> dX[R,E] (overwritten) and parameter gradients. */
/* The forward of an instance-sharded bag (mhimx_merge.own_n > 0), in two halves around ONE all-gather of mhimx_merge_part_floats() floats
 * per shard (99 KB; the replicated form all-reduced the [R, 512] block of rows to merge: 39.7 MB at c5):
 *   fwd_part    the rows pass over this shard's rows and the merge of its tile partials, left raw: part = {max[48] | sum[48] | dropped
 *               sum[48] | pooled rows [48][512]} per score slot (head, query);
 *   fwd_finish  parts [W][mhimx_merge_part_floats()] of all shards (the same on every shard, shard order = merge order: bit-identical
 *               replicas) -> pooled rows Y + softmax statistics in this shard's workspace, tokens z, the queries' EMA.
 * R, ws: the same on both calls and on mhimx_merge_bwd. */
int64_t mhimx_merge_part_floats(void);
int mhimx_merge_fwd_part(void* stream, const mhimx_merge* m, const float* X, int64_t R, float* part, void* ws, int64_t ws_bytes);
int mhimx_merge_fwd_finish(void* stream, const mhimx_merge* m, const float* parts, int32_t W, int64_t R, float* z, float* q_new,
                           int32_t update_q, void* ws, int64_t ws_bytes);
int mhimx_merge_bwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX,
                    const mhimx_merge_grad* g, void* ws, int64_t ws_bytes);
/* The same arguments BEFORE the pool backward that produces dz (= a row block of its dT1 buffer, e.g. the k token rows behind a bag's
 * feature gradients): parks the backward's first stage on g->defer (mhimx_reduce_list.pre) so that it rides in the pool backward's rows
 * launch; a no-op (returns 0, nothing parked) without a list or for shapes that take the general Merge path.  Enqueues nothing itself. */
int mhimx_merge_bwd_park(const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* g, void* ws,
                         int64_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Feature projection backward pieces                            
 * ---------------------------------------------------------------------------------------- */
/* dPre = dH * act'(pre or H) * keep/(1-p), in place on dH.  For relu `pre` may be NULL (uses H>0).
 * colsum_out[E] (optional, (+)= if accumulate): column sums of dPre = the feature-bias gradient, produced in the same pass
 * (ws: 1024*E floats). */
int mhimx_act_bwd(void* stream, float* dH, const float* H, const float* pre, int64_t M, int64_t E, int32_t act,
                  float drop_p, uint64_t drop_seed, const uint8_t* drop_mask, const int64_t* rows, float* colsum_out,
                  int32_t accumulate, void* ws, int64_t ws_bytes, const uint64_t* drop_tick);
/* dH *= dact in place (dact from mhimx_gemm_nt's `dact` output) and colsum_out[e] (+)= sum_m dH[m,e]: the backward
 * through activation + dropout and the bias gradient in one streaming pass.  ws: 1024*E floats. */
int mhimx_mul_colsum(void* stream, float* dH, const float* dact, int64_t M, int64_t E, float* colsum_out, int32_t accumulate,
                     void* ws, int64_t ws_bytes, mhimx_reduce_list* defer /* optional */);
/* dpre[p,:] = dH[rows ? rows[p] : p, :] * dact16[same row, :] for p < L (compact [L,E]): the backward through activation + dropout of
 * the rows that took part in the step, gathered from the bag-ordered gradient buffer dH [*,E] and the fp16 d out / d pre image that
 * mhimx_bag_project wrote; colsum_out[e] (+)= sum_p dpre[p,e] (the projection's bias gradient) in the same pass.  ws: 1024*E floats. */
int mhimx_rows_dpre(void* stream, const float* dH, const void* dact16, const int64_t* rows, int64_t L, int64_t E, float* dpre,
                    float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer /* optional */);
/* The same backward with dPre emitted as the MATRIX-CORE IMAGE of its transpose instead of an fp32 matrix (E % 128 == 0): for every
 * 32-row step ks and 128-column block it, 16 KiB at img + (ks * E/128 + it) * 16384 laid out [row octet 4][hi | lo][slot 128][8 bf16],
 * slot of column c = (c % 4) * 32 + (c % 128) / 4, value = hi + lo to ~2^-16; rows past L are zero.  mhimx_wgrad_image_bytes(L, E)
 * bytes.  colsum_out as above (ws: ceil(L/32) * E floats).  Feeds mhimx_bag_wgrad.  dact16 NULL: dPre = dH[rows] as it is - the
 * image of ANY [L, E] gradient matrix (the Linear layers of the TransMIL encoder: nystrom_attention.py:55,57 under autograd). */
int64_t mhimx_wgrad_image_bytes(int64_t L, int64_t E);
int mhimx_rows_dpre_image(void* stream, const float* dH, const void* dact16, const int64_t* rows, int64_t L, int64_t E, void* img,
                          float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer /* optional */);
/* the same with dH COMPACT (row p of dH belongs to list position p) while dact16 stays bag-ordered (row rows[p]): the gradient of token
 * rows that were gathered out of a bag-ordered projection (the TransMIL / DSMIL students: masking.py:107 after mhim.py:335-336) */
int mhimx_rows_dpre_image_c(void* stream, const float* dH_compact, const void* dact16, const int64_t* rows, int64_t L, int64_t E, void* img,
                            float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer /* optional */);
/* the same in BAG order: row n of the image is dH[n] * dact16[n] if keep[n] (uint8 [N]; NULL: every row), a zero row otherwise - the
 * E-side operand of mhimx_bag_wgrad with .ximg (the k dimension runs over the bag's rows as they lie in memory: no gather) */
int mhimx_rows_dpre_image_k(void* stream, const float* dH, const void* dact16, const uint8_t* keep, int64_t N, int64_t E, void* img,
                            float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer /* optional */);
/* The projection's weight gradient (modules/mhim.py:69-76 backward; the last GEMM of the step):
 *     C[E,D] (+)= dPre[L,E]^T . X[rows ? rows[p] : p][D],   p < L
 * dPre as the image above, X the raw fp32 bag (split to bf16 hi/lo on its way into LDS, transposed with v_permlane32_swap); 3-term bf16
 * (~2^-16); the L reduction is split over slabs in ws (mhimx_wgrad_ws_floats(L, E, D) floats), summed by a queued reduction job (defer)
 * or a second launch.  Needs E % 128 == 0, D % 256 == 0, ldx % 4 == 0 and n_bag_rows * ldx * 4 < 2^32 (row offsets are 32-bit). */
typedef struct {
  const void* img;                                       /* mhimx_rows_dpre_image output                                */
  const float* X; int64_t ldx; int64_t n_bag_rows;       /* the bag [n_bag_rows, D], row pitch ldx                      */
  const int64_t* rows;                                   /* optional [L] row ids into X                                 */
  int64_t L, E, D;
  float* C; int64_t ldc; int32_t accumulate;
  float* ws; int64_t ws_floats;
  mhimx_reduce_list* defer;                              /* optional: queue the slab sum                                */
  int32_t ride_tail;                                     /* with defer: every reduction ALREADY queued on the list (their inputs are final
                                                            before this launch starts) and the last stage of a parked Merge-backward tail run
                                                            as trailing workgroups of this launch - in the CUs its last round of tiles leaves
                                                            idle - instead of in mhimx_reduce_flush; the list then holds this launch's own
                                                            slab sum only (which mhimx_optim_step can fold into the update: `fold`)   */
  const void* ximg;                                      /* optional: the bag as the product's D-side operand image (mhimx_prep_batch kind 9 of
                                                            X[n_bag_rows, D], contiguous rows).  Then img is the dPRE image in BAG order
                                                            (mhimx_rows_dpre_image_k: rows that did not take part are zero rows), rows = NULL,
                                                            L = n_bag_rows, and both operands reach LDS by linear DMA: no register path, no
                                                            split in the loop (the four E-side tiles of a k-step no longer split X four times) */
} mhimx_bag_wgrad_args;
int64_t mhimx_wgrad_ws_floats(int64_t L, int64_t E, int64_t D);
int mhimx_bag_wgrad(void* stream, const mhimx_bag_wgrad_args* a);
/* The same product SUMMED over the n_bags <= 8 bags of an accumulation window (base_engine.py:100-119: the gradients of the window's bags
 * add up before the one optimiser step) in ONE launch: dW (+)= sum_b dPRE_b^T X_b.  Every bag brings its image, bag and row ids and owns
 * its slabs of k-steps; the slab sum (and everything a launch pays once: row tables, first tiles, 34 MB of slab traffic) happens once per
 * window instead of once per bag.  All bags: the same L, E, D, ldx; C / ldc / accumulate / ws / defer of bags[0] are used;
 * ws_floats >= mhimx_wgrad_multi_ws_floats(L, E, D, n_bags). */
int64_t mhimx_wgrad_multi_ws_floats(int64_t L, int64_t E, int64_t D, int32_t n_bags);
int mhimx_bag_wgrad_multi(void* stream, const mhimx_bag_wgrad_args* bags, int32_t n_bags);
/* Instance-sharded bags: a shard holds bag rows [lo, lo + n).  The student's row lists are replicated
 * (every rank runs the same select); these three turn them into fixed-shape local work, with no data-dependent count on the host:
 *   shard_flags   excl[i] = 1 for i < n + k_tokens, then excl[row - lo] = 0 for every stay row (rows_all[R .. R+Lk)) inside the shard and
 *                 excl[n ..] = !tokens_live  (the merged tokens are counted on one rank only)                  -> mhimx_pool_io.excl
 *   shard_gather  out[j,:] = H[rows[j] - lo, :] if the shard owns rows[j], else 0   (j < R; the [R,E] block is then all-reduced: exact)
 *   shard_scatter dH[rows[j] - lo, :] = dX[j,:] for the rows the shard owns */
int mhimx_shard_flags(void* stream, const int64_t* rows_all, int64_t R, int64_t Lk, int64_t lo, int64_t n, int64_t k_tokens,
                      int32_t tokens_live, uint8_t* excl);
int mhimx_shard_gather(void* stream, const float* H, int64_t E, const int64_t* rows, int64_t R, int64_t lo, int64_t n, float* out);
int mhimx_shard_scatter(void* stream, const float* dX, int64_t E, const int64_t* rows, int64_t R, int64_t lo, int64_t n, float* dH);
/* out[e] (+)= sum_m X[m,e].  ws: min(128, ceil(M/16)) x E floats at least; up to 512 x E are used (more, shorter row chunks). */
int mhimx_colsum(void* stream, const float* X, int64_t M, int64_t E, float* out, int32_t accumulate,
                 void* ws, int64_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Head: predictor + CE + SoftTargetCE, forward and backward in one launch
 * replaces: mhim.py:97,371 (predictor), losses.py:26-45, mhim.py:300-316, base_engine.py:99-102.
 *  logits[C] = Wp z + bp;  ce = -log softmax(logits)[label];  cl = -sum softmax(t/temp_t) log_softmax(z)
 *  loss = (main_alpha*ce + aux_alpha*cl)/accum;  out: losses[3] = {loss*accum, ce, cl};
 *  g_z[E], d_wp[C,E], d_bp[C] (of `loss`).  t may be NULL (aux term skipped, common_mil.py:24).
 *  Autograd form: label_dev == NULL and g_logits_in[C] / g_cl_in[1] (device) carry the upstream gradients
 *  of an externally computed criterion (the reference's trainer applies nn.CrossEntropyLoss itself).
 * ---------------------------------------------------------------------------------------- */
int mhimx_head_fwd_bwd(void* stream, const float* z, const float* t, const float* wp, const float* bp,
                       const int64_t* label_dev, int64_t E, int64_t C, float temp_t, float main_alpha,
                       float aux_alpha, float inv_accum, float* logits, float* losses, float* g_z, float* d_wp,
                       float* d_bp, int32_t accumulate, const float* g_logits_in, const float* g_cl_in);

/* ------------------------------------------------------------------------------------------
 * DSMIL encoder pieces                        replaces: mhim_modules/baseline.py:112-194
 * The encoder is composed from the GEMMs, mhimx_softmax_rows and these; see mhim_mil_amd/dsmil.py.
 * ---------------------------------------------------------------------------------------- */
/* y[:,c] = softmax over the M rows of alpha * x[:,c] (x[M,C] contiguous, few columns); bwd: dx = alpha*y*(dy - sum_m y*dy).
 * replaces: F.softmax(A, 0) (baseline.py:147) and its autograd. */
int mhimx_softmax_cols(void* stream, const float* x, float* y, int64_t M, int64_t C, float alpha);
int mhimx_softmax_cols_bwd(void* stream, const float* y, const float* dy, float* dx, int64_t M, int64_t C, float alpha);
/* vals[c] = max_m x[m,c], idx[c] = arg max (lowest m on ties): the critical instance per class (baseline.py:139-140) and the
 * max-instance logits (:172).  x[M,C] contiguous, C <= 16. */
int mhimx_colmax(void* stream, const float* x, int64_t M, int64_t C, float* vals, int64_t* idx);
/* out[m] = max_c x[m,c]: the per-instance score of DSMIL with cls_attn (baseline.py:176). */
int mhimx_rowmax(void* stream, const float* x, int64_t M, int64_t C, float* out);
/* Head: logits = 0.5 (bag + max instance) (common_mil.py:26-28), CE, and cl = mean_c SoftTargetCE(Bs[c], Bt[c]) on the
 * per-class bag features [C,V] (mhim.py:355-364).  losses[3] = {main_alpha*ce + aux_alpha*cl, ce, cl}; g_* of loss*inv_accum.
 * Autograd form: label_dev == NULL (no CE) and g_cl_in[1] (device) = upstream d loss / d cl.  Bt may be NULL. */
int mhimx_dsmil_head(void* stream, const float* logits_bag, const float* logits_ins, const int64_t* label_dev, const float* Bs,
                     const float* Bt, int64_t C, int64_t V, float temp_t, float main_alpha, float aux_alpha, float inv_accum,
                     float* losses, float* g_logits_bag, float* g_logits_ins, float* g_B, const float* g_cl_in);

/* ------------------------------------------------------------------------------------------
 * Optimiser + EMA teacher                                       
 * replaces: torch.optim.Adam as built in train_utils.py:58-65 (L2 weight decay added to the gradient)
 *           and the per-parameter EMA loop base_engine.py:166-167.
 *  p,g,m,v [n_train]; teacher[n_all] (n_all >= n_train: the tail holds non-trainable parameters such as
 *  merge.global_q_mm which only take part in the EMA); step >= 1 is the Adam step AFTER this update.
 *  grad_scale multiplies g first (1/world_size after a SUM all-reduce). teacher may be NULL.
 * ---------------------------------------------------------------------------------------- */
int mhimx_adam_ema(void* stream, float* p, const float* g, float* m, float* v, float* teacher, int64_t n_train,
                   int64_t n_all, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, float ema_mm, int32_t zero_grad /* also clears g (cast away const) */,
                   const uint64_t* step_dev /* optional: Adam step read from device memory (graph replay) */,
                   const float* mm_table, int64_t mm_len /* optional device table: EMA momentum of iteration `step`
                                                           (the reference's cosine `mm_sche`, base_engine.py:160-161) */);

/* The same update with the trainer's optional pieces, all device-side so that a captured hipGraph keeps stepping them:
 *  lr_table   per-update learning-rate schedule (train_utils.py:69-77 builds it, base_engine.py:152-153 steps it once per update):
 *             the update with Adam step t uses lr_table[min(t - 1, lr_len - 1)]; NULL: the constant `lr`;
 *  g_extra    n_extra gradient slabs (slab z at g_extra + z * extra_pitch) added to g before anything else, in slab order: the other
 *             streams of an accumulation window (--accumulation_steps, base_engine.py:29,100-102) accumulate into their own slabs;
 *  clip_norm  > 0: torch.nn.utils.clip_grad_norm_(parameters, clip_norm) on the scaled, summed gradient, i.e. --clip_grad
 *             (base_engine.py:115-119, timm dispatch_clip_grad mode 'norm'): g *= min(1, clip_norm / (||g||_2 + 1e-6)); one extra
 *             launch (per-block sums of squares into ws, >= 1024 floats), the final sum in every block of the update in a fixed order;
 *  fold       see the field. */
typedef struct {
  float* p; float* g; float* m; float* v; float* teacher;
  int64_t n_train, n_all;
  int64_t step; const uint64_t* step_dev;
  float lr; const float* lr_table; int64_t lr_len;
  float beta1, beta2, eps, weight_decay, grad_scale;
  float ema_mm; const float* mm_table; int64_t mm_len;
  int32_t zero_grad;
  const float* g_extra; int64_t n_extra; int64_t extra_pitch;
  float clip_norm;
  float* ws; int64_t ws_floats;
  mhimx_reduce_list* fold;      /* optional: a step's deferred reductions.  Split-K slab sums on it (kind 1, <= 4 of them, <= 64 slabs each) whose
                                   contiguous, 16-byte aligned output lies inside g[0, n_train) are taken off the list and summed by the update
                                   kernel itself while it reads the gradient - same order, same bits, no separate pass over the slabs;
                                   whatever else the list holds is flushed first (mhimx_reduce_flush).  Not with clip_norm (the norm needs the
                                   final gradient): the whole list is flushed then. */
  int64_t extra_lo;             /* (round 6) the slabs of g_extra cover the gradient elements [extra_lo, n_train) only (a multiple of 4; 0: all)  */
  int32_t extra_only;           /* 1: from extra_lo on the gradient is the slabs' sum ALONE - g is neither read nor expected to hold anything
                                   there (mhimx_window_run: g carries feature.0.weight's gradient, the bags' slabs everything else)            */
} mhimx_optim_args;
int mhimx_optim_step(void* stream, const mhimx_optim_args* a);

/* ------------------------------------------------------------------------------------------
 * The whole MHIM(ABMIL) train step of one bag behind ONE call        (round 5)
 * replaces: engines/common_mil.py:14-48 (forward_func: model_ema.forward_teacher -> model(bag, score, teacher_feat)) together with
 *           engines/base_engine.py:76-167 (criterion, loss.backward(), optimizer.step(), the per-parameter EMA) for one bag, i.e.
 *           modules/mhim.py:181-227 (forward_teacher) -> :109-179 (get_mask) -> :318-378 (forward) -> their autograd -> Adam + EMA.
 * One call enqueues the step's ~17 launches (the entry points above, in the order mhim_mil_amd/engine.py issues them: same kernels, same
 * arguments, same bits) on `stream`: no allocation, no host sync, nothing read back - the host cost of a step is the launches themselves
 * (tens of microseconds) instead of ~1700 interpreter calls, and any language binds the step with one function.  Capturable into a hipGraph.
 *   teacher and student projection in one pass over the bag -> teacher scorer + pool (+ pseudo score) -> device-drawn HAM mask and Merge
 *   split (mhimx_select_rows) -> Merge -> student scorer + pool -> head (CE + distillation) -> backward -> [fused Adam + EMA teacher].
 * Shapes: the single-pass ABMIL step's (E = 512, A = 128, plain scorer, C <= 4, 8 x 64 Merge heads with 8 k <= 48, D % 256 == 0,
 * 64 <= N <= MHIMX_STEP_MAX_ROWS; up to 16 384 rows the one-workgroup select with both random subsets drawn in the kernel (k_top <= 4096),
 * above it the multi-workgroup select with the two draws as keyed permutations (k_top <= 16384) - whole-slide bags; rows to merge <= 32768);
 * anything else returns < 0 and the caller composes the step from the building blocks.
 * ---------------------------------------------------------------------------------------- */
#define MHIMX_STEP_MAX_ROWS 262144
#define MHIMX_WINDOW_MAX 8                        /* bags of one mhimx_window_run */
typedef struct {                                  /* one model's parameters: device pointers (views of a flat parameter buffer)          */
  const float* w1; const float* b1;               /* feature.0.weight [E,D], feature.0.bias [E]                                          */
  const float* wa; const float* wc;               /* online_encoder.attention.attention.0.weight [A,E], attention.2.weight [A]           */
  const float* wp; const float* bp;               /* predictor.weight [C,E], predictor.bias [C]                                          */
  float* q;                                       /* merge.global_q_mm [k,E]: the student's is EMA-updated by its own forward (merge.py:142-143) */
  const float* ln_w; const float* ln_b;           /* merge.norm                                                                          */
  const float* wkv; const float* wq; const float* wo; const float* bo;   /* merge.attn.to_kv [2I,E], to_q [I,E], to_out.0 [E,I], bias [E] */
} mhimx_step_params;
typedef struct {                                  /* where the gradients go (views of the flat gradient buffer), overwritten               */
  float* w1; float* b1; float* wa; float* wc; float* wp; float* bp; float* ln_w; float* ln_b; float* wkv; float* wq; float* wo; float* bo;
} mhimx_step_grads;
typedef struct {
  int64_t D, E, A, C, k;                          /* input_dim, mlp_dim, scorer width, classes, merge_k                                  */
  int32_t act, da_act;                            /* MHIMX_ACT_* of the feature / the scorer                                             */
  int32_t attn2score;                             /* the teacher's instance score: 1 = pseudo score (scoring.py:37-58), 0 = attention    */
  float drop_p_teacher, drop_p_student;           /* feature dropout (mhim.py:76); the teacher's is 0 unless it runs in train mode        */
  float merge_drop_p, merge_mm;                   /* MCA dropout (merge.py:33,40), EMA momentum of the global queries                    */
  float temp_t, main_alpha, aux_alpha;
  mhimx_step_params student, teacher;             /* (teacher: the merge.* fields are not read)                                          */
  mhimx_step_grads grad;
  float* p; float* g; float* m; float* v; float* p_teacher;    /* the flat buffers of mhimx_optim_step (p_teacher NULL: no EMA)           */
  int64_t n_train, n_all;
  float lr, beta1, beta2, eps, weight_decay, ema_mm;
  const float* mm_table; int64_t mm_len; const float* lr_table; int64_t lr_len;
  uint64_t* tick; uint64_t* opt_step;             /* device counters: dropout / draw stream position, Adam step (advanced by the step)   */
  float* q_out;                                   /* optional [k,E] (round 6): where the forward's EMA of the global queries goes; student.q then stays what
                                                     it was.  A data-parallel rank's step (options.py:287; every rank's forward sees the update's first
                                                     queries, the ranks' EMA terms are composed after the all-reduce: engine.QueryChain) and the bags of
                                                     an accumulation window.  NULL: in place, as merge.py:142-143 */
  int32_t time_project;                           /* 1 (eager steps only - event records cannot be captured with timing): bracket the step's projection
                                                     launch with a pair of HIP events on `stream`; mhimx_step_project_ms reads the pairs back.  How
                                                     bench.py times the dominant kernel on the step's own issue path (round 6) */
  void* side_stream;                              /* optional second hipStream_t (round 6): the step is enqueued as a DAG instead of a chain - the
                                                     launches that share no data run as two branches that fork from and join `stream` through
                                                     events (capturable: a hipGraph of the step then has parallel branches):
                                                       forward   student scorer over the rows that stay  ||  Merge (rows pass, partial merge, O, to_out);
                                                       backward  scorer-weight-gradient product, Merge parameter-gradient tail, queued reductions
                                                                 ||  Merge rows backward -> dPRE image -> projection weight gradient.
                                                     Same kernels on the same data: the results have the bits of the chain.  NULL: one stream, the chain. */
} mhimx_step_cfg;
/* row counts of one step (masking.py:30-35,61 and merge.py:163 in float64, as the reference computes them):
 * k_top = ceil(N r), r = mask_ratio_h / mask_ratio_hr (r > 1: r = 1, hr = mask_ratio_h); n_sel = ceil(k_top hr) if hr < 1 else k_top;
 * len_keep = N - n_sel; Lk = int(len_keep merge_ratio); R = len_keep - Lk.   returns < 0 when the recipe leaves nothing to mask or merge */
typedef struct { int64_t k_top, n_sel, len_keep, Lk, R; } mhimx_step_counts;
int mhimx_step_counts_of(int64_t N, double mask_ratio_h, double mask_ratio_hr, double merge_ratio, mhimx_step_counts* out);
/* seeds of the step's four counter-hash streams (mixed with *tick on the device) */
typedef struct { uint64_t drop_teacher, drop_student, select, mca; } mhimx_step_seeds;
/* byte offsets, inside the workspace, of what a caller may want to look at after a step (valid until the next step on that workspace) */
typedef struct {
  int64_t total;                                  /* bytes mhimx_step_run needs for this (N, counts)                                      */
  int64_t logits, losses;                         /* float [C], float [3] = {main ce + aux cl, ce, cl}                                    */
  int64_t score;                                  /* float [N]: the teacher's instance score                                             */
  int64_t rows_all;                               /* int64 [len_keep + k] = [rows to merge (R) | rows that stay (Lk) | N .. N+k-1]        */
  int64_t H_teacher, H_student;                   /* float [N,E], float [N+k,E] (rows N..: the merged tokens)                             */
  int64_t dact;                                   /* fp16 [N,E]: d out / d pre of the student's projection                                */
  int64_t z_teacher, z_student, g_z;              /* float [E] each                                                                       */
  int64_t dH;                                     /* float [N+k,E]: gradient of the student's feature rows                                */
} mhimx_step_layout;
int mhimx_step_layout_of(const mhimx_step_cfg* cfg, int64_t N, const mhimx_step_counts* cnt, mhimx_step_layout* out);
/* one bag.  update = 1: the fused Adam + EMA follows (the backward's last reductions ride / fold as in the trainer's own step);
 * update = 0: forward + backward only, the complete gradient in cfg->grad.  host_step: the Adam step after this update when
 * cfg->opt_step is NULL.  X [N, ldx] fp32 device, label_dev int64 [1] device. */
int mhimx_step_run(void* stream, const mhimx_step_cfg* cfg, const float* X, int64_t ldx, int64_t N, const int64_t* label_dev,
                   const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step, void* ws, int64_t ws_bytes, int32_t update);
/* elapsed times (ms) of the projection launches bracketed since the last call on the current device (mhimx_step_cfg.time_project; at most 256
 * are kept): waits for each bracket's last event.  empty_ms_out (optional): what a bracket with NOTHING inside read at the same place in the
 * queue (a third event recorded right behind the second) - the event pair's own share of ms_out.  Returns the number written (<= cap),
 * < 0 on error. */
int mhimx_step_project_ms(float* ms_out, float* empty_ms_out, int32_t cap);
/* n_bags consecutive complete steps (one update each), bag after bag, on one workspace of max_b layout.total bytes (SURVEY.md 7 H4
 * "run_steps"): a resident dataset's epoch as one call per chunk of bags. */
int mhimx_step_run_many(void* stream, const mhimx_step_cfg* cfg, int32_t n_bags, const float* const* X, const int64_t* ldx, const int64_t* N,
                        const int64_t* const* labels_dev, const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step0,
                        void* ws, int64_t ws_bytes);

/* An accumulation window (--accumulation_steps n, base_engine.py:29,47-49,100-119: n bags share the weights, their gradients add up, ONE
 * optimiser step) with every launch over ALL its bags (round 6): one preparation, both projections of the n bags in one launch, the step's
 * middle (teacher scorer, select, Merge, student scorer, head, backward to the dPRE image) issued once with one grid plane per bag, ONE
 * weight-gradient launch over the n images, the EMA chain of the global queries, Adam + EMA: ~22 launches instead of n x 16.
 * The bags have ONE shape: X[b] [N, ldx] fp32 device (b < n_bags, 2 <= n_bags <= MHIMX_WINDOW_MAX), N <= 16384, merge_k <= 6; labels_dev[b]
 * int64 [1] on the device; seeds [n_bags]; cfg as for mhimx_step_run with every cfg->grad view inside cfg->g[0, n_train) and
 * feature.0.weight's FIRST (cfg->grad.w1 == cfg->g: the flat buffer then holds that gradient and nothing else until the update adds the
 * bags' slabs; q_out, side_stream, time_project unused: NULL / 0).  Every bag's loss is scaled by 1 / n_bags.  Per bag the arithmetic is
 * mhimx_step_run(update = 0)'s with that bag's seeds and the window's FIRST global queries (the queries' EMA is chained over the n token sets
 * afterwards: q <- mm^n q + (1 - mm) sum_b mm^(n-1-b) z_b - second order in 1 - merge_mm against the bag-after-bag order).
 * update = 1: Adam + EMA on the summed gradient; update = 0: the summed gradient is left in cfg->g.
 * Workspace: mhimx_window_layout_of(...).total bytes, 256-byte aligned; bag b's copy of a per-bag buffer lies at layout.bag.<field> +
 * b * bag_stride (logits, losses, score, rows_all, ...); bag b's gradient slab (everything but feature.0.weight) at grad_slab + b * bag_stride. */
typedef struct { int64_t total, bag0, bag_stride, grad_slab; mhimx_step_layout bag; } mhimx_window_layout;
int mhimx_window_layout_of(const mhimx_step_cfg* cfg, int32_t n_bags, int64_t N, const mhimx_step_counts* cnt, mhimx_window_layout* out);
int mhimx_window_run(void* stream, const mhimx_step_cfg* cfg, int32_t n_bags, const float* const* X, int64_t ldx, int64_t N,
                     const int64_t* const* labels_dev, const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step, void* ws,
                     int64_t ws_bytes, int32_t update);

/* dst = src (float4 grid-stride stream copy): the on-box HBM copy rate bench.py reports beside the nominal 8 TB/s */
int mhimx_stream_copy(void* stream, const float* src, float* dst, int64_t n_floats);

/* *counter += 1 (device-resident step counters for dropout streams / Adam under hipGraph replay) */
int mhimx_tick(void* stream, uint64_t* counter);

/* ------------------------------------------------------------------------------------------
 * Validation metrics on the device                                   
 * replaces: engines/metrics.py:125-159 (metrics_base: the torchmetrics collection), :104-123 (get_cls_metrics),
 *           :35-78 (DeterministicBootStrapper: B resamples, mean / std taken by the caller)
 * logits [n, C] (row pitch ld), labels [n] int64.  out [B, 7] = Acc (macro), AUC (macro one-vs-rest, exact ROC area),
 * Precision, Recall, F1 (macro), Cohen kappa, Acc_micro.  bin_metric (C == 2): the binary task on logits[:, 1].
 * Predictions not all inside [0,1] go through softmax / sigmoid first (torchmetrics' format rule), per resample.
 * sample_idx [B, n] (NULL with B == 1): row ids of every bootstrap resample. */
int64_t mhimx_cls_metrics_ws_bytes(int64_t n, int64_t C, int64_t B);
int mhimx_cls_metrics(void* stream, const float* logits, int64_t ld, const int64_t* labels, int64_t n, int64_t C,
                      int32_t bin_metric, const int64_t* sample_idx, int64_t B, float* out, void* ws, int64_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Communicator handle: the data-parallel update's ONE collective behind the C boundary.
 * replaces: DistributedDataParallel's bucketed all-reduce (options.py:287, engines/base_engine.py:112-139); RCCL is loaded at run
 * time (dlopen).  unique_id: 128 bytes made on rank 0 and handed to every rank by the host (any side channel); init is collective.
 * allreduce: in-place fp32 SUM on the caller's stream; mode 0 = ncclAllReduce, 1 = reduce-scatter + all-gather (full-mesh form).
 * ---------------------------------------------------------------------------------------- */
typedef struct mhimx_comm mhimx_comm;
int mhimx_comm_unique_id(void* id128);
int mhimx_comm_init(mhimx_comm** out, const void* id128, int32_t rank, int32_t world);
int mhimx_comm_allreduce(mhimx_comm* c, void* stream, float* buf, int64_t count, int32_t mode);
int mhimx_comm_destroy(mhimx_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* MHIMX_H */
