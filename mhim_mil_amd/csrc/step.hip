// step.hip — mhimx_step_run: the whole MHIM(ABMIL) train step of one bag behind ONE C call (include/mhimx.h).
//
// replaces: engines/common_mil.py:14-48 + engines/base_engine.py:76-167 for one bag - and, on this side of the boundary, the Python
// orchestration of mhim_mil_amd/engine.py (FusedTrainer._nat_prep / _nat_bag / _apply): the SAME entry points in the SAME order with the SAME
// arguments, so a step through this executor has the bits of a step through the Python path given the same seeds.  Host code only: every
// launch goes through the library's own extern "C" entry points.  No allocation, no synchronisation: the caller owns the workspace.
//
//   launch  1  mhimx_prep_batch         counters, both projection weight images, the teacher's scorer image (+ the row list's constant tail)
//           2  mhimx_bag_project        teacher + student feature rows in ONE pass over the raw bag
//         3,4  mhimx_abmil_pool_fwd     teacher scorer + pool partials (the student-side preparation jobs ride here) | finalize + pseudo score
//           5  mhimx_select_rows        HAM mask + Merge split, both random subsets drawn on the device
//        6..9  mhimx_merge_fwd          rows pass | partial merge | O | to_out + EMA of the global queries
//       10,11  mhimx_abmil_pool_fwd     student scorer over [stay | tokens] | finalize
//          12  mhimx_head_fwd_bwd       predictor, CE, distillation, their gradients
//          13  mhimx_abmil_pool_bwd     one-pass rows backward (the Merge backward's first stage rides behind its gate)
//          14  mhimx_merge_bwd          rows backward + the parked scorer-weight-gradient product in one launch
//       15,16  rows_dpre_image | bag_wgrad   the projection's gradient pair (the Merge tail and the last reductions ride)
//          17  mhimx_optim_step         Adam + EMA teacher (folds the split-K slab sum)
//
// Round 6 - the step as a DAG (mhimx_step_cfg.side_stream).  The chain above serialises launches that share no data;
// with a second stream the executor forks and joins through events (capturable: the trainer's hipGraph of the step gets parallel branches):
//   forward   after the select:   side: student scorer over the rows that stay   ||  main: Merge rows pass -> partial merge -> O -> to_out
//             join -> the finalize that scores the tokens -> head
//   backward  after the pool backward's rows launch:
//             side: the scorer-weight-gradient product (d_wa) + the reductions queued so far -> [wait: Merge rows backward] -> the Merge
//                   parameter-gradient tail (three stages) + its reductions
//             main: Merge rows backward -> dPRE image -> projection weight gradient (no riders: they are on the side branch)
//             join -> Adam + EMA
// The same entry points with the same arguments on the same buffers - only WHICH launch a rider sits in changes, and riders compute the
// same sums in the same order wherever they run: a DAG step has the bits of a chain step (tests/test_round6_gpu.py).
// MEASURED (profiles/r06_dag.md): on this runtime every fork or join between two queues of a hipGraph costs 5-13 us of idle chip - more than
// the 3-15 us of launches a branch takes off the chain: c2 0.300 -> 0.317 (forward fork alone) / 0.333 (backward alone) / 0.341 ms (both).
// The trainer therefore passes no side stream unless MHIMX_STEP_DAG=1; the form stays as the measured answer to "run the step as a DAG".
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace mhimx {

int64_t merge2_rows_tiles(int64_t R);            // mca2.hip: row tiles of a Merge over R rows

namespace {

// fork / join events of the DAG form: six per device, created on first use (a trainer's first step is eager: not under capture)
struct DagEvents {
  hipEvent_t e[64][6];
  bool have[64] = {};
  int get(hipEvent_t** out) {
    int dev = 0;
    MHIMX_HIP(hipGetDevice(&dev));
    MHIMX_CHECK_ARG(dev >= 0 && dev < 64, "step: device index %d", dev);
    if (!have[dev]) {
      for (int i = 0; i < 6; ++i) MHIMX_HIP(hipEventCreateWithFlags(&e[dev][i], hipEventDisableTiming));
      have[dev] = true;
    }
    *out = e[dev];
    return 0;
  }
};
DagEvents g_dag_events;
// event pairs around the projection launch of eager steps (mhimx_step_cfg.time_project -> mhimx_step_project_ms): a ring per device
constexpr int TP_RING = 256;
struct ProjTimes {
  hipEvent_t e[64][TP_RING][3];            // before | after the launch | right after that one (an EMPTY bracket: what the event pair itself reads)
  bool have[64] = {};
  int n[64] = {};
  int next(int* dev_out, hipEvent_t** pair) {
    int dev = 0;
    MHIMX_HIP(hipGetDevice(&dev));
    MHIMX_CHECK_ARG(dev >= 0 && dev < 64, "step: device index %d", dev);
    if (!have[dev]) {
      for (int i = 0; i < TP_RING; ++i)
        for (int j = 0; j < 3; ++j) MHIMX_HIP(hipEventCreate(&e[dev][i][j]));
      have[dev] = true;
    }
    *dev_out = dev;
    *pair = n[dev] < TP_RING ? e[dev][n[dev]] : nullptr;       // (a full ring: the launch goes unbracketed)
    return 0;
  }
};
ProjTimes g_proj_times;
// `to` continues after everything enqueued on `from` so far
int dag_edge(hipEvent_t ev, hipStream_t from, hipStream_t to) {
  MHIMX_HIP(hipEventRecord(ev, from));
  MHIMX_HIP(hipStreamWaitEvent(to, ev, 0));
  return 0;
}

// MHIMX_FUSE_DPRE=0: the rows' gradient goes to memory in fp32 and mhimx_rows_dpre_image makes the whole image, as rounds 3-5 had it
bool fuse_dpre() {
  static const bool on = getenv("MHIMX_FUSE_DPRE") == nullptr || atoi(getenv("MHIMX_FUSE_DPRE")) != 0;
  return on;
}

struct Carve {
  char* base;
  int64_t off = 0;
  explicit Carve(void* p) : base(static_cast<char*>(p)) {}
  int64_t take_off(int64_t bytes) {
    const int64_t o = off;
    off = (off + bytes + 255) / 256 * 256;
    return o;
  }
  template <typename T> T* at(int64_t o) const { return base ? reinterpret_cast<T*>(base + o) : nullptr; }
};

// every buffer of a step as a byte offset into the workspace
struct StepBufs {
  int64_t w1p_t, wa_frag_t, w1p_s, wa_frag_s, wa_t, wa_t_frag, wo_t, q_old, merge_ws, merge_ws_bytes;
  int64_t H_t, Hbuf, dact;
  int64_t s_t, stats_t, z_t, cproj_t, pscore, attn, pool_ws_t, pool_ws_t_bytes;
  int64_t rows_all, sel_ws, sel_ws_bytes, sel_perm, sel_ids, sel_rows, sel_lk;      // (sel_perm ..: bags of more than 16 384 rows)
  int64_t s_s, stats_s, z_s, pool_ws_s, pool_ws_s_bytes;
  int64_t logits, losses, g_z;
  int64_t dH, img, ws_b, ws_b_bytes, wg_ws, wg_ws_floats;
  int64_t q_scr, gslab;              // (a window's bag: where its forward's query EMA goes, its gradient slab)
  int64_t rows_img, P0, L_img;       // (round 6, fuse_img) the kept rows in the dPRE image's order [stay | 0.. | merge from P0 | 0..], the image's rows
  bool fuse_img;
  int64_t bag0, bag_stride;          // the per-bag part of the workspace: bag b's copy of everything from merge_ws on starts bag0 + b * bag_stride
  int64_t total;
};

// fuse_img: the scorer backward wrote the stay rows' share of the dPRE image (image rows 0 .. P0-1, one partial row of column sums per tile);
// this is the rest - the rows to merge from image row P0 on (mhimx_rows_dpre_image on that part of the image; the Merge backward's tail
// stage still rides there) - and the two sets of column-sum partials become ONE queued reduction.
int dpre_merge_rows(void* stream, const float* dH, const void* dact, const int64_t* rows_merge, int64_t R, int64_t E, char* img, int64_t P0, float* d_b1,
                    float* ws_b, int64_t ws_b_bytes, mhimx_reduce_list* lst) {
  const int64_t t0 = P0 / 32;
  const int n0 = lst->n;
  float* part_m = ws_b + t0 * E;
  if (int r = mhimx_rows_dpre_image(stream, dH, dact, rows_merge, R, E, img + t0 * (E / 128) * 16384, d_b1, 0, part_m, ws_b_bytes - t0 * E * 4, lst)) return r;
  MHIMX_CHECK_ARG(lst->n == n0 + 1 && lst->j[n0].kind == 0 && lst->j[n0].parts == part_m, "step: the bias gradient's partial sum was not queued");
  lst->j[n0].parts = ws_b;
  lst->j[n0].G += t0;
  return 0;
}

int check_cfg(const mhimx_step_cfg* c, int64_t N, const mhimx_step_counts* n) {
  MHIMX_CHECK_ARG(c && n, "step: null configuration / counts");
  MHIMX_CHECK_ARG(c->E == 512 && c->A == 128 && c->C >= 1 && c->C <= 4 && c->k >= 1 && 8 * c->k <= 48 && c->D > 0 && c->D % 256 == 0,
                  "step: shapes outside the single-pass ABMIL step (E = 512, A = 128, C <= 4, 8 k <= 48, D %% 256 == 0)");
  // (round 6: bags beyond the one-workgroup select's 16 384 rows - whole-slide bags, datasets/dataset_feat.py:93-111 - take the multi-workgroup
  // select: select_large below)
  MHIMX_CHECK_ARG(N >= 64 && N <= MHIMX_STEP_MAX_ROWS && n->k_top >= 1 && n->k_top <= (N <= 16384 ? 4096 : 16384) && n->n_sel >= 1 && n->n_sel <= n->k_top &&
                      n->len_keep == N - n->n_sel && n->Lk >= 1 && n->R >= 1 && n->R <= 32768 && n->Lk + n->R == n->len_keep,
                  "step: row counts outside the device-drawn select (64 <= N <= %d, k_top <= 4096 up to 16384 rows / 16384 above, rows to merge "
                  "<= 32768, rows to merge and rows that stay >= 1)", MHIMX_STEP_MAX_ROWS);
  const mhimx_step_params &s = c->student, &t = c->teacher;
  MHIMX_CHECK_ARG(s.w1 && s.b1 && s.wa && s.wc && s.wp && s.bp && s.q && s.ln_w && s.ln_b && s.wkv && s.wq && s.wo && s.bo, "step: null student parameter");
  MHIMX_CHECK_ARG(t.w1 && t.b1 && t.wa && t.wc && (!c->attn2score || (t.wp && t.bp)), "step: null teacher parameter");
  const mhimx_step_grads& g = c->grad;
  MHIMX_CHECK_ARG(g.w1 && g.b1 && g.wa && g.wc && g.wp && g.bp && g.ln_w && g.ln_b && g.wkv && g.wq && g.wo && g.bo, "step: null gradient view");
  MHIMX_CHECK_ARG(c->tick, "step: the device dropout / draw counter (tick) is required");
  return 0;
}

// n_bags > 1: the workspace of an accumulation window (mhimx_window_run) - the parameter images once, ONE weight-gradient workspace for the
// window's multi-bag product, then n_bags copies of the per-bag part (the same offsets as a single step's, plus the bag's query scratch and
// gradient slab) at a fixed stride
void layout(const mhimx_step_cfg* c, int64_t N, const mhimx_step_counts* n, StepBufs* b, int n_bags = 1) {
  const int64_t D = c->D, E = c->E, A = c->A, C = c->C, k = c->k, I = 512;
  Carve cv(nullptr);
  const int64_t F = sizeof(float);
  b->w1p_t = cv.take_off(E * D * F);
  b->wa_frag_t = cv.take_off(A * E * F);
  b->w1p_s = cv.take_off(E * D * F);
  b->wa_frag_s = cv.take_off(A * E * F);
  b->wa_t = cv.take_off(E * A * F);
  b->wa_t_frag = cv.take_off(E * A * F);
  b->wo_t = cv.take_off(I * E * F);
  b->q_old = cv.take_off(k * E * F);
  if (n_bags > 1) {
    b->wg_ws_floats = mhimx_wgrad_multi_ws_floats(fuse_dpre() && N <= 16384 ? (n->Lk + k + 31) / 32 * 32 + n->R : n->len_keep, E, D, n_bags);
    b->wg_ws = cv.take_off(b->wg_ws_floats * F);
  }
  b->bag0 = cv.off;
  b->merge_ws_bytes = mhimx_merge_ws_bytes(n->R, E, k, 8, 64);
  b->merge_ws = cv.take_off(b->merge_ws_bytes);
  // (round 6) up to 16 384 rows the stay rows' share of the projection's dPRE image is written by the scorer backward itself
  // (mhimx_pool_grad.img): the image is ordered [rows that stay, tile for tile of that launch | rows to merge from P0 on]
  b->fuse_img = fuse_dpre() && N <= 16384;
  b->P0 = b->fuse_img ? (n->Lk + k + 31) / 32 * 32 : 0;
  b->L_img = b->fuse_img ? b->P0 + n->R : n->len_keep;
  b->H_t = cv.take_off(N * E * F);
  b->Hbuf = cv.take_off((N + k) * E * F);
  b->dact = cv.take_off(N * E * 2);
  b->s_t = cv.take_off(N * F);
  b->stats_t = cv.take_off(2 * F);
  b->z_t = cv.take_off(E * F);
  b->cproj_t = cv.take_off(N * C * F);
  b->pscore = cv.take_off(N * F);
  b->attn = cv.take_off(N * F);
  b->pool_ws_t_bytes = mhimx_abmil_pool_ws_bytes(N, E, A, 0);
  b->pool_ws_t = cv.take_off(b->pool_ws_t_bytes);
  b->rows_all = cv.take_off((n->len_keep + k) * 8);
  b->sel_ws_bytes = mhimx_select_ws_bytes(N);
  b->sel_ws = cv.take_off(b->sel_ws_bytes);
  b->rows_img = b->fuse_img ? cv.take_off((b->L_img + 31) / 32 * 32 * 8) : 0;
  b->sel_perm = b->sel_ids = b->sel_rows = b->sel_lk = 0;
  if (N > 16384) {
    b->sel_perm = cv.take_off(n->k_top * 8);
    b->sel_ids = cv.take_off(N * 8);
    b->sel_rows = cv.take_off(n->len_keep * 8);
    b->sel_lk = cv.take_off(8);
  }
  const int64_t M = n->Lk + k;
  b->s_s = cv.take_off(M * F);
  b->stats_s = cv.take_off(2 * F);
  b->z_s = cv.take_off(E * F);
  b->pool_ws_s_bytes = mhimx_abmil_pool_ws_bytes(M, E, A, 0);
  b->pool_ws_s = cv.take_off(b->pool_ws_s_bytes);
  b->logits = cv.take_off(16 * F);
  b->losses = cv.take_off(4 * F);
  b->g_z = cv.take_off(E * F);
  b->dH = cv.take_off((N + k) * E * F);
  b->img = cv.take_off(mhimx_wgrad_image_bytes(b->L_img, E));
  b->ws_b_bytes = (b->fuse_img ? b->P0 / 32 + (n->R + 31) / 32 : (n->len_keep + 31) / 32) * E * F;
  b->ws_b = cv.take_off(b->ws_b_bytes);
  b->q_scr = b->gslab = 0;
  if (n_bags > 1) {
    b->q_scr = cv.take_off(k * E * F);
    b->gslab = cv.take_off(c->n_all * F);
  } else {
    b->wg_ws_floats = mhimx_wgrad_ws_floats(b->L_img, E, D);
    b->wg_ws = cv.take_off(b->wg_ws_floats * F);
  }
  b->bag_stride = cv.off - b->bag0;
  b->total = b->bag0 + (int64_t)n_bags * b->bag_stride;
}

}  // namespace

}  // namespace mhimx

using namespace mhimx;

extern "C" int mhimx_step_counts_of(int64_t N, double mask_ratio_h, double mask_ratio_hr, double merge_ratio, mhimx_step_counts* out) {
  MHIMX_CHECK_ARG(out && N >= 1 && mask_ratio_h > 0.0 && mask_ratio_hr > 0.0 && merge_ratio > 0.0, "step_counts: bad arguments");
  // masking.py:30-35: mask_ratio = mask_ratio / random_ratio; if mask_ratio > 1: random_ratio, mask_ratio = mask_ratio_h, 1  (float64, as numpy)
  double eff = mask_ratio_h / mask_ratio_hr, rr = mask_ratio_hr;
  if (eff > 1.0) { rr = mask_ratio_h; eff = 1.0; }
  const int64_t k = (int64_t)ceil((double)N * eff);
  const int64_t n_sel = rr < 1.0 ? (int64_t)ceil((double)k * rr) : k;
  const int64_t len_keep = N - n_sel;
  const int64_t Lk = (int64_t)((double)len_keep * merge_ratio);            // merge.py:163 int(L * merge_ratio)
  *out = mhimx_step_counts{k, n_sel, len_keep, Lk, len_keep - Lk};
  MHIMX_CHECK_ARG(k >= 1 && n_sel >= 1 && len_keep >= 1 && Lk >= 1 && len_keep - Lk >= 1, "step_counts: the recipe leaves nothing to mask, keep or merge");
  return 0;
}

extern "C" int mhimx_step_layout_of(const mhimx_step_cfg* cfg, int64_t N, const mhimx_step_counts* cnt, mhimx_step_layout* out) {
  MHIMX_CHECK_ARG(out, "step_layout: null output");
  if (int r = check_cfg(cfg, N, cnt)) return r;
  StepBufs b;
  layout(cfg, N, cnt, &b);
  *out = mhimx_step_layout{b.total, b.logits, b.losses, cfg->attn2score ? b.pscore : b.attn, b.rows_all, b.H_t, b.Hbuf, b.dact, b.z_t, b.z_s, b.g_z, b.dH};
  return 0;
}

extern "C" int mhimx_step_run(void* stream, const mhimx_step_cfg* cfg, const float* X, int64_t ldx, int64_t N, const int64_t* label_dev,
                              const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step, void* ws, int64_t ws_bytes, int32_t update) {
  if (int r = check_cfg(cfg, N, cnt)) return r;
  MHIMX_CHECK_ARG(X && label_dev && seeds && ws && ldx >= cfg->D && ldx % 4 == 0 && N * ldx * 4 < ((int64_t)1 << 32) && aligned16(X),
                  "step: null bag / label / seeds / workspace, or a row pitch the weight-gradient product does not take");
  MHIMX_CHECK_ARG(!update || (cfg->p && cfg->g && cfg->m && cfg->v && cfg->n_train > 0 && cfg->n_all >= cfg->n_train), "step: update needs the flat optimiser buffers");
  StepBufs b;
  layout(cfg, N, cnt, &b);
  MHIMX_CHECK_ARG(ws_bytes >= b.total && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "step: workspace too small (%lld < %lld) or not 256-byte aligned",
                  (long long)ws_bytes, (long long)b.total);
  const mhimx_step_cfg& c = *cfg;
  const mhimx_step_params &S = c.student, &T = c.teacher;
  const int64_t D = c.D, E = c.E, A = c.A, C = c.C, k = c.k, I = 512;
  const int64_t R = cnt->R, Lk = cnt->Lk, len_keep = cnt->len_keep;
  Carve cv(ws);
  float* w1p_t = cv.at<float>(b.w1p_t);
  float* wa_frag_t = cv.at<float>(b.wa_frag_t);
  float* w1p_s = cv.at<float>(b.w1p_s);
  float* wa_frag_s = cv.at<float>(b.wa_frag_s);
  float* wa_t = cv.at<float>(b.wa_t);
  float* wa_t_frag = cv.at<float>(b.wa_t_frag);
  float* wo_t = cv.at<float>(b.wo_t);
  float* q_old = cv.at<float>(b.q_old);
  void* merge_ws = cv.at<char>(b.merge_ws);
  float* H_t = cv.at<float>(b.H_t);
  float* Hbuf = cv.at<float>(b.Hbuf);
  void* dact = cv.at<char>(b.dact);
  int64_t* rows_all = cv.at<int64_t>(b.rows_all);
  float* dH = cv.at<float>(b.dH);
  const uint64_t* tick = c.tick;
  // the DAG form (header comment): a second stream and its fork / join events.  MHIMX_STEP_DAG=0: the chain, whatever the caller passes
  static const bool dag_on = getenv("MHIMX_STEP_DAG") == nullptr || atoi(getenv("MHIMX_STEP_DAG")) != 0;      // (the caller decides: side_stream)
  static const bool dag_fwd = getenv("MHIMX_STEP_DAG_FWD") == nullptr || atoi(getenv("MHIMX_STEP_DAG_FWD")) != 0;
  static const bool dag_bwd = getenv("MHIMX_STEP_DAG_BWD") == nullptr || atoi(getenv("MHIMX_STEP_DAG_BWD")) != 0;
  const hipStream_t main_st = (hipStream_t)stream, side_st = (hipStream_t)c.side_stream;
  const bool dag = dag_on && c.side_stream != nullptr && c.side_stream != stream;
  hipEvent_t* ev = nullptr;
  if (dag)
    if (int r = g_dag_events.get(&ev)) return r;

  // ---- 1. the step's first launch: counters, both projection weight images, the teacher's scorer image, the row list's constant tail.
  //         Everything else the step prepares (late[]) rides in the teacher's scorer launch, off the head of the chain.
  mhimx_merge mw_prep = {};            // the parameter-only part of the student's Merge (read while enqueueing: prep job kind 6)
  mw_prep.E = E; mw_prep.k = k; mw_prep.heads = 8; mw_prep.dim_head = 64;
  mw_prep.q_param = S.q; mw_prep.ln_w = S.ln_w; mw_prep.ln_b = S.ln_b; mw_prep.wkv = S.wkv; mw_prep.wq = S.wq; mw_prep.wo = S.wo; mw_prep.bo = S.bo;
  mw_prep.mm = c.merge_mm; mw_prep.prec = MHIMX_PREC_BF16X3; mw_prep.drop_tick = tick; mw_prep.rep = 1.f;
  {
    mhimx_prep_job early[8];
    int n = 0;
    early[n++] = mhimx_prep_job{3, nullptr, reinterpret_cast<float*>(c.tick), 1, 1};
    if (c.opt_step) early[n++] = mhimx_prep_job{3, nullptr, reinterpret_cast<float*>(c.opt_step), 1, 1};
    early[n++] = mhimx_prep_job{1, T.w1, w1p_t, E, D};
    early[n++] = mhimx_prep_job{4, T.wa, wa_frag_t, A, E};
    early[n++] = mhimx_prep_job{1, S.w1, w1p_s, E, D};
    early[n++] = mhimx_prep_job{10, nullptr, reinterpret_cast<float*>(rows_all + len_keep), N, k};
    if (int r = mhimx_prep_batch(stream, early, n)) return r;
  }
  mhimx_prep_job late[6];
  late[0] = mhimx_prep_job{4, S.wa, wa_frag_s, A, E};
  late[1] = mhimx_prep_job{0, S.wa, wa_t, A, E};
  late[2] = mhimx_prep_job{5, S.wa, wa_t_frag, A, E};
  late[3] = mhimx_prep_job{0, S.wo, wo_t, E, I};
  late[4] = mhimx_prep_job{2, S.q, q_old, 1, k * E};
  late[5] = mhimx_prep_job{6, reinterpret_cast<const float*>(&mw_prep), static_cast<float*>(merge_ws), R, b.merge_ws_bytes};

  // ---- 2. both models' feature rows in one pass over the raw bag (mhim.py:186 and :335-336)
  {
    mhimx_bag_project_args a = {};
    a.X = X; a.ldx = ldx; a.N = N; a.D = D; a.E = E; a.act = c.act; a.n_heads = 2; a.drop_tick = tick;
    a.head[0].wp = w1p_t; a.head[0].bias = T.b1; a.head[0].H = H_t; a.head[0].ldh = E; a.head[0].drop_p = c.drop_p_teacher; a.head[0].drop_seed = seeds->drop_teacher;
    a.head[1].wp = w1p_s; a.head[1].bias = S.b1; a.head[1].H = Hbuf; a.head[1].ldh = E; a.head[1].dact = dact; a.head[1].drop_p = c.drop_p_student;
    a.head[1].drop_seed = seeds->drop_student;
    hipEvent_t* tp = nullptr;
    int tp_dev = 0;
    if (c.time_project) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      MHIMX_HIP(hipStreamIsCapturing(main_st, &cs));
      if (cs == hipStreamCaptureStatusNone)
        if (int r = g_proj_times.next(&tp_dev, &tp)) return r;
    }
    if (tp) MHIMX_HIP(hipEventRecord(tp[0], main_st));
    if (int r = mhimx_bag_project(stream, &a)) return r;
    if (tp) {
      MHIMX_HIP(hipEventRecord(tp[1], main_st));
      MHIMX_HIP(hipEventRecord(tp[2], main_st));
      ++g_proj_times.n[tp_dev];
    }
  }

  // ---- 3, 4. the teacher: scorer + softmax pool (+ class projections and the pseudo score, scoring.py:37-58)
  mhimx_scorer sc_t = {};
  sc_t.E = E; sc_t.A = A; sc_t.act = c.da_act; sc_t.prec = MHIMX_PREC_BF16X3; sc_t.wa = T.wa; sc_t.wc = T.wc; sc_t.wa_frag = wa_frag_t;
  const float* score = nullptr;
  {
    mhimx_pool_io io = {};
    io.T1 = H_t; io.M1 = N; io.s = cv.at<float>(b.s_t); io.stats = cv.at<float>(b.stats_t); io.z = cv.at<float>(b.z_t);
    io.ws = cv.at<char>(b.pool_ws_t); io.ws_bytes = b.pool_ws_t_bytes;
    if (c.attn2score) { io.wp = T.wp; io.C = C; io.cproj = cv.at<float>(b.cproj_t); io.bp = T.bp; io.pscore = cv.at<float>(b.pscore); }
    io.no_backward = 1;
    io.ride_jobs = late; io.n_ride_jobs = 6;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_t, &io)) return r;
    if (c.attn2score) score = io.pscore;
    else {
      if (int r = mhimx_softmax_from_stats(stream, io.s, io.stats, cv.at<float>(b.attn), N)) return r;
      score = cv.at<float>(b.attn);
    }
  }
  const float* z_t = cv.at<float>(b.z_t);

  // ---- 5. HAM mask + Merge split: rows_all = [rows to merge (R) | rows that stay (Lk) | N .. N + k - 1]
  if (N <= 16384) {
    if (b.fuse_img) {
      if (int r = mhimx_select_rows_img(stream, score, N, cnt->k_top, cnt->n_sel, 1, seeds->select, tick, R, rows_all, cv.at<int64_t>(b.rows_img), b.P0,
                                        cv.at<char>(b.sel_ws), b.sel_ws_bytes, 1))
        return r;
    } else if (int r = mhimx_select_rows(stream, score, N, cnt->k_top, cnt->n_sel, 1, seeds->select, tick, R, rows_all, nullptr, cv.at<char>(b.sel_ws),
                                         b.sel_ws_bytes, 1))
      return r;
  } else {
    // select_large: the two-stage form of masking.py:61-86 + merge.py:163-170 as MHIM.student_rows issues it for such bags (mhim.py of this
    // package: the same four launches + the [merge | stay] swap, the same seeds - the bits of the Python path):
    //   perm  = pi_1 of 0 .. k_top-1 (masking.py:67's torch.randperm, keyed by (seed + 0x51ED270B, tick))
    //   ids   = select_mask(score, k_top, n_sel, perm)          [kept ascending | masked]   (multi-workgroup select)
    //   rows  = ids[pi_2(j)], j < len_keep                       (merge.py:165's shuffle of the kept rows, keyed by (seed ^ 0x3C6E.., tick))
    //   rows_all = [rows[Lk:] (to merge) | rows[:Lk] (stay)]
    int64_t* perm = cv.at<int64_t>(b.sel_perm);
    int64_t* ids = cv.at<int64_t>(b.sel_ids);
    int64_t* rows = cv.at<int64_t>(b.sel_rows);
    if (int r = mhimx_random_perm(stream, cnt->k_top, seeds->select + 0x51ED270Bull, tick, nullptr, perm)) return r;
    if (int r = mhimx_select_mask(stream, score, N, cnt->k_top, cnt->n_sel, 1, cnt->n_sel < cnt->k_top ? perm : nullptr, nullptr, 0, ids, cv.at<int64_t>(b.sel_lk),
                                  nullptr, cv.at<char>(b.sel_ws), b.sel_ws_bytes))
      return r;
    if (int r = mhimx_random_perm(stream, len_keep, seeds->select ^ 0x3C6EF372FE94F82Bull, tick, ids, rows)) return r;
    MHIMX_HIP(hipMemcpyAsync(rows_all, rows + Lk, (size_t)R * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    MHIMX_HIP(hipMemcpyAsync(rows_all + R, rows, (size_t)Lk * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }

  // ---- 6..9. Merge (merge.py:127-203): the k tokens land behind the bag's rows, the queries' EMA in place
  mhimx_merge mw = mw_prep;
  mw.drop_p = c.merge_drop_p; mw.drop_seed = seeds->mca; mw.x_rows = rows_all; mw.prepared = 1;
  // ---- 6..11. the student's forward.  Its scorer over the rows that stay and Merge's rows pass are independent: ONE launch runs both (the Merge
  //      row tiles ride at its front), the Merge tail then makes the tokens, and the pool's finalize launch scores those k rows itself
  //      (mhimx_pool_io.phase; MHIMX_SPLIT_POOL=0: Merge, then the scorer over [stay | tokens], as rounds 2-4 had it)
  mhimx_scorer sc_s = sc_t;
  sc_s.wa = S.wa; sc_s.wc = S.wc; sc_s.wa_frag = wa_frag_s;
  mhimx_pool_io io_s = {};
  io_s.T1 = Hbuf; io_s.M1 = Lk + k; io_s.s = cv.at<float>(b.s_s); io_s.stats = cv.at<float>(b.stats_s); io_s.z = cv.at<float>(b.z_s);
  io_s.ws = cv.at<char>(b.pool_ws_s); io_s.ws_bytes = b.pool_ws_s_bytes; io_s.rows1 = rows_all + R;
  static const bool split_pool = getenv("MHIMX_SPLIT_POOL") == nullptr || atoi(getenv("MHIMX_SPLIT_POOL")) != 0;
  io_s.tail_row0 = -1;
  if (dag && dag_fwd && split_pool && k <= 6) {
    // two branches: the scorer over the rows that stay (side) || Merge's whole chain (main); they meet at the finalize that scores the tokens
    if (int r = dag_edge(ev[0], main_st, side_st)) return r;
    io_s.phase = 1; io_s.tail_tokens = (int32_t)k;
    if (int r = mhimx_abmil_pool_fwd(side_st, &sc_s, &io_s)) return r;
    if (int r = mhimx_merge_fwd(stream, &mw, Hbuf, R, Hbuf + N * E, c.q_out ? c.q_out : S.q, 1, merge_ws, b.merge_ws_bytes)) return r;
    if (int r = dag_edge(ev[1], side_st, main_st)) return r;
    io_s.phase = 2; io_s.tail_wa_t = wa_t; io_s.tail_row0 = N;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
    io_s.phase = 0;
  } else if (split_pool && k <= 6) {
    io_s.phase = 1; io_s.tail_tokens = (int32_t)k;
    io_s.ride_merge = &mw; io_s.ride_X = Hbuf; io_s.ride_R = R; io_s.ride_ws = merge_ws; io_s.ride_ws_bytes = b.merge_ws_bytes;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
    mw.rows_done = io_s.rode_merge;
    if (int r = mhimx_merge_fwd(stream, &mw, Hbuf, R, Hbuf + N * E, c.q_out ? c.q_out : S.q, 1, merge_ws, b.merge_ws_bytes)) return r;
    mw.rows_done = 0;
    io_s.phase = 2; io_s.tail_wa_t = wa_t; io_s.tail_row0 = N;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
    io_s.phase = 0; io_s.ride_merge = nullptr;
  } else {
    if (int r = mhimx_merge_fwd(stream, &mw, Hbuf, R, Hbuf + N * E, c.q_out ? c.q_out : S.q, 1, merge_ws, b.merge_ws_bytes)) return r;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
  }

  // ---- 12. head: predictor, CE, distillation against the teacher's bag feature, and their gradients
  float* g_z = cv.at<float>(b.g_z);
  if (int r = mhimx_head_fwd_bwd(stream, io_s.z, c.aux_alpha != 0.f ? z_t : nullptr, S.wp, S.bp, label_dev, E, C, c.temp_t, c.main_alpha, c.aux_alpha, 1.f,
                                 cv.at<float>(b.logits), cv.at<float>(b.losses), g_z, c.grad.wp, c.grad.bp, 0, nullptr, nullptr))
    return r;

  // ---- 13..16. backward
  mhimx_reduce_list lst;
  memset(&lst, 0, sizeof(lst));
  mhimx_merge mwb = mw;
  mwb.q_param = q_old; mwb.wo_t = wo_t; mwb.prepared = 0;
  mhimx_merge_grad mg = {};
  mg.d_ln_w = c.grad.ln_w; mg.d_ln_b = c.grad.ln_b; mg.d_wkv = c.grad.wkv; mg.d_wq = c.grad.wq; mg.d_wo = c.grad.wo; mg.d_bo = c.grad.bo;
  mg.accumulate = 0; mg.splits = 8; mg.defer = &lst;
  if (int r = mhimx_merge_bwd_park(&mwb, Hbuf, R, dH + N * E, dH, &mg, merge_ws, b.merge_ws_bytes)) return r;
  {
    mhimx_scorer sc_b = sc_s;
    sc_b.wa_frag = nullptr;
    mhimx_pool_grad pg = {};
    pg.g_z = g_z; pg.dT1 = dH; pg.d_wa = c.grad.wa; pg.d_wc = c.grad.wc; pg.wa_t = wa_t; pg.accumulate = 0; pg.splits = 8; pg.defer = &lst; pg.wa_t_frag = wa_t_frag;
    if (b.fuse_img) { pg.img = cv.at<char>(b.img); pg.img_dact = dact; pg.img_part = cv.at<float>(b.ws_b); pg.img_rows = Lk; }
    if (int r = mhimx_abmil_pool_bwd(stream, &sc_b, &io_s, &pg)) return r;
  }
  mhimx_reduce_list lst_main;                    // (DAG form: what the main branch queues - the bias partials, the weight gradient's slab sum)
  memset(&lst_main, 0, sizeof(lst_main));
  const bool dagb = dag && dag_bwd;
  mhimx_reduce_list* lm = dagb ? &lst_main : &lst;
  if (dagb) {
    // side branch, first half: the parked scorer-weight-gradient product (its inputs are the pool backward's) and the reductions queued so far
    if (int r = dag_edge(ev[2], main_st, side_st)) return r;
    // (the chain runs the product beside the Merge rows backward's tiles and sizes it for the CUs they leave - when the first stage rode:
    // the same slab count here keeps the chain's summation order)
    static const bool fuse_rows = getenv("MHIMX_MERGE_BWD_FUSE") == nullptr || atoi(getenv("MHIMX_MERGE_BWD_FUSE")) != 0;
    if (lst.parked.pending && lst.pre.pending == 2 && fuse_rows) lst.parked.reserved = (int32_t)merge2_rows_tiles(R) + 1;
    if (int r = mhimx_reduce_flush(side_st, &lst)) return r;
  }
  // main: the Merge rows backward (DAG form: a launch of its own - the product it used to share a launch with is on the side branch)
  if (int r = mhimx_merge_bwd(stream, &mwb, Hbuf, R, dH + N * E, dH, &mg, merge_ws, b.merge_ws_bytes)) return r;
  if (dagb) {
    // side branch, second half: the Merge parameter-gradient tail reads what the rows backward wrote (the pooled-row partials U)
    if (int r = dag_edge(ev[3], main_st, side_st)) return r;
    if (int r = mhimx_reduce_flush(side_st, &lst)) return r;      // (the tail's stages as launches, the last one inside the reduction launch)
  }
  if (b.fuse_img) {
    if (int r = dpre_merge_rows(stream, dH, dact, rows_all, R, E, cv.at<char>(b.img), b.P0, c.grad.b1, cv.at<float>(b.ws_b), b.ws_b_bytes, lm)) return r;
  } else if (int r = mhimx_rows_dpre_image(stream, dH, dact, rows_all, len_keep, E, cv.at<char>(b.img), c.grad.b1, 0, cv.at<char>(b.ws_b), b.ws_b_bytes, lm))
    return r;
  {
    mhimx_bag_wgrad_args g = {};
    g.img = cv.at<char>(b.img); g.X = X; g.ldx = ldx; g.n_bag_rows = N; g.rows = b.fuse_img ? cv.at<int64_t>(b.rows_img) : rows_all; g.L = b.L_img; g.E = E; g.D = D;
    g.C = c.grad.w1; g.ldc = D;
    g.accumulate = 0; g.ws = cv.at<float>(b.wg_ws); g.ws_floats = b.wg_ws_floats; g.defer = lm; g.ride_tail = update ? 1 : 0;
    if (int r = mhimx_bag_wgrad(stream, &g)) return r;
  }
  if (dagb)
    if (int r = dag_edge(ev[4], side_st, main_st)) return r;
  if (!update) return mhimx_reduce_flush(stream, lm);

  // ---- 17. Adam + EMA teacher; the weight gradient's split-K slab sum is folded into the update
  mhimx_optim_args o = {};
  o.p = c.p; o.g = c.g; o.m = c.m; o.v = c.v; o.teacher = c.p_teacher; o.n_train = c.n_train; o.n_all = c.n_all; o.step = host_step; o.step_dev = c.opt_step;
  o.lr = c.lr; o.lr_table = c.lr_table; o.lr_len = c.lr_len; o.beta1 = c.beta1; o.beta2 = c.beta2; o.eps = c.eps; o.weight_decay = c.weight_decay;
  o.grad_scale = 1.f; o.ema_mm = c.ema_mm; o.mm_table = c.mm_table; o.mm_len = c.mm_len; o.zero_grad = 1; o.fold = lm;
  return mhimx_optim_step(stream, &o);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 6 - an accumulation window with every launch over ALL its bags (base_engine.py:29,47-49,100-119).
//   prep (once)  ->  both projections of the n bags in ONE launch  ->  the step's middle (teacher scorer ... dPRE image: launches 3-15 of
//   the list at the top of this file) issued ONCE with gridDim.z = n (common.hpp: BagBatch - every kernel moves the pointers it was given to
//   its own bag's copy of the workspace)  ->  ONE weight-gradient launch over the n images  ->  the queries' EMA chain  ->  Adam + EMA.
// ~22 launches per window instead of ~127: the latency-bound links of the chain (select, Merge tail, finalizes, head: one to a few dozen
// workgroups each) run for 8 bags in the time of one, which HIP streams / graph branches never delivered (at most two queues make progress
// at a time on this runtime: profiles/r06_window_batched.md).  Same kernels, same arithmetic per bag as mhimx_step_run(update = 0) with
// that bag's seeds; the forward has its bits, the gradient differs only where a split-K slab count follows the launch's size.
// ------------------------------------------------------------------------------------------------------------------------------------
namespace mhimx {
namespace {

// q <- wq q + sum_b w[b] z_b   (the window's EMA chain of the global queries on the tokens its forwards produced: merge.py:142-143 applied bag
// after bag, q <- mm q + (1 - mm) z_b, with every z_b computed from the window's first queries - engine.py window_step's contract)
struct QChainW { float wq; float w[MHIMX_WINDOW_MAX]; };
// ... and, in the same launch, zeros for the elements of the bags' gradient slabs that NO gradient view covers (the flat buffer pads every
// tensor to 16 bytes - predictor.bias [2] leaves two floats - and may hold parameters this step does not train): nobody writes them, the
// update adds them.  (Found by the window tests: stale workspace memory there made two runs of one window differ in two elements.)
struct SlabGaps { int n; int64_t off[16], len[16]; };
__global__ __launch_bounds__(256) void window_q_chain_kernel(float* __restrict__ q, const float* __restrict__ z0, int64_t z_stride, int n_bags, int64_t n,
                                                             QChainW cw, float* __restrict__ slab0, int64_t pitch, SlabGaps gaps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < n_bags; ++b) acc += z0[b * z_stride + i] * cw.w[b];
    q[i] = q[i] * cw.wq + acc;
  }
  for (int gi = 0; gi < gaps.n; ++gi)
    for (int b = 0; b < n_bags; ++b)
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < gaps.len[gi]; i += (int64_t)gridDim.x * blockDim.x)
        slab0[b * pitch + gaps.off[gi] + i] = 0.f;
}
// g = sum_b slab_b   (update = 0: the window's complete gradient in the flat buffer; bag order, as the update kernel adds them)
__global__ __launch_bounds__(256) void window_sum_slabs_kernel(float* __restrict__ g, const float* __restrict__ slab0, int64_t pitch, int n_bags, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < n_bags; ++b) acc += slab0[b * pitch + i];
    g[i] = acc;
  }
}

int check_window(const mhimx_step_cfg* c, int32_t n_bags, int64_t N, const mhimx_step_counts* n) {
  if (int r = check_cfg(c, N, n)) return r;
  MHIMX_CHECK_ARG(n_bags >= 2 && n_bags <= MHIMX_WINDOW_MAX, "window: 2..%d bags", MHIMX_WINDOW_MAX);
  MHIMX_CHECK_ARG(N <= 16384 && c->k <= 6, "window: bags of up to 16384 rows (the one-workgroup select), merge_k <= 6");
  MHIMX_CHECK_ARG(c->g && c->n_train > 0 && c->n_all >= c->n_train && c->n_all % 4 == 0, "window: the flat gradient buffer (n_all %% 4 == 0) is required");
  MHIMX_CHECK_ARG(!c->q_out && !c->side_stream, "window: q_out / side_stream are single-step options");
  const mhimx_step_grads& g = c->grad;
  const float* gp[12] = {g.w1, g.b1, g.wa, g.wc, g.wp, g.bp, g.ln_w, g.ln_b, g.wkv, g.wq, g.wo, g.bo};
  for (int i = 0; i < 12; ++i)
    MHIMX_CHECK_ARG(gp[i] >= c->g && gp[i] < c->g + c->n_train, "window: every gradient view lies inside the flat gradient buffer g[0, n_train)");
  MHIMX_CHECK_ARG(g.w1 == c->g && (c->E * c->D) % 4 == 0, "window: feature.0.weight's gradient is the FIRST block of the flat gradient buffer");
  for (int i = 1; i < 12; ++i) MHIMX_CHECK_ARG(gp[i] >= c->g + c->E * c->D, "window: gradient views overlap feature.0.weight's");
  return 0;
}

}  // namespace
}  // namespace mhimx

extern "C" int mhimx_window_layout_of(const mhimx_step_cfg* cfg, int32_t n_bags, int64_t N, const mhimx_step_counts* cnt, mhimx_window_layout* out) {
  MHIMX_CHECK_ARG(out, "window_layout: null output");
  if (int r = check_window(cfg, n_bags, N, cnt)) return r;
  StepBufs b;
  layout(cfg, N, cnt, &b, n_bags);
  out->total = b.total; out->bag0 = b.bag0; out->bag_stride = b.bag_stride; out->grad_slab = b.gslab;
  out->bag = mhimx_step_layout{b.total, b.logits, b.losses, cfg->attn2score ? b.pscore : b.attn, b.rows_all, b.H_t, b.Hbuf, b.dact, b.z_t, b.z_s, b.g_z, b.dH};
  return 0;
}

extern "C" int mhimx_window_run(void* stream, const mhimx_step_cfg* cfg, int32_t n_bags, const float* const* X, int64_t ldx, int64_t N,
                                const int64_t* const* labels_dev, const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step, void* ws,
                                int64_t ws_bytes, int32_t update) {
  if (int r = check_window(cfg, n_bags, N, cnt)) return r;
  MHIMX_CHECK_ARG(X && labels_dev && seeds && ws && ldx >= cfg->D && ldx % 4 == 0 && N * ldx * 4 < ((int64_t)1 << 32),
                  "window: null bags / labels / seeds / workspace, or a row pitch the weight-gradient product does not take");
  for (int32_t i = 0; i < n_bags; ++i) MHIMX_CHECK_ARG(X[i] && aligned16(X[i]) && labels_dev[i], "window: bag %d / its label null or not 16-byte aligned", i);
  MHIMX_CHECK_ARG(!update || (cfg->p && cfg->m && cfg->v), "window: update needs the flat optimiser buffers");
  StepBufs b;
  layout(cfg, N, cnt, &b, n_bags);
  MHIMX_CHECK_ARG(ws_bytes >= b.total && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "window: workspace too small (%lld < %lld) or not 256-byte aligned",
                  (long long)ws_bytes, (long long)b.total);
  const mhimx_step_cfg& c = *cfg;
  const mhimx_step_params &S = c.student, &T = c.teacher;
  const int64_t D = c.D, E = c.E, A = c.A, C = c.C, k = c.k, I = 512;
  const int64_t R = cnt->R, Lk = cnt->Lk, len_keep = cnt->len_keep;
  const hipStream_t st = (hipStream_t)stream;
  Carve cv(ws);
  const int64_t BS = b.bag_stride;                                     // bytes from a bag's copy of a buffer to the next bag's
  auto bag = [&](int64_t off, int i) { return cv.at<char>(off) + (int64_t)i * BS; };
  float* w1p_t = cv.at<float>(b.w1p_t);
  float* wa_frag_t = cv.at<float>(b.wa_frag_t);
  float* w1p_s = cv.at<float>(b.w1p_s);
  float* wa_frag_s = cv.at<float>(b.wa_frag_s);
  float* wa_t = cv.at<float>(b.wa_t);
  float* wa_t_frag = cv.at<float>(b.wa_t_frag);
  float* wo_t = cv.at<float>(b.wo_t);
  float* q_old = cv.at<float>(b.q_old);
  // bag 0's copies: what the middle is enqueued with
  void* merge_ws = cv.at<char>(b.merge_ws);
  float* H_t = cv.at<float>(b.H_t);
  float* Hbuf = cv.at<float>(b.Hbuf);
  int64_t* rows_all = cv.at<int64_t>(b.rows_all);
  float* dH = cv.at<float>(b.dH);
  float* slab0 = cv.at<float>(b.gslab);
  const uint64_t* tick = c.tick;
  auto gs = [&](float* view) { return slab0 + (view - c.g); };          // a gradient view's place in bag 0's slab

  // the flat gradient takes the weight gradient of the projection (the window's ONE product writes it: elements [0, E D)) and nothing else;
  // from E D on the gradient is the sum of the bags' slabs - taken by the update kernel itself (g_extra, extra_lo, extra_only) or written
  // into the flat buffer by window_sum_slabs_kernel (update = 0)
  const int64_t g_lo = E * D;

  // ---- 1. preparation, once for the window (nothing rides: the middle's launches are bag-batched)
  mhimx_merge mw_prep = {};
  mw_prep.E = E; mw_prep.k = k; mw_prep.heads = 8; mw_prep.dim_head = 64;
  mw_prep.q_param = S.q; mw_prep.ln_w = S.ln_w; mw_prep.ln_b = S.ln_b; mw_prep.wkv = S.wkv; mw_prep.wq = S.wq; mw_prep.wo = S.wo; mw_prep.bo = S.bo;
  mw_prep.mm = c.merge_mm; mw_prep.prec = MHIMX_PREC_BF16X3; mw_prep.drop_tick = tick; mw_prep.rep = 1.f;
  {
    mhimx_prep_job jobs[MHIMX_PREP_MAX];
    int n = 0;
    jobs[n++] = mhimx_prep_job{3, nullptr, reinterpret_cast<float*>(c.tick), 1, 1};
    if (c.opt_step) jobs[n++] = mhimx_prep_job{3, nullptr, reinterpret_cast<float*>(c.opt_step), 1, 1};
    jobs[n++] = mhimx_prep_job{1, T.w1, w1p_t, E, D};
    jobs[n++] = mhimx_prep_job{4, T.wa, wa_frag_t, A, E};
    jobs[n++] = mhimx_prep_job{1, S.w1, w1p_s, E, D};
    jobs[n++] = mhimx_prep_job{4, S.wa, wa_frag_s, A, E};
    jobs[n++] = mhimx_prep_job{0, S.wa, wa_t, A, E};
    jobs[n++] = mhimx_prep_job{5, S.wa, wa_t_frag, A, E};
    jobs[n++] = mhimx_prep_job{0, S.wo, wo_t, E, I};
    jobs[n++] = mhimx_prep_job{2, S.q, q_old, 1, k * E};
    for (int i = 0; i < n_bags; ++i) {
      jobs[n++] = mhimx_prep_job{10, nullptr, reinterpret_cast<float*>(reinterpret_cast<int64_t*>(bag(b.rows_all, i)) + len_keep), N, k};
      jobs[n++] = mhimx_prep_job{6, reinterpret_cast<const float*>(&mw_prep), reinterpret_cast<float*>(bag(b.merge_ws, i)), R, b.merge_ws_bytes};
    }
    static_assert(10 + 2 * MHIMX_WINDOW_MAX <= MHIMX_PREP_MAX, "the window's preparation is one launch");
    if (int r = mhimx_prep_batch(stream, jobs, n)) return r;
  }

  // ---- 2. both models' feature rows of every bag: ONE launch (mhim.py:186 and :335-336)
  {
    mhimx_bag_project_args pa[MHIMX_WINDOW_MAX];
    // the launch projects the bags LAST to FIRST: the middle's planes are dispatched bag 0 first, and its first readers then find the rows
    // written most recently (the window's 400 MB of feature rows do not fit the 256 MB Infinity Cache).  Same bits; 0.5 % of a window,
    // same box, twice (MHIMX_WINDOW_PROJ_REV=0: first to last)
    static const bool proj_rev = getenv("MHIMX_WINDOW_PROJ_REV") == nullptr || atoi(getenv("MHIMX_WINDOW_PROJ_REV")) != 0;
    for (int i = 0; i < n_bags; ++i) {
      mhimx_bag_project_args a = {};
      a.X = X[i]; a.ldx = ldx; a.N = N; a.D = D; a.E = E; a.act = c.act; a.n_heads = 2; a.drop_tick = tick;
      a.head[0].wp = w1p_t; a.head[0].bias = T.b1; a.head[0].H = reinterpret_cast<float*>(bag(b.H_t, i)); a.head[0].ldh = E; a.head[0].drop_p = c.drop_p_teacher;
      a.head[0].drop_seed = seeds[i].drop_teacher;
      a.head[1].wp = w1p_s; a.head[1].bias = S.b1; a.head[1].H = reinterpret_cast<float*>(bag(b.Hbuf, i)); a.head[1].ldh = E; a.head[1].dact = bag(b.dact, i);
      a.head[1].drop_p = c.drop_p_student; a.head[1].drop_seed = seeds[i].drop_student;
      pa[proj_rev ? n_bags - 1 - i : i] = a;
    }
    if (int r = mhimx_bag_project_multi(stream, pa, n_bags)) return r;
  }

  // ---- 3..15. the middle, enqueued once with bag 0's pointers, gridDim.z = n_bags
  BagBatch bb = {};
  bb.n = n_bags;
  bb.lo[0] = reinterpret_cast<uint64_t>(cv.at<char>(b.bag0)); bb.span[0] = (uint64_t)BS; bb.stride[0] = BS;
  bb.tab_key = reinterpret_cast<uint64_t>(labels_dev[0]);
  for (int i = 0; i < n_bags; ++i) {
    bb.dsel[i] = seeds[i].select - seeds[0].select;
    bb.dmca[i] = seeds[i].mca - seeds[0].mca;
    bb.tab[i] = reinterpret_cast<uint64_t>(labels_dev[i]);
  }
  struct BatchScope {                                                 // (every return path leaves the thread without a batch)
    explicit BatchScope(const BagBatch* p) { set_batch(p); }
    ~BatchScope() { set_batch(nullptr); }
  };
  {
    BatchScope scope(&bb);
    // teacher: scorer + softmax pool (+ class projections and the pseudo score, scoring.py:37-58)
    mhimx_scorer sc_t = {};
    sc_t.E = E; sc_t.A = A; sc_t.act = c.da_act; sc_t.prec = MHIMX_PREC_BF16X3; sc_t.wa = T.wa; sc_t.wc = T.wc; sc_t.wa_frag = wa_frag_t;
    const float* score = nullptr;
    {
      mhimx_pool_io io = {};
      io.T1 = H_t; io.M1 = N; io.s = cv.at<float>(b.s_t); io.stats = cv.at<float>(b.stats_t); io.z = cv.at<float>(b.z_t);
      io.ws = cv.at<char>(b.pool_ws_t); io.ws_bytes = b.pool_ws_t_bytes;
      if (c.attn2score) { io.wp = T.wp; io.C = C; io.cproj = cv.at<float>(b.cproj_t); io.bp = T.bp; io.pscore = cv.at<float>(b.pscore); }
    io.no_backward = 1;
      if (int r = mhimx_abmil_pool_fwd(stream, &sc_t, &io)) return r;
      if (c.attn2score) score = io.pscore;
      else {
        if (int r = mhimx_softmax_from_stats(stream, io.s, io.stats, cv.at<float>(b.attn), N)) return r;
        score = cv.at<float>(b.attn);
      }
    }
    const float* z_t = cv.at<float>(b.z_t);
    // HAM mask + Merge split
    if (b.fuse_img) {
      if (int r = mhimx_select_rows_img(stream, score, N, cnt->k_top, cnt->n_sel, 1, seeds[0].select, tick, R, rows_all, cv.at<int64_t>(b.rows_img), b.P0,
                                        cv.at<char>(b.sel_ws), b.sel_ws_bytes, 1))
        return r;
    } else if (int r = mhimx_select_rows(stream, score, N, cnt->k_top, cnt->n_sel, 1, seeds[0].select, tick, R, rows_all, nullptr, cv.at<char>(b.sel_ws),
                                         b.sel_ws_bytes, 1))
      return r;
    // the student's forward: scorer over the rows that stay (Merge's row tiles ride at its front), Merge tail, the finalize that scores the tokens
    mhimx_merge mw = mw_prep;
    mw.drop_p = c.merge_drop_p; mw.drop_seed = seeds[0].mca; mw.x_rows = rows_all; mw.prepared = 1;
    mhimx_scorer sc_s = sc_t;
    sc_s.wa = S.wa; sc_s.wc = S.wc; sc_s.wa_frag = wa_frag_s;
    mhimx_pool_io io_s = {};
    io_s.T1 = Hbuf; io_s.M1 = Lk + k; io_s.s = cv.at<float>(b.s_s); io_s.stats = cv.at<float>(b.stats_s); io_s.z = cv.at<float>(b.z_s);
    io_s.ws = cv.at<char>(b.pool_ws_s); io_s.ws_bytes = b.pool_ws_s_bytes; io_s.rows1 = rows_all + R;
    io_s.tail_row0 = -1;
    io_s.phase = 1; io_s.tail_tokens = (int32_t)k;
    io_s.ride_merge = &mw; io_s.ride_X = Hbuf; io_s.ride_R = R; io_s.ride_ws = merge_ws; io_s.ride_ws_bytes = b.merge_ws_bytes;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
    mw.rows_done = io_s.rode_merge;
    if (int r = mhimx_merge_fwd(stream, &mw, Hbuf, R, Hbuf + N * E, cv.at<float>(b.q_scr), 1, merge_ws, b.merge_ws_bytes)) return r;
    mw.rows_done = 0;
    io_s.phase = 2; io_s.tail_wa_t = wa_t; io_s.tail_row0 = N;
    if (int r = mhimx_abmil_pool_fwd(stream, &sc_s, &io_s)) return r;
    io_s.phase = 0; io_s.ride_merge = nullptr;

    // head: every bag's loss / n_bags (base_engine.py:102)
    float* g_z = cv.at<float>(b.g_z);
    if (int r = mhimx_head_fwd_bwd(stream, io_s.z, c.aux_alpha != 0.f ? z_t : nullptr, S.wp, S.bp, labels_dev[0], E, C, c.temp_t, c.main_alpha, c.aux_alpha,
                                   1.f / (float)n_bags, cv.at<float>(b.logits), cv.at<float>(b.losses), g_z, gs(c.grad.wp), gs(c.grad.bp), 0, nullptr, nullptr))
      return r;

    // backward up to the dPRE image; every parameter gradient but the projection weight's lands in the bag's slab
    mhimx_reduce_list lst;
    memset(&lst, 0, sizeof(lst));
    mhimx_merge mwb = mw;
    mwb.q_param = q_old; mwb.wo_t = wo_t; mwb.prepared = 0;
    mhimx_merge_grad mg = {};
    mg.d_ln_w = gs(c.grad.ln_w); mg.d_ln_b = gs(c.grad.ln_b); mg.d_wkv = gs(c.grad.wkv); mg.d_wq = gs(c.grad.wq); mg.d_wo = gs(c.grad.wo); mg.d_bo = gs(c.grad.bo);
    mg.accumulate = 0; mg.splits = 8; mg.defer = &lst;
    if (int r = mhimx_merge_bwd_park(&mwb, Hbuf, R, dH + N * E, dH, &mg, merge_ws, b.merge_ws_bytes)) return r;
    {
      mhimx_scorer sc_b = sc_s;
      sc_b.wa_frag = nullptr;
      mhimx_pool_grad pg = {};
      pg.g_z = g_z; pg.dT1 = dH; pg.d_wa = gs(c.grad.wa); pg.d_wc = gs(c.grad.wc); pg.wa_t = wa_t; pg.accumulate = 0; pg.splits = 8; pg.defer = &lst;
      pg.wa_t_frag = wa_t_frag;
      if (b.fuse_img) { pg.img = cv.at<char>(b.img); pg.img_dact = cv.at<char>(b.dact); pg.img_part = cv.at<float>(b.ws_b); pg.img_rows = Lk; }
      if (int r = mhimx_abmil_pool_bwd(stream, &sc_b, &io_s, &pg)) return r;
    }
    if (int r = mhimx_merge_bwd(stream, &mwb, Hbuf, R, dH + N * E, dH, &mg, merge_ws, b.merge_ws_bytes)) return r;
    if (b.fuse_img) {
      if (int r = dpre_merge_rows(stream, dH, cv.at<char>(b.dact), rows_all, R, E, cv.at<char>(b.img), b.P0, gs(c.grad.b1), cv.at<float>(b.ws_b), b.ws_b_bytes, &lst))
        return r;
    } else if (int r = mhimx_rows_dpre_image(stream, dH, cv.at<char>(b.dact), rows_all, len_keep, E, cv.at<char>(b.img), gs(c.grad.b1), 0, cv.at<char>(b.ws_b),
                                             b.ws_b_bytes, &lst))
      return r;
    if (int r = mhimx_reduce_flush(stream, &lst)) return r;          // (the Merge tail's remaining stages + every queued reduction, bag-batched)
  }

  // ---- 16. dW1 = sum_bags dPRE_b^T X_b: ONE launch, straight into the flat gradient
  mhimx_reduce_list lst_w;
  memset(&lst_w, 0, sizeof(lst_w));
  {
    mhimx_bag_wgrad_args ga[MHIMX_WINDOW_MAX];
    for (int i = 0; i < n_bags; ++i) {
      mhimx_bag_wgrad_args g = {};
      g.img = bag(b.img, i); g.X = X[i]; g.ldx = ldx; g.n_bag_rows = N; g.rows = reinterpret_cast<const int64_t*>(bag(b.fuse_img ? b.rows_img : b.rows_all, i));
      g.L = b.L_img; g.E = E; g.D = D;
      g.C = c.grad.w1; g.ldc = D; g.accumulate = 0; g.ws = cv.at<float>(b.wg_ws); g.ws_floats = b.wg_ws_floats; g.defer = &lst_w;
      ga[i] = g;
    }
    if (int r = mhimx_bag_wgrad_multi(stream, ga, n_bags)) return r;
  }
  // ---- the queries' EMA chain over the window's tokens: q <- mm^n q + (1 - mm) sum_b mm^(n-1-b) z_b
  {
    QChainW cw;
    const double mm = (double)c.merge_mm;
    cw.wq = (float)pow(mm, (double)n_bags);
    for (int i = 0; i < MHIMX_WINDOW_MAX; ++i) cw.w[i] = i < n_bags ? (float)((1.0 - mm) * pow(mm, (double)(n_bags - 1 - i))) : 0.f;
    // the slab elements from E D on that none of the eleven gradient views covers
    SlabGaps gaps = {};
    {
      struct View { int64_t off, len; };
      const mhimx_step_grads& gr = c.grad;
      View v[11] = {{gr.b1 - c.g, E}, {gr.wa - c.g, A * E}, {gr.wc - c.g, A}, {gr.wp - c.g, C * E}, {gr.bp - c.g, C}, {gr.ln_w - c.g, E}, {gr.ln_b - c.g, E},
                    {gr.wkv - c.g, 2 * I * E}, {gr.wq - c.g, I * E}, {gr.wo - c.g, E * I}, {gr.bo - c.g, E}};
      for (int i = 1; i < 11; ++i)                                     // (insertion sort by offset)
        for (int j = i; j > 0 && v[j].off < v[j - 1].off; --j) { const View t = v[j]; v[j] = v[j - 1]; v[j - 1] = t; }
      int64_t at = g_lo;
      for (int i = 0; i <= 11; ++i) {
        const int64_t nxt = i < 11 ? v[i].off : c.n_train;
        if (nxt > at) { gaps.off[gaps.n] = at; gaps.len[gaps.n] = nxt - at; ++gaps.n; }
        if (i < 11 && v[i].off + v[i].len > at) at = v[i].off + v[i].len;
      }
    }
    hipLaunchKernelGGL(window_q_chain_kernel, dim3((unsigned)cdiv(k * E, 256)), dim3(256), 0, st, S.q, Hbuf + N * E, BS / 4, (int)n_bags, k * E, cw, slab0, BS / 4,
                       gaps);
    MHIMX_LAUNCH_CHECK();
  }
  if (!update) {
    if (int r = mhimx_reduce_flush(stream, &lst_w)) return r;
    hipLaunchKernelGGL(window_sum_slabs_kernel, dim3((unsigned)(cdiv(c.n_train - g_lo, 256) < 2048 ? cdiv(c.n_train - g_lo, 256) : 2048)), dim3(256), 0, st,
                       c.g + g_lo, slab0 + g_lo, BS / 4, (int)n_bags, c.n_train - g_lo);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  // ---- 17. Adam + EMA teacher on g + the bags' slabs; the weight gradient's split-K slab sum is folded into the update
  mhimx_optim_args o = {};
  o.p = c.p; o.g = c.g; o.m = c.m; o.v = c.v; o.teacher = c.p_teacher; o.n_train = c.n_train; o.n_all = c.n_all; o.step = host_step; o.step_dev = c.opt_step;
  o.lr = c.lr; o.lr_table = c.lr_table; o.lr_len = c.lr_len; o.beta1 = c.beta1; o.beta2 = c.beta2; o.eps = c.eps; o.weight_decay = c.weight_decay;
  o.grad_scale = 1.f; o.ema_mm = c.ema_mm; o.mm_table = c.mm_table; o.mm_len = c.mm_len; o.zero_grad = 1; o.fold = &lst_w;
  o.g_extra = slab0; o.n_extra = n_bags; o.extra_pitch = BS / 4; o.extra_lo = g_lo; o.extra_only = 1;
  return mhimx_optim_step(stream, &o);
}

extern "C" int mhimx_step_project_ms(float* ms_out, float* empty_ms_out, int32_t cap) {
  MHIMX_CHECK_ARG(ms_out && cap >= 0, "step_project_ms: null output");
  int dev = 0;
  MHIMX_HIP(hipGetDevice(&dev));
  MHIMX_CHECK_ARG(dev >= 0 && dev < 64, "step_project_ms: device index %d", dev);
  const int n = g_proj_times.n[dev] < cap ? g_proj_times.n[dev] : cap;
  for (int i = 0; i < n; ++i) {
    MHIMX_HIP(hipEventSynchronize(g_proj_times.e[dev][i][2]));
    MHIMX_HIP(hipEventElapsedTime(ms_out + i, g_proj_times.e[dev][i][0], g_proj_times.e[dev][i][1]));
    if (empty_ms_out) MHIMX_HIP(hipEventElapsedTime(empty_ms_out + i, g_proj_times.e[dev][i][1], g_proj_times.e[dev][i][2]));
  }
  g_proj_times.n[dev] = 0;
  return n;
}

extern "C" int mhimx_step_run_many(void* stream, const mhimx_step_cfg* cfg, int32_t n_bags, const float* const* X, const int64_t* ldx, const int64_t* N,
                                   const int64_t* const* labels_dev, const mhimx_step_counts* cnt, const mhimx_step_seeds* seeds, int64_t host_step0,
                                   void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(n_bags >= 1 && X && ldx && N && labels_dev && cnt && seeds, "step_run_many: null arguments");
  for (int32_t i = 0; i < n_bags; ++i)
    if (int r = mhimx_step_run(stream, cfg, X[i], ldx[i], N[i], labels_dev[i], cnt + i, seeds + i, host_step0 + i, ws, ws_bytes, 1)) return r;
  return 0;
}
