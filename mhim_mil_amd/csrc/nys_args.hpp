// nys_args.hpp - the kernel argument block shared by the two translation units of the streamed Nystrom attention
// (nys_flash.hip: landmark-column kernels, 8 waves; nys_flash_tok.hip: token-column kernels, 4 waves).
#pragma once
#include "mma_tile.hpp"

namespace mhimx {

#define NY_EXP2(x) __builtin_amdgcn_exp2f(x)

constexpr int NY_H = 8, NY_D = 64, NY_M = 256, NY_TT = 64, NY_MAXCH = 32, NY_TOKCH = 32;     // token chunks per head: landmark-column kernels (= workspace bound) / token-owning kernels
constexpr int NY_IMG = 16384;                        // one fragment image of a 64 x 64 tile (hi + lo)
constexpr int NY_PART = NY_M * NY_D;                 // floats of one [256, 64] partial

struct NyArgs {
  const float* q; const float* k; const float* v;
  int64_t ld, T;
  const float* ql; const float* kl; int64_t ldl;
  float scale, sl2e;                                 // sl2e = scale * log2(e): P = exp2(s * sl2e - lse2)
  int nch;
  // per entry point
  const float* w2; const float* dout; int64_t ldd;
  const float* lse1; float* lse1_o; float* delta; const float* delta_i;
  const float* lse3; const float* delta3; const float* da3v; const float* u;
  float* out; int64_t ldo;
  float* out2; int64_t ldo2;
  float* part; float* part2;
  int accumulate;
};


// token-column kernels (nys_flash_tok.hip): enqueue only
int nytok_out_fwd(hipStream_t st, const NyArgs& g);
int nytok_out_bwd_q(hipStream_t st, const NyArgs& g);
int nytok_a3v_bwd_t(hipStream_t st, const NyArgs& g, int mode);

}  // namespace mhimx
