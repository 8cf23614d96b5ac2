#!/bin/bash
# usage (GPU box, repo root): tools/pmc_nys.sh  -> gpurun_out/pmc_nys.md
# Two rocprofv3 --pmc passes (kernel trace only) over tools/exp_nys.py: issue / wait / MFMA counters and LDS counters of the streamed
# Nystrom attention kernels at T = 50 176.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
REP=3 MIN_US=40 HEAD=200 $ROOT/tools/pmc_any.sh nysa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" $ROOT/tools/exp_nys.py > /dev/null
REP=3 MIN_US=40 HEAD=200 $ROOT/tools/pmc_any.sh nysb "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $ROOT/tools/exp_nys.py > /dev/null
{
  echo "# rocprofv3 --pmc: the streamed Nystrom attention kernels at T = 50 176 tokens (tools/exp_nys.py, 8 heads x 256 landmarks) - round 3"
  echo
  echo "\`tools/pmc_nys.sh\`: two passes, \`--kernel-trace\` only.  Per dispatch the counter instances are summed, averages over the dispatches of a kernel;"
  echo "\"per SIMD-cycle\" = value / (GRBM_GUI_ACTIVE per instance x 1024 SIMDs).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (x 4 for"
  echo "cycles: SQ_WAVE_CYCLES 0.40 = 1.6 resident waves per SIMD), SQ_VALU_MFMA_BUSY_CYCLES counts cycles (its per-SIMD-cycle figure IS the matrix-pipe busy fraction)."
  echo
  echo '## pass 1: issue, wait, matrix pipe'
  echo '```'
  grep -v "at::native" -A0 $ROOT/gpurun_out/pmc_nysa.txt | grep -v "^--"
  echo '```'
  echo
  echo '## pass 2: LDS'
  echo '```'
  grep -v "at::native" $ROOT/gpurun_out/pmc_nysb.txt
  echo '```'
} > $ROOT/gpurun_out/pmc_nys.md
head -30 $ROOT/gpurun_out/pmc_nys.md
