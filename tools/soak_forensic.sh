#!/bin/bash
# forensic soak: merge_bwd passes next to a c5 bench that loads the GPU; args: R, passes, library names (default: the product library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
R=${1:-970}; P=${2:-30000}; shift 2
LIBS=${@:-libmhimx.so}
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 200 python bench.py --workload c5 --cpu-steps 0 --steps 1500 --warmup 5 > /tmp/load.log 2>&1; done ) &
LOAD=$!
sleep 20
for lib in $LIBS; do
  MHIMX_LIB_NAME=$lib timeout 600 python tools/exp_merge_forensic.py $R $P gpurun_out > gpurun_out/forensic_${R}_$lib.log 2>&1
  echo "== $lib: $(grep -c EVENT gpurun_out/forensic_${R}_$lib.log) events; $(tail -1 gpurun_out/forensic_${R}_$lib.log)"
  grep EVENT gpurun_out/forensic_${R}_$lib.log | head -4 | cut -c1-400
done
for c in $(pgrep -P $LOAD 2>/dev/null); do kill $c 2>/dev/null; done
kill $LOAD 2>/dev/null
