#!/bin/bash
# time ablation variants of bag_project (separate prebuilt libs): tools/ablate_proj.sh (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in mhim_mil_amd/libabl_*.so; do
  n=$(basename $f)
  echo "== $n"; MHIMX_LIB_NAME=$n python tools/exp_proj.py 2>&1 | grep "bag_project p"
done
