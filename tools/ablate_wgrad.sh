#!/bin/bash
# time ablation variants of bag_wgrad (separate prebuilt libs mhim_mil_amd/libwg_*.so): tools/ablate_wgrad.sh (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in mhim_mil_amd/libmhimx.so mhim_mil_amd/libwg_*.so; do
  n=$(basename $f)
  echo "== $n $(MHIMX_LIB_NAME=$n python tools/exp_wgrad.py wgrad wgrad 2>&1 | grep wgrad | tail -1)"
done
