cd $GRAFT_REPO_ROOT
python tools/_dbg_win.py 2>&1 | tail -8
echo ---- no fuse
MHIMX_FUSE_DPRE=0 python tools/_dbg_win.py 2>&1 | tail -8
