#!/bin/bash
cd $GRAFT_REPO_ROOT
GV_SELECT=1 GV_MODES=ring GV_PATHS=0 ITERS=20 bash tools/prof_any.sh r4_sel_ring tools/bench_msda_gv.py 2>&1 | head -8
