#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python $R/scripts/probe_scatter3.py --dense 2>&1 | grep scatter; }
run NSAMD_X=default
run NSAMD_SCATTER_COMBINE_RES=100
run NSAMD_SCATTER_TABLE_BITS=11
