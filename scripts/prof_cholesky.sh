#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/scripts/bench_cholesky.py 2>&1 | tail -5
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_chol -o trace -- python $R/scripts/bench_cholesky.py > /dev/null 2>&1
cd $R; python scripts/rocpd_summary.py $O/prof_chol/trace_results.db | awk 'NR<=2 || $0 ~ /chol/' | cut -c1-170 | head -12; rm -rf $O/prof_chol
