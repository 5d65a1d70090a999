#!/bin/bash
# the rocprofv3 passes of gpu_round.sh alone (kernel trace + the two PMC passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-p}
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --steps 1000 --warmup 100 --profile-cold-only --eager --no-spin > $O/prof_bench_$TAG.log 2>&1; echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$TAG -o pmc -- python $R/bench.py --steps 300 --warmup 30 --profile-cold-only --eager --no-spin > $O/pmc_${c}_$TAG.log 2>&1; echo "rc=$?"
done
cd $R
python scripts/rocpd_summary.py $O/prof_$TAG/trace_results.db $O/pmc_FETCH_SIZE_$TAG/pmc_results.db $O/pmc_WRITE_SIZE_$TAG/pmc_results.db > $O/rocprof_summary_$TAG.txt 2>&1
python scripts/rocpd_summary.py --k2-json $O/pmc_FETCH_SIZE_$TAG/pmc_results.db $O/pmc_WRITE_SIZE_$TAG/pmc_results.db $O/k2_pmc_$TAG.json
grep -E "dq_h" $O/rocprof_summary_$TAG.txt | cut -c1-200
rm -rf $O/prof_$TAG $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG
