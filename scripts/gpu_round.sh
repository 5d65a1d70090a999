#!/bin/bash
# Runs on the GPU box via gpurun: gpu tests, bench, rocprofv3 kernel trace + PMC passes. Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
TAG=${1:-r1}
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_$TAG.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke_$TAG.log
echo "== bench"; timeout 600 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
cd /tmp
echo "== rocprof kernel-trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --eager > $O/prof_bench_$TAG.log 2>&1; echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprof pmc $c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$TAG -o pmc -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --eager > $O/pmc_${c}_$TAG.log 2>&1; echo "rc=$?"
done
find $O -name '*.csv' | head -30
du -sh $O
