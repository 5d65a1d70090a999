#!/bin/bash
# Runs on the GPU box via gpurun: gpu tests, smoke, bench, rocprofv3 kernel trace + PMC passes. Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
TAG=${1:-r1}
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_$TAG.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke_$TAG.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json; tail -3 $O/bench_$TAG.err
cd /tmp
echo "== rocprof kernel-trace (cold regime, eager launches)"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --steps 1000 --warmup 100 --profile-cold-only --eager --no-spin --preheat-ms 0 > $O/prof_bench_$TAG.log 2>&1; echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprof pmc $c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$TAG -o pmc -- python $R/bench.py --steps 300 --warmup 30 --profile-cold-only --eager --no-spin --preheat-ms 0 > $O/pmc_${c}_$TAG.log 2>&1; echo "rc=$?"
done
echo "== rocprof pmc: MFMA utilisation of the MFMA-bound legs (bs 256 weight stream, prefill)"
for shape in "28672 7168 256" "4096 4096 2048"; do
  n=$(echo $shape | tr ' ' 'x')
  timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/pmc_mfma_${n}_$TAG -o pmc -- python $R/scripts/prof_k2_shape.py $shape 10 > $O/pmc_mfma_${n}_$TAG.log 2>&1; echo "rc=$?"
done
cd $R
python scripts/pmc_mfma_summary.py "28672x7168 bs256" $O/pmc_mfma_28672x7168x256_$TAG/pmc_results.db > $O/mfma_util_$TAG.txt 2>&1
python scripts/pmc_mfma_summary.py "4096x4096 bs2048 (prefill)" $O/pmc_mfma_4096x4096x2048_$TAG/pmc_results.db >> $O/mfma_util_$TAG.txt 2>&1
cat $O/mfma_util_$TAG.txt | cut -c1-200
python scripts/rocpd_summary.py $O/prof_$TAG/trace_results.db $O/pmc_FETCH_SIZE_$TAG/pmc_results.db $O/pmc_WRITE_SIZE_$TAG/pmc_results.db > $O/rocprof_summary_$TAG.txt 2>&1
python scripts/rocpd_summary.py --k2-json $O/pmc_FETCH_SIZE_$TAG/pmc_results.db $O/pmc_WRITE_SIZE_$TAG/pmc_results.db $O/k2_pmc_$TAG.json
grep -E "dqgemm" $O/rocprof_summary_$TAG.txt | cut -c1-200
du -sh $O
