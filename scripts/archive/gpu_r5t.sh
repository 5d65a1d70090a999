#!/bin/bash
# round 5: the two-launch threshold by layer size under test; fresh kernel traces of the many-sequence decode steps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_bigp.py -q -x > $O/r05t_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05t_pytest.log
cd /tmp
for spec in "llama 16" "llama 1" "opt 16" "opt 4"; do
  set -- $spec
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/r05t_prof_$1_$2 -o trace -- python $R/scripts/decode_engine_bench.py --arch $1 --layers 4 --prompt 8 --tokens 24 --bs $2 > $O/r05t_prof_$1_$2.log 2>&1; echo "$spec rc=$?"
  python $R/scripts/rocpd_summary.py $O/r05t_prof_$1_$2/trace_results.db > $O/r05t_decode_$1_bs$2_kernel_trace.txt 2>&1
  rm -rf $O/r05t_prof_$1_$2
done
