#!/bin/bash
# round 4, call O: where a 64-sequence step spends its time (kernel traces, 4 layers)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for a in opt llama; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04o_$a -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch $a --layers 4 --prompt 8 --tokens 24 --bs 64 > $GRAFT_REPO_ROOT/gpurun_out/r04o_prof_$a.log 2>&1; echo "prof $a rc=$?"
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r04o_prof_$a.log | cut -c1-300
done
cd $GRAFT_REPO_ROOT
for a in opt llama; do
  python scripts/rocpd_summary.py gpurun_out/prof_r04o_$a/trace_results.db > gpurun_out/r04o_decode_${a}_bs64_kernel_trace.txt 2>&1
  head -16 gpurun_out/r04o_decode_${a}_bs64_kernel_trace.txt | cut -c1-200
  rm -rf gpurun_out/prof_r04o_$a
done
