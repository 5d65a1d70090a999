#!/bin/bash
# round 4, call V: smoke() with the fp32 gate; kernel traces of the blocked decode step on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04v_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04v_smoke.log
cd /tmp
for a in opt llama; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04v_$a -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch $a --blocked --layers 4 --prompt 8 --tokens 24 > $GRAFT_REPO_ROOT/gpurun_out/r04v_prof_$a.log 2>&1; echo "prof $a rc=$?"
done
cd $GRAFT_REPO_ROOT
for a in opt llama; do
  python scripts/rocpd_summary.py gpurun_out/prof_r04v_$a/trace_results.db > gpurun_out/r04v_decode_${a}_blocked_kernel_trace.txt 2>&1
  grep -E "blk_stage|dqgemm|dq_h|decode_attn|argmax|Cijk_Alik_Bljk_HHS" gpurun_out/r04v_decode_${a}_blocked_kernel_trace.txt | cut -c1-200
  rm -rf gpurun_out/prof_r04v_$a
done
