#!/bin/bash
# round 4, call X: kernel traces of the one-launch-per-group decode step (package engine, full models) and the HBM traffic of K2's weight-stream leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for a in opt llama; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r04x_$a -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch $a --prompt 8 --tokens 56 > $O/r04x_prof_$a.log 2>&1; echo "prof $a rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_r04x_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/prof_k2_shape.py 28672 7168 16 20 > $O/r04x_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
for a in opt llama; do
  python scripts/rocpd_summary.py $O/prof_r04x_$a/trace_results.db > $O/r04x_decode_${a}_v3_head_kernel_trace.txt 2>&1
  grep -E "fused_gemm|fused_pair|attn|head_kernel|embed|bigp|u_only" $O/r04x_decode_${a}_v3_head_kernel_trace.txt | cut -c1-200
  rm -rf $O/prof_r04x_$a
done
python scripts/rocpd_summary.py $O/pmc_r04x_FETCH_SIZE/pmc_results.db $O/pmc_r04x_WRITE_SIZE/pmc_results.db > $O/r04x_k2_s_pmc.txt 2>&1
grep -E "dq_s_kernel" $O/r04x_k2_s_pmc.txt | cut -c1-220
rm -rf $O/pmc_r04x_FETCH_SIZE $O/pmc_r04x_WRITE_SIZE
