#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ll_$TAG -o trace -- python $R/scripts/decode_llama.py --layers 8 --prompt 8 --tokens 64 --no-dense > $O/decode_llama_$TAG.json 2> $O/decode_llama_$TAG.err
echo "rc=$?"
cd $R; python scripts/rocpd_summary.py $O/prof_ll_$TAG/trace_results.db | awk 'NR<=2 || $0 ~ /ortho|dq|decode_attn|rope|Cijk|elementwise|silu|mul/' | cut -c1-170 | head -30 > $O/decode_llama_trace_$TAG.txt
cat $O/decode_llama_trace_$TAG.txt; rm -rf $O/prof_ll_$TAG
