#!/bin/bash
# GPU box: kernel trace of the packed decode loop (chained + V-fused variants), summary -> gpurun_out/decode_trace_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_dec_$TAG -o trace -- python $R/scripts/decode_opt.py --only-chained --layers 24 --prompt 8 --tokens 96 > $O/decode_$TAG.json 2> $O/decode_$TAG.err
echo "rc=$?"; cat $O/decode_$TAG.json
cd $R; python scripts/rocpd_summary.py $O/prof_dec_$TAG/trace_results.db | awk 'NR<=2 || $0 ~ /ortho|dq|decode_attn|Cijk|elementwise|layer_norm|argmax|reduce/' | cut -c1-170 | head -40 > $O/decode_trace_$TAG.txt
cat $O/decode_trace_$TAG.txt; rm -rf $O/prof_dec_$TAG
