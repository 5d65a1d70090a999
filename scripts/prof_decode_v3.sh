#!/bin/bash
# GPU box: kernel trace of the v3 decode loop only -> gpurun_out/decode_v3_trace_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_v3_$TAG -o trace -- python $R/scripts/decode_opt.py --only-chained --v3-only --layers 24 --prompt 8 --tokens 96 > $O/decode_v3_$TAG.json 2> $O/decode_v3_$TAG.err
echo "rc=$?"; python -c "import json,sys; d=json.load(open('$O/decode_v3_$TAG.json')); print({k:(round(v['tok_per_s'],1) if isinstance(v,dict) and 'tok_per_s' in v else '') for k,v in d.items()})"
cd $R; python scripts/rocpd_summary.py $O/prof_v3_$TAG/trace_results.db | awk 'NR<=2 || $0 ~ /fused|attn_u|head_kernel|embed_kernel|u_only|Cijk_Alik_Bljk_HHS|layer_norm|argmax|ArgMax/' | cut -c1-175 | head -20 > $O/decode_v3_trace_$TAG.txt
cat $O/decode_v3_trace_$TAG.txt; rm -rf $O/prof_v3_$TAG
