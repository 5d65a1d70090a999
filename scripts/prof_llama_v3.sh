#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; TAG=${1:-x}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/run_llama_v3.py <<PY
import sys, json, torch
sys.path.insert(0, "$R/scripts"); sys.path.insert(0, "$R")
import decode_llama as L, decode_opt as D
dev, dtype = torch.device("cuda:0"), torch.float16
model = L.build(8, dev, dtype)
L.pack_model(model, 2, dev, twin=False)
model.v3 = True
model.fused_head = True
med, _, _ = D.time_decode(model, 1, 8, 48, 80, dev, dtype, False)
print(json.dumps({"layers": 8, "ms_per_token": med * 1e3}))
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ll3_$TAG -o trace -- python /tmp/run_llama_v3.py > $O/llama_v3_$TAG.json 2> $O/llama_v3_$TAG.err
echo "rc=$?"; cat $O/llama_v3_$TAG.json
cd $R; python scripts/rocpd_summary.py $O/prof_ll3_$TAG/trace_results.db | awk 'NR<=2 || $0 ~ /fused|attn_u|bigp|head_kernel|embed_kernel|u_only/' | head -24 | cut -c1-175 > $O/llama_v3_trace_$TAG.txt
cat $O/llama_v3_trace_$TAG.txt; rm -rf $O/prof_ll3_$TAG
