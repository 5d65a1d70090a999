#!/bin/bash
# round 3y: K8 kernel breakdown at d = 8192 / 4096 under rocprofv3; head kernel re-check
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_decode_head.py -q 2>&1 | tail -3
cat > /tmp/k8run.py <<PY
import sys, torch, time
sys.path.insert(0, "$R")
from quip_amd import ops
dev = "cuda:0"
d = int(sys.argv[1])
X = torch.randn(d + 256, d, device=dev); H = X.T @ X / d + 0.01 * torch.eye(d, device=dev)
ops.cholesky_lt(H); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): ops.cholesky_lt(H, check=False)
torch.cuda.synchronize(); print(d, "ms", (time.perf_counter() - t0) / 3 * 1e3)
PY
cd /tmp; export TMPDIR=/tmp
for d in 8192 4096; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k8_$d -o trace -- python /tmp/k8run.py $d 2>/dev/null | grep " ms "
  (cd $R; python scripts/rocpd_summary.py $O/prof_k8_$d/trace_results.db | grep -E "kernel|chol_|copyBuffer|fill" | cut -c1-170 > $O/k8_trace_$d.txt; cat $O/k8_trace_$d.txt); rm -rf $O/prof_k8_$d
done
cd $R; timeout 600 python scripts/decode_opt.py --only-chained --v3-only > $O/r3y_opt.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r3y_opt.json')); print({k: round(v['tok_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'tok_per_s' in v})"
