#!/bin/bash
# round 3z: K8 with the branch-free trailing update + look-ahead: parity (bit-identical to the round-1 form), timings, kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_cholesky_sched.py tests/test_gpu_ortho_ldlq.py -q 2>&1 | tail -5
timeout 600 python scripts/bench_cholesky.py 2>&1 | tee $O/r3z_bench_cholesky.txt
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
for d in 8192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k8_$d -o trace -- python /tmp/k8run.py $d 2>/dev/null | grep " ms "
  (cd $R; python scripts/rocpd_summary.py $O/prof_k8_$d/trace_results.db | grep -E "kernel|chol_" | cut -c1-170 > $O/k8_trace_r3z_$d.txt; cat $O/k8_trace_r3z_$d.txt); rm -rf $O/prof_k8_$d
done
