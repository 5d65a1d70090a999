#!/bin/bash
# K8 A/B on one box: round-1 form, branch-free pipelined trailing update, + two-stream look-ahead
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_cholesky_sched.py tests/test_gpu_ortho_ldlq.py -q 2>&1 | grep -E 'passed|failed|FAILED|^E  ' | head
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/r3z2_k8_ab.txt
import sys, time, torch
sys.path.insert(0, "$R")
from quip_amd import ops
dev = "cuda:0"
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for d in (2048, 4096, 8192, 11008, 16384):
    X = torch.randn(d + 256, d, device=dev); H = X.T @ X / d + 0.01 * torch.eye(d, device=dev)
    row = []
    for name, kw in (("round1", dict(old_syrk=True, lookahead=False, unblocked_diag=True)), ("full-syrk", dict(lookahead=False, unblocked_diag=True)),
                     ("+blocked-diag", dict(lookahead=False)), ("+lookahead", dict(lookahead=True)), ("default", dict())):
        ops.cholesky_config(**kw)
        row.append("%s %.2f" % (name, t(lambda: ops.cholesky_lt(H, check=False))))
    ops.cholesky_config()
    print(d, " | ".join(row), "| TF(last) %.1f" % (d ** 3 / 3 / 1e9 / float(row[-1].split()[-1])), flush=True)
PY
