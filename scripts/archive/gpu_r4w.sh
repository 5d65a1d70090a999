#!/bin/bash
# round 4, call W: argmax_rows with 16-byte loads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_e2e.py -x -q -m gpu -k "argmax or e2e or end_to_end or decode" > gpurun_out/r04w_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/r04w_pytest.log | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from quip_amd import ops
for n, dt in ((50272, torch.float16), (32000, torch.float16), (50272, torch.float32)):
    x = torch.randn(1, n, device="cuda").to(dt)
    out = torch.empty(1, dtype=torch.int64, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.argmax_rows(x, out=out)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                ops.argmax_rows(x, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("argmax_rows", n, dt, round(e0.elapsed_time(e1) * 1e3 / 200, 2), "us per launch (back to back in a graph)")
PY
timeout 500 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:-1 2>/dev/null | grep '^{' | cut -c1-400
