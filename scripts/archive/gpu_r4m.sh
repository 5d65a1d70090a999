#!/bin/bash
# round 4, call M: p x 16 operators on csrc/ortho_bigp.hip up to 64 rows (the Llama cliff of r04l beyond 8 sequences); the general launch
# sequence ("fused" mode) against the one-launch-per-group one (v3_head) at 2 / 4 sequences
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ortho_tile.py tests/test_gpu_decode_step.py -x -q -m gpu > gpurun_out/r04m_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r04m_pytest.log | tail -2
rm -f gpurun_out/r04m_decode_batch.jsonl
timeout 500 python scripts/decode_engine_bench.py --arch llama --sweep 8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04m_decode_batch.jsonl; echo "llama kron rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --mode fused --sweep 1:-1,2:-1,4:-1 2>/dev/null | grep '^{' >> gpurun_out/r04m_decode_batch.jsonl; echo "llama fused rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch opt --mode fused --sweep 1:-1,2:-1,4:-1 2>/dev/null | grep '^{' >> gpurun_out/r04m_decode_batch.jsonl; echo "opt fused rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 16:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04m_decode_batch.jsonl; echo "llama blocked rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04m_decode_batch.jsonl"):
    d = json.loads(l)
    print({k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("arch", "operators", "engine_mode", "bs", "ms_per_step_median", "tok_per_s", "error") if d.get(k) is not None})
PY
