#!/bin/bash
# round 4, call N: csrc/ortho_blk.hip up to 64 rows (16 per workgroup, groups side by side): blocked models at 16 .. 64 sequences per step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_step.py -x -q -m gpu > gpurun_out/r04n_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/r04n_pytest.log | tail -3
rm -f gpurun_out/r04n_decode_batch.jsonl
timeout 500 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04n_decode_batch.jsonl; echo "opt blocked rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04n_decode_batch.jsonl; echo "llama blocked rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04n_decode_batch.jsonl"):
    d = json.loads(l)
    print({k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("arch", "operators", "engine_mode", "bs", "ms_per_step_median", "tok_per_s", "error") if d.get(k) is not None})
PY
