#!/bin/bash
# round 4, call T: ortho_blk.hip without the zero fill of the unused MFMA columns (it doubled when a workgroup went from 8 to 16 rows)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_step.py -x -q -m gpu > gpurun_out/r04t_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error" gpurun_out/r04t_pytest.log | tail -3
rm -f gpurun_out/r04t_decode_batch.jsonl
timeout 500 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:-1,2:-1,4:-1,8:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04t_decode_batch.jsonl; echo "opt blocked rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:-1,2:-1,4:-1,8:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04t_decode_batch.jsonl; echo "llama blocked rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04t_decode_batch.jsonl"):
    d = json.loads(l)
    print(d.get("arch", "")[:10], d.get("operators", "")[:8], d.get("engine_mode"), d.get("bs"), round(d.get("ms_per_step_median", 0), 3), round(d.get("tok_per_s", 0)), d.get("error", ""))
PY
