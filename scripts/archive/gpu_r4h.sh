#!/bin/bash
# round 4, call H: 4-bit bigp tail, kernarg pinning in ortho_blk
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_bigp.py tests/test_gpu_decode_step.py tests/test_gpu_ortho_blk.py tests/test_gpu_decode_e2e.py -x -q -m gpu > gpurun_out/r04h_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04h_pytest.log
rm -f gpurun_out/r04h_decode_engine.jsonl
for cfg in "--arch opt --blocked" "--arch llama --blocked" "--arch llama --bits 4" "--arch opt --bits 4"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04h_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cut -c1-330 gpurun_out/r04h_decode_engine.jsonl
