#!/bin/bash
# round 6, call k: blocked stage with the rows in registers, DPP sums, SOLO finish (ortho_blk.hip).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== pytest blk + decode"; timeout 1500 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_step.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py -q -x 2>&1 | tail -15
echo "== blocked OPT-1.3B, blk_fused_n 2048 / 8192"
timeout 900 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:2048,1:8192,1:2048,1:8192,2:2048,4:2048,4:8192 2>/dev/null | tee -a $O/r06k_decode_blocked.jsonl | cut -c1-330
echo "== blocked Llama-2-7B, blk_fused_n 2048 / 4096 / 16384"
timeout 900 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:2048,1:4096,1:16384,1:2048 2>/dev/null | tee -a $O/r06k_decode_blocked.jsonl | cut -c1-330
echo "== in-situ stamps, blocked"
timeout 600 python scripts/decode_stamps.py --blocked > $O/r06k_decode_stamps_blocked.txt 2>&1; grep -E "^\[|span" $O/r06k_decode_stamps_blocked.txt | head -40
