#!/bin/bash
# round 6, call i: the blocked-stage prologue rework (ortho_blk.hip: rows + permutation first, one factor chunk per thread at n <= 2048, fast
# divisions, residual as loaded).  Parity, blocked decode tok/s at three fusion bounds, in-situ stamps.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== pytest blk + decode"; timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_step.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_fused.py -q -x 2>&1 | tail -4
echo "== blocked OPT-1.3B, blk_fused_n 2048 / 4096 / 8192, alternating"
timeout 900 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:2048,1:8192,1:2048,1:8192,1:2048,4:2048,4:8192 2>/dev/null | tee -a $O/r06i_decode_blocked.jsonl | cut -c1-330
echo "== blocked Llama-2-7B, blk_fused_n 2048 / 4096 / 16384"
timeout 900 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:2048,1:4096,1:16384,1:2048 2>/dev/null | tee -a $O/r06i_decode_blocked.jsonl | cut -c1-330
echo "== in-situ stamps, blocked"
timeout 600 python scripts/decode_stamps.py --blocked > $O/r06i_decode_stamps_blocked.txt 2>&1; grep -E "^\[|span" $O/r06i_decode_stamps_blocked.txt | head -40
