#!/bin/bash
# round 4, call K: the single-launch blocked operator with its prologue rewritten (16-byte loads, gather from LDS), A/B per size threshold;
# the Kronecker engine at 1 / 2 / 4 sequences side by side
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ortho_blk.py -x -q -m gpu > gpurun_out/r04k_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04k_pytest.log
rm -f gpurun_out/r04k_decode_engine.jsonl
timeout 700 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:0,1:4096,1:16384,1:0,1:4096 2>/dev/null | grep '^{' >> gpurun_out/r04k_decode_engine.jsonl; echo "opt blocked rc=$?"
timeout 700 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:0,1:4096,1:16384,1:0,1:4096 2>/dev/null | grep '^{' >> gpurun_out/r04k_decode_engine.jsonl; echo "llama blocked rc=$?"
timeout 600 python scripts/decode_engine_bench.py --arch opt --sweep 1:-1,2:-1,4:-1 2>/dev/null | grep '^{' >> gpurun_out/r04k_decode_engine.jsonl; echo "opt bs rc=$?"
timeout 600 python scripts/decode_engine_bench.py --arch llama --sweep 1:-1,2:-1,4:-1 2>/dev/null | grep '^{' >> gpurun_out/r04k_decode_engine.jsonl; echo "llama bs rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04k_decode_engine.jsonl"):
    d = json.loads(l)
    print({k: d.get(k) for k in ("arch", "operators", "engine_mode", "bs", "blk_fused_n", "ms_per_step_median", "tok_per_s", "error") if d.get(k) is not None})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04k_opt -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch opt --blocked --layers 4 --prompt 8 --tokens 24 --blk-fused-n 16384 > $GRAFT_REPO_ROOT/gpurun_out/r04k_prof_opt.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04k_llama -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch llama --blocked --layers 4 --prompt 8 --tokens 24 --blk-fused-n 16384 > $GRAFT_REPO_ROOT/gpurun_out/r04k_prof_llama.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/rocpd_summary.py gpurun_out/prof_r04k_opt/trace_results.db > gpurun_out/r04k_decode_opt_blocked_kernel_trace.txt 2>&1; grep -E "blk_stage" gpurun_out/r04k_decode_opt_blocked_kernel_trace.txt | cut -c1-220
python scripts/rocpd_summary.py gpurun_out/prof_r04k_llama/trace_results.db > gpurun_out/r04k_decode_llama_blocked_kernel_trace.txt 2>&1; grep -E "blk_stage" gpurun_out/r04k_decode_llama_blocked_kernel_trace.txt | cut -c1-220
rm -rf gpurun_out/prof_r04k_opt gpurun_out/prof_r04k_llama
