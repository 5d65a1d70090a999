#!/bin/bash
# round 6, call m: fp16 x~ in the blocked decode path (grouped one-pass GEMM for q / k / v), half-slab one-pass kernel for d <= 8192 (fc2).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== pytest dqgemm + blk + decode"; timeout 2400 python -m pytest tests/test_gpu_dqgemm_v2.py tests/test_gpu_dqgemm.py tests/test_gpu_ortho_blk.py tests/test_gpu_decode_step.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py -q -x 2>&1 | tail -15
echo "== blocked OPT-1.3B"
timeout 900 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:2048,1:2048,2:2048,4:2048,8:2048 2>/dev/null | tee -a $O/r06m_decode_blocked.jsonl | cut -c1-330
echo "== blocked Llama-2-7B"
timeout 900 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:2048,1:2048 2>/dev/null | tee -a $O/r06m_decode_blocked.jsonl | cut -c1-330
echo "== Kronecker OPT-1.3B / Llama (unchanged paths: control)"
timeout 900 python scripts/decode_engine_bench.py --arch opt --sweep 1:2048,4:2048,8:2048 2>/dev/null | tee -a $O/r06m_decode_kron.jsonl | cut -c1-330
echo "== trace blocked"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_dec_blocked_r06m -o trace -- python $R/scripts/decode_engine_bench.py --arch opt --blocked --prompt 16 --tokens 64 > $O/decode_bench_blocked_r06m.json 2> $O/decode_bench_blocked_r06m.err); echo "rc=$?"
db=$(ls $O/prof_dec_blocked_r06m/*/*results.db $O/prof_dec_blocked_r06m/*results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/decode_timeline.py $db --tokens 48 --head argmax_rows > $O/r06m_decode_timeline_blocked.txt 2>&1; cat $O/r06m_decode_timeline_blocked.txt | cut -c1-170
rm -rf $O/prof_dec_blocked_r06m
