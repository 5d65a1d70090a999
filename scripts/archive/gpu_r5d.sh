#!/bin/bash
# round 5, call d: x-ingest floor lab, S configs for the short-wide fc2 shape, Llama bs-16 trace, checkpoint test, per-dispatch durations of the headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== xingest lab"; timeout 120 build_gpu/xingest_lab 2>&1 | grep -v amdgpu.ids > $O/r05d_xingest_lab.txt; echo "rc=$?"; cat $O/r05d_xingest_lab.txt
echo "== S cfgs 2048x8192 f16"; timeout 300 python scripts/bench_k2_s_cfgs.py --shapes 2048x8192,4096x11008 --dtype f16 > $O/r05d_k2_s_shortwide.jsonl 2>&1; echo "rc=$?"; cut -c1-200 $O/r05d_k2_s_shortwide.jsonl
echo "== checkpoint test"; timeout 1200 python -m pytest tests/test_gpu_checkpoint.py -x -q -s > $O/r05d_pytest_checkpoint.log 2>&1; echo "rc=$?"; grep -E "passed|failed|checkpoint|Error" $O/r05d_pytest_checkpoint.log | tail
cd /tmp; export TMPDIR=/tmp
echo "== kernel trace llama bs16"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/r05d_prof_l16 -o trace -- python $R/scripts/decode_engine_bench.py --arch llama --layers 4 --prompt 8 --tokens 24 --bs 16 > $O/r05d_prof_l16.log 2>&1; echo "rc=$?"
cd $R; python scripts/rocpd_summary.py $O/r05d_prof_l16/trace_results.db > $O/r05d_decode_llama_bs16_kernel_trace.txt 2>&1; grep -E "anonymous|Cijk" $O/r05d_decode_llama_bs16_kernel_trace.txt | head -14 | cut -c1-200
rm -rf $O/r05d_prof_l16
cd /tmp
echo "== headline per-dispatch durations"; timeout 600 rocprofv3 --kernel-trace -d $O/r05d_prof_h -o trace -- python $R/bench.py --steps 1000 --warmup 100 --profile-cold-only --eager --no-spin > $O/r05d_prof_h.log 2>&1; echo "rc=$?"
cd $R; python scripts/k2h_duration_seq.py $O/r05d_prof_h/trace_results.db 96 > $O/r05d_k2h_duration_seq.txt 2>&1; cat $O/r05d_k2h_duration_seq.txt
cd /tmp; timeout 600 rocprofv3 --kernel-trace -d $O/r05d_prof_hs -o trace -- python $R/bench.py --steps 200 --warmup 20 --profile-cold-only --eager > $O/r05d_prof_hs.log 2>&1; echo "rc=$?"
cd $R; python scripts/k2h_duration_seq.py $O/r05d_prof_hs/trace_results.db 96 > $O/r05d_k2h_duration_seq_spin.txt 2>&1; head -12 $O/r05d_k2h_duration_seq_spin.txt
rm -rf $O/r05d_prof_h $O/r05d_prof_hs
