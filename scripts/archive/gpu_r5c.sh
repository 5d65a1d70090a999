#!/bin/bash
# round 5, call c: grouped h kernel + multi-row bigp: tests, decode sweeps for OPT and Llama, kernel trace of one bs-16 step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_e2e.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_decode_step.py -x -q -s > $O/r05c_pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|engine|Error|assert" $O/r05c_pytest.log | tail -15
echo "== decode batch sweep opt"; timeout 600 python scripts/decode_engine_bench.py --arch opt --prompt 32 --tokens 32 --sweep 1:-1,4:-1,8:-1,16:-1,32:-1 > $O/r05c_decode_batch_opt.jsonl 2> $O/r05c_decode_batch_opt.err; echo "rc=$?"; python -c "
import json,sys
for l in open('$O/r05c_decode_batch_opt.jsonl'):
    r=json.loads(l); print(r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"
echo "== decode batch sweep llama"; timeout 900 python scripts/decode_engine_bench.py --arch llama --prompt 32 --tokens 32 --sweep 1:-1,4:-1,8:-1,16:-1 > $O/r05c_decode_batch_llama.jsonl 2> $O/r05c_decode_batch_llama.err; echo "rc=$?"; python -c "
import json,sys
for l in open('$O/r05c_decode_batch_llama.jsonl'):
    r=json.loads(l); print(r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"
cd /tmp; export TMPDIR=/tmp
echo "== kernel trace opt bs16"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/r05c_prof_opt16 -o trace -- python $R/scripts/decode_engine_bench.py --arch opt --layers 4 --prompt 8 --tokens 24 --bs 16 > $O/r05c_prof_opt16.log 2>&1; echo "rc=$?"
cd $R; python scripts/rocpd_summary.py $O/r05c_prof_opt16/trace_results.db > $O/r05c_decode_opt_bs16_kernel_trace.txt 2>&1; head -30 $O/r05c_decode_opt_bs16_kernel_trace.txt | cut -c1-220
rm -rf $O/r05c_prof_opt16
