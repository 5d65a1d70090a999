#!/bin/bash
# round 4, call E: multi-operator blocked launches, fp4 pairing matrix, kernel trace of the blocked decode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out build_gpu
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_hf.py -x -q -m gpu > gpurun_out/r04e_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04e_pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result scripts/fp4layout.hip -o build_gpu/fp4layout > /dev/null 2>&1
timeout 120 build_gpu/fp4layout 2>&1 | grep -v amdgpu.ids > gpurun_out/r04e_fp4layout.txt; head -12 gpurun_out/r04e_fp4layout.txt; tail -3 gpurun_out/r04e_fp4layout.txt
rm -f gpurun_out/r04e_decode_engine.jsonl
for cfg in "--arch opt --blocked" "--arch llama --blocked"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04e_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cut -c1-330 gpurun_out/r04e_decode_engine.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04e_opt -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch opt --blocked --layers 4 --prompt 8 --tokens 24 > $GRAFT_REPO_ROOT/gpurun_out/r04e_prof_opt.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04e_llama -o trace -- python $GRAFT_REPO_ROOT/scripts/decode_engine_bench.py --arch llama --blocked --layers 4 --prompt 8 --tokens 24 > $GRAFT_REPO_ROOT/gpurun_out/r04e_prof_llama.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/rocpd_summary.py gpurun_out/prof_r04e_opt/trace_results.db > gpurun_out/r04e_decode_opt_blocked_kernel_trace.txt 2>&1; head -30 gpurun_out/r04e_decode_opt_blocked_kernel_trace.txt | cut -c1-220
python scripts/rocpd_summary.py gpurun_out/prof_r04e_llama/trace_results.db > gpurun_out/r04e_decode_llama_blocked_kernel_trace.txt 2>&1; head -30 gpurun_out/r04e_decode_llama_blocked_kernel_trace.txt | cut -c1-220
rm -rf gpurun_out/prof_r04e_opt gpurun_out/prof_r04e_llama
