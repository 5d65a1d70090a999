#!/bin/bash
# round 4, call D: the blocked-operator decode kernels, w4 fused launches, fp4 operand pairing, gptq rows-per-workgroup A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out build_gpu
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ortho_blk.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py tests/test_gpu_decode_e2e.py tests/test_gpu_shard_rccl.py tests/test_gpu_gptq_qfnb.py -x -q -m gpu > gpurun_out/r04d_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04d_pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result scripts/fp4layout.hip -o build_gpu/fp4layout > /dev/null 2>&1
timeout 120 build_gpu/fp4layout 2>&1 | grep -v amdgpu.ids > gpurun_out/r04d_fp4layout.txt; tail -3 gpurun_out/r04d_fp4layout.txt
rm -f gpurun_out/r04d_decode_engine.jsonl
for cfg in "--arch opt --blocked" "--arch opt --bits 4" "--arch llama --blocked" "--arch opt --blocked --bits 4"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04d_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cut -c1-330 gpurun_out/r04d_decode_engine.jsonl
timeout 600 python scripts/bench_gptq_qfnb_rows.py > gpurun_out/r04d_gptq_qfnb_rows.jsonl 2>&1; cat gpurun_out/r04d_gptq_qfnb_rows.jsonl | cut -c1-600
