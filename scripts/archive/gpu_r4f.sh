#!/bin/bash
# round 4, call F: gptq_qfnb with 64 rows per workgroup (tests vs the column walk + A/B), fp4 lab with the measured layouts, blocked decode again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out build_gpu
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gptq_qfnb.py tests/test_gpu_gptq.py tests/test_gpu_ortho_blk.py tests/test_gpu_decode_e2e.py tests/test_gpu_driver.py -x -q -m gpu > gpurun_out/r04f_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04f_pytest.log
timeout 600 python scripts/bench_gptq_qfnb_rows.py > gpurun_out/r04f_gptq_qfnb_rows.jsonl 2>&1; grep shape gpurun_out/r04f_gptq_qfnb_rows.jsonl | cut -c1-640
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -I include -I quip_amd/csrc scripts/fp4lab.hip -o build_gpu/fp4lab > /dev/null 2>&1
timeout 300 build_gpu/fp4lab 2>&1 | grep -v amdgpu.ids > gpurun_out/r04f_fp4lab.txt; head -3 gpurun_out/r04f_fp4lab.txt
rm -f gpurun_out/r04f_decode_engine.jsonl
for cfg in "--arch opt --blocked" "--arch llama --blocked"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04f_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cut -c1-330 gpurun_out/r04f_decode_engine.jsonl
