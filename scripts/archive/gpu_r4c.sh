#!/bin/bash
# round 4, call C: fp4 lab, blocked / w4 decode through the engine, then the whole gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out build_gpu
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -I include -I quip_amd/csrc scripts/fp4lab.hip -o build_gpu/fp4lab > gpurun_out/r04c_fp4lab_build.log 2>&1
timeout 300 build_gpu/fp4lab 2>&1 | grep -v amdgpu.ids > gpurun_out/r04c_fp4lab.txt; echo "fp4lab rc=$?"; cat gpurun_out/r04c_fp4lab.txt
for cfg in "--arch opt" "--arch opt --blocked" "--arch opt --bits 4" "--arch llama" "--arch llama --blocked"; do
  timeout 600 python scripts/decode_engine_bench.py $cfg 2>/dev/null | tail -1 >> gpurun_out/r04c_decode_engine.jsonl; echo "decode $cfg rc=$?"
done
cat gpurun_out/r04c_decode_engine.jsonl | cut -c1-420
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04c_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04c_pytest_gpu.log
