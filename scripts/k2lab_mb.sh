#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out; mkdir -p $O
L=${GRAFT_REPO_ROOT:-.}/build_gpu/k2lab
run() { echo "### $*"; timeout 150 $L "$@" 2>&1 | grep -v amdgpu.ids; echo "rc=$?"; }
{
run mb 28672 7168 256 2 bf16
run mb 4096 4096 2048 2 bf16
run mb 8192 8192 256 2 bf16
run mb 28672 7168 256 4 bf16
run mb 28672 7168 64 2 bf16
} > $O/k2lab_$1.log 2>&1
tail -5 $O/k2lab_$1.log
