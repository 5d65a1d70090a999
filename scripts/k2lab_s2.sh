#!/bin/bash
O=${GRAFT_REPO_ROOT:-.}/gpurun_out; mkdir -p $O
L=${GRAFT_REPO_ROOT:-.}/build_gpu/k2lab
run() { echo "### $*"; timeout 150 $L "$@" 2>&1 | grep -v amdgpu.ids; echo "rc=$?"; }
{
run s 28672 7168 16 2 bf16
run s 28672 7168 1 2 bf16
run s 28672 7168 16 4 bf16
run s 32768 8192 16 2 bf16
} > $O/k2lab_$1.log 2>&1
tail -5 $O/k2lab_$1.log
