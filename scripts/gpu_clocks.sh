#!/bin/bash
# what the clock and the power do under the MFMA loops and under dq_mb_kernel: rocm-smi sampled while the load runs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
sample() {   # label, then samples until the background job $1 ends
  local pid=$1 label=$2
  while kill -0 $pid 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed "s/^/$label: /; s/  */ /g"; echo
    sleep 0.3
  done
}
{
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed 's/^/idle: /; s/  */ /g'; echo
build_gpu/mfma_lab 3000000 0 > $O/clk_mfma_const.txt 2>&1 & sample $! "mfma constant operands"
build_gpu/mfma_lab 3000000 1 > $O/clk_mfma_rand.txt 2>&1 & sample $! "mfma random operands"
K2LAB_STEPS=15000 build_gpu/k2lab mb 28672 7168 256 2 bf16 "mb<2,4x2,4x4,nl4" > $O/clk_k2lab.txt 2>&1 & sample $! "dq_mb_kernel 28672x7168 bs256"
cat $O/clk_mfma_const.txt $O/clk_mfma_rand.txt $O/clk_k2lab.txt | grep -E "bf16|mb<|---"
} > $O/r05z_clocks_under_load.txt 2>&1
cut -c1-220 $O/r05z_clocks_under_load.txt
