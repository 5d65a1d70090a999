#!/bin/bash
# round 6, call v: does where the runtime keeps kernel arguments move the launch gap?  (HIP_FORCE_DEV_KERNARG 0 / 1, decode tok/s)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 0 1 unset; do
  for mode in "" "--blocked"; do
    echo "== HIP_FORCE_DEV_KERNARG=$v $mode"
    if [ "$v" = "unset" ]; then env -u HIP_FORCE_DEV_KERNARG timeout 600 python scripts/decode_engine_bench.py --arch opt $mode 2>/dev/null | tail -1 | sed "s/^{/{\"HIP_FORCE_DEV_KERNARG\": \"$v\", /" | tee -a $O/r06v_kernarg.jsonl | cut -c1-60,280-420
    else HIP_FORCE_DEV_KERNARG=$v timeout 600 python scripts/decode_engine_bench.py --arch opt $mode 2>/dev/null | tail -1 | sed "s/^{/{\"HIP_FORCE_DEV_KERNARG\": \"$v\", /" | tee -a $O/r06v_kernarg.jsonl | cut -c1-60,280-420; fi
  done
done
env | grep -i "HIP_\|HSA_\|AMD_\|GPU_" | head -20
