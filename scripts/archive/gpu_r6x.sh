#!/bin/bash
# round 6, call x: the fused Kronecker launches with their per-thread operand loads unconditional (variant unc) against the default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in default unc; do
  L=""; [ "$lib" = "unc" ] && L=$R/quip_amd/csrc/libquip_amd_unc.so
  for arch in opt llama; do
    QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch $arch --reps 1 2>/dev/null | grep '"operand_prefetch": false' | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/r06x_decode_ab_unc.jsonl | cut -c1-200
  done
done
done
