#!/bin/bash
# round 6, call F: three-heads attention form with / without the early K rows at head dim 64 (variant mhnokpf), OPT-1.3B 16 / 8 sequences
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
for lib in default mhnokpf; do
  L=""; [ "$lib" = "mhnokpf" ] && L=$R/quip_amd/csrc/libquip_amd_mhnokpf.so
  QUIP_AMD_LIB=$L timeout 600 python scripts/decode_engine_bench.py --arch opt --sweep 16:2048,12:2048 2>/dev/null | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/r06F_mh_kpf.jsonl | cut -c1-30,250-360
done
done
