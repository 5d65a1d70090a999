#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06h
echo "== pytest decode"; timeout 1800 python -m pytest -q -x -m gpu tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_head.py \
   tests/test_gpu_decode_bigp.py tests/test_gpu_checkpoint.py > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_$TAG.log | cut -c1-300
for rep in 1 2; do
  for lib in noshf default; do
    L=""; [ "$lib" = "noshf" ] && L=$R/quip_amd/csrc/libquip_amd_noshf.so
    for arch in opt llama; do
      echo "== decode $arch batch 1: library $lib (rep $rep)"
      QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch $arch --reps 2 2> $O/decode_ab_${arch}_${lib}_$TAG.err | grep '"operand_prefetch": false' | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/decode_ab_shf_$TAG.jsonl
    done
  done
done
for lib in noshf default; do
  L=""; [ "$lib" = "noshf" ] && L=$R/quip_amd/csrc/libquip_amd_noshf.so
  for b in 4 16; do
    echo "== decode opt $b sequences: library $lib"
    QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch opt --bs $b --reps 2 2>> $O/decode_ab_bs_$TAG.err | grep '"operand_prefetch": false' | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/decode_ab_shf_$TAG.jsonl
  done
  echo "== decode llama 16 sequences: library $lib"
  QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch llama --bs 16 --reps 2 2>> $O/decode_ab_bs_$TAG.err | grep '"operand_prefetch": false' | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/decode_ab_shf_$TAG.jsonl
done
echo "== stamps llama (shared fragments)"; timeout 600 python scripts/decode_stamps.py --arch llama > $O/decode_stamps_llama_$TAG.txt 2> $O/decode_stamps_llama_$TAG.err; grep -E "^\[|U: row landed|reduce" $O/decode_stamps_llama_$TAG.txt | cut -c1-150
echo "== stamps opt"; timeout 600 python scripts/decode_stamps.py --arch opt > $O/decode_stamps_kron_$TAG.txt 2> $O/decode_stamps_kron_$TAG.err; grep -E "^\[|U: row landed|reduce" $O/decode_stamps_kron_$TAG.txt | cut -c1-150
