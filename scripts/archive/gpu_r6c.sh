#!/bin/bash
# round 6, call c: the decode changes behind the register fix (csrc/prefetch.h), correctness first, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06c
echo "== pytest (changed areas)"; timeout 1800 python -m pytest -q -x -m gpu tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py \
   tests/test_gpu_decode_e2e.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_gptq.py tests/test_gpu_shard_rccl.py tests/test_gpu_ortho_ldlq.py tests/test_gpu_feedback_stress.py tests/test_gpu_checkpoint.py \
   tests/test_gpu_decode_bigp.py tests/test_gpu_decode_head.py > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_$TAG.log | cut -c1-300
for rep in 1 2; do
  for lib in nokpf default; do
    echo "== decode A/B OPT-1.3B Kronecker batch 1: library $lib (rep $rep)"
    L=""; [ "$lib" = "nokpf" ] && L=$R/quip_amd/csrc/libquip_amd_nokpf.so
    QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch opt --reps 2 2> $O/decode_ab_opt_${lib}_$TAG.err | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/decode_ab_opt_$TAG.jsonl
  done
done
for lib in nokpf default; do
  echo "== decode A/B Llama-2-7B batch 1: library $lib"
  L=""; [ "$lib" = "nokpf" ] && L=$R/quip_amd/csrc/libquip_amd_nokpf.so
  QUIP_AMD_LIB=$L timeout 600 python scripts/bench_decode_ab.py --arch llama --reps 2 2> $O/decode_ab_llama_${lib}_$TAG.err | sed "s/^{/{\"lib\": \"$lib\", /" | tee -a $O/decode_ab_llama_$TAG.jsonl
done
for v in "" "--blocked"; do
  n=kron; hd=head_kernel; [ -n "$v" ] && n=blocked && hd=argmax_rows
  echo "== stamps $n"; timeout 600 python scripts/decode_stamps.py --arch opt $v > $O/decode_stamps_${n}_$TAG.txt 2> $O/decode_stamps_${n}_$TAG.err; echo "rc=$?"; head -2 $O/decode_stamps_${n}_$TAG.txt | cut -c1-250; tail -2 $O/decode_stamps_${n}_$TAG.err
  echo "== trace $n"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_dec_${n}_$TAG -o trace -- python $R/scripts/decode_engine_bench.py --arch opt $v --prompt 16 --tokens 64 > $O/decode_bench_${n}_$TAG.json 2> $O/decode_bench_${n}_$TAG.err); echo "rc=$?"
  db=$(ls $O/prof_dec_${n}_$TAG/*/*results.db $O/prof_dec_${n}_$TAG/*results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/decode_timeline.py $db --tokens 48 --head $hd --dump $O/decode_timeline_${n}_$TAG.npz > $O/decode_timeline_${n}_$TAG.txt 2>&1; cat $O/decode_timeline_${n}_$TAG.txt | cut -c1-170
  rm -rf $O/prof_dec_${n}_$TAG
done
echo "== stamps llama"; timeout 600 python scripts/decode_stamps.py --arch llama > $O/decode_stamps_llama_$TAG.txt 2> $O/decode_stamps_llama_$TAG.err; echo "rc=$?"; head -2 $O/decode_stamps_llama_$TAG.txt | cut -c1-250
du -sh $O
