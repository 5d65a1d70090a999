#!/bin/bash
# round 6, call b: correctness of the round's kernel changes, then the A/B runs they were made for
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06b
echo "== pytest (changed areas)"; timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py \
   tests/test_gpu_decode_e2e.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_gptq.py tests/test_gpu_shard_rccl.py tests/test_gpu_ortho_ldlq.py tests/test_gpu_feedback_stress.py tests/test_gpu_checkpoint.py \
   > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_$TAG.log
echo "== decode A/B: operand prefetch (OPT-1.3B Kronecker, batch 1)"; timeout 900 python scripts/bench_decode_ab.py --arch opt --reps 3 > $O/decode_ab_opt_$TAG.jsonl 2> $O/decode_ab_opt_$TAG.err; echo "rc=$?"; cat $O/decode_ab_opt_$TAG.jsonl; tail -2 $O/decode_ab_opt_$TAG.err
echo "== fusedlab"; bash scripts/fusedlab.sh $TAG 5 "1 3" > /dev/null 2>&1; grep -E "us per launch|regime" $O/fusedlab_$TAG.log | cut -c1-200
for v in "" "--blocked"; do
  n=kron; hd=head_kernel; [ -n "$v" ] && n=blocked && hd=argmax_rows
  echo "== stamps $n"; timeout 600 python scripts/decode_stamps.py --arch opt $v > $O/decode_stamps_${n}_$TAG.txt 2> $O/decode_stamps_${n}_$TAG.err; echo "rc=$?"; head -2 $O/decode_stamps_${n}_$TAG.txt | cut -c1-250; tail -2 $O/decode_stamps_${n}_$TAG.err
  echo "== trace $n"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_dec_${n}_$TAG -o trace -- python $R/scripts/decode_engine_bench.py --arch opt $v --prompt 16 --tokens 64 > $O/decode_bench_${n}_$TAG.json 2> $O/decode_bench_${n}_$TAG.err); echo "rc=$?"
  db=$(ls $O/prof_dec_${n}_$TAG/*/*results.db $O/prof_dec_${n}_$TAG/*results.db 2>/dev/null | head -1)
  python scripts/decode_timeline.py $db --tokens 48 --head $hd --dump $O/decode_timeline_${n}_$TAG.npz > $O/decode_timeline_${n}_$TAG.txt 2>&1; cat $O/decode_timeline_${n}_$TAG.txt | cut -c1-170
  rm -rf $O/prof_dec_${n}_$TAG
done
echo "== grouped forms (Llama-2-7B, 8 / 16 sequences)"; timeout 900 python scripts/bench_grouped_forms.py > $O/grouped_forms_$TAG.jsonl 2> $O/grouped_forms_$TAG.err; echo "rc=$?"; cat $O/grouped_forms_$TAG.jsonl; tail -2 $O/grouped_forms_$TAG.err
du -sh $O
