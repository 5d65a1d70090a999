#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06f
echo "== pytest (changed areas)"; timeout 1800 python -m pytest -q -x -m gpu tests/test_gpu_decode_fused.py tests/test_gpu_decode_attn.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py \
   tests/test_gpu_decode_e2e.py tests/test_gpu_ortho_blk.py tests/test_gpu_cholesky_sched.py tests/test_gpu_ortho_ldlq.py tests/test_gpu_feedback_stress.py tests/test_gpu_opt30b.py tests/test_gpu_method.py \
   > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_$TAG.log | cut -c1-300
for rep in 1 2; do
  echo "== decode A/B OPT-1.3B Kronecker batch 1 (rep $rep)"
  timeout 600 python scripts/bench_decode_ab.py --arch opt --reps 2 2> $O/decode_ab_opt_$TAG.err | tee -a $O/decode_ab_opt_$TAG.jsonl
done
echo "== decode A/B Llama-2-7B batch 1"; timeout 600 python scripts/bench_decode_ab.py --arch llama --reps 2 2> $O/decode_ab_llama_$TAG.err | tee -a $O/decode_ab_llama_$TAG.jsonl
echo "== blocked OPT-1.3B"; timeout 600 python scripts/bench_decode_ab.py --arch opt --blocked --reps 3 2> $O/decode_ab_blocked_$TAG.err | tee -a $O/decode_ab_blocked_$TAG.jsonl
echo "== OPT-1.3B 4 / 16 sequences"; for b in 4 16; do timeout 600 python scripts/bench_decode_ab.py --arch opt --bs $b --reps 2 2>> $O/decode_ab_opt_$TAG.err | tee -a $O/decode_ab_opt_bs_$TAG.jsonl; done
n=kron; hd=head_kernel
echo "== trace $n"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_dec_${n}_$TAG -o trace -- python $R/scripts/decode_engine_bench.py --arch opt --prompt 16 --tokens 64 > $O/decode_bench_${n}_$TAG.json 2> $O/decode_bench_${n}_$TAG.err); echo "rc=$?"
db=$(ls $O/prof_dec_${n}_$TAG/*/*results.db $O/prof_dec_${n}_$TAG/*results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/decode_timeline.py $db --tokens 48 --head $hd > $O/decode_timeline_${n}_$TAG.txt 2>&1; cat $O/decode_timeline_${n}_$TAG.txt | cut -c1-170
rm -rf $O/prof_dec_${n}_$TAG
du -sh $O
