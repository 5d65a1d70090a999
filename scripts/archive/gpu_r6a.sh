#!/bin/bash
# round 6, call a: (1) the GPU suite with the feedback-matrix stress IN SUITE ORDER (500 poisoned repetitions per ragged width);
# (2) fused-launch lab in four cache regimes; (3) in-situ phase stamps of a decode step, Kronecker + blocked; (4) kernel-trace timelines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
TAG=r06a
echo "== pytest gpu, QUIP_FEEDBACK_REPS=500"; QUIP_FEEDBACK_REPS=500 timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_$TAG.log
echo "== fusedlab"; bash scripts/fusedlab.sh $TAG 5 "0 1 2 3" > /dev/null 2>&1; grep -E "us per launch|regime" $O/fusedlab_$TAG.log | cut -c1-200
for v in "" "--blocked"; do
  n=kron; [ -n "$v" ] && n=blocked
  echo "== stamps $n"; timeout 600 python scripts/decode_stamps.py --arch opt $v > $O/decode_stamps_${n}_$TAG.txt 2> $O/decode_stamps_${n}_$TAG.err; echo "rc=$?"; head -3 $O/decode_stamps_${n}_$TAG.txt | cut -c1-250; tail -3 $O/decode_stamps_${n}_$TAG.err
  echo "== trace $n"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_dec_${n}_$TAG -o trace -- python $R/scripts/decode_engine_bench.py --arch opt $v --prompt 16 --tokens 64 > $O/decode_bench_${n}_$TAG.json 2> $O/decode_bench_${n}_$TAG.err); echo "rc=$?"; cat $O/decode_bench_${n}_$TAG.json | cut -c1-400
  db=$(ls $O/prof_dec_${n}_$TAG/*/*results.db $O/prof_dec_${n}_$TAG/*results.db 2>/dev/null | head -1)
  python scripts/decode_timeline.py $db --tokens 48 > $O/decode_timeline_${n}_$TAG.txt 2>&1; cat $O/decode_timeline_${n}_$TAG.txt | cut -c1-160
  rm -rf $O/prof_dec_${n}_$TAG
done
du -sh $O
