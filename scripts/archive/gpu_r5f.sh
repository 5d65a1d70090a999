#!/bin/bash
# round 5, call f: chunk order x fine-grained last-chunk waits on the headline kernel (8 lab builds, alternating, 3 rounds), bigp tail with the reduce launch,
# dqgemm + bigp tests, decode sweeps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2 3; do
  for v in c0f0 c1f0 c2f0 c3f0 c0f1 c1f1 c2f1 c3f1; do echo -n "$v rep $rep: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2" ; done
done
for v in c0f0 c1f1 c2f1 c3f1; do echo -n "$v f16: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 2 f16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
for v in c0f0 c1f1 c2f1 c3f1; do echo -n "$v bs8: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 8 2 bf16 "h<2,rt1,nw8,nch2>" 2>&1 | grep "h<2"; done
for v in c0f0 c1f1 c2f1 c3f1; do echo -n "$v w4: "; timeout 120 build_gpu/k2lab_$v h 4096 4096 16 4 bf16 "h<4,rt1,nw8,nch4>" 2>&1 | grep "h<4"; done
for v in c0f0 c1f1 c2f1 c3f1; do echo "$v 2048:"; timeout 120 build_gpu/k2lab_$v h 2048 2048 16 2 bf16 2>&1 | grep "h<2"; done
} > $O/r05f_k2lab_chunks_fine_ab.txt 2>&1
cat $O/r05f_k2lab_chunks_fine_ab.txt | cut -c1-150
echo "== bigp tail"; timeout 300 python scripts/bench_bigp_tail.py > $O/r05f_bigp_tail.jsonl 2>&1; cat $O/r05f_bigp_tail.jsonl | cut -c1-300
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_dqgemm.py tests/test_gpu_dqgemm_v2.py tests/test_gpu_decode_bigp.py tests/test_gpu_decode_e2e.py tests/test_gpu_decode_hf.py tests/test_gpu_checkpoint.py -x -q > $O/r05f_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/r05f_pytest.log
echo "== decode sweeps"; for arch in opt llama; do timeout 900 python scripts/decode_engine_bench.py --arch $arch --prompt 32 --tokens 32 --sweep 1:-1,8:-1,16:-1 > $O/r05f_decode_batch_$arch.jsonl 2> $O/r05f_decode_batch_$arch.err; python -c "
import json,sys
for l in open('$O/r05f_decode_batch_$arch.jsonl'):
    r=json.loads(l); print('$arch', r.get('bs'), r.get('engine_mode'), round(r.get('ms_per_step_median',0),3), round(r.get('tok_per_s',0)), r.get('error'))"; done
