#!/bin/bash
# round 4, call L: batch sweeps of the decode engine (the fused launches take 1..4 sequences, the general launches up to 64), blocked and
# Kronecker, then the round-end sequence (scripts/gpu_round.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r04l_decode_batch.jsonl
timeout 500 python scripts/decode_engine_bench.py --arch opt --sweep 1:-1,4:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04l_decode_batch.jsonl; echo "opt kron rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:-1,2:-1,4:-1,8:-1 2>/dev/null | grep '^{' >> gpurun_out/r04l_decode_batch.jsonl; echo "opt blocked rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --sweep 8:-1,16:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04l_decode_batch.jsonl; echo "llama kron rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:-1,4:-1,8:-1 2>/dev/null | grep '^{' >> gpurun_out/r04l_decode_batch.jsonl; echo "llama blocked rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04l_decode_batch.jsonl"):
    d = json.loads(l)
    print({k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("arch", "operators", "engine_mode", "bs", "ms_per_step_median", "tok_per_s", "error") if d.get(k) is not None})
PY
bash scripts/gpu_round.sh r04Z
