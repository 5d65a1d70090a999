#!/bin/bash
# round 4, call S: the round-end sequence on the final tree + the decode batch sweeps of the four models
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_round.sh r04ZZ
rm -f gpurun_out/r04s_decode_batch.jsonl
timeout 500 python scripts/decode_engine_bench.py --arch opt --sweep 1:-1,2:-1,4:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04s_decode_batch.jsonl; echo "opt kron rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch opt --blocked --sweep 1:-1,2:-1,4:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04s_decode_batch.jsonl; echo "opt blocked rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --sweep 1:-1,2:-1,4:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04s_decode_batch.jsonl; echo "llama kron rc=$?"
timeout 500 python scripts/decode_engine_bench.py --arch llama --blocked --sweep 1:-1,2:-1,4:-1,8:-1,16:-1,32:-1,64:-1 2>/dev/null | grep '^{' >> gpurun_out/r04s_decode_batch.jsonl; echo "llama blocked rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04s_decode_batch.jsonl"):
    d = json.loads(l)
    print(d.get("arch", "")[:10], d.get("operators", "")[:8], d.get("engine_mode"), d.get("bs"), round(d.get("ms_per_step_median", 0), 3), round(d.get("tok_per_s", 0)), d.get("error", ""))
PY
