#!/bin/bash
# round 3v: decode_bigp tests after the wave-count fix, HF decode parity, the 32-block Llama-2-7B-architecture decode
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_decode_bigp.py tests/test_gpu_decode_fused.py tests/test_gpu_decode_hf.py tests/test_gpu_decode.py -q > $O/r3v_tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/r3v_tests.log
timeout 600 python scripts/decode_llama.py --check > $O/r3v_llama_check.json 2> $O/r3v_llama_check.err; cat $O/r3v_llama_check.json
timeout 900 python scripts/decode_llama.py --no-dense > $O/r3v_llama.json 2> $O/r3v_llama.err; echo "rc=$?"; python - <<PY
import json
d = json.load(open("$O/r3v_llama.json"))
print({k: (round(v["tok_per_s"], 1) if isinstance(v, dict) and "tok_per_s" in v else None) for k, v in d.items()})
PY
