#!/bin/bash
# round 3w: csrc/decode_head.hip -- tests, HF parity with the fused ends, OPT-1.3B and Llama-2-7B decode
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_decode_head.py tests/test_gpu_decode_hf.py -q > $O/r3w_tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/r3w_tests.log
timeout 900 python scripts/decode_opt.py --only-chained --v3-only > $O/r3w_opt.json 2> $O/r3w_opt.err; echo "opt rc=$?"; tail -3 $O/r3w_opt.err
python - <<PY
import json
for f in ("r3w_opt.json",):
    try:
        d = json.load(open("$O/" + f))
        print({k: (round(v["tok_per_s"], 1), v.get("logits_rel_diff_vs_v3", v.get("logits_rel_diff_vs_chained"))) for k, v in d.items() if isinstance(v, dict) and "tok_per_s" in v})
    except Exception as e:
        print(f, e)
PY
timeout 900 python scripts/decode_llama.py --no-dense > $O/r3w_llama.json 2> $O/r3w_llama.err; echo "llama rc=$?"; tail -3 $O/r3w_llama.err
python - <<PY
import json
d = json.load(open("$O/r3w_llama.json"))
print({k: round(v["tok_per_s"], 1) for k, v in d.items() if isinstance(v, dict) and "tok_per_s" in v})
PY
