#!/bin/bash
# round 6, call B: the bench line and the GPU suite once more on another box (box-to-box spread of the final tree)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r06B_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r06B_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06B_bench.json 2> $O/r06B_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06B_bench.json"))
print("headline us", round(d["ms_per_step"]*1e3,3), "frac", d["roofline"]["frac"], "blocked", d["decode"]["value"], "kron", d["decode"]["kronecker_operators"]["value"], "llama", d["decode_llama"]["value"],
      "gptq_qfnb", d["quantise_linear"]["4096x4096"]["gptq_qfnb_ms"], "quantise_model", d["quantise_model"]["wall_s"])
PY
