#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_r03m.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_r03m.log | tail -3; grep -E "^E  |^FAILED" $O/pytest_gpu_r03m.log | head -20
timeout 600 python scripts/decode_llama.py --layers 32 --prompt 16 --tokens 48 --no-dense 2>$O/llama_r03m.err | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['tok_per_s'],1) if isinstance(v,dict) and 'tok_per_s' in v else '') for k,v in d.items()})"; tail -2 $O/llama_r03m.err
