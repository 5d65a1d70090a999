#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py -m gpu -q -rP -x -p no:cacheprovider > $O/pytest_gpu_r03k.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_r03k.log | tail -3; grep -E "decode vs HF" $O/pytest_gpu_r03k.log; grep -E "^E  " $O/pytest_gpu_r03k.log | head -20
timeout 600 python scripts/decode_llama.py --layers 32 --prompt 16 --tokens 48 --no-dense 2>$O/llama_r03k.err | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['tok_per_s'],1) if isinstance(v,dict) and 'tok_per_s' in v else '') for k,v in d.items()})"; tail -3 $O/llama_r03k.err
timeout 600 python scripts/decode_opt.py --only-chained --v3-only --layers 24 --prompt 8 --tokens 96 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['tok_per_s'],1) if isinstance(v,dict) and 'tok_per_s' in v else '') for k,v in d.items()})"
