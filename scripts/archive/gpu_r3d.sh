#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
bash scripts/fusedlab.sh r03d 5 > /dev/null 2>&1; cat $O/fusedlab_r03d.log
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_decode_step.py tests/test_gpu_decode_hf.py -m gpu -q -rP -x -p no:cacheprovider > $O/pytest_gpu_r03d.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu_r03d.log | tail -3; grep -E "decode vs HF" $O/pytest_gpu_r03d.log; grep -E "^E  " $O/pytest_gpu_r03d.log | head -20
timeout 600 python scripts/decode_opt.py --only-chained --v3-only --layers 24 --prompt 8 --tokens 96 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['tok_per_s'],1) if isinstance(v,dict) and 'tok_per_s' in v else '') for k,v in d.items()})"
