#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_shard_rccl.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
timeout 600 python scripts/quantize_opt_sharded.py --hidden 2048 --ffn 8192 --heads 32 --layers 2 --nsamples 64 --seqlen 2048 --vocab 4096 --incoh --force-exchange 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('world','calibration','wall_s','phase_seconds_rank0','bytes_broadcast_weights','mean_proxy_error')})"
timeout 600 python scripts/quantize_opt_sharded.py --hidden 2048 --ffn 8192 --heads 32 --layers 2 --nsamples 64 --seqlen 2048 --vocab 4096 --incoh --force-exchange --calibration owner 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('world','calibration','wall_s','phase_seconds_rank0','mean_proxy_error')})"
