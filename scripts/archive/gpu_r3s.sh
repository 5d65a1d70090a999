#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_shard_rccl.py -m gpu -q -x -p no:cacheprovider > $O/shard_r03s.log 2>&1; tail -25 $O/shard_r03s.log | cut -c1-250
timeout 600 python scripts/quantize_opt_sharded.py --hidden 2048 --ffn 8192 --heads 32 --layers 2 --nsamples 64 --seqlen 2048 --vocab 4096 --incoh --force-exchange > $O/sharded_run_r03s.json 2> $O/sharded_run_r03s.err; tail -5 $O/sharded_run_r03s.err | cut -c1-300; cut -c1-900 $O/sharded_run_r03s.json
