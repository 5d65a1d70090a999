#!/bin/bash
# round 3G: whole-model runs (OPT-1.3B architecture, 24 blocks, 128 x 2048 calibration tokens): LDLQ and GPTQ, both with incoherence processing
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for q in ldlq gptq; do
  timeout 600 python scripts/quantize_opt.py --hidden 2048 --ffn 8192 --heads 32 --layers 24 --vocab 50272 --nsamples 128 --seqlen 2048 --wbits 2 --quant $q --incoh > $O/r3G_full_$q.json 2> $O/r3G_full_$q.err
  echo "$q rc=$?"; python -c "
import json; d=json.loads(open('$O/r3G_full_$q.json').read().strip().splitlines()[-1]); print('$q', 'wall_s', d['wall_s'], 'linears', d['linears'], 'mean_proxy_error', round(d['mean_proxy_error'],2), 'sum fasterquant s', round(sum(p['seconds'] for p in d['per_linear']),2))"
done
