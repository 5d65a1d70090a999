#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python scripts/decode_engine_bench.py --arch opt --sweep 16:2048,1:2048 2>/dev/null | cut -c1-30,250-360
