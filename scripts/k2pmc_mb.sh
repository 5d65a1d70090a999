#!/bin/bash
# rocprofv3 counter passes over single K2 lab variants (each pass: counters only + kernel trace, as gpurun requires)
# usage: k2pmc_mb.sh <tag> ; results -> gpurun_out/k2pmc_<tag>.txt   (LDS counters of dq_mb_kernel, both MFMA shapes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export K2LAB_STEPS=10
pass() {  # name counters... -- lab args
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- $R/build_gpu/k2lab "$@" > /tmp/pmc_$name.log 2>&1
  echo "== $name rc=$? : $(grep -v simple_timer /tmp/pmc_$name.log | grep -v amdgpu.ids | tail -4 | tr '\n' '|' | cut -c1-600)"
  find /tmp/pmc_$name -type f | head -5
  python3 - "$name" /tmp/pmc_$name <<'PY'
import sys, csv, glob, collections
name, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not f: print(name, 'NO COUNTER FILE'); sys.exit()
print(name, f[0], flush=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'].replace('(anonymous namespace)::', '')[:90]
    acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in acc.items():
    if 'dq_' not in k and 'dqgemm' not in k: continue
    print(name, k)
    for cn, v in sorted(c.items()): print('    %-28s mean %14.1f  (n=%d)' % (cn, sum(v) / len(v), len(v)))
PY
}
{
M16="mb 28672 7168 256 2 bf16 mb<2,4x2,4x4,nl4"
M32="mb 28672 7168 256 2 bf16 mb32<2,4x2,4x4,nl4"
pass mb16_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $M16
pass mb32_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $M32
} > $O/k2pmc_$1.txt 2>&1
grep -v "^/tmp" $O/k2pmc_$1.txt | cut -c1-200
