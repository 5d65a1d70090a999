#!/bin/bash
# rocprofv3 counter passes over single K2 lab variants (each pass: counters only + kernel trace, as gpurun requires)
# usage: k2pmc.sh <tag> ; results -> gpurun_out/k2pmc_<tag>.txt
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
    k = row['Kernel_Name'].split('(')[0][:70]
    acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in acc.items():
    if 'dq_' not in k and 'dqgemm' not in k: continue
    print(name, k)
    for cn, v in sorted(c.items()): print('    %-28s mean %14.1f  (n=%d)' % (cn, sum(v) / len(v), len(v)))
PY
}
{
S="s 28672 7168 16 2 bf16 nw7,ksp2,d3"
pass s_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $S
pass s_sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -- $S
pass s_sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT -- $S
pass s_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- $S
H="h 4096 4096 16 2 bf16 nw8,nch2,mixfalse"
pass h_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $H
M="mb 28672 7168 256 2 bf16 4x2,4x4"
pass m_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $M
pass m_sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -- $M
pass m_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- $M
} > $O/k2pmc_$1.txt 2>&1
mkdir -p $O/pmc_csv; for d in /tmp/pmc_*/; do n=$(basename $d); cp $d/p_counter_collection.csv $O/pmc_csv/$n.csv 2>/dev/null; done; du -sh $O/pmc_csv
