#!/usr/bin/env python3
"""rocprofv3 --pmc results (rocpd sqlite) of scripts/prof_k2_shape.py -> per K2 kernel: counters averaged per dispatch and the MFMA
utilisation they imply:  util = SQ_INSTS_MFMA x cycles per instruction / (kernel duration x 2.4 GHz x 1024 SIMDs)
(v_mfma_f32_16x16x32_bf16: 16 cycles = 4 passes at the 2.5 PFLOP/s peak; ROCm 7.2 has no gfx950 formulas for the derived MfmaUtil).
usage: pmc_mfma_summary.py <label> <results.db> [...]"""
import sqlite3
import sys

label = sys.argv[1]
for p in sys.argv[2:]:
    con = sqlite3.connect(p)
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by 1, 2").fetchall()
    byk = {}
    for k, c, n, v, dur in rows:
        if "dq_" in k or "dqgemm" in k:
            byk.setdefault(k, {})[c] = (n, v, dur)
    for k, cs in byk.items():
        print(f"{label}  {k[:100]}")
        for c, (n, v, dur) in sorted(cs.items()):
            print(f"    {c:<28} {v:>16.1f}   (dispatches {n}, avg kernel {dur / 1e3:.2f} us under the counter pass)")
        if "SQ_INSTS_MFMA" in cs:
            n, v, dur = cs["SQ_INSTS_MFMA"]
            print(f"    => MFMA utilisation {100.0 * v * 16 / (dur * 2.4 * 1024):.1f} % of the bf16 matrix pipe (16 cycles per 16x16x32 instruction, 1024 SIMDs, 2.4 GHz)")
