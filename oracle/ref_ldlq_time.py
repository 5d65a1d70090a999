"""oracle/ref_ldlq_time.py -- time the REFERENCE's own CPU LDLQ path on this machine's host cores.

TEST / MEASUREMENT INFRASTRUCTURE (bench.py's `ldlq_cpu_reference` leg runs it in a subprocess; nothing else does).
Imports the reference's quant.py / method.py / vector_balance.py / bal.py from the byte copies oracle/stage_ref.py stages into
the git-ignored oracle/_ref/cpu/ (plus the primefac shim tests/golden/_shims/primefac.py: method.py:8 imports a package that is
not installed) and runs, unmodified,

    Balance(layer).configure('ldlq', 2, 0, unbiased=False); quantizer qfn b            (bal.py:15-19, quant.py:138-163)
    .H = X^T X / (d + 256);  .preproc(gptqH, percdamp .01, rescale, proj, extra 0)     (method.py:125-193)
    .fasterquant(lazy_batch in {False, True})                                          (bal.py:21-48 -> vector_balance.py:155-199 / 218-291)

on BASELINE.md section 2's synthetic layer (W = 0.02 randn(m, d) fp16, X = randn(d + 256, d), seeds 0), printing one JSON line per
measurement: the `.time` attribute fasterquant sets (grid map + rounding + postproc, bal.py:27,47) and preproc's wall time.

usage: python oracle/ref_ldlq_time.py [--sizes 2048,4096] [--budget 120]"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2048,4096")
    ap.add_argument("--budget", type=float, default=120.0, help="stop starting new measurements after this many seconds")
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads (0: torch's default = the box's physical cores)")
    ap.add_argument("--order", default="as-given", choices=["as-given", "lazy-first"])
    a = ap.parse_args()
    cpu = os.path.join(HERE, "_ref", "cpu")
    if not os.path.exists(os.path.join(cpu, "bal.py")):
        print(json.dumps({"error": "oracle/_ref/cpu not staged (oracle/stage_ref.py needs the reference checkout)"}))
        return 1
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden", "_shims"))
    sys.path.insert(0, cpu)
    import numpy as np
    import torch
    import torch.nn as nn
    if a.threads:
        torch.set_num_threads(a.threads)
    import bal as ref_bal                  # the reference's files
    import quant as ref_quant
    assert os.path.dirname(os.path.abspath(ref_bal.__file__)) == cpu and os.path.dirname(os.path.abspath(ref_quant.__file__)) == cpu
    t_start = time.perf_counter()
    cases = [(int(v), lazy) for v in a.sizes.split(",") for lazy in (False, True)]
    if a.order == "lazy-first":                                    # the cheaper half of every size first (a time budget then cuts the dearer one)
        cases = sorted(cases, key=lambda c: (c[0], not c[1]))
    for d, lazy in cases:
        m = d
        if True:
            if time.perf_counter() - t_start > a.budget:
                print(json.dumps({"m": m, "d": d, "lazy_batch": lazy, "skipped": "time budget"}), flush=True)
                continue
            torch.manual_seed(0)
            np.random.seed(0)
            lin = nn.Linear(d, m, bias=False)
            lin.weight.data = (0.02 * torch.randn(m, d)).half()
            X = torch.randn(d + 256, d)
            meth = ref_bal.Balance(lin)
            meth.configure('ldlq', 2, 0, unbiased=False)
            meth.quantizer = ref_quant.Quantizer()
            meth.quantizer.configure(2, perchannel=True, sym=False, qfn='b', mse=False)
            meth.H = (X.T @ X / (d + 256)).to(torch.float32)
            meth.nsamples = 1
            del X
            t0 = time.perf_counter()
            meth.preproc(preproc_gptqH=True, percdamp=.01, preproc_rescale=True, preproc_proj=True, preproc_proj_extra=0)
            t_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            meth.fasterquant(lazy_batch=lazy)
            t_fq = time.perf_counter() - t0
            print(json.dumps({"m": m, "d": d, "lazy_batch": lazy, "fasterquant_time_attr_s": round(float(meth.time), 3),
                              "fasterquant_wall_s": round(t_fq, 3), "preproc_wall_s": round(t_pre, 3), "proxy_error": float(meth.error),
                              "threads": torch.get_num_threads(), "logical_cpus": os.cpu_count()}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
