#!/usr/bin/env python3
"""GPTQ.fasterquant through the K4 kernel path vs the reference-order column loop (GPTQ on K4)."""
import time, torch, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import gptq as G, quant as Q
dev = "cuda:0"
for (m, d) in [(2048, 2048), (2048, 8192)]:
    X = torch.randn(d + 256, d, device=dev); H = (X.T @ X / (d + 256))
    W = (0.02 * torch.randn(m, d, device=dev)).half()
    res = {}
    for use in (True, False):
        G.USE_KERNEL = use
        for rep in range(2):
            lin = torch.nn.Linear(d, m, bias=False).to(dev).half(); lin.weight.data = W.clone()
            meth = G.GPTQ(lin); meth.quantizer = Q.Quantizer(); meth.quantizer.configure(4, perchannel=True, sym=False, qfn='a', mse=False)
            meth.H = H.clone(); meth.preproc(preproc_gptqH=True, percdamp=.01)
            torch.cuda.synchronize(); t0 = time.perf_counter(); meth.fasterquant(); torch.cuda.synchronize()
            res["kernel" if use else "column_loop"] = round(time.perf_counter() - t0, 4)
            err = meth.error
        res["error_" + ("kernel" if use else "loop")] = err
    print(json.dumps({"shape": f"{m}x{d}", **res}), flush=True)
