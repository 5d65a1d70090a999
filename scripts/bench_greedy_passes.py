#!/usr/bin/env python3
"""round_ldl with 0 and 9 greedy passes at full size: time and proxy loss (greedy passes)."""
import time, torch, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_amd import ops, vector_balance as VB
dev="cuda:0"
for (m,d) in [(4096,4096),(2048,8192)]:
    X=torch.randn(d+256,d,device=dev); H=X.T@X/(d+256); H+=0.01*H.diag().mean()*torch.eye(d,device=dev)
    w=(torch.rand(m,d,device=dev)*3.6-0.3).clamp(0,3)
    def run(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); c=VB.round_ldl(w,H,2,n_greedy_passes=n); torch.cuda.synchronize(); return time.perf_counter()-t0, c
    run(0); run(2)
    t0,c0=run(0); t9,c9=run(9)
    def proxy(c):
        dw=c.double()-w.double(); return float(((dw@H.double())*dw).sum())
    print(json.dumps({"shape":f"{m}x{d}","round_ldl_npasses0_ms":round(t0*1e3,2),"round_ldl_npasses9_ms":round(t9*1e3,2),"proxy0":proxy(c0),"proxy9":proxy(c9)}),flush=True)
