#!/usr/bin/env python3
"""In-situ phase stamps of the HEADLINE launch (dq_h_kernel, 4 waves x 4 adjacent chunks) on the probe build of the library (csrc/probe.h):
every launch streams a different cold weight copy out of a ring (> Infinity Cache), runs alone, and leaves slot i of its four waves in the
probe buffer.  VERDICT r5 next #3: "re-stamp the CURRENT 4 x 4 kernel -- the meet figure is from round-2 code", and "why is 8192 x 2048 (half
the x) not faster cold".  Medians over the launches, clocks since the first wave's first stamp: first wave .. last wave."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QUIP_AMD_LIB"] = os.path.join(ROOT, "quip_amd", "csrc", "libquip_amd_probe.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from quip_amd import _lib, ops  # noqa: E402

NAMES = ["start", "every request issued (weights, then the x slabs)", "first chunk landed (wait -> first dequant)", "last chunk landed",
         "last MFMA issued", "partials parked in LDS", "block barrier passed", "reduce + epilogue + store issued"]


def run(m, d, bs, dt, launches=48, warm=False, cfg=None):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 4, (m, d), generator=g, dtype=torch.uint8).to(dev)
    q0 = ops.pack(codes, 2, ops.LAYOUT_STREAM)
    nring = 1 if warm else max(2, (400 << 20) // (m * d // 4) + 1)
    ring = [q0] + [q0.clone() for _ in range(nring - 1)]
    x = torch.randn(bs, d, generator=g).to(dt).to(dev)
    y = torch.empty(bs, m, dtype=dt, device=dev)
    sc = torch.tensor([0.05], device=dev)
    buf = torch.zeros(256, dtype=torch.int64, device=dev)
    _lib.call("quipamd_probe_set", buf.data_ptr())
    for i in range(nring):                                              # one walk: pages mapped, data long evicted again when its turn comes
        ops.dequant_gemm(x, ring[i], 2, "b", sc, None, None, out=y, cfg=cfg)
    torch.cuda.synchronize()
    rows = []
    for i in range(launches):
        buf.zero_()
        torch.cuda.synchronize()
        ops.dequant_gemm(x, ring[i % nring], 2, "b", sc, None, None, out=y, cfg=cfg)
        torch.cuda.synchronize()
        st = buf.cpu().numpy().astype(np.int64).reshape(16, 16)
        live = st[:, 0] > 0
        t0 = st[live][:, 0].min()
        rows.append(np.stack([np.where(st[live][:, k] > 0, st[live][:, k] - t0, -1) for k in range(8)], 1))
    _lib.call("quipamd_probe_set", None)
    a = np.stack(rows)                                                   # [launch, wave, slot]
    print(f"\n{'kernel cfg ' + str(cfg) if cfg else 'default kernel'}: {m} x {d}, bs {bs}, {str(dt).split('.')[-1]}, {'WARM (one copy)' if warm else f'COLD (ring of {nring} copies)'}: {a.shape[1]} waves stamped in workgroup 5, "
          f"median over {launches} launches; clocks (us at 2.4 GHz)")
    prev = 0
    for k in range(8):
        lo, hi = np.median(a[:, :, k].min(1)), np.median(a[:, :, k].max(1))
        print(f"  {k} {NAMES[k]:<52} {lo:7.0f} .. {hi:7.0f}   (+{hi - prev:6.0f} on the last wave; {hi / 2400:5.2f} us)")
        prev = hi


if __name__ == "__main__":
    run(4096, 4096, 16, torch.bfloat16, cfg=(2, 4, 4))               # dq_h_kernel 4 x 4: x through LDS-DMA slabs
    run(4096, 4096, 16, torch.bfloat16, warm=True, cfg=(2, 4, 4))
    run(4096, 4096, 16, torch.bfloat16, cfg=(2, 44, 4))              # dq_hr_kernel: x straight into registers (round 6)
    run(4096, 4096, 16, torch.bfloat16, warm=True, cfg=(2, 44, 4))
    run(8192, 2048, 16, torch.bfloat16)
    run(8192, 2048, 16, torch.bfloat16, warm=True)
