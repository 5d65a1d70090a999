#!/usr/bin/env python3
"""Phase budget of every launch of one decode step, IN SITU (VERDICT r5 next #1a -> profiles/r06_decode_stamps.txt).

The decode engine (quip_amd.decode.DecodeEngine) runs a full-size packed model eagerly, launch by launch, on the PROBE build of the library
(quip_amd/csrc/libquip_amd_probe.so: csrc/probe.h, -DQA_PROBE) -- the shipped kernels plus s_memtime stamps at their phase boundaries, written
by every wave of one workgroup of each launch.  Every C-ABI call of one token is intercepted: the stamp buffer is cleared, the launch runs
alone (synchronised), the buffer is read.  So each launch meets the caches as a real step leaves them -- its weights, factor fragments, index
vectors and gains were last touched one token (508 MB of traffic) ago -- which the lab harnesses with their handful of operand copies do not.

Printed per launch of ONE block in the middle of the model: for every phase the clock (since the first wave's first stamp) at which the
FIRST and the LAST wave pass the stamp, and the kernel's own span.  Next to it scripts/decode_timeline.py turns a rocprofv3 kernel trace of
the graph-replayed engine into (duration, gap to the next launch) per position in the block: span + gap = the period the token pays.

    python scripts/decode_stamps.py [--arch opt|llama] [--blocked] [--block 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "quip_amd", "csrc", "libquip_amd_probe.so")
os.environ["QUIP_AMD_LIB"] = PROBE                       # before quip_amd is imported
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# slot names per C entry point (the order of the QA_STAMP / FG_STAMP / K2_STAMP calls in the kernels)
FUSED = ["start (kernarg in)", "U: row landed, copied", "barrier", "U mix (its fragments landed)", "barrier", "gather (idx, bias, residual landed) / x row",
         "norm (gains landed)", "V scatter (colscale, idx landed)", "barrier", "V stage 1 (fragments landed)", "barrier", "V stage 2 -> x~", "barrier",
         "dequant + MFMA (weights landed)", "park + barrier", "reduce + store"]
NAMES = {
    "quipamd_decode_fused_gemm": FUSED,
    "quipamd_decode_attention_fused": ["start", "position landed", "requests issued", "y landed, copied", "barrier", "stage 1 + barrier", "stage 2 + barrier",
                                       "gather + barrier", "rotary / append + barrier", "scores (K rows landed)", "softmax", "p V", "end"],
    "quipamd_ortho_blocked_rows": ["start", "requests issued", "row statistics (unfused + norm)", "FUSED: rows pre-processed (registers) / staged (general form)",
                                   "barrier", "FUSED: rows in LDS (general form: + statistics, gains)", "FUSED: first-stage products", "input vector in LDS",
                                   "barrier", "MFMAs (factors landed)", "barrier (not SOLO)", "store"],
    "quipamd_dequant_gemm": ["start", "requests issued", "first chunk landed", "last chunk landed", "MFMAs done", "parked", "barrier", "reduce + store"],
}
NAMES["quipamd_decode_attention"] = NAMES["quipamd_decode_attention_fused"]     # the plain launch stamps slots 0, 1, 8 .. 12 of the same list
NAMES["quipamd_ortho_blocked_rows_multi"] = NAMES["quipamd_ortho_blocked_rows"]
NAMES["quipamd_dequant_gemm_grouped"] = NAMES["quipamd_dequant_gemm"]
NAMES["quipamd_dequant_gemm_cfg"] = NAMES["quipamd_dequant_gemm"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="opt", choices=["opt", "llama"])
    ap.add_argument("--blocked", action="store_true")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--block", type=int, default=-1, help="the block whose launches are printed (default: the middle one)")
    ap.add_argument("--position", type=int, default=96, help="tokens decoded before the stamped step")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--per-wave", action="store_true", help="print every wave's clock for every slot (who is late)")
    a = ap.parse_args()
    if not os.path.exists(PROBE):
        raise SystemExit(f"{PROBE} missing: python __graft_entry__.py --probe")
    from quip_amd import _lib, decode
    import decode_engine_bench as B
    model, nbytes, arch = B.build(a)
    dev = torch.device("cuda:0")
    eng = decode.DecodeEngine(model, bs=1, max_len=a.position + 16, mode="auto", graph=False)
    ids = torch.randint(0, 30000, (1, a.position + 2), device=dev)
    for i in range(a.position):                                        # fill the cache, warm every table and the allocator
        eng.forward(ids[:, i])
    torch.cuda.synchronize()
    buf = torch.zeros(256, dtype=torch.int64, device=dev)
    _lib.call("quipamd_probe_set", buf.data_ptr())
    rec = []
    orig = _lib.call

    def hooked(name, *args):
        buf.zero_()
        torch.cuda.synchronize()
        rc = orig(name, *args)
        torch.cuda.synchronize()
        rec.append((name, buf.cpu().numpy().astype(np.uint64).reshape(16, 16).copy()))
        return rc
    _lib.call = hooked
    try:
        eng.forward(ids[:, a.position])
    finally:
        _lib.call = orig
        orig("quipamd_probe_set", None)
    nb = len(model.blocks)
    launches = [r for r in rec if r[0] != "quipamd_probe_set"]
    names_seq = [r[0] for r in launches]
    per, first = 0, 0
    n_ = len(names_seq)
    for p_ in range(2, 80):                                             # the block's launch sequence = the shortest period that holds over the whole step
        lo_, hi_ = 2, n_ - p_ - 3
        if hi_ - lo_ < 2 * p_:
            break
        if all(names_seq[i] == names_seq[i + p_] for i in range(lo_, hi_)):
            per = p_
            break
    first = (n_ - nb * per) // 2 if per else 0                         # what is not a block sits at the two ends (embed | head, argmax)
    if eng.mode == "v3_head":
        first = 1
    blk = a.block if a.block >= 0 else nb // 2
    first += min(blk, nb - 2) * per
    print(f"# {arch}; operators: {'blocked butterfly (the shipped flag, preproc_proj_extra = 0)' if a.blocked else 'Kronecker (preproc_proj_extra = 1)'}; "
          f"engine mode {eng.mode}; {len(launches)} C-ABI launches in the step, {per} per block x {nb} blocks; position {a.position}")
    print(f"# stamps of the {per} launches from launch {first} on (block {blk}): clocks since the first stamp of the launch (first wave .. last wave to pass "
          f"it); us at {a.clock_ghz} GHz")
    for k in range(first, first + per):
        name, st = launches[k]
        live = st[:, 0] > 0
        if not live.any():
            print(f"\n[{k - first}] {name}: no stamps (a kernel without probe points)")
            continue
        t0 = st[live][:, :].reshape(-1)
        t0 = int(t0[t0 > 0].min())
        names = NAMES.get(name, [])
        span = int(st.max()) - t0
        print(f"\n[{k - first}] {name}: {int(live.sum())} waves stamped, span {span} clocks = {span / a.clock_ghz / 1e3:.2f} us")
        prev = 0
        for i in range(16):
            col = st[:, i]
            col = col[col > 0]
            if col.size == 0:
                continue
            lo, hi = int(col.min()) - t0, int(col.max()) - t0
            nm = names[i] if i < len(names) else f"slot {i}"
            print(f"    {i:2d} {nm:<52} {lo:7d} .. {hi:7d}   (+{hi - prev:6d} on the last wave)")
            prev = hi
            if a.per_wave:
                print("         waves: " + " ".join(f"{int(v) - t0:6d}" if v > 0 else "     -" for v in st[:, i]))


if __name__ == "__main__":
    main()
