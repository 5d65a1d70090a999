"""CPU test (needs hipcc, no GPU): audit of the generated gfx950 code of dq_h_kernel (csrc/dqgemm_v2.h).

That kernel hides its weight loads in inline asm (hipcc would otherwise drain the LDS-DMA queue at their first use).  hipcc
does not track an asm load: it may read, copy or reuse the destination registers before the data has landed
(cdna_hip_programming.md 5.7 item 1) -- the first version of the streaming kernels failed exactly like that on the GPU.
The audit replays the vector-memory queue of every dq_h_kernel instantiation in program order: an instruction that
touches the destination registers of an asm load which no s_waitcnt vmcnt(N) has retired yet is a failure.
It also checks that no instantiation spills and that no waterfall loop wraps the DMA instructions."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    """'v[6:9]' -> {6,7,8,9}; 'v12' -> {12}"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "dqgemm_v2.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "quip_amd", "csrc"),
                           os.path.join(ROOT, "quip_amd", "csrc", "dqgemm_v2.hip"), "-o", str(out)])
    return out.read_text().splitlines()


def _functions(lines, prefix):
    cur, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if ".end_amdhsa_kernel" in ln:
                if prefix in cur:
                    yield cur, body
                cur = None


def _h_kernels(asm):
    """every instantiation of the one-problem kernel and of the grouped one (round 5: dq_hg_kernel, the same body text with the problem
    picked by blockIdx.y)"""
    yield from _functions(asm, "dq_h_kernel")
    yield from _functions(asm, "dq_hg_kernel")
    yield from _functions(asm, "dq_hr_kernel")       # round 6: x as asm loads into registers too (32 per wave behind the 4 weight tiles)


def test_h_kernel_asm_loads_are_not_touched_before_their_wait(asm):
    n = 0
    assert sum(1 for _ in _functions(asm, "dq_hg_kernel")) >= 4, "no dq_hg_kernel instantiations found"
    assert sum(1 for _ in _functions(asm, "dq_hr_kernel")) >= 2, "no dq_hr_kernel instantiations found"
    for name, body in _h_kernels(asm):
        n += 1
        queue = []          # outstanding vector-memory operations, oldest first: (dest registers of an asm load | empty set)
        in_asm = False
        meta = "\n".join(body)
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), f"{name}: spills"
        for i, ln in enumerate(body):
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            op = s.split()[0]
            touched = _regs(s.split(";")[0])
            pending = set().union(*[q for q in queue]) if queue else set()
            is_asm_load = in_asm and op.startswith("global_load")
            if not is_asm_load and (touched & pending):
                raise AssertionError(f"{name}: line {i}: `{s}` touches registers {sorted(touched & pending)} of an asm load "
                                     f"that no s_waitcnt has retired")
            if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "scratch_", "flat_")):
                if is_asm_load:
                    dst = _regs(s.split(",")[0])
                    addr = _regs(",".join(s.split(",")[1:]))
                    assert not (addr & pending), f"{name}: line {i}: asm load address in flight"
                    queue.append(dst)
                else:
                    queue.append(set())
            m = re.search(r"vmcnt\((\d+)\)", s)
            if op == "s_waitcnt" and m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            if op == "s_endpgm":
                break
    assert n >= 8, "no dq_h_kernel instantiations found"


def test_no_waterfall_loops_around_the_dma(asm):
    """a buffer descriptor hipcc cannot prove wave-uniform gets every buffer_load wrapped in a readfirstlane loop
    (cdna_hip_programming.md T20): the streaming kernels must not have any next to their LDS-DMA instructions."""
    for prefix in ("dq_h_kernel", "dq_hg_kernel", "dq_s_kernel", "dq_mb_kernel"):
        for name, body in _functions(asm, prefix):
            for i, ln in enumerate(body):
                if "buffer_load_dwordx4" in ln and " lds" in ln:
                    window = "\n".join(body[max(0, i - 6):i])
                    assert "v_readfirstlane_b32" not in window or "s_and_saveexec" not in window, f"{name}: waterfall loop at line {i}"
