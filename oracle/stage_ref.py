"""oracle/stage_ref.py -- stage the reference's own DRIVER files where a GPU box can see them.

TEST INFRASTRUCTURE.  The reference is pure Python (nothing to compile), but the drop-in claim "the reference's
opt_sequential / llama_sequential run unmodified on quip_amd" can only be EXECUTED on a machine that has both a GPU and the
reference's driver files; the authoring container has /root/reference and no GPU, the GPU boxes have a GPU and no reference
tree.  This recipe copies the three driver files, byte for byte, from where they lie under the reference checkout into
oracle/_ref/ -- git-ignored (never in history), not gpurun-ignored (travels with the snapshot like the built .so files):

    opt.py        opt_sequential (opt.py:29-190), opt_eval, benchmark (opt.py:431-482)
    llama.py      llama_sequential (llama.py:36-171), llama_eval, benchmark (llama.py:418-471)
    datautils.py  imported by llama.py:12 at module load

and, into oracle/_ref/cpu/ (a separate directory, so that they can never shadow quip_amd's aliases when a staged driver runs), the
four files of the reference's CPU LDLQ path that bench.py's `ldlq_cpu_reference` leg times on the GPU box's host cores
(oracle/ref_ldlq_time.py, run in a subprocess):

    quant.py  method.py  vector_balance.py  bal.py        Balance.fasterquant -> round_ldl / round_ldl_block (bal.py:21-48,
                                                          vector_balance.py:155-199,218-291)

Nothing under quip_amd/, bench.py's timed region or smoke() reads oracle/_ref; scripts/run_reference_driver.py and
tests/test_gpu_driver.py import the staged drivers ON TOP of quip_amd (module aliasing) -- the
staged files are the caller under test, quip_amd is what they call.  A MANIFEST with the SHA-256 of every staged file is
written beside them so that a test can say which bytes ran.

usage: python oracle/stage_ref.py [/path/to/QuIP]        (default /root/reference; called by __graft_entry__.build())"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
FILES = ("opt.py", "llama.py", "datautils.py")
CPU_FILES = ("quant.py", "method.py", "vector_balance.py", "bal.py")      # -> oracle/_ref/cpu/


def stage(ref="/root/reference"):
    """copy FILES from `ref` into oracle/_ref/; returns the manifest dict, or None when there is no reference checkout"""
    if not all(os.path.exists(os.path.join(ref, f)) for f in FILES):
        return None
    os.makedirs(DEST, exist_ok=True)
    manifest = {}
    for f in FILES:
        src, dst = os.path.join(ref, f), os.path.join(DEST, f)
        shutil.copyfile(src, dst)
        with open(dst, "rb") as fh:
            manifest[f] = hashlib.sha256(fh.read()).hexdigest()
    if all(os.path.exists(os.path.join(ref, f)) for f in CPU_FILES):
        os.makedirs(os.path.join(DEST, "cpu"), exist_ok=True)
        for f in CPU_FILES:
            dst = os.path.join(DEST, "cpu", f)
            shutil.copyfile(os.path.join(ref, f), dst)
            with open(dst, "rb") as fh:
                manifest["cpu/" + f] = hashlib.sha256(fh.read()).hexdigest()
    with open(os.path.join(DEST, "MANIFEST.json"), "w") as fh:
        json.dump({"source": ref, "sha256": manifest}, fh, indent=1)
    return manifest


def staged_cpu():
    """the directory holding the reference's CPU LDLQ modules, or None"""
    d = os.path.join(DEST, "cpu")
    return d if all(os.path.exists(os.path.join(d, f)) for f in CPU_FILES) else None


def staged():
    """the directory holding the staged drivers, or None"""
    return DEST if all(os.path.exists(os.path.join(DEST, f)) for f in FILES) else None


if __name__ == "__main__":
    print(stage(sys.argv[1] if len(sys.argv) > 1 else "/root/reference"))
