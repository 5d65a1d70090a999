"""oracle/build.py -- compile oracle/ldlq_oracle.c into oracle/liboracle.so with gcc.

TEST INFRASTRUCTURE.  Called by __graft_entry__.build() and lazily by
oracle/quip_oracle.py.  Nothing of the reference is compiled (it is pure Python);
oracle/stage_ref.py stages its driver files into oracle/_ref/ for the GPU driver
tests, see DESIGN.md "Oracle".
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    src = os.path.join(HERE, "ldlq_oracle.c")
    out = os.path.join(HERE, "liboracle.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", out, src, "-lm"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True))
