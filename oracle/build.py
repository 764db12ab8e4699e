"""Build the C part of the oracle (vq_oracle.c + fsq_oracle.c): oracle/_build/libvq_oracle.so (gcc, no GPU needed).

There is no oracle/_ref: the reference is pure Python (no C/C++ sources to compile), so
the restatement is pinned by fixtures generated from the importable reference modules
(tests/golden/make_golden.py) instead.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libvq_oracle.so")


SOURCES = ["vq_oracle.c", "fsq_oracle.c"]


def build(force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                           "-o", LIB] + srcs + ["-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
