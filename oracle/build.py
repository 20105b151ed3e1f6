"""Build the C oracle (test infrastructure) into oracle/_build/libswk_oracle.so.

The reference is pure Python whose heavy arithmetic lives in third-party wheels
(OpenCV, NumPy/LAPACK, SciPy, scikit-image), so there is no reference C/C++ to
compile into oracle/_ref/: the reference is "unbuildable here" in the sense of
the task statement and the oracle is pinned by fixtures generated from the
importable Python reference instead (oracle/make_goldens.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libswk_oracle.so")
SRC = os.path.join(HERE, "swk_oracle.c")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off",
           "-fvisibility=hidden", "-Wall", "-o", LIB, SRC, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
