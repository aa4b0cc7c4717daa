"""Compile the oracle's C pieces (gcc) into oracle/_build/liboracle.so.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "liboracle.so")
SRCS = [os.path.join(HERE, "nms_oracle.c")]


def build(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(s) for s in SRCS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", OUT, *SRCS, "-lm"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(True))
