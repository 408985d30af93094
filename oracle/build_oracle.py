"""Compile the C part of the CPU oracle (test infrastructure) into oracle/_build/."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_oracle_c(force=False):
    src = os.path.join(_HERE, "natac_oracle_c.c")
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libnatac_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # -ffp-contract=off: keep the reference's a*b then += rounding (no FMA fusion)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", so])
    return so


if __name__ == "__main__":
    print(build_oracle_c(force=True))
