#!/usr/bin/env python3
"""Build libtitanet_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python titanet_amd/csrc/build.py [--force]

The library lands next to the package (titanet_amd/libtitanet_amd.so): it is git-ignored
but travels with the repo snapshot to the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["tn_api.hip", "tn_bwd.hip", "tn_mel.hip"]
# every header next to the sources feeds the rebuild digest (a stale .so must never ship: it travels prebuilt)
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + ["../../include/titanet_amd.h"]
OUT = os.path.join(PKG, "libtitanet_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + ["build.py"]:
        path = os.path.join(HERE, f)
        if os.path.exists(path):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stamp = os.path.join(HERE, ".build_stamp")
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return OUT
    objs = []

    def cc(src):
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
