"""Build recipe for the gfx950 HIP library (libstemgnn_hip.so), in-tree.

    python -m stem_gnn_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstemgnn_hip.so")
SOURCES = ["graph_build.hip", "sage_agg.hip", "bn_act.hip", "vq.hip", "edge_ops.hip", "linear.hip", "graph_aug.hip", "sampler.hip", "loss_ops.hip", "optim_ops.hip", "phases.hip", "heads.hip", "wsgemm.hip", "wspair.hip", "bigtile.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-munsafe-fp-atomics"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(PKG, "..", "include", "stemgnn.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(sp, obj) or any(_newer(h, obj) for h in headers):
            cmd = [HIPCC, *FLAGS, "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if verbose:
        print(f"compiled {len(procs)} of {len(objs)} sources" + (" (clean build)" if force else
              ": " + (", ".join(s for s, _ in procs) or "all objects up to date")), flush=True)
    if procs or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
