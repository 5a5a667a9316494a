"""Build ``libsrgd_hip.so`` (the C-ABI engine, include/srgd_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``srgd_amd/libsrgd_hip.so`` travels to the GPU box with the tree (git-ignored, not
gpurun-ignored).  ``python -m srgd_amd.build`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrgd_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
SOURCES = ["conv_igemm.hip", "conv3x3_bf16.hip", "conv3x3_split.hip", "conv3x3_mx2.hip", "conv1x1_split.hip", "conv1x1_bf16.hip", "conv3x3_mxfp8.hip", "conv1x1_mxfp8.hip", "quant_mxfp8.hip", "norm_act.hip", "attention.hip", "linattn_fused.hip", "linattn_fused256.hip", "cond.hip", "sampler.hip", "imageio.hip", "engine.hip", "kernel_api.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
         "-fno-gpu-rdc", "-DNDEBUG", "-fvisibility=hidden"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built (there is no CPU fallback)")


def _digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for name in sorted(os.listdir(root)):
            with open(os.path.join(root, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(b"version-script:srgd_*")
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    stamp = os.path.join(OBJ_DIR, "stamp")
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # version script: the dynamic symbol table holds the C ABI (srgd_*) and nothing else - without it the weak template
    # instantiations of libstdc++ types (default visibility by the standard library's own attribute) leak out as exports
    vers = os.path.join(OBJ_DIR, "exports.map")
    with open(vers, "w") as f:
        f.write("{ global: srgd_*; local: *; };\n")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vers}", "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
