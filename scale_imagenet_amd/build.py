"""Builds libttnet.so (the C-ABI library of include/ttnet.h) in-tree with hipcc for gfx950.

``python -m scale_imagenet_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU; the .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libttnet.so")
SOURCES = ["plan.hip", "lut_build.hip", "stem.hip", "gate.hip", "gate_fused.hip", "gate_xs.hip", "gate_full.hip", "gate_va.hip", "head.hip", "preproc.hip"]
# gate_full.hip: the same flag keeps the float32 GELU of the fast kernels out of v_pk_* with shuffling moves.
# stem.hip: without -fno-slp-vectorize the producers' pooling adds become v_pk_add_f32 behind shuffling moves, which
# beside the consumers' MFMAs cost 82 us per launch instead of 65 (tools/ubench/stem_parts.hip)
EXTRA = {"stem.hip": ["-fno-slp-vectorize"], "gate_full.hip": ["-fno-slp-vectorize"]}
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "ttnet_common.h"), os.path.join(HERE, "..", "include", "ttnet.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
