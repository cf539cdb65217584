"""Builds libdae_hip.so (hipcc, gfx950 only) in-tree.  `python -m spotify_recsys_challenge_2018_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdae_hip.so")
SOURCES = ["api.hip", "encode.hip", "decode_f32.hip", "topk.hip", "refine.hip", "mixexact.hip", "train.hip", "csr.hip", "title.hip", "pipeline.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
if os.environ.get("DAE_EXPERIMENTS"):          # A/B switches and stage early-outs (csrc/dae_internal.h); never the default
    FLAGS.append("-DDAE_EXPERIMENTS")
FLAGS += os.environ.get("DAE_EXTRA_FLAGS", "").split()        # e.g. -DDAE_SMALL_PRIO=0 for an A/B build


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    objs = []

    def cc(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
