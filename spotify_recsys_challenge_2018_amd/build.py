"""Builds libdae_hip.so (hipcc, gfx950 only) in-tree.  `python -m spotify_recsys_challenge_2018_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdae_hip.so")
SOURCES = ["api.hip", "encode.hip", "decode_f32.hip", "topk.hip", "refine.hip", "audit.hip", "mixexact.hip", "train.hip", "csr.hip", "title.hip", "pipeline.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-vectorize: hipcc's LOOP vectorizer miscompiles per-lane strided loops of these kernels (a loop `for (i = lane; i < n;
# i += 64) { kl[b + i] = f(x[i]); ku[b + i] = g(x[i]); min / max of f }` leaves the key of entry i + 64 m in slot i of the
# first array: scripts/probe/vec_repro.hip is the 60-line reproducer, profiles/r05_notes.md the analysis -- the source has no
# aliasing or ordering bug, -fno-vectorize or -O1 give the right keys).  A SIMT kernel has nothing to gain from it: the
# hardware already runs 64 iterations per instruction.  (Round 4 carried `#pragma clang loop vectorize(disable)` on ten loops
# and -Wno-pass-failed instead; both are gone.)  The SLP vectorizer (packed 2 x fp32 / bf16 operations) stays on.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-vectorize", "-Wall", "-Wno-unused-function",
         # (only what is left of it: "loop not unrolled" notes for `#pragma unroll` loops of the generic-hidden-size template
         # instances of decode_f32_kernel, whose trip counts are run-time values there)
         "-Wno-pass-failed"]
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
