"""Builds the EXPERIMENTS variant of the library (-DDAE_EXPERIMENTS: A/B switches, stage stamps) next to the
default one, as scripts/probe/libdae_hip_exp.so -- the default libdae_hip.so is not touched.  Use it through
DAE_LIB_AB=<path> with scripts/time_modes.py."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spotify_recsys_challenge_2018_amd import build as b      # noqa: E402

out_dir = "/tmp/dae_expbuild"
os.makedirs(out_dir, exist_ok=True)
flags = b.FLAGS + ([] if "-DDAE_EXPERIMENTS" in b.FLAGS else ["-DDAE_EXPERIMENTS"])


def cc(src):
    obj = os.path.join(out_dir, src.replace(".hip", ".o"))
    subprocess.check_call([b.HIPCC] + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj])
    return obj


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, b.SOURCES))
dst = os.path.join(ROOT, "scripts", "probe", "libdae_hip_exp.so")
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", dst] + objs)
print(dst)
