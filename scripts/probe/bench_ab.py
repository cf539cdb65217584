"""bench.py's own main() on another build of the library (DAE_LIB_AB): an A/B harness, never a published number.
usage: DAE_LIB_AB=<so> python scripts/probe/bench_ab.py <bench.py arguments>"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spotify_recsys_challenge_2018_amd import _lib
if os.environ.get("DAE_LIB_AB"):
    _lib.LIB_PATH = os.environ["DAE_LIB_AB"]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
