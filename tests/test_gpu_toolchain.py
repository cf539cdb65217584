"""GPU (-m gpu): the reason behind a build flag stays checked (VERDICT r5 Weak #13).  The library is built with -fno-vectorize
because hipcc's LOOP vectorizer miscompiles a per-lane strided staging loop of these kernels (scripts/probe/vec_repro.hip: the
60-line reproducer, a one-thread recomputation of every staged key as the checker; profiles/r05_notes.md 4).  If a toolchain
update moves the bug -- into the SLP vectorizer that stays on, say -- the parity tests would catch wrong lists, but not say why;
this test does: the reproducer built WITH the library's flags must stage every key right."""
import os
import shutil
import subprocess

import pytest

from spotify_recsys_challenge_2018_amd import build as hip_build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scripts", "probe", "vec_repro.hip")


def _build_and_run(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call([hip_build.HIPCC] + flags + [SRC, "-o", exe], cwd=ROOT)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("entries")][-1]
    # "entries N | pragma: lower-bound keys wrong a, upper b | no pragma: lower c, upper d"
    nums = [int(tok.strip(",")) for tok in line.replace("|", " ").split() if tok.strip(",").isdigit()]
    return r.returncode, nums, line


@pytest.mark.skipif(not (os.path.exists(hip_build.HIPCC) or shutil.which(hip_build.HIPCC)), reason="no hipcc on this box")
def test_the_librarys_flags_keep_the_staging_loop_right(tmp_path):
    flags = [f for f in hip_build.FLAGS if f not in ("-fPIC",) and not f.startswith("-W")]
    assert "-fno-vectorize" in flags, "the library is no longer built with -fno-vectorize: say why in build.py and here"
    rc, nums, line = _build_and_run(tmp_path, "vec_ok", flags)
    assert rc == 0 and nums[1:] == [0, 0, 0, 0], "wrong staged keys WITH the library's flags: " + line
    # without the flag: the bug as it was found (reported, not asserted -- a fixed compiler is good news, and then says so here)
    rc2, nums2, line2 = _build_and_run(tmp_path, "vec_vec", [f for f in flags if f != "-fno-vectorize"])
    print("without -fno-vectorize:", line2, "(the loop vectorizer %s)" % ("still miscompiles the loop" if rc2 else "no longer miscompiles it"))
