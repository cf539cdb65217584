"""Does a small kernel get a slot next to a running fp32 filter launch?  Stream A: dae_score_topk (fp32) in a loop; stream B:
M launches of a small kernel (encode / a torch elementwise add).  Prints B's time alone and under A.
usage: coresidency_probe.py [mode f32|bf16|exact]"""
import os
import sys
import time
import numpy as np
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
dt = {"f32": 0, "bf16": 1, "exact": 2}[mode]
B, V, nt, H, k = 256, 170000, 140000, 256, 500
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
rp, col, val = coo_to_csr(pos, ones, B, V)
srp, sc = seeds_to_csr(seeds, B, nt)
d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cA, cB = _lib.Context(0), _lib.Context(0)
with torch.cuda.stream(sA):
    cA.bind_stream(); cA.prepack_decoder(d[5], d[6], dtype=dt)
with torch.cuda.stream(sB):
    cB.bind_stream()
torch.cuda.synchronize()
out = (torch.empty((B, k), device="cuda"), torch.empty((B, k), dtype=torch.int32, device="cuda"))
h = torch.empty((B, H), device="cuda")
x = torch.zeros(4096, device="cuda")


def run_A(n):
    with torch.cuda.stream(sA):
        for _ in range(n):
            cA.score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, out[0], out[1], dtype=dt)


def small(kind):
    if kind == "encode":
        cB.encode(d[0], d[1], d[2], d[3], d[4], h)
    elif kind == "probe":
        small_probe()
    else:
        x.add_(1.0)


def run_B(kind, m):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sB):
        e0.record()
        for _ in range(m):
            small(kind)
        e1.record()
    return e0, e1


import ctypes
probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libreg_probe.so"))
probe.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
cur = {}


def small_probe():
    rc = probe.probe_launch(sB.cuda_stream, cur["regs"], cur["threads"], cur["blocks"], cur["lds"], 1500, x.data_ptr(), cur.get("prio", 0))
    assert rc == 0, rc


run_A(20); torch.cuda.synchronize()
cases = [("encode", None), ("add", None)]
for prio in (0, 1):
    for regs in (32, 96, 200):
        cases.append(("probe", dict(regs=regs, threads=64, blocks=1024, lds=0, prio=prio)))
    cases.append(("probe", dict(regs=96, threads=256, blocks=256, lds=0, prio=prio)))
for kind, cfg in cases:
    if cfg:
        cur.update(cfg)
    M = 300
    e0, e1 = run_B(kind, M); torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) * 1e3 / M
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sA):
        a0.record()
    run_A(150)                                   # ~30 ms of work queued on A
    with torch.cuda.stream(sA):
        a1.record()
    e0, e1 = run_B(kind, M)
    torch.cuda.synchronize()
    print("%s %-6s %s: alone %.1f us per launch; under stream A %.1f us per launch (A: %.1f us per step)"
          % (mode, kind, cfg or "", alone, e0.elapsed_time(e1) * 1e3 / M, a0.elapsed_time(a1) * 1e3 / 150), flush=True)
