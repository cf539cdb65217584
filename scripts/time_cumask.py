"""ONE spatial-partition experiment.  N batches in flight, each batch's context on a stream confined to its
own set of CUs (hipExtStreamCreateWithCUMask) -- against the same contexts on ordinary streams, same process, same box.
Partitions tried (scripts/probe/cumask_probe.hip tells how mask bits map to XCDs):
  none      ordinary streams (the shipped arrangement: every launch may use every CU)
  slice     stream j owns 1 / N of the CUs of EVERY XCD
  half      streams j and j + 2 share one half of every XCD's CUs, the others the other half
usage: time_cumask.py [B] [mode exact|bf16] [streams]"""
import ctypes
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "exact"
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
XCC_OF_BIT = os.environ.get("XCC_OF_BIT", "mod8")          # how mask bit i maps to an XCD: "mod8" (bit i -> XCD i % 8) or "div32"
dt = {"f32": 0, "bf16": 1, "exact": 2}[mode]
V, nt, H, k = 170000, 140000, 256, 500
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
rp, col, val = coo_to_csr(pos, ones, B, V)
srp, sc = seeds_to_csr(seeds, B, nt)
d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
hip = ctypes.CDLL("libamdhip64.so")


def xcc_of(i):
    return i % 8 if XCC_OF_BIT == "mod8" else i // 32


def masked_stream(pred):
    m = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            m[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, m)
    assert rc == 0, rc
    return s


def partition(kind, j):
    # (whole XCDs per stream are not expressible: scripts/probe/cumask_probe.hip -- a mask that leaves an XCD without a CU is
    # ignored, the stream then reaches all 256 CUs; mask bit i is CU i // 8 of XCD i % 8)
    if kind == "half":                    # two groups of streams, each group on half of every XCD's CUs
        return lambda i: (i // 8) % 2 == j % 2
    # slice: 1 / NS of the CUs of every XCD (the index of a CU inside its XCD: bit // 8 under mod8, bit % 32 under div32)
    inner = (lambda i: i // 8) if XCC_OF_BIT == "mod8" else (lambda i: i % 32)
    return lambda i: inner(i) % NS == j


ctxs = [_lib.Context(0) for _ in range(NS)]
ctxs[0].prepack_decoder(d[5], d[6], dtype=dt)
torch.cuda.synchronize()
for c in ctxs[1:]:
    c.share_decoder(ctxs[0], dt)
for c in ctxs:
    c.set_overlap_hint(NS)
torch.cuda.synchronize()
outs = [(torch.empty((B, k), device="cuda"), torch.empty((B, k), dtype=torch.int32, device="cuda")) for _ in range(NS)]
hs = [c.score_topk_handle(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, outs[j][0], outs[j][1], dtype=dt) for j, c in enumerate(ctxs)]
torch_streams = [torch.cuda.Stream() for _ in range(NS)]
ref = None
for kind in ("none", "slice", "half", "none"):
    if kind == "none":
        raw = [ctypes.c_void_p(s.cuda_stream) for s in torch_streams]
    else:
        raw = [masked_stream(partition(kind, j)) for j in range(NS)]
    for c, s in zip(ctxs, raw):
        c.check(c.lib.dae_set_stream(c.h, s))
    for _ in range(40):
        for h in hs:
            h()
    torch.cuda.synchronize()
    N = 800
    t0 = time.perf_counter()
    for i in range(N):
        hs[i % NS]()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    same = True
    if ref is None:
        ref = [o[1].clone() for o in outs]
    else:
        same = all(torch.equal(a, o[1]) for a, o in zip(ref, outs))
    print(f"B={B} {mode} streams={NS} partition={kind:5s}: {el / N * 1e6:.1f} us/step = {B * N / el / 1e6:.3f} M playlists/s  same_lists={same}", flush=True)
