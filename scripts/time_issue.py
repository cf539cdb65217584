"""Is the step loop bound by the host's issue rate?  dae_score_topk through pre-marshalled handles, N batches in flight:
(a) one thread issuing round robin -- time to ISSUE the steps against the time until they have run; (b) one issuing thread per
stream (ctypes drops the GIL in the foreign call); (c) each step replayed from a captured graph.
usage: time_issue.py [B] [mode exact|bf16|f32] [streams]"""
import os
import sys
import threading
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
dt = {"f32": 0, "bf16": 1, "exact": 2}[mode]
V, nt, H, k = 170000, 140000, 256, 500
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
rp, col, val = coo_to_csr(pos, ones, B, V)
srp, sc = seeds_to_csr(seeds, B, nt)
d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
ctxs = [_lib.Context(0) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
for n_, (c, st) in enumerate(zip(ctxs, streams)):
    with torch.cuda.stream(st):
        c.bind_stream()
        if n_ == 0:
            c.prepack_decoder(d[5], d[6], dtype=dt)
    torch.cuda.synchronize()
    if n_ > 0:
        c.share_decoder(ctxs[0], dt)
    c.set_overlap_hint(NS)
torch.cuda.synchronize()
outs = [(torch.empty((B, k), device="cuda"), torch.empty((B, k), dtype=torch.int32, device="cuda")) for _ in range(NS)]
hs = [c.score_topk_handle(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, outs[j][0], outs[j][1], dtype=dt) for j, c in enumerate(ctxs)]
for _ in range(30):
    for h in hs:
        h()
torch.cuda.synchronize()
N = 800

# (a) one thread
t0 = time.perf_counter()
for i in range(N):
    hs[i % NS]()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} {mode} streams={NS} one thread : issue {(t1 - t0) / N * 1e6:.1f} us/step, done {(t2 - t0) / N * 1e6:.1f} us/step "
      f"= {B * N / (t2 - t0) / 1e6:.3f} M playlists/s", flush=True)


# (b) one thread per stream
def worker(h, n):
    for _ in range(n):
        h()
ths = [threading.Thread(target=worker, args=(hs[j], N // NS)) for j in range(NS)]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} {mode} streams={NS} {NS} threads  : issue {(t1 - t0) / N * 1e6:.1f} us/step, done {(t2 - t0) / N * 1e6:.1f} us/step "
      f"= {B * N / (t2 - t0) / 1e6:.3f} M playlists/s", flush=True)

# (c) captured graphs
try:
    graphs = []
    for j in range(NS):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[j], capture_error_mode="relaxed"):
            hs[j]()
        graphs.append(g)
    torch.cuda.synchronize()
    for _ in range(10):
        for j in range(NS):
            with torch.cuda.stream(streams[j]):
                graphs[j].replay()
    torch.cuda.synchronize()
    ref = [o[1].clone() for o in outs]
    t0 = time.perf_counter()
    for i in range(N):
        graphs[i % NS].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} {mode} streams={NS} graphs     : issue {(t1 - t0) / N * 1e6:.1f} us/step, done {(t2 - t0) / N * 1e6:.1f} us/step "
          f"= {B * N / (t2 - t0) / 1e6:.3f} M playlists/s  same_lists={all(torch.equal(a, o[1]) for a, o in zip(ref, outs))}", flush=True)
except Exception as e:                       # noqa: BLE001
    print("graph capture failed:", repr(e)[:300], flush=True)
