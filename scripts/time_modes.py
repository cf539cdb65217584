"""Steady-state timing of dae_score_topk per decode mode at full size: ms per step with N batches in flight (N contexts
on N streams, round robin).  usage: time_modes.py [B] [bias] [modes f32,bf16,exact] [streams 1,2,3]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib
if os.environ.get("DAE_LIB_AB"):       # A/B against another build of the library
    _lib.LIB_PATH = os.environ["DAE_LIB_AB"]
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bias = sys.argv[2] if len(sys.argv) > 2 else "zipf"
only = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "bf16", "exact"]
nstr = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1]
V, nt, H, k = 170000, 140000, 256, 500
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias=bias, n_tracks=nt)
if os.environ.get("HEAVY"):           # decoder rows of very different norms (popular tracks of a trained model): x (1 + HEAVY u^8)
    fac = 1.0 + float(os.environ["HEAVY"]) * np.random.default_rng(5).random(V) ** 8
    W_dec = (W_dec * fac[:, None]).astype(np.float32)
if os.environ.get("SCALE"):            # a model whose rows rank the tracks differently (x40: scripts/tf_gap.py's third case)
    W_enc = (W_enc * float(os.environ["SCALE"])).astype(np.float32)
    W_dec = (W_dec * float(os.environ["SCALE"])).astype(np.float32)
pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
if os.environ.get("TRAINED"):          # a model TRAINED on clustered synthetic playlists (utils/synthetic.py), its own kind of batch
    from spotify_recsys_challenge_2018_amd.utils.synthetic import train_clustered_model
    W_enc, b_enc, W_dec, b_dec, gen, info = train_clustered_model(nt, V - nt, H, steps=int(os.environ["TRAINED"]), log=print)
    print("trained:", info, flush=True)
    pos, ones, seeds = gen.scoring_feed(B, np.random.default_rng(77))
    print("b_dec range", float(b_dec.min()), float(b_dec.max()), "|W_dec| row-norm1 mean/max",
          float(np.abs(W_dec).sum(1).mean()), float(np.abs(W_dec).sum(1).max()), flush=True)
rp, col, val = coo_to_csr(pos, ones, B, V)
srp, sc = seeds_to_csr(seeds, B, nt)
d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
NS = max(nstr)
ctxs = [_lib.Context(0) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
idle_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("EXTRA_STREAMS", "0")))]      # A/B: idle streams (hardware queues) in the process
for s_ in idle_streams:
    with torch.cuda.stream(s_):
        torch.zeros(1, device="cuda")
share = os.environ.get("SHARE", "1") != "0"          # one packed image for all contexts (dae_share_decoder)
for n_, (c, st) in enumerate(zip(ctxs, streams)):
    with torch.cuda.stream(st):
        c.bind_stream()
        if n_ == 0 or not share:
            if "f32" in only:
                c.prepack_decoder(d[5], d[6])
            c.prepack_decoder(d[5], d[6], dtype=_lib.DAE_DTYPE_BF16_EXACT)
    torch.cuda.synchronize()
    if n_ > 0 and share:
        if "f32" in only:
            c.share_decoder(ctxs[0], 0)
        c.share_decoder(ctxs[0], _lib.DAE_DTYPE_BF16_EXACT)
torch.cuda.synchronize()
print("shared image:", share, flush=True)
outs = [(torch.empty((B, k), device="cuda"), torch.empty((B, k), dtype=torch.int32, device="cuda")) for _ in range(NS)]
ref = None
for name, dt in (("f32", 0), ("bf16", 1), ("exact", 2)):
    if name not in only:
        continue
    for n in nstr:
        for c in ctxs:
            c.set_overlap_hint(n if os.environ.get("HINT", "1") != "0" else 1)
            c.profile_enable(os.environ.get("PROFILE", "0") == "1")     # event pairs around the dominant launch, as bench.py

        def step(i):
            j = i % n
            with torch.cuda.stream(streams[j]):
                ctxs[j].score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, outs[j][0], outs[j][1], dtype=dt)
        for i in range(60):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 600
        for i in range(N):
            step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / N * 1e3
        for c in ctxs:
            c.profile_read()
        s, i_ = outs[0]
        if name == "f32" and ref is None:
            ref = (s.clone(), i_.clone())
        same = ref is not None and bool(torch.equal(i_, ref[1]) and torch.equal(s.view(torch.int32), ref[0].view(torch.int32)))
        extra = ""
        if name == "exact":
            st = [c.exact_stats_read() for c in ctxs[:n]]
            extra = "  candidates/row=%.0f recomputed/row=%.0f guard=%s" % (
                np.mean([x["candidates_per_row"] for x in st]), np.mean([x["recomputed_per_row"] for x in st]),
                [c.exact_guard_read() for c in ctxs[:n]][0])
        print(f"{name} streams={n}: {ms:.4f} ms/step  {B / ms * 1e3 / 1e6:.3f} M playlists/s  identical_to_f32={same}{extra}", flush=True)
