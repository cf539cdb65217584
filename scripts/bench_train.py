"""Training-step timing at BASELINE.json configs[3] shape (V=170 000, H=256, B=256, fp32 MFMA):
forward with dropout + loss + backward (dae_train_forward_backward) + dense Adam on all variables."""
import json, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib
if os.environ.get("DAE_LIB_AB"):       # A/B against another build of the library
    _lib.LIB_PATH = os.environ["DAE_LIB_AB"]
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights
tied = "--tied" in sys.argv
V, nt, H, B = 170000, 140000, 256, 256
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zeros", n_tracks=nt, tied=tied)
pos, ones, _ = make_playlists(B, nt, V - nt, seed=1, seed_counts=(20, 40, 66, 100))
m = pos[:, 1] < nt
xr, xc, xv = coo_to_csr(pos[m], ones[m], B, V)
yr, yc, yv = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ctx = _lib.Context(0)
if "--bf16" in sys.argv:
    ctx.set_train_dtype(_lib.DAE_DTYPE_BF16)      # bf16 operands in the three GEMMs (BASELINE configs[3])
P = _lib._ptr
t = {k: dev(v) for k, v in dict(xr=xr, xc=xc, xv=xv, yr=yr, yc=yc, yv=yv, We=W_enc, be=b_enc, Wd=W_dec, bd=b_dec).items()}
names = ["We", "be", "bd"] + ([] if tied else ["Wd"])
g = {n: torch.zeros_like(t[n]) for n in names}
mom = {n: (torch.zeros_like(t[n]), torch.zeros_like(t[n])) for n in names}
cost = torch.zeros(1, device="cuda")
default_f32 = "--default-f32" in sys.argv and not tied      # the same step with train_dtype = f32 (config.ini's default)
default = ("--default" in sys.argv or default_f32) and not tied      # models/DAEs.py train_step: the fused / row-sparse Adam forms (bf16 GEMMs)
if default:
    ctx.set_train_dtype(_lib.DAE_DTYPE_F32 if default_f32 else _lib.DAE_DTYPE_BF16)
    lz = {"state": torch.zeros(2 * V, dtype=torch.int32, device="cuda"), "tab": torch.zeros(1 << 12, dtype=torch.float32, device="cuda"),
          "flushed": 0}
    ctx.check(ctx.lib.dae_set_enc_grad_prezeroed(ctx.h, 1))
    import ctypes
    rows_arg = (P(t["xc"]), ctypes.c_void_p(t["xr"].data_ptr() + 4 * B), int(t["xc"].numel()))
def step(i):
    if default:
        ts = i + 1
        ctx.check(ctx.lib.dae_arm_decoder_adam(ctx.h, P(mom["Wd"][0]), P(mom["Wd"][1]), 0.005, 0.9, 0.999, 1e-8, ts))
        ctx.check(ctx.lib.dae_adam_rows_begin(ctx.h, P(t["We"]), P(mom["We"][0]), P(mom["We"][1]), P(lz["state"]), P(lz["tab"]),
                                              lz["tab"].numel(), V, H, rows_arg[0], rows_arg[1], rows_arg[2], 0.9, 0.999, 1e-8, ts))
        ctx.check(ctx.lib.dae_train_forward_backward(ctx.h, P(t["xr"]), P(t["xc"]), P(t["xv"]), P(t["yr"]), P(t["yc"]), P(t["yv"]),
            P(t["We"]), P(t["be"]), P(t["Wd"]), P(t["bd"]), V, H, B, B, 0, 0.75, 0.8, 100 + i, 0.0,
            P(g["We"]), P(g["be"]), P(g["Wd"]), P(g["bd"]), P(cost)))
        ctx.check(ctx.lib.dae_adam_rows_apply(ctx.h, P(t["We"]), P(mom["We"][0]), P(mom["We"][1]), P(g["We"]), P(lz["state"]),
                                              P(lz["tab"]), lz["tab"].numel(), V, H, rows_arg[0], rows_arg[1], rows_arg[2],
                                              0.005, 0.9, 0.999, 1e-8, ts))
        if ts - lz["flushed"] >= 32:
            ctx.check(ctx.lib.dae_adam_rows_flush(ctx.h, P(t["We"]), P(mom["We"][0]), P(mom["We"][1]), P(lz["state"]), P(lz["tab"]),
                                                  lz["tab"].numel(), V, H, 0.9, 0.999, 1e-8, ts))
            lz["flushed"] = ts
        for n in ("be", "bd"):
            ctx.check(ctx.lib.dae_adam_step(ctx.h, P(t[n]), P(mom[n][0]), P(mom[n][1]), P(g[n]), t[n].numel(), 0.005, 0.9, 0.999, 1e-8, ts))
        return
    ctx.check(ctx.lib.dae_train_forward_backward(ctx.h, P(t["xr"]), P(t["xc"]), P(t["xv"]), P(t["yr"]), P(t["yc"]), P(t["yv"]),
        P(t["We"]), P(t["be"]), P(t["Wd"]), P(t["bd"]), V, H, B, B, 1 if tied else 0, 0.75, 0.8, 100 + i, 0.0,
        P(g["We"]), P(g["be"]), None if tied else P(g["Wd"]), P(g["bd"]), P(cost)))
    for n in names:
        ctx.check(ctx.lib.dae_adam_step(ctx.h, P(t[n]), P(mom[n][0]), P(mom[n][1]), P(g[n]), t[n].numel(), 0.005, 0.9, 0.999, 1e-8, i + 1))
for i in range(3): step(i)
torch.cuda.synchronize(); costs = []
t0 = time.perf_counter(); K = 32 if default else 20
for i in range(K):
    step(3 + i)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
flop = 3 * 2.0 * B * V * H
print(json.dumps({"what": "training step (%s%s), fwd+loss+bwd+Adam" % ("tied" if tied else "untied", ", bf16 GEMMs" if "--bf16" in sys.argv else ""), "ms_per_step": round(ms, 3),
                  "playlists_per_s": round(B / ms * 1e3, 1), "gemm_tflops_incl_everything": round(flop / ms / 1e9, 1),
                  "cost_first_last": [float(cost.item())]}))

# ---- what a rank of an N-GPU vocabulary-row sharded training job would run (sharding.ShardedTrainer), on ONE GPU:
# the three stages on shard 0 of N, the two [B,H] all-reduces replaced by 1-rank copies (communication NOT included)
if "--sim-world" in sys.argv:
    from spotify_recsys_challenge_2018_amd.sharding import HipTrainStages, ShardedTrainer, shard_bounds
    N = int(sys.argv[sys.argv.index("--sim-world") + 1])
    lo, hi = shard_bounds(V, N, 0)
    full = [W_enc, W_dec, b_enc, b_dec]
    sub = [W_enc[lo:hi], W_dec[lo:hi], b_enc, b_dec[lo:hi]]
    tr = ShardedTrainer(full, B, 0.005, 0.0, tied, HipTrainStages(ctx), device="cuda", rank=0, world=1)
    # shrink the trainer to the shard (world stays 1 so that no process group is needed)
    tr.lo, tr.hi = lo, hi
    for n_, a_ in (("W_enc", sub[0]), ("W_dec", sub[1]), ("b_dec", sub[3])):
        if n_ == "W_dec" and tied:
            continue
        tt = dev(np.ascontiguousarray(a_))
        setattr(tr, n_, tt)
        tr.params[n_] = tt
        tr.grads[n_] = torch.zeros_like(tt)
        tr.moments[n_] = (torch.zeros_like(tt), torch.zeros_like(tt))
    if tied:
        tr.W_dec = tr.W_enc
    x = (t["xr"], t["xc"], t["xv"]); y = (t["yr"], t["yc"], t["yv"])
    for _ in range(3):
        tr.train_step(x, y, 0.8, 0.75)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        tr.train_step(x, y, 0.8, 0.75)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / K * 1e3
    print(json.dumps({"what": "per-rank compute of a %d-GPU row-sharded training step (%s), shard [%d,%d), no communication" % (N, "tied" if tied else "untied", lo, hi),
                      "ms_per_step": round(ms2, 3), "playlists_per_s_if_comm_free": round(B / ms2 * 1e3, 1),
                      "vs_one_gpu_step": round(ms / ms2, 2)}))
