#!/usr/bin/env python
"""Odd shapes at the full vocabulary (V = 170 000): the fused scoring path against decode_dense + topk_dense through the
C ABI, fp32 bit for bit, bf16 fused against bf16 unfused.  Not part of the test suite (about a minute of GPU time)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib                                # noqa: E402
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr   # noqa: E402
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights   # noqa: E402


def main():
    import torch
    V, nt, H = 170000, 140000, 256
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()              # noqa: E731
    ctx = _lib.Context(0)
    bad = n = 0
    for bias in ("zipf", "zeros"):
        W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=3, bias=bias, n_tracks=nt)
        d_We, d_be, d_Wd, d_bd = dev(W_enc), dev(b_enc), dev(W_dec), dev(b_dec)
        for dtype in (_lib.DAE_DTYPE_F32, _lib.DAE_DTYPE_BF16):
            ctx.prepack_decoder(d_Wd, d_bd, 0, V, dtype)
            for B, k in ((1, 500), (37, 1), (150, 500), (257, 1024), (750, 500), (1000, 7), (1025, 500), (2049, 100)):
                pos, ones, seeds = make_playlists(B, nt, V - nt, seed=B + k)
                if B > 3:
                    seeds[2] = list(range(0, nt, 3))                              # a row that seeds a third of the tracks
                    seeds[3] = []
                rp, col, val = coo_to_csr(pos, ones, B, V)
                srp, sc = seeds_to_csr(seeds, B, nt)
                d = [dev(a) for a in (rp, col, val, srp, sc if sc.size else np.zeros(1, np.int32))]
                s = torch.empty((B, k), device="cuda"); i = torch.empty((B, k), dtype=torch.int32, device="cuda")
                ctx.score_topk(d[0], d[1], d[2], d_We, d_be, nt, d[3], d[4], k, s, i, dtype=dtype)
                h = torch.empty((B, H), device="cuda"); ctx.encode(d[0], d[1], d[2], d_We, d_be, h)
                z = torch.empty((B, V), device="cuda"); ctx.decode_dense(h, z, apply_sigmoid=False, dtype=dtype)
                su = torch.empty_like(s); iu = torch.empty_like(i)
                ctx.topk_dense(z, nt, 0, d[3], d[4], k, su, iu)
                ok = torch.equal(i, iu) and torch.equal(s, su)
                n += 1
                if not ok:
                    bad += 1
                    print("MISMATCH bias=%s dtype=%d B=%d k=%d: %d index rows differ" % (
                        bias, dtype, B, k, int((i != iu).any(dim=1).sum())))
    print("full-size sweep: %d cases, %d bad" % (n, bad))


if __name__ == "__main__":
    main()
