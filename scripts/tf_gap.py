"""How much of "parity unpinned at the TensorFlow boundary" can matter for the RANKING?  

TensorFlow 1.x is not installable here, so nobody can run the reference's encode / decode (models/DAEs.py:64-77,
:141-145) and compare bits.  What can be computed is whether the reference's top-500 SET could differ from ours at
all: TF evaluates the same dot products in fp32 with SOME summation order (Eigen's blocked GEMM), and any fp32
evaluation of a length-n dot product sum_k a_k b_k is within gamma_n * sum_k |a_k b_k| of the exact value
(gamma_n = n u / (1 - n u), u = 2^-24; Higham, Accuracy and Stability of Numerical Algorithms, eq. 3.5).

Per row this script computes, in float64,
    z_c      exact logits from the oracle's fp32 hidden activations,
    E_c      = gamma_(H+1) (sum_k |h_k| |W_dec[c,k]| + |b_c|)           any-order fp32 error of the decoder
             + dh * sum_k |W_dec[c,k]|                                  what an any-order ENCODER can move h by:
               dh = 1/4 gamma_(nnz+1) sum |x^| |W_enc| + 4 u            (sigmoid' <= 1/4; 4 ulp for the sigmoid itself)
and calls the row ORDER-INDEPENDENT when   min over the top-500 of (z - E)  >  max over the rest of (z + E)   (rankable,
non-seed columns): then EVERY fp32 evaluation -- TF's included -- selects the same 500 tracks, and only their order
inside the list can differ.  Otherwise it counts the columns whose interval [z - E, z + E] straddles the cut.

    python scripts/tf_gap.py                 full size (V = 170 000, B = 256; ~1 min), writes profiles/r03_tf_gap.json
    python scripts/tf_gap.py --small         the size tests/test_tf_gap_cpu.py runs

TEST INFRASTRUCTURE: imports the oracle; nothing in the product imports this.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

U = 2.0 ** -24


def gamma(n):
    return n * U / (1.0 - n * U)


def analyse(W_enc, b_enc, W_dec, b_dec, rp, col, val, srp, sc, n_tracks, k=500):
    """-> dict of per-batch statistics (see the module docstring)."""
    import oracle
    B = len(rp) - 1
    H = W_enc.shape[1]
    h = oracle.encode(rp, col, val, W_enc, b_enc).astype(np.float64)
    Wd = W_dec[:n_tracks].astype(np.float64)
    bd = b_dec[:n_tracks].astype(np.float64)
    z = h @ Wd.T + bd[None, :]
    absW = np.abs(Wd)
    n_c = absW.sum(1)
    E_dec = gamma(H + 1) * (np.abs(h) @ absW.T + np.abs(bd)[None, :])
    # encoder: x^ = val / sum(val) per row (DAEs.py:41-42), a = sum x^ W_enc
    dh = np.zeros(B)
    absWe = np.abs(W_enc.astype(np.float64))
    for r in range(B):
        c_, v_ = col[rp[r]:rp[r + 1]], val[rp[r]:rp[r + 1]].astype(np.float64)
        if len(c_):
            xh = v_ / (v_.sum() + 1e-10)
            dh[r] = 0.25 * gamma(len(c_) + 1) * float(np.max(xh @ absWe[c_])) + 4 * U
        else:
            dh[r] = 4 * U
    E = E_dec + dh[:, None] * n_c[None, :]
    indep, ambiguous, gaps, margins = 0, [], [], []
    for r in range(B):
        ok = np.ones(n_tracks, bool)
        ok[sc[srp[r]:srp[r + 1]]] = False
        zr, er = z[r][ok], E[r][ok]
        if zr.size <= k:
            indep += 1
            continue
        order = np.argsort(-zr, kind="stable")
        top, rest = order[:k], order[k:]
        lo_top = np.min(zr[top] - er[top])
        hi_rest = np.max(zr[rest] + er[rest])
        gaps.append(float(zr[order[k - 1]] - zr[order[k]]))
        margins.append(float(lo_top - hi_rest))
        if lo_top > hi_rest:
            indep += 1
        else:
            # columns that could be on either side of the cut under some fp32 evaluation
            cut_lo, cut_hi = min(lo_top, hi_rest), max(lo_top, hi_rest)
            amb = int(np.sum((zr + er >= cut_lo) & (zr - er <= cut_hi)))
            ambiguous.append(amb)
    return {
        "rows": B, "k": k,
        "rows_order_independent": indep,
        "fraction_order_independent": round(indep / B, 4),
        "median_gap_at_cut": float(np.median(gaps)) if gaps else None,
        "median_error_bound_at_cut": float(np.median(E)),
        "ambiguous_columns_per_flagged_row": {"median": float(np.median(ambiguous)) if ambiguous else 0,
                                              "max": int(max(ambiguous)) if ambiguous else 0},
        "saturated_scores": int(np.sum(z > 17.0)),       # sigmoid(z) == 1.0f: the reference would rank these by argsort's tie order
    }


def problem(V, nt, H, B, bias, scale=1.0, seed=0):
    from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
    from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=seed, bias=bias, n_tracks=nt)
    W_dec = (W_dec * scale).astype(np.float32)
    W_enc = (W_enc * scale).astype(np.float32)
    pos, ones, seeds = make_playlists(B, nt, V - nt, seed=seed + 1)
    rp, col, val = coo_to_csr(pos, ones, B, V)
    srp, sc = seeds_to_csr(seeds, B, nt)
    return W_enc, b_enc, W_dec, b_dec, rp, col, val, srp, sc


def run(V, nt, H, B):
    out = {}
    for name, bias, scale in (("bench_model_zipf_bias", "zipf", 1.0), ("zero_bias", "zeros", 1.0),
                              ("weights_x40_zipf_bias", "zipf", 40.0)):
        p = problem(V, nt, H, B, bias, scale)
        out[name] = analyse(*p, nt)
        out[name]["model"] = "V=%d n_tracks=%d H=%d, b_dec=%s, weights x%g (synthetic: utils/synthetic.py)" % (V, nt, H, bias, scale)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_tf_gap.json"))
    a = ap.parse_args()
    V, nt, H, B = (20000, 16000, 64, 48) if a.small else (170000, 140000, 256, 256)
    res = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "results": run(V, nt, H, B)}
    if not a.small:
        json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res["results"], indent=1))


if __name__ == "__main__":
    main()
