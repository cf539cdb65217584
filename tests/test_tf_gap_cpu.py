"""CPU: scripts/tf_gap.py -- the part of "parity unpinned at the TensorFlow boundary" that can reach the ranking.
For every row the script decides whether ANY fp32 evaluation of the reference's dot products (TensorFlow's included)
must select the same 500 tracks; this test runs it at a small size, checks its arithmetic on a hand-made row, and
checks that the committed full-size report (profiles/r03_tf_gap.json, quoted in DESIGN.md section 1) is present."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import tf_gap  # noqa: E402


def test_report_on_a_small_model():
    res = tf_gap.run(6000, 5000, 32, 16)
    for name in ("bench_model_zipf_bias", "zero_bias", "weights_x40_zipf_bias"):
        r = res[name]
        assert r["rows"] == 16 and 0 <= r["rows_order_independent"] <= 16
        assert r["median_error_bound_at_cut"] > 0
    # a popularity prior separates neighbouring ranks by far more than fp32 can move a logit
    assert res["bench_model_zipf_bias"]["fraction_order_independent"] >= 0.9
    assert res["bench_model_zipf_bias"]["median_gap_at_cut"] > 10 * res["bench_model_zipf_bias"]["median_error_bound_at_cut"]


def test_the_criterion_flags_a_row_whose_cut_is_closer_than_the_bound():
    """Two decoder rows that differ by one ulp in one weight sit exactly at ranks k and k + 1: the row is order-dependent;
    move them apart and it is not."""
    from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
    rng = np.random.default_rng(0)
    V = nt = 64; H = 16; k = 8
    W_enc = (rng.standard_normal((V, H)) * 0.1).astype(np.float32)
    b_enc = np.zeros(H, np.float32)
    W_dec = (rng.standard_normal((V, H)) * 1e-3).astype(np.float32)      # the bias decides the order
    b_dec = np.linspace(3.0, -3.0, V).astype(np.float32)          # ranks 0..63 far apart
    W_dec[k] = W_dec[k - 1]
    b_dec[k] = b_dec[k - 1]
    W_dec[k, 3] = np.nextafter(W_dec[k, 3], np.float32(-10.0))  # rank k just below rank k - 1 ... by one ulp of a weight
    pos = np.array([[0, 1], [0, 5]], np.int64)
    rp, col, val = coo_to_csr(pos, np.ones(2, np.float32), 1, V)
    srp, sc = seeds_to_csr([[]], 1, nt)
    r = tf_gap.analyse(W_enc, b_enc, W_dec, b_dec, rp, col, val, srp, sc, nt, k=k)
    assert r["rows_order_independent"] == 0 and r["ambiguous_columns_per_flagged_row"]["max"] == 2
    b_dec[k] -= 0.05
    r = tf_gap.analyse(W_enc, b_enc, W_dec, b_dec, rp, col, val, srp, sc, nt, k=k)
    assert r["rows_order_independent"] == 1


def test_committed_full_size_report():
    rep = json.load(open(os.path.join(ROOT, "profiles", "r03_tf_gap.json")))["results"]
    for name in ("bench_model_zipf_bias", "zero_bias", "weights_x40_zipf_bias"):
        assert rep[name]["rows"] == 256 and "V=170000" in rep[name]["model"]
        assert 0.0 <= rep[name]["fraction_order_independent"] <= 1.0
