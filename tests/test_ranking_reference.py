"""The ranking half of the hot path against outputs of the REAL reference function
`main_runner/main_challenge.py:26-41 cand_generate` (tests/golden/expected_ranking.npz, written by
tests/golden/make_golden.py importing it in the build container -- numpy argsort of the negated scores,
list.remove per seed, [:500], id -> 'spotify:track:<uri>').

numpy's argsort order among EQUAL scores is unspecified, so the reference answer is one of several valid ones
whenever ties cross rank 500: rows are compared exactly where the reference answer is unique (as a set: no tie
between the last kept and the first dropped score; as a sequence: no tie inside the kept range either), and
"valid under the rule" (oracle.dae_numpy.topk_valid_under_reference_rule) + score-sequence-identical otherwise.

CPU: the oracle's orc_topk (what every GPU parity test is checked against) and the product's URI formatting.
GPU (-m gpu): dae_topk_dense(DAE_OUT_SCORE) through the C ABI on the same rows."""
import os

import numpy as np
import pytest

import oracle
from oracle import dae_numpy as dn

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 500


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(G, "expected_ranking.npz"))
    rows = []
    for r in range(z["scores"].shape[0]):
        rows.append((z["scores"][r], z["seed_flat"][z["seed_ptr"][r]:z["seed_ptr"][r + 1]].tolist(),
                     z["expected_ids"][r], str(z["kinds"][r])))
    return z, rows


def _seed_csr(seeds_per_row, n_cols):
    from spotify_recsys_challenge_2018_amd.models.DAEs import seeds_to_csr
    return seeds_to_csr(seeds_per_row, len(seeds_per_row), n_cols)


def _check_row(y, seeds, ref_ids, got_idx, got_score):
    """got_* = this build's answer for the row; ref_ids = what the reference function returned."""
    ref = [int(i) for i in ref_ids if i >= 0]
    got = [int(i) for i in got_idx if i >= 0]
    assert len(got) == len(ref)
    ok, boundary_tie = dn.topk_valid_under_reference_rule(y, seeds, got, K)
    assert ok
    # the reference's answer is itself valid under the rule (sanity of the checker against the real function)
    assert dn.topk_valid_under_reference_rule(y, seeds, ref, K)[0]
    # same score at every rank, whatever the tie order
    assert np.array_equal(y[got], y[ref])
    if not boundary_tie:
        assert set(got) == set(ref)
    if len(np.unique(y[ref])) == len(ref) and not boundary_tie:
        assert got == ref                                       # unique answer: same sequence
    # this build's tie rule: equal scores in ascending column order
    g = np.asarray(got)
    same = y[g][1:] == y[g][:-1]
    assert np.all(g[1:][same] > g[:-1][same])
    assert np.array_equal(np.asarray(got_score[:len(got)], np.float32), y[got])
    return boundary_tie


def test_fixture_covers_the_cases(fx):
    z, rows = fx
    kinds = {k for _, _, _, k in rows}
    assert kinds == {"plain", "plateau_top", "ties_everywhere", "distinct"}
    assert any(len(s) != len(set(s)) for _, s, _, _ in rows)            # duplicated seeds
    assert any(max(s, default=0) >= z["scores"].shape[1] for _, s, _, _ in rows)   # seed ids beyond the columns
    assert any(np.sum(y == np.float32(1.0)) > 30 for y, _, _, _ in rows)           # saturated plateau
    assert any(len(s) == 0 for _, s, _, _ in rows)                      # 0-seed rows


def test_oracle_topk_against_the_reference_function(fx):
    z, rows = fx
    n = z["scores"].shape[1]
    srp, sc = _seed_csr([s for _, s, _, _ in rows], n)
    # the oracle ranks fp32 keys; handing it the reference's SCORES as "logits" ranks exactly those values
    s_o, i_o = oracle.topk(z["scores"], K, srp, sc, out_kind=1)
    unique_rows = 0
    for r, (y, seeds, ref_ids, _kind) in enumerate(rows):
        tie = _check_row(y, seeds, ref_ids, i_o[r], s_o[r])
        unique_rows += (not tie)
    assert unique_rows >= len(rows) // 2
    # fewer rankable columns than k: cand[:500] of a short list
    ys = z["short_scores"]
    srp, sc = _seed_csr([z["short_seeds"].tolist()], ys.size)
    s1, i1 = oracle.topk(ys[None, :], K, srp, sc, out_kind=1)
    _check_row(ys, z["short_seeds"].tolist(), z["short_expected"], i1[0], s1[0])
    assert np.all(i1[0, z["short_expected"].size:] == -1)


def test_product_uri_formatting_matches_reference_strings(fx):
    from spotify_recsys_challenge_2018_amd.main_runner.main_challenge import cand_to_uris
    z, rows = fx
    id2uri = {str(i): "U%06d" % i for i in range(z["scores"].shape[1])}
    assert cand_to_uris(rows[0][2], id2uri) == [str(u) for u in z["uris_row0"]]
    assert cand_to_uris(np.array([3, -1, -1]), id2uri) == ["spotify:track:U000003"]


@pytest.mark.gpu
def test_gpu_topk_dense_against_the_reference_function(fx):
    import torch
    from spotify_recsys_challenge_2018_amd import _lib
    z, rows = fx
    ctx = _lib.Context(0)
    n = z["scores"].shape[1]

    def run(scores, seeds_per_row):
        srp, sc = _seed_csr(seeds_per_row, scores.shape[1])
        d_y = torch.from_numpy(np.ascontiguousarray(scores)).cuda()
        d_srp = torch.from_numpy(srp).cuda()
        d_sc = torch.from_numpy(sc if sc.size else np.zeros(1, np.int32)).cuda()
        B = scores.shape[0]
        s = torch.empty((B, K), device="cuda"); i = torch.empty((B, K), dtype=torch.int32, device="cuda")
        # the values ARE the reference's y_pred: rank them as they are (no sigmoid on the way out)
        ctx.topk_dense(d_y, scores.shape[1], 0, d_srp, d_sc, K, s, i, out_kind=_lib.DAE_OUT_LOGIT)
        return s.cpu().numpy(), i.cpu().numpy()
    s_g, i_g = run(z["scores"], [s for _, s, _, _ in rows])
    for r, (y, seeds, ref_ids, _kind) in enumerate(rows):
        _check_row(y, seeds, ref_ids, i_g[r], s_g[r])
    # and bit-identical to the oracle (the parity bar of every other GPU test)
    srp, sc = _seed_csr([s for _, s, _, _ in rows], n)
    s_o, i_o = oracle.topk(z["scores"], K, srp, sc, out_kind=1)
    assert np.array_equal(i_g, i_o) and np.array_equal(s_g.view(np.uint32), s_o.view(np.uint32))
    ys = z["short_scores"]
    s1, i1 = run(ys[None, :], [z["short_seeds"].tolist()])
    _check_row(ys, z["short_seeds"].tolist(), z["short_expected"], i1[0], s1[0])
    ctx.close()
