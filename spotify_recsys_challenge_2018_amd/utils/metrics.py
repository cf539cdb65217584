"""Ranking metrics of the reference's utils/metrics.py, kept as the drop-in contract
(north_star: "utils/metrics.py r-precision").  Citations relative to /root/reference.

Differences from the snapshot: `get_metrics`/`single_eval` there unpack three
values from a scalar and always raise; here they return the r-precision, which is what the
training driver accumulates (main_train.py:89-100).  The ranking inside `single_eval` is NOT done
on the host: dense scores are ranked by the HIP top-k kernel (dae_topk_dense) with the canonical
tie rule, and the drivers of this repo never build dense scores at all (they call
`model.recommend`).
"""
import math

import numpy as np


def get_r_precision(answer, cand, answer_cls=None, class_divpnt=None):
    """metrics.py:20-27: |set(answer) & set(cand[:len(answer)])| / len(answer).
    `answer` may contain -1 (tracks outside the vocabulary, spotify_reader.py:271-273): they can
    never be hit but do count in the denominator.  The two class arguments are accepted and
    ignored, as in the reference."""
    n = len(answer)
    if n == 0:
        raise ZeroDivisionError("empty answer list")
    return len(set(answer) & set(cand[:n])) / n


def get_ndcg(answer, cand):
    """metrics.py:29-42 (NB: the reference's IDCG grows with the number of HITS, not with
    len(answer); kept as is)."""
    dcg = 1.0 if cand[0] in answer else 0.0
    idcg, next_ideal = 1.0, 2
    for pos in range(1, len(cand)):
        if cand[pos] in answer:
            dcg += 1.0 / math.log(pos + 1, 2)
            idcg += 1.0 / math.log(next_ideal, 2)
            next_ideal += 1
    return dcg / idcg


def get_rsc(answer, cand):
    """metrics.py:44-49: recommended-songs clicks = index of the first hit // 10, 51 if none."""
    for pos, c in enumerate(cand):
        if c in answer:
            return pos // 10
    return 51


def get_metrics(answer, cand, answer_cls=None, num_cls=None):
    """metrics.py:51-56, repaired: r-precision only (ndcg / rsc are commented out upstream)."""
    return get_r_precision(answer, cand, answer_cls, num_cls)


_ctx = None


def _rank_dense_on_gpu(scores, seed, k):
    """metrics.py:59-68 ranking (argsort desc, remove seeds, first 500) on the GPU."""
    import torch

    from .. import _lib
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(0)
    _ctx.bind_stream()
    s = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32).reshape(1, -1)).cuda()
    n = s.shape[1]
    sd = np.unique(np.asarray([x for x in seed if 0 <= x < n], dtype=np.int32))
    srp = torch.tensor([0, sd.size], dtype=torch.int32, device="cuda")
    sc = torch.from_numpy(sd if sd.size else np.zeros(1, np.int32)).cuda()
    out_s = torch.empty((1, k), dtype=torch.float32, device="cuda")
    out_i = torch.empty((1, k), dtype=torch.int32, device="cuda")
    # scores are already sigmoid outputs: rank them as they are (DAE_OUT_LOGIT = raw values)
    _ctx.topk_dense(s, n, 0, srp, sc, k, out_s, out_i, out_kind=_lib.DAE_OUT_LOGIT)
    idx = out_i.cpu().numpy()[0]
    return [int(i) for i in idx if i >= 0]


def single_eval(scores, seed, answer, answer_cls=None, num_cls=None, k=500):
    """metrics.py:58-70, repaired: rank one row of track scores, drop the seeds, keep 500, return
    the r-precision."""
    cand = _rank_dense_on_gpu(np.asarray(scores), seed, k)
    return get_metrics(answer, cand, answer_cls, num_cls)


def eval_topk(cand_idx, answer):
    """r-precision of an already ranked candidate row (output of model.recommend)."""
    cand = [int(i) for i in cand_idx if i >= 0]
    return get_r_precision(answer, cand)
