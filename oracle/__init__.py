"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/liboracle.so (dae_oracle.c).

PARITY UNPINNED at the TensorFlow boundary (see dae_oracle.c).  Import allowed only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package must never import it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dae_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_sigmoidf.restype = ctypes.c_float
        _LIB.orc_sigmoidf.argtypes = [ctypes.c_float]
        _LIB.orc_uniform.restype = ctypes.c_float
        _LIB.orc_uniform.argtypes = [ctypes.c_uint32] * 4
        _LIB.orc_okey.restype = ctypes.c_uint32
        _LIB.orc_okey.argtypes = [ctypes.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sigmoid(x):
    x = _f32(x)
    out = np.empty_like(x)
    l = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = l.orc_sigmoidf(float(flat_in[i]))
    return out


def encode(row_ptr, col, val, W_enc, b_enc, ikp=1.0, kp=1.0, seed=0):
    row_ptr, col, val = _i32(row_ptr), _i32(col), _f32(val)
    W_enc, b_enc = _f32(W_enc), _f32(b_enc)
    V, H = W_enc.shape
    B = row_ptr.size - 1
    h = np.empty((B, H), dtype=np.float32)
    lib().orc_encode(_p(row_ptr), _p(col), _p(val), _p(W_enc), _p(b_enc),
                     ctypes.c_int(V), ctypes.c_int(H), ctypes.c_int(B),
                     ctypes.c_float(ikp), ctypes.c_float(kp), ctypes.c_uint32(seed), _p(h))
    return h


def decode(h, W_dec, b_dec, col_lo=0, col_hi=None, apply_sigmoid=False, bf16=False):
    h, W_dec, b_dec = _f32(h), _f32(W_dec), _f32(b_dec)
    B, H = h.shape
    if col_hi is None:
        col_hi = W_dec.shape[0]
    n = col_hi - col_lo
    out = np.empty((B, n), dtype=np.float32)
    fn = lib().orc_decode_bf16 if bf16 else lib().orc_decode
    fn(_p(h), _p(W_dec), _p(b_dec), ctypes.c_int(H), ctypes.c_int(B),
       ctypes.c_int(col_lo), ctypes.c_int(col_hi), ctypes.c_int(1 if apply_sigmoid else 0),
       _p(out), ctypes.c_int64(n))
    return out


def topk(logits, k, seed_row_ptr=None, seed_col=None, col_base=0, ncols=None, out_kind=0):
    logits = _f32(logits)
    B, ld = logits.shape
    if ncols is None:
        ncols = ld
    srp = _i32(seed_row_ptr) if seed_row_ptr is not None else None
    sc = _i32(seed_col) if seed_col is not None else None
    if sc is not None and sc.size == 0:
        sc = np.zeros(1, dtype=np.int32)
    score = np.empty((B, k), dtype=np.float32)
    idx = np.empty((B, k), dtype=np.int32)
    lib().orc_topk(_p(logits), ctypes.c_int64(ld), ctypes.c_int(B), ctypes.c_int(ncols),
                   ctypes.c_int(col_base), _p(srp), _p(sc) if srp is not None else None,
                   ctypes.c_int(k), ctypes.c_int(out_kind), _p(score), _p(idx))
    return score, idx


def topk_merge(cand_logit, cand_idx, out_kind=0):
    cand_logit, cand_idx = _f32(cand_logit), _i32(cand_idx)
    G, B, k = cand_logit.shape
    score = np.empty((B, k), dtype=np.float32)
    idx = np.empty((B, k), dtype=np.int32)
    lib().orc_topk_merge(ctypes.c_int(G), ctypes.c_int(B), ctypes.c_int(k),
                         _p(cand_logit), _p(cand_idx), ctypes.c_int(out_kind), _p(score), _p(idx))
    return score, idx


def score_batch(row_ptr, col, val, W_enc, b_enc, W_dec, b_dec, n_cols_decoded, n_tracks,
                seed_row_ptr, seed_col, k):
    """encode -> decode -> rank on one thread (bench.py cpu_baseline 'port', smoke())."""
    row_ptr, col, val = _i32(row_ptr), _i32(col), _f32(val)
    W_enc, b_enc, W_dec, b_dec = _f32(W_enc), _f32(b_enc), _f32(W_dec), _f32(b_dec)
    srp, sc = _i32(seed_row_ptr), _i32(seed_col)
    if sc.size == 0:
        sc = np.zeros(1, dtype=np.int32)
    V, H = W_enc.shape
    B = row_ptr.size - 1
    score = np.empty((B, k), dtype=np.float32)
    idx = np.empty((B, k), dtype=np.int32)
    lib().orc_score_batch(_p(row_ptr), _p(col), _p(val), _p(W_enc), _p(b_enc), _p(W_dec),
                          _p(b_dec), ctypes.c_int(V), ctypes.c_int(H), ctypes.c_int(B),
                          ctypes.c_int(n_cols_decoded), ctypes.c_int(n_tracks), _p(srp), _p(sc),
                          ctypes.c_int(k), _p(score), _p(idx))
    return score, idx
