"""ctypes binding of libdae_hip.so (C ABI: include/dae_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdae_hip.so")

DAE_OUT_SCORE, DAE_OUT_LOGIT = 0, 1
DAE_DTYPE_F32, DAE_DTYPE_BF16, DAE_DTYPE_BF16_EXACT = 0, 1, 2

# every symbol include/dae_hip.h declares (tests/test_abi_cpu.py checks the .so exports all of them)
EXPORTS = [
    "dae_version", "dae_create", "dae_destroy", "dae_set_stream", "dae_last_error",
    "dae_scratch_bytes", "dae_profile_enable", "dae_profile_read", "dae_profile_kernel", "dae_clock_probe", "dae_last_plan",
    "dae_coo_to_csr", "dae_seeds_from_csr", "dae_encode", "dae_prepack_decoder", "dae_prepack_decoder_rows", "dae_share_decoder", "dae_exact_bounds",
    "dae_exact_guard_read", "dae_exact_guard_words", "dae_exact_guard_snapshot", "dae_exact_stats_read", "dae_set_exact_margin", "dae_set_exact_margin_range", "dae_set_exact_audit", "dae_exact_audit_read", "dae_decode_dense", "dae_decode_topk",
    "dae_score_topk", "dae_score_topk_begin", "dae_score_topk_finish", "dae_topk_dense", "dae_topk_merge", "dae_set_train_dtype", "dae_train_forward_backward",
    "dae_train_shard_encode", "dae_train_shard_decode", "dae_train_shard_finish", "dae_title_features", "dae_title_prepack_features",
    "dae_mix_scores", "dae_decode_mix_term", "dae_set_score_mix", "dae_mix_topk_exact", "dae_title_score_exact", "dae_title_score", "dae_row_sums", "dae_mix_weights", "dae_title_loss_backward", "dae_title_conv_backward", "dae_adam_step",
    "dae_adam_rows_begin", "dae_adam_rows_apply", "dae_adam_rows_flush", "dae_set_enc_grad_prezeroed",
    "dae_arm_decoder_adam", "dae_set_decode_gate", "dae_set_overlap_hint",
    "dae_pipeline_create", "dae_pipeline_create_titled", "dae_pipeline_destroy", "dae_pipeline_submit", "dae_pipeline_submit_titled", "dae_pipeline_flush", "dae_pipeline_poll",
    "dae_pipeline_release", "dae_pipeline_stats", "dae_pipeline_exact_margin", "dae_pipeline_times", "dae_pipeline_last_error",
]
DAE_PIPE_BUSY = 1

_lib = None


class DaeError(RuntimeError):
    pass


def load():
    """Load libdae_hip.so; raise loudly if it has not been built (python -m <pkg>.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DaeError(
            "libdae_hip.so not found at %s -- build it with "
            "`python -m spotify_recsys_challenge_2018_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the scoring path." % LIB_PATH)
    # torch must come first: it brings its own libamdhip64; loading ours before it puts two HIP
    # runtimes in the process and the second one finds no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_f, c_u32, vp = (ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                    ctypes.c_uint32, ctypes.c_void_p)
    lib.dae_version.restype = c_int
    lib.dae_create.argtypes = [c_int, ctypes.POINTER(vp)]
    lib.dae_destroy.argtypes = [vp]
    lib.dae_set_stream.argtypes = [vp, vp]
    lib.dae_last_error.argtypes = [vp]
    lib.dae_last_error.restype = ctypes.c_char_p
    lib.dae_scratch_bytes.argtypes = [vp]
    lib.dae_scratch_bytes.restype = ctypes.c_size_t
    lib.dae_profile_enable.argtypes = [vp, c_int]
    lib.dae_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]
    lib.dae_profile_kernel.argtypes = [vp]
    lib.dae_profile_kernel.restype = ctypes.c_char_p
    lib.dae_clock_probe.argtypes = [vp, vp, c_int, vp, ctypes.POINTER(c_int)]
    lib.dae_last_plan.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    lib.dae_coo_to_csr.argtypes = [vp, vp, vp, c_int, c_i64, c_int, c_int, vp, vp, vp, vp]
    lib.dae_seeds_from_csr.argtypes = [vp, vp, vp, c_int, c_int, vp, vp]
    lib.dae_encode.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_f, c_f, c_u32, vp]
    lib.dae_prepack_decoder.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int]
    lib.dae_prepack_decoder_rows.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int]
    lib.dae_share_decoder.argtypes = [vp, vp, c_int]
    lib.dae_exact_bounds.argtypes = [vp, vp]
    lib.dae_exact_guard_read.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.dae_exact_guard_words.argtypes = [vp, ctypes.POINTER(vp)]
    lib.dae_exact_guard_snapshot.argtypes = [vp, vp]
    lib.dae_set_exact_margin.argtypes = [vp, c_f]
    lib.dae_exact_stats_read.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.dae_set_exact_margin_range.argtypes = [vp, c_int, c_int, c_f]
    lib.dae_set_exact_audit.argtypes = [vp, c_int, c_int]
    lib.dae_exact_audit_read.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.dae_decode_dense.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, c_i64]
    lib.dae_decode_topk.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, vp, c_int, c_int, vp, vp]
    lib.dae_score_topk.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp,
                                   c_int, c_int, vp, vp]
    lib.dae_score_topk_begin.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, c_int, vp]
    lib.dae_score_topk_finish.argtypes = [vp, vp, vp, vp, c_int, vp, vp]
    lib.dae_topk_dense.argtypes = [vp, vp, c_i64, c_int, c_int, c_int, vp, vp, c_int, c_int, vp, vp]
    lib.dae_topk_merge.argtypes = [vp, c_int, c_int, c_int, vp, vp, c_int, vp, vp]
    lib.dae_set_train_dtype.argtypes = [vp, c_int]
    lib.dae_set_train_dtype.restype = c_int
    lib.dae_train_forward_backward.argtypes = (
        [vp] + [vp] * 6 + [vp] * 4 + [c_int] * 5 + [c_f, c_f, c_u32, c_f] + [vp] * 5)
    lib.dae_train_shard_encode.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_f, c_u32, vp]
    lib.dae_train_shard_decode.argtypes = (
        [vp, vp, vp] + [vp] * 3 + [vp] * 3 + [c_int] * 6 + [c_f, c_u32, c_f] + [vp] * 4)
    lib.dae_train_shard_finish.argtypes = (
        [vp, vp] + [vp] * 3 + [vp] * 4 + [c_int] * 5 + [c_f, c_f, c_u32, c_f] + [vp] * 4)
    lib.dae_title_features.argtypes = [vp, vp, c_int, c_int, vp, c_int, c_int, vp, vp, ctypes.POINTER(ctypes.c_int32),
                                       c_int, c_int, c_f, c_u32, vp, c_i64, vp, vp]
    lib.dae_title_prepack_features.argtypes = [vp, vp, c_int, c_int, vp, ctypes.POINTER(ctypes.c_int32), c_int, c_int]
    lib.dae_mix_scores.argtypes = [vp, vp, c_i64, vp, c_i64, vp, vp, c_int, c_int]
    lib.dae_decode_mix_term.argtypes = [vp, vp, c_int, c_int, c_int, vp, c_int, vp, c_i64]
    lib.dae_set_score_mix.argtypes = [vp, vp, c_i64, c_int, vp]
    lib.dae_mix_topk_exact.argtypes = [vp, vp, vp, c_i64, vp, c_i64, c_int, vp, vp, c_int, vp, vp, c_int, vp, vp, vp]
    lib.dae_row_sums.argtypes = [vp, vp, vp, vp, c_int, c_f, c_u32, vp]
    lib.dae_title_score_exact.argtypes = [vp, vp, vp, vp, c_int, c_i64, c_int, c_int, vp, vp, c_int, vp, c_int, vp, c_int, c_int, vp, vp,
                                          ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_int, vp, c_int, c_int, vp, vp, vp, vp]
    lib.dae_mix_weights.argtypes = [vp, vp, vp, vp, c_int, c_f, c_u32, vp, vp, vp]
    lib.dae_title_loss_backward.argtypes = [vp, vp, c_i64, vp, c_i64, vp, vp, vp, vp, vp, c_int, c_int, c_int,
                                            vp, c_int, vp, vp, vp, vp, vp]
    lib.dae_title_conv_backward.argtypes = [vp, vp, c_int, c_int, vp, c_int, c_int, vp, ctypes.POINTER(ctypes.c_int32),
                                            c_int, c_int, vp, vp, vp, c_i64, c_f, c_u32, vp, vp, vp]
    lib.dae_adam_step.argtypes = [vp, vp, vp, vp, vp, c_i64, c_f, c_f, c_f, c_f, c_int]
    lib.dae_adam_rows_begin.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, c_int, c_f, c_f, c_f, c_int]
    lib.dae_adam_rows_apply.argtypes = [vp, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, c_int,
                                        c_f, c_f, c_f, c_f, c_int]
    lib.dae_adam_rows_flush.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_f, c_f, c_f, c_int]
    lib.dae_set_enc_grad_prezeroed.argtypes = [vp, c_int]
    lib.dae_arm_decoder_adam.argtypes = [vp, vp, vp, c_f, c_f, c_f, c_f, c_int]
    lib.dae_set_decode_gate.argtypes = [vp, vp, vp]
    lib.dae_set_overlap_hint.argtypes = [vp, c_int]
    u64p, i32pp, f32pp, ip = (ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                              ctypes.POINTER(ctypes.POINTER(ctypes.c_float)), ctypes.POINTER(c_int))
    lib.dae_pipeline_create.argtypes = [c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_int, c_int, c_int,
                                        ctypes.POINTER(vp)]
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.dae_pipeline_create_titled.argtypes = ([c_int, vp, vp, vp, vp, c_int, c_int, c_int,        # device, the DAE's arrays, V, H, n_tracks
                                                vp, c_int, c_int, vp, vp, i32p, c_int, c_int,       # emb, n_char, E, conv_w, conv_b, filter_sizes, n_sizes, F
                                                vp, vp, c_int, c_int] +                             # Output_WT, Output_b, ld_feat, L
                                               [c_int] * 3 + [c_i64] + [c_int] * 3 + [ctypes.POINTER(vp)])
    lib.dae_pipeline_submit_titled.argtypes = [vp, vp, vp, c_int, c_i64, c_int, vp, vp, u64p]
    lib.dae_title_score.argtypes = [vp, vp, c_int, vp, vp, c_int, c_i64, c_int, c_int, vp, vp, c_int, vp, c_int, vp, c_int, c_int, vp, vp,
                                    i32p, c_int, c_int, c_int, vp, c_int, c_int, vp, vp, vp, vp]
    lib.dae_pipeline_destroy.argtypes = [vp]
    lib.dae_pipeline_submit.argtypes = [vp, vp, vp, c_int, c_i64, c_int, u64p]
    lib.dae_pipeline_flush.argtypes = [vp]
    lib.dae_pipeline_poll.argtypes = [vp, c_int, u64p, i32pp, f32pp, ip, ip]
    lib.dae_pipeline_release.argtypes = [vp, c_int]
    lib.dae_pipeline_stats.argtypes = [vp, u64p]
    lib.dae_pipeline_exact_margin.argtypes = [vp, c_f]
    lib.dae_pipeline_times.argtypes = [vp, u64p]
    lib.dae_pipeline_last_error.argtypes = [vp]
    lib.dae_pipeline_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        if name not in ("dae_last_error", "dae_scratch_bytes", "dae_profile_kernel", "dae_pipeline_last_error"):
            getattr(lib, name).restype = c_int
    _lib = lib
    return lib


def _ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


class Context:
    """Owns one dae_ctx bound to a device + the caller's current torch stream."""

    def __init__(self, device_index=0):
        import torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise DaeError("no GPU visible: the DAE scoring path runs only on MI355X (gfx950)")
        h = ctypes.c_void_p()
        rc = self.lib.dae_create(int(device_index), ctypes.byref(h))
        if rc != 0:
            raise DaeError("dae_create failed (%d): %s"
                           % (rc, self.lib.dae_last_error(None).decode()))
        self.h = h
        self.device_index = int(device_index)
        self.bind_stream()

    def bind_stream(self):
        import torch
        s = torch.cuda.current_stream(self.device_index).cuda_stream
        self.check(self.lib.dae_set_stream(self.h, ctypes.c_void_p(s)))

    def check(self, rc):
        if rc != 0:
            raise DaeError("libdae_hip error %d: %s"
                           % (rc, self.lib.dae_last_error(self.h).decode()))

    def coo_to_csr(self, positions, values, n_rows, n_cols):
        """Device COO (feed order, duplicates: last wins) -> CSR.  positions: int64 [nnz,2] CUDA tensor,
        values: float32 [nnz] or [1].  Returns (row_ptr, col, val, status) CUDA tensors; col / val have
        room for nnz entries, row_ptr[n_rows] of them are valid.  status != 0: an index was out of range."""
        import torch
        nnz = int(positions.shape[0])
        dev = positions.device
        rp = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        bcast = 1 if (values.numel() == 1 and nnz != 1) else 0
        self.check(self.lib.dae_coo_to_csr(self.h, _ptr(positions), _ptr(values), bcast, nnz, int(n_rows),
                                           int(n_cols), _ptr(rp), _ptr(col), _ptr(val), _ptr(status)))
        return rp, col, val, status

    def seeds_from_csr(self, row_ptr, col, n_tracks):
        """Seed CSR = the track columns of the input CSR (seeds are the playlist's own tracks).  -> (seed_row_ptr, seed_col)."""
        import torch
        B = row_ptr.numel() - 1
        srp = torch.empty(B + 1, dtype=torch.int32, device=row_ptr.device)
        sc = torch.empty(max(int(col.numel()), 1), dtype=torch.int32, device=row_ptr.device)
        self.check(self.lib.dae_seeds_from_csr(self.h, _ptr(row_ptr), _ptr(col), B, int(n_tracks), _ptr(srp), _ptr(sc)))
        return srp, sc

    def set_train_dtype(self, dtype):
        """Arithmetic of the training forward GEMM: DAE_DTYPE_F32 (default) or DAE_DTYPE_BF16."""
        self.check(self.lib.dae_set_train_dtype(self.h, int(dtype)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.dae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin wrappers (torch CUDA tensors in, results written into caller tensors) ------------
    def encode(self, row_ptr, col, val, W_enc, b_enc, h_out, ikp=1.0, kp=1.0, seed=0):
        V, H = W_enc.shape
        B = row_ptr.numel() - 1
        self.check(self.lib.dae_encode(self.h, _ptr(row_ptr), _ptr(col), _ptr(val), _ptr(W_enc),
                                       _ptr(b_enc), V, H, B, float(ikp), float(kp),
                                       int(seed) & 0xFFFFFFFF, _ptr(h_out)))

    def prepack_decoder(self, W_dec, b_dec, col_lo=0, col_hi=None, dtype=DAE_DTYPE_F32):
        V, H = W_dec.shape
        if col_hi is None:
            col_hi = V
        self.check(self.lib.dae_prepack_decoder(self.h, _ptr(W_dec), _ptr(b_dec), V, H,
                                                int(col_lo), int(col_hi), int(dtype)))
        self._drop_share(dtype)

    @staticmethod
    def _slot(dtype):
        return "f32" if int(dtype) == DAE_DTYPE_F32 else "bf16"      # the two image slots of a context (dae_packed)

    def _drop_share(self, dtype):
        """The slot holds an image of its own again: the context it borrowed from need not be kept alive for it."""
        self.__dict__.setdefault("_shares", {}).pop(self._slot(dtype), None)

    def prepack_decoder_rows(self, W_rows, b_rows, col_lo, dtype=DAE_DTYPE_F32):
        """Prepack a decoder image from a RANK-LOCAL copy of its rows: W_rows [n, H] / b_rows [n] hold the global
        columns col_lo .. col_lo + n (dae_prepack_decoder_rows)."""
        n, H = W_rows.shape
        assert W_rows.is_contiguous() and b_rows.is_contiguous() and b_rows.numel() == n
        self.check(self.lib.dae_prepack_decoder_rows(self.h, _ptr(W_rows), _ptr(b_rows), int(n), H, int(col_lo), int(dtype)))
        self._drop_share(dtype)

    def set_overlap_hint(self, batches_in_flight):
        """Other batches are in flight on other streams: kernel shapes that share CUs (include/dae_hip.h)."""
        self.check(self.lib.dae_set_overlap_hint(self.h, int(batches_in_flight)))

    def share_decoder(self, src, dtype=DAE_DTYPE_F32):
        """Use `src`'s prepacked image of `dtype` instead of an own copy (see include/dae_hip.h dae_share_decoder: `src`
        outlives every use; order src's prepack before this context's first launch)."""
        self.check(self.lib.dae_share_decoder(self.h, src.h, int(dtype)))
        self.__dict__.setdefault("_shares", {})[self._slot(dtype)] = src     # one owner per slot, alive while borrowed

    def exact_bounds(self, eps_out):
        """Per-column bounds |fp32 logit - bf16 logit| <= eps of the image prepacked with DAE_DTYPE_BF16_EXACT."""
        self.check(self.lib.dae_exact_bounds(self.h, _ptr(eps_out)))

    def set_exact_margin(self, scale):
        """Factor on every eps_c at the NEXT exact prepack (include/dae_hip.h: > 1 widens, < 1 voids the bound -- the
        guard's test hook)."""
        self.check(self.lib.dae_set_exact_margin(self.h, float(scale)))
        self._exact_margin = float(scale)        # (models/DAEs.py hands it on to the pipelines it builds from this model)

    def set_exact_margin_range(self, col_from, col_to, scale):
        """The same for the columns [col_from, col_to) alone (the audit's test hook: a column every row drops)."""
        self.check(self.lib.dae_set_exact_margin_range(self.h, int(col_from), int(col_to), float(scale)))

    def set_exact_audit(self, every_n, n_tiles=16):
        """Every every_n-th exact scoring launch audits n_tiles random tiles of DROPPED columns (0: never; include/dae_hip.h)."""
        self.check(self.lib.dae_set_exact_audit(self.h, int(every_n), int(n_tiles)))

    def exact_audit_read(self):
        """{audits, checked, violations} of the dropped-column audit since the context was created (synchronises)."""
        a = (ctypes.c_uint64 * 3)()
        self.check(self.lib.dae_exact_audit_read(self.h, a))
        return {"audits": int(a[0]), "checked": int(a[1]), "violations": int(a[2])}

    def exact_guard_read(self):
        """(violations, column) of the exact mode's bound guard since the last non-zero read; synchronises the stream."""
        n, c = ctypes.c_int32(), ctypes.c_int32()
        self.check(self.lib.dae_exact_guard_read(self.h, ctypes.byref(n), ctypes.byref(c)))
        if n.value:
            self._guard_seen = 0             # (the read reset the cumulative words: snapshots compare against zero again)
        return int(n.value), int(c.value)

    def exact_guard_snapshot(self, words_out):
        """The cumulative guard words after everything enqueued so far -> words_out (CUDA int32[2]), in stream order."""
        self.check(self.lib.dae_exact_guard_snapshot(self.h, _ptr(words_out)))

    def guard_moved(self, n_bad):
        """A launch's snapshot against the previous one taken on this context: True when the count changed under it."""
        seen = getattr(self, "_guard_seen", 0)
        self._guard_seen = int(n_bad)
        return int(n_bad) != seen

    def exact_stats_read(self):
        """{rows, candidates_per_row, recomputed_per_row} of the refine launches since the last read (synchronises)."""
        a = (ctypes.c_uint64 * 3)()
        self.check(self.lib.dae_exact_stats_read(self.h, a))
        rows = max(int(a[0]), 1)
        return {"rows": int(a[0]), "candidates_per_row": round(int(a[1]) / rows, 1), "recomputed_per_row": round(int(a[2]) / rows, 1)}

    def exact_guard_words(self):
        """Device address of the guard's two int32 words (for a fetch alongside the results)."""
        p = ctypes.c_void_p()
        self.check(self.lib.dae_exact_guard_words(self.h, ctypes.byref(p)))
        return int(p.value)

    def decode_dense(self, h, out, apply_sigmoid=True, dtype=DAE_DTYPE_F32):
        B, H = h.shape
        self.check(self.lib.dae_decode_dense(self.h, _ptr(h), B, H, int(dtype),
                                             1 if apply_sigmoid else 0, _ptr(out),
                                             int(out.stride(0))))

    def decode_topk(self, h, n_tracks, seed_row_ptr, seed_col, k, out_score, out_idx,
                    out_kind=DAE_OUT_SCORE, dtype=DAE_DTYPE_F32):
        B, H = h.shape
        self.check(self.lib.dae_decode_topk(self.h, _ptr(h), B, H, int(dtype), int(n_tracks),
                                            _ptr(seed_row_ptr), _ptr(seed_col), int(k),
                                            int(out_kind), _ptr(out_score), _ptr(out_idx)))

    def score_topk(self, row_ptr, col, val, W_enc, b_enc, n_tracks, seed_row_ptr, seed_col, k,
                   out_score, out_idx, out_kind=DAE_OUT_SCORE, dtype=DAE_DTYPE_F32):
        V, H = W_enc.shape
        B = row_ptr.numel() - 1
        self.check(self.lib.dae_score_topk(self.h, _ptr(row_ptr), _ptr(col), _ptr(val),
                                           _ptr(W_enc), _ptr(b_enc), V, H, B, int(dtype),
                                           int(n_tracks), _ptr(seed_row_ptr), _ptr(seed_col),
                                           int(k), int(out_kind), _ptr(out_score), _ptr(out_idx)))

    def score_topk_handle(self, row_ptr, col, val, W_enc, b_enc, n_tracks, seed_row_ptr, seed_col, k,
                          out_score, out_idx, out_kind=DAE_OUT_SCORE, dtype=DAE_DTYPE_F32):
        """`score_topk` with its arguments marshalled ONCE: returns a zero-argument callable that enqueues the call on
        the context's stream (a driver that scores resident batches issues one every ~30 us: building 17 ctypes
        arguments and a torch stream context per call was a third of that).  The tensors must stay alive and in place."""
        V, H = W_enc.shape
        B = row_ptr.numel() - 1
        args = (self.h, _ptr(row_ptr), _ptr(col), _ptr(val), _ptr(W_enc), _ptr(b_enc), V, H, B, int(dtype), int(n_tracks),
                _ptr(seed_row_ptr), _ptr(seed_col), int(k), int(out_kind), _ptr(out_score), _ptr(out_idx))
        fn, check = self.lib.dae_score_topk, self.check
        keep = (row_ptr, col, val, W_enc, b_enc, seed_row_ptr, seed_col, out_score, out_idx)

        def call(_keep=keep):
            rc = fn(*args)
            if rc:
                check(rc)
        return call

    def score_topk_begin(self, row_ptr, col, val, W_enc, b_enc, n_tracks, seed_row_ptr, k, tau_out, dtype=DAE_DTYPE_F32):
        """First half of score_topk: encode + threshold sample -> tau_out [B] (this image's per-row lower bounds)."""
        V, H = W_enc.shape
        B = row_ptr.numel() - 1
        self.check(self.lib.dae_score_topk_begin(self.h, _ptr(row_ptr), _ptr(col), _ptr(val), _ptr(W_enc), _ptr(b_enc),
                                                 V, H, B, int(dtype), int(n_tracks), _ptr(seed_row_ptr), int(k),
                                                 _ptr(tau_out)))

    def score_topk_finish(self, tau, seed_row_ptr, seed_col, out_score, out_idx, out_kind=DAE_OUT_SCORE):
        """Second half: filter with `tau` [B] (the own bounds or the maximum over the shards) + selection."""
        self.check(self.lib.dae_score_topk_finish(self.h, _ptr(tau), _ptr(seed_row_ptr), _ptr(seed_col), int(out_kind),
                                                  _ptr(out_score), _ptr(out_idx)))

    def topk_dense(self, logits, ncols, col_base, seed_row_ptr, seed_col, k, out_score, out_idx,
                   out_kind=DAE_OUT_SCORE):
        B = logits.shape[0]
        ld = int(logits.stride(0)) if B > 1 else max(int(logits.stride(0)), int(logits.shape[1]))   # a 1-row view may carry stride 0
        self.check(self.lib.dae_topk_dense(self.h, _ptr(logits), ld, B,
                                           int(ncols), int(col_base), _ptr(seed_row_ptr),
                                           _ptr(seed_col), int(k), int(out_kind),
                                           _ptr(out_score), _ptr(out_idx)))

    def decode_mix_term(self, h, row_scale, n_cols, outT, dtype=DAE_DTYPE_F32):
        """outT[c, r] = row_scale[r] * sigmoid(logit[r, c]) for the prepacked columns c < n_cols (DAEs.py:180, DAE term)."""
        B, H = h.shape
        self.check(self.lib.dae_decode_mix_term(self.h, _ptr(h), B, H, int(dtype), _ptr(row_scale), int(n_cols),
                                                _ptr(outT), int(outT.stride(0))))

    def set_score_mix(self, mixT=None, w_title=None):
        """dae_decode_topk on this context ranks sigmoid(.) * w_title[r] + mixT[c, r] until cleared (no arguments)."""
        self.check(self.lib.dae_set_score_mix(self.h, _ptr(mixT), int(mixT.stride(0)) if mixT is not None else 0,
                                              int(mixT.shape[0]) if mixT is not None else 0, _ptr(w_title)))

    def title_score_exact(self, dae_ctx, d_pos, d_val, n_rows, V, W_enc, b_enc, d_titles, tm, d_use, n_tracks, k, out_score, out_idx,
                          guard_out, status):
        """One titled launch of the streamed loop in one call (dae_title_score_exact); `tm`: the Char_CNN model object."""
        nnz = int(d_pos.shape[0])
        bcast = 1 if (d_val.numel() == 1 and nnz != 1) else 0
        self.check(self.lib.dae_title_score_exact(
            self.h, dae_ctx.h, _ptr(d_pos), _ptr(d_val), bcast, nnz, int(n_rows), int(V), _ptr(W_enc), _ptr(b_enc),
            int(W_enc.shape[1]), _ptr(d_titles), tm.input_len, _ptr(tm.p["char_embedding"]), tm.char_size, tm.embedding,
            _ptr(tm.p["conv_w"]), _ptr(tm.p["conv_b"]), tm._fs, len(tm.filter_sizes), tm.filter_num, tm.ld, _ptr(d_use),
            int(n_tracks), int(k), _ptr(out_score), _ptr(out_idx), _ptr(guard_out), _ptr(status)))

    def mix_topk_exact(self, dae_ctx, feat, h, w_title, w_playlist, n_tracks, seed_row_ptr, seed_col, k, out_score, out_idx,
                       guard_out=None):
        """On the title scorer's context: top-k of the title mix, bit-identical to the fp32 path, both GEMMs on bf16
        operands (dae_mix_topk_exact; both contexts prepacked with DAE_DTYPE_BF16_EXACT and bound to the same stream)."""
        B = h.shape[0]
        self.check(self.lib.dae_mix_topk_exact(self.h, dae_ctx.h, _ptr(feat), int(feat.stride(0)), _ptr(h), int(h.stride(0)),
                                               int(B), _ptr(w_title), _ptr(w_playlist), int(n_tracks), _ptr(seed_row_ptr),
                                               _ptr(seed_col), int(k), _ptr(out_score), _ptr(out_idx), _ptr(guard_out)))

    def topk_merge(self, cand_logit, cand_idx, out_score, out_idx, out_kind=DAE_OUT_SCORE):
        G, B, k = cand_logit.shape
        self.check(self.lib.dae_topk_merge(self.h, G, B, k, _ptr(cand_logit), _ptr(cand_idx),
                                           int(out_kind), _ptr(out_score), _ptr(out_idx)))

    def profile_enable(self, on=True):
        self.check(self.lib.dae_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_int()
        self.check(self.lib.dae_profile_read(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_kernel(self):
        """Symbol of the kernel the last profiled launch ran (what rocprofv3's kernel trace calls it)."""
        return self.lib.dae_profile_kernel(self.h).decode()

    def clock_probe(self, out2, stream=None, window_us=50):
        """Enqueue the one-wave clock sampler (`out2`: CUDA int64/uint64 tensor of 2); returns the wall clock's kHz.
        GHz = out2[0] / out2[1] * kHz / 1e6 once the stream has run it."""
        khz = ctypes.c_int()
        self.check(self.lib.dae_clock_probe(self.h, ctypes.c_void_p(stream.cuda_stream) if stream is not None else None,
                                            int(window_us), _ptr(out2), ctypes.byref(khz)))
        return khz.value

    def last_plan(self):
        arr = (ctypes.c_int32 * 8)()
        self.lib.dae_last_plan(arr)
        keys = ["R_TILE", "n_rg", "nb_rg", "S", "n_sample_tiles", "n_filter_tiles", "fused",
                "n_tiles"]
        return dict(zip(keys, list(arr)))

    def scratch_bytes(self):
        return int(self.lib.dae_scratch_bytes(self.h))


class _Block:
    """One polled feed's claim on a pinned result block of a Pipeline: released when the last array viewing it dies."""

    def __init__(self, pipe, block):
        self.pipe, self.block = pipe, block
        pipe._held[block] = pipe._held.get(block, 0) + 1

    def __del__(self):
        try:
            p = self.pipe
            n = p._held.get(self.block, 0) - 1
            if n > 0:
                p._held[self.block] = n
            else:
                p._held.pop(self.block, None)
            if p.h:
                p.lib.dae_pipeline_release(p.h, self.block)
                if p._closing and not p._held:           # close() was asked for while this array was alive
                    p.close()
        except Exception:
            pass


_live_pipelines = None          # weak set of open pipelines: closed at interpreter exit, BEFORE the HIP runtime tears down


def _close_pipelines_at_exit():
    for p in list(_live_pipelines or ()):
        try:
            p._held = {}
            p.close()
        except Exception:
            pass


class Pipeline:
    """The drivers' loop inside the library (include/dae_hip.h dae_pipeline_*): host feeds in, host top-k lists out; a
    library-owned thread issues the launches on `lanes` contexts.  The weights are CUDA tensors the caller keeps alive.
    `submit` / `results` from one thread."""

    def __init__(self, W_enc, b_enc, W_dec, b_dec, n_tracks, dtype=DAE_DTYPE_F32, k=500, group_rows=1024, max_nnz=1 << 20,
                 lanes=2, want_scores=True, result_blocks=None, device_index=0, title=None):
        """title: a models.title_models.Char_CNN with its variables on the device -> a titled pipeline
        (dae_pipeline_create_titled): `submit(..., titles=, titles_use=)` ranks the title-mixed score."""
        self.lib = load()
        V, H = W_enc.shape
        self._keep = (W_enc, b_enc, W_dec, b_dec, title)
        self.title_len = None if title is None else int(title.input_len)
        self.k, self.want_scores, self.group_rows, self.max_nnz = int(k), bool(want_scores), int(group_rows), int(max_nnz)
        h = ctypes.c_void_p()
        n_slots = 2 * lanes + 2              # launches the pipeline holds at once (csrc/pipeline.hip)
        n_blocks = max(int(result_blocks or 0), n_slots + 6)
        self._held = {}                      # result block -> claims of arrays still alive in the caller's hands
        self._closing = False
        self.max_held = n_blocks - n_slots                    # more blocks out than this: poll() copies instead of lending
        if title is None:
            rc = self.lib.dae_pipeline_create(int(device_index), _ptr(W_enc), _ptr(b_enc), _ptr(W_dec), _ptr(b_dec), V, H, int(n_tracks),
                                              int(dtype), int(k), int(group_rows), int(max_nnz), int(lanes), 1 if want_scores else 0,
                                              n_blocks, ctypes.byref(h))
        else:
            tm = title
            rc = self.lib.dae_pipeline_create_titled(
                int(device_index), _ptr(W_enc), _ptr(b_enc), _ptr(W_dec), _ptr(b_dec), V, H, int(n_tracks),
                _ptr(tm.p["char_embedding"]), tm.char_size, tm.embedding, _ptr(tm.p["conv_w"]), _ptr(tm.p["conv_b"]), tm._fs,
                len(tm.filter_sizes), tm.filter_num, _ptr(tm.p["Output_WT"]), _ptr(tm.p["Output_b"]), tm.ld, tm.input_len,
                int(dtype), int(k), int(group_rows), int(max_nnz), int(lanes), 1 if want_scores else 0, n_blocks, ctypes.byref(h))
        if rc != 0:
            raise DaeError("dae_pipeline_create failed (%d): %s" % (rc, self.lib.dae_pipeline_last_error(None).decode()))
        self.h = h
        self.pending = 0                     # feeds submitted and not yet yielded
        # the foreign calls' out-parameters, made once (a feed is ~25 us of this thread: five ctypes objects and their byref()
        # per poll were 3 us of it)
        self._t, self._n, self._b = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int()
        self._ip, self._sp = ctypes.POINTER(ctypes.c_int32)(), ctypes.POINTER(ctypes.c_float)()
        self._poll_args = (ctypes.byref(self._t), ctypes.byref(self._ip), ctypes.byref(self._sp) if self.want_scores else None,
                           ctypes.byref(self._n), ctypes.byref(self._b))
        self._t_ref = ctypes.byref(self._t)
        # a pipeline owns a library thread and streams: it must be gone before the runtime is (a destructor running HIP calls
        # during interpreter teardown is undefined -- ADVICE r4)
        global _live_pipelines
        if _live_pipelines is None:
            import atexit
            import weakref
            _live_pipelines = weakref.WeakSet()
            atexit.register(_close_pipelines_at_exit)
        _live_pipelines.add(self)

    def _check(self, rc):
        if rc < 0:
            raise DaeError("dae_pipeline error %d: %s" % (rc, self.lib.dae_pipeline_last_error(self.h).decode()))
        return rc

    def submit(self, positions, values, n_rows, titles=None, titles_use=None):
        """positions: int64 [nnz, 2] (C-contiguous numpy), values: float32 [nnz] or one value.  -> True, or False when every
        lane holds lists that were not fetched yet (take `results()` first).  titles [n_rows, L] int32 + titles_use [n_rows]
        (a titled pipeline): the launch ranks the title-mixed score."""
        import numpy as np
        if self._closing:
            raise DaeError("dae_pipeline: closed")
        pos = positions if (type(positions) is np.ndarray and positions.dtype == np.int64 and positions.flags.c_contiguous) \
            else np.ascontiguousarray(positions, np.int64)
        val = values if (type(values) is np.ndarray and values.dtype == np.float32 and values.flags.c_contiguous) \
            else np.ascontiguousarray(values, np.float32)
        nnz = pos.size >> 1
        if val.size != nnz and val.size != 1:
            raise ValueError("positions (%d) and values (%d) differ in length" % (nnz, val.size))
        if titles is not None:
            tt = np.ascontiguousarray(titles, np.int32).reshape(-1)
            uu = np.ascontiguousarray(titles_use, np.float32).reshape(-1)
            if tt.size != int(n_rows) * self.title_len or uu.size != int(n_rows):
                raise ValueError("titles must be [n_rows, %d] and titles_use [n_rows]" % self.title_len)
            rc = self._check(self.lib.dae_pipeline_submit_titled(
                self.h, pos.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p),
                1 if (val.size == 1 and nnz != 1) else 0, nnz, int(n_rows), tt.ctypes.data_as(ctypes.c_void_p),
                uu.ctypes.data_as(ctypes.c_void_p), self._t_ref))
        else:
            rc = self.lib.dae_pipeline_submit(self.h, ctypes.c_void_p(pos.__array_interface__["data"][0]),
                                              ctypes.c_void_p(val.__array_interface__["data"][0]),
                                              1 if (val.size == 1 and nnz != 1) else 0, nnz, int(n_rows), self._t_ref)
            if rc < 0:
                self._check(rc)
        if rc == DAE_PIPE_BUSY:
            return False
        self.pending += 1
        return True

    def flush(self):
        self._check(self.lib.dae_pipeline_flush(self.h))

    def poll(self, wait=True, copy=False):
        """The next feed in submission order -> (idx [n_rows, k] int32, score or None), or None when it is not ready
        (wait=False) or nothing is pending.  The arrays VIEW a pinned result block that returns to the pipeline when they
        are garbage collected (copy=True: private copies, the block returns at once)."""
        import numpy as np
        if self.pending == 0:
            return None
        n, b, ip_, sp_ = self._n, self._b, self._ip, self._sp
        rc = self.lib.dae_pipeline_poll(self.h, 1 if wait else 0, *self._poll_args)
        if rc < 0:
            self._check(rc)
        if rc == DAE_PIPE_BUSY:
            raise DaeError("dae_pipeline: every result block is held (drop or copy the arrays of earlier feeds)")
        if n.value == 0:
            return None
        self.pending -= 1
        if b.value not in self._held and len(self._held) >= self.max_held:
            copy = True                      # the caller keeps many results alive: this one is copied, its block goes back
        owner = _Block(self, b.value)
        n_el = n.value * self.k

        def view(ptr, ctype, dt):
            # the ctypes array object is the memory owner numpy sees: it carries the block's claim, so every array derived
            # from the result (slices, views) keeps the block out of the pipeline's hands
            buf = (ctype * n_el).from_address(ctypes.addressof(ptr.contents))
            buf._owner = owner
            return np.frombuffer(buf, dtype=dt).reshape(n.value, self.k)
        idx = view(ip_, ctypes.c_int32, np.int32)
        score = view(sp_, ctypes.c_float, np.float32) if self.want_scores else None
        if copy:
            idx, score = idx.copy(), (None if score is None else score.copy())
        return idx, score

    def exact_margin(self, scale):
        self._check(self.lib.dae_pipeline_exact_margin(self.h, float(scale)))

    def times(self):
        """ms since creation: the library thread issuing / idle, the caller in submit / waiting in poll."""
        a = (ctypes.c_uint64 * 4)()
        self._check(self.lib.dae_pipeline_times(self.h, a))
        return dict(zip(("issue_ms", "idle_ms", "submit_ms", "wait_ms"), [round(int(x) / 1e6, 2) for x in a]))

    def stats(self):
        a = (ctypes.c_uint64 * 3)()
        self._check(self.lib.dae_pipeline_stats(self.h, a))
        return {"launches": int(a[0]), "feeds": int(a[1]), "guard_fallbacks": int(a[2])}

    def close(self):
        """Destroys the pipeline -- once no array handed out by poll() views its pinned blocks any more (until then the
        pipeline only stops taking feeds)."""
        if getattr(self, "h", None):
            if self._held:
                self._closing = True
                return
            self.lib.dae_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self._held = {}              # (nothing can view the blocks any more: the claims hold a reference to this object)
            self.close()
        except Exception:
            pass
