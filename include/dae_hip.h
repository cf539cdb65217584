/*
 * dae_hip.h -- C ABI of libdae_hip.so: the MI355X (gfx950) implementation of the
 * multi-hot denoising-autoencoder scoring path of hojinYang/spotify_recSys_challenge_2018.
 *
 * The reference has no FFI: its boundary is `sess.run(model.y_pred | [optimizer, cost], feed_dict)`
 * on the TF1 graph built in models/DAEs.py.  Each entry point below names the graph section
 * (reference file:line) it replaces.  A reference maintainer binds these with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; dae_last_error(ctx) gives the message;
 *   - all data pointers are CALLER-OWNED DEVICE pointers (e.g. torch tensors); the library only
 *     allocates scratch inside the ctx (grown lazily, never inside a steady-state call once sized);
 *   - all work is enqueued on the ctx stream (dae_set_stream), no hidden synchronisation;
 *   - one ctx is not thread-safe; independent ctxs are;
 *   - indices are int32, values fp32; rows = playlists, columns = item ids (tracks then artists).
 *
 * Numerics contract (DESIGN.md "canonical order"): the fp32 path is bit-reproducible and equals
 * oracle/dae_oracle.c bit-for-bit: encode = fmaf chain over the de-duplicated non-zeros in
 * ascending column order; decode = fmaf chain over k = 0..H-1 (v_mfma_f32_32x32x2_f32 is exactly
 * that chain); ranking key = (fp32 logit desc, column index asc); scores = dae canonical sigmoid.
 */
#ifndef DAE_HIP_H
#define DAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dae_ctx dae_ctx;

/* error codes */
#define DAE_OK            0
#define DAE_ERR_ARG      -1   /* bad argument (shape, alignment, null pointer)           */
#define DAE_ERR_HIP      -2   /* a HIP runtime call failed                                */
#define DAE_ERR_NOMEM    -3   /* scratch allocation failed                                */
#define DAE_ERR_STATE    -4   /* call sequence error (e.g. weights not prepacked)         */

/* what dae_decode_topk / dae_topk_* write into out_score */
#define DAE_OUT_SCORE     0   /* canonical sigmoid(logit)  (what the reference ranks on)  */
#define DAE_OUT_LOGIT     1   /* raw fp32 logit (needed when shards are merged later)     */

/* decode arithmetic */
#define DAE_DTYPE_F32     0   /* v_mfma_f32_32x32x2_f32, exact fp32, bit-equal to oracle  */
#define DAE_DTYPE_BF16    1   /* v_mfma_f32_32x32x16_bf16, fp32 accumulate (cfg 5)        */
/* BASELINE.json north_star: the decode as a bf16 MFMA GEMM whose top-k lists are BIT-IDENTICAL to the fp32 path
 * (indices and scores; main_challenge.py:26-36 ranks the fp32 y_pred).  The bf16 GEMM only FILTERS: with a rigorous
 * per-column bound eps_c >= |z_fp32 - z_bf16| (bf16 rounding of both operands + both accumulations; computed by
 * dae_prepack_decoder, readable through dae_exact_bounds), the threshold is taken from lower bounds z_bf16 - eps_c,
 * every column with z_bf16 + eps_c >= threshold survives, and the survivors' logits are recomputed with the canonical
 * fp32 fmaf chain before they are ranked.  Accepted by dae_prepack_decoder (bf16 image + bounds + a row-major fp32
 * copy of the decoder rows), dae_score_topk and dae_decode_topk.  Precondition of the bound: hidden activations in
 * [0, 1] (sigmoid outputs, DAEs.py:66-67) -- always true in dae_score_topk; a row passed to dae_decode_topk that
 * violates it returns no recommendations (idx -1, score -inf).  Not available with dae_set_score_mix: the title mix
 * has its own exact entry point, dae_mix_topk_exact. */
#define DAE_DTYPE_BF16_EXACT 2

/* ---- lifecycle -------------------------------------------------------------------------- */

/* library ABI version (major*1000 + minor). */
int dae_version(void);

/* Create a context bound to HIP device `device`.  Replaces tf.Session() (main_challenge.py:66,
 * main_train.py:172). */
int dae_create(int device, dae_ctx** out);
int dae_destroy(dae_ctx* ctx);

/* Stream all later calls are enqueued on (a hipStream_t passed as void*; NULL = default stream). */
int dae_set_stream(dae_ctx* ctx, void* hip_stream);

/* Last error message of this ctx (never NULL). dae_last_error(NULL) = last creation error. */
const char* dae_last_error(const dae_ctx* ctx);

/* Bytes of device scratch currently held by the ctx. */
size_t dae_scratch_bytes(const dae_ctx* ctx);

/* Per-kernel timing for roofline reporting: when enabled, the dominant kernel of each
 * dae_decode_* call is bracketed by hipEvents on the ctx stream.  dae_profile_read
 * synchronises those events and returns total milliseconds and the launch count, then resets. */
int dae_profile_enable(dae_ctx* ctx, int on);
int dae_profile_read(dae_ctx* ctx, double* ms_total, int* launches);
/* Symbol (as rocprofv3's kernel trace prints it) of the kernel the last event pair bracketed; "" before the first. */
const char* dae_profile_kernel(const dae_ctx* ctx);

/* Engine clock actually sustained while other work runs (the chip clocks to its power budget: the fp32 matrix-core
 * roof moves with it).  Enqueues ONE wave on `hip_stream` (any stream; NULL = the ctx stream) that reads the shader
 * cycle counter (s_memtime) and the constant-rate wall clock (s_memrealtime) `window_us` apart and stores
 * {shader cycles, wall-clock ticks} in out2_dev (device, 2 x uint64).  wall_khz_out (host, may be NULL) receives the
 * wall clock's rate (hipDeviceAttributeWallClockRate).  GHz = cycles / ticks * wall_khz / 1e6. */
int dae_clock_probe(dae_ctx* ctx, void* hip_stream, int window_us, uint64_t* out2_dev, int* wall_khz_out);

/* Geometry of the last dae_decode_topk issued from the calling thread, for roofline accounting:
 * {R_TILE, n_row_groups, blocks_per_row_group, sample_stride S, n_sample_tiles (phase A),
 *  n_filter_tiles (phase B), fused(0/1), n_tiles}.  A tile is 32 vocabulary columns. */
int dae_last_plan(int32_t out[8]);

/* ---- input: COO -> CSR (DAEs.py:33-38 SparseTensor + sparse_tensor_to_dense) --------------- */

/* The reference feeds COO (row, col) pairs with duplicates and ASSIGNMENT semantics
 * (last occurrence wins).  The host shim (models/DAEs.py of this repo) de-duplicates and sorts
 * them into CSR (row_ptr[B+1], col ascending per row, val).  All encode/train entry points take
 * that CSR. */

/* ---- encode (DAEs.py:40-42 dropout+normalise, :64-70 encoder) ----------------------------- */

/* The feed of the reference graph on the device (models/DAEs.py:23-35; utils/data_reader.py builds
 * it row by row): COO entries (row, col) -> value in FEED ORDER, duplicates allowed;
 * tf.sparse_tensor_to_dense(validate_indices=False) scatters them by assignment, so the LAST
 * occurrence of a (row, col) wins.  Produces the CSR every other entry point takes: columns
 * ascending per row, one entry per (row, col), explicit zeros dropped.
 *   positions [nnz,2] int64 (row in batch, column), values [nnz] fp32 (or ONE value for all entries
 *   when values_broadcast != 0: the reference feeds np.ones / a scalar).
 *   row_ptr [n_rows+1], col / val with room for nnz entries (row_ptr[n_rows] of them are written).
 *   status: device int32, 0 = ok, bit 0 = an entry had row/col out of range (it is skipped; the host
 *   restatement raises ValueError for the same input). */
int dae_coo_to_csr(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                   int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                   int32_t* status);

/* The seed lists of a scoring call whose seeds are the playlist's OWN tracks -- what both reference drivers pass
 * (main_challenge.py:76-88 `seed` = the track ids x_positions feeds; main_train.py:64-89 `test_seed` likewise): the
 * columns < n_tracks of every row of the input CSR, as a CSR (seed_row_ptr [B+1], seed_col with room for
 * row_ptr[B] entries; sorted and unique per row because the input rows are).  Saves the host the list handling and
 * two uploads per batch.  Any B (slabs of 16384 rows inside). */
int dae_seeds_from_csr(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, int B, int n_tracks,
                       int32_t* seed_row_ptr, int32_t* seed_col);

/* h[r,:] = hidden_dropout( sigmoid( sum_c (x[r,c]/(s_r+1e-10)) * W_enc[c,:] + b_enc ) )
 *   x      = input_dropout(CSR row r), s_r = sum of surviving weights.
 *   ikp/kp = input / hidden keep probabilities (1.0 = identity, the inference setting);
 *   seed   = counter-based RNG seed for both dropouts (ignored when ikp == kp == 1).
 * W_enc [V,H] row-major fp32, b_enc [H], h_out [B,H] row-major fp32.  H % 4 == 0. */
int dae_encode(dae_ctx* ctx,
               const int32_t* row_ptr, const int32_t* col, const float* val,
               const float* W_enc, const float* b_enc,
               int V, int H, int B,
               float ikp, float kp, uint32_t seed,
               float* h_out);

/* ---- decoder weights ------------------------------------------------------------------------ */

/* Re-tile W_dec[col_lo:col_hi, :] (row-major [V,H] fp32, DAEs.py:123 / tied :54) and
 * b_dec[col_lo:col_hi] into the MFMA operand order the decode kernels stream (DESIGN.md "HBM
 * layout").  Call once per weight update (model load / after a training epoch); the packed copy
 * lives in the ctx.  dtype selects the fp32 or bf16 packed image (both may be resident). */
int dae_prepack_decoder(dae_ctx* ctx, const float* W_dec, const float* b_dec,
                        int V, int H, int col_lo, int col_hi, int dtype);

/* The same from a RANK-LOCAL copy of the rows (vocabulary shards: a rank holds only its own rows of W_dec):
 * W_rows [n_rows, H] / b_rows [n_rows] are the global columns col_lo .. col_lo + n_rows; the image covers exactly
 * those.  (dae_prepack_decoder indexes its arguments by global column; this entry point does the shift inside the
 * library, where the prepack kernels' access range is known.)  Replaces the same section as dae_prepack_decoder. */
int dae_prepack_decoder_rows(dae_ctx* ctx, const float* W_rows, const float* b_rows, int n_rows, int H, int col_lo,
                             int dtype);

/* Let `dst` use the decoder image `src` prepacked for `dtype` instead of holding a copy of its own (the contexts of
 * several batches in flight score with the same weights: one image stays resident in the 256 MB Infinity Cache where
 * two or three copies of 87 / 174 MB evict each other; it also saves their memory and re-tiling).  `dst` borrows the
 * buffers: `src` must outlive every use, must not be re-prepacked while `dst` has work in flight, and the caller orders
 * `src`'s prepack before `dst`'s first launch (streams!).  A later dae_prepack_decoder on `dst` gives it an image of its
 * own again.  Same device only. */
int dae_share_decoder(dae_ctx* dst, const dae_ctx* src, int dtype);

/* DAE_DTYPE_BF16_EXACT: copy the per-column bounds eps_c of the prepacked image (col_lo <= c < col_hi) to
 * eps_out (device, col_hi - col_lo floats).  |fp32 logit - bf16 logit| <= eps_c for every hidden row in [0, 1]^H;
 * DESIGN.md section 2b derives it, tests/test_gpu_exact.py checks it against measured differences. */
int dae_exact_bounds(dae_ctx* ctx, float* eps_out);

/* DAE_DTYPE_BF16_EXACT, the BOUND GUARD.  One term of eps_c -- the accumulation inside v_mfma_f32_32x32x16_bf16 -- rests
 * on an error model of that instruction, not on a specification.  So every survivor the refine launch recomputes in
 * fp32 is tested against what the filter launch promised for it (u - 2 eps_c <= z_fp32 <= u, u the stored upper bound;
 * both numbers are in registers: the test is free), and a violation is COUNTED in two device words of the context
 * instead of silently costing a true top-k column.  A caller that sees a count != 0 must treat the results of the calls
 * since the last read as unproven and re-run them with DAE_DTYPE_F32 (models/DAEs.py recommend / recommend_iter do).
 *   dae_exact_guard_read : synchronises the ctx stream; violations (count since the last read that returned != 0) and
 *                          one violating global column (-1 when none); resets the words when the count is non-zero.
 *   dae_exact_guard_words: the device address of {int32 violations, int32 column}, for a caller that fetches the words
 *                          with its results (no synchronisation here); valid for the life of the ctx.
 * The ranking values that reach main_challenge.py:26-36's argsort are fp32 either way: the guard protects the SET. */
int dae_exact_guard_read(dae_ctx* ctx, int32_t* violations, int32_t* column);
int dae_exact_guard_words(dae_ctx* ctx, const int32_t** words_dev);
/* The guard words {violations so far, a violating column} as they stand after everything enqueued on the context's stream so
 * far, copied to words_out_dev (DEVICE int32[2]) in stream order: a snapshot that travels with a launch's lists.  The words
 * are cumulative until a dae_exact_guard_read finds them non-zero (and resets them): a count that differs from the previous
 * launch's snapshot on the same context means THIS launch is unproven (models/DAEs.py: the interpreter loop). */
int dae_exact_guard_snapshot(dae_ctx* ctx, int32_t* words_out_dev);
/* How selective the filter was (the exact mode's rate depends on it, its results never): of the LAST exact scoring launch
 * of this context, out3 = {rows refined, candidates the bf16 filter launch left for them (sum over the rows), candidates
 * recomputed in fp32 after the narrowing step (sum)}.  Synchronises the ctx stream. */
int dae_exact_stats_read(dae_ctx* ctx, uint64_t out3[3]);

/* Factor on every eps_c computed by the NEXT dae_prepack_decoder(DAE_DTYPE_BF16_EXACT) of this context (default 1).
 * > 1 widens the bounds: a safety margin for a caller who distrusts the error model (more survivors to recompute, same
 * results).  < 1 VOIDS the guarantee and exists so that the guard can be exercised: results may then differ from the
 * fp32 path, and dae_exact_guard_read reports it (tests/test_gpu_exact.py).  0 < scale <= 1024. */
int dae_set_exact_margin(dae_ctx* ctx, float scale);
/* The same hook for the columns [col_from, col_to) alone (global ids; applied by the NEXT exact prepack, the other columns
 * keep dae_set_exact_margin's factor): lets a test void the bound of a column that every row DROPS, which only the audit
 * below can see (tests/test_gpu_exact.py::test_audit_sees_a_violation_on_a_dropped_column).  col_from == col_to: none.
 * scale < 0 (>= -1024) FORGES the filter outright: the UPPER bound of those columns is put |scale| logits too low, so rows
 * drop columns that belong in their lists -- the failure the audits exist for (tests/test_gpu_title_exact.py). */
int dae_set_exact_margin_range(dae_ctx* ctx, int col_from, int col_to, float scale);

/* DAE_DTYPE_BF16_EXACT, the AUDIT of what the guard cannot see (csrc/audit.hip).  The guard tests the survivors the refine
 * launch recomputes; a column the bf16 filter launch DROPPED (upper bound u < the row's threshold) is never recomputed, so a
 * bound that fails there -- the only failure that can change a top-k list of main_challenge.py:28-36 -- would go unseen.
 * Every every_n-th exact scoring launch of the context (dae_decode_topk / dae_score_topk / dae_score_topk_finish; default 64,
 * 0 = never) therefore takes n_tiles (default 16, <= 64) pseudo-random 32-column tiles of the ranked columns -- different ones
 * each time; nearly all of their elements are dropped ones -- and for EVERY row of the launch recomputes both the filter
 * launch's upper bound u (the same bf16 MFMA sequence on the same operands: the same bits) and the canonical fp32 logit, and
 * counts every element outside [u - 2 eps_c, u] in the GUARD WORDS above: callers react exactly as to a survivor's violation.
 *   dae_exact_audit_read: out3 = {audits run, (row, column) elements checked, violations among them} since the context was
 *                         created; synchronises the ctx stream.
 * Proven: the bound, given the accumulation model of the bf16 MFMA.  Checked always: survivors.  Checked on a sample: dropped
 * columns (all rows x n_tiles x 32 columns per audit).  Assumed: nothing else.
 * Under the exact title mix (dae_mix_topk_exact / dae_title_score; set on the TITLE context, same defaults) the audit tests the
 * OUTCOME instead: the sampled tiles' mixed scores, recomputed with the canonical chains of both images and the fp32 path's mix
 * (DAEs.py:153-181), against the lists the launch just wrote -- an element above its row's k-th listed score that is neither
 * listed nor one of the row's seeds counts as a violation in the title context's guard words (csrc/mixexact.hip mix_audit). */
int dae_set_exact_audit(dae_ctx* ctx, int every_n, int n_tiles);
int dae_exact_audit_read(dae_ctx* ctx, uint64_t out3[3]);

/* ---- decode (DAEs.py:73-77 tied / :141-145 untied) ---------------------------------------- */

/* out[r, c-col_lo] = (apply_sigmoid ? sigmoid : id)( h[r,:] . W_dec[c,:] + b_dec[c] )
 * for c in the prepacked range.  out is [B, ld_out] fp32 with ld_out >= col_hi-col_lo.
 * This is the reference's y_pred (main_train.py:66, main_challenge.py:80). */
int dae_decode_dense(dae_ctx* ctx, const float* h, int B, int H, int dtype,
                     int apply_sigmoid, float* out, int64_t ld_out);

/* ---- rank (main_challenge.py:26-36, :87; metrics.py:59-68) ------------------------------- */

/* Fused decode + top-k over the prepacked column range restricted to track columns
 * (c < n_tracks, main_challenge.py:87), excluding each row's seed tracks
 * (seed_row_ptr[B+1], seed_col sorted ascending & unique per row; may be NULL = no seeds).
 * Order: logit descending, then column index ascending.  Writes k entries per row:
 * out_idx[r, i] = GLOBAL column id (or -1 when fewer than k candidates exist, as the reference's
 * cand[:500] of a short list), out_score per `out_kind`.  1 <= k <= 1024. */
int dae_decode_topk(dae_ctx* ctx, const float* h, int B, int H, int dtype,
                    int n_tracks,
                    const int32_t* seed_row_ptr, const int32_t* seed_col,
                    int k, int out_kind,
                    float* out_score, int32_t* out_idx);

/* The whole scoring path in one call: dae_encode (inference keep-probs) -> dae_decode_topk,
 * with the hidden activations kept inside the ctx in the decode kernels' operand order (no
 * [B,H] round trip, no re-pack pass).  Results are identical to calling the two separately.
 * This is what main_challenge.py:80-90 / main_train.py:66-89 do per batch. */
int dae_score_topk(dae_ctx* ctx,
                   const int32_t* row_ptr, const int32_t* col, const float* val,
                   const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                   int n_tracks,
                   const int32_t* seed_row_ptr, const int32_t* seed_col,
                   int k, int out_kind,
                   float* out_score, int32_t* out_idx);

/* dae_score_topk in two halves, for vocabulary-sharded scoring with a THRESHOLD EXCHANGE between them:
 *   begin:  encode + the threshold sample of this image's columns -> tau_out[B] (device): per row a valid lower bound
 *           of the k-th largest rankable non-seed logit among THIS image's columns (-inf when the image is small enough
 *           to be ranked densely).  The k-th largest logit over ALL shards is at least every shard's bound, so the
 *           element-wise MAXIMUM of the shards' tau_out (one all-gather of 4 B per row and rank) is a valid -- and far
 *           tighter -- threshold for every shard.
 *   finish: the filter launch with `tau` (tau_out, or that maximum), the selection: out_score / out_idx hold the image's
 *           columns with logit >= tau in rank order, at most k, padded with (-inf, -1): merged over the shards
 *           (dae_topk_merge) they give exactly what the unsharded call returns.
 * One begin / finish pair at a time per context, same stream, 1 <= B <= 4096; seed_row_ptr of finish is the one
 * begin was given.  Replaces the same graph section as dae_score_topk (main_challenge.py:80-90, one batch). */
int dae_score_topk_begin(dae_ctx* ctx,
                         const int32_t* row_ptr, const int32_t* col, const float* val,
                         const float* W_enc, const float* b_enc,
                         int V, int H, int B, int dtype, int n_tracks,
                         const int32_t* seed_row_ptr, int k, float* tau_out);
int dae_score_topk_finish(dae_ctx* ctx, const float* tau,
                          const int32_t* seed_row_ptr, const int32_t* seed_col,
                          int out_kind, float* out_score, int32_t* out_idx);

/* Unfused ranking of caller-provided dense logits/scores [B, ld] whose column 0 is global
 * column `col_base`; ranks columns [0, ncols).  Same order/seed rules as dae_decode_topk.
 * (parity path: dae_decode_dense + dae_topk_dense must equal dae_decode_topk.) */
int dae_topk_dense(dae_ctx* ctx, const float* logits, int64_t ld, int B, int ncols, int col_base,
                   const int32_t* seed_row_ptr, const int32_t* seed_col,
                   int k, int out_kind, float* out_score, int32_t* out_idx);

/* Merge G per-shard candidate lists (as produced with DAE_OUT_LOGIT, gathered by RCCL
 * all-gather into cand_logit/cand_idx [G, B, k]) into the global top-k.  idx == -1 entries are
 * ignored. */
int dae_topk_merge(dae_ctx* ctx, int G, int B, int k,
                   const float* cand_logit, const int32_t* cand_idx,
                   int out_kind, float* out_score, int32_t* out_idx);

/* Tell the context that `batches_in_flight` batches are being scored concurrently on different streams (their own
 * contexts).  With more than one, the latency-bound launches of the bf16 / exact paths take shapes that fit on a CU NEXT to
 * another batch's filter workgroup (fewer threads and registers): each is a little slower alone and the step is faster
 * (7.0 vs 6.5 M playlists/s with three batches in flight).  Results do not change.  Default: 1. */
int dae_set_overlap_hint(dae_ctx* ctx, int batches_in_flight);

/* Two batches in flight on two contexts / streams (bench.py): the dominant launch of the fused path (the threshold
 * filter over ~90 % of the vocabulary) occupies every CU, so two of them in flight only queue behind each other.  With a
 * gate, this context's launch waits for `wait_event` (a hipEvent_t the OTHER context records after its own launch) and
 * records `record_event` after itself: the two dominant launches alternate, everything else still overlaps.  Events are
 * caller-owned; NULL, NULL removes the gate.  Results are unaffected. */
int dae_set_decode_gate(dae_ctx* ctx, void* wait_event, void* record_event);

/* ---- the drivers' loop (main_challenge.py:72-93, main_train.py:62-96) ---------------------------------------------------
 *
 * The reference's loop per batch -- reader.next_batch() -> sess.run(y_pred, feed) -> cand_generate per row -- as a streaming
 * pipeline inside the library: HOST feeds in (the COO positions / values the reference's readers emit, utils/data_reader.py),
 * HOST top-k index lists out, nothing of the interpreter in between.  dae_pipeline_submit copies a feed into pinned staging
 * memory and returns; a library-owned thread issues the launches (upload, dae_coo_to_csr, dae_seeds_from_csr, dae_score_topk,
 * download) on `lanes` contexts / streams that take them in turn and share one packed decoder image; consecutive feeds share
 * a launch of up to `group_rows` rows (rows are scored independently: every feed gets the bits dae_score_topk gives it alone).
 * Seeds are the playlist's own tracks (the columns < n_tracks of its feed), what both reference drivers pass.
 * This group is the exception to "device pointers only": feeds and results are HOST memory.  One caller thread.
 *
 *   create : W_enc / b_enc / W_dec / b_dec are caller-owned DEVICE arrays ([V,H], [H], [V,H], [V]) that outlive the pipeline;
 *            dtype = DAE_DTYPE_*; a feed holds at most group_rows (<= 16384) rows and max_nnz entries; result_blocks pinned
 *            result blocks (>= lanes + 1) bound how many launches' lists the caller may hold at once.
 *   submit : positions [nnz,2] int64 (row in the FEED, column), values [nnz] fp32 (or one value: values_broadcast); n_rows
 *            rows.  Returns DAE_OK and the feed's ticket, or DAE_PIPE_BUSY (> 0): every lane holds lists that were not
 *            polled / released yet -- poll, then submit again.
 *   flush  : close the launch being filled (end of input; poll(wait) does it as well).
 *   poll   : the next feed IN SUBMISSION ORDER: *idx -> its [n_rows,k] global column ids (-1 = fewer candidates), *score
 *            (may be NULL) -> canonical sigmoid scores, both inside pinned result block *block, valid until
 *            dae_pipeline_release(block) (once per polled feed).  wait = 0: returns *n_rows = 0 when the feed is not ready.
 *            DAE_DTYPE_BF16_EXACT: a launch whose bound guard fired (dae_exact_guard_read) is re-scored with DAE_DTYPE_F32
 *            before its lists go out (dae_pipeline_stats counts it).
 *   stats  : {launches issued, feeds submitted, launches re-scored in fp32 after a bound-guard hit}. */
typedef struct dae_pipeline dae_pipeline;
#define DAE_PIPE_BUSY 1
int dae_pipeline_create(int device, const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
                        int V, int H, int n_tracks, int dtype, int k, int group_rows, int64_t max_nnz, int lanes,
                        int want_scores, int result_blocks, dae_pipeline** out);
int dae_pipeline_destroy(dae_pipeline* p);
int dae_pipeline_submit(dae_pipeline* p, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                        int n_rows, uint64_t* ticket_out);
/* The TITLED loop -- what the reference's `--challenge` really runs (main_challenge.py:58-59, :72-93: every batch goes through
 * DAE_title with `titles` / `titles_use` in the feed; DAEs.py:153-181).  create_titled: the pipeline above plus the title
 * scorer's variables (caller-owned DEVICE arrays in the layout of dae_title_features / dae_title_score: emb [n_char][E], conv_w /
 * conv_b back to back, filter_sizes a HOST array of n_sizes entries, Output_WT [V][ld_feat] = Output_W^T zero padded,
 * Output_b [V]); titles are rows of L characters.  Every lane holds a second context for the title scorer on its stream.
 * submit_titled: a feed with its titles [n_rows][L] int32 (-1 = no character) and titles_use [n_rows] (HOST arrays) -> the
 * launch ranks y = sigmoid(z_title) * w_title + sigmoid(z_dae) * w_playlist (dae_title_score) with seeds = the playlist's own
 * tracks.  Feeds submitted WITHOUT titles (dae_pipeline_submit) on the same pipeline rank the plain DAE logits
 * (dae_score_topk), as `DAE_title.recommend` does for batches whose titles_use is all zero; the two kinds never share a launch.
 * At most 4096 rows per launch.  DAE_DTYPE_BF16_EXACT: a launch under which the title context's guard words moved (a bound
 * violated, or rows with more survivors than the refine launch lists) is re-scored with DAE_DTYPE_F32 before its lists go out;
 * after two launches in a row with overflowing rows the mode pauses for 64 launches (they run on the fp32 kernels). */
int dae_pipeline_create_titled(int device, const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
                               int V, int H, int n_tracks, const float* emb, int n_char, int E, const float* conv_w,
                               const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, const float* Output_WT,
                               const float* Output_b, int ld_feat, int L, int dtype, int k, int group_rows, int64_t max_nnz,
                               int lanes, int want_scores, int result_blocks, dae_pipeline** out);
int dae_pipeline_submit_titled(dae_pipeline* p, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                               int n_rows, const int32_t* titles, const float* titles_use, uint64_t* ticket_out);
int dae_pipeline_flush(dae_pipeline* p);
int dae_pipeline_poll(dae_pipeline* p, int wait, uint64_t* ticket, const int32_t** idx, const float** score, int* n_rows,
                      int* block);
int dae_pipeline_release(dae_pipeline* p, int block);
int dae_pipeline_stats(dae_pipeline* p, uint64_t out3[3]);
/* Where the host side of the loop spent its time since creation, nanoseconds: {the library thread issuing launches, the
 * library thread waiting for a staged launch, the caller inside dae_pipeline_submit, the caller waiting in dae_pipeline_poll}. */
int dae_pipeline_times(dae_pipeline* p, uint64_t out4[4]);
/* dae_set_exact_margin for the pipeline's own decoder image (re-tiled here; no feed may be in flight). */
int dae_pipeline_exact_margin(dae_pipeline* p, float scale);
const char* dae_pipeline_last_error(const dae_pipeline* p);

/* ---- training step (DAEs.py:98-102) ------------------------------------------------------ */

/* Arithmetic of the three GEMMs of the training step of this context (forward hidden x W_dec^T, gW_dec = dz^T h,
 * dh = dz W_dec): DAE_DTYPE_F32 (default; fp32 MFMA) or DAE_DTYPE_BF16 (BASELINE.json configs[3]: operands
 * rounded to bf16 in registers / by the prepack, fp32 accumulate; the loss, dL/dz, the encoder gradient, the
 * parameters and Adam stay fp32; the backward GEMMs switch only when H % 128 == 0).  Sticky; applies to
 * dae_train_forward_backward and dae_train_shard_decode.  Replaces nothing in the reference (TF1 is fp32). */
int dae_set_train_dtype(dae_ctx* ctx, int dtype);

/* Forward + loss + backward for one batch.  x_* = input CSR (after the host coin flip
 * tracks-only / artists-only, main_train.py:202-213), y_* = target CSR (values are the y_ones
 * of the feed).  n_batch = the fixed graph batch the mean divides by (DAEs.py:100).
 * tied != 0: W_dec aliases W_enc (DAE_tied); gW_dec is then ignored and gW_enc receives both
 * gradients.  Gradients are dense fp32 buffers the caller owns (overwritten).
 * cost_out: device scalar = mean_rows(L) + reg_lambda * l2 (DAEs.py:79-82/:147-150, :100). */
int dae_train_forward_backward(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
        int V, int H, int B, int n_batch, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc, float* gb_enc, float* gW_dec, float* gb_dec,
        float* cost_out);

/* ---- the same step with the vocabulary ROW-SHARDED over ranks (training) ------
 * Rank g owns rows [col_lo, col_hi) of W_enc, W_dec and b_dec (pointers *_loc address the shard's
 * own [col_hi-col_lo, H] / [col_hi-col_lo] arrays); b_enc is replicated.  Every rank receives the
 * whole batch CSR with GLOBAL column ids.  One step = three stages around the two exchanges the
 * path really has (the caller issues them: RCCL all-reduce(sum) over [B,H] fp32):
 *
 *   dae_train_shard_encode   pre_partial[B,H] = sum over own columns of xhat * W_enc_loc rows
 *                            (xhat normalised by the row's global sum, DAEs.py:40-42, :66)
 *        -- all-reduce pre_partial --
 *   dae_train_shard_decode   h = dropout(sigmoid(pre + b_enc)) (:67-68); decode + loss + dz over own
 *                            columns (:75/:143, :98-100); gW_dec_loc, gb_dec_loc; dh_partial[B,H];
 *                            cost_partial (device scalar: own columns' loss / n_batch + lambda * l2
 *                            of the tensors this rank owns; b_enc counted by the col_lo == 0 rank)
 *        -- all-reduce dh_partial and cost_partial --
 *   dae_train_shard_finish   dpre, gb_enc (replicated, identical on all ranks), row-sparse gW_enc_loc
 *
 * tied != 0: W_dec_loc is ignored (W_enc_loc is the decoder), stage "decode" writes the decoder
 * gradient into gW_out = gW_enc_loc and stage "finish" accumulates the encoder part on top.
 * untied: gW_out = gW_dec_loc.  Adam is local: dae_adam_step on the owned rows.
 * The stages keep h / sigmoid in ctx scratch between calls: run them in order on one ctx.
 * With a single shard [0, V) the result equals dae_train_forward_backward up to fp32
 * re-association in the encoder sum. */
int dae_train_shard_encode(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const float* W_enc_loc, int col_lo, int col_hi, int H, int B,
        float ikp, uint32_t seed, float* pre_partial);
int dae_train_shard_decode(dae_ctx* ctx, const float* pre, const float* b_enc,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc_loc, const float* W_dec_loc, const float* b_dec_loc,
        int col_lo, int col_hi, int H, int B, int n_batch, int tied,
        float kp, uint32_t seed, float reg_lambda,
        float* gW_out, float* gb_dec_loc, float* dh_partial, float* cost_partial);
int dae_train_shard_finish(dae_ctx* ctx, const float* dh,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const float* W_enc_loc, const float* b_enc, const float* W_dec_loc, const float* b_dec_loc,
        int col_lo, int col_hi, int H, int B, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc_loc, float* gb_enc, float* gW_dec_loc, float* gb_dec_loc);

/* ---- title scorer of the challenge path (models/title_models/Char_CNN.py, models/DAEs.py:153-181) ---- */

/* Char_CNN.py:23-62: titles [B,L] int32 character ids (-1 = padding, embeds to zero) -> embedding [n_char,E]
 * -> for each of n_sizes filter sizes (host array filter_sizes) a VALID convolution with F filters
 * (conv_w = the TF variables Conv_W0.. [fs_i, E, 1, F] back to back, conv_b = Conv_b0.. [n_sizes, F]) -> ReLU
 * -> max over time -> concat -> dropout(keep_prob) -> feat [B, ld] (ld >= n_sizes*F, zero padded so that the
 * result can go straight into dae_decode_dense as a "hidden" matrix of size ld).  argmax / feat_raw (both
 * [B, n_sizes*F], nullable) keep the max positions and the pre-dropout features for the backward pass.
 * The vocabulary-wide output layer sigmoid(feat . Output_W + Output_b) (:64-72) is the decoder GEMM:
 * dae_prepack_decoder(Output_W^T [V, ld], Output_b) + dae_decode_dense(apply_sigmoid = 1). */
int dae_title_features(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char, int E,
                       const float* conv_w, const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F,
                       float keep_prob, uint32_t seed, float* feat, int64_t ld, int32_t* argmax, float* feat_raw);

/* A FROZEN title scorer's convolutions as a table (scoring: main_challenge.py:80-90 runs Char_CNN.py:23-62 with fixed
 * variables on every batch).  Titles are rows of character ids, so the inner sum of the convolution over the embedding has
 * only fs x (n_char + 1) distinct terms per filter: T[d][c][f] = sum_e emb[c][e] Conv_W[d][e][f].  After this call,
 * dae_title_features / dae_title_score on this context, called with THESE emb / conv_w arrays, keep_prob = 1 and no argmax /
 * feat_raw, sum fs table entries per (position, filter) instead of fs x E products -- the same real number in another
 * summation order (a different fp32 rounding; the title path is specified to a tolerance, as TF's conv2d has no documented
 * order).  The caller calls it again after the variables changed, or with emb = NULL to drop the table (training does:
 * calls with keep_prob < 1 or argmax never use it). */
int dae_title_prepack_features(dae_ctx* ctx, const float* emb, int n_char, int E, const float* conv_w,
                               const int32_t* filter_sizes, int n_sizes, int F);

/* DAEs.py:180: dae_score[r, c] = title_score[r, c] * w_title[r] + dae_score[r, c] * w_playlist[r] over the first
 * ncols columns; the weights are DAEs.py:159-162 (x_count = row_sum * input_keep_prob; u / (u + x_count + 1e-10),
 * x_count / (u + x_count + 1e-10)), computed by the caller from the feed. */
int dae_mix_scores(dae_ctx* ctx, const float* title_score, int64_t ld_title, float* dae_score, int64_t ld_dae,
                   const float* w_title, const float* w_playlist, int B, int ncols);

/* The same mix WITHOUT the two [B, V] score matrices (what `--challenge` runs for titled batches, main_challenge.py:80-90
 * with DAE_title.y_pred of DAEs.py:176-181):
 *
 * dae_decode_mix_term (on the DAE's context): outT[c * ldT + r] = w_playlist[r] * sigmoid(h[r,:] . W_dec[c,:] + b_dec[c])
 *   for the prepacked columns c < n_cols (pass n_tracks: only track columns are ranked) -- the second term of
 *   DAEs.py:180, stored TRANSPOSED ([column][row], ldT >= B) so that both sides touch whole cache lines.  Whole
 *   32-column tiles are written: outT needs room for n_cols rounded up to a multiple of 32 (but never past the image).
 * dae_set_score_mix (on the title scorer's context, whose prepacked "decoder" is Output_W^T / Output_b): until
 *   cleared with (NULL, 0, 0, NULL), dae_decode_topk on this context ranks
 *       y[r, c] = sigmoid(feat[r,:] . Output_W[:, c] + Output_b[c]) * w_title[r] + mixT[c * ld + r]
 *   instead of the logit -- same operations and order as dae_mix_scores, so the result is bit-identical to
 *   dae_mix_scores + dae_topk_dense(DAE_OUT_LOGIT) -- through the fused threshold path (sample tiles -> tau -> filter
 *   -> top-k); out_score holds y whatever out_kind says.  mixT ([n_cols][ld], columns past n_cols count as 0) and
 *   w_title are caller-owned device arrays. */
int dae_decode_mix_term(dae_ctx* ctx, const float* h, int B, int H, int dtype, const float* row_scale, int n_cols,
                        float* outT, int64_t ldT);
int dae_set_score_mix(dae_ctx* ctx, const float* mixT, int64_t ld, int n_cols, const float* w_title);

/* DAE_DTYPE_BF16_EXACT for the title mix: the top-k of y = sigmoid(z_title) * w_title + sigmoid(z_dae) * w_playlist
 * (DAEs.py:176-181; what main_challenge.py:80-90 ranks for every titled batch) BIT-IDENTICAL to the fp32 path above
 * (dae_decode_mix_term + dae_set_score_mix + dae_decode_topk, or dae_mix_scores + dae_topk_dense), with both vocabulary-
 * wide GEMMs on v_mfma_f32_32x32x16_bf16: one launch decodes each 32-column tile against the hidden rows of BOTH scorers
 * on bounds of the two logits (sample: lower bounds -> threshold; filter: upper bounds -> candidates), and the candidates'
 * logits are recomputed with the canonical fp32 chains and mixed with dae_mix_scores' operations before they are ranked.
 * No [B, V] matrix and no transposed term.  Called on the TITLE scorer's context; `dae` is the DAE's.  Both must hold
 * their weights prepacked with DAE_DTYPE_BF16_EXACT over the same columns [0, V) (DAE: hidden 256; title: Output_W^T in
 * rows of 448), and be bound to the same stream.  feat [B][ld_feat]: title features (dae_title_features; any finite
 * values -- the bound scales with the row's largest |feature|), h [B][ld_h]: the DAE's fp32 hidden rows (dae_encode),
 * w_title / w_playlist [B] in [0, 1] (DAEs.py:159-162).  Rows that violate a precondition return no recommendations
 * (idx -1).  The bound guard of the plain exact mode covers both GEMMs: dae_exact_guard_read / _words on the title
 * context; a row with more than 8192 columns left to recompute also counts there (column -3).  guard_out (nullable, DEVICE int32[2]):
 * receives the guard words {violations so far, a violating column} in stream order, for a fetch alongside the lists --
 * a count that grew since the previous launch's means THIS launch is not to be trusted (re-score it with
 * DAE_DTYPE_F32).  out_score holds y.  B <= 4096. */
/* One titled launch of the drivers' loop under DAE_DTYPE_BF16_EXACT in ONE call (main_challenge.py:72-93 with DAE_title:
 * `reader.next_batch()` -> `sess.run(model.y_pred)` -> `cand_generate`): the feed as the reader emits it (COO positions /
 * values ON THE DEVICE, as for dae_coo_to_csr), the titles [n_rows][L] int32 and titles_use [n_rows] on the device ->
 * dae_title_features, dae_coo_to_csr, dae_encode, dae_seeds_from_csr (seeds = the playlist's own tracks), dae_mix_weights,
 * dae_mix_topk_exact, with the intermediates in the title context's scratch.  Same results as the calls made one by one
 * (`DAE_title.recommend_iter` made them from Python: eight library calls and a dozen allocations, ~0.2 ms of a 0.6 ms
 * launch).  filter_sizes: HOST array.  csr_status: device int32, as dae_coo_to_csr's. */
/* dae_title_score: the same launch for any dtype -- DAE_DTYPE_BF16_EXACT as above; DAE_DTYPE_F32 / DAE_DTYPE_BF16 through the
 * fused mix (dae_decode_mix_term on the DAE's context, dae_set_score_mix + dae_decode_topk on the title context; both hold
 * their weights prepacked with that dtype), guard_out zeroed.  dae_title_score_exact = dae_title_score(DAE_DTYPE_BF16_EXACT). */
int dae_title_score(dae_ctx* title_ctx, dae_ctx* dae, int dtype, const int64_t* positions, const float* values, int values_broadcast,
                    int64_t nnz, int n_rows, int V, const float* W_enc, const float* b_enc, int H,
                    const int32_t* titles, int L, const float* emb, int n_char, int E, const float* conv_w,
                    const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, int ld_feat,
                    const float* titles_use, int n_tracks, int k, float* out_score, int32_t* out_idx,
                    int32_t* guard_out, int32_t* csr_status);
int dae_title_score_exact(dae_ctx* title_ctx, dae_ctx* dae, const int64_t* positions, const float* values, int values_broadcast,
                          int64_t nnz, int n_rows, int V, const float* W_enc, const float* b_enc, int H,
                          const int32_t* titles, int L, const float* emb, int n_char, int E, const float* conv_w,
                          const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, int ld_feat,
                          const float* titles_use, int n_tracks, int k, float* out_score, int32_t* out_idx,
                          int32_t* guard_out, int32_t* csr_status);
int dae_mix_topk_exact(dae_ctx* title_ctx, dae_ctx* dae, const float* feat, int64_t ld_feat, const float* h, int64_t ld_h,
                       int B, const float* w_title, const float* w_playlist, int n_tracks, const int32_t* seed_row_ptr,
                       const int32_t* seed_col, int k, float* out_score, int32_t* out_idx, int32_t* guard_out);

/* Training of the title variables (main_train.py:214-221 feeds the playlist as x AND y, titles_use = 1; the DAE
 * arrays are constants, DAEs.py:165-171).  The caller runs the forward pieces -- dae_encode (dropout on) +
 * dae_decode_dense for the DAE scores, dae_title_features(argmax, feat_raw) + dae_decode_dense(apply_sigmoid = 0)
 * for the title logits, dae_row_sums for x_count -- and then:
 *
 * dae_row_sums: out[r] = reduce_sum of DAEs.py:41, the row sums of the dropped-out input with dae_encode's draws.
 *
 * dae_title_loss_backward: cost = mean_rows(weighted BCE of the MIXED score) (DAEs.py:193-195) and the gradients
 *   of the output layer: gOutput_WT [V, ld] (the transposed layout the library keeps Output_W in), gOutput_b [V],
 *   and dfeat [B, ld] = d cost / d features (after dropout).  ld % 32 == 0, B <= 256.
 *
 * dae_title_conv_backward: dfeat back through dropout, max over time, ReLU, the convolutions and the embedding:
 *   g_emb [n_char, E], g_conv_w / g_conv_b in the layout of dae_title_features (overwritten).
 * Then dae_adam_step on each variable. */
int dae_row_sums(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B,
                 float input_keep_prob, uint32_t seed, float* out);
/* The mixing weights of DAE_title (DAEs.py:159-162) in one launch: x_count = (the row sum of dae_row_sums) * input_keep_prob,
 * deno = titles_use + x_count + 1e-10, w_title = titles_use / deno, w_playlist = x_count / deno -- fp32 operations in the
 * reference's order.  titles_use, w_title, w_playlist: device arrays [B]. */
int dae_mix_weights(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B, float input_keep_prob,
                    uint32_t seed, const float* titles_use, float* w_title, float* w_playlist);
int dae_title_loss_backward(dae_ctx* ctx, const float* title_logits, int64_t ld_z, const float* dae_score, int64_t ld_d,
                            const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                            const float* w_title, const float* w_playlist, int B, int V, int n_batch,
                            const float* feat, int ld, const float* Output_WT, float* gOutput_WT, float* gOutput_b,
                            float* dfeat, float* cost_out);
int dae_title_conv_backward(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char, int E,
                            const float* conv_w, const int32_t* filter_sizes, int n_sizes, int F,
                            const int32_t* argmax, const float* feat_raw, const float* dfeat, int64_t ld,
                            float keep_prob, uint32_t seed, float* g_emb, float* g_conv_w, float* g_conv_b);

/* TF1 AdamOptimizer update (DAEs.py:102): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
 * m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; p -= lr_t*m/(sqrt(v)+eps).  Dense over n elements.
 * t = 1-based step count. */
int dae_adam_step(dae_ctx* ctx, float* param, float* m, float* v, const float* grad,
                  int64_t n, float lr, float beta1, float beta2, float eps, int t);

/* The same dense Adam on a [n_rows, row_len] tensor whose gradient is ROW-SPARSE (the untied encoder: DAEs.py:66's
 * sparse_tensor_dense_matmul has a non-zero gradient only on the rows the batch's input names), without the seven
 * HBM passes over the rows that have none.  A row without gradient evolves by a recurrence nobody else reads, so its
 * (param, m, v) may stay at the step they were last current for and be brought up to date later by running the SAME
 * per-element update with g = 0 once per missed step, with the alpha of that step: the result is bit-identical to
 * calling dae_adam_step every step (tests/test_gpu_train.py).  Replaces nothing in the reference (TF1 applies the
 * dense update); it is how this build avoids 1.2 GB of traffic per step.
 *
 *   state   int32 [2 * n_rows], zero-initialised by the caller: the step each row is current for, and a claim mark
 *   lr_tab  float [tab_cap]: alpha of every step so far, written by dae_adam_rows_apply (entry t); t < tab_cap
 *   rows    int32 device list of the rows the coming / finished step touches (duplicates allowed, e.g. the column
 *           array of the input CSR); its length is read on the DEVICE from *n_listed_dev (e.g. &row_ptr[B]) when
 *           that pointer is not NULL, and is at most n_listed_max (which sizes the launch)
 *
 * Per step t = 1, 2, ...:
 *   dae_adam_rows_begin(rows of step t)   listed rows become current for step t - 1 -- BEFORE anything reads them
 *   ... forward / backward: the gradient rows land in the dense buffer `grad` (all other rows are zero) ...
 *   dae_adam_rows_apply(same rows)        listed rows take the update of step t; their gradient rows are zeroed
 *                                         again, so `grad` stays all-zero between steps (dae_set_enc_grad_prezeroed
 *                                         tells dae_train_forward_backward not to clear it)
 * and before anyone reads the WHOLE tensor (evaluation, saving, exchanging shards):
 *   dae_adam_rows_flush(t)                every row becomes current for step t */
int dae_adam_rows_begin(dae_ctx* ctx, float* param, float* m, float* v, int32_t* state, float* lr_tab, int tab_cap,
                        int n_rows, int row_len, const int32_t* rows, const int32_t* n_listed_dev, int n_listed_max,
                        float beta1, float beta2, float eps, int t);
int dae_adam_rows_apply(dae_ctx* ctx, float* param, float* m, float* v, float* grad, int32_t* state, float* lr_tab,
                        int tab_cap, int n_rows, int row_len, const int32_t* rows, const int32_t* n_listed_dev,
                        int n_listed_max, float lr, float beta1, float beta2, float eps, int t);
int dae_adam_rows_flush(dae_ctx* ctx, float* param, float* m, float* v, int32_t* state, const float* lr_tab,
                        int tab_cap, int n_rows, int row_len, float beta1, float beta2, float eps, int t);

/* Arms the NEXT dae_train_forward_backward of this context (untied model, reg_lambda = 0, H % 128 == 0) to apply the
 * dense Adam update of W_dec INSIDE the decoder-gradient kernel: each gradient tile is consumed in registers, W_dec
 * (which the call then writes through its const pointer), m and v are updated in place and gW_dec is not written
 * (it may be NULL).  The per-element operations are dae_adam_step's, so the parameters are bit-identical to writing
 * the gradient and calling dae_adam_step(W_dec, m, v, gW_dec, V*H, lr, ..., t); what disappears is one pass over the
 * gradient in each direction (1.2 GB -> 1.0 GB for Adam + the gradient at V = 170 000, H = 256).  m = NULL disarms. */
int dae_arm_decoder_adam(dae_ctx* ctx, float* m, float* v, float lr, float beta1, float beta2, float eps, int t);

/* on != 0: the untied gW_enc buffer handed to dae_train_forward_backward is all-zero on entry (kept so by
 * dae_adam_rows_apply), so the step does not clear its 4*V*H bytes.  Sticky. */
int dae_set_enc_grad_prezeroed(dae_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* DAE_HIP_H */
