"""--challenge driver (reference main_runner/main_challenge.py; citations relative to
/root/reference): score every challenge playlist, write [pid, 500 track URIs] rows as a pickle.

Reference hot loop (main_challenge.py:72-93) per batch: COO build -> sess.run(y_pred) of the whole
[batch, n_input] matrix -> host slice to tracks -> per row argsort + list.remove + [:500].
Here: the raw COO feed goes to the device, the CSR and the seed lists are built there (dae_coo_to_csr,
dae_seeds_from_csr), ONE library call (dae_score_topk) encodes, decodes and ranks; only [batch, 500]
indices come back, batches streamed through `model.recommend_iter`.

Titles: the reference mixes a character-CNN title score into y_pred (DAE_title, DAEs.py:176-181;
weights restored from a TF checkpoint, main_challenge.py:68-69).  Here the title variables come from this
package's own pickle `<[TITLE] save>.pkl` (models/title_models.py; a TF checkpoint cannot be read in this
environment).  When that file exists, batches that carry titles are scored by DAE_title -- the mix fused into
the threshold path (dae_decode_mix_term + dae_set_score_mix: no [batch, n_input] matrix of either scorer).
When it does not, the run fails like the reference's restore does, unless [CHALLENGE] allow_no_title = True
(optional key of this build) asks for the plain DAE: titles_use = 0 reduces the mix exactly to it (App. B.6)
on the weights the [TITLE] section names (DAEval); playlists without seeds still make that an error.
"""
import datetime
import os
import pickle

from ..utils.data_reader import data_reader_challenge


def log_write(conf, log):
    """main_challenge.py:17-23 (rank 0 only under torch.distributed.run)."""
    if int(os.environ.get("RANK", "0")) != 0:
        return
    with open(os.path.join(conf.dir, 'log.txt'), "a") as f:
        f.write(log)
        f.write('\n')
    if conf.verbose:
        print(log)


def _init_distributed(conf):
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the env): the RCCL
    process group (backend "nccl"; DAE_DIST_BACKEND=gloo rehearses the flow on CPU or with several ranks on one
    GPU) and this rank's device.  Returns (rank, world).  BASELINE.json configs[2]: the vocabulary columns are
    sharded over the ranks, every rank reads the same challenge file and feeds the same batches."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    backend = os.environ.get("DAE_DIST_BACKEND", "nccl")
    if torch.cuda.is_available():
        n_dev = torch.cuda.device_count()
        conf.device_index = int(os.environ.get("LOCAL_RANK", rank)) % max(n_dev, 1) if backend == "gloo" \
            else int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(conf.device_index)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def cand_to_uris(cand_idx, id2uri):
    """main_challenge.py:37-41: ranked track ids -> 'spotify:track:<uri>' strings."""
    return ['spotify:track:' + id2uri[str(int(i))] for i in cand_idx if i >= 0]


class UriTable:
    """The same mapping for whole batches: the 'spotify:track:<uri>' string of every track id built ONCE (an object
    array), a batch of ranked ids turned into rows by one fancy index.  The per-id str() / dict lookup / concatenation of
    `cand_to_uris` was 42 % of a full-size --challenge run (scripts/bench_challenge.py), and because every occurrence of
    a track is now the same str object, pickle writes it once and refers back to it (the result file loads to equal
    lists)."""

    def __init__(self, id2uri, n_tracks):
        import numpy as np
        self.table = np.empty(n_tracks, dtype=object)
        self.missing = np.zeros(n_tracks, dtype=bool)
        for i in range(n_tracks):
            u = id2uri.get(str(i))
            self.missing[i] = u is None
            self.table[i] = None if u is None else 'spotify:track:' + u
        self.any_missing = bool(self.missing.any())

    def rows(self, idx):
        """idx [rows, k] ranked ids (-1 = padding of a short list) -> list of lists of URI strings."""
        import numpy as np
        idx = np.asarray(idx)
        if self.any_missing and idx.size and self.missing[idx[idx >= 0]].any():
            raise KeyError("a ranked track id has no entry in id2uri")
        if idx.size and idx.min() >= 0:
            return self.table[idx].tolist()
        return [self.table[r[r >= 0]].tolist() for r in idx]


def run(conf, model=None):
    """`model`: None -> built from conf (DAE / DAE_title on the GPU); tests pass their own object with the same
    `recommend` / `shard_scoring` / `owned_rows` protocol."""
    rank, world = _init_distributed(conf)
    reader = data_reader_challenge(data_dir=conf.data_dir, filename=conf.challenge_data,
                                   batch_size=conf.batch)
    conf.n_tracks = reader.num_tracks                      # main_challenge.py:49-53
    conf.n_input = reader.num_items
    conf.n_output = reader.num_items
    conf.charsize = reader.num_char
    conf.strmaxlen = reader.max_title_len
    print(conf.n_input)

    log_write(conf, '*' * 10)
    log_write(conf, '[challenge mode] start at ' + str(datetime.datetime.now()))

    use_titles = False
    exchange = getattr(conf, 'shard_exchange', 'allgather')
    tau_x = bool(getattr(conf, 'shard_tau_exchange', True))
    if model is None:
        from ..models.DAEs import DAE, DAE_title
        from ..models.title_models import get_model
        conf.initval = getattr(conf, 'DAEval', conf.initval)   # weights the TITLE stage froze
        title_pkl = str(getattr(conf, 'title_save', '')) + '.pkl'
        if getattr(conf, 'char_model', None) == 'Char_CNN' and os.path.exists(title_pkl):
            model_title = get_model(conf)                       # main_challenge.py:58-59
            model_title.fit()
            model_title.load(title_pkl)
            model = DAE_title(conf, model_title)
            use_titles = True
            log_write(conf, 'title scorer: ' + title_pkl)
        else:
            # The reference fails here (saver.restore of a missing checkpoint, main_challenge.py:68-69).  A silent
            # plain-DAE run would turn a mistyped path into a submission without title mixing -- and on playlists
            # without seed tracks every row would get the same popularity ranking -- so it needs an explicit opt-in:
            # optional key of this build, [CHALLENGE] allow_no_title = True.
            seedless = sum(1 for p in reader.playlists if not p[0] and not p[1])
            if not getattr(conf, 'allow_no_title', False) or seedless:
                raise FileNotFoundError(
                    "title variables %s not found (train them with --title)%s; set [CHALLENGE] allow_no_title = True "
                    "to score with the plain DAE (titles_use = 0)" % (
                        title_pkl, "; %d playlists of %s have no seeds and can only be ranked by title"
                        % (seedless, conf.challenge_data) if seedless else ""))
            model = DAE(conf)
            log_write(conf, 'no title variables at %s: plain DAE (titles_use = 0), allow_no_title is set' % title_pkl)
        sharded = world > 1 and not use_titles
        if sharded:
            # optional key of this build, [CHALLENGE] shard_exchange = allgather (default; configs[2] as written:
            # every rank ends with every row) | alltoall (every rank ends with the rows it owns)
            model.shard_scoring(rank, world, exchange=exchange, tau_exchange=tau_x)
        model.fit()
    else:
        use_titles = bool(getattr(model, 'title_model', None))
        sharded = world > 1 and not use_titles
        if sharded:
            if exchange == 'alltoall' and model.n_batch % world and getattr(model, 'ctx', None) is not None:
                # shard_scoring can only round n_batch up BEFORE fit(): say so here instead of failing inside it
                raise ValueError("[CHALLENGE] shard_exchange = alltoall needs the model's batch (%d) to be a multiple of the "
                                 "world size (%d) when a fitted model is passed in: build the model with such a batch, or "
                                 "use shard_exchange = allgather" % (model.n_batch, world))
            import inspect
            if 'tau_exchange' in inspect.signature(model.shard_scoring).parameters:
                model.shard_scoring(rank, world, exchange=exchange, tau_exchange=tau_x)
            else:                                              # a model object of the older protocol
                model.shard_scoring(rank, world, exchange=exchange)
                tau_x = False                                  # (what the log line below reports)
    if sharded:
        log_write(conf, 'vocabulary columns sharded over %d ranks (%s exchange of the per-shard top-500%s)'
                  % (world, exchange, ', thresholds exchanged first' if tau_x else ''))
    elif world > 1:
        # title-mixed scores are not vocabulary-sharded: the batches are dealt round-robin to the ranks instead
        # (whole model on every GPU, no collective in the data path)
        log_write(conf, 'title model: batches partitioned over %d ranks' % world)

    cands = []                  # (position in the file, [pid, uris...]) of the rows THIS rank holds results for
    # The challenge seeds ARE the playlist's tracks (reader: seed = playlists[i][0], the ids x_positions feeds), so the
    # model cuts the seed lists out of the input on the device; batches are streamed through recommend_iter (upload and
    # launch of batch n + 1 before the results of batch n are fetched).
    from ..models.DAEs import SEEDS_FROM_INPUT
    meta = []                   # per submitted batch: (first position in the file, pids, rows in the batch)

    def feeds():
        n_seen, n_batches = 0, 0
        while True:
            x_positions, seed, titles, titles_exist, pid, x_ones = reader.next_batch()
            mine = sharded or world == 1 or n_batches % world == rank
            if mine:
                meta.append((n_seen, pid, len(seed)))
                if getattr(reader, "last_titles", None) is not None:      # this build's reader: the same titles as arrays
                    titles, titles_exist = reader.last_titles, reader.last_titles_use.reshape(-1, 1)
                yield x_positions, x_ones, seed, titles, titles_exist
            n_seen += len(seed)
            n_batches += 1
            if reader.ch_idx == 0:
                break

    def results():
        if use_titles and hasattr(model, 'recommend_iter'):
            stream = ((xp, xo, SEEDS_FROM_INPUT, len(seed), t, [e[0] for e in te]) for xp, xo, seed, t, te in feeds())
            for idx, _score in model.recommend_iter(stream, k=500, want_scores=False):
                yield idx
        elif use_titles:
            for x_positions, x_ones, seed, titles, titles_exist in feeds():
                yield model.recommend(x_positions, x_ones, seed, k=500, n_rows=len(seed), titles=titles,
                                      titles_use=[t[0] for t in titles_exist])[0]
        elif hasattr(model, 'recommend_iter'):
            stream = ((xp, xo, SEEDS_FROM_INPUT, len(seed)) for xp, xo, seed, _t, _e in feeds())
            for idx, _score in model.recommend_iter(stream, k=500, want_scores=False):
                yield idx
        else:                   # a model object without the streaming entry point (tests)
            for x_positions, x_ones, seed, _t, _e in feeds():
                yield model.recommend(x_positions, x_ones, seed, k=500, n_rows=len(seed))[0]

    uris = UriTable(reader.id2uri, reader.num_tracks)
    # one full collection now and the survivors frozen: the loop allocates a few tracked objects per batch, and the
    # generation-2 collection they eventually trigger walks everything torch and the model loaded (33 - 38 ms measured in
    # the middle of a 70 ms scoring loop: profiles/r05_notes.md)
    import gc
    gc.collect()
    gc.freeze()
    for b_no, idx in enumerate(results()):
        first, pid, n_in_batch = meta[b_no]
        if sharded and exchange == 'allgather' and rank != 0:        # all-gather: rank 0 formats and writes
            continue
        r0, _r1 = model.owned_rows() if sharded else (0, n_in_batch)
        for j, row in enumerate(uris.rows(idx)):            # rows r0 .. of the batch (all of them unless alltoall)
            i = r0 + j
            cands.append((first + i, [pid[i]] + row))

    if world > 1:
        import torch.distributed as dist
        if not (sharded and exchange == 'allgather'):
            parts = [None] * world if rank == 0 else None
            dist.gather_object(cands, parts, dst=0)         # every rank's rows -> rank 0, back in file order
            if rank == 0:
                cands = sorted((c for part in parts for c in part), key=lambda c: c[0])
        dist.barrier()
    total_cands = [c for _pos, c in cands]
    if rank == 0:
        with open(conf.result, 'wb') as f:                 # main_challenge.py:95-96
            pickle.dump(total_cands, f)
        log_write(conf, 'wrote %d playlists to %s' % (len(total_cands), conf.result))
    return total_cands
