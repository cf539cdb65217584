"""--pretrain / --dae driver (reference main_runner/main_train.py, INTENDED behaviour: the
snapshot does not run; citations relative to /root/reference).

Loop (main_train.py:193-253): next_batch -> input keep-prob ~ U(input_kp range) -> fair coin:
feed tracks-only or artists-only as x, tracks+artists as y -> one Adam step.  When the reader
wraps, evaluate r-precision on every test split (seed tracks in, top-500 out, main_train.py:48-100)
and save the parameters if the summed r-precision over `update_seed` did not get worse.
"""
import datetime
import os
import random

import numpy as np

from ..models.DAEs import DAE, DAE_tied, DAE_title
from ..models.title_models import get_model
from ..utils import metrics as met
from ..utils.data_reader import data_reader, data_reader_firstN, data_reader_test


def log_write(conf, log):
    if int(os.environ.get("RANK", "0")) != 0:
        return
    with open(os.path.join(conf.dir, 'log.txt'), "a") as f:
        f.write(log)
        f.write('\n')
    if getattr(conf, 'verbose', True):
        print(log)


def eval(reader_test, conf, model):
    """Mean r-precision over one test split (main_train.py:48-100): seed tracks only, both
    keep-probs 1.0, rank the track columns, drop the seeds, top 500."""
    total, count = 0.0, len(reader_test.playlists)
    answers = []

    def feeds():
        while True:
            x_positions, test_seed, test_answer, titles, x_ones = reader_test.next_batch_test()
            answers.append(test_answer)
            yield x_positions, x_ones, test_seed, titles
            if reader_test.test_idx == 0:
                break

    def results():
        if conf.mode == 'title':                                   # main_train.py:69-79: titles_use = 1 everywhere
            # ... for rows that HAVE a title (every row of a file this repo's or the reference's generator
            # writes).  A 5-field row (the snapshot reader's layout) carries none: mixing in the score of an
            # all-padding title would rank a 0-seed row on noise, so such rows take titles_use = 0.
            pad = [-1] * conf.strmaxlen
            from ..models.DAEs import SEEDS_FROM_INPUT

            def titled():
                for x_positions, x_ones, test_seed, titles in feeds():
                    use = np.array([0.0 if t is None else 1.0 for t in titles], np.float32)
                    yield (x_positions, x_ones, SEEDS_FROM_INPUT, len(test_seed),
                           [t if t is not None else pad for t in titles], use)
            for idx, _score in model.recommend_iter(titled(), k=500, want_scores=False):
                yield idx
        elif hasattr(model, 'recommend_iter'):
            # the evaluation seeds are the seed tracks the reader feeds (main_train.py:64-68): cut out of the input on
            # the device; batches streamed (upload / launch of batch n + 1 before the fetch of batch n)
            from ..models.DAEs import SEEDS_FROM_INPUT
            stream = ((xp, xo, SEEDS_FROM_INPUT, len(seed)) for xp, xo, seed, _t in feeds())
            for idx, _score in model.recommend_iter(stream, k=500, want_scores=False):
                yield idx
        else:
            for x_positions, x_ones, test_seed, _t in feeds():
                yield model.recommend(x_positions, x_ones, test_seed, k=500, n_rows=len(test_seed))[0]

    for b_no, idx in enumerate(results()):
        for i in range(len(idx)):
            total += met.eval_topk(idx[i], answers[b_no][i])
    return total / max(count, 1)


def _eval_splits(readers_test, conf, model, rank, world):
    """r-precision of every test split.  Under torch.distributed.run the splits are dealt round-robin to the ranks
    (every rank holds a full, freshly synchronised replica for inference) and the values meet in one all-reduce, so
    the evaluation of an epoch costs 1/world of what every-rank-evaluates-everything did."""
    names = list(readers_test)
    vals = [eval(readers_test[n], conf, model) if i % world == rank else 0.0 for i, n in enumerate(names)]
    if world > 1 and names:
        import torch
        import torch.distributed as dist
        t = torch.tensor(vals, dtype=torch.float64, device=torch.device("cuda", conf.device_index))
        dist.all_reduce(t)
        vals = t.tolist()
    return dict(zip(names, vals))


def _init_distributed(conf):
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the env):
    RCCL process group, this rank's device, and the SAME host RNG streams on every rank so that all
    ranks draw the same batches, coin flips and keep-probabilities (the batch is replicated, the
    vocabulary rows are sharded).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    conf.device_index = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(conf.device_index)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world)
    random.seed(int(getattr(conf, "seed", 0)))
    np.random.seed(int(getattr(conf, "seed", 0)))
    conf.verbose = getattr(conf, "verbose", True) and rank == 0
    return rank, world


def run(conf, only_testmode):
    rank, world = _init_distributed(conf)
    # the training reader (main_train.py:129-133): whole playlists, or -- with a [DAE] firstN_range -- their first-N prefixes
    reader_args = dict(data_dir=conf.data_dir, filename='train', batch_size=conf.batch)
    reader = data_reader(**reader_args) if -1 in conf.firstN else data_reader_firstN(from_to=conf.firstN, **reader_args)
    # what the models read off `conf` (main_train.py:134-148): the vocabulary and title shapes come from the data file
    for attr, value in (("class_divpnt", reader.class_divpnt), ("n_tracks", reader.num_tracks), ("n_input", reader.num_items),
                        ("n_output", reader.num_items), ("charsize", reader.num_char), ("strmaxlen", reader.max_title_len)):
        setattr(conf, attr, value)
    kp_range = conf.input_kp

    readers_test = {}
    for seed in conf.test_seed:
        path = os.path.join(conf.data_dir, seed)
        if not os.path.exists(path):
            log_write(conf, "test split %s not found, skipped" % path)
            continue
        readers_test[seed] = data_reader_test(data_dir=conf.data_dir, filename=seed,
                                              batch_size=conf.batch, test_num=conf.testsize)
    print(conf.n_input)

    if conf.mode == 'pretrain':                                     # main_train.py:154-161
        info, model = '[pretrain mode]', DAE_tied(conf)
    elif conf.mode == 'dae':
        if only_testmode:
            conf.initval = conf.save
        info, model = '[dae mode]', DAE(conf)
    elif conf.mode == 'title':                                      # main_train.py:162-165
        model_title = get_model(conf)
        model_title.fit()
        if only_testmode:
            model_title.load(conf.save + '.pkl')
        info, model = '[title mode]', DAE_title(conf, model_title)
    else:
        raise ValueError("unknown mode %r" % conf.mode)
    log_write(conf, '*' * 10)
    log_write(conf, info + ' start at ' + str(datetime.datetime.now()))
    model.fit()
    if world > 1 and not only_testmode and conf.mode != 'title':
        model.shard_training(rank, world)

    if only_testmode:                                               # main_train.py:181-191
        log_write(conf, '<<only test mode>>')
        out = _eval_splits(readers_test, conf, model, rank, world)
        for seed_num in readers_test:
            log_write(conf, "seed num: %s rprecision: %f" % (seed_num, out[seed_num]))
        return out

    epoch, it, loss, max_eval = 0, 0, 0.0, 0.0
    history = []
    while True:
        start_idx = reader.train_idx
        trk_positions, art_positions, y_positions, _titles, trk_val, art_val = reader.next_batch()
        end_idx = reader.train_idx
        input_kp = random.uniform(kp_range[0], kp_range[-1])        # main_train.py:199
        if conf.mode == 'title':                                    # :214-221: the whole playlist is input and target
            ones = np.ones(len(y_positions), np.float32)
            l = model.train_step(y_positions, ones, y_positions, ones, conf.kp, input_kp, titles=_titles,
                                 titles_use=np.ones(conf.batch, np.float32), title_keep_prob=conf.title_kp)
        else:
            if np.random.randint(2) == 0:                           # :202 fair coin
                x_pos, x_val = trk_positions, trk_val
            else:
                x_pos, x_val = art_positions, art_val
            # the cost stays on the device (a 0-dim tensor, summed there and fetched once per epoch) so the reader
            # builds the next batch while this step runs -- single GPU and vocabulary-sharded alike
            kw = {"fetch_cost": False}
            l = model.train_step(x_pos, x_val, y_positions, np.ones(len(y_positions), np.float32),
                                 conf.kp, input_kp, **kw)
        loss = loss + l
        it += 1
        if start_idx > end_idx or end_idx == 0:                     # :227 reader wrapped
            epoch += 1
            loss = float(loss)                                      # drains the stream
            if hasattr(model, "check_feed"):
                model.check_feed()
            log_write(conf, "epoch " + str(epoch))
            log_write(conf, "training loss: " + str(loss / it))
            cur_eval = 0.0
            model.sync_params()            # collective when sharded: every rank refreshes its replica first
            rprecs = _eval_splits(readers_test, conf, model, rank, world)
            for seed_num in readers_test:
                log_write(conf, "seed num: %s rprecision: %f" % (seed_num, rprecs[seed_num]))
                if seed_num in conf.update_seed:
                    cur_eval += rprecs[seed_num]
            history.append((epoch, loss / it, cur_eval))
            if cur_eval >= max_eval:                                # :243-249
                if rank == 0:
                    if conf.mode == 'title':
                        model.title_model.save(conf.save + '.pkl')   # the reference: saver.save(sess, conf.save)
                    else:
                        model.save_model()
                max_eval = cur_eval
                log_write(conf, "The highest score is updated. Parameters are saved")
            loss, it = 0.0, 0
            if epoch == conf.epochs:
                break
    return history
