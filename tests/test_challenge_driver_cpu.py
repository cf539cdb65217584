"""CPU, world_size 2 over gloo: the PRODUCT's --challenge driver (main_runner/main_challenge.py) with the
vocabulary columns sharded over the ranks (BASELINE.json configs[2]) on the golden challenge file -- the result
pickle must equal the 1-rank run's, for both exchanges.  The device is replaced by an oracle-backed object with the
model protocol the driver uses (recommend / shard_scoring / owned_rows); the exchange, the row ownership, the
rank-0 write and the gather of owned rows are the product's own code (sharding.ShardedRanker is what
DAE.shard_scoring builds as well).  The GPU counterpart (real kernels, two ranks on one device over gloo) is
tests/test_gpu_sharded_scoring.py."""
import os
import pickle
import shutil

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.sharding import ShardedRanker, row_owner_bounds, scoring_shard
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_weights

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class OracleDAE:
    """Stands in for models.DAEs.DAE on a machine without a GPU: same protocol, oracle arithmetic."""
    title_model = None

    def __init__(self, n_batch, n_input, n_tracks, hidden=32):
        self.n_batch, self.n_input, self.n_tracks = n_batch, n_input, n_tracks
        self.W_enc, self.b_enc, self.W_dec, self.b_dec = make_weights(n_input, hidden, seed=4, bias="zipf",
                                                                      n_tracks=n_tracks)
        self.shard = None

    def shard_scoring(self, rank, world, group=None, exchange="allgather"):
        if exchange == "alltoall" and self.n_batch % world:
            self.n_batch += world - self.n_batch % world
        lo, hi = scoring_shard(self.n_tracks, self.n_input, world, rank)[0]      # the rank's slice of the TRACK columns
        self.shard = (rank, world, exchange, lo, hi)

    def owned_rows(self):
        if self.shard is None or self.shard[2] != "alltoall":
            return 0, self.n_batch
        return row_owner_bounds(self.n_batch, self.shard[1], self.shard[0])

    def recommend(self, x_positions, x_ones, seeds, k=500, n_rows=None):
        B = self.n_batch
        rp, col, val = coo_to_csr(x_positions, x_ones, B, self.n_input)
        srp, sc = seeds_to_csr(seeds, B, self.n_tracks)
        h = oracle.encode(rp, col, val, self.W_enc, self.b_enc)
        n_rows = B if n_rows is None else n_rows
        if self.shard is None:
            s, i = oracle.topk(oracle.decode(h, self.W_dec, self.b_dec, 0, self.n_tracks), k, srp, sc)
            return i[:n_rows], s[:n_rows]
        _rank, _world, exchange, lo, hi = self.shard

        def local_topk(_feed, kk):
            if hi <= lo:            # a shard of artist columns only: nothing rankable
                return (torch.full((B, kk), -np.inf), torch.full((B, kk), -1, dtype=torch.int32))
            s, i = oracle.topk(oracle.decode(h, self.W_dec, self.b_dec, lo, hi), kk, srp, sc, col_base=lo, out_kind=1)
            return torch.from_numpy(s), torch.from_numpy(i)

        def merge(gl, gi):
            s, i = oracle.topk_merge(gl.numpy(), gi.numpy())
            return torch.from_numpy(s), torch.from_numpy(i)
        s, i = ShardedRanker(local_topk, merge, exchange=exchange).rank_batch(None, k)
        r0, r1 = self.owned_rows()
        n_own = max(0, min(r1, n_rows) - r0)
        return i[:n_own].numpy(), s[:n_own].numpy()


def _conf(tmp, exchange, batch):
    from spotify_recsys_challenge_2018_amd import main as cli
    run = os.path.join(tmp, "run_%s_%d" % (exchange or "one", batch))
    os.makedirs(run, exist_ok=True)
    ini = open(os.path.join(G, "config.ini")).read().replace("./data", os.path.join(G, "data"))
    ini = ini.replace("./challenge_results", os.path.join(run, "res"))
    ini = ini.replace("[CHALLENGE]\nbatch = 5", "[CHALLENGE]\nbatch = %d" % batch)
    if exchange:
        ini = ini.replace("[CHALLENGE]", "[CHALLENGE]\nshard_exchange = %s" % exchange)
    open(os.path.join(run, "config.ini"), "w").write(ini)
    conf = cli.load_conf(run)
    conf.set_dae_conf(); conf.set_title_conf(); conf.set_challenge_oonf()
    return conf


def _run(conf):
    import io
    from contextlib import redirect_stdout
    from spotify_recsys_challenge_2018_amd.main_runner import main_challenge
    from spotify_recsys_challenge_2018_amd.utils.data_reader import data_reader_challenge
    with redirect_stdout(io.StringIO()):
        rd = data_reader_challenge(conf.data_dir, conf.challenge_data, conf.batch)
        model = OracleDAE(conf.batch, rd.num_items, rd.num_tracks)
        return main_challenge.run(conf, model=model)


def _worker(rank, world, port, tmp, exchange, batch):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DAE_DIST_BACKEND="gloo")
    _run(_conf(tmp, exchange, batch))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,batch", [("allgather", 5), ("alltoall", 5), ("alltoall", 4)])
def test_two_rank_challenge_driver_writes_the_one_rank_pickle(tmp_path, exchange, batch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    one = _conf(str(tmp_path), None, batch)
    want = _run(one)
    assert len(want) == 13 and pickle.load(open(one.result, "rb")) == want
    assert all(50 <= len(row) - 1 <= 500 and len(set(row[1:])) == len(row) - 1 for row in want)   # < 500 tracks exist
    ctx = mp.get_context("spawn")
    port = 29500 + (os.getpid() * 7 + len(exchange) + batch) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), exchange, batch)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    two = _conf(str(tmp_path), exchange, batch)
    got = pickle.load(open(two.result, "rb"))
    assert got == want
    log = open(os.path.join(two.dir, "log.txt")).read()
    assert "sharded over 2 ranks (%s exchange" % exchange in log and log.count("wrote 13 playlists") == 1


def test_missing_title_variables_need_an_explicit_opt_in(tmp_path):
    """a mistyped [TITLE] save must not silently produce a submission without title mixing."""
    from spotify_recsys_challenge_2018_amd.main_runner import main_challenge
    conf = _conf(str(tmp_path), None, 5)
    with pytest.raises(FileNotFoundError, match="allow_no_title"):
        main_challenge.run(conf)              # model=None -> the product path; stops before any device work


def test_uri_table_equals_the_per_row_mapping():
    """The batched id -> URI mapping (one fancy index per batch) gives the rows main_challenge.py:37-41 builds one id at
    a time, including short lists padded with -1; the result pickle loads to the same lists."""
    import io
    from spotify_recsys_challenge_2018_amd.main_runner.main_challenge import UriTable, cand_to_uris
    rng = np.random.default_rng(0)
    nt = 300
    id2uri = {str(i): "u%05d" % (7 * i) for i in range(nt)}
    t = UriTable(id2uri, nt)
    full = rng.integers(0, nt, (9, 40))
    assert t.rows(full) == [cand_to_uris(r, id2uri) for r in full]
    short = full.copy(); short[2, 25:] = -1; short[5, :] = -1
    assert t.rows(short) == [cand_to_uris(r, id2uri) for r in short]
    assert t.rows(np.zeros((0, 40), np.int32)) == []
    rows = [[11] + r for r in t.rows(full)]
    assert pickle.loads(pickle.dumps(rows)) == [[11] + cand_to_uris(r, id2uri) for r in full]
    with pytest.raises(KeyError):
        UriTable({"0": "a"}, 2).rows(np.array([[1]]))
