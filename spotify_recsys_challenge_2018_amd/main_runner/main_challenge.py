"""--challenge driver (reference main_runner/main_challenge.py; citations relative to
/root/reference): score every challenge playlist, write [pid, 500 track URIs] rows as a pickle.

Reference hot loop (main_challenge.py:72-93) per batch: COO build -> sess.run(y_pred) of the whole
[batch, n_input] matrix -> host slice to tracks -> per row argsort + list.remove + [:500].
Here: COO -> CSR on the host, then ONE library call (dae_score_topk) that encodes, decodes and
ranks on the GPU; only [batch, 500] indices come back.

Titles: the reference mixes a character-CNN title score into y_pred (DAE_title, DAEs.py:176-181;
weights restored from a TF checkpoint, main_challenge.py:68-69).  Here the title variables come from this
package's own pickle `<[TITLE] save>.pkl` (models/title_models.py; a TF checkpoint cannot be read in this
environment).  When that file exists, batches that carry titles are scored by DAE_title (mixed, unfused);
when it does not, or for rows without a title, titles_use = 0 and the mix reduces exactly to the plain DAE
(App. B.6) on the weights the [TITLE] section names (DAEval), through the fused path.
"""
import datetime
import os
import pickle

from ..models.DAEs import DAE, DAE_title
from ..models.title_models import get_model
from ..utils.data_reader import data_reader_challenge


def log_write(conf, log):
    """main_challenge.py:17-23."""
    with open(os.path.join(conf.dir, 'log.txt'), "a") as f:
        f.write(log)
        f.write('\n')
    if conf.verbose:
        print(log)


def cand_to_uris(cand_idx, id2uri):
    """main_challenge.py:37-41: ranked track ids -> 'spotify:track:<uri>' strings."""
    return ['spotify:track:' + id2uri[str(int(i))] for i in cand_idx if i >= 0]


def run(conf, model=None):
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
    if model is None:
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
            model = DAE(conf)
            log_write(conf, 'no title variables at %s: plain DAE (titles_use = 0)' % title_pkl)
        model.fit()
    else:
        use_titles = isinstance(model, DAE_title)

    total_cands = []
    while True:
        x_positions, seed, titles, titles_exist, pid, x_ones = reader.next_batch()
        if use_titles:
            idx, _score = model.recommend(x_positions, x_ones, seed, k=500, n_rows=len(seed), titles=titles,
                                          titles_use=[t[0] for t in titles_exist])
        else:
            idx, _score = model.recommend(x_positions, x_ones, seed, k=500, n_rows=len(seed))
        for i in range(len(seed)):
            total_cands.append([pid[i]] + cand_to_uris(idx[i], reader.id2uri))
        if reader.ch_idx == 0:
            break

    with open(conf.result, 'wb') as f:                     # main_challenge.py:95-96
        pickle.dump(total_cands, f)
    log_write(conf, 'wrote %d playlists to %s' % (len(total_cands), conf.result))
    return total_cands
