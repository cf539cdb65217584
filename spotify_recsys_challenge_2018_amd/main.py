"""Command line of the reference (main.py; citations relative to /root/reference), same flags and
the same config.ini schema:

    python -m spotify_recsys_challenge_2018_amd.main --dir D {--pretrain|--dae|--challenge} [--testmode]

    [BASE] verbose data_dir result_dir testsize   (+ optional, this build only: train_dtype = f32 | bf16, decode_dtype = f32 | bf16 | exact_bf16)
    [DAE] epochs batch lr reg_lambda hidden test_seed update_seed keep_prob input_kp firstN_range initval save
    [PRETRAIN] epochs batch lr reg_lambda save
    [TITLE] ... (parsed for compatibility; the title models are outside the scoring path)
    [CHALLENGE] batch challenge_data result   (+ optional, this build only: allow_no_title, shard_exchange, shard_tau_exchange)

Kept on purpose: `[DAE]` is always read first, so --pretrain inherits hidden / keep_prob /
input_kp / firstN_range / test_seed from it (main.py:121, load-bearing per SURVEY App. A).
Repaired: booleans are parsed ("False" is false; main.py:19 uses bool(str)).
"""
import argparse
import configparser
import os


def _csv(text, cast=str):
    return [cast(t.strip()) for t in text.split(',')]


def _truth(text):
    return str(text).strip().lower() in ("1", "true", "yes", "on")


def _seeds(text):
    return ['test-' + t for t in _csv(text)]


def _floats(text):
    return _csv(text, float)


def _ints(text):
    return _csv(text, int)


# config.ini schema (reference main.py:12-94) as data: section -> (attribute, key, cast).
# "@dir" / "@result" casts join the value onto the run directory / the result directory.
_SCHEMA = {
    'BASE': (('data_dir', 'data_dir', str), ('result_dir', 'result_dir', str), ('testsize', 'testsize', int),
             ('verbose', 'verbose', _truth)),
    'DAE': (('epochs', 'epochs', int), ('batch', 'batch', int), ('lr', 'lr', float),
            ('reg_lambda', 'reg_lambda', float), ('test_seed', 'test_seed', _seeds),
            ('update_seed', 'update_seed', _seeds), ('input_kp', 'input_kp', _floats), ('kp', 'keep_prob', float),
            ('firstN', 'firstN_range', _floats), ('initval', 'initval', '@dir'), ('save', 'save', '@dir'),
            ('hidden', 'hidden', int)),
    'PRETRAIN': (('epochs', 'epochs', int), ('batch', 'batch', int), ('lr', 'lr', float),
                 ('reg_lambda', 'reg_lambda', float), ('save', 'save', '@dir')),
    # [TITLE] OVERRIDES epochs / batch / lr / input_kp / test_seed / update_seed / save of [DAE] (main.py:59-83),
    # for --title and, as in the reference, for --challenge as well; keep_prob stays the [DAE] one
    'TITLE': (('epochs', 'epochs', int), ('batch', 'batch', int), ('lr', 'lr', float), ('title_lr', 'lr', float),
              ('input_kp', 'input_kp', _floats), ('title_kp', 'title_kp', float),
              ('test_seed', 'test_seed', _seeds), ('update_seed', 'update_seed', _seeds),
              ('char_emb', 'char_emb', int), ('char_model', 'char_model', str), ('DAEval', 'DAEval', '@dir'),
              ('save', 'save', '@dir'), ('title_save', 'save', '@dir')),
    'CHALLENGE': (('challenge_data', 'challenge_data', str), ('result', 'result', '@result'),
                  ('batch', 'batch', int)),
}


class Conf:
    """Plain attribute bag the drivers and models read."""

    def __init__(self, dir, ini):
        self.dir = dir
        self.ini = ini
        self._load('BASE')
        # two OPTIONAL [BASE] keys this build adds (absent from the reference's files, which then mean fp32):
        #   train_dtype  = f32 | bf16   arithmetic of the training step's three GEMMs (BASELINE.json configs[3])
        #   decode_dtype = f32 | bf16 | exact_bf16   arithmetic of the scoring decode: fp32 MFMA (bit-exact path), bf16
        #                  (configs[4]), or the bf16 GEMM as a filter with the survivors recomputed in fp32 (north_star:
        #                  the fp32 lists, bit for bit)
        for key, allowed in (('train_dtype', ('f32', 'bf16')), ('decode_dtype', ('f32', 'bf16', 'exact_bf16'))):
            if key in self.ini['BASE']:
                val = self.ini['BASE'][key].strip().lower()
                if val not in allowed:
                    raise ValueError("[BASE] %s must be one of %s, not %r" % (key, ' | '.join(allowed), val))
                setattr(self, key, val)

    def _load(self, section):
        sec = self.ini[section]
        for attr, key, cast in _SCHEMA[section]:
            raw = sec[key]
            if cast == '@dir':
                val = os.path.join(self.dir, raw)
            elif cast == '@result':
                val = os.path.join(self.result_dir, raw)
            else:
                val = cast(raw)
            setattr(self, attr, val)
        return sec

    def set_dae_conf(self):
        self._load('DAE')
        self._check_firstN(self.firstN)
        self.mode = 'dae'

    @staticmethod
    def _check_firstN(rng):
        """The reference's range rules (main.py:35-43): a single -1 disables firstN; fractions
        must stay below 1 on both ends; counts must be integers >= 1."""
        if len(rng) == 1:
            assert rng[0] == -1.0
            return
        lo, hi = rng[0], rng[1]
        assert lo <= hi
        if hi < 1:
            assert lo == 0 or not float(lo).is_integer()
        else:
            assert lo >= 1 and float(lo).is_integer() and float(hi).is_integer()

    def set_pretrain_conf(self):
        self._load('PRETRAIN')
        self.is_pretrain = True
        self.mode = 'pretrain'

    def set_title_conf(self):
        """[TITLE] (main.py:58-86): the character-CNN title scorer trained on top of the frozen DAE `DAEval`.
        The variables are saved as `<save>.pkl` (the reference writes a TF checkpoint at `save`)."""
        sec = self._load('TITLE')
        if self.char_model == 'Char_CNN':
            self.filter_num = int(sec['filter_num'])
            self.filter_size = _ints(sec['filter_size'])
        os.makedirs(os.path.dirname(self.save) or '.', exist_ok=True)      # main.py:80-82
        self.mode = 'title'

    def set_challenge_oonf(self):          # (sic) the reference's spelling, main.py:88
        os.makedirs(self.result_dir, exist_ok=True)
        sec = self._load('CHALLENGE')
        # three OPTIONAL [CHALLENGE] keys this build adds (absent from the reference's files):
        #   allow_no_title = True        score with the plain DAE when <[TITLE] save>.pkl is missing (default: error --
        #                                the reference fails there too; also an error while any playlist has no seeds)
        #   shard_exchange = allgather | alltoall    under torch.distributed.run: how the per-shard top-500 meet
        #   shard_tau_exchange = False   ... without the exchange of the shards' thresholds before their filter launches
        #                                (default True: one more collective of 4 bytes per row and rank, same result,
        #                                the ranks holding unpopular tracks stop keeping 10x the candidates)
        if 'allow_no_title' in sec:
            self.allow_no_title = _truth(sec['allow_no_title'])
        if 'shard_tau_exchange' in sec:
            self.shard_tau_exchange = _truth(sec['shard_tau_exchange'])
        if 'shard_exchange' in sec:
            val = sec['shard_exchange'].strip().lower()
            if val not in ('allgather', 'alltoall'):
                raise ValueError("[CHALLENGE] shard_exchange must be allgather or alltoall, not %r" % val)
            self.shard_exchange = val

    set_challenge_conf = set_challenge_oonf


def load_conf(dir):
    ini = configparser.ConfigParser()
    ini.read(os.path.join(dir, 'config.ini'))
    return Conf(dir, ini)


def build_parser():
    ap = argparse.ArgumentParser(description="args")
    ap.add_argument('--dir', type=str, default='qwerty', help="directory name which contains config file")
    ap.add_argument('--pretrain', action='store_true', default=False, help="pretrain mode if Specified")
    ap.add_argument('--dae', action='store_true', default=False, help="DAE mode if Specified")
    ap.add_argument('--title', action='store_true', default=False, help="title mode if Specified")
    ap.add_argument('--challenge', action='store_true', default=False, help="challenge mode if Specified")
    ap.add_argument('--testmode', action='store_true', default=False,
                    help="test mode if Specified(just check the result)")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    dir = os.path.join(".", args.dir)
    if not os.path.isdir(dir):
        print("ERROR: Cannot find " + dir + " ->Create directory and config.ini file first")
        return 0
    if 'config.ini' not in os.listdir(dir):
        print("ERROR: Cannot find config.ini in " + dir + " ->Create config.ini file in the directory first")
        return 0
    conf = load_conf(dir)
    conf.set_dae_conf()                                   # always first (main.py:121)
    from .main_runner import main_challenge, main_train
    if args.pretrain:
        conf.set_pretrain_conf()
        main_train.run(conf, args.testmode)
    elif args.dae:
        conf.set_dae_conf()
        main_train.run(conf, args.testmode)
    elif args.title:
        conf.set_title_conf()
        main_train.run(conf, args.testmode)
    elif args.challenge:
        conf.set_title_conf()
        conf.set_challenge_oonf()
        main_challenge.run(conf)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
