"""Command line of the reference (main.py; citations relative to /root/reference), same flags and
the same config.ini schema:

    python -m spotify_recsys_challenge_2018_amd.main --dir D {--pretrain|--dae|--challenge} [--testmode]

    [BASE] verbose data_dir result_dir testsize
    [DAE] epochs batch lr reg_lambda hidden test_seed update_seed keep_prob input_kp firstN_range initval save
    [PRETRAIN] epochs batch lr reg_lambda save
    [TITLE] ... (parsed for compatibility; the title models are outside the scoring path)
    [CHALLENGE] batch challenge_data result

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


class Conf:
    """Plain attribute bag the drivers and models read (reference main.py:12-94)."""

    def __init__(self, dir, ini):
        self.dir = dir
        self.ini = ini
        base = ini['BASE']
        self.data_dir = base['data_dir']
        self.result_dir = base['result_dir']
        self.testsize = int(base['testsize'])
        self.verbose = _truth(base['verbose'])

    def set_dae_conf(self):
        s = self.ini['DAE']
        self.epochs, self.batch = int(s['epochs']), int(s['batch'])
        self.lr, self.reg_lambda = float(s['lr']), float(s['reg_lambda'])
        self.test_seed = ['test-' + t for t in _csv(s['test_seed'])]
        self.update_seed = ['test-' + t for t in _csv(s['update_seed'])]
        self.input_kp = _csv(s['input_kp'], float)
        self.kp = float(s['keep_prob'])
        self.firstN = _csv(s['firstN_range'], float)
        self._check_firstN(self.firstN)
        self.initval = os.path.join(self.dir, s['initval'])
        self.save = os.path.join(self.dir, s['save'])
        self.hidden = int(s['hidden'])
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
        s = self.ini['PRETRAIN']
        self.epochs, self.batch = int(s['epochs']), int(s['batch'])
        self.lr, self.reg_lambda = float(s['lr']), float(s['reg_lambda'])
        self.is_pretrain = True
        self.save = os.path.join(self.dir, s['save'])
        self.mode = 'pretrain'

    def set_title_conf(self):
        """[TITLE] (main.py:58-86).  Only the fields the scoring path needs are acted on: the
        title scorers (Char-CNN / Char-LSTM) are out of scope."""
        s = self.ini['TITLE']
        self.title_epochs, self.title_batch = int(s['epochs']), int(s['batch'])
        self.title_lr = float(s['lr'])
        self.title_input_kp = _csv(s['input_kp'], float)
        self.title_kp = s['title_kp']
        self.title_test_seed = ['test-' + t for t in _csv(s['test_seed'])]
        self.title_update_seed = ['test-' + t for t in _csv(s['update_seed'])]
        self.char_emb = int(s['char_emb'])
        self.char_model = s['char_model']
        if self.char_model == 'Char_CNN':
            self.filter_num = int(s['filter_num'])
            self.filter_size = _csv(s['filter_size'], int)
        self.DAEval = os.path.join(self.dir, s['DAEval'])
        self.title_save = os.path.join(self.dir, s['save'])

    def set_challenge_oonf(self):          # (sic) reference spelling, main.py:88
        if not os.path.isdir(self.result_dir):
            os.mkdir(self.result_dir)
        s = self.ini['CHALLENGE']
        self.challenge_data = s['challenge_data']
        self.result = os.path.join(self.result_dir, s['result'])
        self.batch = int(s['batch'])

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
        raise SystemExit("--title trains the character CNN on top of a frozen DAE; the title models "
                         "are outside the DAE scoring path this package implements (SURVEY.md 8f)")
    elif args.challenge:
        conf.set_title_conf()
        conf.set_challenge_oonf()
        main_challenge.run(conf)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
