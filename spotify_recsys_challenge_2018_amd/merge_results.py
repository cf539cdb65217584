"""Reference merge_results.py (citations relative to /root/reference): concatenate the per-category
pickles that `main.py --challenge` wrote (main_challenge.py:95-96: lists of [pid, 500 track URIs])
under --dir into the submission file results.csv, first row = the team line (merge_results.py:13).

    python -m spotify_recsys_challenge_2018_amd.merge_results --dir challenge_results [--out results.csv]

The reference goes through pandas (`DataFrame(total_cands).to_csv(index=False, header=False)`), which
pads ragged rows with empty cells; the csv module writes the same bytes without the dependency
(tests/test_cli_cpu.py compares against pandas).  Files are taken in sorted order so the output is
reproducible (os.listdir order, which the reference uses, is file-system dependent)."""
import argparse
import csv
import os
import pickle

TEAM_ROW = ['team_info', 'track', 'team_name', 'email@address.com']      # merge_results.py:13


def merge(directory, out_path='results.csv', team_row=None):
    total = [list(team_row if team_row is not None else TEAM_ROW)]
    for name in sorted(os.listdir(directory)):
        with open(os.path.join(directory, name), 'rb') as f:
            total += pickle.load(f)                                      # merge_results.py:16-19
    width = max(len(r) for r in total)
    with open(out_path, 'w', newline='') as f:
        w = csv.writer(f, lineterminator='\n')
        for r in total:
            w.writerow(list(r) + [''] * (width - len(r)))
    return total


def main(argv=None):
    ap = argparse.ArgumentParser(description="args")
    ap.add_argument('--dir', type=str, default='challenge_results')
    ap.add_argument('--out', type=str, default='results.csv')
    args = ap.parse_args(argv)
    total = merge("./" + args.dir, args.out)
    print("num_playlist: ", len(total) - 1)                              # merge_results.py:21-22
    print("num_rec: ", (len(total[1]) - 1) if len(total) > 1 else 0)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
