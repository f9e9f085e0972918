"""`local-scores` of the reference's CLI (eigentrust-cli/src/cli.rs:458-517) on this engine:

    python -m protocol_amd.cli local-scores [--assets DIR] [--domain 0x...]

reads DIR/attestations.csv, recovers and verifies the signatures, converges on the GPU and writes
DIR/scores.csv in the reference's format.  `--large` lifts the slot limit (no rational columns then:
numerator = floor score is not available, so the file holds peer_address,score_fr only).
`threshold` re-checks one participant of an existing scores.csv against a threshold
(eigentrust/src/lib.rs:470-510 without the proof).
"""
import argparse
import csv
import os
import sys
from fractions import Fraction

from . import scores as sc


def _domain(text):
    raw = bytes.fromhex(text[2:] if text.startswith("0x") else text)
    if len(raw) != 20:
        raise SystemExit("--domain takes 20 bytes of hex")
    return raw


def local_scores(args):
    path = os.path.join(args.assets, "attestations.csv")
    with open(path) as f:
        atts = sc.read_attestations_csv(f.read())
    if not atts:
        raise SystemExit("No attestations found.")          # cli.rs:481-485
    out_path = os.path.join(args.assets, "scores.csv")
    if args.large:
        addrs, values = sc.scores_from_attestations(atts, _domain(args.domain), args.iterations, args.initial_score)
        with open(out_path, "w", newline="") as f:
            w = csv.writer(f, lineterminator="\n")
            w.writerow(["peer_address", "score_fr"])
            for a, v in zip(addrs, values):
                w.writerow(["0x" + a.hex(), "0x%064x" % v])
    else:
        scores = sc.calculate_scores(atts, None, _domain(args.domain), args.neighbours, args.iterations, args.initial_score)
        with open(out_path, "w") as f:
            f.write(sc.write_scores_csv(scores))
    print('Scores saved at "%s".' % out_path)                 # cli.rs:511-514


def threshold(args):
    with open(os.path.join(args.assets, "scores.csv")) as f:
        rows = list(csv.DictReader(f))
    want = args.peer.lower()
    for r in rows:
        if r["peer_address"].lower() == want:
            ok = sc.Threshold(int(r["score_fr"], 16), Fraction(int(r["numerator"]), int(r["denominator"])), args.threshold,
                              num_neighbours=args.neighbours, initial_score=args.initial_score).check_threshold()
            print("above threshold" if ok else "below threshold")
            return 0 if ok else 1
    raise SystemExit("participant not found in scores.csv")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="protocol_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    common = argparse.ArgumentParser(add_help=False)
    common.add_argument("--assets", default="assets", help="directory holding attestations.csv / scores.csv")
    common.add_argument("--neighbours", type=int, default=4)          # circuits/mod.rs:39-45
    common.add_argument("--iterations", type=int, default=20)
    common.add_argument("--initial-score", type=int, default=1000)
    p = sub.add_parser("local-scores", parents=[common])
    p.add_argument("--domain", default="0x" + "00" * 20)
    p.add_argument("--large", action="store_true", help="any number of participants, field scores only")
    p.set_defaults(fn=local_scores)
    p = sub.add_parser("threshold", parents=[common])
    p.add_argument("--peer", required=True)
    p.add_argument("--threshold", type=int, required=True)
    p.set_defaults(fn=threshold)
    args = ap.parse_args(argv)
    return args.fn(args) or 0


if __name__ == "__main__":
    sys.exit(main())
