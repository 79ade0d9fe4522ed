"""Synthetic string blocks for the string DICT sub-stream tests: (name, list of bytes objects,
codes scheme, lengths scheme, fsst_layout)."""
import numpy as np


def _words(rng, count, lo, hi):
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJ0123456789_-/ ", dtype=np.uint8)
    return [bytes(rng.choice(alphabet, size=int(rng.integers(lo, hi + 1)))) for _ in range(count)]


def cases():
    rng = np.random.default_rng(77)
    out = []
    cities = _words(rng, 300, 3, 24)
    # skewed categorical column, short runs: codes as picked by the integer picker
    picks = np.minimum(rng.zipf(1.3, 65536) - 1, len(cities) - 1)
    out.append(("categorical_auto", [cities[i] for i in picks], "auto", "auto", False))
    out.append(("categorical_bp", [cities[i] for i in picks], "bp", "auto", False))
    out.append(("categorical_dict_layout_fsst", [cities[i] for i in picks], "dict(bp)", "bp", True))
    # long runs: average run above 3 -> RLE-optimised path
    runs = np.repeat(rng.integers(0, len(cities), 4000), rng.integers(1, 40, 4000))[:65536]
    out.append(("runs_rle", [cities[i] for i in runs], "auto", "auto", False))
    out.append(("runs_rle_forced_kids", [cities[i] for i in runs], "rle(dict(bp),pfor)", "auto", False))
    out.append(("runs_rle_layout_fsst", [cities[i] for i in runs], "auto", "rle(bp,bp)", True))
    # one distinct value, empty strings, a ragged last block
    out.append(("one_value", [b"same"] * 5000, "auto", "auto", False))
    out.append(("with_empty", [b"", b"x", b"", b"yy", b"x"] * 777, "auto", "auto", True))
    many = _words(rng, 20000, 1, 12)
    out.append(("high_cardinality", [many[i] for i in rng.integers(0, len(many), 33333)], "pfor", "pfor", True))
    out.append(("single_row", [b"only"], "auto", "auto", False))
    return out
