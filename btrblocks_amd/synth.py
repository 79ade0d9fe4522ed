"""Deterministic synthetic columns.

The reference's generators (test/DatasetGenerator.cpp, tools/examples/compression.cpp:10-20) seed
from time()/random_device, so their data is not reproducible; these restate the same
distributions with fixed numpy seeds. Benchmark columns follow the shapes the baseline names:
RLE, Dictionary and FOR+bitpack integer columns and a decimal-like double column.
"""
import numpy as np

BLOCK_ROWS = 65536


def example_runs(size=640000, unique=(1 << 12) - 1, runlength=40, seed=42, dtype=np.int32):
    """tools/examples/compression.cpp:10-20: runs of `runlength` equal values < unique, one row
    skipped between runs (the reference's loop increments i twice at a run boundary)."""
    out = np.zeros(size, dtype=dtype)
    rng = np.random.RandomState(seed)  # MT19937, like std::mt19937(seed)
    i = 0
    while i < size - runlength:
        number = int(rng.randint(0, 2**32, dtype=np.uint64) % unique)
        out[i:i + runlength] = number
        i += runlength + 1
    return out


def int_rle(n, seed=1):
    """values u % 100000 in runs of uniform length 16..64."""
    rng = np.random.default_rng(seed)
    runs = int(n / 40 * 1.05) + 64  # mean run length is 40; 5 % head-room, topped up below
    lengths = rng.integers(16, 65, size=runs)
    while int(lengths.sum()) < n:
        lengths = np.concatenate([lengths, rng.integers(16, 65, size=runs // 8 + 64)])
    runs = lengths.size
    values = rng.integers(0, 100000, size=runs).astype(np.int32)
    out = np.repeat(values, lengths)[:n]
    return np.ascontiguousarray(out, dtype=np.int32)


def int_dict(n, seed=2, distinct=4095):
    """`distinct` random 31-bit values drawn uniformly."""
    rng = np.random.default_rng(seed)
    dictionary = np.unique(rng.integers(0, 2**31 - 1, size=distinct * 2).astype(np.int32))[:distinct]
    return dictionary[rng.integers(0, dictionary.size, size=n)]


def int_for(n, seed=3, bits=12):
    """1e9 + u % 2^bits."""
    rng = np.random.default_rng(seed)
    return (1_000_000_000 + rng.integers(0, 1 << bits, size=n)).astype(np.int32)


def double_decimal(n, seed=4, patch_fraction=0.01):
    """k * 0.01 with k < 1e7 (two decimals), `patch_fraction` non-decimal patches and a few -0.0."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 10_000_000, size=n)
    out = k.astype(np.float64) * 0.01
    m = max(1, int(n * patch_fraction)) if patch_fraction > 0 else 0
    if m:
        rows = rng.choice(n, size=m, replace=False)
        out[rows] = (k[rows].astype(np.float64) + 1.0) / 3.0
        out[rows[: max(1, m // 100)]] = -0.0
    return out


def int_constant(n, seed=0):
    return np.full(n, 123456 + seed, dtype=np.int32)


def int_frequency(n, seed=5, top_fraction=0.99):
    """test/DatasetGenerator.cpp:290-325: one dominant value, rare random exceptions."""
    rng = np.random.default_rng(seed)
    out = np.full(n, 777_777, dtype=np.int32)
    rows = rng.random(n) > top_fraction
    out[rows] = rng.integers(0, 2**30, size=int(rows.sum())).astype(np.int32)
    return out


def double_frequency(n, seed=6, top_fraction=0.99):
    rng = np.random.default_rng(seed)
    out = np.full(n, 1234.5, dtype=np.float64)
    rows = rng.random(n) > top_fraction
    out[rows] = rng.uniform(-999999, 999999, size=int(rows.sum()))
    return out


def dictionary_values(n, distinct, seed=7, dtype=np.int32):
    """test/DatasetGenerator.cpp:72-115, 134-177: `distinct` random values cycled."""
    rng = np.random.default_rng(seed)
    if dtype == np.int32:
        d = rng.integers(-2**31, 2**31 - 1, size=distinct).astype(np.int32)
    else:
        d = rng.uniform(-999999, 999999, size=distinct)
    return d[np.arange(n) % distinct]


def random_nullmap(n, valid_fraction, seed=8):
    rng = np.random.default_rng(seed)
    return (rng.random(n) < valid_fraction).astype(np.uint8)


def _run_lengths(rng, n, lo=16, hi=65):
    runs = int(n / ((lo + hi - 1) / 2) * 1.05) + 64
    lengths = rng.integers(lo, hi, size=runs)
    while int(lengths.sum()) < n:
        lengths = np.concatenate([lengths, rng.integers(lo, hi, size=runs // 8 + 64)])
    return lengths


def double_decimal_runs(n, seed=11, patch_fraction=0.01):
    """Two-decimal prices from a pool of 4096 values in runs of 16..64, with non-decimal patches."""
    rng = np.random.default_rng(seed)
    lengths = _run_lengths(rng, n)
    pool = rng.integers(0, 10_000_000, size=4096).astype(np.float64) * 0.01
    out = np.repeat(pool[rng.integers(0, pool.size, size=lengths.size)], lengths)[:n].copy()
    m = int(n * patch_fraction)
    if m:
        rows = rng.choice(n, size=m, replace=False)
        out[rows] = (rows.astype(np.float64) + 1.0) / 3.0
    return out


def double_dict(n, seed=12, distinct=4095):
    """`distinct` random doubles drawn uniformly."""
    rng = np.random.default_rng(seed)
    dictionary = np.unique(rng.random(distinct * 2))[:distinct] * 1e6
    return dictionary[rng.integers(0, dictionary.size, size=n)]


def double_runs(n, seed=13):
    """random doubles in runs of uniform length 16..64."""
    rng = np.random.default_rng(seed)
    lengths = _run_lengths(rng, n)
    return np.repeat(rng.random(lengths.size) * 1e6, lengths)[:n].copy()


def int_rle_pool(n, seed=7, distinct=1000):
    """runs of uniform length 16..64 over a pool of `distinct` random 31-bit values."""
    rng = np.random.default_rng(seed)
    lengths = _run_lengths(rng, n)
    pool = rng.integers(0, 2**31 - 1, size=distinct).astype(np.int32)
    return np.ascontiguousarray(np.repeat(pool[rng.integers(0, distinct, size=lengths.size)], lengths)[:n])


def int_outliers(n, seed=10, bits=10, outlier_fraction=0.02):
    """u % 2^bits with a few 31-bit outliers: the shape patched bit-packing is made for."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << bits, size=n).astype(np.int32)
    rows = rng.choice(n, size=max(1, int(n * outlier_fraction)), replace=False)
    out[rows] = rng.integers(0, 2**31 - 1, size=rows.size).astype(np.int32)
    return out


def int_frequency_pool(n, seed=9, top_fraction=0.95, distinct=500):
    """one dominant value; the exceptions come from a pool of `distinct` random values."""
    rng = np.random.default_rng(seed)
    out = np.full(n, 123456, dtype=np.int32)
    rows = rng.choice(n, size=int(n * (1 - top_fraction)), replace=False)
    pool = rng.integers(0, 2**31 - 1, size=distinct).astype(np.int32)
    out[rows] = pool[rng.integers(0, distinct, size=rows.size)]
    return out


def double_decimal_clean(n, seed=4):
    """double_decimal without patches: every exponent is 2, no exception rows."""
    return double_decimal(n, seed=seed, patch_fraction=0.0)


def double_decimal_mixed(n, seed=4):
    """Decimals of mixed magnitude (digits of 4 to 24 bits, one to four decimals) with 1 % patches:
    digit and exponent groups whose widths change from group to group."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(4, 25, size=(n + 31) // 32).repeat(32)[:n]
    k = rng.integers(0, 1 << 24, size=n) >> (24 - bits)
    e = rng.integers(1, 5, size=(n + 127) // 128).repeat(128)[:n]
    out = k.astype(np.float64) / np.power(10.0, e)
    m = max(1, n // 100)
    rows = rng.choice(n, size=m, replace=False)
    out[rows] = (k[rows].astype(np.float64) + 1.0) / 3.0
    return out


def int_outliers_few(n, seed=10):
    """int_outliers with one outlier in a million values: PFOR blocks almost free of exceptions."""
    return int_outliers(n, seed=seed, outlier_fraction=1e-6)
