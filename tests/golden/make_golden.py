"""Regenerates the golden fixtures: small known inputs compressed by the CPU compressor
(btrblocks_amd.compress_chunk) together with the values a correct decoder must return.

The reference's own tests hold no golden bytes (they are round trips through one library, seeded
from time()); the inputs here are the known vectors its tools hold - the 10-value vector of
tools/playground/for_tests.cpp:48, the generateData() pattern of tools/examples/compression.cpp:10-20,
the distributions of test/DatasetGenerator.cpp - plus Pseudodecimal edge values derived from
btrblocks/scheme/double/Pseudodecimal.cpp:82-123. Expected output = input (every scheme is lossless).

Usage: python tests/golden/make_golden.py [name ...]   (writes tests/golden/*.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import btrblocks_amd as B  # noqa: E402
from btrblocks_amd import synth  # noqa: E402

KAT10 = np.array([1, -1, 0, -10, 10, -300, -2**31, 2**31 - 1, -2, 128], dtype=np.int32)
EDGE = np.array([0.0, -0.0, 1.0, -1.5, 0.1, 0.25, 123456.789, 2147483647.0, 2147483648.0, 1e-22, 3e-22, 1e22,
                 0.30000000000000004, 5e-324, 1.7976931348623157e308, 99.99, 0.07, 1234.5678], dtype=np.float64)


def vectors():
    rng = np.random.default_rng(2024)
    yield "kat10_bp", np.tile(KAT10, 13), None, "bp"
    yield "kat10_for_bp", np.tile(KAT10, 13), None, "for(bp)"
    yield "kat10_dict", np.tile(KAT10, 13), None, "dict(bp)"
    yield "example_int_auto", synth.example_runs(4000, unique=4095, runlength=40, seed=42), None, "auto"
    yield "example_dbl_auto", synth.example_runs(4000, unique=4095, runlength=40, seed=69, dtype=np.float64), None, "auto"
    yield "rle_int", synth.int_rle(3000), None, "rle(bp,bp)"
    yield "rle_dict_int", synth.int_rle(3000), None, "rle(dict(bp),bp)"
    yield "frequency_int", synth.int_frequency(3000, top_fraction=0.97), None, "frequency(bp)"
    yield "frequency_dbl", synth.double_frequency(3000, top_fraction=0.97), None, "frequency(uncompressed)"
    yield "dict8_int", synth.dictionary_values(2000, 255), None, "dict8"
    yield "dict16_dbl", synth.dictionary_values(2000, 1020, dtype=np.float64), None, "dict16"
    yield "pseudo_edges", np.tile(EDGE, 20), None, "pseudodecimal(bp,bp,uncompressed)"
    yield "pseudo_decimal", synth.double_decimal(3000, patch_fraction=0.03), None, "pseudodecimal(bp,bp,uncompressed)"
    yield "doublebp", rng.integers(0, 2**20, 1500).astype(np.int64).view(np.float64), None, "doublebp"
    yield "onevalue_int", np.full(1000, -7, dtype=np.int32), None, "auto"
    yield "nulls_regular", synth.int_dict(3000, distinct=50), synth.random_nullmap(3000, 0.2), "dict(bp)"
    yield "nulls_flipped", synth.int_rle(3000), synth.random_nullmap(3000, 0.8), "rle(bp,bp)"
    yield "nulls_all", synth.int_for(500), np.zeros(500, dtype=np.uint8), "for(bp)"
    yield "empty_int", np.zeros(0, dtype=np.int32), None, "auto"
    # for_tests.cpp:48-58 runs its vector through PFOR as well as BP
    yield "kat10_pfor", np.tile(KAT10, 40), None, "pfor"
    outliers = rng.integers(0, 1000, 1000).astype(np.int32)
    outliers[::37] = rng.integers(0, 2**31 - 1, outliers[::37].size).astype(np.int32)
    yield "pfor_outliers", outliers, None, "pfor"
    yield "pfor_delta_sorted", np.cumsum(rng.integers(0, 50, 1000)).astype(np.int32), None, "pfor_delta"


def main():
    only = set(sys.argv[1:])  # optional: regenerate just these fixtures
    for name, values, nullmap, spec in vectors():
        if only and name not in only:
            continue
        chunk = B.compress_chunk(values, nullmap, spec)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), chunk=chunk, values=values,
                            nullmap=np.ones(values.size, np.uint8) if nullmap is None else nullmap,
                            spec=np.array(spec), tree=np.array(B.describe(chunk)))
        print(f"{name:20s} {spec:40s} -> {B.describe(chunk):45s} {chunk.size} bytes")


if __name__ == "__main__":
    main()
