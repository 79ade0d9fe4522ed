"""Shared parity cases: (name, values, nullmap, scheme spec, compress kwargs).

Distributions follow the reference's test inputs (test/DatasetGenerator.cpp, test-cases/V1.cpp,
V2.cpp, tools/examples/compression.cpp:10-20, tools/playground/for_tests.cpp:47-58) with fixed
seeds; edge cases cover empty, single-row, ragged (non multiple of 32/64/128/256) and full blocks.
"""
import numpy as np

from btrblocks_amd import synth

N = 65536
KAT10 = np.array([1, -1, 0, -10, 10, -300, -2**31, 2**31 - 1, -2, 128], dtype=np.int32)  # for_tests.cpp:48


def _rng(seed):
    return np.random.default_rng(seed)


def cases():
    out = []

    def add(name, values, spec, nullmap=None, **kw):
        out.append((name, np.ascontiguousarray(values), nullmap, spec, kw))

    # ---- integers
    add("int_rle_bp", synth.int_rle(N), "rle(bp,bp)")
    add("int_dict_bp", synth.int_dict(N), "dict(bp)")
    add("int_for_bp", synth.int_for(N), "for(bp)")
    add("int_bp_12bit", _rng(10).integers(0, 4096, N).astype(np.int32), "bp")
    add("int_bp_mixed_widths", (_rng(11).integers(0, 2**31, N) >> _rng(12).integers(0, 31, N)).astype(np.int32), "bp")
    add("int_bp_negative", _rng(13).integers(-2**31, 2**31, N).astype(np.int32), "bp")
    add("int_bp_zeros", np.zeros(N, dtype=np.int32), "bp")
    add("int_bp_kat10_x40", np.tile(KAT10, 40), "bp")
    add("int_uncompressed", _rng(14).integers(-2**31, 2**31, 5000).astype(np.int32), "uncompressed")
    add("int_onevalue", np.full(N, -12345, dtype=np.int32), "onevalue")
    add("int_auto_onevalue", np.full(65000, 7, dtype=np.int32), "auto")
    add("int_frequency_bp", synth.int_frequency(N), "frequency(bp)")
    add("int_frequency_dense", synth.int_frequency(N, top_fraction=0.5), "frequency(dict(bp))")
    add("int_frequency_none", np.full(N, 5, dtype=np.int32), "frequency(bp)")
    add("int_dict8", synth.dictionary_values(65000, 255), "dict8")
    add("int_dict16", synth.dictionary_values(65000, 1020), "dict16")
    add("int_dict_big", _rng(15).integers(0, 2**31, N).astype(np.int32), "dict(bp)")
    add("int_rle_dict", synth.int_rle(N), "rle(dict(bp),bp)")
    add("int_for_rle", synth.int_rle(N) + 1_000_000, "for(rle(bp,bp))")
    add("int_for_dict", synth.int_dict(N), "for(dict(bp))")
    add("int_dict_rle_codes", synth.int_rle(N), "dict(rle(bp,bp))")
    add("int_rle_short_runs", np.repeat(_rng(16).integers(0, 1000, N // 2 + 1), 2)[:N].astype(np.int32), "rle(bp,bp)")
    add("int_rle_one_run", np.full(N, 9, dtype=np.int32), "rle(bp,bp)")
    add("int_rle_all_distinct", np.arange(3000, dtype=np.int32), "rle(bp,bp)")
    add("int_depth4", synth.int_rle(N), "rle(rle(bp,bp),for(bp))", max_cascade_depth=4)
    add("int_depth5_for_chain", synth.int_for(N), "for(for(for(bp)))", max_cascade_depth=5)
    add("int_example_auto", synth.example_runs(65000), "auto")
    add("int_auto_random", _rng(17).integers(0, 100000, N).astype(np.int32), "auto")
    # dictionaries around the size where the plan trades the walker's jump table for one more
    # workgroup per CU: equally wide code groups (table given up) and varying ones (table kept)
    pool = _rng(40).integers(-2**31, 2**31, 4400).astype(np.int32)
    even = _rng(41).integers(0, 4400, N)
    even[::32] = np.argsort(pool)[-1]  # every miniblock holds the largest code: one width throughout
    add("int_dict_4400_even_codes", pool[even], "dict(bp)")
    add("int_dict_4400_varying_codes", pool[(_rng(42).integers(0, 4400, N) >> _rng(43).integers(0, 12, N // 32 + 1).repeat(32)[:N])], "dict(bp)")
    # PFOR (SIMDFastPFor<8> + VariableByte) and the legacy PFOR_DELTA
    outliers = _rng(30).integers(0, 1000, N).astype(np.int32)
    outliers[_rng(31).integers(0, N, N // 50)] = _rng(32).integers(0, 2**31 - 1, N // 50).astype(np.int32)
    add("int_pfor_outliers", outliers, "pfor")
    add("int_pfor_kat10_x40", np.tile(KAT10, 40), "pfor")
    add("int_pfor_zeros", np.zeros(3000, dtype=np.int32), "pfor")
    w1 = _rng(33).integers(0, 4, 4096).astype(np.int32)
    w1[::7] += 4
    add("int_pfor_width1_exceptions", w1, "pfor")
    add("int_pfor_mixed_widths", (_rng(34).integers(0, 2**31, N) >> _rng(35).integers(0, 31, N)).astype(np.int32), "pfor")
    add("int_pfor_two_pages", outliers.repeat(2)[:100000] ^ 5, "pfor")
    add("int_pfor_delta_sorted", np.cumsum(_rng(36).integers(0, 50, N)).astype(np.int32), "pfor_delta")
    add("int_pfor_delta_outliers", outliers, "pfor_delta")
    add("int_rle_pfor", synth.int_rle(N), "rle(pfor,pfor)")
    add("int_for_pfor", synth.int_for(N), "for(pfor)")
    add("int_dict_pfor_codes", synth.int_dict(N), "dict(pfor)")
    add("int_frequency_pfor", synth.int_frequency(N), "frequency(pfor)")
    for n in (1, 255, 256, 257, 511, 513, 1000, 16960, 65535):
        add(f"int_pfor_n{n}", _rng(200 + n).integers(0, 1 << 17, n).astype(np.int32), "pfor")
        add(f"int_pfor_delta_n{n}", np.cumsum(_rng(300 + n).integers(0, 9, n)).astype(np.int32), "pfor_delta")
    # ragged sizes
    for n in (0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 16960, 65535):
        add(f"int_bp_n{n}", _rng(100 + n).integers(0, 1 << 17, n).astype(np.int32), "bp")
        add(f"int_rle_n{n}", synth.int_rle(max(n, 1))[:n], "rle(bp,bp)")
        add(f"int_dict_n{n}", synth.int_dict(max(n, 1), distinct=97)[:n], "dict(bp)")
    # nulls
    nm = synth.random_nullmap(N, 0.7)
    add("int_rle_nulls_flipped", synth.int_rle(N), "rle(bp,bp)", nullmap=nm)
    add("int_dict_nulls_regular", synth.int_dict(N), "dict(bp)", nullmap=synth.random_nullmap(N, 0.2, seed=9))
    add("int_for_allnull", synth.int_for(N), "for(bp)", nullmap=np.zeros(N, dtype=np.uint8))
    add("int_nulls_runs", synth.int_for(N), "bp", nullmap=np.repeat(_rng(18).integers(0, 2, N // 64), 64).astype(np.uint8))

    # ---- doubles
    add("dbl_pseudo_bp", synth.double_decimal(N), "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_pseudo_auto_children", synth.double_decimal(N), "pseudodecimal")
    add("dbl_pseudo_nopatch", synth.double_decimal(N, patch_fraction=0), "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_pseudo_rle_children", np.repeat(synth.double_decimal(N // 16, patch_fraction=0.02), 16),
        "pseudodecimal(rle(bp,bp),rle(bp,bp),dict(bp))")
    edge = np.array([0.0, -0.0, 1.0, -1.5, 0.1, 0.25, 123456.789, 2147483647.0, 2147483648.0, 1e-22, 3e-22,
                     1e22, 0.30000000000000004, 5e-324, 1.7976931348623157e308, 99.99], dtype=np.float64)
    add("dbl_pseudo_edges", np.tile(edge, 64), "pseudodecimal(bp,bp,uncompressed)")
    # exception placement against the streamed kernel's row scheduling (digits lag rows by the
    # exceptions before them): many patches, patches bunched at either end, wide digit strings
    add("dbl_pseudo_many_patches", synth.double_decimal(N, patch_fraction=0.45), "pseudodecimal(bp,bp,uncompressed)")
    bunch = synth.double_decimal(N, patch_fraction=0)
    tail_bunch = bunch.copy()
    tail_bunch[-9000:] = 1.0 / 3.0 * np.arange(1, 9001)
    add("dbl_pseudo_patches_at_end", tail_bunch, "pseudodecimal(bp,bp,uncompressed)")
    head_bunch = bunch.copy()
    head_bunch[:9000] = 1.0 / 7.0 * np.arange(1, 9001)
    add("dbl_pseudo_patches_at_start", head_bunch, "pseudodecimal(bp,bp,uncompressed)")
    mid_bunch = bunch.copy()
    mid_bunch[30000:33500] = 1.0 / 3.0 * np.arange(1, 3501)
    add("dbl_pseudo_patches_in_the_middle", mid_bunch, "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_pseudo_one_exponent", _rng(62).integers(0, 10**7, N).astype(np.float64) * 0.01 + 0.0, "pseudodecimal(bp,bp,uncompressed)")
    wide = (_rng(60).integers(0, 2**31 - 1, N) >> _rng(61).integers(0, 30, N)).astype(np.float64) / 1000.0
    add("dbl_pseudo_wide_digits", wide, "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_pseudo_two_blocks", synth.double_decimal(100000, patch_fraction=0.02), "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_uncompressed", _rng(20).uniform(-1e6, 1e6, 5000), "uncompressed")
    add("dbl_onevalue", np.full(N, 3.25), "onevalue")
    add("dbl_dict_bp", synth.dictionary_values(N, 1020, dtype=np.float64), "dict(bp)")
    dpool = _rng(44).uniform(-1e9, 1e9, 6000)
    add("dbl_dict_4000_even_codes", dpool[_rng(45).integers(0, 4000, N)], "dict(bp)")
    add("dbl_dict_4000_varying_codes", dpool[_rng(46).integers(0, 4000, N) >> _rng(47).integers(0, 12, N // 32 + 1).repeat(32)[:N]], "dict(bp)")
    add("dbl_dict_6000_global_gather", dpool[_rng(48).integers(0, 6000, N)], "dict(bp)")
    add("dbl_dict8", synth.dictionary_values(65000, 255, dtype=np.float64), "dict8")
    add("dbl_dict16", synth.dictionary_values(65000, 1020, dtype=np.float64), "dict16")
    add("dbl_rle", np.repeat(_rng(21).uniform(-999999, 999999, N // 21 + 1), 21)[:N], "rle(uncompressed,bp)")
    add("dbl_rle_dict", np.repeat(synth.dictionary_values(N // 8, 300, dtype=np.float64), 8), "rle(dict(bp),bp)")
    add("dbl_frequency", synth.double_frequency(N), "frequency(uncompressed)")
    add("dbl_frequency_pseudo", np.where(_rng(22).random(N) < 0.95, 10.5, synth.double_decimal(N, patch_fraction=0)),
        "frequency(pseudodecimal(bp,bp,uncompressed))")
    add("dbl_doublebp", (_rng(23).integers(0, 1000, N)).astype(np.float64).view(np.int64).astype(np.int64).view(np.float64)
        if False else _rng(23).integers(0, 2**20, N).astype(np.int64).view(np.float64), "doublebp")
    add("dbl_example_auto", synth.example_runs(65000, seed=69, dtype=np.float64), "auto")
    add("dbl_auto_decimal", synth.double_decimal(N), "auto")
    add("dbl_auto_random", _rng(24).uniform(-999999, 999999, N), "auto")
    for n in (0, 1, 3, 4, 5, 63, 64, 65, 257, 16960):
        add(f"dbl_pseudo_n{n}", synth.double_decimal(max(n, 1), patch_fraction=0.05)[:n], "pseudodecimal(bp,bp,uncompressed)")
        add(f"dbl_dict_n{n}", synth.dictionary_values(max(n, 1), 50, dtype=np.float64)[:n], "dict(bp)")
    # NaN (several payloads), infinities and signed zeros: every scheme keeps their bits (the
    # reference's std::map-based statistics are undefined for NaN; value-equality schemes may
    # merge -0.0 with 0.0)
    special = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, 1.5, -2.25], dtype=np.float64)
    special_bits = special.view(np.uint64).copy()
    special_bits[1] = 0x7FF0000000000001  # a signalling NaN pattern
    special = special_bits.view(np.float64)
    sp = special[_rng(70).integers(0, special.size, N)]
    sp_runs = np.repeat(special[_rng(71).integers(0, special.size, N // 16)], 16)
    for spec in ("auto", "dict(bp)", "dict8", "dict16", "frequency(uncompressed)", "rle(uncompressed,bp)",
                 "pseudodecimal(bp,bp,uncompressed)", "uncompressed"):
        add(f"dbl_special_{spec.split('(')[0]}", sp_runs if spec.startswith("rle") else sp, spec)
    add("dbl_special_mostly_nan", np.where(_rng(72).random(N) < 0.97, np.nan, sp), "frequency(dict(bp))")
    add("dbl_onevalue_nan", np.full(N, np.nan), "auto")
    add("dbl_pseudo_special_patches", np.where(_rng(73).random(N) < 0.03, sp, synth.double_decimal(N, patch_fraction=0)),
        "pseudodecimal(bp,bp,uncompressed)")
    add("dbl_pseudo_special_patches_dict", np.where(_rng(74).random(N) < 0.03, sp, synth.double_decimal(N, patch_fraction=0)),
        "pseudodecimal(bp,bp,dict(bp))")
    add("dbl_pseudo_nulls", synth.double_decimal(N), "pseudodecimal(bp,bp,uncompressed)", nullmap=synth.random_nullmap(N, 0.9))
    # a nullmap that is not one constant sends these trees through the generic kernels
    add("dbl_dict_nulls", synth.dictionary_values(N, 1020, dtype=np.float64), "dict(bp)", nullmap=synth.random_nullmap(N, 0.6, seed=3))
    add("int_for_nulls", synth.int_for(N), "for(bp)", nullmap=synth.random_nullmap(N, 0.6, seed=4))
    add("int_dict_big_nulls", _rng(15).integers(0, 2**31, N).astype(np.int32), "dict(bp)", nullmap=synth.random_nullmap(N, 0.6, seed=5))
    add("int_dict_allnull", synth.int_dict(N), "dict(bp)", nullmap=np.zeros(N, dtype=np.uint8))
    return out
