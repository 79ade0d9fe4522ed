"""Builds the HIP shared library (gfx950 only) and the CPU oracle used by the tests.

`python -m btrblocks_amd.build` compiles in-tree:
  btrblocks_amd/libbtrblocks_amd.so   the product: C ABI + HIP kernels + CPU compressor
  oracle/_build/libbtr_oracle.so      test infrastructure: CPU restatement of the decoder
  examples/compression                round trip through the C++ interface (include/btrblocks_amd.hpp)
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "btrblocks_amd", "csrc")
LIB = os.path.join(ROOT, "btrblocks_amd", "libbtrblocks_amd.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "btr_oracle.cpp")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libbtr_oracle.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources():
    out = []
    for d, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h")):
                out.append(os.path.join(d, f))
    out.append(os.path.join(ROOT, "include", "btrblocks_amd.h"))
    return out


DEBUG_LIB = os.path.join(ROOT, "btrblocks_amd", "libbtrblocks_amd_dbg.so")


def build_variant(name, defines):
    """Experiment build: the library compiled with extra -D macros into libbtrblocks_amd_<name>.so
    (load it with BTR_LIB_PATH). `python -m btrblocks_amd.build --variant NAME -DX -DY=1`."""
    target = os.path.join(ROOT, "btrblocks_amd", f"libbtrblocks_amd_{name}.so")
    cmd = [
        HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off", "-Wall", "-Wno-unused-function", *defines,
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        os.path.join(CSRC, "c_api.hip"),
        os.path.join(CSRC, "plan_builder.cpp"),
        os.path.join(CSRC, "host", "compress.cpp"),
        os.path.join(CSRC, "host", "codecs.cpp"),
        "-o", target, "-lpthread",
    ]
    subprocess.check_call(cmd)
    return target


def build_library(force=False, verbose=False, debug_steps=False):
    """debug_steps: diagnostic build (-DBTR_DEBUG_STEPS) whose kernels honour the step-skipping
    bits of BTR_DEBUG_FLAGS and record per-role wait times. It goes to a file of its own
    (libbtrblocks_amd_dbg.so, loaded with BTR_LIB_PATH) and is never shipped."""
    srcs = _sources()
    target = DEBUG_LIB if debug_steps else LIB
    if not force and _newer(target, srcs):
        return target
    cmd = [
        HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off", "-Wall", "-Wno-unused-function",
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        os.path.join(CSRC, "c_api.hip"),
        os.path.join(CSRC, "plan_builder.cpp"),
        os.path.join(CSRC, "host", "compress.cpp"),
        os.path.join(CSRC, "host", "codecs.cpp"),
        "-o", target, "-lpthread",
    ]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if debug_steps:
        cmd.insert(1, "-DBTR_DEBUG_STEPS")
    subprocess.check_call(cmd)
    return target


def build_oracle(force=False):
    if not force and _newer(ORACLE_LIB, [ORACLE_SRC]):
        return ORACLE_LIB
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    subprocess.check_call([
        "g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-Wall",
        ORACLE_SRC, "-o", ORACLE_LIB,
    ])
    return ORACLE_LIB


EXAMPLE_SRC = os.path.join(ROOT, "examples", "compression.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "compression")


def build_example(force=False):
    """The C++ example, linked against the in-tree library (found at run time next to the package)."""
    srcs = [EXAMPLE_SRC, os.path.join(ROOT, "include", "btrblocks_amd.hpp"), os.path.join(ROOT, "include", "btrblocks_amd.h")]
    if not force and _newer(EXAMPLE_BIN, srcs + [LIB]):
        return EXAMPLE_BIN
    subprocess.check_call([
        "g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), EXAMPLE_SRC, "-o", EXAMPLE_BIN,
        "-L", os.path.dirname(LIB), "-lbtrblocks_amd", "-Wl,-rpath,$ORIGIN/../btrblocks_amd",
        "-L", "/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64",
    ])
    return EXAMPLE_BIN


def main():
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build_variant(name, [a for a in sys.argv if a.startswith("-D")]))
        return
    force = "--force" in sys.argv
    verbose = "--verbose" in sys.argv
    print(build_library(force, verbose))
    if "--debug-steps" in sys.argv:
        print(build_library(force, verbose, True))
    print(build_oracle(force))
    print(build_example(force))


if __name__ == "__main__":
    main()
