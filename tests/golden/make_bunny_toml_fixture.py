"""Generates tests/golden/bunny_toml_clouds.npz: the INPUT clouds of the reference's example run test/bunny.toml, as the CLI's loader
produces them — data/bunny/model_bunny.txt subsampled at 0.5 and data_bunny.txt at 0.1 (src/utilities.hpp:193-217 semantics: keep each
point with probability `subsample` until floor(total * subsample) are kept), with the fixed seeds 1 / 2 (the reference seeds from
std::random_device; `params.seed` is this repository's addition).  Data only — read here from /root/reference, stored as float32.

    python tests/golden/make_bunny_toml_fixture.py        (needs tests/host_harness/libcli_harness.so: built on the fly)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.join(REPO, "tests", "host_harness")
REF = "/root/reference/data/bunny"


def main():
    so = os.path.join(HERE, "libcli_harness.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "cli_harness.cpp")], check=True)
    L = C.CDLL(so)
    L.cli_load_cloud.argtypes = [C.c_char_p, C.c_float, C.c_longlong, C.POINTER(C.c_float), C.c_long, C.c_char_p, C.c_int]
    L.cli_load_cloud.restype = C.c_long

    def load(path, sub, seed):
        buf = np.empty((200000, 3), np.float32)
        err = C.create_string_buffer(512)
        n = L.cli_load_cloud(path.encode(), sub, seed, buf.ctypes.data_as(C.POINTER(C.c_float)), len(buf), err, 512)
        assert n > 0, err.value
        return buf[:n].copy()

    tgt = load(os.path.join(REF, "model_bunny.txt"), 0.5, 1)
    src = load(os.path.join(REF, "data_bunny.txt"), 0.1, 2)
    assert 17000 < len(tgt) <= 35947 // 2 and 2800 < len(src) <= 30379 // 10, (len(tgt), len(src))  # the sampler may stop short of its budget
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bunny_toml_clouds.npz")
    np.savez_compressed(out, tgt=tgt, src=src, lut_resolution=np.float32(0.002), mse_threshold=np.float32(1e-3),
                        note="test/bunny.toml: target_subsample 0.5 (seed 1), source_subsample 0.1 (seed 2) of the Stanford-bunny demo clouds")
    print(out, tgt.shape, src.shape, os.path.getsize(out))


if __name__ == "__main__":
    sys.exit(main())
