"""Writes the HDF5 fixtures of tests/test_cpu_h5lite.py with the REAL h5py / libhdf5 - run where h5py exists, e.g.

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py        (h5py 3.3.0, libhdf5 1.10.6 in the build image)

The files follow the two layouts the reference produces and consumes:
  lazyload_fixture.h5   sample store as src/preprocessing/videollama2_vlb_lazyloading.py:141-164 writes it - one
                        `h5py.File(path, "a")` per sample, group "{i}" with six datasets created by
                        `create_dataset(name, data=...)` (contiguous, no filters), root dataset `dset_len`.
  episodes_fixture.h5   per-episode groups of gzip-4 chunked datasets (..._extractfeatures.py:443-508), plus cases
                        that stress the reader: a big-endian dataset, a chunked dataset with partial edge chunks and
                        the shuffle filter, > 20 links in one group (a multi-node symbol table), a 70-group root.
  *_expected.npz        the same arrays through numpy - what the pure-Python reader must reproduce.
Data are seeded synthetic numbers in the shapes of the schema (a small image size keeps the files small).
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(1234)


def lazyload():
    path = os.path.join(HERE, "lazyload_fixture.h5")
    if os.path.exists(path):
        os.remove(path)
    expect = {}
    n = 5
    for idx in range(n):
        sample = {
            "timeseries": rng.randn(128).astype(np.float32),
            "vision": rng.randn(8, 3, 14, 14).astype(np.float32),
            "vis_weights": rng.rand(5),                                   # float64, like np.array of python floats
            "language": np.concatenate([rng.randint(3, 500, 20), [-201], rng.randint(3, 500, 20), np.zeros(8)]).astype(np.float64),
            "lang_weights": np.concatenate([rng.rand(6), np.zeros(58)]),
            "padvals": np.array([8, 9, 6], dtype=np.int64),
        }
        with h5py.File(path, "a") as f:                                   # one append-open per sample, like the reference
            g = f.create_group(f"{idx}")
            for k in ("timeseries", "vision", "vis_weights", "language", "lang_weights", "padvals"):
                g.create_dataset(f"{idx}_{k}", data=sample[k])
        for k, v in sample.items():
            expect[f"{idx}/{idx}_{k}"] = v
    with h5py.File(path, "a") as f:
        f.create_dataset("dset_len", data=[n])
    expect["dset_len"] = np.array([n])
    np.savez_compressed(os.path.join(HERE, "lazyload_fixture_expected.npz"), **expect)
    print("wrote", path, os.path.getsize(path), "bytes")


def episodes():
    path = os.path.join(HERE, "episodes_fixture.h5")
    if os.path.exists(path):
        os.remove(path)
    expect = {}
    with h5py.File(path, "w") as f:
        for ep in ("s01e01a", "s01e01b"):
            g = f.create_group(ep)
            arrs = {
                "video_features": rng.randn(7, 4, 3, 10, 10).astype(np.float32),
                "transcript_features": rng.randint(0, 32000, (7, 40)).astype(np.int64),
                "transcript_onsets": rng.rand(7, 40),
                "masking_params": rng.randint(0, 50, (7, 3)).astype(np.int64),
            }
            for k, v in arrs.items():
                g.create_dataset(k, data=v, compression="gzip", compression_opts=4)
                expect[f"{ep}/{k}"] = v
        s = f.create_group("stress")
        be = rng.randn(6, 5).astype(">f8")
        s.create_dataset("big_endian", data=be)
        expect["stress/big_endian"] = be.astype("<f8")
        edge = rng.randint(-1000, 1000, (37, 23)).astype(np.int16)
        s.create_dataset("edge_chunks_shuffle", data=edge, chunks=(16, 10), compression="gzip", shuffle=True)
        expect["stress/edge_chunks_shuffle"] = edge
        u8 = rng.randint(0, 255, (3, 1000)).astype(np.uint8)
        s.create_dataset("chunked_plain", data=u8, chunks=(1, 256))
        expect["stress/chunked_plain"] = u8
        s.create_dataset("scalar_like", data=np.array([3.5], dtype=np.float32))
        expect["stress/scalar_like"] = np.array([3.5], dtype=np.float32)
        many = f.create_group("many")
        for i in range(45):
            v = np.arange(i + 1, dtype=np.int32)
            many.create_dataset(f"d{i:03d}", data=v)
            expect[f"many/d{i:03d}"] = v
        for i in range(70):
            f.create_group(f"g{i:02d}").create_dataset("x", data=np.float64([i, i + 0.5]))
            expect[f"g{i:02d}/x"] = np.float64([i, i + 0.5])
    np.savez_compressed(os.path.join(HERE, "episodes_fixture_expected.npz"), **expect)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    lazyload()
    episodes()
