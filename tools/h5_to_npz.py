"""Convert reference HDF5 files to the .npz layouts this package reads when h5py is not installed.
Run where h5py exists:

    python tools/h5_to_npz.py friends_llFile_sub-01_s1_n0.h5 [out.npz]        # lazy-load sample store
                                                                              # (videollama2_vlb_lazyloading.py:141-164)
    python tools/h5_to_npz.py --episodes friends_s1_features.h5 [out.npz]     # per-episode features or BOLD file:
                                                                              # every 'group/dataset' becomes a key
                                                                              # (phantom_vlb_amd/episodes.py reads it)
"""
import sys

import numpy as np


def convert_samples(src, dst):
    import h5py
    out = {}
    with h5py.File(src, "r") as f:
        n = int(np.array(f["dset_len"])[0])
        out["dset_len"] = np.array([n])
        for i in range(n):
            for mod in ("timeseries", "vision", "language", "padvals", "vis_weights", "lang_weights"):
                out[f"{i}_{mod}"] = np.array(f[f"{i}"][f"{i}_{mod}"])
    np.savez(dst, **out)          # uncompressed: samples are memory-mapped on read
    print(f"wrote {dst}: {n} samples")


def convert_groups(src, dst):
    import h5py
    out = {}
    with h5py.File(src, "r") as f:
        for g, grp in f.items():
            for d, ds in grp.items():
                out[f"{g}/{d}"] = np.array(ds)
    np.savez(dst, **out)
    print(f"wrote {dst}: {len(out)} datasets")


def main():
    args = [a for a in sys.argv[1:] if a != "--episodes"]
    src = args[0]
    dst = args[1] if len(args) > 1 else src.rsplit(".", 1)[0] + ".npz"
    (convert_groups if "--episodes" in sys.argv else convert_samples)(src, dst)


if __name__ == "__main__":
    main()
