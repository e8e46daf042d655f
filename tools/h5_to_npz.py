"""Convert a reference lazy-load HDF5 file (src/preprocessing/videollama2_vlb_lazyloading.py:141-164) to the
.npz layout VLBDataModule reads when h5py is not installed.  Run where h5py exists:

    python tools/h5_to_npz.py friends_llFile_sub-01_s1_n0.h5 [out.npz]
"""
import sys

import numpy as np


def main():
    import h5py
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src.rsplit(".", 1)[0] + ".npz"
    out = {}
    with h5py.File(src, "r") as f:
        n = int(np.array(f["dset_len"])[0])
        out["dset_len"] = np.array([n])
        for i in range(n):
            for mod in ("timeseries", "vision", "language", "padvals", "vis_weights", "lang_weights"):
                out[f"{i}_{mod}"] = np.array(f[f"{i}"][f"{i}_{mod}"])
    np.savez(dst, **out)          # uncompressed: samples are memory-mapped on read
    print(f"wrote {dst}: {n} samples")


if __name__ == "__main__":
    main()
