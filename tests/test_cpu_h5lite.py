"""The pure-Python HDF5 reader (phantom_vlb_amd/h5lite.py) against files written by the REAL h5py 3.3 / libhdf5 1.10.6
(tests/golden/make_h5_fixtures.py, run with the image's /opt/conda/bin/python3.9): the reference's lazy-load sample-store
layout read the way its VLB_Dataset reads it (src/datamodule/...:83-109), and the gzip-4 chunked per-episode layout."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["lazyload", "episodes"])
def test_every_dataset_equals_what_numpy_wrote(name):
    from phantom_vlb_amd import h5lite
    f = h5lite.File(os.path.join(GOLD, f"{name}_fixture.h5"))
    exp = np.load(os.path.join(GOLD, f"{name}_fixture_expected.npz"))
    assert len(exp.files) > 30
    for k in exp.files:
        d = f[k]
        a = np.array(d)
        assert d.shape == exp[k].shape and a.dtype == exp[k].dtype and np.array_equal(a, exp[k]), k
    groups = {k.split("/")[0] for k in exp.files}
    assert set(f.keys()) == groups                      # incl. the 70-group root and the 45-link group (multi-node symbol tables)
    if name == "episodes":
        assert set(f["many"].keys()) == {f"d{i:03d}" for i in range(45)}
        assert f["stress"]["edge_chunks_shuffle"][36, 22] == exp["stress/edge_chunks_shuffle"][36, 22]
    with pytest.raises(KeyError):
        f["nope"]
    with pytest.raises(NotImplementedError):
        h5lite.File(os.path.join(GOLD, f"{name}_fixture.h5"), "a")


def test_vlb_dataset_reads_the_reference_sample_store_layout(monkeypatch):
    """VLB_Dataset over an .h5 written like the reference writes it: same items, same dtypes as the reference's
    __getitem__ produces (timeseries / vision / language -> float32 tensors, the rest raw numpy)."""
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py hidden for this test")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_h5py)
    from phantom_vlb_amd.datamodule import VLB_Dataset
    from torch.utils.data import DataLoader
    path = os.path.join(GOLD, "lazyload_fixture.h5")
    ds = VLB_Dataset([path, path])
    exp = np.load(os.path.join(GOLD, "lazyload_fixture_expected.npz"))
    assert len(ds) == 10 and ds.ranges == [(0, 5), (5, 10)]
    for idx in (0, 4, 7):
        item, k = ds[idx], idx % 5
        assert item["vision"].dtype == torch.float32 and tuple(item["vision"].shape) == (8, 3, 14, 14)
        assert torch.equal(item["vision"], torch.from_numpy(exp[f"{k}/{k}_vision"]))
        assert torch.equal(item["language"], torch.from_numpy(exp[f"{k}/{k}_language"]).float())
        assert item["padvals"].dtype == np.int64 and np.array_equal(item["padvals"], exp[f"{k}/{k}_padvals"])
        assert item["vis_weights"].dtype == np.float64
    batch = next(iter(DataLoader(ds, batch_size=3)))
    assert batch["vision"].shape == (3, 8, 3, 14, 14) and batch["padvals"].shape == (3, 3) and batch["lang_weights"].dtype == torch.float64


def test_episode_groups_view_reads_gzip_chunked_datasets(monkeypatch):
    import builtins
    real_import = builtins.__import__
    monkeypatch.setattr(builtins, "__import__", lambda name, *a, **k: (_ for _ in ()).throw(ImportError()) if name == "h5py" else real_import(name, *a, **k))
    from phantom_vlb_amd.episodes import _Hdf5Groups
    g = _Hdf5Groups(os.path.join(GOLD, "episodes_fixture.h5"))
    exp = np.load(os.path.join(GOLD, "episodes_fixture_expected.npz"))
    assert "s01e01a" in g and "s01e01b" in g.keys()
    ep = g["s01e01b"]
    assert set(ep) == {"video_features", "transcript_features", "transcript_onsets", "masking_params"}
    for k, v in ep.items():
        assert np.array_equal(v, exp[f"s01e01b/{k}"])
