"""Mirror of the reference's ``src.datamodule`` (VLBDataModule, VLBDataModuleConfig, VLB_Dataset,
VLBDatasets; src/datamodule/videollama2_vlb_datamodule.py:24-238).

Same sample schema (6 arrays per sample: timeseries, vision, language as fp32 tensors; padvals,
vis_weights, lang_weights as numpy), same file-level train/val split (one random file is the
validation set, ``np.random.RandomState(random_state).choice``), same ``$SCRATCH_PATH`` /
``s*`` -> season substitution, same DataLoader settings.  HDF5 sample stores are read with h5py
when it is installed and with the package's own pure-Python reader (``h5lite``, pinned against files written by the real
h5py) otherwise.  Two additions because this environment has no CNeuroMod files:
  * ``.npz`` lazy-load files with the same keys (``{i}_{mod}`` + ``dset_len``) are read too
    (tools/h5_to_npz.py converts);
  * ``lazyload_path: synthetic:<n_files>x<samples>`` yields seeded synthetic samples.
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


try:                                   # a LightningDataModule when Lightning is installed (reference datamodule :156), so the
    from lightning.pytorch import LightningDataModule as _Base      # reference's train.py can pass it to Trainer.fit
except Exception:
    class _Base:
        """Stand-in for ``lightning.pytorch.LightningDataModule`` where Lightning is absent: the built-in Trainer only calls
        ``train_dataloader`` / ``val_dataloader``."""

        def __init__(self):
            pass

MODS_T = ("timeseries", "vision", "language")
MODS_N = ("padvals", "vis_weights", "lang_weights")


def get_idx(ranges, val):
    for i, (lo, hi) in enumerate(ranges):
        if lo <= val < hi:
            return i
    return -1


@dataclass
class VLBDataModuleConfig:
    lazyload_path: str
    subject: str
    seasons: list
    delay: int
    window: int
    random_state: int
    shuffle_val_data: bool
    batch_size: int = 1
    num_workers: int = 0
    # extras
    geometry: str = "7b"
    num_target: int = 1000


def open_h5(path):
    """h5py when it is installed, else the package's own read-only HDF5 subset (h5lite) - same indexing surface."""
    try:
        import h5py
        return h5py.File(path, "r")
    except ImportError:
        from . import h5lite
        return h5lite.File(path)


class _H5File:
    """The reference's lazy-load sample store (src/preprocessing/videollama2_vlb_lazyloading.py:141-164), read exactly
    as its VLB_Dataset does (src/datamodule/...:83-109): f["{i}"]["{i}_{mod}"], f["dset_len"][0]."""

    def __init__(self, path):
        self.f = open_h5(path)
        self.length = int(np.array(self.f["dset_len"])[0])

    def get(self, i, mod):
        return np.array(self.f[f"{i}"][f"{i}_{mod}"])


class _NpzFile:
    def __init__(self, path):
        self.f = np.load(path, mmap_mode="r")
        self.length = int(self.f["dset_len"][0])

    def get(self, i, mod):
        return np.array(self.f[f"{i}_{mod}"])


class _SyntheticFile:
    def __init__(self, spec, geometry, num_target):
        from .geometry import geometry_7b, geometry_mini
        from .synthetic import synthetic_batch
        self.seed, self.length = spec
        self.g = geometry_mini(num_target=num_target) if geometry == "mini" else geometry_7b(num_target=num_target)
        self._gen = synthetic_batch

    def get(self, i, mod):
        b = self._gen(self.g, 1, seed=self.seed * 100003 + i)
        return b[mod][0].numpy()


class VLB_Dataset(Dataset):
    def __init__(self, ds_paths, geometry="7b", num_target=1000):
        self.ds_files, self.length, self.ranges = {}, 0, []
        for i, p in enumerate(ds_paths):
            if isinstance(p, tuple):
                f = _SyntheticFile(p, geometry, num_target)
            elif str(p).endswith(".npz"):
                f = _NpzFile(p)
            else:
                f = _H5File(p)
            self.ds_files[i] = {"ds_file": f, "idx_from": self.length}
            self.ranges.append((self.length, self.length + f.length))
            self.length += f.length

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        i = get_idx(self.ranges, idx)
        f = self.ds_files[i]["ds_file"]
        k = idx - self.ds_files[i]["idx_from"]
        item = {m: torch.from_numpy(np.asarray(f.get(k, m))).float() for m in MODS_T}
        item.update({m: np.asarray(f.get(k, m)) for m in MODS_N})
        return item


@dataclass
class VLBDatasets:
    config: VLBDataModuleConfig
    train: VLB_Dataset | None = None
    val: VLB_Dataset | None = None
    test: VLB_Dataset | None = None

    def __post_init__(self):
        c = self.config
        if c.lazyload_path.startswith("synthetic:"):
            n_files, n_samples = (int(x) for x in c.lazyload_path.split(":", 1)[1].split("x"))
            f_list = [(c.random_state + k, n_samples) for k in range(n_files)]
            names = [f"synthetic_{k}" for k in range(n_files)]
        else:
            f_list = []
            for s in c.seasons:
                f_list += sorted(glob.glob(
                    c.lazyload_path.replace("$SCRATCH_PATH", os.environ.get("SCRATCH_PATH", ".")).replace("s*", f"{s}")))
            names = [os.path.basename(x) for x in f_list]
        if not f_list:
            raise FileNotFoundError(f"no lazy-load files match {c.lazyload_path!r}")
        r = np.random.RandomState(c.random_state)
        vi = int(r.choice(len(f_list), 1)[0])
        val_file = [f_list[vi]]
        train_files = [x for x in f_list if x != f_list[vi]]
        self.dset_names = {"val_set": [names[vi]], "train_set": [n for n, x in zip(names, f_list) if x != f_list[vi]]}
        self.val = VLB_Dataset(val_file, c.geometry, c.num_target)
        self.train = VLB_Dataset(train_files, c.geometry, c.num_target)


class VLBDataModule(_Base):
    def __init__(self, config: VLBDataModuleConfig) -> None:
        super().__init__()
        self.config = config
        self.datasets = VLBDatasets(self.config)

    def x_dataloader(self, dataset, shuffle: bool = True, sampler=None):
        if dataset is None:
            raise AttributeError
        # pinned batches so DevicePrefetcher's host->device copies are truly asynchronous
        return DataLoader(dataset=dataset, batch_size=self.config.batch_size, shuffle=shuffle and sampler is None,
                          sampler=sampler, num_workers=self.config.num_workers, pin_memory=torch.cuda.is_available())

    def train_dataloader(self, rank: int = 0, world: int = 1):
        """reference :219-225 (shuffle=True).  The permutation is a function of (random_state, epoch) - set by
        ``sampler.set_epoch`` - instead of the global torch RNG, so a run resumed from a checkpoint continues with
        the same clip order; under data parallelism ranks draw disjoint, rank-strided clips of that permutation
        (the reference has no DistributedSampler: it never ran multi-GPU in mainline)."""
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(self.datasets.train, num_replicas=world, rank=rank, shuffle=True,
                                     seed=self.config.random_state, drop_last=world > 1)
        return self.x_dataloader(dataset=self.datasets.train, sampler=sampler)

    def val_dataloader(self):
        return self.x_dataloader(dataset=self.datasets.val, shuffle=self.config.shuffle_val_data)


class DevicePrefetcher:
    """Iterates a DataLoader one batch ahead of the consumer: the large tensors of batch i+1 (16.3 MB of fp32
    pixels per clip) are copied host->device on a side stream while step i computes, and are
    handed over with an event, so the step never waits on PCIe.  The token ids and ``padvals`` stay on the
    host: the step sizes its unpadded row layout from them without a device sync (they are tiny and are
    uploaded by the step itself).  Device tensors are tied to the consumer stream with ``record_stream`` so
    the caching allocator cannot recycle them while a kernel still reads them."""

    def __init__(self, loader, device, keep_on_host=("language", "padvals"), on_staged=None, on_discard=None):
        """``on_staged(batch, event)``: called for every batch right after its copies were enqueued (event = copies done),
        i.e. one step before the batch is handed over - VLBLitModule.prefetch_vision hooks in here.
        ``on_discard(batch)``: called for a batch that was staged but will never be handed over (the consumer stopped
        iterating: ``limit_val_batches``, ``max_steps``, an exception) - VLBLitModule.discard_prefetched_vision."""
        self.loader, self.device, self.keep = loader, torch.device(device), tuple(keep_on_host)
        self.on_staged, self.on_discard = on_staged, on_discard
        self.sampler = getattr(loader, "sampler", None)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        out, ev = {}, torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                out[k] = v if (k in self.keep or not torch.is_tensor(v)) else v.to(self.device, non_blocking=True)
            ev.record(self.stream)
        if self.on_staged is not None:
            self.on_staged(out, ev)
        return out, ev

    def __iter__(self):
        for _, batch in self.iter_selected():
            yield batch

    def iter_selected(self, select=None, limit=None):
        """Yields ``(index in the underlying loader, batch)`` for the batches with ``select(index)`` true (all when None),
        stopping in front of index ``limit``.  Only selected batches are copied to the device and announced through
        ``on_staged``: a batch the consumer skips (another rank's validation batch, the part of an epoch a resumed run has
        already seen) never starts a side-stream computation nobody would pick up."""
        def wanted():
            for i, b in enumerate(self.loader):
                if limit is not None and i >= limit:
                    return
                if select is None or select(i):
                    yield i, b
        it = wanted()
        nxt = None
        try:
            first = next(it, None)
            if first is None:
                return
            nxt = (first[0],) + self._stage(first[1])
            while nxt is not None:
                idx, cur, ev = nxt
                nxt = None
                follow = next(it, None)               # batch i+1 starts moving before step i is enqueued
                if follow is not None:
                    nxt = (follow[0],) + self._stage(follow[1])
                consumer = torch.cuda.current_stream(self.device)
                consumer.wait_event(ev)
                for v in cur.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(consumer)
                yield idx, cur
        finally:
            if nxt is not None and self.on_discard is not None:      # staged, never handed over
                self.on_discard(nxt[1])
            it.close()
