"""Per-episode feature wire format -> lazy-load sample store (consume side).

The reference extracts, per Friends episode, four arrays into a gzip-4 HDF5 group named after the episode
(src/preprocessing/videollama2_vlb_extractfeatures.py:443-508):

    video_features       (n_tr, 12, 3, 336, 336) f32   one 3-TR frame window per TR
    transcript_features  (n_tr, 866)             int   token ids, right-padded with 0, one -201 video slot
    transcript_onsets    (n_tr, 64)              f64   onset (s) of each dialogue token, right-padded with 0
    masking_params       (n_tr, 3)               int   (pad_len, inst_len, dialog_len)

and a second script aligns them with the subject's BOLD runs into the per-sample store the DataModule reads
(src/preprocessing/videollama2_vlb_lazyloading.py:52-166).  This module is that second step as a library:
host-side numpy only (it runs once, offline, before training), reading HDF5 when h5py is installed and
dict / .npz episodes otherwise, and writing the same sample schema `VLB_Dataset` consumes
(`{i}_timeseries, {i}_vision, {i}_vis_weights, {i}_language, {i}_lang_weights, {i}_padvals`, `dset_len`).

What is NOT here: producing the four arrays from video files and transcripts (needs the VideoLLaMA2
tokenizer, its video processor and a video decoder - none available offline).

HRF weights: the reference calls nilearn (`get_hrf_weight`, src/utils.py:14-37; pinned nilearn==0.12.0,
requirements_rorqual.txt:41).  nilearn is absent here, so `glover_hrf_weight` restates the published
`compute_regressor(..., hrf_model="glover")` algorithm for that exact call; it is UNPINNED (no nilearn to
check against) and `get_hrf_weight` prefers nilearn whenever it is importable.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

TR_SECONDS = 1.49                 # videollama2_vlb_extractfeatures.py:81; hard-coded in the aligner (:97,109)
EPISODE_KEYS = ("video_features", "transcript_features", "transcript_onsets", "masking_params")
SAMPLE_MODS = ("timeseries", "vision", "vis_weights", "language", "lang_weights", "padvals")


# --------------------------------------------------------------------------------------------------
# Glover HRF weight (restatement of nilearn.glm.first_level.compute_regressor for the reference's call)
# --------------------------------------------------------------------------------------------------
def _glover_kernel(tr: float, oversampling: int = 50, time_length: float = 32.0) -> np.ndarray:
    """Difference of two gamma densities (Glover 1999 parameters as nilearn uses them: delay 6,
    undershoot 12, dispersion 0.9 / 0.9, ratio 0.35), sampled every tr/oversampling over 32 s, unit sum."""
    from scipy.stats import gamma
    dt = tr / oversampling
    t = np.linspace(0.0, time_length, int(np.rint(time_length / dt)))
    peak = gamma.pdf(t, 6.0 / 0.9, loc=dt, scale=0.9)
    under = gamma.pdf(t, 12.0 / 0.9, loc=dt, scale=0.9)
    h = peak - 0.35 * under
    return h / h.sum()


def glover_hrf_weight(time_diff: float, oversampling: int = 50, min_onset: float = -24.0) -> float:
    """Value at `time_diff` seconds of a unit boxcar (onset 0 s, duration 1 s, amplitude 1) convolved with
    the Glover HRF - what `compute_regressor(exp_condition=[[0],[1],[1]], hrf_model="glover",
    frame_times=[0, time_diff])[0][-1, 0]` returns (src/utils.py:29-36)."""
    t = float(time_diff)
    if not t > 0.0:
        raise ValueError("time_diff must be > 0 s (the reference's two-point frame grid degenerates otherwise)")
    # two frame times (0, t): n = 2, so the implied TR is t and the high-resolution grid runs from
    # min_onset to 2t with `oversampling` points per TR
    n_hr = (1.0 / t) * (2.0 * t - min_onset) * oversampling + 1
    grid = np.linspace(min_onset, 2.0 * t, int(np.rint(n_hr)))
    box = np.zeros_like(grid)
    tmax = grid.size
    i_on = min(int(np.searchsorted(grid, 0.0)), tmax - 1)
    i_off = min(int(np.searchsorted(grid, 1.0)), tmax - 1)
    if i_off < tmax - 1 and i_off == i_on:
        i_off += 1
    box[i_on] += 1.0
    box[i_off] -= 1.0
    box = np.cumsum(box)
    conv = np.convolve(box, _glover_kernel(t, oversampling))[:tmax]
    return float(np.interp(t, grid, conv))


def get_hrf_weight(time_diff: float) -> float:
    """nilearn when importable (bit-identical to the reference), the restatement above otherwise."""
    try:
        from nilearn.glm.first_level import compute_regressor  # type: ignore
    except Exception:
        return glover_hrf_weight(time_diff)
    reg, _ = compute_regressor(exp_condition=np.array([[0], [1], [1]]), hrf_model="glover",
                               frame_times=np.array([0.0, time_diff]))
    return float(reg[-1, 0])


# --------------------------------------------------------------------------------------------------
# Alignment (videollama2_vlb_lazyloading.py:94-140)
# --------------------------------------------------------------------------------------------------
def vision_weights(num_frames: int, window: int = 3, delay: int = 3,
                   hrf: Callable[[float], float] = get_hrf_weight) -> np.ndarray:
    """HRF weight of each down-sampled frame (12 frames -> floor(12/2)+1 = 7 connector frames) relative to
    the target TR's mid-point; identical for every sample of a run (lazyloading.py:100-115)."""
    n_ds = math.floor(num_frames / 2) + 1
    step = window / (n_ds - 1)
    abs_tr_delay = (window - 1) + delay + 0.5
    onsets = TR_SECONDS * (abs_tr_delay - np.arange(0, window + step, step))
    return np.array([hrf(t) for t in onsets])


def episode_key_map(timeseries_sessions: Mapping[str, Iterable[str]]) -> Dict[str, Tuple[str, str]]:
    """`{episode: (session, run)}` from the BOLD file's `session/run` names, the episode being the token after
    the last '-' of the run name's second '_' field (lazyloading.py:58-60), e.g.
    'ses-001_task-s01e02a_timeseries' -> 's01e02a'."""
    return {run.split("_")[1].split("-")[-1]: (ses, run) for ses, runs in timeseries_sessions.items() for run in runs}


def chunk_assignment(n_episodes: int, n_split: int) -> np.ndarray:
    """Output-file index of each episode (lazyloading.py:86-88)."""
    return np.floor(np.arange(n_episodes) / (n_episodes / n_split)).astype(int)


def align_run(episode: Mapping[str, np.ndarray], run_bold: np.ndarray, window: int = 3, delay: int = 3,
              hrf: Callable[[float], float] = get_hrf_weight) -> List[Dict[str, np.ndarray]]:
    """One episode's features + that run's BOLD matrix `(n_tr, V)` -> list of samples.

    Sample n pairs input window n+(window-1) with BOLD row n+(window-1)+delay (the first window-1 inputs
    have no complete window; the target lags the window's last TR by `delay`), rows beyond the shortest of
    the three streams are dropped, and the 64 token onsets become HRF weights of (TR mid-point - onset)
    for the first dialog_len slots (lazyloading.py:94-140)."""
    for k in EPISODE_KEYS:
        if k not in episode:
            raise KeyError(f"episode group lacks '{k}'")
    lead = window - 1
    tseries = np.asarray(run_bold)[lead + delay:]
    tr_onsets = [(lead + delay + 0.5 + i) * TR_SECONDS for i in range(tseries.shape[0])]
    vision = np.asarray(episode["video_features"])[lead:]
    language = np.asarray(episode["transcript_features"])[lead:]
    lang_onsets = np.array(episode["transcript_onsets"], dtype=np.float64)[lead:]      # private copy, rewritten below
    maskval = np.asarray(episode["masking_params"])[lead:]
    if maskval.shape[0] != language.shape[0]:
        raise ValueError("masking_params and transcript_features disagree on the number of TRs")
    vis_w = vision_weights(vision.shape[1], window, delay, hrf) if vision.shape[0] else np.zeros(0)
    n_rows = min(tseries.shape[0], vision.shape[0], language.shape[0])
    out = []
    for n in range(n_rows):
        dialog_len = int(maskval[n][2])
        lang_onsets[n][:dialog_len] = [hrf(t) for t in tr_onsets[n] - lang_onsets[n][:dialog_len]]
        out.append({"timeseries": tseries[n], "vision": vision[n], "vis_weights": vis_w,
                    "language": language[n], "lang_weights": lang_onsets[n], "padvals": maskval[n]})
    return out


# --------------------------------------------------------------------------------------------------
# Containers
# --------------------------------------------------------------------------------------------------
class _Hdf5Groups:
    """Read-only `{group: {dataset: array}}` view of an HDF5 file: h5py when installed, else the package's own reader
    (h5lite: chunked gzip-4 datasets, the format ..._extractfeatures.py:443-508 writes, are supported)."""

    def __init__(self, path):
        from .datamodule import open_h5
        self.f = open_h5(path)

    def keys(self):
        return list(self.f.keys())

    def __contains__(self, k):
        return k in self.f

    def __getitem__(self, k):
        return {name: np.array(d) for name, d in self.f[k].items()}


class _NpzGroups:
    """Same view over a flat .npz whose keys are 'group/dataset' (what tools/h5_to_npz.py --episodes writes)."""

    def __init__(self, path):
        self.f = np.load(path, mmap_mode="r")
        self._groups: Dict[str, List[str]] = {}
        for k in self.f.files:
            g, _, d = k.partition("/")
            self._groups.setdefault(g, []).append(d)

    def keys(self):
        return list(self._groups)

    def __contains__(self, k):
        return k in self._groups

    def __getitem__(self, k):
        return {d: np.array(self.f[f"{k}/{d}"]) for d in self._groups[k]}


def open_groups(src):
    """Mapping, .npz path or HDF5 path -> `{group: {dataset: array}}`."""
    if isinstance(src, Mapping):
        return src
    return _NpzGroups(src) if str(src).endswith(".npz") else _Hdf5Groups(src)


def _open_bold(src) -> Tuple[Dict[str, Tuple[str, str]], Callable[[str, str], np.ndarray]]:
    """BOLD container `{session: {run: (n_tr, V)}}` -> (episode map, loader)."""
    if isinstance(src, Mapping):
        return episode_key_map({s: list(r.keys()) for s, r in src.items()}), lambda s, r: np.asarray(src[s][r])
    if str(src).endswith(".npz"):
        f = np.load(src, mmap_mode="r")
        sessions: Dict[str, List[str]] = {}
        for k in f.files:
            s, _, r = k.partition("/")
            sessions.setdefault(s, []).append(r)
        return episode_key_map(sessions), lambda s, r: np.array(f[f"{s}/{r}"])
    g = _Hdf5Groups(src)
    return episode_key_map({s: list(g.f[s].keys()) for s in g.keys()}), lambda s, r: np.array(g.f[s][r])


def _write_store(path: str, samples: Sequence[Dict[str, np.ndarray]]) -> None:
    """Sample store in the f1 schema: HDF5 (groups '{i}', uncompressed, root 'dset_len'; lazyloading.py:141-166)
    for a .h5 path, flat .npz with the same dataset names otherwise."""
    if str(path).endswith(".npz"):
        out = {"dset_len": np.array([len(samples)])}
        for i, s in enumerate(samples):
            for m in SAMPLE_MODS:
                out[f"{i}_{m}"] = np.asarray(s[m])
        np.savez(path, **out)
        return
    import h5py
    with h5py.File(path, "w") as f:
        for i, s in enumerate(samples):
            g = f.create_group(f"{i}")
            for m in SAMPLE_MODS:
                g.create_dataset(f"{i}_{m}", data=np.asarray(s[m]))
        f.create_dataset("dset_len", data=[len(samples)])


def make_lazy_loading_dsets(features, timeseries, lazyload_path: str, subject: str, season: str,
                            n_split: int = 4, delay: int = 3, window: int = 3, ext: Optional[str] = None,
                            hrf: Callable[[float], float] = get_hrf_weight) -> List[str]:
    """The reference's entry point of the same name (lazyloading.py:52-166), as a function: every episode of
    `features` that has a BOLD run for this subject is aligned and appended to one of `n_split` files
    `friends_llFile_{subject}_{season}_n{i}.{h5|npz}` under `lazyload_path`.  Returns the paths written.

    `features` / `timeseries`: HDF5 path, .npz path (keys 'group/dataset') or nested dicts.  `ext` defaults to
    'h5' when h5py is importable and 'npz' otherwise; `VLB_Dataset` reads both."""
    if ext is None:
        try:
            import h5py  # noqa: F401
            ext = "h5"
        except ImportError:
            ext = "npz"
    ep_keys, load_run = _open_bold(timeseries)
    feats = open_groups(features)
    epi_list = [e for e in feats.keys() if e in ep_keys]
    if not epi_list:
        raise ValueError("no episode of the features file has a BOLD run in the timeseries file")
    chunk_idx = chunk_assignment(len(epi_list), n_split)
    os.makedirs(lazyload_path, exist_ok=True)
    written = []
    for i in range(n_split):
        samples: List[Dict[str, np.ndarray]] = []
        for ep in np.array(epi_list)[chunk_idx == i].tolist():
            ses, run = ep_keys[ep]
            samples.extend(align_run(feats[ep], load_run(ses, run), window, delay, hrf))
        path = os.path.join(lazyload_path, f"friends_llFile_{subject}_{season}_n{i}.{ext}")
        _write_store(path, samples)
        written.append(path)
    return written
