"""Per-episode wire format -> lazy-load sample store (consume side).

Follows src/preprocessing/videollama2_vlb_lazyloading.py:52-166: alignment offsets, chunking, the HRF rewrite of
the token onsets, and the sample schema the DataModule reads back.  Host logic only (numpy)."""
import numpy as np
import pytest

from phantom_vlb_amd import episodes as E
from phantom_vlb_amd.datamodule import VLB_Dataset


def _episode(n_tr, seed, frames=12, hw=6, text=20, V=5):
    rng = np.random.default_rng(seed)
    vid = rng.standard_normal((n_tr, frames, 3, hw, hw)).astype(np.float32)
    ids = np.zeros((n_tr, text), dtype=np.int64)
    onsets = np.zeros((n_tr, 64))
    mask = np.zeros((n_tr, 3), dtype=np.int64)
    for t in range(n_tr):
        dlg = int(rng.integers(0, 6))
        inst = 3
        used = 1 + 1 + 2 + inst + dlg + 4
        ids[t, :used] = rng.integers(3, 1000, used)
        ids[t, 1] = -201
        # tokens were spoken inside the 3-TR window ending at TR t
        onsets[t, :dlg] = np.sort(rng.uniform(max(0.0, (t - 2) * E.TR_SECONDS), (t + 1) * E.TR_SECONDS, dlg))
        mask[t] = (text - used, inst, dlg)
    return {"video_features": vid, "transcript_features": ids, "transcript_onsets": onsets, "masking_params": mask}


def test_glover_weight_shape_and_known_properties():
    # unit-sum kernel * 1 s boxcar: value ~ h(t)/0.65; peak between 5 and 7 s, undershoot after ~11 s, ~0 by 30 s
    w = {t: E.glover_hrf_weight(t) for t in (0.5, 2.0, 4.0, 5.5, 6.0, 8.0, 12.0, 15.0, 30.0)}
    assert w[0.5] < 1e-4 < w[2.0] < w[4.0] < w[6.0]
    assert w[6.0] > w[8.0] > 0 > w[15.0]
    assert abs(w[30.0]) < 1e-3
    assert 0.25 < max(w.values()) < 0.35
    # smooth in t although the sampling grid changes with t (dt = t/50)
    a, b = E.glover_hrf_weight(6.0), E.glover_hrf_weight(6.01)
    assert abs(a - b) < 2e-3
    with pytest.raises(ValueError):
        E.glover_hrf_weight(0.0)


def test_vision_weights_onsets_follow_reference_grid():
    seen = []
    E.vision_weights(12, window=3, delay=3, hrf=lambda t: seen.append(t) or t)
    # 12 frames -> 7 connector frames, 0.5 TR apart, newest frame 2.5 TR before the target's mid-point
    assert np.allclose(seen, 1.49 * (5.5 - np.arange(0, 3.5, 0.5)))
    assert len(E.vision_weights(8, hrf=lambda t: t)) == 5               # mini geometry: 8 frames -> 5


def test_align_run_offsets_and_lang_weights():
    ep = _episode(n_tr=12, seed=1)
    bold = np.arange(14 * 5, dtype=np.float64).reshape(14, 5)            # BOLD run longer than the features
    hrf = lambda t: 1000.0 + t                                           # invertible marker
    s = E.align_run(ep, bold, window=3, delay=3, hrf=hrf)
    # streams after trimming: bold 14-5 = 9, vision 12-2 = 10, language 10 -> 9 samples
    assert len(s) == 9
    for n, smp in enumerate(s):
        assert np.array_equal(smp["timeseries"], bold[n + 5])
        assert np.array_equal(smp["vision"], ep["video_features"][n + 2])
        assert np.array_equal(smp["language"], ep["transcript_features"][n + 2])
        assert np.array_equal(smp["padvals"], ep["masking_params"][n + 2])
        d = int(smp["padvals"][2])
        tr_mid = (5 + 0.5 + n) * 1.49
        assert np.allclose(smp["lang_weights"][:d], 1000.0 + (tr_mid - ep["transcript_onsets"][n + 2][:d]))
        assert np.all(smp["lang_weights"][d:] == 0) and smp["lang_weights"].shape == (64,)
        assert smp["vis_weights"].shape == (7,)
    # the caller's onsets are not rewritten in place
    assert np.all(ep["transcript_onsets"] < 1000.0)
    # features shorter than BOLD, BOLD shorter than features, and an episode with fewer TRs than the lead
    assert len(E.align_run(ep, bold[:8], hrf=hrf)) == 3
    assert E.align_run(_episode(2, 3), bold, hrf=hrf) == []
    with pytest.raises(KeyError):
        E.align_run({k: v for k, v in ep.items() if k != "masking_params"}, bold)


def test_chunking_and_episode_keys():
    assert E.chunk_assignment(10, 4).tolist() == [0, 0, 0, 1, 1, 2, 2, 2, 3, 3]
    assert E.chunk_assignment(3, 4).tolist() == [0, 1, 2]
    m = E.episode_key_map({"ses-001": ["ses-001_task-s01e02a_timeseries", "ses-001_task-s01e02b_timeseries"],
                           "ses-002": ["ses-002_task-s01e03a_timeseries"]})
    assert m == {"s01e02a": ("ses-001", "ses-001_task-s01e02a_timeseries"),
                 "s01e02b": ("ses-001", "ses-001_task-s01e02b_timeseries"),
                 "s01e03a": ("ses-002", "ses-002_task-s01e03a_timeseries")}


def test_store_roundtrip_through_dataset(tmp_path):
    feats = {f"s01e0{k}a": _episode(9 + k, seed=10 + k) for k in range(1, 6)}
    feats["s09e99z"] = _episode(9, seed=99)                               # no BOLD for this subject: skipped
    bold = {"ses-001": {f"ses-001_task-s01e0{k}a_timeseries": np.random.default_rng(k).standard_normal((11 + k, 5))
                        for k in range(1, 4)},
            "ses-002": {f"ses-002_task-s01e0{k}a_timeseries": np.random.default_rng(k).standard_normal((11 + k, 5))
                        for k in range(4, 6)}}
    paths = E.make_lazy_loading_dsets(feats, bold, str(tmp_path), "sub-99", "s1", n_split=2, ext="npz",
                                      hrf=E.glover_hrf_weight)
    assert [p.split("/")[-1] for p in paths] == ["friends_llFile_sub-99_s1_n0.npz", "friends_llFile_sub-99_s1_n1.npz"]
    ds = VLB_Dataset(paths, geometry="mini", num_target=5)
    # episodes k=1..5 have 9+k feature TRs and 11+k BOLD TRs -> min(6+k, 7+k) = 6+k samples each; chunks 3 + 2
    assert len(ds) == sum(6 + k for k in range(1, 6))
    first = ds[0]
    ep, run = feats["s01e01a"], bold["ses-001"]["ses-001_task-s01e01a_timeseries"]
    assert np.allclose(first["timeseries"].numpy(), run[5].astype(np.float32))
    assert np.array_equal(first["vision"].numpy(), ep["video_features"][2])
    assert np.array_equal(np.asarray(first["language"]).astype(np.int64), ep["transcript_features"][2])
    assert tuple(np.asarray(first["padvals"]).astype(int)) == tuple(ep["masking_params"][2])
    # the same containers through the flat-npz episode layout tools/h5_to_npz.py --episodes writes
    np.savez(tmp_path / "feat.npz", **{f"{g}/{d}": a for g, grp in feats.items() for d, a in grp.items()})
    np.savez(tmp_path / "bold.npz", **{f"{s}/{r}": a for s, runs in bold.items() for r, a in runs.items()})
    (tmp_path / "b").mkdir()
    paths2 = E.make_lazy_loading_dsets(str(tmp_path / "feat.npz"), str(tmp_path / "bold.npz"), str(tmp_path / "b"),
                                       "sub-99", "s1", n_split=2, ext="npz", hrf=E.glover_hrf_weight)
    for a, b in zip(paths, paths2):
        fa, fb = np.load(a), np.load(b)
        assert fa.files == fb.files and all(np.array_equal(fa[k], fb[k]) for k in fa.files)
