"""This repository's host code and oracle against fixtures produced by RUNNING the reference's own functions
(oracle/gen_ref_fixtures.py + oracle/gen_ref_fixtures_py39.py, build container only; SURVEY.md §8 rows a2, a15, a17, c, f4).

The reference files behind each fixture:
  ref_weight_mask.npz    src/litmodule/videollama2_vlb_litmodule.py:178-203  (make_weight_mask)
  ref_linear_names.json  src/litmodule/videollama2_vlb_litmodule.py:36-55    (find_all_linear_names)
  ref_datamodule.json    src/datamodule/videollama2_vlb_datamodule.py:24-28,65-109,130-153
  ref_pipeline/          src/preprocessing/videollama2_vlb_extractfeatures.py:88-145,198-508 and
                         src/preprocessing/videollama2_vlb_lazyloading.py:51-169 - the HDF5 files in there were WRITTEN by
                         those scripts (real h5py 3.3), with the engines of tests/toy_engines.py injected.
Nothing here reads /root/reference; the HIP mask kernel is checked against the same file in tests/test_gpu_kernels.py.
"""
import json
import os

import numpy as np
import pytest
import torch

import toy_engines as T
import vlb_oracle as O
from phantom_vlb_amd import episodes as E
from phantom_vlb_amd import extract as X
from phantom_vlb_amd.datamodule import VLB_Dataset, VLBDataModuleConfig, VLBDatasets, get_idx, open_h5

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PIPE = os.path.join(GOLDEN, "ref_pipeline")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(PIPE, "meta.json")) as f:
        return json.load(f)


def bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


# --------------------------------------------------------------------------------------------------
# a2 / c: the oracle's make_weight_mask == the reference's, bit for bit in the reference's dtype (bf16)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,tokens_per_frame", [("g7b", 169), ("g2f", 169)])
def test_oracle_weight_mask_equals_the_reference(tag, tokens_per_frame):
    z = np.load(os.path.join(GOLDEN, "ref_weight_mask.npz"))
    pv, vw, lw = (torch.from_numpy(z[f"{tag}_{k}"]) for k in ("padvals", "vis_weights", "lang_weights"))
    want = z[f"{tag}_rows_bf16_bits"]
    assert pv.shape[0] >= (32 if tag == "g7b" else 4)
    rows = O.make_weight_mask(pv, vw, lw, int(z[f"{tag}_lang_len"]), int(z[f"{tag}_max_len"]), tokens_per_frame)
    # the reference casts every segment to bf16 before concatenating (:190-194); the oracle keeps fp32 and the callers round
    assert np.array_equal(bf16_bits(rows), want)
    if tag == "g7b":
        tri = {tuple(int(v) for v in r) for r in pv}
        assert {(0, 9, 0), (300, 9, 58), (0, 0, 0)} <= tri                     # the corners VERDICT r03 names


# --------------------------------------------------------------------------------------------------
# a15: find_all_linear_names
# --------------------------------------------------------------------------------------------------
def test_find_all_linear_names_equals_the_reference_on_a_mistral_tree_with_decoys():
    pytest.importorskip("transformers")
    import gen_ref_fixtures as G                     # oracle/ (test infrastructure): the tree builder only
    from phantom_vlb_amd.litmodule import find_all_linear_names
    with open(os.path.join(GOLDEN, "ref_linear_names.json")) as f:
        ref = json.load(f)
    m = G.linear_names_model()
    assert [[n, type(mod).__name__] for n, mod in m.named_modules() if isinstance(mod, torch.nn.Linear)] == ref["linear_modules"]
    assert sorted(find_all_linear_names(m)) == ref["names"] == sorted(
        ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"])


# --------------------------------------------------------------------------------------------------
# a17: get_idx, file split, dataset items
# --------------------------------------------------------------------------------------------------
def test_datamodule_split_and_items_equal_the_reference(monkeypatch):
    with open(os.path.join(GOLDEN, "ref_datamodule.json")) as f:
        ref = json.load(f)
    for c in ref["get_idx"]:
        assert get_idx([tuple(r) for r in c["ranges"]], c["val"]) == c["idx"]
    monkeypatch.setenv("SCRATCH_PATH", PIPE)
    first = None
    for s in ref["splits"]:
        cfg = VLBDataModuleConfig(lazyload_path=s["lazyload_path"], subject="sub-01", seasons=s["seasons"], delay=3, window=3,
                                  random_state=s["random_state"], shuffle_val_data=False, batch_size=2, num_workers=0)
        ds = VLBDatasets(cfg)
        assert ds.dset_names == s["dset_names"]
        assert (len(ds.train), len(ds.val)) == (s["train_len"], s["val_len"])
        assert [list(r) for r in ds.train.ranges] == s["train_ranges"] and [list(r) for r in ds.val.ranges] == s["val_ranges"]
        first = first or ds
    for it in ref["items"]:
        got = first.train[it["idx"]]
        assert {k: str(v.dtype) for k, v in got.items()} == it["dtypes"]
        assert {k: list(v.shape) for k, v in got.items()} == it["shapes"]
        assert got["language"].tolist() == it["language"] and got["padvals"].tolist() == it["padvals"]
        assert [float(x) for x in got["timeseries"]] == it["timeseries"]
        assert float(got["vision"].double().sum()) == it["vision_sum"]
        assert [float(x) for x in got["vis_weights"]] == it["vis_weights"]
        assert [float(x) for x in got["lang_weights"]] == it["lang_weights"]


# --------------------------------------------------------------------------------------------------
# f4 producer: text
# --------------------------------------------------------------------------------------------------
def test_get_max_token_and_scene_onsets_equal_the_reference(meta):
    for c in meta["text"]["get_max_token"]:
        assert X.get_max_token(c["model_max_length"], c["window_duration"], c["frames_per_tr"]) == c["max_tokens"]
    for c in meta["text"]["get_sceneonsets"]:
        assert X.scene_onsets(c["scenes"], c["onsets"]) == c["scene_onsets"]


def test_prep_text_equals_the_reference_including_the_budget_corner_cases(meta):
    names = set()
    for c in meta["text"]["prep_text"]:
        ids, onsets, inst_len = X.prep_text(c["scene_text"], c["seg_text"], c["word_lists"], c["onset_lists"],
                                            T.ToyTokenizer(), c["max_tokens"])
        assert [int(i) for i in ids] == c["input_ids"], c["name"]
        assert [float(o) for o in onsets] == c["token_onsets"] and inst_len == c["inst_len"], c["name"]
        names.add(c["name"])
    # the scene budget <= 0 (reference: `tokens[-max_scene_length:]` keeps everything at 0 and drops the first |k| below)
    assert {"budget_zero", "budget_negative", "long_scene", "silent"} <= names


# --------------------------------------------------------------------------------------------------
# f4 producer: whole episodes == the per-episode HDF5 file the reference's extract_features_videollama2 wrote
# --------------------------------------------------------------------------------------------------
def test_extract_episode_equals_the_file_the_reference_wrote(meta):
    f = open_h5(os.path.join(PIPE, "features_s1.h5"))
    assert sorted(f.keys()) == sorted(meta["episodes"])
    size = meta["processor_size"]
    for ep, spec in meta["episodes"].items():
        text, wl, ol = T.synthetic_transcript(spec["n_tr"], spec["seed"], tuple(spec["silent"]))
        seg_times = X.scene_onsets([s for s, _ in spec["seg"]], [o for _, o in spec["seg"]])
        nfr, fps, h, w = spec["video"]
        arrays = X.extract_episode(text, wl, ol, seg_times, T.ToyTokenizer(), T.ToyVideoReader(n=nfr, fps=fps, h=h, w=w), fps, nfr,
                                   tr=1.49, window_duration=3, frames_per_tr=4, model_max_length=2048, size=size)
        assert set(arrays) == set(E.EPISODE_KEYS) == set(f[ep].keys())
        for k in ("transcript_features", "transcript_onsets", "masking_params"):
            ref = np.array(f[ep][k])
            assert arrays[k].dtype == ref.dtype and arrays[k].shape == ref.shape, (ep, k)
            assert np.array_equal(arrays[k], ref), (ep, k)
        ref = np.array(f[ep]["video_features"])
        assert arrays["video_features"].dtype == ref.dtype == np.float32 and arrays["video_features"].shape == ref.shape
        # same frames, same padding, same order; the reference's processor rescales in f64 (x * (1/255)), ours divides in f32
        assert np.abs(arrays["video_features"] - ref).max() <= 3e-7, ep


# --------------------------------------------------------------------------------------------------
# f4 consumer: the aligner == the sample stores the reference's make_lazy_loading_dsets wrote
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["d3w3", "d2w3"])
def test_aligner_equals_the_sample_stores_the_reference_wrote(meta, tag, tmp_path):
    a = meta["aligner"][tag]
    out = E.make_lazy_loading_dsets(os.path.join(PIPE, "features_s1.h5"), os.path.join(PIPE, "bold_sub-01.h5"), str(tmp_path),
                                    "sub-01", "s1", n_split=a["n_split"], delay=a["delay"], window=a["window"], ext="npz",
                                    hrf=T.toy_hrf)
    assert [os.path.basename(p).replace(".npz", ".h5") for p in out] == a["files"]
    total = 0
    for p, fn in zip(out, a["files"]):
        ref = open_h5(os.path.join(PIPE, f"lazy_{tag}", fn))
        got = np.load(p)
        n = int(np.array(ref["dset_len"])[0])
        assert int(got["dset_len"][0]) == n and np.array(ref["dset_len"]).dtype == got["dset_len"].dtype
        total += n
        for i in range(n):
            assert sorted(ref[f"{i}"].keys()) == sorted(f"{i}_{m}" for m in E.SAMPLE_MODS)
            for m in E.SAMPLE_MODS:
                r, g = np.array(ref[f"{i}"][f"{i}_{m}"]), got[f"{i}_{m}"]
                assert r.dtype == g.dtype and r.shape == g.shape, (fn, i, m)
                assert np.array_equal(r, g), (fn, i, m)
    assert total > 20                               # three episodes with BOLD runs; the fourth (no run) is skipped
    # and the reference-written store feeds this package's dataset unchanged
    ds = VLB_Dataset([os.path.join(PIPE, f"lazy_{tag}", fn) for fn in a["files"]])
    assert len(ds) == total and ds[0]["vision"].dtype == torch.float32 and ds[total - 1]["padvals"].shape == (3,)
