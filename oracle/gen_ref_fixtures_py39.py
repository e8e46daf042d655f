"""Run the reference's OWN preprocessing scripts on synthetic episodes and write tests/golden/ref_pipeline/.
TEST INFRASTRUCTURE - build container only; run by oracle/gen_ref_fixtures.py under /opt/conda/bin/python3.9
(the interpreter of this image that has the real h5py 3.3 and pandas; it has no torch).

What runs, unmodified, loaded by file path from /root/reference:
  * src/preprocessing/videollama2_vlb_extractfeatures.py: get_max_token (:198-212), get_sceneonsets (:131-145),
    get_input_paths (:88-112), prep_text (:215-300), load_video (:303-317), extract_video_chunk (:320-349) and the whole
    per-episode loop extract_features_videollama2 (:352-508), which writes the per-episode gzip-4 HDF5 file itself;
  * src/preprocessing/videollama2_vlb_lazyloading.py: make_lazy_loading_dsets (:51-169), which reads that file plus a
    BOLD file and writes the lazy-load sample stores itself.
What is injected (tests/toy_engines.py; the same objects the tests hand to this repository's code): the tokenizer, the
video reader, the CLIP processor, frame_sample / expand2square / tokenizer_multimodal_token (VideoLLaMA2 functions the
script imports inside its __main__ block; the submodule is empty) and the HRF (nilearn is absent) - plus no-op stand-ins
for `torch.device`, `tqdm` and the two by-NAME loaders prep_video_processor / prep_tokenizer.
The reference text never travels: only the synthetic inputs and the files its functions wrote do.
"""
import importlib.util
import json
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.environ.get("VLB_REF_PIPELINE_OUT") or os.path.join(ROOT, "tests", "golden", "ref_pipeline")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import toy_engines as T  # noqa: E402


def load_by_path(name, path, inject=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    for k, v in (inject or {}).items():
        setattr(mod, k, v)
    spec.loader.exec_module(mod)
    return mod


def load_repo_multimodal_tokenizer():
    """extract.py's restatement of tokenizer_multimodal_token, loaded by path (importing the package needs torch):
    the engine both sides share, see the module docstring."""
    m = load_by_path("_vlb_extract", os.path.join(ROOT, "phantom_vlb_amd", "extract.py"))
    return lambda prompt, tokenizer, modal_token, return_tensors=None: m.tokenizer_multimodal_token(prompt, tokenizer, modal_token)


EPISODES = {                     # episode -> (n_tr of transcript, seed, silent TRs, scene segmentation rows, video spec)
    "s01e01a": (14, 0, (0, 5, 6, 7, 8), [(1, 0.0), (1, 3.0), (2, 6.2), (3, 15.0)], (190, 10.0, 6, 8)),
    "s01e02b": (12, 1, (3,), [(1, 0.0), (2, 9.0)], (170, 10.0, 8, 6)),
    "s01e03a": (10, 2, (), [(1, 0.0), (2, 4.0), (2, 8.0), (4, 11.0)], (150, 10.0, 8, 8)),
    "s01e04a": (9, 3, (0, 1, 2), [(1, 0.0), (2, 100.0)], (140, 10.0, 5, 8)),       # no BOLD run: must be skipped
}
V = 16                           # BOLD targets per TR


def write_inputs(work):
    import pandas as pd
    for d in ("transcripts", "seg", "video"):
        os.makedirs(os.path.join(work, d), exist_ok=True)
    inputs = {}
    for ep, (n, seed, silent, seg, vid) in EPISODES.items():
        text, wl, ol = T.synthetic_transcript(n, seed, silent)
        pd.DataFrame({"text_per_tr": text, "words_per_tr": [repr(w) for w in wl], "onsets_per_tr": [repr(o) for o in ol]}
                     ).to_csv(os.path.join(work, "transcripts", f"friends_{ep}.tsv"), sep="\t", index=False)
        # get_input_paths (:103) looks the segmentation up under the name with "s0" -> "s"
        pd.DataFrame({"scene": [s for s, _ in seg], "onset": [o for _, o in seg]}
                     ).to_csv(os.path.join(work, "seg", f"friends_{ep}_manualseg.tsv".replace("s0", "s")), sep="\t", index=False)
        open(os.path.join(work, "video", f"friends_{ep}.mkv"), "w").close()
        T.TOY_VIDEOS[f"friends_{ep}.mkv"] = vid
        inputs[ep] = {"n_tr": n, "seed": seed, "silent": list(silent), "seg": seg, "video": list(vid)}
    return inputs


def run_extract(work, size):
    from PIL import Image
    tok = T.ToyTokenizer()
    stub_torch = types.SimpleNamespace(device=lambda s: s, cuda=types.SimpleNamespace(is_available=lambda: False))
    inject = dict(torch=stub_torch, tqdm=lambda it, **kw: it, Image=Image, VideoReader=T.ToyVideoReader, cpu=T.toy_cpu,
                  frame_sample=T.toy_frame_sample, expand2square=T.toy_expand2square,
                  tokenizer_multimodal_token=load_repo_multimodal_tokenizer())
    X = load_by_path("_ref_extract", os.path.join(REF, "src", "preprocessing", "videollama2_vlb_extractfeatures.py"), inject)

    def prep_video_processor(a):            # the reference's loads the CLIP tower by model NAME (:148-177): unavailable offline
        a.video_processor = T.ToyProcessor(size)
        return a
    X.prep_video_processor = prep_video_processor
    X.prep_tokenizer = lambda a: tok        # AutoTokenizer.from_pretrained(name) (:180-195): unavailable offline
    feat = os.path.join(OUT, "features_s1.h5")
    args = types.SimpleNamespace(
        input_transcript_path=os.path.join(work, "transcripts"), input_seg_path=os.path.join(work, "seg"),
        input_video_path=os.path.join(work, "video"), lazy_load_path=feat, model_max_length=2048, bf16=True,
        frames_per_tr=4, tr=1.49, window_duration=3)
    X.extract_features_videollama2(args)
    return X, tok, feat


def prep_text_cases(X):
    """prep_text / get_max_token / get_sceneonsets on their own, including the paths the episode loop does not reach:
    a scene context far longer than the budget (cut from the left) and a budget <= 0 (the reference's negative / zero
    slice, `tokens[-max_scene_length:]`, kept as is)."""
    import pandas as pd
    cases = []
    long_scene = " ".join(f"word{i} and" for i in range(600))
    specs = [
        ("plain", "hello there Monica", "so then ", [["so", "then"]], [[1.0, 1.5]], 866),
        ("silent", "earlier line", "", [[], [], []], [[], [], []], 866),
        ("long_scene", long_scene, "okay Rachel ", [[], ["okay"], ["Rachel"]], [[], [4.1], [5.2]], 866),
        ("budget_zero", "a b c d e f g h", "x y ", [["x", "y"]], [[0.1, 0.2]], 82),            # 82 - (80 + 2) = 0
        ("budget_negative", "a b c d e f g h", "x y z ", [["x", "y", "z"]], [[0.1, 0.2, 0.3]], 80),   # 80 - 83 = -3
        ("long_words", "", "extraordinarily coffeehouse ", [["extraordinarily"], ["coffeehouse"]], [[2.0], [3.0]], 300),
    ]
    for name, scene, seg, wl, ol, mx in specs:
        tok = T.ToyTokenizer()
        ids, onsets, inst_len = X.prep_text(scene, seg, wl, ol, tok, mx)
        cases.append(dict(name=name, scene_text=scene, seg_text=seg, word_lists=wl, onset_lists=ol, max_tokens=mx,
                          input_ids=[int(i) for i in ids], token_onsets=[float(o) for o in onsets], inst_len=int(inst_len)))
    mt = []
    for mml, win, fpt in [(2048, 3, 4), (2048, 2, 4), (4096, 3, 4), (1024, 1, 2), (2048, 3, 3)]:
        mt.append(dict(model_max_length=mml, window_duration=win, frames_per_tr=fpt,
                       max_tokens=int(X.get_max_token(types.SimpleNamespace(model_max_length=mml, window_duration=win, frames_per_tr=fpt)))))
    so = []
    for scenes, onsets in [([1, 1, 2, 2, 3], [0.0, 4.0, 9.5, 12.0, 20.0]), ([5, 5, 5], [1.0, 2.0, 3.0]), ([2, 1, 2, 1], [0.5, 1.5, 2.5, 3.5])]:
        so.append(dict(scenes=scenes, onsets=onsets,
                       scene_onsets=[float(v) for v in X.get_sceneonsets(pd.DataFrame({"scene": scenes, "onset": onsets}))]))
    return dict(prep_text=cases, get_max_token=mt, get_sceneonsets=so)


def run_aligner(feat):
    import h5py
    src = types.ModuleType("src")
    src.get_hrf_weight = T.toy_hrf
    sys.modules["src"] = src
    L = load_by_path("_ref_lazy", os.path.join(REF, "src", "preprocessing", "videollama2_vlb_lazyloading.py"))
    bold = os.path.join(OUT, "bold_sub-01.h5")
    rng = np.random.RandomState(7)
    runs = {"ses-001": {"ses-001_task-s01e01a_timeseries": 15, "ses-001_task-s01e02b_timeseries": 11},
            "ses-002": {"ses-002_task-s01e03a_timeseries": 13, "ses-002_task-s02e01a_timeseries": 9}}
    with h5py.File(bold, "w") as f:
        for ses, rr in runs.items():
            g = f.create_group(ses)
            for run, n in rr.items():
                g.create_dataset(run, data=rng.randn(n, V).astype(np.float32))
    outs = {}
    for tag, n_split, delay, window in [("d3w3", 2, 3, 3), ("d2w3", 1, 2, 3)]:
        d = os.path.join(OUT, f"lazy_{tag}")
        os.makedirs(d)
        L.make_lazy_loading_dsets(types.SimpleNamespace(
            timeseries_path=bold, features_path=feat, lazyload_path=d, subject="sub-01", season="s1",
            n_split=n_split, delay=delay, window=window))
        outs[tag] = dict(n_split=n_split, delay=delay, window=window, files=sorted(os.listdir(d)))
    return outs


def main():
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    work = os.path.join("/tmp", f"vlb_ref_fixture_work_{os.getpid()}")
    if os.path.isdir(work):
        shutil.rmtree(work)
    os.makedirs(work)
    size = 8            # = the longer side of every toy video: expand2square output needs no resize (PIL-version independent)
    inputs = write_inputs(work)
    X, tok, feat = run_extract(work, size)
    meta = dict(episodes=inputs, processor_size=size, bold_targets=V, text=prep_text_cases(X), aligner=run_aligner(feat),
                reference_files=["src/preprocessing/videollama2_vlb_extractfeatures.py:88-145,198-508",
                                 "src/preprocessing/videollama2_vlb_lazyloading.py:51-169"])
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    total = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(OUT) for fn in fns)
    print(f"wrote {OUT}: {total / 1e6:.2f} MB")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
