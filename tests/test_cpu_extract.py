"""Producer side of the feature wire format (reference src/preprocessing/
videollama2_vlb_extractfeatures.py:198-349,386-508) with its engines injected: a small deterministic sentencepiece-like
tokenizer and synthetic frames stand in for the VideoLLaMA2 tokenizer and decord, which do not exist offline.  What is
checked is the LAYOUT the rest of the path consumes as given numbers: where the <video> slot, the "+2", ``inst_len``,
the dialogue tokens and the "+4" sit in the 866-long id row, that ``masking_params`` describes exactly that, and that
the rows feed ``make_weight_mask`` / the aligner unchanged."""
import os
import re

import numpy as np
import pytest
import torch

from phantom_vlb_amd import extract as X


class ToyTokenizer:
    """Word-piece tokenizer with the three properties of the real (Llama sentencepiece) one the layout depends on:
    a newline is the two tokens ['▁', '<0x0A>'], ' [/INST]' is the four tokens ['▁[', '/', 'INST', ']'], and the
    tokenisation of space-joined words is the concatenation of the words' tokenisations (long words split in pieces)."""
    bos_token_id = 1

    def __init__(self, piece=4):
        self.ids, self.piece = {}, piece

    def tokenize(self, text):
        out = []
        for word in re.findall(r"\n|[^\s]+", text):
            if word == "\n":
                out += ["▁", "<0x0A>"]
                continue
            first = True
            for piece in re.findall(r"[A-Za-z0-9']+|[^A-Za-z0-9'\s]", word):
                chunks = [piece[i:i + self.piece] for i in range(0, len(piece), self.piece)]
                for c in chunks:
                    out.append(("▁" if first else "") + c)
                    first = False
        return out

    def convert_tokens_to_string(self, tokens):
        return "".join(t.replace("▁", " ") for t in tokens).replace("<0x0A>", "\n").strip()

    def __call__(self, text):
        ids = [self.bos_token_id] + [self.ids.setdefault(t, 3 + len(self.ids)) for t in self.tokenize(text)]
        return type("Enc", (), {"input_ids": ids})()


def test_text_slots_and_the_two_tokenizer_facts_the_layout_needs():
    assert X.get_max_token(2048, 3, 4) == 866                       # 2048 - 7*169 + 1 (extractfeatures.py:204-212)
    assert X.get_max_token(128, 2, 4) == 128 - 5 * 169 + 1          # the formula, whatever the geometry
    tok = ToyTokenizer()
    assert tok.tokenize("\n") == ["▁", "<0x0A>"] and tok.tokenize(" [/INST]") == ["▁[", "/", "INST", "]"]
    ids = X.tokenizer_multimodal_token("a b <video>\nc d", tok)
    assert ids.count(X.MODAL_INDEX_VIDEO) == 1 and ids[0] == tok.bos_token_id and ids.count(tok.bos_token_id) == 1


def _transcript(n=14):
    rng = np.random.RandomState(0)
    words = "well okay so then maybe could you pass the extraordinarily long coffeehouse thing please Rachel Monica".split()
    text, wl, ol = [], [], []
    for i in range(n):
        if i in (0, 5, 6, 7, 8):                                   # silent TRs (a fully silent window at i = 7, 8)
            text.append(None); wl.append([]); ol.append([])
            continue
        k = int(rng.randint(1, 5))
        w = [words[int(rng.randint(len(words)))] for _ in range(k)]
        text.append(" ".join(w) + " ")
        wl.append(w)
        ol.append([round(1.49 * i + 0.3 * q, 2) for q in range(k)])
    return text, wl, ol


def test_episode_text_features_layout_matches_what_make_weight_mask_assumes():
    import vlb_oracle as O
    tok = ToyTokenizer()
    text, wl, ol = _transcript()
    seg_times = [0.0, 6.2, 15.0]                                    # scene changes after TR 4 (reset of the context)
    tf, to, mp = X.episode_text_features(text, wl, ol, seg_times, tok, tr=1.49, window_duration=3, max_tokens=866)
    n = len(text)
    assert tf.shape == (n, 866) and to.shape == (n, 64) and mp.shape == (n, 3)
    inst_len = len(tok.tokenize(X.INSTRUCTION))
    reset = next(i for i in range(n) if i * 1.49 > seg_times[1])    # first TR of the second scene: the window restarts there
    assert reset == 5
    for i in range(n):
        pad_len, il, dialog_len = (int(v) for v in mp[i])
        assert il == inst_len
        lo = max(0, i - 2, reset if i >= reset else 0)
        win_words = [w for j in range(lo, i + 1) for w in wl[j]]
        win_onsets = [o for j in range(lo, i + 1) for o in ol[j]]
        silent = not win_words
        # a silent window is the text "No dialogue." with TWO dummy onsets whatever its token count (extractfeatures.py:243-245):
        # masking_params says 2 there, the row itself holds the real tokens
        n_dlg = len(tok.tokenize("No dialogue.")) if silent else dialog_len
        assert dialog_len == (2 if silent else len([t for w in win_words for t in tok.tokenize(w)]))
        lay = X.layout_of(tf[i], il, n_dlg)                         # exactly one -201; 2 + inst_len + dialogue + 4 tokens behind it
        assert lay["pad_len"] == pad_len and (tf[i][866 - pad_len:] == 0).all() and (tf[i][:866 - pad_len] != 0).all()
        d0, d1 = lay["dialog"]
        inv = {v: k for k, v in tok.ids.items()}
        got = tok.convert_tokens_to_string([inv[int(t)] for t in tf[i][d0:d1]])
        if silent:
            assert got == "No dialogue." and to[i, :2].tolist() == [0.5, 1.0] and (to[i, 2:] == 0).all()
        else:                                                       # the dialogue span holds the window's words, in order, one onset per token
            assert got == " ".join(win_words)
            want = [o for w, o in zip(win_words, win_onsets) for _ in tok.tokenize(w)]
            assert np.allclose(to[i, :dialog_len], want) and (to[i, dialog_len:] == 0).all()
        # the four closing tokens are ' [/INST]' and the two behind the slot are the newline
        c0, c1 = lay["closing"]
        assert [inv[int(t)] for t in tf[i][c0:c1]] == ["▁[", "/", "INST", "]"]
        assert [inv[int(t)] for t in tf[i][lay["P"] + 1:lay["P"] + 3]] == ["▁", "<0x0A>"]
    # ... and that is the layout make_weight_mask (litmodule :178-203) turns into HRF weights: after the splice (+1182 rows
    # for the 1183 video tokens replacing the slot) the non-zero language weights sit on the dialogue tokens
    i = 3
    pad_len, il, dialog_len = (int(v) for v in mp[i])
    lang_w = torch.zeros(1, 64, dtype=torch.float64)
    lang_w[0, :dialog_len] = 0.1 + torch.arange(dialog_len, dtype=torch.float64) * 0.01
    wm = O.make_weight_mask(torch.tensor([[pad_len, il, dialog_len]]), torch.full((1, 7), 0.05, dtype=torch.float64), lang_w, 866, 2048)
    lay = X.layout_of(tf[i], il, dialog_len)
    d0, d1 = lay["dialog"]
    nz = (wm[0, 1183 + lay["P"]:] != 0).nonzero().flatten() + 1183 + lay["P"]
    assert nz.tolist() == list(range(d0 + 1182, d1 + 1182))
    assert (wm[0, lay["P"]:lay["P"] + 1183] != 0).all() and (wm[0, :lay["P"]] == 0).all()


def test_scene_context_is_cut_from_the_left_and_reset_at_scene_changes():
    tok = ToyTokenizer(piece=16)           # whole words: the fixed parts of the prompt then fit the reference's 80-token allowance
    fixed = len(X.prep_text("", "", [[]], [[]], tok, 866)[0])
    assert fixed - 1 - len(tok.tokenize("No dialogue.")) < 80       # (the real tokenizer: 73, extractfeatures.py:258-259)
    n = 400
    text = ["alpha beta gamma delta "] * n
    wl = [["alpha", "beta", "gamma", "delta"]] * n
    ol = [[0.1, 0.2, 0.3, 0.4]] * n
    tf, to, mp = X.episode_text_features(text, wl, ol, [0.0, 1e9], tok, max_tokens=866)
    assert tf.shape == (n, 866)
    pads = mp[:, 0]
    assert pads[0] > pads[50] and pads[-1] >= 0                      # the context grows until the 80-token allowance binds
    assert (pads[300:] == pads[300]).all()                           # ... and then stays cut to the same length
    tf2, _, mp2 = X.episode_text_features(text[:20], wl[:20], ol[:20], [0.0, 9.0, 1e9], tok, max_tokens=866)
    assert mp2[8, 0] > mp2[6, 0]                                     # the scene change at 9 s (TR 7) emptied the context
    assert X.scene_onsets([1, 1, 2, 2, 3], [0.0, 4.0, 9.5, 12.0, 20.0]) == [0.0, 9.5, 20.0]


class _Frames:
    """decord-like frame source: get_batch(indices) -> [n, H, W, 3] uint8; frame f has value f % 251 in channel 0."""

    def __init__(self, n, h=48, w=64):
        self.n, self.h, self.w = n, h, w

    def get_batch(self, idx):
        out = np.zeros((len(idx), self.h, self.w, 3), np.uint8)
        for k, f in enumerate(idx):
            out[k, ..., 0] = f % 251
            out[k, ..., 1] = 100
            out[k, ..., 2] = 200
        return out


def test_video_windows_sampling_padding_and_normalisation():
    fps, nfr, tr = 29.97, 600, 1.49
    ends = X.tr_end_times(nfr, fps, tr)
    assert ends[0] == pytest.approx(1.49) and len(ends) == int(np.ceil(nfr / fps / tr)) - 1
    idx = X.window_frame_indices(ends[5], 3, fps, nfr, tr, 4)        # a full 3-TR window: 12 frames inside it, increasing
    assert len(idx) == 12 and idx == sorted(idx) and idx[0] >= int((ends[5] - 3 * tr) * fps) - 1 and idx[-1] <= int(ends[5] * fps) - 1
    assert len(X.window_frame_indices(ends[0], 3, fps, nfr, tr, 4)) == 4          # first TR: only one TR of video exists
    assert X.frame_sample(10, 5).tolist() == [1, 3, 5, 6, 8]          # segment centres 0.9, 2.7, 4.5, 6.3, 8.1
    v = X.extract_video_chunk(_Frames(nfr), ends[0], 3, fps, nfr, tr, 4, size=336)
    assert v.shape == (12, 3, 336, 336) and v.dtype == np.float32
    # the 8 completing frames are black frames padded with the mean colour: rows outside the letter-box are ~0 after
    # normalisation, rows inside are (0 - mean) / std
    top = v[11, :, 0, 0]
    assert np.allclose(top, (np.array([int(m * 255) for m in X.CLIP_MEAN]) / 255.0 - X.CLIP_MEAN) / X.CLIP_STD, atol=1e-6)
    mid = v[11, :, 168, 168]
    assert np.allclose(mid, (0 - np.array(X.CLIP_MEAN)) / X.CLIP_STD, atol=1e-5)
    sq = X.expand2square(np.full((2, 4, 3), 9, np.uint8), (1, 2, 3))
    assert sq.shape == (4, 4, 3) and (sq[0] == (1, 2, 3)).all() and (sq[1:3] == 9).all() and (sq[3] == (1, 2, 3)).all()
    # a real frame keeps its content: channel 1 is 100 everywhere inside the letter-box
    assert v[0, 1, 168, 168] == pytest.approx((100 / 255.0 - X.CLIP_MEAN[1]) / X.CLIP_STD[1], abs=1e-5)


def test_extract_episode_round_trips_through_the_aligner_and_the_dataset(tmp_path):
    """producer -> per-episode file -> make_lazy_loading_dsets (the consume side) -> VLB_Dataset: the four arrays arrive as
    the six sample tensors of the lazy-load schema."""
    from phantom_vlb_amd import episodes as E
    tok = ToyTokenizer()
    text, wl, ol = _transcript(12)
    fps, nfr = 10.0, 190                                             # 19 s -> 12 TR ends
    arrays = X.extract_episode(text, wl, ol, [0.0, 1e9], tok, _Frames(nfr, 24, 32), fps, nfr, size=28, model_max_length=2048)
    assert set(arrays) == set(E.EPISODE_KEYS)
    assert arrays["video_features"].shape == (12, 12, 3, 28, 28) and arrays["transcript_features"].shape == (12, 866)
    path = X.write_episode(str(tmp_path / "features.npz"), "s01e01a", arrays)
    back = E.open_groups(path)
    assert "s01e01a" in back and np.array_equal(back["s01e01a"]["masking_params"], arrays["masking_params"])
    bold = {"ses-001": {"ses-001_task-s01e01a_timeseries": np.random.RandomState(1).randn(12, 16).astype(np.float32)}}
    out = E.make_lazy_loading_dsets(path, bold, str(tmp_path / "ll"), "sub-01", "s1", n_split=1, ext="npz", hrf=lambda t: 0.1)
    store = np.load(out[0])
    nsmp = int(store["dset_len"][0])
    assert nsmp == 12 - 2 - 3                                        # window-1 inputs without a full window, `delay` targets missing
    assert store["0_language"].shape == (866,) and store["0_vision"].shape == (12, 3, 28, 28) and store["0_padvals"].shape == (3,)
    assert np.array_equal(store["0_language"], arrays["transcript_features"][2])
    assert np.array_equal(store["0_padvals"], arrays["masking_params"][2])
