"""Feature extraction, producer side of the per-episode wire format.

The reference's ``src/preprocessing/videollama2_vlb_extractfeatures.py`` turns one Friends episode (an ``.mkv``, a
per-TR transcript table and a scene segmentation) into four arrays,

    video_features       (n_tr, 12, 3, 336, 336) f32   CLIP-normalised frames of the 3-TR window ending at each TR
    transcript_features  (n_tr, 866)             int   prompt token ids, one -201 <video> slot, right-padded with 0
    transcript_onsets    (n_tr, 64)              f64   onset (s) of every dialogue token of the window, 0-padded
    masking_params       (n_tr, 3)               int   (pad_len, inst_len, dialog_len)

which ``episodes.py`` (the consume side) aligns with BOLD into the lazy-load sample store.  This module is that
producer as a library with its three external engines INJECTED, because none of them exists offline:

  * ``tokenizer`` - any object with the Hugging Face surface the reference uses (``tokenize``,
    ``convert_tokens_to_string``, ``__call__(text).input_ids``, ``bos_token_id``): the real one is
    ``AutoTokenizer.from_pretrained("DAMO-NLP-SG/VideoLLaMA2-7B")`` (extractfeatures.py:180-195);
  * ``frames`` - any object with ``get_batch(indices) -> uint8 array [n, H, W, 3]`` (decord's ``VideoReader`` after
    ``.asnumpy()``, extractfeatures.py:303-317,336), or a plain ndarray of frames;
  * the CLIP image processor - restated here for the fixed ``openai/clip-vit-large-patch14-336`` settings (bicubic
    resize of the square-padded frame to 336, rescale 1/255, normalise with the OpenAI CLIP mean / std).

What the module pins down (and the tests check with a small deterministic tokenizer): the prompt / token LAYOUT that
defines the constants ``make_weight_mask`` consumes as given numbers (litmodule :178-203) -

    ids = [P prompt tokens] [-201] [2 tokens of "\\n"] [inst_len instruction tokens] [dialog_len dialogue tokens]
          [4 tokens of " [/INST]"] [pad_len zeros],          len(ids) == max_text_tokens (866 for the 7B geometry)

UNPINNED (recalled from the public VideoLLaMA2 sources, which are absent from /root/reference - SURVEY Appendix B):
``tokenizer_multimodal_token``, ``frame_sample``, ``expand2square`` and the chat template.  Every one of them is a
parameter or a small function here, so a maintainer with the real sources can swap them in.
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

MODAL_INDEX_VIDEO = -201            # videollama2 constants.py MODAL_INDEX_MAP["<video>"]; extractfeatures.py:235-236
MODAL_TOKEN = "<video>"
INSTRUCTION = "Here are the words spoken in the video:"        # extractfeatures.py:273
TOKENS_PER_FRAME = 13 * 13          # connector output grid (extractfeatures.py:204-209)
ONSET_SLOTS = 64                    # extractfeatures.py:447-451
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# --------------------------------------------------------------------------------------------------
# text
# --------------------------------------------------------------------------------------------------
def get_max_token(model_max_length: int = 2048, window_duration: int = 3, frames_per_tr: int = 4) -> int:
    """extractfeatures.py:198-212: text slots = model_max_length - (floor(frames/2)+1)*169, +1 for the <video> placeholder
    that the model replaces (866 for 2048 / 12 frames)."""
    num_frames = window_duration * frames_per_tr
    return model_max_length - (math.floor(num_frames / 2) + 1) * TOKENS_PER_FRAME + 1


def tokenizer_multimodal_token(prompt: str, tokenizer, multimodal_token: str = MODAL_TOKEN) -> List[int]:
    """VideoLLaMA2 ``mm_utils.tokenizer_multimodal_token`` (UNPINNED restatement): tokenize the text between the modal
    tokens separately, keep the BOS of the first chunk only and put the modal index (-201) between the chunks."""
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split(multimodal_token)]
    ids: List[int] = []
    offset = 0
    bos = getattr(tokenizer, "bos_token_id", None)
    if chunks and chunks[0] and bos is not None and chunks[0][0] == bos:
        offset = 1
        ids.append(chunks[0][0])
    for i, c in enumerate(chunks):
        if i:
            ids.append(MODAL_INDEX_VIDEO)
        ids.extend(c[offset:])
    return ids


def default_chat_template(messages: Sequence[Mapping[str, str]]) -> str:
    """Llama-2 / Mistral-instruct layout of a (system, user) pair as the reference's prompt needs it: the system content
    already carries its own <<SYS>> markers (extractfeatures.py:282-289), the user turn ends in ' [/INST]' (the "+4 tokens
    after", extractfeatures.py:278-279).  UNPINNED: the real string comes from the checkpoint's tokenizer_config.json."""
    system = "".join(m["content"] for m in messages if m["role"] == "system")
    user = "".join(m["content"] for m in messages if m["role"] == "user")
    return f"[INST] {system}\n\n{user} [/INST]"


def prep_text(scene_text: str, seg_text: str, word_lists: Sequence[Sequence[str]], onset_lists: Sequence[Sequence[float]],
              tokenizer, max_tokens: int, chat_template: Optional[Callable] = None) -> Tuple[List[int], List[float], int]:
    """extractfeatures.py:215-300 -> (input_ids with one -201, onset of every dialogue token, inst_len).

    Dialogue of the frame window = the words of its TRs joined by single spaces, each token inheriting its word's onset
    (a silent window becomes the text "No dialogue." with TWO dummy onsets 0.5 / 1.0, whatever number of tokens that text has:
    ``masking_params`` then says dialog_len = 2 - a quirk of the reference kept as is); the scene text spoken before the window is cut from
    the LEFT so that everything fits ``max_tokens`` with an 80-token allowance for instructions and system message (a budget
    <= 0 is not a cut to nothing: Python's ``tokens[-0:]`` / ``tokens[k:]``, unreachable with 866 slots; pinned against the
    reference's own output in tests/test_cpu_ref_fixtures.py)."""
    all_words = [w for wl in word_lists for w in wl]
    all_onsets = [o for ol in onset_lists for o in ol]
    if len(all_words) != len(all_onsets):
        raise ValueError("prep_text: words and onsets differ in number")
    if seg_text == "":
        seg_dialog, token_onsets = "No dialogue.", [0.5, 1.0]
    else:
        token_onsets, seg_dialog = [], ""
        for w, o in zip(all_words, all_onsets):
            token_onsets += [o] * len(tokenizer.tokenize(w))
            seg_dialog += f"{w} "
        if len(token_onsets) != len(tokenizer.tokenize(seg_dialog.strip())):
            raise ValueError("prep_text: per-word and whole-dialogue tokenisations disagree (extractfeatures.py:252)")
    tokens = tokenizer.tokenize(scene_text.strip())
    seg_len = len(tokenizer.tokenize(seg_dialog.strip()))
    max_scene = max_tokens - (80 + seg_len)
    if len(tokens) > max_scene:
        tokens = tokens[-max_scene:]       # the reference's slice as is: a budget of 0 keeps everything, -k drops the first k
    background = tokenizer.convert_tokens_to_string(tokens).strip()
    inst_len = len(tokenizer.tokenize(INSTRUCTION.strip()))
    instructions = f"{INSTRUCTION.strip()} {seg_dialog.strip()}"
    messages = [
        {"role": "system", "content": ("<<SYS>>\nThis video is from a scene from the TV show Friends. Try to understand what is "
                                       "happening in the video.\nFor context, here is the dialogue that was spoken just before "
                                       f"the video onset: {background}.\n<</SYS>>")},
        {"role": "user", "content": MODAL_TOKEN + "\n" + instructions.strip()},
    ]
    if chat_template is None and hasattr(tokenizer, "apply_chat_template"):
        prompt = tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=False)
    else:
        prompt = (chat_template or default_chat_template)(messages)
    return tokenizer_multimodal_token(prompt, tokenizer, MODAL_TOKEN), token_onsets, inst_len


def layout_of(ids: Sequence[int], inst_len: int, dialog_len: int) -> dict:
    """The segment boundaries ``make_weight_mask`` (litmodule :178-203) assumes, read back from a padded id row:
    ``P`` prompt tokens, the -201 slot, ``2 + inst_len`` tokens, ``dialog_len`` dialogue tokens, ``4`` closing tokens,
    ``pad_len`` zeros.  Raises when the row does not have that shape (the reference's implicit contract)."""
    ids = list(ids)
    n = len(ids)
    pad_len = 0
    while pad_len < n and ids[n - 1 - pad_len] == 0:
        pad_len += 1
    slots = [i for i, t in enumerate(ids) if t == MODAL_INDEX_VIDEO]
    if len(slots) != 1:
        raise ValueError(f"expected exactly one {MODAL_INDEX_VIDEO} slot, found {len(slots)}")
    P = slots[0]
    body = n - pad_len - P - 1
    if body != 2 + inst_len + dialog_len + 4:
        raise ValueError(f"layout mismatch: {body} tokens after the video slot, expected 2 + {inst_len} + {dialog_len} + 4")
    d0 = P + 1 + 2 + inst_len
    return {"P": P, "pad_len": pad_len, "dialog": (d0, d0 + dialog_len), "closing": (d0 + dialog_len, n - pad_len)}


def scene_onsets(scenes: Sequence, onsets: Sequence[float]) -> List[float]:
    """extractfeatures.py:131-145: onset of the first segment of every scene, in order of appearance."""
    seen, out = set(), []
    for sc, on in zip(scenes, onsets):
        if sc not in seen:
            seen.add(sc)
            out.append(float(on))
    return out


def episode_text_features(text_per_tr: Sequence[Optional[str]], words_per_tr: Sequence[Sequence[str]],
                          onsets_per_tr: Sequence[Sequence[float]], seg_times: Sequence[float], tokenizer,
                          tr: float = 1.49, window_duration: int = 3, max_tokens: int = 866,
                          chat_template: Optional[Callable] = None):
    """extractfeatures.py:386-455: one row per TR of the transcript table -> (transcript_features [n,max_tokens] int64,
    transcript_onsets [n,64] f64, masking_params [n,3] int64).  The text of the ``window_duration`` most recent TRs is the
    frame window's dialogue; what was spoken earlier in the same scene is the system message's context; a scene change
    (``seg_times``) resets both."""
    run_tokens, run_times, params = [], [], []
    scene_chunk, j = "", 1
    tr_chunk: List[str] = [""] * window_duration
    tr_words: List[Sequence[str]] = [[]] * window_duration
    tr_onsets: List[Sequence[float]] = [[]] * window_duration
    for i in range(len(text_per_tr)):
        if j < len(seg_times) and (i * tr) > seg_times[j] and j < (len(seg_times) - 1):
            scene_chunk = ""
            tr_chunk, tr_words, tr_onsets = [""] * window_duration, [[]] * window_duration, [[]] * window_duration
            j += 1
        txt = text_per_tr[i]
        if txt is None or (isinstance(txt, float) and math.isnan(txt)):
            i_text, i_words, i_times = "", [], []
        else:
            i_text, i_words, i_times = str(txt), list(words_per_tr[i]), list(onsets_per_tr[i])
            if len(i_words) != len(i_times):
                raise ValueError(f"TR {i}: {len(i_words)} words but {len(i_times)} onsets")
        scene_chunk += tr_chunk[0]
        tr_chunk = tr_chunk[1:] + [i_text]
        tr_words = tr_words[1:] + [i_words]
        tr_onsets = tr_onsets[1:] + [i_times]
        ids, onsets, inst_len = prep_text(scene_chunk, "".join(tr_chunk), tr_words, tr_onsets, tokenizer, max_tokens, chat_template)
        pad = max_tokens - len(ids)
        if pad < 0 or len(onsets) > ONSET_SLOTS:
            raise ValueError(f"TR {i}: {len(ids)} prompt tokens / {len(onsets)} dialogue tokens exceed {max_tokens} / {ONSET_SLOTS}")
        run_tokens.append(np.pad(np.asarray(ids, dtype=np.int64), (0, pad)))
        run_times.append(np.pad(np.asarray(onsets, dtype=np.float64), (0, ONSET_SLOTS - len(onsets))))
        params.append(np.array([pad, inst_len, len(onsets)], dtype=np.int64))
    return np.array(run_tokens), np.array(run_times), np.array(params)


# --------------------------------------------------------------------------------------------------
# video
# --------------------------------------------------------------------------------------------------
def frame_sample(duration: int, num_frames: int) -> np.ndarray:
    """VideoLLaMA2 ``mm_utils.frame_sample(mode='uniform')`` (UNPINNED restatement): centres of ``num_frames`` equal
    segments of [0, duration-1], rounded to frame indices."""
    seg = float(duration - 1) / num_frames
    return np.round(np.array([(seg * i + seg * (i + 1)) / 2 for i in range(num_frames)]) + 1e-6).astype(int)


def expand2square(frame: np.ndarray, background: Sequence[int]) -> np.ndarray:
    """LLaVA / VideoLLaMA2 ``expand2square`` on an [H, W, 3] uint8 array: centre the frame on a square canvas of the
    background colour (the processor's mean colour, extractfeatures.py:345)."""
    h, w = frame.shape[:2]
    if h == w:
        return frame
    side = max(h, w)
    out = np.empty((side, side, 3), dtype=frame.dtype)
    out[...] = np.asarray(background, dtype=frame.dtype)
    top, left = (side - h) // 2, (side - w) // 2
    out[top:top + h, left:left + w] = frame
    return out


def clip_preprocess(frames: Iterable[np.ndarray], size: int = 336) -> np.ndarray:
    """``CLIPImageProcessor.preprocess`` of openai/clip-vit-large-patch14-336 on square uint8 frames: bicubic resize to
    ``size``, (centre crop: a no-op on squares), rescale 1/255, normalise -> float32 [n, 3, size, size]."""
    from PIL import Image
    mean, std = np.asarray(CLIP_MEAN, np.float32), np.asarray(CLIP_STD, np.float32)
    out = []
    for f in frames:
        img = Image.fromarray(f).convert("RGB").resize((size, size), resample=Image.BICUBIC)
        x = np.asarray(img, dtype=np.float32) / 255.0
        out.append(((x - mean) / std).transpose(2, 0, 1))
    return np.stack(out).astype(np.float32)


def window_frame_indices(end_time: float, win_dur: int, fps: float, num_frames_of_video: int, tr: float = 1.49,
                         frames_per_tr: int = 4) -> List[int]:
    """extractfeatures.py:320-335: frames of the window [end_time - win_dur*tr, end_time] sampled uniformly, 4 per TR of
    window actually available (windows at the episode's onset are shorter)."""
    start = max(0.0, end_time - tr * win_dur)
    f_start = max(int(start * fps) - 1, 0)
    f_end = min(int(end_time * fps) - 1, num_frames_of_video - 1)
    span = list(range(f_start, f_end + 1))
    n = round((end_time - start) / tr) * frames_per_tr
    return [span[i] for i in frame_sample(len(span), n)]


def extract_video_chunk(frames, end_time: float, win_dur: int, fps: float, num_frames_of_video: int, tr: float = 1.49,
                        frames_per_tr: int = 4, size: int = 336) -> np.ndarray:
    """extractfeatures.py:320-349 -> float32 [win_dur*frames_per_tr, 3, size, size]; short windows are completed with
    black frames (appended at the END, as the reference does) before padding to square and normalising."""
    idx = window_frame_indices(end_time, win_dur, fps, num_frames_of_video, tr, frames_per_tr)
    batch = frames.get_batch(idx) if hasattr(frames, "get_batch") else np.asarray(frames)[idx]
    batch = batch.asnumpy() if hasattr(batch, "asnumpy") else np.asarray(batch)
    data = [np.ascontiguousarray(f) for f in batch]
    while len(data) < win_dur * frames_per_tr:
        data.append(np.zeros_like(data[-1]))
    bg = tuple(int(x * 255) for x in CLIP_MEAN)
    return clip_preprocess([expand2square(f, bg) for f in data], size)


def tr_end_times(num_frames_of_video: int, fps: float, tr: float = 1.49) -> List[float]:
    """extractfeatures.py:303-317: END (s) of the frame window of every TR of the episode."""
    duration = num_frames_of_video / fps
    return (np.array(range(1, math.ceil(duration / tr))) * tr).tolist()


def extract_episode(text_per_tr, words_per_tr, onsets_per_tr, seg_times, tokenizer, frames, fps: float,
                    num_frames_of_video: int, tr: float = 1.49, window_duration: int = 3, frames_per_tr: int = 4,
                    model_max_length: int = 2048, size: int = 336, chat_template: Optional[Callable] = None) -> dict:
    """One episode -> the four arrays of the wire format (EPISODE_KEYS of episodes.py)."""
    max_tokens = get_max_token(model_max_length, window_duration, frames_per_tr)
    tf, to, mp = episode_text_features(text_per_tr, words_per_tr, onsets_per_tr, seg_times, tokenizer, tr, window_duration,
                                       max_tokens, chat_template)
    video = np.stack([extract_video_chunk(frames, e, window_duration, fps, num_frames_of_video, tr, frames_per_tr, size)
                      for e in tr_end_times(num_frames_of_video, fps, tr)])
    return {"video_features": video, "transcript_features": tf, "transcript_onsets": to, "masking_params": mp}


def write_episode(path: str, ep_num: str, arrays: Mapping[str, np.ndarray]) -> str:
    """Append one episode to the feature file: an HDF5 group with gzip-4 datasets (extractfeatures.py:457-508) when the path
    is not ``.npz`` (needs h5py), else a flat ``.npz`` with keys ``<episode>/<dataset>`` - both are what ``episodes.open_groups``
    and ``make_lazy_loading_dsets`` read."""
    if str(path).endswith(".npz"):
        import os
        old = dict(np.load(path)) if os.path.exists(path) else {}
        old.update({f"{ep_num}/{k}": np.asarray(v) for k, v in arrays.items()})
        np.savez(path, **old)
        return path
    import h5py  # type: ignore
    with h5py.File(path, "a") as f:
        grp = f.require_group(ep_num)
        for k, v in arrays.items():
            if k in grp:
                del grp[k]
            grp.create_dataset(k, data=np.asarray(v), compression="gzip", compression_opts=4)
    return path
