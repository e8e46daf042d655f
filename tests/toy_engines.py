"""Small deterministic stand-ins for the three engines the reference's preprocessing scripts load by model NAME or from
absent packages (the VideoLLaMA2 tokenizer, decord's VideoReader, nilearn's Glover regressor).  TEST INFRASTRUCTURE.

The same objects are handed (a) to the reference's own functions by ``oracle/gen_ref_fixtures_py39.py`` when it generates
``tests/golden/ref_pipeline/`` and (b) to this repository's ``extract.py`` / ``episodes.py`` by the tests that compare against
those files, so what the comparison pins is everything AROUND the engines: prompt assembly, truncation, onsets, window
arithmetic, padding, alignment offsets, dtypes and the HDF5 layout.  stdlib + numpy only (it also runs under the build
container's python3.9, which has h5py / pandas but no torch).
"""
import math
import re

import numpy as np


class ToyTokenizer:
    """Word-piece tokenizer with the properties of the real (Llama sentencepiece) one that the id-row layout depends on:
    a newline is the two tokens ['▁', '<0x0A>'], ' [/INST]' is the four tokens ['▁[', '/', 'INST', ']'], and the
    tokenisation of space-joined words is the concatenation of the words' tokenisations (long words split in pieces).
    Ids are assigned from a fixed hash of the piece, so two instances agree without sharing state."""
    bos_token_id = 1
    unk_token = "<unk>"
    pad_token = None

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

    def token_id(self, t):
        if t not in self.ids:
            h = 2166136261
            for b in t.encode("utf-8"):
                h = ((h ^ b) * 16777619) & 0xFFFFFFFF
            self.ids[t] = 3 + h % 31997
        return self.ids[t]

    def __call__(self, text, **kw):
        ids = [self.bos_token_id] + [self.token_id(t) for t in self.tokenize(text)]
        return type("Enc", (), {"input_ids": ids})()

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False):
        """Llama-2 / Mistral-instruct layout of a (system, user) pair (the real string lives in the checkpoint's
        tokenizer_config.json, absent offline)."""
        system = "".join(m["content"] for m in messages if m["role"] == "system")
        user = "".join(m["content"] for m in messages if m["role"] == "user")
        return f"[INST] {system}\n\n{user} [/INST]"


class ToyBatch:
    def __init__(self, a):
        self.a = a

    def asnumpy(self):
        return self.a


class ToyVideoReader:
    """decord.VideoReader surface the reference uses (extractfeatures.py:303-317,336): len(), get_avg_fps(),
    get_batch(indices).asnumpy() -> uint8 [n, H, W, 3].  Frame f is a seeded pattern of its index."""

    def __init__(self, path=None, ctx=None, num_threads=1, n=190, fps=10.0, h=24, w=32):
        spec = None if path is None else TOY_VIDEOS.get(str(path).split("/")[-1])
        if spec is not None:
            n, fps, h, w = spec
        self.n, self.fps, self.h, self.w = n, fps, h, w

    def __len__(self):
        return self.n

    def get_avg_fps(self):
        return self.fps

    def frame(self, f):
        yy, xx = np.mgrid[0:self.h, 0:self.w]
        out = np.empty((self.h, self.w, 3), np.uint8)
        out[..., 0] = (f * 7 + yy * 3 + xx) % 251
        out[..., 1] = (f * 13 + yy + xx * 5) % 241
        out[..., 2] = (f + yy * xx) % 239
        return out

    def get_batch(self, idx):
        return ToyBatch(np.stack([self.frame(int(f)) for f in idx]))


TOY_VIDEOS = {}          # file name -> (n_frames, fps, h, w); filled by whoever creates the fake .mkv files


def toy_cpu(i):
    return None


def toy_hrf(t):
    """A smooth, sign-changing function of the time difference (s) in place of nilearn's Glover regressor."""
    t = float(t)
    return math.exp(-t / 5.0) * math.sin(t / 2.0) * 0.3 + 0.01 * t


def toy_frame_sample(duration, mode="uniform", num_frames=None, fps=None):
    """VideoLLaMA2 mm_utils.frame_sample(mode='uniform') as published: centres of num_frames equal segments of
    [0, duration - 1], rounded (+1e-6 against banker's rounding)."""
    seg = float(duration - 1) / num_frames
    ids = []
    for i in range(num_frames):
        ids.append(np.round((seg * i + seg * (i + 1)) / 2 + 1e-6))
    return np.array(ids).astype(int)


def toy_expand2square(pil_img, background_color):
    """LLaVA / VideoLLaMA2 mm_utils.expand2square as published (PIL images)."""
    from PIL import Image
    width, height = pil_img.size
    if width == height:
        return pil_img
    side = max(width, height)
    result = Image.new(pil_img.mode, (side, side), background_color)
    if width > height:
        result.paste(pil_img, (0, (width - height) // 2))
    else:
        result.paste(pil_img, ((height - width) // 2, 0))
    return result


class ToyProcessor:
    """CLIPImageProcessor surface the reference uses (image_mean, preprocess(images)['pixel_values']) with the
    openai/clip-vit-large-patch14-336 constants at a small target size: bicubic resize, rescale 1/255, normalise."""
    image_mean = [0.48145466, 0.4578275, 0.40821073]
    image_std = [0.26862954, 0.26130258, 0.27577711]

    def __init__(self, size=32):
        self.size = size

    def preprocess(self, images, **kw):
        from PIL import Image
        mean = np.asarray(self.image_mean, np.float32)
        std = np.asarray(self.image_std, np.float32)
        out = []
        for img in images:
            img = img.convert("RGB")
            if img.size != (self.size, self.size):
                img = img.resize((self.size, self.size), resample=Image.BICUBIC)
            x = (np.asarray(img).astype(np.float64) * (1 / 255)).astype(np.float32)
            out.append(((x - mean) / std).transpose(2, 0, 1))
        return {"pixel_values": out}


def synthetic_transcript(n, seed, silent=()):
    """(text_per_tr, words_per_tr, onsets_per_tr) of an n-TR episode; `silent` TRs have no text (NaN in the TSV)."""
    rng = np.random.RandomState(seed)
    words = ("well okay so then maybe could you pass the extraordinarily long coffeehouse thing please Rachel Monica "
             "it's we're don't Chandler's apartment, really? no! yes.").split()
    text, wl, ol = [], [], []
    for i in range(n):
        if i in silent:
            text.append(None)
            wl.append([])
            ol.append([])
            continue
        k = int(rng.randint(1, 6))
        w = [words[int(rng.randint(len(words)))] for _ in range(k)]
        text.append(" ".join(w) + " ")
        wl.append(w)
        ol.append([round(1.49 * i + 0.25 * q, 2) for q in range(k)])
    return text, wl, ol
