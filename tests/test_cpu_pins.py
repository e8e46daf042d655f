"""CPU tier: the oracle's pins as tests, the committed goldens against a fresh generation, and independent
in-container cross-checks for the rows no importable reference exists for (connector, splice, LoRA: "parity
unpinned" - the upstream VideoLLaMA2 / timm / peft sources are absent).

* head  == the reference's OWN src/utils.py layers   (needs /root/reference; skipped on the GPU box)
* CLIP  == transformers.CLIPVisionModel, Mistral == transformers.MistralModel   (needs transformers)
* tests/golden/*.npz == oracle/gen_golden.build_golden() today: the oracle and its fixtures cannot drift apart
* LoRA forward / gradients == torch autograd on the explicitly merged weight W + s.B.A (and the dropout form)
* connector stages == torch.nn modules (Conv2d / Conv3d / Linear / LayerNorm) loaded from the same state-dict names
* splice == an index-by-index Python loop
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

have_reference = os.path.isfile("/root/reference/src/utils.py")
try:
    import transformers  # noqa: F401
    have_transformers = True
except Exception:  # pragma: no cover
    have_transformers = False


@pytest.mark.skipif(not have_reference, reason="the reference tree only exists in the build container")
def test_pin_head_against_the_reference_layers():
    import gen_golden
    gen_golden.pin_head()


@pytest.mark.skipif(not have_transformers, reason="transformers not importable")
def test_pin_clip_tower_against_transformers():
    import gen_golden
    import vlb_oracle as O
    gen_golden.pin_clip(O.geometry_mini())


@pytest.mark.skipif(not have_transformers, reason="transformers not importable")
def test_pin_mistral_decoder_against_transformers():
    import gen_golden
    import vlb_oracle as O
    gen_golden.pin_mistral(O.geometry_mini())


@pytest.mark.skipif(not (have_reference and have_transformers), reason="the reference tree only exists in the build container")
def test_committed_reference_fixtures_equal_a_fresh_run_of_the_reference():
    """tests/golden/ref_{weight_mask.npz,linear_names.json,datamodule.json} == what the reference's own make_weight_mask /
    find_all_linear_names / datamodule produce today (oracle/gen_ref_fixtures.generate(): the python3.10 side)."""
    import json
    import gen_ref_fixtures as G
    fresh = G.generate()
    gold = np.load(os.path.join(GOLD, "ref_weight_mask.npz"))
    assert set(gold.files) == set(fresh["weight_mask"])
    for k in gold.files:
        assert np.array_equal(gold[k], np.asarray(fresh["weight_mask"][k])), k
    for name in ("linear_names", "datamodule"):
        with open(os.path.join(GOLD, f"ref_{name}.json")) as f:
            assert json.load(f) == json.loads(json.dumps(fresh[name])), name


@pytest.mark.skipif(not (have_reference and os.path.exists("/opt/conda/bin/python3.9")), reason="build container only")
def test_committed_reference_pipeline_files_equal_a_fresh_run_of_the_reference(tmp_path):
    """tests/golden/ref_pipeline/ == what the reference's extractfeatures / lazyloading scripts write today
    (oracle/gen_ref_fixtures_py39.py re-run into a scratch directory; datasets compared one by one)."""
    import json
    import subprocess
    from phantom_vlb_amd import h5lite
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    env["VLB_REF_PIPELINE_OUT"] = str(tmp_path / "ref_pipeline")
    subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", os.path.join(ROOT, "oracle", "gen_ref_fixtures_py39.py")],
                   check=True, env=env, capture_output=True)
    gold_dir = os.path.join(GOLD, "ref_pipeline")

    def walk(g, prefix=""):
        for k in g.keys():
            item = g[k]
            if hasattr(item, "keys"):
                yield from walk(item, f"{prefix}{k}/")
            else:
                yield f"{prefix}{k}", np.array(item)

    files = sorted(os.path.relpath(os.path.join(dp, fn), gold_dir) for dp, _, fns in os.walk(gold_dir) for fn in fns)
    fresh_dir = env["VLB_REF_PIPELINE_OUT"]
    assert files == sorted(os.path.relpath(os.path.join(dp, fn), fresh_dir) for dp, _, fns in os.walk(fresh_dir) for fn in fns)
    for rel in files:
        if rel.endswith(".json"):
            with open(os.path.join(gold_dir, rel)) as a, open(os.path.join(fresh_dir, rel)) as b:
                assert json.load(a) == json.load(b)
            continue
        a, b = dict(walk(h5lite.File(os.path.join(gold_dir, rel)))), dict(walk(h5lite.File(os.path.join(fresh_dir, rel))))
        assert a.keys() == b.keys(), rel
        for k in a:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (rel, k)


@pytest.mark.skipif(not have_transformers, reason="transformers not importable")
def test_clip_preprocess_equals_the_hf_clip_image_processor():
    """extract.clip_preprocess (the processor the reference gets from the CLIP tower, extractfeatures.py:148-177,345-347)
    against transformers' CLIPImageProcessor configured as openai/clip-vit-large-patch14-336's preprocessor_config.json
    (shortest edge 336 bicubic, centre crop 336, rescale 1/255, OpenAI mean / std) on square frames."""
    import warnings
    from PIL import Image
    from phantom_vlb_amd import extract as X
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from transformers import CLIPImageProcessor
        proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_resize=True,
                                  do_center_crop=True, do_rescale=True, do_normalize=True, do_convert_rgb=True, resample=3,
                                  image_mean=list(X.CLIP_MEAN), image_std=list(X.CLIP_STD))
    rng = np.random.RandomState(0)
    frames = [rng.randint(0, 256, (480, 480, 3), dtype=np.uint8), rng.randint(0, 256, (336, 336, 3), dtype=np.uint8),
              rng.randint(0, 256, (200, 200, 3), dtype=np.uint8)]
    want = np.stack([np.asarray(proc.preprocess([Image.fromarray(f)])["pixel_values"][0]) for f in frames])
    got = X.clip_preprocess(frames, 336)
    assert got.shape == want.shape == (3, 3, 336, 336) and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-6


@pytest.mark.parametrize("tag", ["frozen", "lora"])
def test_committed_goldens_equal_a_fresh_generation(tag):
    import gen_golden
    fresh = gen_golden.build_golden(tag)
    gold = np.load(os.path.join(GOLD, f"mini_{tag}.npz"))
    assert set(gold.files) == set(fresh)
    for k in gold.files:
        a, b = gold[k], np.asarray(fresh[k])
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype.kind in "iubSU":
            assert np.array_equal(a, b), k
        else:
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(a).max()))), k


def test_committed_mask_vectors_equal_a_fresh_generation():
    import gen_golden
    fresh = gen_golden.build_mask_kat()
    gold = np.load(os.path.join(GOLD, "weight_mask_kat.npz"))
    assert set(gold.files) == set(fresh)
    for k in gold.files:
        assert np.array_equal(gold[k], fresh[k]), k


# ------------------------------------------------------------------ LoRA (peft absent): merged-weight autograd
@pytest.mark.parametrize("r,drop", [(16, False), (8, False), (16, True)])
def test_lora_linear_equals_autograd_on_the_merged_weight(r, drop):
    """y = x W^T + (alpha/r) B A x  ==  x (W + (alpha/r) B A)^T, values and every gradient.  With dropout the adapter
    sees x*keep/(1-p) while the base path sees x: checked against the two-term expression differentiated by hand."""
    import vlb_oracle as O
    torch.manual_seed(0)
    din, dout, M, alpha, p = 48, 40, 33, 32, 0.25
    g = O.Geometry(lora_r=r, lora_alpha=alpha)
    s = alpha / r
    W, A, B = torch.randn(dout, din) * 0.2, torch.randn(r, din) * 0.3, torch.randn(dout, r) * 0.3
    x, dy = torch.randn(M, din), torch.randn(M, dout)
    keep = (torch.rand(M, din) >= p).float() / (1 - p) if drop else None
    P = {"l.weight": W, "l.lora_A.weight": A.clone().requires_grad_(True), "l.lora_B.weight": B.clone().requires_grad_(True)}
    xo = x.clone().requires_grad_(True)
    y = O._lin(P, "l", xo, g, lora_drop=None if keep is None else {"l": keep})
    (y * dy).sum().backward()
    if not drop:
        Am, Bm, xm = A.clone().requires_grad_(True), B.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ym = F.linear(xm, W + s * Bm @ Am)
        (ym * dy).sum().backward()
        assert torch.allclose(y, ym, atol=1e-5)
        assert torch.allclose(xo.grad, xm.grad, atol=1e-5)
        assert torch.allclose(P["l.lora_A.weight"].grad, Am.grad, atol=1e-4)
        assert torch.allclose(P["l.lora_B.weight"].grad, Bm.grad, atol=1e-4)
    else:
        xd = x * keep
        assert torch.allclose(y, x @ W.t() + s * (xd @ A.t()) @ B.t(), atol=1e-5)
        u = s * dy @ B                                  # [M, r]
        assert torch.allclose(P["l.lora_B.weight"].grad, s * dy.t() @ (xd @ A.t()), atol=1e-4)
        assert torch.allclose(P["l.lora_A.weight"].grad, u.t() @ xd, atol=1e-4)
        assert torch.allclose(xo.grad, dy @ W + keep * (u @ A), atol=1e-5)


# ------------------------------------------------------------------ connector (timm / VideoLLaMA2 absent): nn modules
class _LayerNorm2d(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class _ConvNormAct(nn.Module):
    def __init__(self, cin, cout, k, groups=1, act=True, eps=1e-6):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, groups=groups, bias=False)
        self.bn = _LayerNorm2d(cout, eps=eps)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class _SE(nn.Module):
    def __init__(self, c, rd):
        super().__init__()
        self.fc1, self.fc2 = nn.Conv2d(c, rd, 1), nn.Conv2d(rd, c, 1)

    def forward(self, x):
        return x * torch.sigmoid(self.fc2(F.silu(self.fc1(x.mean((2, 3), keepdim=True)))))


class _Bottleneck(nn.Module):
    """RegNet bottleneck as the published timm architecture describes it (bottle_ratio 1, group width 1 -> depthwise
    3x3, SE on the bottleneck width with rd = round(in_chs * 0.25), LayerNorm2d norms, SiLU; conv1x1 shortcut on a
    channel change) - built from stock torch.nn modules, attribute names = the state-dict names."""

    def __init__(self, cin, cout, se_ratio, eps):
        super().__init__()
        self.conv1 = _ConvNormAct(cin, cout, 1, eps=eps)
        self.conv2 = _ConvNormAct(cout, cout, 3, groups=cout, eps=eps)
        self.se = _SE(cout, int(round(cin * se_ratio)))
        self.conv3 = _ConvNormAct(cout, cout, 1, act=False, eps=eps)
        self.downsample = _ConvNormAct(cin, cout, 1, act=False, eps=eps) if cin != cout else None

    def forward(self, x):
        sc = x if self.downsample is None else self.downsample(x)
        return F.silu(self.conv3(self.se(self.conv2(self.conv1(x)))) + sc)


class _STC(nn.Module):
    def __init__(self, g):
        super().__init__()
        def stage(cin):
            return nn.ModuleDict({f"b{i + 1}": _Bottleneck(cin if i == 0 else g.dim, g.dim, g.proj_se_ratio, g.proj_eps)
                                  for i in range(g.proj_depth)})
        self.s1, self.s2 = stage(g.vit_dim), stage(g.dim)
        self.sampler = nn.Sequential(nn.Conv3d(g.dim, g.dim, 2, stride=2, padding=1), nn.SiLU())
        self.readout = nn.Sequential(nn.Linear(g.dim, g.dim), nn.GELU(), nn.Linear(g.dim, g.dim))

    def forward(self, feats, g):
        B, T = feats.shape[:2]
        x = feats.reshape(B * T, g.grid, g.grid, g.vit_dim).permute(0, 3, 1, 2)
        for blk in self.s1.values():
            x = blk(x)
        x = self.sampler(x.reshape(B, T, g.dim, g.grid, g.grid).transpose(1, 2))           # b d t h w
        nt, nh = x.shape[2], x.shape[3]
        x = x.transpose(1, 2).reshape(B * nt, g.dim, nh, nh)
        for blk in self.s2.values():
            x = blk(x)
        x = x.reshape(B, nt, g.dim, nh * nh).permute(0, 1, 3, 2).reshape(B, nt * nh * nh, g.dim)
        return self.readout(x)


def test_stc_connector_equals_torch_nn_modules_with_the_same_state_dict():
    import vlb_oracle as O
    g = O.geometry_mini()
    p = O.init_params(g, seed=11)
    pre = "model.mm_projector."
    sd = {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}
    net = _STC(g)
    missing, unexpected = net.load_state_dict(sd, strict=True)          # every oracle tensor has a home, and vice versa
    assert not missing and not unexpected
    torch.manual_seed(3)
    feats = torch.randn(2, g.num_frames, g.grid * g.grid, g.vit_dim)
    with torch.no_grad():
        ref = net(feats, g)
        got = O.stc_connector(p, feats, g)
    assert ref.shape == got.shape == (2, g.vis_tokens, g.dim)
    assert torch.allclose(ref, got, atol=1e-5), float((ref - got).abs().max())


def test_splice_equals_an_index_by_index_loop():
    import vlb_oracle as O
    g = O.geometry_mini()
    torch.manual_seed(5)
    batch = O.synthetic_batch(g, 3, seed=9)
    ids = batch["language"].long()
    E = torch.randn(g.vocab, 8)
    vid = torch.randn(3, g.vis_tokens, 8)
    emb, mask = O.splice_multimodal(E, ids, vid)
    assert emb.shape == (3, g.max_len, 8) and mask.shape == (3, g.max_len)
    for b in range(3):
        out_row, out_mask = [], []
        for t in ids[b].tolist():
            if t == O.VIDEO_TOKEN_ID:
                out_row += [vid[b, j] for j in range(g.vis_tokens)]
                out_mask += [True] * g.vis_tokens
            else:
                out_row.append(E[t])
                out_mask.append(t != 0)
        assert torch.equal(emb[b], torch.stack(out_row))
        assert mask[b].tolist() == out_mask
