"""Pin the oracle and emit tests/golden/*.  TEST INFRASTRUCTURE - runs in the build container only.

  python oracle/gen_golden.py            # validates, then (re)writes tests/golden/

What it pins (and how):
  1. brain head pooling + ridge   vs the reference's OWN src/utils.py (HRFConvolveLayer,
     RidgeRegressionLayer), loaded by file path from /root/reference with three stub modules
     for its unused imports (lightning Callback, nilearn compute_regressor, torchmetrics).
  2. CLIP tower                   vs installed transformers CLIPVisionModel (random init, mini cfg).
  3. Mistral decoder              vs installed transformers MistralModel (inputs_embeds + padding
     mask, eager attention, hidden_states[-1]).
  4. connector / splice / LoRA / mask layout have no importable reference: UNPINNED
     (hand-checked known-answer vectors only; tests/test_cpu_pins.py adds independent in-container
     cross-checks: LoRA vs autograd on the merged weight, connector vs torch.nn modules, splice vs a loop).
The pins run as tests (tests/test_cpu_pins.py) whenever /root/reference and transformers are present, and the
same file checks that the committed fixtures equal a fresh generation - oracle and goldens cannot drift apart.
The reference itself never travels: only the numbers written here do.
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import vlb_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


def load_reference_head():
    for name, attrs in {
        "lightning": {}, "lightning.pytorch": {}, "lightning.pytorch.callbacks": {"Callback": object},
        "nilearn": {}, "nilearn.glm": {}, "nilearn.glm.first_level": {"compute_regressor": None},
        "torchmetrics": {"PearsonCorrCoef": None},
    }.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("_ref_utils", os.path.join(REF, "src", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pin_head():
    ref = load_reference_head()
    torch.manual_seed(0)
    B, S, E, V = 4, 64, 32, 128
    emb, w = torch.randn(B, S, E), torch.rand(B, S)
    pooled_ref = ref.HRFConvolveLayer()(emb, w)
    pooled = O.hrf_convolve(emb, w)
    assert torch.equal(pooled_ref, pooled)
    ridge = ref.RidgeRegressionLayer(E, V, l2_lambda=1e-3)
    xr = torch.randn(B, E)                 # the two exported layers on their own (src/__init__.py:3-13), as the oracle restates them
    o_ref, l_ref = ridge(xr)
    o_mine, l_mine = O.ridge_regression(ridge.linear.weight.detach(), ridge.linear.bias.detach(), xr, 1e-3)
    assert torch.allclose(o_mine, o_ref, atol=1e-6) and torch.allclose(l_mine, l_ref, rtol=1e-6)
    assert torch.equal(O.ridge_regression(ridge.linear.weight.detach(), ridge.linear.bias.detach(), xr, 1e-3, False), ridge(xr, False))
    g = O.Geometry(dim=E, num_target=V, l2_lambda=1e-3)
    p = {"layer_norm1.weight": torch.ones(E), "layer_norm1.bias": torch.zeros(E),
         "layer_norm2.weight": torch.ones(E), "layer_norm2.bias": torch.zeros(E),
         "ridge_layer.linear.weight": ridge.linear.weight.detach(), "ridge_layer.linear.bias": ridge.linear.bias.detach()}
    # full head composed from the reference's layers exactly as VLBLitModule.forward does (:245-254)
    ln1, ln2 = torch.nn.LayerNorm(E), torch.nn.LayerNorm(E)
    hrf = ref.HRFConvolveLayer()
    out_ref, l2_ref = ridge(ln2(hrf(ln1(emb), w)))
    out, l2, _ = O.brain_head(p, emb, w, g)
    assert torch.allclose(out, out_ref, atol=1e-6), (out - out_ref).abs().max()
    assert torch.allclose(l2, l2_ref, rtol=1e-6)
    print("pinned: brain head == reference src/utils.py layers")


def pin_clip(g):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(hidden_size=g.vit_dim, num_hidden_layers=g.vit_layers, num_attention_heads=g.vit_heads,
                           intermediate_size=g.vit_ff, image_size=g.image_size, patch_size=g.patch)
    cfg._attn_implementation = "eager"
    torch.manual_seed(1)
    m = CLIPVisionModel(cfg).eval()
    sd = m.state_dict()
    pre = "model.vision_tower.vision_tower.vision_model"
    has_vm = any(k.startswith("vision_model.") for k in sd)
    p = {f"{pre}.{k[len('vision_model.'):] if has_vm else k}": v for k, v in sd.items()}
    x = torch.randn(3, 3, g.image_size, g.image_size)
    with torch.no_grad():
        ref = m(pixel_values=x, output_hidden_states=True).hidden_states[g.vit_select_layer][:, 1:]
        got = O.clip_tower(p, x, g)
    err = (ref - got).abs().max().item()
    assert err < 2e-5, err
    print(f"pinned: clip_tower == transformers CLIPVisionModel hidden_states[{g.vit_select_layer}][:,1:]  (max err {err:.2e})")


def pin_mistral(g):
    from transformers import MistralConfig, MistralModel
    cfg = MistralConfig(hidden_size=g.dim, num_hidden_layers=g.layers, num_attention_heads=g.heads,
                        num_key_value_heads=g.kv_heads, intermediate_size=g.ff, vocab_size=g.vocab,
                        head_dim=g.head_dim, sliding_window=None, rms_norm_eps=g.rms_eps,
                        rope_theta=g.rope_theta, max_position_embeddings=4096)
    cfg._attn_implementation = "eager"
    torch.manual_seed(2)
    m = MistralModel(cfg).eval()
    p = {f"model.{k}": v for k, v in m.state_dict().items()}
    B, S = 3, 96
    x = torch.randn(B, S, g.dim) * 0.5
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[0, 80:] = False
    mask[2, 50:] = False
    with torch.no_grad():
        out = m(inputs_embeds=x, attention_mask=mask.long(), output_hidden_states=True)
        ref = out.hidden_states[-1]
        assert torch.equal(ref, out.last_hidden_state)
        got = O.mistral_decoder(p, x, mask, g)
    # padded query rows are don't-care (weight 0 in the head); compare valid rows
    err = ((ref - got).abs() * mask[..., None]).max().item()
    assert err < 5e-5, err
    print(f"pinned: mistral_decoder == transformers MistralModel hidden_states[-1] on valid rows (max err {err:.2e})")


def mask_known_answers():
    """Hand-checkable vectors for make_weight_mask (src/litmodule/...:178-203)."""
    cases = []
    # tokens_per_frame=2, 2 frames, lang_len=17 -> max_len = 4+17-1 = 20
    for pad_len, inst_len, dialog_len in [(0, 1, 3), (2, 1, 0), (0, 0, 0), (3, 2, 2)]:
        pv = torch.tensor([[pad_len, inst_len, dialog_len]])
        vw = torch.tensor([[0.5, 0.25]], dtype=torch.float64)
        lw = torch.zeros(1, 8, dtype=torch.float64)
        lw[0, :dialog_len] = torch.arange(1, dialog_len + 1, dtype=torch.float64) / 10
        row = O.make_weight_mask(pv, vw, lw, 17, 20, tokens_per_frame=2)[0]
        tail = 4 + 2 + inst_len + dialog_len + 4 + pad_len
        expect = [0.0] * (20 - tail) + [0.5, 0.5, 0.25, 0.25] + [0.0] * (2 + inst_len) + \
                 [(i + 1) / 10 for i in range(dialog_len)] + [0.0] * (4 + pad_len)
        assert torch.allclose(row, torch.tensor(expect)), (row, expect)
        cases.append(dict(padvals=pv.numpy(), vis=vw.numpy(), lang=lw.numpy(), row=row.numpy()))
    print("checked: make_weight_mask known-answer vectors")
    return cases


def weights_digest(p):
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(k.encode())
        h.update(p[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def build_mask_kat():
    cases = mask_known_answers()
    return {f"{i}_{k}": v for i, c in enumerate(cases) for k, v in c.items()}


def build_golden(tag: str) -> dict:
    """The arrays of tests/golden/mini_<tag>.npz ('frozen' | 'lora'), regenerated from seeds (configs[0])."""
    g = O.geometry_mini()
    lora = tag == "lora"
    p = O.round_bf16(O.init_params(g, seed=1234, lora=lora, lora_b_std=0.02 if lora else 0.0))
    batch = O.synthetic_batch(g, batch=4, seed=1234)
    names = O.trainable_names(p, freeze_backbone=not lora, use_lora=lora)
    for n in names:
        p[n].requires_grad_(True)
    stages = {}
    loss, pred = O.training_loss(p, batch, g, stages=stages)
    loss.backward()
    out = dict(
        weights_sha256=np.array(weights_digest({k: v.detach() for k, v in p.items()})),
        language=batch["language"].numpy(), padvals=batch["padvals"].numpy(),
        vis_weights=batch["vis_weights"].numpy(), lang_weights=batch["lang_weights"].numpy(),
        timeseries=batch["timeseries"].numpy(),
        vision_probe=batch["vision"][:, 0, 0, :4, :4].numpy(),
        weight_mask=stages["weight_mask"].numpy(),
        vit_tokens=stages["vit_tokens"].detach().numpy().astype(np.float32),
        video_tokens=stages["video_tokens"].detach().numpy(),
        inputs_embeds_probe=stages["inputs_embeds"].detach()[:, ::7, ::5].numpy(),
        key_mask=stages["key_mask"].numpy(),
        layer0=stages["layer_outputs"][0].detach().numpy(),
        hidden=stages["hidden"].detach().numpy(),
        head_pooled=stages["head_pooled"].detach().numpy(),
        head_ln2=stages["head_ln2"].detach().numpy(),
        pred=pred.detach().numpy(), l2=stages["l2"].detach().numpy(), loss=loss.detach().numpy(),
    )
    for n in names:
        if ".lora_" in n and not (n.startswith("model.layers.0.") or n.startswith(f"model.layers.{g.layers - 1}.")):
            continue
        out["grad::" + n] = p[n].grad.numpy()
    out["grad_global_norm"] = np.array(float(torch.sqrt(sum(p[n].grad.double().pow(2).sum() for n in names))))
    return out


def main():
    torch.set_num_threads(8)
    pin_head()
    g = O.geometry_mini()
    pin_clip(g)
    pin_mistral(g)
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, "weight_mask_kat.npz"), **build_mask_kat())
    # ---- mini config end-to-end golden (configs[0]): frozen backbone and LoRA variants
    for tag in ("frozen", "lora"):
        out = build_golden(tag)
        path = os.path.join(GOLDEN, f"mini_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path}: loss={float(out['loss']):.6f}  ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
