"""HF checkpoint directory (config.json + model.safetensors) -> Fish through the C++ loader (kf_safetensors.cpp): the model it builds
must be the model synth.build_from_raw builds from the same tensors -- identical logits and ids (same device quantiser, same kernels)."""
import json

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import Qwen3
from oracle import oracle as O

pytestmark = pytest.mark.gpu
HF = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj", "gate": "mlp.gate_proj", "up": "mlp.up_proj",
      "down": "mlp.down_proj"}
HFN = {"norm_in": "input_layernorm", "norm_post": "post_attention_layernorm", "qn": "self_attn.q_norm", "kn": "self_attn.k_norm"}


def _bf16(a_u16):
    return torch.from_numpy(np.ascontiguousarray(a_u16).view(np.int16)).view(torch.bfloat16)


def write_hf(path, cfg, raw, dtype=torch.bfloat16, sharded=False):
    t = {"model.embed_tokens.weight": _bf16(raw["embed"]).to(dtype), "model.norm.weight": _bf16(raw["final_norm"]).to(dtype)}
    if not cfg.get("tied", True):
        t["lm_head.weight"] = _bf16(raw["head"]).to(dtype)
    for i, lw in enumerate(raw["layers"]):
        for s, n in HF.items():
            t["model.layers.%d.%s.weight" % (i, n)] = _bf16(lw[s]).to(dtype)
        for s, n in HFN.items():
            t["model.layers.%d.%s.weight" % (i, n)] = _bf16(lw[s]).to(dtype)
    card = {"architectures": ["Qwen3ForCausalLM"], "hidden_size": cfg["dim"], "num_hidden_layers": cfg["n_layer"], "num_attention_heads": cfg["n_head"],
            "num_key_value_heads": cfg["n_kv"], "head_dim": cfg["head_dim"], "intermediate_size": cfg["ffn"], "vocab_size": cfg["vocab"], "rms_norm_eps": 1e-6,
            "rope_theta": cfg["theta"], "tie_word_embeddings": bool(cfg.get("tied", True)), "max_position_embeddings": 40960, "torch_dtype": "bfloat16"}
    (path / "config.json").write_text(json.dumps(card, indent=1))
    if not sharded:
        save_file(t, str(path / "model.safetensors"), metadata={"format": "pt"})
    else:
        names = sorted(t)
        half = len(names) // 2
        parts = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
        wm = {}
        for fn, ns in parts.items():
            save_file({n: t[n] for n in ns}, str(path / fn))
            wm.update({n: fn for n in ns})
        (path / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))


@pytest.mark.parametrize("layer_type,head_type,sharded", [(L.Q4, L.BF16, False), (L.BF16, L.BF16, True), (L.F8E5M2, L.BF16, False), (L.BOOL1, L.Q4, False),
                                                            (L.NF4, L.BF16, False), (L.NF4, L.NF4, True)])
def test_hf_directory_builds_the_same_model(tmp_path, layer_type, head_type, sharded):
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 4321, w_std=0.1)
    write_hf(tmp_path, cfg, raw, sharded=sharded)
    a = Qwen3.from_hf(tmp_path, layer_type, head_type, max_seq=cfg["max_seq"])
    assert (a.cfg["dim"], a.cfg["n_layer"], a.cfg["n_kv"], a.cfg["head_dim"], a.cfg["vocab"], a.cfg["tied"]) == (cfg["dim"], cfg["n_layer"], cfg["n_kv"],
                                                                                                              cfg["head_dim"], cfg["vocab"], True)
    b = synth.build_from_raw(cfg, raw, layer_type, head_type)
    prompt = prompt_ids(cfg, 10)
    tok = int(prompt[0])
    for pos in range(12):
        na, la = a.forward(tok, pos)
        nb, lb = b.forward(tok, pos)
        assert na == nb and np.array_equal(la, lb), "step %d" % pos
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else na
    assert a.generate(prompt, 16) == b.generate(prompt, 16)
    a.close()
    b.close()


def test_hf_fp32_checkpoint_and_oracle_ids(tmp_path):
    """an F32 checkpoint is rounded to bf16 on load (the values are bf16-exact here), and the loaded model reproduces the oracle's ids"""
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 99, w_std=0.1)
    write_hf(tmp_path, cfg, raw, dtype=torch.float32)
    m = Qwen3.from_hf(tmp_path, L.Q4, L.BF16, max_seq=cfg["max_seq"])
    om = oracle_model(cfg, raw, L.Q4, L.BF16)
    prompt = prompt_ids(cfg, 12)
    assert m.generate(prompt, 24) == om.generate(prompt.tolist(), 24)
    m.close()


def test_hf_load_errors(tmp_path):
    with pytest.raises(L.KFError, match="config.json"):
        Qwen3.from_hf(tmp_path)
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 1, w_std=0.1)
    write_hf(tmp_path, cfg, raw)
    card = json.loads((tmp_path / "config.json").read_text())
    card["intermediate_size"] = cfg["ffn"] * 2
    (tmp_path / "config.json").write_text(json.dumps(card))
    with pytest.raises(L.KFError, match="unexpected shape"):
        Qwen3.from_hf(tmp_path, max_seq=64)


def test_hf_autoawq_checkpoint(tmp_path):
    """AutoAWQ GEMM-format checkpoint (qweight / qzeros / scales per linear, dense embeddings and norms): the loader takes the triples as
    they are, the model runs the per-kernel path (fuse_level 0: the AWQ layout has its own mat-vec) and follows the oracle's AWQ model."""
    cfg = dict(synth.CONFIGS["small"], n_layer=2, vocab=1024, max_seq=64)     # every in-dimension a multiple of 128 (AWQ group)
    raw = synth.raw_weights_numpy(cfg, 7, w_std=0.05)
    rng = np.random.default_rng(3)
    t = {"model.embed_tokens.weight": _bf16(raw["embed"]), "model.norm.weight": _bf16(raw["final_norm"])}
    ow = {"embed": O.quantize(raw["embed"], cfg["vocab"], cfg["dim"], L.BF16), "final_norm": raw["final_norm"], "layers": []}
    ow["head"] = ow["embed"]
    for i, lw in enumerate(raw["layers"]):
        d = {}
        for s, name in HF.items():
            n_out, n_in = synth.SHAPES[s](cfg)
            q = rng.integers(0, 16, size=(n_in, n_out))
            z = rng.integers(0, 16, size=(n_in // 128, n_out))
            sc = rng.uniform(0.002, 0.01, size=(n_in // 128, n_out)).astype(np.float16)
            aw = O.AWQWeight(n_out, n_in, O.awq_pack(q), O.awq_pack(z), sc)
            d[s] = aw
            p = "model.layers.%d.%s" % (i, name)
            t[p + ".qweight"] = torch.from_numpy(aw.data.view(np.int32).reshape(n_in, n_out // 8).copy())
            t[p + ".qzeros"] = torch.from_numpy(aw.qzeros.view(np.int32).reshape(n_in // 128, n_out // 8).copy())
            t[p + ".scales"] = torch.from_numpy(sc.copy())
        for s, name in HFN.items():
            t["model.layers.%d.%s.weight" % (i, name)] = _bf16(lw[s])
            d[s] = lw[s]
        ow["layers"].append(d)
    card = {"hidden_size": cfg["dim"], "num_hidden_layers": cfg["n_layer"], "num_attention_heads": cfg["n_head"], "num_key_value_heads": cfg["n_kv"],
            "head_dim": cfg["head_dim"], "intermediate_size": cfg["ffn"], "vocab_size": cfg["vocab"], "rms_norm_eps": 1e-6, "rope_theta": cfg["theta"],
            "tie_word_embeddings": True, "quantization_config": {"quant_method": "awq", "bits": 4, "group_size": 128, "version": "gemm"}}
    (tmp_path / "config.json").write_text(json.dumps(card))
    save_file(t, str(tmp_path / "model.safetensors"))
    m = Qwen3.from_hf(tmp_path, L.Q4, L.BF16, max_seq=cfg["max_seq"])
    assert m.fuse_level == 0
    om = O.Qwen3Oracle(cfg, ow)
    prompt = prompt_ids(cfg, 8)
    tok = int(prompt[0])
    for pos in range(10):
        g_next, g_logits = m.forward(tok, pos)
        o_next, o_logits, _ = om.decode(tok, pos)
        gl, ol = O.bf16_to_f32(g_logits), O.bf16_to_f32(o_logits)
        assert np.abs(gl - ol).max() <= 2.0 ** -6 * np.abs(ol).max(), "step %d" % pos
        assert g_next == o_next
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else o_next
    ids = m.generate(prompt, 8)                    # eager per-kernel steps driven by the device state
    assert ids == om.generate(prompt.tolist(), 8)
    m.set_prefill_mode(1)
    assert m.generate(prompt, 8) == ids            # batched prefill falls back to the AWQ mat-vec per token row: same ids
    m.close()


def test_unsupported_shapes_and_families_fail_early_with_a_reason(tmp_path):
    """a group of 5 query heads per kv head (Qwen3-14B) or another model family is refused before any tensor is touched, with the reason"""
    with pytest.raises(L.KFError, match="served: 1, 2, 4, 8"):
        Qwen3(dict(synth.CONFIGS["tiny"], n_head=10, n_kv=2))
    with pytest.raises(L.KFError, match="head_dim 96"):
        Qwen3(dict(synth.CONFIGS["tiny"], head_dim=96))
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 1, w_std=0.1)
    write_hf(tmp_path, cfg, raw)
    card = json.loads((tmp_path / "config.json").read_text())
    (tmp_path / "config.json").write_text(json.dumps(dict(card, model_type="qwen2")))
    with pytest.raises(L.KFError, match="model_type 'qwen2'"):
        Qwen3.from_hf(tmp_path, max_seq=64)
    (tmp_path / "config.json").write_text(json.dumps(dict(card, num_attention_heads=10)))
    with pytest.raises(L.KFError, match="served: 1, 2, 4, 8"):
        Qwen3.from_hf(tmp_path, max_seq=64)
    # a checkpoint with projection biases (Qwen2 style) is refused rather than decoded without them
    (tmp_path / "config.json").write_text(json.dumps(dict(card, model_type="qwen3")))
    t = {"model.layers.0.self_attn.q_proj.bias": torch.zeros(cfg["n_head"] * cfg["head_dim"], dtype=torch.bfloat16)}
    save_file(t, str(tmp_path / "extra.safetensors"))
    import safetensors.torch as stt
    allt = stt.load_file(str(tmp_path / "model.safetensors"))
    allt.update(t)
    save_file(allt, str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
    with pytest.raises(L.KFError, match="projection biases"):
        Qwen3.from_hf(tmp_path, max_seq=64)
