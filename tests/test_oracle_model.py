"""Model semantics of the oracle's Qwen3 decoder, checked two ways:
  * against HF transformers' Qwen3 (third-party; architecture semantics only: rotate-half RoPE, per-head q/k RMSNorm,
    GQA grouping h / (n_head/n_kv), SwiGLU, tied head) at a tiny random shape, fp32 math on the same bf16-valued weights;
  * against the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from this oracle).
"""
import os

import numpy as np
import pytest

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_matches_hf_qwen3_semantics():
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    try:
        from transformers import Qwen3Config, Qwen3ForCausalLM
    except Exception:
        pytest.skip("transformers has no Qwen3")
    cfg = dict(synth.CONFIGS["tiny"])
    raw = synth.raw_weights_numpy(cfg, 4321, w_std=0.05)
    om = oracle_model(cfg, raw, L.BF16, L.BF16)
    kw = dict(hidden_size=cfg["dim"], intermediate_size=cfg["ffn"], num_hidden_layers=cfg["n_layer"], num_attention_heads=cfg["n_head"],
              num_key_value_heads=cfg["n_kv"], head_dim=cfg["head_dim"], vocab_size=cfg["vocab"], rms_norm_eps=1e-6, tie_word_embeddings=True,
              max_position_embeddings=cfg["max_seq"], attention_bias=False, use_sliding_window=False)
    try:
        hc = Qwen3Config(rope_theta=cfg["theta"], **kw)
    except TypeError:
        hc = Qwen3Config(rope_parameters={"rope_type": "default", "rope_theta": cfg["theta"]}, **kw)
    hc._attn_implementation = "eager"
    hf = Qwen3ForCausalLM(hc).to(torch.float32).eval()

    def t(a):
        return torch.from_numpy(O.bf16_to_f32(a).copy())
    sd = {"model.embed_tokens.weight": t(raw["embed"]), "model.norm.weight": t(raw["final_norm"]), "lm_head.weight": t(raw["embed"])}
    names = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj", "gate": "mlp.gate_proj",
             "up": "mlp.up_proj", "down": "mlp.down_proj"}
    for li, lw in enumerate(raw["layers"]):
        for s, n in names.items():
            sd["model.layers.%d.%s.weight" % (li, n)] = t(lw[s])
        sd["model.layers.%d.input_layernorm.weight" % li] = t(lw["norm_in"])
        sd["model.layers.%d.post_attention_layernorm.weight" % li] = t(lw["norm_post"])
        sd["model.layers.%d.self_attn.q_norm.weight" % li] = t(lw["qn"])
        sd["model.layers.%d.self_attn.k_norm.weight" % li] = t(lw["kn"])
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "rotary" not in m], missing
    ids = prompt_ids(cfg, 24, seed=5)
    with torch.no_grad():
        ref = hf(torch.from_numpy(ids.astype(np.int64))[None]).logits[0].numpy()
    agree = 0
    for pos, tok in enumerate(ids):
        nxt, logits, _ = om.decode(int(tok), pos)
        lg = O.bf16_to_f32(logits)
        scale = np.abs(ref[pos]).max()
        assert np.abs(lg - ref[pos]).max() <= 0.04 * scale, "pos %d: %g of max" % (pos, np.abs(lg - ref[pos]).max() / scale)
        agree += int(nxt == int(ref[pos].argmax()))
    assert agree >= len(ids) - 2   # bf16 round trips may flip a near-tie, never the bulk


def _load(name):
    p = os.path.join(GOLD, name)
    if not os.path.exists(p):
        pytest.fail("golden fixture %s missing: run python tests/golden/make_golden.py" % name)
    return np.load(p)


def test_golden_quant_and_linear():
    g = _load("q4_linear.npz")
    ow = O.quantize(g["w"], int(g["m"]), int(g["k"]), O.Q4)
    assert np.array_equal(ow.data.view(np.uint8), g["packed"])
    assert np.array_equal(ow.zero, g["zero"]) and np.array_equal(ow.step, g["step"])
    assert np.array_equal(O.dequant(ow).reshape(-1), g["dequant"])
    assert np.array_equal(O.linear(ow, g["x"]), g["y"])


def test_golden_normal_float_quant_and_linear():
    g = _load("nf4_linear.npz")
    m, k = int(g["m"]), int(g["k"])
    for bits in (4, 3):
        ow = O.quantize_nf4(g["w"], m, k, bits=bits)
        assert np.array_equal(ow.data, g["packed%d" % bits]) and np.array_equal(ow.lut, g["lut%d" % bits])
        assert np.array_equal(O.dequant(ow).reshape(-1), g["dequant%d" % bits])
        assert np.array_equal(O.linear(ow, g["x"]), g["y%d" % bits])


@pytest.mark.parametrize("name", ["tiny_q4", "tiny_bool1", "small_q4", "tiny_q4_std002", "tiny_nf4"])
def test_golden_decode(name):
    g = _load("decode_%s.npz" % name)
    cfg = dict(synth.CONFIGS[str(g["cfg_name"])])
    raw = synth.raw_weights_numpy(cfg, int(g["seed"]), w_std=float(g["w_std"]))
    om = oracle_model(cfg, raw, int(g["layer_type"]), int(g["head_type"]))
    prompt = g["prompt"]
    ids = om.generate(prompt.tolist(), len(g["ids"]))
    assert ids == g["ids"].tolist()
    # the canonical order (the one the kernels reproduce bit for bit) generates the same ids: the GPU test asserts them exactly in that order
    O.set_order(O.ORDER_CANON)
    try:
        omc = oracle_model(cfg, raw, int(g["layer_type"]), int(g["head_type"]), attn_mode=O.ATTN_CANON)
        assert omc.generate(prompt.tolist(), len(g["ids"])) == g["ids"].tolist()
        omc.close()
    finally:
        O.set_order(O.ORDER_DOT16)
    om2 = oracle_model(cfg, raw, int(g["layer_type"]), int(g["head_type"]))
    _, logits, _ = om2.decode(int(prompt[0]), 0)
    assert np.array_equal(logits, g["logits0"])
