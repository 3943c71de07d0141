"""`.kun` checkpoints on the GPU (Fish::SAFETENSOR_Serialize save / load, Serialize.cpp:873-976; SerialGamaData, huTensor.cu:413-458): a model
saved from HBM and loaded back is the SAME model -- every `data||gama` blob in the file is the device blob byte for byte, and the reloaded
model's logits and ids are bit-identical, through the per-layer launches and through the persistent engine."""
import json
import struct

import msgpack
import numpy as np
import pytest

from helpers import oracle_model, prompt_ids
from koifish_amd import lib as L
from koifish_amd import synth
from koifish_amd.runtime import Qwen3

pytestmark = pytest.mark.gpu
HF = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
DTYPE = {L.BF16: "BF16(E8)", L.Q4: "Q<4>", L.T_SIGN: "TERNARY", L.BOOL1: "BOOL<1>", L.T_BINARY: "BINARY", L.F8E5M2: "F8E5M2", L.NF4: "Q<4>"}


def _parse(path):
    raw = path.read_bytes()
    hlen = struct.unpack("<Q", raw[:8])[0]
    return json.loads(raw[8:8 + hlen]), raw[8 + hlen:]


@pytest.mark.parametrize("layer_type,head_type,tied", [(L.Q4, L.BF16, True), (L.Q4, L.Q4, False), (L.T_SIGN, L.BF16, True), (L.BOOL1, L.Q4, True),
                                                         (L.F8E5M2, L.BF16, False), (L.NF4, L.BF16, True), (L.NF4, L.NF4, False), (L.BF16, L.BF16, True)])
def test_save_then_load_is_the_same_model(tmp_path, layer_type, head_type, tied):
    cfg = dict(synth.CONFIGS["tiny"], tied=tied)
    raw = synth.raw_weights_numpy(cfg, 77, w_std=0.1)
    a = synth.build_from_raw(cfg, raw, layer_type, head_type)
    path = tmp_path / "tiny.kun"
    a.save_kun(path)
    hdr, data = _parse(path)
    # the file holds the device blobs as they are
    for (layer, slot), w in a.weights.items():
        if tied and (layer, slot) == (-1, 1):
            continue                                    # the tied head refers to the embedding's tensor: not written (isRefer, Serialize.cpp:936)
        name = ("model.layers.%d.%s.weight" % (layer, HF[slot])) if layer >= 0 else ("model.embed_tokens.weight" if slot == 0 else "lm_head.weight")
        d = hdr[name]
        want = DTYPE[layer_type if layer >= 0 else head_type]
        assert d["dtype"] == want and d["shape"] == [w.ne0, w.ne1] and d["szData"] == w.szData and d["szData"] + d["szGama"] == w.blob.numel()
        assert data[d["data_offsets"][0]:d["data_offsets"][1]] == w.blob.cpu().numpy().tobytes(), name
    assert ("lm_head.weight" in hdr) == (not tied)
    js = msgpack.unpackb(data[hdr["__koifish__config__"]["data_offsets"][0]:])
    p = js["CLI_params"]["config"]["model"]["parameter"]
    assert js["vendor"] == "gruai" and p["Layer"] == cfg["n_layer"] and p["tie_word_embeddings"] == tied
    assert p["transformer"] == {"Ctx": cfg["max_seq"], "Embed": cfg["dim"], "Ffn": cfg["ffn"], "Head": cfg["n_head"], "KVHead": cfg["n_kv"], "head_dim": cfg["head_dim"]}
    assert set(js["tensors"]) == set(hdr) - {"__metadata__", "__koifish__config__"}

    b = Qwen3.from_kun(path)
    assert (b.cfg["dim"], b.cfg["n_layer"], b.cfg["n_kv"], b.cfg["head_dim"], b.cfg["vocab"], b.cfg["tied"], b.cfg["max_seq"]) == (
        cfg["dim"], cfg["n_layer"], cfg["n_kv"], cfg["head_dim"], cfg["vocab"], tied, cfg["max_seq"])
    prompt = prompt_ids(cfg, 10)
    tok = int(prompt[0])
    for pos in range(12):
        na, la = a.forward(tok, pos)
        nb, lb = b.forward(tok, pos)
        assert na == nb and np.array_equal(la, lb), "step %d" % pos
        tok = int(prompt[pos + 1]) if pos + 1 < len(prompt) else na
    assert a.generate(prompt, 16) == b.generate(prompt, 16)
    # and a second generation of the file is the same file
    path2 = tmp_path / "again.kun"
    b.save_kun(path2)
    assert path2.read_bytes() == path.read_bytes()
    a.close()
    b.close()


def test_reloaded_model_runs_the_engine_and_follows_the_oracle(tmp_path):
    cfg = synth.CONFIGS["small"]
    raw = synth.raw_weights_numpy(cfg, 5, w_std=0.05)
    a = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    path = tmp_path / "small.kun"
    a.save_kun(path)
    b = Qwen3.from_kun(path, max_seq=128)              # the context length may be overridden at load
    assert b.cfg["max_seq"] == 128
    prompt = prompt_ids(cfg, 12)
    ids = b.generate(prompt, 24, use_graph=True)
    assert b.engine_steps() > 0, "the reloaded 4-bit model must be served by the persistent engine"
    assert ids == a.generate(prompt, 24, use_graph=True)
    om = oracle_model(cfg, raw, L.Q4, L.BF16)
    assert ids == om.generate(prompt.tolist(), 24)
    a.close()
    b.close()


def test_kun_load_errors(tmp_path):
    with pytest.raises(L.KFError, match="cannot open"):
        Qwen3.from_kun(tmp_path / "nothing.kun")
    cfg = synth.CONFIGS["tiny"]
    raw = synth.raw_weights_numpy(cfg, 3, w_std=0.1)
    a = synth.build_from_raw(cfg, raw, L.Q4, L.BF16)
    path = tmp_path / "t.kun"
    a.save_kun(path)
    a.close()
    hdr, data = _parse(path)
    # a config that promises another group size: the gama no longer fits the card
    off = hdr["__koifish__config__"]["data_offsets"][0]
    js = msgpack.unpackb(data[off:])
    js["CLI_params"]["config"]["quantizer"]["self_attn"]["group_size"] = 64
    js["CLI_params"]["config"]["quantizer"]["group_size"] = 64
    pack = msgpack.packb(js)
    hdr["__koifish__config__"]["data_offsets"] = [off, off + len(pack)]
    hdr["__koifish__config__"]["shape"] = [len(pack)]
    text = json.dumps(hdr, separators=(",", ":")).encode()
    bad = tmp_path / "bad.kun"
    bad.write_bytes(struct.pack("<Q", len(text)) + text + data[:off] + pack)
    with pytest.raises(L.KFError, match="szData/szGama"):
        Qwen3.from_kun(bad)
    # a file without the config tensor
    hdr2 = {k: v for k, v in hdr.items() if k != "__koifish__config__"}
    text = json.dumps(hdr2, separators=(",", ":")).encode()
    bare = tmp_path / "bare.kun"
    bare.write_bytes(struct.pack("<Q", len(text)) + text + data[:off])
    with pytest.raises(L.KFError, match="__koifish__config__"):
        Qwen3.from_kun(bare)
