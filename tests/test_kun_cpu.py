"""`.kun` container (the reference's own checkpoint file: K_SafeTensors::Register / insertJS / _to_ofs, Serialize.cpp:554-665, 849-871;
GTensor::jDesc, Serialize.cpp:61-103): the C++ writer and reader of koifish_amd/host/kf_safetensors.cpp checked without a GPU.
The independent checkers are python's json / struct / msgpack: the file is parsed here byte by byte, and the config tensor -- MessagePack in
the encoding nlohmann::json::to_msgpack chooses -- is compared with msgpack.packb's bytes."""
import ctypes as C
import json
import struct

import msgpack
import numpy as np
import pytest

from koifish_amd import lib as L

# the model / quantizer part of the config a reference checkpoint carries (log/@[._checkpoints_._koifish_state_.ckp].json of the reference,
# the "_detail.json" side file Fish::SAFETENSOR_Serialize writes, Serialize.cpp:957-960; same keys as cases/qwen3/qwen3_596M_q4.json)
REF_CONFIG = {
    "version": "0.1.0",
    "quantizer": {"train_target": "gama", "group_size": 128, "self_attn": {"quant_method": "RTN", "bits": 4}, "mlp": {"quant_method": "RTN", "bits": 4}},
    "model": {"arch": "QWEN3",
              "parameter": {"Layer": 6, "transformer": {"Ctx": 1024, "Embed": 1024, "Ffn": 3072, "Head": 16, "KVHead": 8, "head_dim": 128},
                            "tie_word_embeddings": True, "max_pos_embeddings": 32768},
              "backbone": {"embed_tokens": {"Embedding": []}, "layer": {"self_attn": {"QKV": []}, "mlp": {"FFN": []}}, "norm": {"Normal": []},
                           "output": {"CLASIFY": []}}},
    "train": {"dump-every": 10, "gpt-every": -10, "epoch": 1, "batch": 16},
    "seed": 42, "checkpoint_in": None,
}


@pytest.fixture(scope="module")
def host():
    return L.load()[1]


def _pack(host, doc_text):
    n = host.kfh_json_to_msgpack(doc_text.encode(), None, 0)
    assert n > 0, host.kfh_last_error()
    buf = (C.c_ubyte * n)()
    assert host.kfh_json_to_msgpack(doc_text.encode(), buf, n) == n
    return bytes(buf)


def _unpack(host, raw):
    buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
    n = host.kfh_msgpack_to_json(buf, len(raw), None, 0)
    assert n > 0, host.kfh_last_error()
    out = C.create_string_buffer(n)
    assert host.kfh_msgpack_to_json(buf, len(raw), out, n) == n
    return json.loads(out.value.decode())


DOCS = [
    REF_CONFIG,
    {"ints": [0, 1, 127, 128, 255, 256, 65535, 65536, 2 ** 32 - 1, 2 ** 32, 2 ** 40, -1, -32, -33, -128, -129, -32768, -32769, -2 ** 31, -2 ** 31 - 1, -2 ** 40]},
    {"s31": "a" * 31, "s32": "b" * 32, "s255": "c" * 255, "s256": "d" * 256, "s70000": "e" * 70000, "utf8": "天命玄鸟", "esc": "q\"\\\n\t"},
    {"arr15": list(range(15)), "arr16": list(range(16)), "arr70000": [1] * 70000, "nested": [[[]], {}, [{}]]},
    {("k%d" % i): i for i in range(16)},
    {("k%d" % i): None for i in range(70000)},
    {"t": True, "f": False, "n": None, "e": "", "tensors": {"model.embed_tokens.weight": 0, "model.layers.0.self_attn.q_proj.weight": 1544148992}},
]


@pytest.mark.parametrize("doc", DOCS, ids=lambda d: list(d)[0][:12])
def test_msgpack_bytes_equal_pythons_packer(host, doc):
    text = json.dumps(doc, ensure_ascii=False)
    raw = _pack(host, text)
    assert raw == msgpack.packb(doc), "encoder differs from the MessagePack reference packer"
    assert _unpack(host, raw) == doc
    assert _unpack(host, msgpack.packb(doc)) == doc


def test_msgpack_floats_take_the_compact_form(host):
    # nlohmann's write_compact_float: float32 (0xca) when the double survives the round trip through float, float64 (0xcb) otherwise
    assert _pack(host, "0.5") == b"\xca" + struct.pack(">f", 0.5)
    assert _pack(host, "1000000.0") == b"\xca" + struct.pack(">f", 1e6)
    assert _pack(host, "0.1") == b"\xcb" + struct.pack(">d", 0.1)
    assert _pack(host, "1e-06") == b"\xcb" + struct.pack(">d", 1e-6)
    assert _pack(host, "1e300") == b"\xcb" + struct.pack(">d", 1e300)
    assert _pack(host, "2") == b"\x02"           # an integer token stays an integer
    doc = {"rope_theta": 1000000.0, "rms_norm_eps": 1e-06, "lr": 0.0006, "neg": -2.5}
    assert msgpack.unpackb(_pack(host, json.dumps(doc))) == doc
    assert _unpack(host, msgpack.packb(doc)) == doc
    assert _unpack(host, msgpack.packb(doc, use_single_float=True)) == {k: struct.unpack(">f", struct.pack(">f", v))[0] for k, v in doc.items()}


def test_msgpack_malformed_is_refused(host):
    good = msgpack.packb({"a": [1, 2, 3], "b": "xyz"})
    for bad in (good[:-1], good + b"\x00", b"\xc1", b"\xdd\xff\xff\xff\xff", b"\x81\x01\x01", b""):
        buf = (C.c_ubyte * max(1, len(bad))).from_buffer_copy(bad or b"\x00")
        assert host.kfh_msgpack_to_json(buf, len(bad), None, 0) < 0


def _entries(seed=0):
    """tensors as a 4-bit model would store them: data||gama blobs with the reference's K_FLOATS dtype names"""
    rng = np.random.default_rng(seed)
    ne0, ne1, g = 64, 256, 128
    n = ne0 * ne1
    out = []
    for name, dtype, bits in (("model.embed_tokens.weight", "BF16(E8)", 16), ("model.layers.0.self_attn.q_proj.weight", "Q<4>", 4),
                              ("model.layers.0.mlp.gate_proj.weight", "TERNARY", 2), ("model.layers.0.mlp.down_proj.weight", "BINARY", 1),
                              ("model.layers.0.self_attn.k_proj.weight", "F8E5M2", 8)):
        szData = n * bits // 8
        szGama = (ne0 + ne1 + 2 * (n // g)) * 2 if bits < 8 else 0
        out.append((name, dtype, (ne0, ne1), szData, szGama, rng.integers(0, 256, szData + szGama, dtype=np.uint8)))
    out.append(("model.norm.weight", "BF16(E8)", (ne1,), ne1 * 2, 0, rng.integers(0, 256, ne1 * 2, dtype=np.uint8)))
    return out


def _write(host, path, entries, config):
    n = len(entries)
    names = (C.c_char_p * n)(*[e[0].encode() for e in entries])
    dtypes = (C.c_char_p * n)(*[e[1].encode() for e in entries])
    shape4 = np.zeros((n, 4), np.int64)
    for i, e in enumerate(entries):
        shape4[i, :len(e[2])] = e[2]
    szd = np.array([e[3] for e in entries], np.uint64)
    szg = np.array([e[4] for e in entries], np.uint64)
    blobs = (C.c_void_p * n)(*[e[5].ctypes.data for e in entries])
    return host.kfh_kun_write(str(path).encode(), n, names, dtypes, shape4.ctypes.data, szd.ctypes.data, szg.ctypes.data, blobs, json.dumps(config).encode())


def test_kun_file_layout_byte_by_byte(host, tmp_path):
    entries = _entries()
    config = {"vendor": "gruai", "CLI_params": {"config": REF_CONFIG}, "tokenizer": {"tokens": ""}}
    path = tmp_path / "model.kun"
    assert _write(host, path, entries, config) == 0, host.kfh_last_error()
    raw = path.read_bytes()
    hlen = struct.unpack("<Q", raw[:8])[0]
    hdr = json.loads(raw[8:8 + hlen])                       # dicts keep the file's key order
    data = raw[8 + hlen:]
    keys = list(hdr)
    assert keys[0] == "__metadata__" and hdr["__metadata__"] == {"format": "pt", "writer": "koifish"}          # UpdateMetaData
    assert keys[1:-1] == [e[0] for e in entries] and keys[-1] == "__koifish__config__"                        # registration order; config last
    off = 0
    for name, dtype, shape, szd, szg, blob in entries:
        d = hdr[name]
        assert list(d) == ["dtype", "shape", "data_offsets", "loAB", "szGama", "szData"]                       # GTensor::jDesc order
        assert (d["dtype"], d["shape"], d["data_offsets"], d["loAB"], d["szGama"], d["szData"]) == (dtype, list(shape), [off, off + szd + szg], 0, szg, szd)
        assert data[off:off + szd + szg] == blob.tobytes()                                                     # data||gama in one piece
        off += szd + szg
    c = hdr["__koifish__config__"]
    assert c["dtype"] == "U8" and c["data_offsets"][0] == off and c["data_offsets"][1] == len(data) and c["shape"] == [len(data) - off]
    js = msgpack.unpackb(data[off:])
    want = dict(config, tensors={e[0]: o for e, o in zip(entries, np.cumsum([0] + [e[3] + e[4] for e in entries])[:-1].tolist())})
    assert js == want and list(js) == ["vendor", "CLI_params", "tokenizer", "tensors"]                         # jsConfig["tensors"][name] = offset
    assert data[off:] == msgpack.packb(want)
    assert not (tmp_path / "model.kun.tmp").exists()


def test_kun_reader_round_trip(host, tmp_path):
    entries = _entries(1)
    config = {"vendor": "gruai", "CLI_params": {"config": REF_CONFIG}, "tokenizer": {"tokens": ""}}
    path = tmp_path / "m.kun"
    assert _write(host, path, entries, config) == 0
    h = C.c_void_p(host.kfh_st_open(str(path).encode(), 0))
    assert h, host.kfh_last_error()
    assert host.kfh_st_count(h) == len(entries) + 1
    for i, (name, dtype, shape, szd, szg, blob) in enumerate(entries):
        nm, dt = C.create_string_buffer(256), C.create_string_buffer(16)
        sh = (C.c_int64 * 4)()
        nd, b, e = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        assert host.kfh_st_info(h, i, nm, 256, dt, 16, sh, C.byref(nd), C.byref(b), C.byref(e)) == 0
        assert (nm.value.decode(), dt.value.decode(), tuple(sh[:nd.value]), e.value - b.value) == (name, dtype, tuple(shape), szd + szg)
        d, g = C.c_uint64(0), C.c_uint64(0)
        assert host.kfh_st_blob_sizes(h, i, C.byref(d), C.byref(g)) == 0 and (d.value, g.value) == (szd, szg)
        buf = (C.c_ubyte * (szd + szg))()
        assert host.kfh_st_read(h, name.encode(), buf, szd + szg) == 0 and bytes(buf) == blob.tobytes()
    n = host.kfh_st_config_json(h, None, 0)
    out = C.create_string_buffer(n)
    host.kfh_st_config_json(h, out, n)
    js = json.loads(out.value.decode())
    assert js["CLI_params"]["config"] == REF_CONFIG and js["vendor"] == "gruai" and set(js["tensors"]) == {e[0] for e in entries}
    host.kfh_st_close(h)


def test_kun_empty_and_malformed(host, tmp_path):
    # no tensors at all: header + config only
    path = tmp_path / "empty.kun"
    assert _write(host, path, [], {"vendor": "gruai"}) == 0
    h = C.c_void_p(host.kfh_st_open(str(path).encode(), 0))
    assert h and host.kfh_st_count(h) == 1
    host.kfh_st_close(h)
    # an entry whose szData + szGama disagree with its data_offsets is refused
    entries = _entries(2)
    good = tmp_path / "good.kun"
    assert _write(host, good, entries, {"vendor": "gruai"}) == 0
    raw = good.read_bytes()
    hlen = struct.unpack("<Q", raw[:8])[0]
    hdr = json.loads(raw[8:8 + hlen])
    hdr[entries[1][0]]["szGama"] += 2
    text = json.dumps(hdr, separators=(",", ":")).encode()
    bad = tmp_path / "bad.kun"
    bad.write_bytes(struct.pack("<Q", len(text)) + text + raw[8 + hlen:])
    assert not host.kfh_st_open(str(bad).encode(), 0)
    assert b"data_offsets" in host.kfh_last_error()
    # truncated file: the last tensor runs past the end
    cut = tmp_path / "cut.kun"
    cut.write_bytes(raw[:-7])
    assert not host.kfh_st_open(str(cut).encode(), 0)
    # a safetensors file without the config tensor has no config
    h = C.c_void_p(host.kfh_st_open(str(good).encode(), 0))
    assert host.kfh_st_config_json(h, None, 0) > 0
    host.kfh_st_close(h)
    # config text that is not an object is refused by the writer
    assert _write(host, tmp_path / "x.kun", [], [1, 2]) != 0
