"""bench.py prints ONE compact JSON line (the 20.6 KB line of round 5 was not parsed by the driver).  CPU test: the compact record built from a
canned full result -- the round-5 line as it was committed (profiles/r05f_bench_line.json), i.e. every side object at its real size -- stays under the hard limit,
is strict JSON and carries the contract's fields."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("kf_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _strict(line):
    def no_const(c):
        raise ValueError("non-finite constant %s in the line" % c)
    return json.loads(line, parse_constant=no_const)


def _canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05f_bench_line.json")))
    full["wall_s"] = 123.4
    return full


def test_compact_line_is_small_strict_and_complete():
    B = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 15000   # the canned result is the oversized one
    line = json.dumps(B.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < B.LINE_LIMIT == 8192
    assert len(line) < 4608, len(line)   # the target (<= ~4 KB) with room for the keys added this round
    d = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["roofline"]["frac"] == full["roofline"]["frac"] and d["roofline"]["bound"] == "hbm" and "traffic" in d["roofline"] and d["roofline"]["peak"] == 8000.0
    assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["kind"] == "port"
    assert d["cpu_baseline"]["ids_equal"] == [49, 49] and d["cpu_baseline"]["logits_equal"] == [151936, 151936]
    assert "workload" in d["config"] and "model" not in d["config"]
    side = d["side"]
    assert side["config3_train_step"]["ms"] == full["config3_train_step"]["ms"]
    assert side["config5_sparse_1bit"]["parity"] is True
    assert side["config4_one_gpu"]["cpu_4_layer_slice"]["parity"] is True
    assert side["config4_one_gpu"]["tp8_ranks_as_xcds"]["parity"] is True
    assert d["detail"] == "bench_detail.json"


def test_emit_writes_the_detail_file_and_prints_one_line(tmp_path, capsys, monkeypatch):
    B = _bench()
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    full = _canned()
    B.emit(full)
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8192
    d = _strict(lines[0])
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    det = json.load(open(tmp_path / "bench_detail.json"))
    assert det["xcd_replicas"]["parity"]["sequence_0_ids_equal_single_sequence_engine"] is True   # nothing is lost: the detail keeps every object


def test_a_line_that_would_not_fit_drops_its_side_objects_not_the_contract(capsys, tmp_path, monkeypatch):
    B = _bench()
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    monkeypatch.setattr(B, "LINE_LIMIT", 1500)
    B.emit(_canned())
    d = _strict(capsys.readouterr().out.strip())
    assert "dropped" in d["side"] and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0


def test_nan_is_refused():
    B = _bench()
    full = _canned()
    full["value"] = float("nan")
    try:
        json.dumps(B.compact_line(full), allow_nan=False)
    except ValueError:
        return
    raise AssertionError("a NaN reached the line")


def test_compact_line_of_the_round_6_result():
    """the same on this round's own full result (profiles/r06g_bench_detail.json: batched decoders, request queue, prompt batches, every side leg): under the target size,
    strict JSON, and the queue's figures ride in the line"""
    B = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06g_bench_detail.json")))
    line = json.dumps(B.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < 4608, len(line)
    d = _strict(line)
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"] and d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    x = d["side"]["xcd_replicas"]
    assert x["streams"] == 32 and x["parity"] is True
    q = x["request_queue"]
    assert q["generated_tokens_per_s"] > 0 and q["prefill_batch"]["ms_per_prompt"] < 1.0
    assert d["side"]["config3_train_step"]["host_loop"].startswith("C++")
