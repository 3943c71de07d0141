"""ctypes front-end of the CPU oracle (oracle/kf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (koifish_amd/) never imports this module.

All tensors are numpy arrays; bf16 is carried as uint16 bit patterns.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# typNUMBER order restated from src/g_float.hpp:84-117
F32, F64, F16, BF16, F8E5M2, F8E4M3, U8, I8, U16, I16, U32, I32, U64, I64, Q4, Q3, Q2, T_SIGN, T_SEQ, BOOL1, T_BINARY, T_BINARY_3, T_BINARY_TILE = range(23)
BITS = {BF16: 16, F16: 16, F8E5M2: 8, Q4: 4, T_SIGN: 2, Q2: 2, BOOL1: 1, T_BINARY: 1}

Q4_AWQ = 100  # oracle-internal tag: Q4 in the AutoAWQ GEMM layout
Q4_LUT = 101  # oracle-internal tag: Q4 in the row-codebook storage (GeQuant::RT_NormalF)
Q3_LUT, Q2_LUT, Q2_ROWRTN = 102, 103, 104  # 8- / 4-entry row codebooks (3- / 2-bit streams); (zero, step) per row (CU_Q22X_RTN)
ATTN_REF, ATTN_FUSED, ATTN_CANON = 0, 1, 2
ORDER_DOT16, ORDER_CANON = 0, 1  # summation order of the mat-vec: the reference's CPU primitive (dotprod_fp16) / the canonical order kernels and oracle share


def build(force=False):
    so = os.path.join(_HERE, "libkf_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("kf_oracle.c", "kfo_math.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.kfo_rtn_x.restype = C.c_float
        L.kfo_yinyang.restype = C.c_float
        L.kfo_expf_export.restype = C.c_float
        L.kfo_expf_export.argtypes = [C.c_float]
        L.kfo_qwen3_new.restype = C.c_void_p
        L.kfo_qwen3_new.argtypes = [C.c_int] * 8 + [C.c_float] * 3 + [C.c_int]
        L.kfo_qwen3_kcache.restype = C.c_void_p
        L.kfo_qwen3_vcache.restype = C.c_void_p
        L.kfo_qwen3_kcache.argtypes = [C.c_void_p]
        L.kfo_qwen3_vcache.argtypes = [C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_order(order):
    """ORDER_DOT16 (default) or ORDER_CANON: the canonical mat-vec order (kf_oracle.c section 4c).  Process-wide."""
    lib().kfo_set_order(int(order))


def get_order():
    return int(lib().kfo_get_order())


def lpr_log2(n_blk, rows):
    return int(lib().kfo_lpr_log2(int(n_blk), C.c_long(int(rows))))


class canonical:
    """with O.canonical(rows=...): the canonical order for the mat-vecs inside; rows = the rows of ALL matrices of the launch being mirrored
    (None: each matrix's own rows)"""

    def __init__(self, rows=None):
        self.rows = rows

    def __enter__(self):
        self.prev = get_order()
        set_order(ORDER_CANON)
        lib().kfo_set_launch_rows(C.c_long(int(self.rows or 0)))
        return self

    def __exit__(self, *a):
        lib().kfo_set_launch_rows(C.c_long(0))
        set_order(self.prev)


# ---------------------------------------------------------------- bf16 helpers (numpy, RNE)
def f32_to_bf16(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_to_f32(h):
    return (np.ascontiguousarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


# ---------------------------------------------------------------- layout
def pack(q, bits):
    """q: int32[..., n] with n % (128/bits) == 0 -> uint8 packed stream (Packed128 blocks)."""
    q = np.ascontiguousarray(q, dtype=np.int32).reshape(-1)
    per = 128 // bits
    assert q.size % per == 0
    out = np.zeros(q.size // per * 16, dtype=np.uint8)
    fn = getattr(lib(), "kfo_pack%d_128" % bits)
    for b in range(q.size // per):
        fn(_p(q[b * per:]), C.c_void_p(out.ctypes.data + 16 * b))
    return out


def unpack(packed, bits):
    packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    per = 128 // bits
    nb = packed.size // 16
    out = np.zeros(nb * per, dtype=np.int32)
    fn = getattr(lib(), "kfo_unpack%d_128" % bits)
    for b in range(nb):
        fn(C.c_void_p(packed.ctypes.data + 16 * b), C.c_void_p(out.ctypes.data + 4 * per * b))
    return out


def quant_range(bits, symmetric=False, yyang=False):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().kfo_quant_range(bits, int(symmetric), int(yyang), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class QWeight:
    """A weight as the reference holds it: `data||gama` (GTensor.cpp:456-510) + quant card fields."""

    def __init__(self, type_, ne0, ne1, data, zero=None, step=None, lGroup=128, qBias=0):
        self.type, self.ne0, self.ne1 = type_, ne0, ne1
        self.data = np.ascontiguousarray(data)
        self.zero = None if zero is None else np.ascontiguousarray(zero, dtype=np.uint16)
        self.step = None if step is None else np.ascontiguousarray(step, dtype=np.uint16)
        self.lGroup, self.qBias = lGroup, qBias

    @property
    def bits(self):
        return BITS[self.type]

    @property
    def nGroup(self):
        return 0 if self.zero is None else self.zero.size

    def blob(self):
        """bytes of `data || R_SCALE[ne0] C_SCALE[ne1] ZERO[nGroup] STEP[nGroup]` (bf16), gama_T layout."""
        if self.zero is None:
            return self.data.view(np.uint8).reshape(-1).copy()
        rc = np.full(self.ne0 + self.ne1, 0x3F80, dtype=np.uint16)  # 1.0; unused when rc_normal == 0
        return np.concatenate([self.data.view(np.uint8).reshape(-1), rc.view(np.uint8), self.zero.view(np.uint8), self.step.view(np.uint8)])

    def nbytes_algorithmic(self):
        """bytes a GEMV must read: packed data + zero/step (R/C scales are never read)."""
        n = self.data.nbytes
        if self.zero is not None:
            n += self.zero.nbytes + self.step.nbytes
        return n

    def cdesc(self):
        class _W(C.Structure):
            _fields_ = [("type", C.c_int), ("ne0", C.c_int), ("ne1", C.c_int), ("data", C.c_void_p), ("zero", C.c_void_p),
                        ("step", C.c_void_p), ("lGroup", C.c_int), ("qBias", C.c_int)]
        return _W(self.type, self.ne0, self.ne1, _p(self.data), _p(self.zero), _p(self.step), self.lGroup, self.qBias)


def quantize(w_bf16, ne0, ne1, type_, lGroup=128, symmetric=False):
    """GeQuant::RTN_x (4/2-bit RTN) or GeQuant::YinYang (T_SIGN ternary / BOOL1 1-bit)."""
    w_bf16 = np.ascontiguousarray(w_bf16, dtype=np.uint16).reshape(-1)
    assert w_bf16.size == ne0 * ne1 and (ne0 * ne1) % lGroup == 0
    if type_ == BF16:
        return QWeight(BF16, ne0, ne1, w_bf16.copy())
    if type_ == F16:   # BASELINE config 1: IEEE half weights (float_to_half = _cvtss_sh, round to nearest even; GST_float.cpp:60-62)
        return QWeight(F16, ne0, ne1, bf16_to_f32(w_bf16).astype(np.float16).view(np.uint16))
    if type_ == F8E5M2:
        out = np.zeros(w_bf16.size, dtype=np.uint8)
        lib().kfo_bf16_to_f8e5m2(_p(w_bf16), C.c_size_t(w_bf16.size), _p(out))
        return QWeight(F8E5M2, ne0, ne1, out)
    bits = BITS[type_]
    nG = w_bf16.size // lGroup
    packed = np.zeros(w_bf16.size * bits // 8, dtype=np.uint8)
    zero = np.zeros(nG, dtype=np.uint16)
    step = np.zeros(nG, dtype=np.uint16)
    yy = type_ in (T_SIGN, BOOL1, T_BINARY)
    if yy:
        lib().kfo_yinyang(_p(w_bf16), C.c_size_t(nG), lGroup, bits, _p(packed), _p(zero), _p(step))
    else:
        lib().kfo_rtn_x(_p(w_bf16), C.c_size_t(nG), lGroup, bits, int(symmetric), 0, _p(packed), _p(zero), _p(step))
    _, _, qBias = quant_range(bits, symmetric, yy)
    return QWeight(type_, ne0, ne1, packed, zero, step, lGroup, qBias)


def dequant(w):
    out = np.zeros(w.ne0 * w.ne1, dtype=np.uint16)
    d = w.cdesc()
    lib().kfo_dequant_weight(C.byref(d), _p(out))
    return out.reshape(w.ne0, w.ne1)


class AWQWeight(QWeight):
    """qweight int32 [in, out/8], qzeros int32 [in/128, out/8], scales fp16 [in/128, out]; logical W[out, in]."""

    def __init__(self, n_out, n_in, qweight, qzeros, scales):
        self.type, self.ne0, self.ne1 = Q4_AWQ, n_out, n_in
        self.data = np.ascontiguousarray(qweight, dtype=np.uint32)
        self.qzeros = np.ascontiguousarray(qzeros, dtype=np.uint32)
        self.scales = np.ascontiguousarray(scales, dtype=np.float16)
        self.zero, self.step = self.qzeros, self.scales  # carried in the zero/step slots of the C descriptor
        self.lGroup, self.qBias = 128, 0

    @property
    def bits(self):
        return 4

    def cdesc(self):
        class _W(C.Structure):
            _fields_ = [("type", C.c_int), ("ne0", C.c_int), ("ne1", C.c_int), ("data", C.c_void_p), ("zero", C.c_void_p),
                        ("step", C.c_void_p), ("lGroup", C.c_int), ("qBias", C.c_int)]
        return _W(self.type, self.ne0, self.ne1, _p(self.data), _p(self.qzeros), _p(self.scales), 128, 0)

    def nbytes_algorithmic(self):
        return self.data.nbytes + self.qzeros.nbytes + self.scales.nbytes


class LutWeight(QWeight):
    """Row-codebook weight (GeQuant::RT_NormalF storage): MSB-first `bits`-wide stream + a 2^bits-entry bf16 table per row (bits 4, 3 or 2);
    rtn=True (bits 2): a (zero, step) pair per row instead of a table (CU_Q22X_RTN)."""

    def __init__(self, ne0, ne1, data, lut, bits=4, rtn=False):
        self.type = Q2_ROWRTN if rtn else {4: Q4_LUT, 3: Q3_LUT, 2: Q2_LUT}[bits]
        self.ne0, self.ne1, self._bits, self.rtn = ne0, ne1, bits, rtn
        self.data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        assert self.data.size == ne0 * ne1 * bits // 8
        self.lut = np.ascontiguousarray(lut, dtype=np.uint16).reshape(ne0, 2 if rtn else 1 << bits)
        self.zero, self.step = self.lut, None
        self.lGroup, self.qBias = 0, 0

    @property
    def bits(self):
        return self._bits

    def blob(self):
        """bytes of `data || R_SCALE[ne0] C_SCALE[ne1] LUT[ne0 x 16]` (bf16): gama_T layout with the LUT at +ne0+ne1 (GeQuant.cpp:710)"""
        rc = np.full(self.ne0 + self.ne1, 0x3F80, dtype=np.uint16)
        return np.concatenate([self.data, rc.view(np.uint8), self.lut.reshape(-1).view(np.uint8)])

    def nbytes_algorithmic(self):
        return self.data.nbytes + self.lut.nbytes

    def cdesc(self):
        class _W(C.Structure):
            _fields_ = [("type", C.c_int), ("ne0", C.c_int), ("ne1", C.c_int), ("data", C.c_void_p), ("zero", C.c_void_p),
                        ("step", C.c_void_p), ("lGroup", C.c_int), ("qBias", C.c_int)]
        return _W(self.type, self.ne0, self.ne1, _p(self.data), _p(self.lut), None, 0, 0)


def quantize_nf4(w_bf16, ne0, ne1, want_err=False, bits=4):
    """GeQuant::RT_NormalF (4- or 3-bit normal-float row codebooks)."""
    w_bf16 = np.ascontiguousarray(w_bf16, dtype=np.uint16).reshape(-1)
    assert bits in (4, 3) and w_bf16.size == ne0 * ne1 and ne1 % 8 == 0
    packed = np.zeros(ne0 * ne1 * bits // 8, dtype=np.uint8)
    lut = np.zeros(ne0 << bits, dtype=np.uint16)
    lib().kfo_lut_quantize_nf.restype = C.c_float
    err = lib().kfo_lut_quantize_nf(_p(w_bf16), ne0, ne1, bits, _p(packed), _p(lut))
    w = LutWeight(ne0, ne1, packed, lut, bits)
    return (w, float(err)) if want_err else w


def quantize_nf3(w_bf16, ne0, ne1, want_err=False):
    return quantize_nf4(w_bf16, ne0, ne1, want_err, bits=3)


def nf4_table():
    lib().kfo_nf4_table.restype = C.POINTER(C.c_float)
    return np.array([lib().kfo_nf4_table()[i] for i in range(16)], dtype=np.float32)


def nf3_table():
    lib().kfo_nf3_table.restype = C.POINTER(C.c_float)
    return np.array([lib().kfo_nf3_table()[i] for i in range(8)], dtype=np.float32)


def dequant_awq(w):
    """[in, out] bf16, as the reference's GetDataX leaves it (TransA = 0)"""
    out = np.zeros((w.ne1, w.ne0), dtype=np.uint16)
    lib().kfo_dequant_awq(_p(w.data), _p(w.qzeros), _p(w.scales), w.ne1, w.ne0, _p(out))
    return out


def awq_pack(q_int, order=(0, 2, 4, 6, 1, 3, 5, 7)):
    """int array [..., n] (values 0..15) -> uint32 [..., n/8] in AutoAWQ nibble order (AWQ_ORDER of src/Python/test_awq.py:50)"""
    q = np.asarray(q_int, dtype=np.uint32)
    q = q.reshape(q.shape[:-1] + (-1, 8))
    out = np.zeros(q.shape[:-1], dtype=np.uint32)
    for pos, k in enumerate(order):    # nibble `pos` of the word holds element order[pos]
        out |= (q[..., k] & 0xF) << np.uint32(4 * pos)
    return out


def dequant_q128(packed, zero, step, lGroup, bits, qBias):
    out = np.zeros(zero.size * lGroup, dtype=np.uint16)
    lib().kfo_dequant_q128(_p(packed), _p(zero), _p(step), C.c_size_t(zero.size), lGroup, bits, qBias, _p(out))
    return out


def linear(w, x, bias=None, alpha=1.0, beta=0.0, y=None):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    y = np.zeros(w.ne0, dtype=np.uint16) if y is None else np.ascontiguousarray(y, dtype=np.uint16).copy()
    d = w.cdesc()
    lib().kfo_linear(C.byref(d), _p(x), _p(y), _p(bias), C.c_float(alpha), C.c_float(beta))
    return y


def linear_masked(w, x, hot, bias=None):
    """D_matmul_sparse: y[i] = (hot[i] == 1 ? W[i,:].x : 0) (+ bias[i])"""
    x = np.ascontiguousarray(x, dtype=np.uint16)
    hot = np.ascontiguousarray(hot, dtype=np.int32)
    y = np.zeros(w.ne0, dtype=np.uint16)
    d = w.cdesc()
    lib().kfo_linear_masked(C.byref(d), _p(x), _p(y), _p(bias), _p(hot))
    return y


def linear_f32(w, x, c0=0, c1=None):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    y = np.zeros(w.ne0, dtype=np.float32)
    d = w.cdesc()
    lib().kfo_linear_f32(C.byref(d), _p(x), _p(y), c0, w.ne1 if c1 is None else c1)
    return y


def rmsnorm(x, w, eps=1e-6):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    rows = 1 if x.ndim == 1 else x.shape[0]
    dim = x.shape[-1]
    y = np.zeros_like(x)
    lib().kfo_rmsnorm(_p(x), _p(np.ascontiguousarray(w, dtype=np.uint16)), _p(y), rows, dim, C.c_float(eps))
    return y


def headnorm(x, w, nHead, hd, eps=1e-6):
    x = np.ascontiguousarray(x, dtype=np.uint16).copy()
    lib().kfo_headnorm(_p(x), _p(np.ascontiguousarray(w, dtype=np.uint16)), nHead, hd, C.c_float(eps))
    return x


def rope_table(pos, hd, theta):
    c = np.zeros(hd // 2, dtype=np.float32)
    s = np.zeros(hd // 2, dtype=np.float32)
    lib().kfo_rope_table(pos, hd, C.c_float(theta), _p(c), _p(s))
    return c, s


def rope(x, nHead, hd, pos, theta):
    x = np.ascontiguousarray(x, dtype=np.uint16).copy()
    lib().kfo_rope(_p(x), nHead, hd, pos, C.c_float(theta))
    return x


def swiglu(gate, up):
    gate = np.ascontiguousarray(gate, dtype=np.uint16)
    out = np.zeros_like(gate)
    lib().kfo_swiglu(_p(gate), _p(np.ascontiguousarray(up, dtype=np.uint16)), _p(out), gate.size)
    return out


def add(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    out = np.zeros_like(a)
    lib().kfo_add(_p(a), _p(np.ascontiguousarray(b, dtype=np.uint16)), _p(out), a.size)
    return out


def embed(w, token):
    out = np.zeros(w.ne1, dtype=np.uint16)
    d = w.cdesc()
    lib().kfo_embed(C.byref(d), int(token), _p(out))
    return out


def argmax_bf16(logits):
    logits = np.ascontiguousarray(logits, dtype=np.uint16)
    return int(lib().kfo_argmax_bf16(_p(logits), logits.size))


def sample(logits, top_k, temperature, top_p, rng_state, want_detail=False, true_topk=False):
    """GeneratOnPrompt::Sample (non-greedy branch); rng_state is a 1-element uint64 array advanced in place.
    true_topk: the k largest logits as candidates instead of the set the reference's heap keeps (kfo_sample_topk)."""
    logits = np.ascontiguousarray(logits, dtype=np.uint16)
    assert rng_state.dtype == np.uint64 and rng_state.size == 1
    k = min(top_k, logits.size)
    picks = np.zeros(max(k, 1), dtype=np.int32)
    probs = np.zeros(max(k, 1), dtype=np.float32)
    npick = C.c_int(0)
    fn = lib().kfo_sample_topk if true_topk else lib().kfo_sample
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    tok = int(fn(_p(logits), logits.size, int(top_k), float(temperature), float(top_p), _p(rng_state), _p(picks), _p(probs), C.byref(npick)))
    if want_detail:
        return tok, picks, probs, npick.value
    return tok


def layernorm(x, w, b, eps=1e-5, want_stats=False):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    rows, dim = (1, x.size) if x.ndim == 1 else x.shape
    y = np.zeros_like(x)
    mean, rstd = np.zeros(rows, np.float32), np.zeros(rows, np.float32)
    fn = lib().kfo_layernorm
    fn.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    fn(_p(x), _p(np.ascontiguousarray(w, dtype=np.uint16)), _p(np.ascontiguousarray(b, dtype=np.uint16)) if b is not None else None, _p(y), rows, dim, eps,
       _p(mean), _p(rstd))
    return (y, mean, rstd) if want_stats else y


def norm_backward(dinp, dweight, dbias, dout, inp, weight, mean, rstd):
    """LayerNorm (mean given) / RMSNorm (mean None) backward, in place on uint16 dinp [rows, C], dweight [C], dbias [C] or None"""
    for a in (dinp, dweight, dout, inp, weight):
        assert a.dtype == np.uint16 and a.flags.c_contiguous
    rows, C_ = dinp.shape
    rs = np.ascontiguousarray(rstd, dtype=np.float32)
    mn = np.ascontiguousarray(mean, dtype=np.float32) if mean is not None else None
    fn = lib().kfo_norm_backward
    fn.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int]
    fn(_p(dinp), _p(dweight), _p(dbias) if dbias is not None else None, _p(dout), _p(inp), _p(weight), _p(mn) if mn is not None else None, _p(rs), rows, C_)


def attn_backward(q, k, v, o, dO, n_head, hd, n_kv=None):
    """causal attention backward for one sequence.  MHA (n_kv None): q, k, v, o, dO uint16 [T, n_head * hd] -> (dq, dk, dv) of the same shape.
    GQA: k, v [T, n_kv * hd]; every tensor is laid into a common row width internally."""
    n_kv = n_kv or n_head
    T = q.shape[0]
    Cw = n_head * hd
    pad = lambda a: np.ascontiguousarray(np.pad(np.asarray(a, dtype=np.uint16), ((0, 0), (0, Cw - a.shape[1]))))
    qa, ka, va, oa, da = (pad(a) for a in (q, k, v, o, dO))
    dq, dk, dv = (np.zeros((T, Cw), np.uint16) for _ in range(3))
    fn = lib().kfo_attn_backward
    fn.argtypes = [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]
    fn(_p(qa), _p(ka), _p(va), Cw, _p(oa), _p(da), Cw, _p(dq), _p(dk), _p(dv), Cw, T, n_head, n_kv, hd)
    return dq, dk[:, :n_kv * hd].copy(), dv[:, :n_kv * hd].copy()


def embed_backward(dwte, dwpe, dout, tokens, B, T, V):
    """in place on uint16 dwte [V, ldw] (or None) and dwpe [T, C] (or None); dout [B*T, C]"""
    assert dout.dtype == np.uint16 and dout.flags.c_contiguous
    C_ = dout.shape[1]
    tk = np.ascontiguousarray(tokens, dtype=np.int32)
    fn = lib().kfo_embed_backward
    fn.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    fn(_p(dwte) if dwte is not None else None, dwte.shape[1] if dwte is not None else C_, _p(dwpe) if dwpe is not None else None, _p(dout), _p(tk), B, T, C_, V)


def colsum_add(x, dst):
    """dst[c] = bf16(sum_r x[r, c] + dst[c]) in place (uint16 arrays)"""
    assert x.dtype == np.uint16 and dst.dtype == np.uint16 and x.flags.c_contiguous and dst.flags.c_contiguous
    fn = lib().kfo_colsum_add
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    fn(_p(x), _p(dst), x.shape[0], x.shape[1])


def rope_backward(d, nHead, hd, pos, theta):
    """transpose of rope(): gradient row d (uint16 [nHead * hd]) at position pos"""
    out = np.ascontiguousarray(d, dtype=np.uint16).copy()
    c, s_ = rope_table(pos, hd, theta)
    fn = lib().kfo_rope_backward
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn(_p(out), nHead, hd, _p(c), _p(s_))
    return out


def gelu_backward(d, x):
    """returns gelu'(x) * d (uint16 bf16 arrays)"""
    out = np.ascontiguousarray(d, dtype=np.uint16).copy()
    fn = lib().kfo_gelu_backward
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn(_p(out), _p(np.ascontiguousarray(x, dtype=np.uint16)), out.size)
    return out


def swiglu_backward(delta, gate, up):
    """returns (delta_up, delta_gate) of out = silu(gate) * up"""
    d_up = np.ascontiguousarray(delta, dtype=np.uint16).copy()
    d_gate = np.zeros_like(d_up)
    fn = lib().kfo_swiglu_backward
    fn.argtypes = [C.c_void_p] * 4 + [C.c_size_t]
    fn(_p(d_up), _p(d_gate), _p(np.ascontiguousarray(gate, dtype=np.uint16)), _p(np.ascontiguousarray(up, dtype=np.uint16)), d_up.size)
    return d_up, d_gate


def fused_classifier(logits, losses, targets, V, dloss=1.0, mask=None, write_dlogits=True, want_probs=False):
    """fused_classifier on uint16 (bf16) logits [rows, P]: losses (float32, accumulated in place) and the rows overwritten by the logit gradient.
    Returns probs (uint16 [rows, P]) when want_probs."""
    assert logits.dtype == np.uint16 and logits.flags.c_contiguous and logits.ndim == 2
    assert losses.dtype == np.float32 and losses.flags.c_contiguous
    rows, P = logits.shape
    tg = np.ascontiguousarray(targets, dtype=np.int32)
    mk = np.ascontiguousarray(mask, dtype=np.int32) if mask is not None else None
    probs = np.zeros_like(logits) if want_probs else None
    fn = lib().kfo_fused_classifier
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int]
    fn(_p(logits), _p(losses), _p(probs) if probs is not None else None, dloss, _p(tg), rows, V, P, _p(mk) if mk is not None else None, int(write_dlogits))
    return probs


def logf(x):
    fn = lib().kfo_logf_export
    fn.argtypes, fn.restype = [C.c_float], C.c_float
    return np.array([fn(float(v)) for v in np.asarray(x, dtype=np.float32).ravel()], dtype=np.float32)


def gelu(x):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    y = np.zeros_like(x)
    fn = lib().kfo_gelu
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn(_p(x), _p(y), x.size)
    return y


def adamw(params, grads, m, v, lr, beta1, beta2, b1c, b2c, eps, wd, grad_scale, seed):
    """CU_adamw_p in place on uint16 (bf16) params / grads and bf16 (uint16) or float32 m / v arrays. Returns 0, or -1 when a thread met a
    non-finite value (KOIFISH_ADAMW_MV)."""
    for a in (params, grads):
        assert a.dtype == np.uint16 and a.flags.c_contiguous
    assert m.dtype == v.dtype and m.dtype in (np.uint16, np.float32)
    fn = lib().kfo_adamw
    fn.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_int] + [C.c_float] * 8 + [C.c_uint32]
    return int(fn(_p(params), _p(grads), _p(m), _p(v), params.size, int(m.dtype == np.uint16), lr, beta1, beta2, b1c, b2c, eps, wd, grad_scale, seed))


def attn_decode(q, kc, vc, pos, n_head, n_kv, hd, kv_stride=None, mode=ATTN_FUSED):
    q = np.ascontiguousarray(q, dtype=np.uint16)
    kc = np.ascontiguousarray(kc, dtype=np.uint16)
    vc = np.ascontiguousarray(vc, dtype=np.uint16)
    out = np.zeros(n_head * hd, dtype=np.uint16)
    lib().kfo_attn_decode(_p(q), _p(kc), _p(vc), _p(out), pos, n_head, n_kv, hd, kv_stride or n_kv * hd, mode)
    return out


def expf(x):
    return np.array([lib().kfo_expf_export(float(v)) for v in np.asarray(x, dtype=np.float32).reshape(-1)], dtype=np.float32)


# ---------------------------------------------------------------- Qwen3 decoder
SLOTS = ("q", "k", "v", "o", "gate", "up", "down")
NORMS = ("norm_in", "norm_post", "qn", "kn")


class Qwen3Oracle:
    """weights: dict with 'embed', 'head' (QWeight; may be the same object when tied), 'final_norm' (uint16[dim]),
    'layers': list of dicts with QWeight q,k,v,o,gate,up,down and uint16 norm_in,norm_post,qn,kn."""

    def __init__(self, cfg, weights, attn_mode=ATTN_FUSED, tp=1):
        self.cfg, self.w = cfg, weights
        L = lib()
        self.h = L.kfo_qwen3_new(cfg["dim"], cfg["n_layer"], cfg["n_head"], cfg["n_kv"], cfg["head_dim"], cfg["ffn"], cfg["vocab"],
                                 cfg["max_seq"], cfg.get("rms_eps", 1e-6), cfg.get("qk_eps", 1e-6), cfg.get("theta", 1e6), attn_mode)
        self.h = C.c_void_p(self.h)
        L.kfo_qwen3_set_tp(self.h, tp)
        self._keep = []

        def setw(layer, slot, w):
            self._keep.append(w)
            r = L.kfo_qwen3_set_weight(self.h, layer, slot, w.type, w.ne0, w.ne1, _p(w.data), _p(w.zero), _p(w.step), w.lGroup, w.qBias)
            assert r == 0

        setw(-1, 0, weights["embed"])
        setw(-1, 1, weights["head"])
        fn = np.ascontiguousarray(weights["final_norm"], dtype=np.uint16)
        self._keep.append(fn)
        L.kfo_qwen3_set_norm(self.h, -1, 0, _p(fn))
        for li, lw in enumerate(weights["layers"]):
            for si, s in enumerate(SLOTS):
                setw(li, si, lw[s])
            for si, s in enumerate(NORMS):
                if lw.get(s) is not None:
                    a = np.ascontiguousarray(lw[s], dtype=np.uint16)
                    self._keep.append(a)
                    L.kfo_qwen3_set_norm(self.h, li, si, _p(a))

    def prepare_fast(self):
        """16-bit weight views for the vectorised (AVX2 / F16C) mat-vec: same bits as the scalar path, an order of magnitude faster (the timed CPU
        baseline).  Returns the bytes allocated (quantised tensors are dequantised once into bf16 copies), or a negative value when not applicable."""
        lib().kfo_qwen3_prepare_fast.restype = C.c_longlong
        return int(lib().kfo_qwen3_prepare_fast(self.h))

    def set_hot(self, layer, hot):
        """sparse forward: hot[ffn] int32, 1 = the FFN row is computed (D_matmul_sparse); None = dense"""
        if not hasattr(self, "_hot"):
            self._hot = {}
        if hot is None:
            self._hot.pop(layer, None)
            lib().kfo_qwen3_set_hot(self.h, int(layer), None)
            return
        a = np.ascontiguousarray(hot, dtype=np.int32)
        self._hot[layer] = a   # the C side keeps the pointer
        lib().kfo_qwen3_set_hot(self.h, int(layer), _p(a))

    def decode(self, token, pos, want_logits=True, want_hidden=False):
        logits = np.zeros(self.cfg["vocab"], dtype=np.uint16) if want_logits else None
        hidden = np.zeros(self.cfg["dim"], dtype=np.uint16) if want_hidden else None
        nxt = lib().kfo_qwen3_decode(self.h, int(token), int(pos), _p(logits), _p(hidden))
        return int(nxt), logits, hidden

    def kv(self):
        c = self.cfg
        n = c["n_layer"] * c["max_seq"] * c["n_kv"] * c["head_dim"]
        k = np.ctypeslib.as_array(C.cast(lib().kfo_qwen3_kcache(self.h), C.POINTER(C.c_uint16)), shape=(n,))
        v = np.ctypeslib.as_array(C.cast(lib().kfo_qwen3_vcache(self.h), C.POINTER(C.c_uint16)), shape=(n,))
        shp = (c["n_layer"], c["max_seq"], c["n_kv"] * c["head_dim"])
        return k.reshape(shp), v.reshape(shp)

    def generate(self, prompt, n_new, sampler=None):
        """Token-serial prefill (Fish::Chat, GoPT.cpp:1139-1146) then decode. Returns new ids.
        sampler = None: greedy; else dict(top_k, temperature, top_p, seed) -> GeneratOnPrompt::Sample on every step's logits
        (the coin is drawn once per sampled token, starting with the token that follows the prompt)."""
        greedy = sampler is None or sampler["temperature"] == 0.0 or sampler["top_k"] == 1
        rng = None if greedy else np.array([sampler["seed"]], dtype=np.uint64)

        def pick(nxt, logits):
            return nxt if greedy else sample(logits, sampler["top_k"], sampler["temperature"], sampler["top_p"], rng)
        pos, nxt = 0, None
        for i, t in enumerate(prompt):
            last = i == len(prompt) - 1
            nxt, lg, _ = self.decode(t, pos, want_logits=(last and not greedy))
            if last:
                nxt = pick(nxt, lg)
            pos += 1
        out = []
        for _ in range(n_new):
            out.append(nxt)
            nxt, lg, _ = self.decode(nxt, pos, want_logits=not greedy)
            nxt = pick(nxt, lg)
            pos += 1
        return out

    def close(self):
        if self.h:
            lib().kfo_qwen3_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def from_device_model(m, attn_mode=ATTN_FUSED):
    """Builds the CPU decoder from a koifish_amd.runtime.Qwen3 (duck-typed: .cfg, .weights[(layer, slot)] with .blob/.szData/
    .type/.ne0/.ne1/.nGroup/.lGroup/.qBias, ._norms[(layer, slot)]): copies the SAME packed weights D2H, so GPU and CPU decode
    one model.  Used by tests and by bench.py's cpu_baseline leg."""
    import torch

    def host_weight(w):
        blob = w.blob.cpu().numpy()
        data = blob[:w.szData]
        if getattr(w, "lGroup", 128) == 0:  # LutDevWeight: nibble stream || bf16 [R][C][LUT ne0 x 16]
            g = blob[w.szData:].view(np.uint16)
            return LutWeight(w.ne0, w.ne1, data, g[w.ne0 + w.ne1:].copy())
        if w.type == BF16:
            return QWeight(BF16, w.ne0, w.ne1, data.view(np.uint16))
        if w.type == F8E5M2:
            return QWeight(F8E5M2, w.ne0, w.ne1, data)
        g = blob[w.szData:].view(np.uint16)
        z0 = w.ne0 + w.ne1
        return QWeight(w.type, w.ne0, w.ne1, data, g[z0:z0 + w.nGroup].copy(), g[z0 + w.nGroup:z0 + 2 * w.nGroup].copy(), w.lGroup, w.qBias)

    def host_norm(t):
        return t.view(torch.int16).cpu().numpy().view(np.uint16)

    cfg = m.cfg
    ow = {"embed": host_weight(m.weights[(-1, 0)]), "final_norm": host_norm(m._norms[(-1, 0)]), "layers": []}
    ow["head"] = ow["embed"] if m.weights[(-1, 1)] is m.weights[(-1, 0)] else host_weight(m.weights[(-1, 1)])
    for li in range(cfg["n_layer"]):
        d = {s: host_weight(m.weights[(li, si)]) for si, s in enumerate(SLOTS)}
        for si, s in enumerate(NORMS):
            d[s] = host_norm(m._norms[(li, si)])
        ow["layers"].append(d)
    return Qwen3Oracle(cfg, ow, attn_mode=attn_mode)


def set_num_threads(n):
    lib().kfo_set_num_threads(int(n))


def bench_matvec(M, K, reps=10):
    """seconds per M x K 16-bit mat-vec at the current thread count (first-touch-local pages)"""
    lib().kfo_bench_matvec.restype = C.c_double
    return float(lib().kfo_bench_matvec(int(M), int(K), int(reps)))


def num_threads():
    return int(lib().kfo_num_threads())
