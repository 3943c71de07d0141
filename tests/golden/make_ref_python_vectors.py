"""Golden vectors from the reference's OWN Python statements of three operators of the path, executed here (this container only; the GPU box has no
/root/reference): the function objects are taken out of the reference's files with `ast` at generation time and run on seeded inputs; only inputs and
outputs are stored (tests/golden/ref_python_vectors.npz).  Nothing of the reference's text is kept.

  src/Python/test_awq.py:32-66, 103-128   unpack_awq / reverse_awq_order / Dequant_1   (AutoAWQ GEMM layout -> bf16 weights)
  src/Python/tile_wrapper/tl_qkv.py:384-403   ref_program(Q, K, V, is_causal, groups)    (causal grouped-query attention; row a13)
  src/Python/tile_wrapper/tl_norm.py:62-63     ref_program(x)                              (RMS normalisation without a weight; row a9)
  src/Python/tile_wrapper/tl_gemm.py:150-151   ref_program(A, B) = A @ B.T                 (the token-batch product SLP::Forw hands to cuBLASLt: x [n, K] . W [M, K]^T; row a8)

    python tests/golden/make_ref_python_vectors.py        (needs /root/reference)
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/src/Python"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_python_vectors.npz")


def take(path, names, extra=None):
    """the named top-level functions / assignments of a reference file, compiled into a fresh namespace (the file's own imports -- awq, tilelang --
    are not installed here and are not needed by these functions)"""
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in names) or
            (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in n.targets))]
    assert {getattr(n, "name", None) or n.targets[0].id for n in keep} == set(names), "reference file changed"
    ns = {"torch": torch, "F": F, "np": np}
    ns.update(extra or {})
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def take_def(path, name):
    """one top-level function of a reference file that this interpreter cannot parse as a whole (tl_gemm.py uses 3.12 f-string syntax further down): the lines from its
    `def` to the next top-level statement, compiled alone"""
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith("def %s(" % name))
    i1 = next(i for i in range(i0 + 1, len(lines)) if lines[i] and not lines[i][0].isspace())
    ns = {"torch": torch, "F": F, "np": np}
    exec(compile("\n".join(lines[i0:i1]), path, "exec"), ns)
    return ns


def bf16_bits(t):
    return t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def main():
    out = {}
    g = torch.Generator().manual_seed(20260301)
    # ---- AutoAWQ: qweight [in, out/8] int32, qzeros [in/128, out/8] int32, scales [in/128, out] fp16 -> dequantised [in, out] bf16
    awq = take(os.path.join(REF, "test_awq.py"), {"unpack_awq", "reverse_awq_order", "Dequant_1", "AWQ_ORDER", "AWQ_REVERSE_ORDER"},
               {"save_dequantized_to_csv": lambda *a, **k: None, "print": lambda *a, **k: None})
    n_in, n_out = 256, 64
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_in, n_out // 8), generator=g, dtype=torch.int32)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_in // 128, n_out // 8), generator=g, dtype=torch.int32)
    scales = (torch.rand(n_in // 128, n_out, generator=g) * 0.01 + 0.002).to(torch.float16)
    iw, iz = awq["unpack_awq"](qweight, qzeros, 4)
    iw, iz = torch.bitwise_and(iw, 15), torch.bitwise_and(iz, 15)
    iw_r, iz_r = awq["reverse_awq_order"](iw, iz, 4)
    deq = awq["Dequant_1"]("x.w", qweight, scales, qzeros, 128, 4)
    out.update(awq_qweight=qweight.numpy(), awq_qzeros=qzeros.numpy(), awq_scales=scales.view(torch.int16).numpy().view(np.uint16),
               awq_iweight=iw_r.numpy().astype(np.uint8), awq_izeros=iz_r.numpy().astype(np.uint8), awq_dequant=bf16_bits(deq))
    # ---- attention: Q [B, T, HQ, D], K / V [B, T, HK, D], causal, groups = HQ / HK; inputs are bf16-exact values, the reference computes in fp32
    att = take(os.path.join(REF, "tile_wrapper", "tl_qkv.py"), {"ref_program"})
    for tag, (T, HQ, HK, D) in (("a", (48, 4, 2, 64)), ("b", (33, 8, 1, 128)), ("c", (20, 2, 2, 64))):
        Q = torch.randn(1, T, HQ, D, generator=g).to(torch.bfloat16).float()
        K = torch.randn(1, T, HK, D, generator=g).to(torch.bfloat16).float()
        V = torch.randn(1, T, HK, D, generator=g).to(torch.bfloat16).float()
        O = att["ref_program"](Q, K, V, True, HQ // HK)
        out.update({"att_%s_q" % tag: bf16_bits(Q[0]), "att_%s_k" % tag: bf16_bits(K[0]), "att_%s_v" % tag: bf16_bits(V[0]), "att_%s_out" % tag: O[0].numpy().astype(np.float32)})
    # ---- RMS normalisation (eps 1e-12, no weight)
    nrm = take(os.path.join(REF, "tile_wrapper", "tl_norm.py"), {"ref_program"})
    x = (torch.randn(7, 1024, generator=g) * 3.0).to(torch.bfloat16).float()
    out.update(rms_x=bf16_bits(x), rms_out=nrm["ref_program"](x).numpy().astype(np.float32))
    # ---- the product y = x . W^T on bf16-exact operands (one token row, and a token batch), fp32 result; drawn AFTER everything above: the earlier vectors keep their bits
    gm = take_def(os.path.join(REF, "tile_wrapper", "tl_gemm.py"), "ref_program")
    for tag, (n, M, K) in (("a", (1, 96, 256)), ("b", (40, 64, 384))):
        A = torch.randn(n, K, generator=g).to(torch.bfloat16).float()
        B = (torch.randn(M, K, generator=g) * 0.05).to(torch.bfloat16).float()
        out.update({"gemm_%s_x" % tag: bf16_bits(A), "gemm_%s_w" % tag: bf16_bits(B), "gemm_%s_out" % tag: gm["ref_program"](A, B).numpy().astype(np.float32)})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
