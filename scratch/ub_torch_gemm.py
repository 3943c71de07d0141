"""What the vendor library reaches on the config-3 GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, fp32 accumulate): the yardstick for
the hand-written token-batch kernels."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
n = 8192
for M, K in ((4800, 1600), (1600, 1600), (6400, 1600), (1600, 6400), (50304, 1600)):
    x = torch.randn(n, K, device=dev, dtype=torch.bfloat16); w = torch.randn(M, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(n, M, device=dev, dtype=torch.bfloat16)
    f = t(lambda: x @ w.t()); bx = t(lambda: dy @ w); bw = t(lambda: dy.t() @ x)
    fl = 2.0 * n * M * K
    print("M %5d K %5d: fwd %.3f ms %.0f TF | dX %.3f ms %.0f TF | dW %.3f ms %.0f TF" % (M, K, f, fl / f / 1e9, bx, fl / bx / 1e9, bw, fl / bw / 1e9))
