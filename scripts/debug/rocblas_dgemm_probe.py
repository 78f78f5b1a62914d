"""What the vendor DGEMM reaches on this part, for the shapes the dense solver runs (scripts/debug: a yardstick for the hand-written
FP64-MFMA tiles of esl_chol.hpp, not part of the product -- the library never calls rocBLAS)."""
import torch, time, sys
dev = torch.device("cuda:0")
PEAK = 78.6
def run(M, N, K, reps=5, trans=False):
    A = torch.randn(M, K, dtype=torch.float64, device=dev)
    B = torch.randn(N, K, dtype=torch.float64, device=dev) if trans else torch.randn(K, N, dtype=torch.float64, device=dev)
    C = torch.randn(M, N, dtype=torch.float64, device=dev)
    f = (lambda: torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)) if trans else (lambda: torch.addmm(C, A, B, beta=1.0, alpha=-1.0, out=C))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"M={M:6d} N={N:6d} K={K:5d} {'NT' if trans else 'NN'}  {ms:9.3f} ms  {tf:6.2f} TF  ({100*tf/PEAK:5.1f} % of {PEAK})", flush=True)
for trans in (False, True):
    run(8192, 8192, 512, trans=trans)
    run(8192, 8192, 1536, trans=trans)
    run(8192, 8192, 3744, trans=trans)
    run(16384, 16384, 512, trans=trans)
    run(16384, 16384, 4096, trans=trans)
    run(18000, 18000, 3744, reps=3, trans=trans)
