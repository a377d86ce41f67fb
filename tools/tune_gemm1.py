"""First-layer GEMM (patch matrix [4.2 M, 48] x Toeplitz weights [48, 192] + bias) with PyTorch's default
hipBLASLt heuristic against TunableOp's pick."""
import time
import torch
dev = torch.device("cuda:0")
M, K, N = 978 * 18 * 80 * 3, 48, 192
a = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev)
b = torch.randn(N, device=dev)
def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
print("default  %.3f ms" % timeit(lambda: torch.addmm(b, a, w)), flush=True)
print("mm+add   %.3f ms" % timeit(lambda: torch.mm(a, w).add_(b)), flush=True)
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_filename("/tmp/tunableop_results.csv")
t0 = time.time()
torch.addmm(b, a, w); torch.cuda.synchronize()
print("tuning took %.1f s" % (time.time() - t0), flush=True)
print("tunable  %.3f ms" % timeit(lambda: torch.addmm(b, a, w)), flush=True)
