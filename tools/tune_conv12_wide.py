"""conv1_2 (16 -> 16, k(3,9,1), s(1,2,1)) against a Toeplitz-widened equivalent that produces two
output rows per position as 32 channels (k(3,11,1), s(1,4,1)): more MACs (x1.22) but a wider GEMM N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
n = 978
x = torch.randn(n, 16, 18, 80, 36, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(16, 16, 3, 9, 1, device=dev) * 0.05
b = torch.randn(16, device=dev)
w2 = torch.zeros(2, 16, 16, 3, 11, 1, device=dev)
for hs in range(2):
    w2[hs, :, :, :, 2 * hs:2 * hs + 9, :] = w
w2 = w2.reshape(32, 16, 3, 11, 1).contiguous(memory_format=torch.channels_last_3d)
b2 = b.repeat(2)
wc = w.contiguous(memory_format=torch.channels_last_3d)

def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2], y

t1, y1 = timeit(lambda: F.conv3d(x, wc, b, stride=(1, 2, 1)))
t2, y2 = timeit(lambda: F.conv3d(x, w2, b2, stride=(1, 4, 1)))
# y2 [n, 32, 16, 18, 36] -> (hs, co) channels -> rows h = 2 hp + hs
y2v = y2.reshape(n, 2, 16, 16, 18, 36).permute(0, 2, 3, 4, 1, 5).reshape(n, 16, 16, 36, 36)
print("orig %.3f ms   wide %.3f ms   max diff %.2e" % (t1, t2, (y1 - y2v).abs().max().item()))
# with the pool (max over W pairs) absorbing the permutation
def pool_orig():
    y = F.conv3d(x, wc, b, stride=(1, 2, 1))
    return torch.maximum(y[..., 0::2], y[..., 1::2])
def pool_wide():
    y = F.conv3d(x, w2, b2, stride=(1, 4, 1))
    v = y.reshape(n, 2, 16, 16, 18, 36).permute(0, 2, 3, 4, 1, 5)        # [n, co, d, hp, hs, w]
    m = torch.maximum(v[..., 0::2], v[..., 1::2])                          # [n, co, d, hp, hs, 18]
    return m.reshape(n, 16, 16, 36, 18).contiguous(memory_format=torch.channels_last_3d)
t3, p1 = timeit(pool_orig)
t4, p2 = timeit(pool_wide)
print("orig+pool %.3f ms   wide+pool %.3f ms   max diff %.2e" % (t3, t4, (p1 - p2).abs().max().item()))
