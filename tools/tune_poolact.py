"""Pool (max over column pairs) + PReLU after conv1_2 on the folded channels-last tensor: PyTorch formulations."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
x = torch.randn(978, 32, 16, 18, 36, device=dev).contiguous(memory_format=torch.channels_last_3d)
slope = torch.tensor([0.25], device=dev)
def timeit(fn, reps=10):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2], y
t0, y0 = timeit(lambda: F.prelu(torch.maximum(x[..., 0::2], x[..., 1::2]), slope))
print(f"maximum + prelu            {t0:.3f} ms")
t1, y1 = timeit(lambda: F.prelu(torch.amax(x.unflatten(-1, (18, 2)), dim=-1), slope))
print(f"amax(unflatten) + prelu    {t1:.3f} ms  equal {torch.equal(y0, y1)}")
t2, y2 = timeit(lambda: F.leaky_relu_(torch.maximum(x[..., 0::2], x[..., 1::2]), 0.25))
print(f"maximum + leaky_relu_      {t2:.3f} ms  equal {torch.equal(y0, y2)}")
t3, y3 = timeit(lambda: F.prelu(F.max_pool3d(x, (1, 1, 2), (1, 1, 2)), slope))
print(f"max_pool3d + prelu         {t3:.3f} ms  equal {torch.equal(y0, y3)}")
def one_pass():
    m = torch.maximum(x[..., 0::2], x[..., 1::2])
    return torch.maximum(m, m * 0.25)
t4, y4 = timeit(one_pass)
print(f"maximum, maximum(m, s m)   {t4:.3f} ms  max diff {(y0 - y4).abs().max().item():.1e}")
big = torch.randn(978, 16, 18, 80, 36, device=dev).contiguous(memory_format=torch.channels_last_3d)
t5, _ = timeit(lambda: F.prelu(big, slope))
t6, _ = timeit(lambda: F.leaky_relu(big, 0.25))
bb = big.clone()
t7, _ = timeit(lambda: F.leaky_relu_(bb, 0.25))
print(f"first activation (3.2 GB): prelu {t5:.3f}  leaky_relu {t6:.3f}  leaky_relu_ (in place) {t7:.3f} ms")
