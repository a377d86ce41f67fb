"""conv1_2 with R output rows per position folded into channels (R = 1 reference, 2 shipped, 3, 4), and
conv2_1 on the folded layout as a 2-group conv against a dilated conv over the (column, parity) axis."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
CL = torch.channels_last_3d
n = 978
x = torch.randn(n, 16, 18, 80, 36, device=dev).contiguous(memory_format=CL)
w = torch.randn(16, 16, 3, 9, 1, device=dev) * 0.05

def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]

for R in (1, 2, 3, 4):
    kh = 9 + 2 * (R - 1)
    wf = torch.zeros(R, 16, 16, 3, kh, 1, device=dev)
    for p in range(R):
        wf[p, :, :, :, 2 * p:2 * p + 9, :] = w
    wf = wf.reshape(R * 16, 16, 3, kh, 1).contiguous(memory_format=CL)
    h_out = (80 - kh) // (2 * R) + 1
    t = timeit(lambda: F.conv3d(x, wf, None, stride=(1, 2 * R, 1)))
    print(f"conv1_2 rows/position {R}: kernel (3,{kh},1) stride (1,{2*R},1) -> {R*16} ch x {h_out} rows ({h_out*R} of 36)  {t:.3f} ms", flush=True)

a = torch.randn(n, 32, 16, 18, 18, device=dev).contiguous(memory_format=CL)     # (parity, c) channels
w21 = torch.randn(32, 16, 3, 1, 4, device=dev) * 0.05
w21g = w21.repeat(2, 1, 1, 1, 1).contiguous(memory_format=CL)
t_g = timeit(lambda: F.conv3d(a, w21g, None, groups=2))
# same bytes viewed as 16 channels over W' = (w, parity): memory (n, d, hp, w, parity, c)
a_view = a.permute(0, 2, 3, 4, 1).reshape(n, 16, 18, 18, 2, 16).permute(0, 5, 1, 2, 3, 4).reshape(n, 16, 16, 18, 36)
print("view is channels_last:", a_view.is_contiguous(memory_format=CL), a_view.stride())
w21c = w21.contiguous(memory_format=CL)
t_d = timeit(lambda: F.conv3d(a_view, w21c, None, dilation=(1, 1, 2)))
print(f"conv2_1: 2-group {t_g:.3f} ms   dilated over (w, parity) {t_d:.3f} ms")
