"""conv1_2 .. conv2_2 in two formulations: the reference's, and "H folded into channels": conv1_2 as a
Toeplitz-widened conv producing two output rows per position (32 channels = (h parity, co)), conv2_1 as
a 2-group conv (it does not mix rows), conv2_2 (8 taps, stride 2 over rows) as a 4-tap stride-1 conv over
row pairs with 64 input channels.  Pure re-indexing: same sums, different order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
CL = torch.channels_last_3d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 978
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s, scale=0.05):
    return (torch.randn(*s, generator=g) * scale).to(dev)
x1 = rnd(n, 16, 18, 80, 36, scale=1.0).contiguous(memory_format=CL)
w12, b12, s12 = rnd(16, 16, 3, 9, 1), rnd(16), torch.rand(16, generator=g).to(dev) * 0.5
w21, b21, s21 = rnd(32, 16, 3, 1, 4), rnd(32), torch.rand(32, generator=g).to(dev) * 0.5
w22, b22, s22 = rnd(32, 32, 3, 8, 1), rnd(32), torch.rand(32, generator=g).to(dev) * 0.5

def prelu(x, s):
    return torch.where(x >= 0, x, x * s.view(1, -1, 1, 1, 1))
def pool_w(x):
    w2 = x.shape[-1] // 2 * 2
    return torch.maximum(x[..., 0:w2:2], x[..., 1:w2:2])

w12c, w21c, w22c = (t.contiguous(memory_format=CL) for t in (w12, w21, w22))
def chain_ref():
    a = prelu(pool_w(F.conv3d(x1, w12c, b12, stride=(1, 2, 1))), s12)
    b = prelu(F.conv3d(a, w21c, b21), s21)
    c = prelu(pool_w(F.conv3d(b, w22c, b22, stride=(1, 2, 1))), s22)
    return c

w12f = torch.zeros(2, 16, 16, 3, 11, 1, device=dev)
for hs in range(2):
    w12f[hs, :, :, :, 2 * hs:2 * hs + 9, :] = w12
w12f = w12f.reshape(32, 16, 3, 11, 1).contiguous(memory_format=CL)
b12f, s12f = b12.repeat(2), s12.repeat(2)
w21g = w21.repeat(2, 1, 1, 1, 1).contiguous(memory_format=CL)          # [64, 16, 3, 1, 4], groups = 2
b21g, s21g = b21.repeat(2), s21.repeat(2)
# w22f[co, hs * 32 + ci, kd, khp] = w22[co, ci, kd, 2 khp + hs]
w22f = w22.reshape(32, 32, 3, 4, 2, 1).permute(0, 4, 1, 2, 3, 5).reshape(32, 64, 3, 4, 1).contiguous(memory_format=CL)
def chain_fold():
    a = prelu(pool_w(F.conv3d(x1, w12f, b12f, stride=(1, 4, 1))), s12f)
    b = prelu(F.conv3d(a, w21g, b21g, groups=2), s21g)
    c = prelu(pool_w(F.conv3d(b, w22f, b22)), s22)
    return c

def timeit(fn, reps=5):
    for _ in range(2):
        y = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return sorted(ts)[len(ts) // 2], y
t1, y1 = timeit(chain_ref)
t2, y2 = timeit(chain_fold)
print("reference chain %.3f ms   folded chain %.3f ms   max diff %.2e (scale %.2f)" % (t1, t2, (y1 - y2).abs().max().item(), y1.abs().max().item()))
# layer by layer
a_r = prelu(pool_w(F.conv3d(x1, w12c, b12, stride=(1, 2, 1))), s12)
a_f = prelu(pool_w(F.conv3d(x1, w12f, b12f, stride=(1, 4, 1))), s12f)
b_r = prelu(F.conv3d(a_r, w21c, b21), s21)
b_f = prelu(F.conv3d(a_f, w21g, b21g, groups=2), s21g)
for name, f_r, f_f in [("conv1_2", lambda: F.conv3d(x1, w12c, b12, stride=(1, 2, 1)), lambda: F.conv3d(x1, w12f, b12f, stride=(1, 4, 1))),
                       ("conv2_1", lambda: F.conv3d(a_r, w21c, b21), lambda: F.conv3d(a_f, w21g, b21g, groups=2)),
                       ("conv2_2", lambda: F.conv3d(b_r, w22c, b22, stride=(1, 2, 1)), lambda: F.conv3d(b_f, w22f, b22))]:
    tr, _ = timeit(f_r)
    tf, _ = timeit(f_f)
    print("%s: reference %.3f ms   folded %.3f ms" % (name, tr, tf))
