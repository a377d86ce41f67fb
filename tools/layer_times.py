"""Per-op HIP-event times of FusedEmbedder on one micro-batch (978 cubes), measured on the embedder's own
code path: its F.conv3d / F.prelu / torch.maximum / torch.addmm / F.linear calls are wrapped to drop an
event after each op."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_verification_amd import model as M
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 978
emb = M.seeded_model(1).to(dev).eval().fused_inference(channels_last=True)
x0 = torch.randn(n, 1, 20, 80, 40, device=dev)
events, recording = [], [False]

def wrap(fn, label):
    def inner(*a, **k):
        out = fn(*a, **k)
        if recording[0]:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            shape = tuple(out.shape)
            events.append((label, shape, e))
        return out
    return inner

M.F.conv3d = wrap(M.F.conv3d, "conv3d")
M.F.prelu = wrap(M.F.prelu, "prelu")
M.F.linear = wrap(M.F.linear, "linear")
M.torch.addmm = wrap(M.torch.addmm, "addmm (conv1_1)")
M.torch.maximum = wrap(M.torch.maximum, "maximum (pool)")
for _ in range(3):
    emb(x0)
torch.cuda.synchronize()
recording[0] = True
start = torch.cuda.Event(enable_timing=True)
start.record()
emb(x0)
torch.cuda.synchronize()
prev, total = start, 0.0
for label, shape, e in events:
    t = prev.elapsed_time(e)
    total += t
    print(f"{label:18s} -> {str(shape):28s} {t:7.3f} ms")
    prev = e
print(f"total {total:.3f} ms  (the first interval includes the strided gather of the cube when called on a cube)")
