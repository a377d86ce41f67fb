"""Per-stage HIP-event times of FusedEmbedder on one micro-batch (978 cubes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speaker_verification_amd.model import seeded_model, _FLAT
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 978
model = seeded_model(1).to(dev).eval()
emb = model.fused_inference(channels_last=True)
x0 = torch.randn(n, 1, 20, 80, 40, device=dev)
names = ["1_1", "1_2", "2_1", "2_2", "3_1", "3_2", "4_1", "4_2"]

def run(timed):
    ev = []
    def mark(label):
        if timed:
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((label, e))
    with torch.no_grad():
        x = x0.contiguous(memory_format=torch.channels_last_3d)
        mark("start")
        fold = None
        for li, (w, b, slope, stride, pool, pool_first) in enumerate(emb.stages):
            groups = 1
            if li == 1 and emb.row_fold is not None:
                fold = emb.row_fold
            if fold is not None and li in (1, 2, 3):
                w, b, slope, stride, groups = fold[li]
            if li == 0:
                nb, _, d, h, wd = x.shape
                kd, kw = w.shape[2], w.shape[4]
                od, ow = d - kd + 1, wd - kw + 1
                G, wt, bt = emb._first_layer_tables(ow)
                xs = x.reshape(nb, d, h, wd)
                windows = xs.as_strided((nb, od, h, ow // G, kd, kw + G - 1), (d * h * wd, h * wd, wd, G, h * wd, 1))
                x = torch.addmm(bt, windows.reshape(nb * od * h * (ow // G), kd * (kw + G - 1)), wt)
                x = x.view(nb, od, h, ow, w.shape[0]).permute(0, 4, 1, 2, 3)
            else:
                x = F.conv3d(x, w, b, stride=stride, groups=groups)
            mark("conv" + names[li])
            if pool_first:
                w2 = x.shape[-1] // 2 * 2
                x = F.prelu(torch.maximum(x[..., 0:w2:2], x[..., 1:w2:2]), slope)
            else:
                x = F.prelu(x, slope)
            mark("act" + names[li])
        y = F.linear(x.reshape(x.shape[0], _FLAT), emb.fc_w, emb.fc_b)
        mark("fc")
    return ev
for _ in range(3):
    run(False)
torch.cuda.synchronize()
ev = run(True)
torch.cuda.synchronize()
tot = 0.0
for (l0, e0), (l1, e1) in zip(ev[:-1], ev[1:]):
    t = e0.elapsed_time(e1); tot += t
    print(f"{l1:10s} {t:7.3f} ms")
print(f"total      {tot:7.3f} ms")
