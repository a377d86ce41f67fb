import sys, os, torch
sys.path.insert(0, os.getcwd())
from speaker_verification_amd.engine import get_engine
from speaker_verification_amd.model import seeded_model
eng = get_engine(0)
emb = seeded_model(1, n_labels=4).to(eng.device).eval().fused_inference(channels_last=True)
t = emb.conv32_tables()
x = torch.randn((1024, 10, 15, 5, 64), device=eng.device)
for _ in range(10): eng.c3d2_conv32(x, t)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): eng.c3d2_conv32(x, t)
b.record(); torch.cuda.synchronize()
print("svk_c3d2_conv32, 1024 cubes: %.3f ms" % (a.elapsed_time(b) / 20))
