"""Per-step device time of the headline batch right after start-up (HIP events on the stream the model runs on): where in
the driver's burst (5 warm-up + 20 timed steps) does the step reach its sustained time?"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from basic_pitch_amd.inference import Model

B = 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
audio = (torch.rand((B, 43844), device=dev, generator=g) * 2 - 1) * 0.5
out = {"note": torch.empty((B, 172, 88), device=dev), "onset": torch.empty((B, 172, 88), device=dev),
       "contour": torch.empty((B, 172, 264), device=dev)}
torch.cuda.synchronize()
s = torch.cuda.Stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for trial in range(2):
    model = Model(device=0, max_windows=B)
    with torch.cuda.stream(s):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record(s)
        for i in range(n):
            model._predict_device(audio, out=out, sync=False)
            ev[i + 1].record(s)
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("steps 0-4 %.4f | 5-24 %.4f | 25-49 %.4f | 50-99 %.4f | 100- %.4f" % (
        np.mean(ms[:5]), np.mean(ms[5:25]), np.mean(ms[25:50]), np.mean(ms[50:100]), np.mean(ms[100:])))
    print(" ".join("%.3f" % v for v in ms[:60]))
    model.close()
