import sys, time, torch
sys.path.insert(0, '.')
from basic_pitch_amd.inference import Model
B = 256
dev = torch.device('cuda', 0)
audio = (torch.rand((B, 43844), device=dev) * 2 - 1).contiguous()
out = {"note": torch.empty((B,172,88), device=dev), "onset": torch.empty((B,172,88), device=dev), "contour": torch.empty((B,172,264), device=dev)}
for timing in (True, False, True, False):
    m = Model(device=0, max_windows=B, stage_timing=timing)
    for _ in range(5): m._predict_device(audio, out=out, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): m._predict_device(audio, out=out, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("timing", timing, "ms/step", round(dt * 1e3, 4), "windows/s", round(B / dt))
    m.close()
