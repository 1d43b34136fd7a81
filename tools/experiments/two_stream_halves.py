"""Does splitting a 256-window batch into two halves on two streams (two handles, intermediates of their own) beat one
handle on one stream?  The halves' kernels can overlap (one half's HBM / latency-bound CQT kernels under the other's
matrix kernels).  Prints windows/s for: 1 x 256, 2 x 128 on two streams, 2 x 256 on two streams, 4 x 64."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from basic_pitch_amd import Model

def rate(parts, per, seconds=1.5):
    models = [Model(max_windows=per) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = [(torch.rand(per, 43844, device="cuda") * 2 - 1).float() for _ in range(parts)]
    outs = [m._predict_device(x) for m, x in zip(models, xs)]
    torch.cuda.synchronize()
    def step():
        for m, s, x, o in zip(models, streams, xs, outs):
            with torch.cuda.stream(s):
                m._predict_device(x, out=o, sync=False)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    for m in models:
        m.close()
    return n * parts * per / dt

for parts, per in ((1, 256), (2, 128), (2, 256), (4, 64), (1, 512)):
    print(f"{parts} x {per}: {rate(parts, per):,.0f} windows/s", flush=True)
