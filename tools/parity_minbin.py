# Min-bin diagnostic of the CQT front ends (tools/; round-3 review, parity item 1).  The per-window MINIMUM of the log-power
# map (basic_pitch/layers/signal.py:177) shifts every output of the window, and it sits on the weakest bin, where the CQT's
# rounding error is largest relative to the value: an fp32-class evaluation of the graph is as good as its log-power at
# that bin.  For bench.py's 256 uniform[-1, 1) windows and 256 normal(0, 0.01) windows, per window and evaluation:
#   e_at   = lp_eval[t*, b*] - lp_fp64[t*, b*]   at the fp64 oracle's arg-min bin (t*, b*)
#   e_min  = min(lp_eval) - min(lp_fp64)         what the normalisation actually subtracts
# for the default path (planes CQT, split-f16 on 16x16x32), the exact-f32 A/B path (cqt_pyramid / cqt_filterbank on f32
# MFMA) — both through the C ABI stage hook: pyramid then filterbank — and the torch fp32 oracle.  Summary: median / p90 /
# p99 / max of |e| and the two-sample Kolmogorov-Smirnov distance of each HIP sample to the fp32 oracle's.
#   python tools/parity_minbin.py [n]  ->  markdown on stdout
import os, sys, time
import numpy as np
import torch
from scipy.stats import ks_2samp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_windows
from oracle import bp_oracle as O
from basic_pitch_amd import Model
from stage_harness import StageRunner, pyr_pack, pyr_unpack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = O.load_weights()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
xb = (torch.rand((256, 43844), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()[:n].cpu().numpy()
fam = {"bench uniform (torch seed 1234)": xb, "normal sigma 0.01 (numpy seed 1)": make_windows("normal", n, seed=1)}
F32 = torch.float32

def hip_lp(runner, x):
    out = []
    for i in range(0, len(x), 32):
        a = x[i:i + 32]; m = len(a)
        pyr = runner.run("pyramid", m, {"audio": a}, {"pyr": ((m, 43712), F32)})["pyr"]
        lp = runner.run("filterbank", m, {"audio": a, "pyr": pyr}, {"lp": ((m, 172, 309), F32), "mm": ((m, 2), torch.int32)})["lp"]
        out.append(lp)
    return np.concatenate(out)

def oracle_lp(x, dt):
    out = []
    for i in range(0, len(x), 32):
        out.append(np.asarray(O.forward(x[i:i + 32], W, dt, intermediates=True)["lp"], dtype=np.float64))
    return np.concatenate(out)

runners = {"default path (planes CQT, split-f16)": StageRunner(Model(max_windows=32)),
           "exact-f32 A/B path": StageRunner(Model(max_windows=32, exact_f32_mfma=True))}
def q(v): return f"{np.median(v):.2e} | {np.quantile(v, 0.9):.2e} | {np.quantile(v, 0.99):.2e} | {v.max():.2e}"
for name, x in fam.items():
    t0 = time.time()
    lp64 = oracle_lp(x, np.float64)
    ev = {"fp32 oracle (torch CPU)": oracle_lp(x, np.float32)}
    for k, r in runners.items(): ev[k] = hip_lp(r, x).astype(np.float64)
    flat = lp64.reshape(len(x), -1)
    am = flat.argmin(1); idx = np.arange(len(x))
    e_at = {k: v.reshape(len(x), -1)[idx, am] - flat[idx, am] for k, v in ev.items()}
    e_min = {k: v.reshape(len(x), -1).min(1) - flat.min(1) for k, v in ev.items()}
    e_all = {k: np.abs(v - lp64).reshape(len(x), -1).max(1) for k, v in ev.items()}
    rng_ = flat.max(1) - flat.min(1)
    print(f"\n### {name}, {len(x)} windows ({time.time() - t0:.0f} s); fp64 log-power: min {flat.min(1).mean():.1f} dB, range {rng_.mean():.1f} dB on average\n")
    print("| evaluation | quantity | median | p90 | p99 | max | mean (signed) | KS distance to the fp32 oracle's sample (p) |")
    print("|---|---|---|---|---|---|---|---|")
    ref = "fp32 oracle (torch CPU)"
    for k in ev:
        for qn, e in (("|e_at| (dB) at the fp64 arg-min bin", e_at), ("|e_min| (dB) of the window minimum", e_min)):
            ks = ks_2samp(np.abs(e[k]), np.abs(e[ref])) if k != ref else None
            print(f"| {k} | {qn} | {q(np.abs(e[k]))} | {e[k].mean():+.2e} | " + ("—" if ks is None else f"{ks.statistic:.3f} ({ks.pvalue:.2f})") + " |")
        print(f"| {k} | max |lp - lp64| over all bins (dB) | {q(e_all[k])} | | |")
    d = np.abs(e_min["default path (planes CQT, split-f16)"]); o = np.abs(e_min[ref])
    worst = np.argsort(-d)[:5]
    print("\nfive largest |e_min| of the default path: " + ", ".join(
        f"#{i}: {d[i]:.2e} dB (fp32 oracle {o[i]:.2e}, exact-f32 {abs(e_min['exact-f32 A/B path'][i]):.2e}; min lp64 {flat[i].min():.1f} dB, "
        f"range {rng_[i]:.1f} dB)" for i in worst))
    print(f"\nin output units: a shift of the minimum by d dB moves z by bn_a * d / range = {2.48:.2f} * d / {rng_.mean():.0f} on average")
