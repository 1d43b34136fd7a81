# How much does the choice among filter designs that all meet libsoxr's SOXR_HQ specification move the posteriorgrams?
# The reference's golden clip (44.1 kHz) is brought to 48 kHz and to 16 kHz (a long Kaiser polyphase filter, float64) and
# from there to 22.05 kHz by
#   (a) one poly-phase stage with the SINGLE-RATE design rule (rho = .5, taps = 1 mod 4) — what the product ran in round 2,
#   (b) one poly-phase stage with lsx_design_lpf's POLY-PHASE rule (rho = .75, taps = k * phases - 1) — what libsoxr's own
#       stage plan (oracle/soxr_oracle.py stage_plan) runs for these two ratios; its even tap count leaves half a tick
#       (1 / 294 of an input sample at 48 kHz) of alignment open: both roundings are measured,
# and the fp32 graph oracle turns each into posteriorgrams.  CPU only:  python tools/soxr_plan_distance.py
import os, sys
import numpy as np
import scipy.signal
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basic_pitch_amd import audio as A
from oracle import bp_oracle as O, soxr_oracle as S

W = O.load_weights()
pcm, sr = A.read_wav(os.path.join(ROOT, "tests", "golden", "vocadito_10.wav"))
x = A.to_mono(pcm).astype(np.float64)
gold = np.load(os.path.join(ROOT, "tests", "golden", "vocadito_10_model_output.npz"))


def direct(xs, up, down, h, c, n_out):
    y = np.zeros(n_out)
    for k in range(n_out):
        base = k * down + c
        j_lo = max(0, -(-(base - (len(h) - 1)) // up))
        j_hi = min(len(xs) - 1, base // up)
        j = np.arange(j_lo, j_hi + 1)
        y[k] = np.dot(xs[j], h[base - j * up])
    return y.astype(np.float32)


def posteriorgrams(y):
    return O.run_track(y, W, np.float32, batch=8)


for rate, (u0, d0) in ((48000, (160, 147)), (16000, (160, 441))):
    xs = scipy.signal.resample_poly(x, u0, d0, window=("kaiser", 16.0)).astype(np.float32).astype(np.float64)
    fr_up, fr_down = (147, 320) if rate == 48000 else (441, 320)
    n_out = int(np.ceil(len(xs) * 22050 / rate))
    variants = {}
    h = S.taps(fr_up, fr_down, poly_rule=False)
    variants["single-rate rule (round 2)"] = direct(xs, fr_up, fr_down, h, (len(h) - 1) // 2, n_out)
    h = S.taps(fr_up, fr_down, poly_rule=True)
    variants["poly-phase rule, centre floor"] = direct(xs, fr_up, fr_down, h, (len(h) - 1) // 2, n_out)
    variants["poly-phase rule, centre ceil"] = direct(xs, fr_up, fr_down, h, len(h) // 2, n_out)
    post = {k: posteriorgrams(v) for k, v in variants.items()}
    ref = "poly-phase rule, centre floor"
    print(f"\n## {rate} Hz -> 22050 Hz ({fr_up} : {fr_down}), {len(xs)} samples in, stage plan {S.stage_plan(rate / 22050.0)}\n")
    print("| design | taps | max |sample - (b floor)| | posteriorgram max-abs vs (b floor): note / onset / contour | vs the reference's golden (44.1 kHz path) |")
    print("|---|---|---|---|---|")
    for k, v in variants.items():
        n = min(len(v), len(variants[ref]))
        ds = np.abs(v[:n] - variants[ref][:n]).max()
        dp = [np.abs(post[k][m] - post[ref][m]).max() for m in ("note", "onset", "contour")]
        T = min(post[k]["note"].shape[0], gold["note"].shape[0])
        dg = [np.abs(post[k][m][:T] - gold[m][:T]).max() for m in ("note", "onset", "contour")]
        nt = len(S.taps(fr_up, fr_down, poly_rule=not k.startswith("single")))
        print(f"| {k} | {nt} | {ds:.2e} | {dp[0]:.2e} / {dp[1]:.2e} / {dp[2]:.2e} | {dg[0]:.2e} / {dg[1]:.2e} / {dg[2]:.2e} |")
