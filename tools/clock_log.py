#!/usr/bin/env python
"""Sample the GPU's clocks, socket power and temperature while a command runs (round 5).

    python tools/clock_log.py --tag mfma_power --out gpurun_out/clocks -- tools/ubench/bin/mfma_power

Sources, in order of preference (whatever the box exposes is recorded, the header of the CSV says which):
  * sysfs hwmon of the amdgpu device: freq1_input (sclk, Hz), freq2_input (mclk), power1_average / power1_input (uW),
    temp1_input — read at ~20 Hz, no subprocess;
  * `amd-smi metric -c -p --json` in a side loop (~1 Hz: the tool takes a few hundred ms to start) as a cross-check and as
    the only source when hwmon is absent.
Writes <out>/<tag>.csv (t_s, source, sclk_mhz, mclk_mhz, power_w, temp_c), <out>/<tag>.stdout (the command's output) and
prints a summary: over the samples taken while the command ran and the power was above half of its maximum (the loaded
phase), min / median / max of sclk and power.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def find_hwmon():
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "vendor")):
            continue
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
            return hw
    return None


def read_int(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


def hwmon_sample(hw):
    sclk = read_int(os.path.join(hw, "freq1_input"))
    mclk = read_int(os.path.join(hw, "freq2_input"))
    p = read_int(os.path.join(hw, "power1_average"))
    if p is None:
        p = read_int(os.path.join(hw, "power1_input"))
    t = read_int(os.path.join(hw, "temp1_input"))
    return (sclk / 1e6 if sclk else None, mclk / 1e6 if mclk else None, p / 1e6 if p else None, t / 1e3 if t else None)


def _dig(d, *names):
    """first numeric value found under any of the key names, searching nested dicts / lists"""
    stack = [d]
    while stack:
        x = stack.pop()
        if isinstance(x, dict):
            for k, v in x.items():
                if k in names:
                    if isinstance(v, dict) and "value" in v:
                        v = v["value"]
                    try:
                        return float(v)
                    except (TypeError, ValueError):
                        pass
                stack.append(v)
        elif isinstance(x, list):
            stack.extend(x)
    return None


def amdsmi_sample():
    try:
        out = subprocess.run(["amd-smi", "metric", "-c", "-p", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out[out.index("[") if "[" in out and out.index("[") < out.index("{") else out.index("{"):])
    except Exception:
        return None
    clk = None
    # gfx clocks: take the largest current gfx clock reported (one entry per XCD)
    vals = []
    stack = [d]
    while stack:
        x = stack.pop()
        if isinstance(x, dict):
            for k, v in x.items():
                if k.startswith("gfx") and isinstance(v, dict) and "clk" in v:
                    c = v["clk"]
                    if isinstance(c, dict):
                        c = c.get("value")
                    try:
                        vals.append(float(c))
                    except (TypeError, ValueError):
                        pass
                stack.append(v)
        elif isinstance(x, list):
            stack.extend(x)
    if vals:
        clk = max(vals)
    mclk = _dig(d, "mem_0")
    return (clk, mclk, _dig(d, "socket_power", "current_socket_power", "average_socket_power"), None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--out", default="gpurun_out/clocks")
    ap.add_argument("--interval", type=float, default=0.05)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    os.makedirs(a.out, exist_ok=True)
    hw = find_hwmon()
    rows = []
    stop = threading.Event()
    t0 = time.time()

    def smi_loop():
        while not stop.is_set():
            s = amdsmi_sample()
            if s:
                rows.append((time.time() - t0, "amd-smi") + s)
            stop.wait(0.5)

    th = threading.Thread(target=smi_loop, daemon=True)
    th.start()
    with open(os.path.join(a.out, a.tag + ".stdout"), "w") as fo:
        proc = subprocess.Popen(cmd, stdout=fo, stderr=subprocess.STDOUT)
        while proc.poll() is None:
            if hw:
                rows.append((time.time() - t0, "hwmon") + hwmon_sample(hw))
            time.sleep(a.interval)
    stop.set()
    th.join(timeout=15)
    rows.sort()
    with open(os.path.join(a.out, a.tag + ".csv"), "w") as f:
        f.write(f"# hwmon={hw} cmd={' '.join(cmd)} rc={proc.returncode}\n")
        f.write("t_s,source,sclk_mhz,mclk_mhz,power_w,temp_c\n")
        for r in rows:
            f.write(",".join("" if v is None else (f"{v:.3f}" if isinstance(v, float) else str(v)) for v in r) + "\n")
    for src in ("hwmon", "amd-smi"):
        rr = [r for r in rows if r[1] == src and r[4] is not None]
        if not rr:
            print(f"[{a.tag}] {src}: no samples")
            continue
        pmax = max(r[4] for r in rr)
        busy = [r for r in rr if r[4] >= 0.5 * pmax]
        sc = [r[2] for r in busy if r[2] is not None]
        pw = [r[4] for r in busy]
        mc = [r[3] for r in busy if r[3] is not None]

        def q(v):
            return "n/a" if not v else f"{min(v):.0f} / {statistics.median(v):.0f} / {max(v):.0f}"

        print(f"[{a.tag}] {src}: {len(rr)} samples, {len(busy)} loaded; sclk MHz min/med/max {q(sc)}; "
              f"mclk {q(mc)}; power W {q(pw)}; idle power {min(r[4] for r in rr):.0f} W")
    return proc.returncode


if __name__ == "__main__":
    sys.exit(main())
