"""bench.py on hardware: the N > 1 path of the benchmark contract executed before the driver does.

The box has one GPU, so the two ranks share device 0 (`--share-gpu`): what is checked is the control flow the driver's
8-GPU launch goes through — self-launch over torch.distributed.run, rendezvous on 127.0.0.1, the gloo control group
(barrier + max over ranks on CPU tensors; the data path has no collective), one JSON line from rank 0 —
not a scaling number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_bench_two_ranks_share_one_gpu():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-exact-f32", "--no-fp8-extra", "--sustained-s", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout[-2000:]  # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["unit"] == "windows/s" and line["value"] > 0 and line["outputs_finite"] is True
    # value = windows of ALL ranks / max-over-ranks time
    assert abs(line["value"] - 2 * 256 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    assert "fp8" not in line["dtype"] and line["roofline"]["frac"] > 0


def test_bench_single_rank_line_has_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--sustained-s", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-4000:]
    (line,) = _json_lines(res.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and "workload" in line["config"] and "fp8" not in line["dtype"]
    # the exact-f32 A/B rate is reported BESIDE the headline; the fp8-corrections mode left the product in round 6
    assert "fp8_corrections_windows_per_s" not in line and line["exact_f32_windows_per_s"] > 0
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the kernel's own I/O (c1 materialised: an implementation choice) is not called algorithmic; the branch's is beside it
    assert "algorithmic_bytes_per_launch" not in r
    assert r["kernel_io_bytes_per_launch"] > 3 * r["branch_algorithmic_bytes_per_launch"] > 0
    # every BASELINE.json config and the reference's batch-1 host call pattern are in the driver's record (round 4)
    for k in ("b1024", "bf16_b1024", "ext44k_b512", "tracks_256x3min", "b16"):
        assert line["configs"][k]["windows_per_s"] > 0, k
        assert k == "tracks_256x3min" or (line["configs"][k]["steps"] >= 10 and line["configs"][k]["warmup"] == 3), k
    assert line["seam_b1_host"]["windows_per_s"] > 0 and line["seam_b1_host"]["ms_per_call"] > 0
    st = r["step_traffic"]
    assert st is None or (st["ratio"] > 1.0 and st["bytes_per_step"] > st["algorithmic_bytes_per_step"])


def test_library_then_torch_share_one_hip_runtime():
    """Load order must not matter: PyTorch bundles its own copy of libamdhip64 and asks for it by another name than this
    library does.  Loaded in the order [this library, torch], a process used to end up with TWO HIP runtimes and the one
    that initialised second saw no device (build() followed by smoke() in one process).  _native.load_library() now
    brings the runtime in under torch's name first: one copy in the process, every order works."""
    code = (
        "from basic_pitch_amd import _native; _native.load_library()\n"
        "import torch; assert torch.cuda.is_available(); x = torch.ones(3).cuda()\n"
        "from basic_pitch_amd import Model; m = Model(max_windows=1); assert m.info()['arch'].startswith('gfx950')\n"
        "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
        "assert len(libs) == 1, libs\n"
        "print('ok', float(x.sum()))\n"
    )
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0 and "ok 3.0" in res.stdout, res.stderr[-3000:]
