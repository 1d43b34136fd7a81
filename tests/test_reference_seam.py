"""BASELINE.json configs[0] ("single WAV through the reference's ONNX CPU predict(): plumbing, no GPU") and the
zero-patch seam of SURVEY.md §8b: the UNMODIFIED reference package (`/root/reference/basic_pitch`: `__init__`,
`inference.py`, `note_creation.py`, `constants.py`) is imported for real and its own `predict()` is run on its own test
clip, with
  * `onnxruntime`      = basic_pitch_amd.ort_shim (installed with `ort_shim.install()`), so `Model(ICASSP_2022_MODEL_PATH)`
                         goes down the reference's ONNX leg: get_available_providers -> InferenceSession(nmp.onnx) ->
                         session.run([...], {"serving_default_input_2:0": x}) one window at a time (inference.py:130-182);
  * `librosa.load`     = basic_pitch_amd.audio.load (decode + soxr_hq-design resampling);
  * `pretty_midi`      = basic_pitch_amd.midi;
  * the session's compute = the fp32 ORACLE behind the shim's `Model` (there is no GPU in this container; on the GPU box
    the same shim drives libbasicpitch_amd.so, tests/test_gpu_parity.py::test_ort_shim_session_runs_reference_call_pattern).
What is checked is the reference's own known-answer test (tests/test_inference.py:43-76): 28 note events and the
posteriorgrams at atol = 1e-4.  Needs the reference checkout: skipped on the GPU box.
"""
import os
import sys
import types

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "basic_pitch")), reason="needs the reference checkout")


class _OracleModel:
    """Stands where basic_pitch_amd.inference.Model stands inside the shim's session: same constructor, same predict."""

    calls = []

    def __init__(self, model_path, device=0, max_windows=256):
        from basic_pitch_amd import weights
        from oracle import bp_oracle as O

        self._O = O
        blob_path = os.fspath(model_path)
        assert open(blob_path, "rb").read()[:8] == weights.MAGIC
        self._W = O.load_weights(blob_path)

    def predict(self, x):
        x = np.asarray(x, dtype=np.float32)
        _OracleModel.calls.append(x.shape)
        return self._O.forward(x[:, :, 0] if x.ndim == 3 else x, self._W, np.float32)


def test_unmodified_reference_predict_through_the_onnxruntime_seam(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_stubs

    from basic_pitch_amd import audio, inference as amd_inference, midi, ort_shim

    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in
             ("basic_pitch", "librosa", "pretty_midi", "mir_eval", "resampy", "onnxruntime")}
    try:
        ref_stubs.install_stubs()
        sys.modules["pretty_midi"] = midi
        sys.modules["librosa"].load = lambda path, sr=22050, mono=True: audio.load(path, sr=sr, mono=mono)
        for k in [k for k in sys.modules if k.split(".")[0] == "basic_pitch"]:
            del sys.modules[k]
        ort_shim.install()
        monkeypatch.setattr(amd_inference, "Model", _OracleModel)
        monkeypatch.syspath_prepend(REF)
        import basic_pitch
        from basic_pitch import inference as ref_inference

        assert basic_pitch.ONNX_PRESENT and not basic_pitch.TF_PRESENT
        assert str(basic_pitch.ICASSP_2022_MODEL_PATH).endswith("nmp.onnx")
        assert ref_inference.__file__.startswith(REF)
        _OracleModel.calls.clear()
        model_output, midi_data, note_events = ref_inference.predict(os.path.join(GOLDEN, "vocadito_10.wav"))
        assert _OracleModel.calls == [(1, 43844, 1)] * 6  # the reference's batch-1 loop, six windows
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in
                  ("basic_pitch", "librosa", "pretty_midi", "mir_eval", "resampy", "onnxruntime")]:
            del sys.modules[k]
        sys.modules.update(saved)
    g = np.load(os.path.join(GOLDEN, "vocadito_10_model_output.npz"))
    for k in ("note", "onset", "contour"):
        assert model_output[k].shape == g[k].shape
        assert np.abs(model_output[k] - g[k]).max() <= 1e-4, (k, np.abs(model_output[k] - g[k]).max())
    ge = np.load(os.path.join(GOLDEN, "vocadito_10_note_events.npz"))
    assert len(note_events) == 28
    for i, e in enumerate(note_events):
        assert e[0] == ge["start_s"][i] and e[1] == ge["end_s"][i] and e[2] == ge["pitch"][i]
        assert abs(float(e[3]) - float(ge["amplitude"][i])) <= 1e-4
        assert list(e[4]) == list(ge["bend_values"][ge["bend_offsets"][i] : ge["bend_offsets"][i + 1]])
    assert len(midi_data.instruments) == 1 and len(midi_data.instruments[0].notes) == 28
    assert midi_data.to_bytes()[:4] == b"MThd"
