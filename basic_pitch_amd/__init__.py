"""basic_pitch_amd — MI355X (gfx950) native executor of the Basic Pitch inference hot path.

Host-side mirror of the reference's `basic_pitch.inference` surface over a C-ABI HIP library
(include/basic_pitch_amd.h).  No CPU execution path: importing is cheap, creating a `Model` needs
the built library and an MI355X.
"""
from .constants import *  # noqa: F401,F403
from .inference import (  # noqa: F401
    ICASSP_2022_MODEL_PATH,
    Model,
    get_audio_input,
    run_inference,
    unwrap_output,
    window_audio_file,
)
from .inference import predict, predict_and_save, predict_and_save_many, predict_many, transcribe_files  # noqa: F401
from .sharding import predict_and_save_sharded, predict_many_sharded  # noqa: F401

__version__ = "0.1.0"
