#!/bin/bash
# Host code of the library under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU needed, ~5 minutes):
#   fuzz_flac: 20,000 mutated copies (byte noise in the headers, truncation, bit flips, spliced garbage) of each FLAC fixture
#              through bp_flac_info / bp_flac_decode, every input in an exact-size heap block;
#   fuzz_host: 6,000 mutated WAV headers through bp_wav_info / bp_wav_decode, and 300 random posteriorgram maps (NaN cells,
#              thresholds from -0.1 to 1.5, every switch) through bp_notes_decode, bp_notes_to_midi and bp_notes_to_csv
#              (the device entry points file_pipeline.cpp links against are stubbed).
# Any report is a bug; a clean run prints the two summary lines only.
set -e
cd "$(dirname "$0")"
src=../../basic_pitch_amd/csrc
flags="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -pthread -w"
g++ $flags fuzz_flac.cpp $src/flac_decode.cpp -o /tmp/bp_fuzz_flac
g++ $flags fuzz_host.cpp $src/flac_decode.cpp $src/note_decode.cpp $src/file_pipeline.cpp -o /tmp/bp_fuzz_host
for f in ../../tests/golden/*.flac; do /tmp/bp_fuzz_flac $f; done
/tmp/bp_fuzz_host ../../tests/golden/vocadito_10.wav
