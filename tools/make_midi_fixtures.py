#!/usr/bin/env python
"""Dev-time fixture builder: the Standard MIDI File bytes `pretty_midi.PrettyMIDI.write` (pretty_midi 0.2.x over mido
1.2/1.3) produces for the MIDI objects the UNMODIFIED reference builds (`note_creation.py:222-267`), written out from
the documented structure of those two libraries — neither is installable here, so this script is their algorithm
spelled out with message objects and a comparator, independently of basic_pitch_amd/midi.py (which works on byte
tuples and sort keys):

  pretty_midi.PrettyMIDI.write
    timing track: time_signature 4/4 at 0 (no user signature), set_tempo per tick-scale entry
      (tempo = int(6e7 / (60. / (tick_scale * resolution)))), sorted by event_compare, end_of_track at last + 1;
    per instrument n: channel = [0..8, 10..15][n % 15]; program_change at 0; per note note_on(start) and
      note_on(end, velocity 0); per bend pitchwheel; sorted(cmp_to_key(event_compare)); the note-off-first fix-up;
      end_of_track at last + 1; absolute ticks -> deltas;
    event_compare: equal times -> difference of secondary keys (set_tempo 1, time_signature 2, program_change 6,
      pitchwheel 7 (+ pitch), note_on 10 (+ note * 256 + velocity), end_of_track 11; all * 65536), else time difference;
    time_to_tick(t) = int(round(t / tick_scale)) beyond the last known tick (the only tick known is 0);
  mido.MidiFile.save (type 1)
    MThd = pack('>hhh', type, n_tracks, ticks_per_beat); per track MTrk + length + messages;
    delta time as variable-length quantity; meta = FF type len data (and resets running status); channel messages
    with running status; pitchwheel value + 8192 as LSB, MSB; time_signature data = numerator, log2(denominator),
    clocks_per_click = 24, notated_32nd_notes_per_beat = 8; set_tempo data = 3 bytes big endian.

Input: tests/golden/note_fixtures.npz (the reference's MIDI object contents).  Output: tests/golden/midi/<case>.mid.
"""
from __future__ import annotations

import functools
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("clip_default", "clip_multi_bends", "clip_tempo_90", "syn_multi_bends", "syn_dense_overlaps")


class Msg:
    def __init__(self, type, time, **kw):
        self.type, self.time = type, time
        self.__dict__.update(kw)


SECONDARY = {
    "set_tempo": lambda e: 1 * 256 * 256,
    "time_signature": lambda e: 2 * 256 * 256,
    "program_change": lambda e: 6 * 256 * 256,
    "pitchwheel": lambda e: (7 * 256 * 256) + e.pitch,
    "note_on": lambda e: (10 * 256 * 256) + (e.note * 256) + e.velocity,
    "end_of_track": lambda e: 11 * 256 * 256,
}


def event_compare(e1, e2):
    if e1.time == e2.time and e1.type in SECONDARY and e2.type in SECONDARY:
        return SECONDARY[e1.type](e1) - SECONDARY[e2.type](e2)
    return e1.time - e2.time


def varlen(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.insert(0, (v & 0x7F) | 0x80)
        v >>= 7
    return bytes(out)


def msg_bytes(m):
    if m.type == "set_tempo":
        return b"\xff\x51\x03" + m.tempo.to_bytes(3, "big")
    if m.type == "time_signature":
        return bytes([0xFF, 0x58, 4, m.numerator, {1: 0, 2: 1, 4: 2, 8: 3, 16: 4}[m.denominator], 24, 8])
    if m.type == "end_of_track":
        return b"\xff\x2f\x00"
    if m.type == "program_change":
        return bytes([0xC0 | m.channel, m.program])
    if m.type == "note_on":
        return bytes([0x90 | m.channel, m.note, m.velocity])
    if m.type == "pitchwheel":
        v = m.pitch + 8192
        return bytes([0xE0 | m.channel, v & 0x7F, v >> 7])
    raise ValueError(m.type)


def write_track(track):
    data = bytearray()
    running = None
    for m in track:
        data += varlen(m.time)
        b = msg_bytes(m)
        if b[0] == 0xFF:
            data += b
            running = None
        else:
            data += b[1:] if b[0] == running else b
            running = b[0]
    return b"MTrk" + struct.pack(">L", len(data)) + bytes(data)


def smf_bytes(initial_tempo, resolution, instruments):
    """instruments: list of (program, [(velocity, pitch, start, end)], [(pitch, time)])."""
    tick_scale = 60.0 / (initial_tempo * resolution)

    def time_to_tick(t):
        if not t > 0:  # searchsorted([0], t, side="left") == 0 -> tick 0
            return 0
        return int(round((t - 0.0) / tick_scale))

    timing = [Msg("time_signature", 0, numerator=4, denominator=4),
              Msg("set_tempo", 0, tempo=int(6e7 / (60.0 / (tick_scale * resolution))))]
    timing.sort(key=functools.cmp_to_key(event_compare))
    timing.append(Msg("end_of_track", timing[-1].time + 1))
    tracks = [timing]
    channels = list(range(16))
    channels.remove(9)
    for n, (program, notes, bends) in enumerate(instruments):
        channel = channels[n % len(channels)]
        track = [Msg("program_change", 0, program=program, channel=channel)]
        for velocity, pitch, start, end in notes:
            track.append(Msg("note_on", time_to_tick(start), channel=channel, note=pitch, velocity=velocity))
            track.append(Msg("note_on", time_to_tick(end), channel=channel, note=pitch, velocity=0))
        for pitch, time in bends:
            track.append(Msg("pitchwheel", time_to_tick(time), channel=channel, pitch=pitch))
        track = sorted(track, key=functools.cmp_to_key(event_compare))
        for i, (e1, e2) in enumerate(zip(track[:-1], track[1:])):
            if (e1.time == e2.time and e1.type == "note_on" and e2.type == "note_on" and e1.note == e2.note
                    and e1.velocity != 0 and e2.velocity == 0):
                track[i] = e2
                track[i + 1] = e1
        track.append(Msg("end_of_track", track[-1].time + 1))
        tracks.append(track)
    for track in tracks:
        tick = 0
        for e in track:
            e.time -= tick
            tick += e.time
    out = b"MThd" + struct.pack(">L", 6) + struct.pack(">hhh", 1, len(tracks), resolution)
    for track in tracks:
        out += write_track(track)
    return out


def main() -> None:
    fx = np.load(os.path.join(ROOT, "tests", "golden", "note_fixtures.npz"))
    out_dir = os.path.join(ROOT, "tests", "golden", "midi")
    os.makedirs(out_dir, exist_ok=True)
    for name in CASES:
        tempo, res = fx[f"{name}/midi_tempo_resolution"]
        instruments = []
        n0 = b0 = 0
        for prog, nn, nb in zip(fx[f"{name}/inst_program"], fx[f"{name}/inst_n_notes"], fx[f"{name}/inst_n_bends"]):
            notes = [(int(fx[f"{name}/note_velocity"][i]), int(fx[f"{name}/note_pitch"][i]),
                      float(fx[f"{name}/note_start"][i]), float(fx[f"{name}/note_end"][i])) for i in range(n0, n0 + nn)]
            bends = [(int(fx[f"{name}/pb_pitch"][i]), float(fx[f"{name}/pb_time"][i])) for i in range(b0, b0 + nb)]
            instruments.append((int(prog), notes, bends))
            n0, b0 = n0 + nn, b0 + nb
        data = smf_bytes(float(tempo), int(res), instruments)
        with open(os.path.join(out_dir, f"{name}.mid"), "wb") as f:
            f.write(data)
        print(name, len(instruments), "instruments", len(data), "bytes")


if __name__ == "__main__":
    main()
