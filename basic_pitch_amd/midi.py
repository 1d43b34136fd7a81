"""Minimal stand-in for the part of `pretty_midi` the reference's note path uses.

`basic_pitch/note_creation.py:222-267` builds a `pretty_midi.PrettyMIDI(initial_tempo=...)` with
`Instrument(program=instrument_name_to_program("Electric Piano 1"))`, `Note(velocity, pitch, start, end)`
and `PitchBend(pitch, time)` objects and later calls `.write(path)`.  pretty_midi is not installable
here, so these classes carry the same attributes and `write()` emits a type-1 Standard MIDI File with
pretty_midi's conventions (resolution 220 ticks per quarter note, tempo + 4/4 time signature on track 0,
one track per instrument, channel 0..15 skipping 9, events ordered by tick).  Byte-level identity with
pretty_midi's own writer (mido) is NOT pinned — there is no pretty_midi here to compare against.
"""
from __future__ import annotations

import struct
from typing import List

ELECTRIC_PIANO_1 = 4  # pretty_midi.instrument_name_to_program("Electric Piano 1")


def instrument_name_to_program(name: str) -> int:
    if name != "Electric Piano 1":
        raise ValueError("only 'Electric Piano 1' is mapped (the reference uses nothing else)")
    return ELECTRIC_PIANO_1


class Note:
    def __init__(self, velocity: int, pitch: int, start: float, end: float):
        self.velocity, self.pitch, self.start, self.end = int(velocity), int(pitch), float(start), float(end)

    def __repr__(self) -> str:
        return f"Note(start={self.start:f}, end={self.end:f}, pitch={self.pitch}, velocity={self.velocity})"


class PitchBend:
    def __init__(self, pitch: int, time: float):
        self.pitch, self.time = int(pitch), float(time)


class Instrument:
    def __init__(self, program: int, is_drum: bool = False, name: str = ""):
        self.program, self.is_drum, self.name = int(program), bool(is_drum), name
        self.notes: List[Note] = []
        self.pitch_bends: List[PitchBend] = []


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


class PrettyMIDI:
    def __init__(self, initial_tempo: float = 120.0, resolution: int = 220):
        self.resolution = int(resolution)
        self.initial_tempo = float(initial_tempo)
        self.instruments: List[Instrument] = []

    def time_to_tick(self, t: float) -> int:
        return int(round(t * self.resolution * self.initial_tempo / 60.0))

    def get_end_time(self) -> float:
        ends = [n.end for i in self.instruments for n in i.notes] + [b.time for i in self.instruments for b in i.pitch_bends]
        return max(ends) if ends else 0.0

    def _track(self, events) -> bytes:
        # events: (tick, order, bytes); stable sort by tick then order (note-offs before note-ons at a tick)
        data = bytearray()
        last = 0
        for tick, _, payload in sorted(events, key=lambda e: (e[0], e[1])):
            data += _vlq(tick - last) + payload
            last = tick
        data += _vlq(0) + b"\xff\x2f\x00"
        return b"MTrk" + struct.pack(">I", len(data)) + bytes(data)

    def write(self, filename: str) -> None:
        tempo_us = int(round(60_000_000.0 / self.initial_tempo))
        tracks = [
            self._track(
                [
                    (0, 0, b"\xff\x51\x03" + struct.pack(">I", tempo_us)[1:]),
                    (0, 1, b"\xff\x58\x04\x04\x02\x18\x08"),
                ]
            )
        ]
        channels = [c for c in range(16) if c != 9]
        for n, inst in enumerate(self.instruments):
            ch = 9 if inst.is_drum else channels[n % len(channels)]
            ev = [(0, 0, bytes([0xC0 | ch, inst.program & 0x7F]))]
            for b in inst.pitch_bends:
                v = max(-8192, min(8191, b.pitch)) + 8192
                ev.append((self.time_to_tick(b.time), 1, bytes([0xE0 | ch, v & 0x7F, (v >> 7) & 0x7F])))
            for note in inst.notes:
                ev.append((self.time_to_tick(note.start), 3, bytes([0x90 | ch, note.pitch & 0x7F, note.velocity & 0x7F])))
                ev.append((self.time_to_tick(note.end), 2, bytes([0x90 | ch, note.pitch & 0x7F, 0])))
            tracks.append(self._track(ev))
        with open(filename, "wb") as f:
            f.write(b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), self.resolution))
            for t in tracks:
                f.write(t)
