"""Minimal stand-in for the part of `pretty_midi` the reference's note path uses.

`basic_pitch/note_creation.py:222-267` builds a `pretty_midi.PrettyMIDI(initial_tempo=...)` with
`Instrument(program=instrument_name_to_program("Electric Piano 1"))`, `Note(velocity, pitch, start, end)`
and `PitchBend(pitch, time)` objects and `inference.py:586` calls `.write(path)`.  pretty_midi (and mido, which
does its byte encoding) are not installable here, so these classes carry the same attributes and `write()` follows
pretty_midi 0.2.x `PrettyMIDI.write` + mido `MidiFile.save` step by step:

  * type-1 file, resolution 220 ticks per quarter note;
  * track 0: `set_tempo` (int(6e7 / (60 / (tick_scale * resolution))) us per quarter) then `time_signature` 4/4
    (24 clocks per click, 8 notated 32nds) at tick 0, `end_of_track` one tick after the last event;
  * one track per instrument, channel n mod 15 over 0..15 without 9: `program_change` at tick 0, per note a
    `note_on` at its start and a `note_on` with velocity 0 at its end, per pitch bend a `pitchwheel`; stable sort by
    (tick, class) with pretty_midi's class keys (program change < pitchwheel by value < note_on by
    note * 256 + velocity), its note-off-before-note-on fix-up, `end_of_track` one tick after the last event;
  * tick = int(round(time / tick_scale)), tick_scale = 60 / (tempo * resolution), Python's round-half-even;
  * mido's encoding: delta times as variable-length quantities, running status within a track (a status byte equal
    to the previous channel message's is omitted; any meta event resets it).

`synthesize()` restates pretty_midi's sine sonification (what `note_creation.sonify_midi`, note_creation.py:119-128,
renders and `inference.py:588-594` saves as `*_basic_pitch.wav`); no pretty_midi output exists to pin its samples, the
test checks the spectrum.

tests/golden/midi/*.mid are those steps written out independently (tools/make_midi_fixtures.py) for the events the
UNMODIFIED reference produced; `write()` must reproduce them byte for byte (tests/test_note_decode.py).
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
    __slots__ = ("velocity", "pitch", "start", "end")

    def __init__(self, velocity: int, pitch: int, start: float, end: float):
        self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    def __repr__(self) -> str:
        return f"Note(start={self.start:f}, end={self.end:f}, pitch={self.pitch}, velocity={self.velocity})"


class PitchBend:
    __slots__ = ("pitch", "time")

    def __init__(self, pitch: int, time: float):
        self.pitch, self.time = pitch, time


def note_number_to_hz(note_number: float) -> float:
    """pretty_midi.note_number_to_hz: A4 = 440 Hz, 12-tone equal temperament."""
    return 440.0 * (2.0 ** ((note_number - 69) / 12.0))


def pitch_bend_to_semitones(pitch_bend: int, semitone_range: float = 2.0) -> float:
    """pretty_midi.pitch_bend_to_semitones: +-8192 <-> +-semitone_range."""
    return semitone_range * pitch_bend / 8192.0


class Instrument:
    """pretty_midi.Instrument's attributes.  An instrument built by `from_arrays` (note_creation.note_events_to_midi) keeps
    its notes and pitch bends as numpy arrays and only turns them into `Note` / `PitchBend` objects when `.notes` /
    `.pitch_bends` are first read: a batch job that only writes the file never creates the ~11 k objects of a 3-minute
    track (object creation runs under the GIL, which is what bounds a file job — DESIGN.md §5)."""

    def __init__(self, program: int, is_drum: bool = False, name: str = ""):
        self.program, self.is_drum, self.name = program, is_drum, name
        self._notes: List[Note] = []
        self._pitch_bends: List[PitchBend] = []
        self._lazy = None  # (pitch, velocity, start, end, bend_pitch, bend_time) arrays, or None once materialised

    @classmethod
    def from_arrays(cls, program: int, pitch, velocity, start, end, bend_pitch, bend_time) -> "Instrument":
        inst = cls(program)
        inst._lazy = (pitch, velocity, start, end, bend_pitch, bend_time)
        return inst

    def _materialise(self) -> None:
        if self._lazy is not None:
            pitch, velocity, start, end, bend_pitch, bend_time = self._lazy
            self._lazy = None
            self._notes = list(map(Note, velocity.tolist(), pitch.tolist(), start.tolist(), end.tolist()))
            self._pitch_bends = list(map(PitchBend, bend_pitch.tolist(), bend_time.tolist()))

    @property
    def notes(self) -> List[Note]:
        self._materialise()
        return self._notes

    @notes.setter
    def notes(self, value) -> None:
        self._materialise()
        self._notes = value

    @property
    def pitch_bends(self) -> List[PitchBend]:
        self._materialise()
        return self._pitch_bends

    @pitch_bends.setter
    def pitch_bends(self, value) -> None:
        self._materialise()
        self._pitch_bends = value

    def _event_arrays(self):
        """(pitch, velocity, start, end, bend_pitch, bend_time) as int64 / float64 arrays."""
        import numpy as np

        if self._lazy is not None:
            return self._lazy
        nn, nb = len(self._notes), len(self._pitch_bends)
        return (np.fromiter((int(x.pitch) for x in self._notes), dtype=np.int64, count=nn),
                np.fromiter((int(x.velocity) for x in self._notes), dtype=np.int64, count=nn),
                np.fromiter((x.start for x in self._notes), dtype=np.float64, count=nn),
                np.fromiter((x.end for x in self._notes), dtype=np.float64, count=nn),
                np.fromiter((int(x.pitch) for x in self._pitch_bends), dtype=np.int64, count=nb),
                np.fromiter((x.time for x in self._pitch_bends), dtype=np.float64, count=nb))

    def get_end_time(self) -> float:
        _, _, _, end, _, bend_time = self._event_arrays()
        return float(max(end.max() if end.size else 0.0, bend_time.max() if bend_time.size else 0.0))

    def synthesize(self, fs: int = 44100, wave=None):
        """pretty_midi.Instrument.synthesize restated: every note a `wave` (default np.sin) oscillator at its pitch,
        frequency multiplied by the instrument's pitch bends (with a phase offset at each bend so the waveform stays
        continuous), an exp(-t) envelope with a 0.1 s linear fade-out, scaled by the note's velocity."""
        import numpy as np

        wave = np.sin if wave is None else wave
        synthesized = np.zeros(int(fs * (self.get_end_time() + 1)))
        if self.is_drum:
            return synthesized
        fade_out = np.linspace(1, 0, int(0.1 * fs))
        bend_multiplier = np.ones(synthesized.shape)
        ordered_bends = sorted(self.pitch_bends, key=lambda bend: bend.time)
        end_bend = PitchBend(0, self.get_end_time())
        for start_bend, stop_bend in zip(ordered_bends, ordered_bends[1:] + [end_bend]):
            start, end = int(start_bend.time * fs), int(stop_bend.time * fs)
            bend_multiplier[start:end] = 2 ** (pitch_bend_to_semitones(start_bend.pitch) / 12.0)
        for note in self.notes:
            start, end = int(fs * note.start), int(fs * note.end)
            frequency = note_number_to_hz(note.pitch)
            offsets = np.zeros(end - start)
            for bend in ordered_bends:
                bend_sample = int(bend.time * fs)
                if start < bend_sample < end:
                    bend_so_far = bend_multiplier[start:bend_sample].mean()
                    bend_amount = bend_multiplier[bend_sample]
                    offsets[bend_sample - start :] = (bend_so_far - bend_amount) * (bend_sample - start)
            frequencies = 2 * np.pi * frequency * (bend_multiplier[start:end]) / fs
            note_waveform = wave(frequencies * np.arange(end - start) + 2 * np.pi * frequency * offsets / fs)
            envelope = np.exp(-np.arange(end - start) / (1.0 * fs))
            if envelope.shape[0] > fade_out.shape[0]:
                envelope[-fade_out.shape[0] :] *= fade_out
            else:
                envelope *= np.linspace(1, 0, envelope.shape[0])
            envelope *= note.velocity
            synthesized[start:end] += envelope * note_waveform
        return synthesized


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _chunk(tag: bytes, data: bytes) -> bytes:
    return tag + struct.pack(">I", len(data)) + data


class PrettyMIDI:
    def __init__(self, midi_file=None, resolution: int = 220, initial_tempo: float = 120.0):
        if midi_file is not None:
            raise NotImplementedError("reading MIDI files is not part of this stand-in")
        self.resolution = resolution
        self.initial_tempo = initial_tempo
        self._tick_scale = 60.0 / (initial_tempo * resolution)  # seconds per tick
        self.instruments: List[Instrument] = []

    def time_to_tick(self, time: float) -> int:
        """pretty_midi.PrettyMIDI.time_to_tick for a file with a single tempo."""
        if not time > 0:
            return 0
        return int(round(float(time) / self._tick_scale))

    def get_end_time(self) -> float:
        return max((i.get_end_time() for i in self.instruments), default=0.0)

    def synthesize(self, fs: int = 44100, wave=None):
        """pretty_midi.PrettyMIDI.synthesize restated: the instruments' waveforms summed (zero padded to the longest)
        and normalised to a peak of 1."""
        import numpy as np

        if len(self.instruments) == 0:
            return np.array([])
        waveforms = [i.synthesize(fs=fs, wave=wave) for i in self.instruments]
        synthesized = np.zeros(np.max([w.shape[0] for w in waveforms]))
        for waveform in waveforms:
            synthesized[: waveform.shape[0]] += waveform
        peak = np.abs(synthesized).max()
        return synthesized / peak if peak > 0 else synthesized

    @staticmethod
    def _encode_track(events) -> bytes:
        """events: (absolute tick, status byte or None for meta, payload) already in file order."""
        data = bytearray()
        last_tick, running = 0, None
        for tick, status, payload in events:
            data += _vlq(tick - last_tick)
            last_tick = tick
            if status is None:
                data += payload
                running = None
            else:
                if status != running:
                    data.append(status)
                data += payload
                running = status
        return _chunk(b"MTrk", bytes(data))

    def _instrument_track(self, n: int, inst: "Instrument") -> bytes:
        """One instrument's MTrk chunk, byte for byte what pretty_midi + mido write: program change, then note-ons /
        note-offs / pitch bends in pretty_midi's `event_compare` order (tick, then class key; stable), its fix-up swap of a
        note-on directly followed by the same pitch's note-off at the same tick, delta times as variable-length quantities,
        running status, end-of-track one tick after the last event.  Vectorised: a 3-minute track has ~11 k events and this
        runs under the GIL."""
        import numpy as np

        channels = [c for c in range(16) if c != 9]
        ch = 9 if inst.is_drum else channels[n % len(channels)]
        pitch, vel, start_s, end_s, bend, bend_s = inst._event_arrays()
        nn, nb = pitch.size, bend.size

        def ticks_of(times) -> "np.ndarray":
            # time_to_tick: 0 unless time > 0, else int(round(time / tick_scale)) (round half to even, like np.rint)
            t = np.asarray(times, dtype=np.float64)
            return np.where(t > 0, np.rint(t / self._tick_scale), 0.0).astype(np.int64)

        t_on, t_off, t_bend = ticks_of(start_s), ticks_of(end_s), ticks_of(bend_s)
        bad = (bend < -8192) | (bend > 8191)
        if bad.any():
            raise ValueError(f"pitch bend {int(bend[bad][0])} outside [-8192, 8191]")
        v14 = bend + 8192

        m = 1 + 2 * nn + nb
        tick = np.empty(m, np.int64)
        key = np.empty(m, np.int64)      # classes as in pretty_midi's event_compare
        note = np.full(m, -1, np.int64)  # pitch of note events (the fix-up pass compares them)
        velo = np.full(m, -1, np.int64)
        status = np.empty(m, np.int64)
        d1 = np.empty(m, np.int64)
        d2 = np.full(m, -1, np.int64)    # -1: one data byte
        tick[0], key[0], status[0], d1[0] = 0, 6 << 16, 0xC0 | ch, inst.program
        on, off = slice(1, 1 + 2 * nn, 2), slice(2, 2 + 2 * nn, 2)
        tick[on], tick[off] = t_on, t_off
        key[on], key[off] = (10 << 16) + pitch * 256 + vel, (10 << 16) + pitch * 256
        note[on] = note[off] = pitch
        velo[on], velo[off] = vel, 0
        status[1 : 1 + 2 * nn] = 0x90 | ch
        d1[on] = d1[off] = pitch
        d2[on], d2[off] = vel, 0
        bs = slice(1 + 2 * nn, m)
        tick[bs], key[bs], status[bs], d1[bs], d2[bs] = t_bend, (7 << 16) + bend, 0xE0 | ch, v14 & 0x7F, v14 >> 7

        order = np.lexsort((key, tick))  # stable, like sorted(cmp_to_key(event_compare))
        tick, note, velo, status, d1, d2 = tick[order], note[order], velo[order], status[order], d1[order], d2[order]
        # the fix-up pass looks at the ORIGINAL neighbours; its swaps cannot overlap (the second event of a pair has
        # velocity 0, the first of the next pair must not)
        sw = np.nonzero((tick[:-1] == tick[1:]) & (note[:-1] >= 0) & (note[:-1] == note[1:]) & (velo[:-1] != 0)
                        & (velo[1:] == 0))[0]
        if sw.size:
            perm = np.arange(m)
            perm[sw], perm[sw + 1] = sw + 1, sw
            tick, status, d1, d2 = tick[perm], status[perm], d1[perm], d2[perm]

        delta = np.diff(tick, prepend=0)
        if (delta < 0).any() or (delta >= 1 << 28).any():
            raise ValueError("MIDI delta time out of range")
        nv = 1 + (delta >= 1 << 7) + (delta >= 1 << 14) + (delta >= 1 << 21)   # bytes of the variable-length quantity
        has_status = np.ones(m, bool)
        has_status[1:] = status[1:] != status[:-1]                               # running status
        has_d2 = d2 >= 0
        size = nv + has_status + 1 + has_d2
        start = np.cumsum(size) - size
        out = np.zeros(int(size.sum()) + 4, np.uint8)
        for j in range(4):  # VLQ byte j counted from the least significant group; written at start + nv - 1 - j
            sel = nv > j
            val = (delta[sel] >> (7 * j)) & 0x7F
            if j:
                val = val | 0x80
            out[start[sel] + nv[sel] - 1 - j] = val
        at = start + nv
        out[at[has_status]] = status[has_status]
        at = at + has_status
        out[at] = d1
        out[(at + 1)[has_d2]] = d2[has_d2]
        out[-4:] = (1, 0xFF, 0x2F, 0x00)  # end of track, one tick after the last event
        return _chunk(b"MTrk", out.tobytes())

    def to_bytes(self) -> bytes:
        tempo = int(6e7 / (60.0 / (self._tick_scale * self.resolution)))
        timing = [
            (0, None, b"\xff\x51\x03" + struct.pack(">I", tempo)[1:]),
            (0, None, b"\xff\x58\x04\x04\x02\x18\x08"),
            (1, None, b"\xff\x2f\x00"),
        ]
        tracks = [self._encode_track(timing)]
        for n, inst in enumerate(self.instruments):
            tracks.append(self._instrument_track(n, inst))
        return _chunk(b"MThd", struct.pack(">hhh", 1, len(tracks), self.resolution)) + b"".join(tracks)

    def write(self, filename) -> None:
        with open(filename, "wb") as f:
            f.write(self.to_bytes())
