"""NSynth input pipeline on the host (reference dataset.py:12-91), without TensorFlow.

`nsynth_input_fn(filenames, batch_size, num_epochs, shuffle, buffer_size, pitches, sources)` keeps the reference's
signature and semantics: every record names a 16-bit PCM WAV file, its pitch and its instrument source; records are
(optionally) shuffled with a buffer, repeated `num_epochs` times, decoded to 64000 mono samples in [-1, 1), filtered to
`min(pitches) <= pitch <= max(pitches)` and `source in sources`, labelled with the one-hot index of the pitch in
`sorted(pitches)`, batched with the remainder dropped, and prefetched.  It returns a zero-argument callable -- the eager
counterpart of `iterator.get_next()` -- that yields `(waveforms [B, 64000] float32, labels [B, len(pitches)] float32)`
and raises `StopIteration` when the data is exhausted (tf.errors.OutOfRangeError in the reference, models.py:193).

Record sources: the reference's own `*.tfrecord` files (read by the small TFRecord / tf.train.Example wire-format
parser below -- features path / pitch / source, dataset.py:19-25), or NSynth's `examples.json` index next to an `audio/`
directory.  `synthetic_nsynth_input_fn` produces notes of the same shapes and ranges without any file (bench, tests).
Decoding runs in a background thread (the step consumes a batch in ~4 ms; a WAV decode is host work) into pinned
buffers, so the copy to the GPU overlaps the previous step.
"""
import json
import os
import queue
import struct
import threading
import wave

import numpy as np
import torch

WAVEFORM_LENGTH = 64000   # dataset.py:35 desired_samples


# ------------------------------------------------------------------------------ record readers
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) triples of one protobuf message (wire types 0 and 2 are all tf.train.Example uses)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def parse_example(serialized):
    """tf.train.Example -> {name: bytes | int | float | list}: Example{1: Features{1: map<string, Feature>}}; Feature is a oneof
    of BytesList(1) / FloatList(2) / Int64List(3), each with repeated field 1 (int64 / float lists may be packed)."""
    out = {}
    for num, _, features in _fields(serialized):
        if num != 1:
            continue
        for fnum, _, entry in _fields(features):
            if fnum != 1:
                continue
            name, feature = None, b""
            for enum, _, val in _fields(entry):
                if enum == 1:
                    name = bytes(val).decode()
                elif enum == 2:
                    feature = val
            values = []
            for kind, _, lst in _fields(feature):
                for vnum, wt, val in _fields(lst):
                    if vnum != 1:
                        continue
                    if kind == 1:
                        values.append(bytes(val))
                    elif kind == 3:
                        if wt == 0:
                            values.append(val - (1 << 64) if val >= (1 << 63) else val)
                        else:   # packed
                            p = 0
                            while p < len(val):
                                v, p = _varint(val, p)
                                values.append(v - (1 << 64) if v >= (1 << 63) else v)
                    elif kind == 2:
                        values.extend(struct.unpack(f"<{len(val) // 4}f", bytes(val)) if wt == 2 else struct.unpack("<f", bytes(val)))
            out[name] = values[0] if len(values) == 1 else values
    return out


def tfrecord_iterator(filename):
    """Serialized records of a TFRecord file: [u64 length][u32 crc][data][u32 crc] (the CRCs are not verified)."""
    with open(filename, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                return
            (n,) = struct.unpack("<Q", head[:8])
            data = f.read(n)
            f.read(4)
            if len(data) < n:
                return
            yield data


def _records(filename):
    """(path, pitch, source) triples of one record file."""
    if filename.endswith(".json"):   # NSynth's examples.json: {note name: {"pitch": .., "instrument_source": ..}}, audio/<name>.wav
        base = os.path.join(os.path.dirname(os.path.abspath(filename)), "audio")
        with open(filename) as f:
            index = json.load(f)
        for name in sorted(index):
            meta = index[name]
            yield os.path.join(base, name + ".wav"), int(meta["pitch"]), int(meta.get("instrument_source", meta.get("source", 0)))
        return
    for rec in tfrecord_iterator(filename):
        ex = parse_example(rec)
        yield ex["path"].decode(), int(ex["pitch"]), int(ex["source"])


def decode_wav(path, desired_samples=WAVEFORM_LENGTH):
    """audio_ops.decode_wav(desired_channels=1, desired_samples=64000) (dataset.py:32-37): 16-bit PCM scaled by 1/32768, first
    channel, cropped or zero-padded to `desired_samples`."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: decode_wav handles 16-bit PCM only")
        channels = w.getnchannels()
        pcm = np.frombuffer(w.readframes(min(w.getnframes(), desired_samples)), dtype="<i2")
    if channels > 1:
        pcm = pcm.reshape(-1, channels)[:, 0]
    out = np.zeros(desired_samples, dtype=np.float32)
    out[:len(pcm)] = pcm.astype(np.float32) * np.float32(1.0 / 32768.0)
    return out


# ---------------------------------------------------------------------------------- pipeline
class _Prefetcher(object):
    """Background producer of (waveforms, labels) batches; `__call__` is the eager `iterator.get_next()`."""

    def __init__(self, make_batches, depth, device):
        self._q = queue.Queue(maxsize=max(1, depth))
        self._device = device
        self._done = False
        self._thread = threading.Thread(target=self._work, args=(make_batches,), daemon=True)
        self._thread.start()

    def _work(self, make_batches):
        try:
            for wav, lab in make_batches():
                wav_t, lab_t = torch.from_numpy(wav), torch.from_numpy(lab)
                if self._device is not None and torch.device(self._device).type == "cuda":
                    wav_t, lab_t = wav_t.pin_memory(), lab_t.pin_memory()
                self._q.put((wav_t, lab_t))
            self._q.put(None)
        except BaseException as e:   # surfaces in the consumer
            self._q.put(e)

    def __call__(self):
        if self._done:
            raise StopIteration
        item = self._q.get()
        if item is None:
            self._done = True
            raise StopIteration
        if isinstance(item, BaseException):
            self._done = True
            raise item
        wav, lab = item
        if self._device is not None:
            wav, lab = wav.to(self._device, non_blocking=True), lab.to(self._device, non_blocking=True)
        return wav, lab


def nsynth_input_fn(filenames, batch_size, num_epochs, shuffle, buffer_size=None, pitches=None, sources=None, device=None, seed=0,
                    prefetch=2):
    """dataset.py:12-91.  -> callable returning (waveforms [B, 64000], labels [B, len(pitches)]); StopIteration at the end."""
    filenames = [filenames] if isinstance(filenames, str) else list(filenames)
    if not filenames:
        raise ValueError("nsynth_input_fn: no input files")
    if not pitches:
        raise ValueError("nsynth_input_fn: `pitches` is required (the label table, dataset.py:15)")
    table = {p: i for i, p in enumerate(sorted(pitches))}
    lo, hi = min(pitches), max(pitches)
    src_ok = None if not sources else set(int(s) for s in sources)

    def make_batches():
        records = [r for fn in filenames for r in _records(fn)]
        rng = np.random.default_rng(seed)
        epoch = 0
        wavs, labs = [], []
        while num_epochs is None or epoch < num_epochs:
            order = np.arange(len(records))
            if shuffle:   # buffer_size=None = a full shuffle (dataset.py:52-55), reshuffled each epoch
                if buffer_size is None or buffer_size >= len(records):
                    rng.shuffle(order)
                else:     # tf.data's streaming shuffle: a buffer of `buffer_size` records, random pick, refill
                    buf, out, it = list(order[:buffer_size]), [], iter(order[buffer_size:])
                    while buf:
                        j = int(rng.integers(len(buf)))
                        out.append(buf[j])
                        nxt = next(it, None)
                        if nxt is None:
                            buf.pop(j)
                        else:
                            buf[j] = nxt
                    order = np.asarray(out)
            for k in order:
                path, pitch, source = records[k]
                if not (lo <= pitch <= hi) or (src_ok is not None and source not in src_ok) or pitch not in table:
                    continue
                wavs.append(decode_wav(path))
                lab = np.zeros(len(table), dtype=np.float32)
                lab[table[pitch]] = 1.0
                labs.append(lab)
                if len(wavs) == batch_size:
                    yield np.stack(wavs), np.stack(labs)
                    wavs, labs = [], []
            epoch += 1
            if not records:
                break
        # drop_remainder=True (dataset.py:82): a trailing partial batch is discarded

    fn = _Prefetcher(make_batches, prefetch, device)
    fn.finite = num_epochs is not None   # (models.GANSynth.train: data-parallel ranks vote before each step when the input can run dry)
    return fn


def synthetic_nsynth_input_fn(batch_size, pitches=range(24, 85), num_batches=None, device=None, seed=0, prefetch=2):
    """Notes with NSynth's shapes and ranges and no files: MIDI pitch p -> f = 440 * 2^((p - 69) / 12) with a few harmonics, an
    attack / decay envelope, a little noise; peak amplitude < 1.  Same return convention as nsynth_input_fn."""
    pitches = sorted(pitches)

    def make_batches():
        rng = np.random.default_rng(seed)
        t = np.arange(WAVEFORM_LENGTH, dtype=np.float32) / np.float32(16000.0)
        k = 0
        while num_batches is None or k < num_batches:
            idx = rng.integers(0, len(pitches), size=batch_size)
            wav = np.zeros((batch_size, WAVEFORM_LENGTH), dtype=np.float32)
            for b, i in enumerate(idx):
                f0 = 440.0 * 2.0 ** ((pitches[i] - 69) / 12.0)
                env = (1.0 - np.exp(-t / 0.02)) * np.exp(-t / rng.uniform(0.4, 2.0))
                sig = sum(rng.uniform(0.2, 1.0) / h * np.sin(2.0 * np.pi * f0 * h * t + rng.uniform(0, 2 * np.pi))
                          for h in range(1, 5) if f0 * h < 8000.0)
                sig = sig * env + rng.normal(0.0, 0.002, WAVEFORM_LENGTH)
                wav[b] = (0.7 * sig / (np.abs(sig).max() + 1e-9)).astype(np.float32)
            lab = np.zeros((batch_size, len(pitches)), dtype=np.float32)
            lab[np.arange(batch_size), idx] = 1.0
            yield wav, lab
            k += 1

    fn = _Prefetcher(make_batches, prefetch, device)
    fn.finite = num_batches is not None
    return fn
