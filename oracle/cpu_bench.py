"""CPU timing of the numpy spectral oracle on all host cores (bench.py's `spectral.cpu_baseline` leg).

Test / measurement infrastructure (see oracle/__init__.py): run as its own process so that the worker pool can be forked from
an interpreter that holds no HIP state:

    python -m oracle.cpu_bench --examples 256 --procs 64

One worker process per core (BLAS pinned to one thread each), every worker converts whole examples with
oracle.spectral_np.convert_to_spectrogram (spectral_ops.py:45-94 restated); the timed region is the pool.map over all examples
after every worker has converted one warm-up example.  Prints one JSON object.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)
_W = None


def _init(seed, examples):
    global _W
    import numpy as np
    rng = np.random.default_rng(seed)
    _W = np.clip(rng.normal(0.0, 0.1, (examples, 64000)), -1, 1).astype(np.float32)   # SURVEY.md 8(d) synthetic waveforms
    from oracle import spectral_np as S
    S.convert_to_spectrogram(_W[:1], **P)


def _work(span):
    from oracle import spectral_np as S
    lo, hi = span
    lm, mi = S.convert_to_spectrogram(_W[lo:hi], **P)
    return float(lm.sum()) + float(mi.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--examples", type=int, default=256)
    ap.add_argument("--procs", type=int, default=0, help="worker processes (0: one per host core, at most one per example)")
    ap.add_argument("--chunk", type=int, default=1)
    args = ap.parse_args()
    total = os.cpu_count() or 1
    procs = args.procs or total
    procs = max(1, min(procs, (args.examples + args.chunk - 1) // args.chunk))
    spans = [(i, min(i + args.chunk, args.examples)) for i in range(0, args.examples, args.chunk)]
    with mp.get_context("fork").Pool(procs, initializer=_init, initargs=(4000, args.examples)) as pool:
        pool.map(_work, [(0, 1)] * procs)   # every worker is up (its initializer has run) before the clock starts
        t0 = time.time()
        checks = pool.map(_work, spans, chunksize=1)
        dt = time.time() - t0
    print(json.dumps({"examples": args.examples, "seconds": dt, "procs": procs, "host_cores": total, "checksum": sum(checks)}))


if __name__ == "__main__":
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
