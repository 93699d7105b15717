"""CPU oracle for the GANSynth hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy float64 + torch-CPU fp32/fp64) of the
reference algorithm on the hot path named by BASELINE.json: ops.py:149-348,
networks.py:6-290, models.py:22-89, spectral_ops.py:8-149 of skmhrk1209/GANSynth.
Each function cites the reference file:line it follows.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this
path and its arithmetic lives in TensorFlow 1.13.1 +
tensorflow_probability, which are not vendored under /root/reference and are not
installable in the build image.  The oracle is therefore pinned only by
(1) hand-derivable known-answer tests (tests/test_oracle_kat.py),
(2) the published TF-1.13 op semantics restated in the docstrings (SAME-pad
    asymmetry, conv2d_transpose = d conv / d input, HTK mel, periodic Hann,
    floor-mod unwrap, TF-form Adam), and
(3) a cross-check of two independent restatements (numpy direct-definition
    loops vs torch-CPU library ops).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (gansynth_amd/) never does.
"""
