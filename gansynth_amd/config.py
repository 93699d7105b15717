"""Every environment switch of the Python layer, in one place.

The switches are read through `flag()` / `value()` only, and only names registered in KNOBS are accepted: a typo or an undocumented
switch is an error at import time, and `tests/test_abi_cpu.py::test_every_environment_switch_is_registered` keeps the table complete.
Three kinds:
  operational  -- a deployment may need it (transport, probes, debugging aids);
  schedule     -- alternative schedules that were built, are parity-tested and measured, and are not the default (the number that decided it
                  is in the description; DESIGN.md has the measurement);
  ablation     -- "GS_NO_*": switches ONE optimisation off so that its gain can be re-measured on a new box / ROCm build
                  (scripts/ab_multi.sh).  None of them changes results beyond fp32 association.
The C++ side reads its own tuning switches (tile / grid overrides of the kernels, `getenv` in csrc/): measurement only, listed in
scripts/README.md.
"""
import os

KNOBS = {
    # ---------------------------------------------------------------- operational
    "GS_TORCH_COLLECTIVES": ("operational", "gradient all-reduce through torch.distributed's communicator instead of the library's own (gs_comm_*)"),
    "GS_NO_GRAPH_ALLREDUCE": ("operational", "no RCCL collective inside captured graphs: eager all-reduce behind each replay (bench.DP_LADDER's tamer modes)"),
    "GS_FORK_PROBE": ("operational", "auto | always | never: probe a forked hipGraph replay in a child process before the first forked capture "
                                     "(auto: only on HIP builds the workaround was not debugged on, models._forked_replay_ok)"),
    "GS_FORK_PROBED": ("operational", "(set by the probe for child processes: ok | died)"),
    "GS_FORK_PROBE_TIMEOUT_S": ("operational", "budget of that probe, seconds (600)"),
    "GS_LEVEL_STREAMS": ("operational", "throw-away streams alive while a graph is instantiated (128; models.GANSynth._leveled_queues)"),
    "GS_COMM_MARKER_US": ("operational", "tests / profiles at world size 1: the one-rank all-reduce becomes a kernel that holds its stream this long "
                                         "(read once per communicator, comm.RcclComm; gs_comm_set_marker_us)"),
    "GS_CAPTURE_MODE": ("operational", "debugging: force torch.cuda.graph's capture_error_mode"),
    "GS_DEBUG_POISON_WS": ("operational", "debugging: kernel workspaces start as NaN bit patterns"),
    "GS_CHECK_FUSION": ("operational", "debugging: every fused cross-node form also runs its unfused definition and the two are compared"),
    "GS_FORK_EAGER": ("operational", "tests: the forked branches with eager launches on two streams"),
    "GS_NO_FORK_MARKS": ("operational", "debugging: branches start where they are opened, not at their marks"),
    # ---------------------------------------------------------------- schedule alternatives (built, tested, measured, not the default)
    "GS_NO_FORK": ("schedule", "no parallel branches in the runs' graphs (5.9 against 5.2 ms; also taken when GPU_MAX_HW_QUEUES != 4 or the probe fails)"),
    "GS_NO_MERGED_RUNS": ("schedule", "part A of the generator run NOT inside the discriminator run's graph (5.13 against 4.99 ms)"),
    "GS_NO_FUSED_ITERATION": ("schedule", "two graphs per iteration with eager optimizer steps between them (neutral on one GPU; data parallel both "
                                          "all-reduces are then the last node of a graph)"),
    "GS_OVERLAP_REDUCE": ("schedule", "data parallel: round 4's four-graph iteration, each all-reduce on a forked branch beside part A of the other run "
                                      "(hides both collectives; +0.7 ms at world size 1)"),
    "GS_NO_OVERLAP_REDUCE": ("schedule", "(overrides GS_OVERLAP_REDUCE)"),
    "GS_FORK_DIST": ("schedule", "compute branches also in the four-graph data-parallel iteration (host-bound: 7.4 ms of replay calls)"),
    "GS_PIPELINE": ("schedule", "round 2's pipelined iteration with the update on a side stream (cross-stream hops between replays: -6 %)"),
    "GS_PIPE_SIDE": ("schedule", "0 | 1: the side stream of that form"),
    "GS_DEBUG_DP_BUCKET": ("operational", "print the range the first message of a two-step discriminator all-reduce covers"),
    "GS_DP_BUCKET_D": ("schedule", "data parallel: the discriminator's gradient all-reduced in two steps, ~98 % of it beside the end of its backward (300-us stand-ins: -0.09 ... -0.14 ms fully grown, +0.15 in a fade-in)"),
    "GS_SUB_RUNS": ("schedule", "the discriminator run as two independent sub-runs: split loss launches, two backward calls (4.91 -> 5.33 ms)"),
    "GS_FAKE_FIRST": ("schedule", "the discriminator run's fake pass issued before the real pass (+0.04 ms; hides 0.13 ms of a 0.3 ms all-reduce "
                                  "stand-in in one process, none in another)"),
    "GS_EARLY_FLUSH_DIV": ("schedule", "a layer is 'large' from 1/DIV of the full resolution's pixels (16)"),
    "GS_EARLY_FLUSH_CUS": ("schedule", "CUs the early weight-gradient contraction is sized for (192; 160 / 224 / 256 within noise, round 6)"),
    "GS_D_TAIL_LEVELS": ("schedule", "levels of the discriminator's tail run over [real; fake] as one batch when the runs do not fork (3)"),
    # ---------------------------------------------------------------- ablations
    "GS_NO_EARLY_FLUSH": ("ablation", "weight gradients of the full-chip levels contracted at the end of the run only"),
    "GS_NO_DEFERRED_REDUCE": ("ablation", "every weight gradient contracted where autograd produces it"),
    "GS_NO_DEFERRED_FOLDS": ("ablation", "bias-gradient partial rows folded by their producers"),
    "GS_NO_WGRAD_GROUPS": ("ablation", "no grouped weight-gradient launches"),
    "GS_NO_SPLIT_G_LOSS": ("ablation", "the generator's loss as one launch over both halves: the mode-seeking second-order pass waits for the discriminator's forward"),
    "GS_NO_SPLIT_FINAL_FLUSH": ("ablation", "the final weight-gradient contraction of a run on one stream (HBM-bound and MFMA-bound jobs one after the other)"),
    "GS_NO_FUSED_LOSSES": ("ablation", "per-sample loss algebra in torch instead of the one-launch loss heads"),
    "GS_NO_D_TAIL_BATCH": ("ablation", "real and fake pass through the discriminator's tail separately (no-fork schedule)"),
    "GS_NO_FUSED_NORM": ("ablation", "pixel norm as its own node behind every generator conv"),
    "GS_NO_NORM_EPILOGUE": ("ablation", "no pixel norm in conv epilogues"),
    "GS_NO_NORM_BWD_EPILOGUE": ("ablation", "no previous-block norm backward in data-gradient epilogues"),
    "GS_NO_NORM_BWD2_EPILOGUE": ("ablation", "no second-order norm kernel in the forward-on-cotangent conv"),
    "GS_NO_NORM_BWD_BIAS": ("ablation", "bias sums outside the norm's backward"),
    "GS_NO_MASK_BITS": ("ablation", "leaky-relu masks read as the bf16 activations themselves, not as the sign bits stored behind them"),
    "GS_NO_PREMASK": ("ablation", "activation derivative never folded into the consuming data-gradient kernel"),
    "GS_NO_PREMASK_GRAPH": ("ablation", "... not across autograd nodes"),
    "GS_NO_PREMASK_GRAPH2": ("ablation", "... not in second-order graphs"),
    "GS_NO_PARAMS_ONLY": ("ablation", "leaf activations receive gradients like tf.gradients would not ask for"),
    "GS_NO_DERIVED_SLICES": ("ablation", "the 257-channel conv's weight slices copied on every use"),
    "GS_NO_UNITS_NHWC": ("ablation", "dense -> reshape -> activation as three steps"),
    "GS_NO_DENSE_NHWC": ("ablation", "flatten copy + plain dense kernels"),
}


def _check(name):
    if name not in KNOBS:
        raise KeyError("gansynth_amd.config: %s is not a registered environment switch (add it to KNOBS with its description)" % name)


def flag(name):
    """True when the switch is set to anything non-empty."""
    _check(name)
    return bool(os.environ.get(name))


def value(name, default=None):
    _check(name)
    return os.environ.get(name, default)
