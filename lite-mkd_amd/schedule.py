"""ONE object for the execution schedule and the arithmetic of the episode loop.

The schedule switches grew up as module-level flags in ops.py / trainloop.py / model/backbone/resnet.py (each introduced with the
measurement that justified it); bench.py flipped twenty of them from environment variables and the tests flipped others.  `Schedule`
names them all in one place: `Schedule.bench()` is what the benchmark times, `Schedule.serial()` the single-stream reference schedule
the parity tests compare it with, `with schedule.applied(): ...` installs one for a block and restores the previous state.
trainloop.make(config, schedule=...) / trainloop.train(..., schedule=...) take it; `python -m litemkd_amd.train` builds it from the
command line."""
import contextlib
import dataclasses
import os


_PIPELINE = [False]      # the installed schedule's loop-level choice (Schedule.apply / Schedule.current; read by trainloop.train)


@dataclasses.dataclass
class Schedule:
    # ---- arithmetic (process-wide in liblmkd_hip.so: lmkd_conv_set_compute_dtype / lmkd_set_activation_dtype)
    conv_dtype: str = "fp32x3"             # fp32x3 (default: fp32 as 3 x bf16 on the matrix pipe) | fp32h2 (bench.py: + the 3x3 kernels on 2 x fp16) | fp32 (native MFMA) | bf16 | fp32x3_9
    act_dtype: str = "fp32"                # storage of the trunk's activations: fp32 | bf16 (BASELINE configs[2], needs conv_dtype bf16)
    # ---- streams
    overlap_trunk_calls: bool = True       # support / query trunk call on two streams (resnet.OVERLAP_TRUNK_CALLS; where the calls are not merged)
    merge_trunk_calls: bool = False        # both calls as one launch per layer (resnet.MERGE_TRUNK_CALLS; bf16-plane modes)
    pipeline_episodes: bool = False        # the forward of episode i + 1 beside the backward of episode i (trainloop.PipelinedEpisodes): a
                                           # property of the LOOP (trainloop.train / bench.py), not a module switch
    side_wgrad: bool = True                # convolution weight gradients on their own stream, accumulated into .grad there (ops.SIDE_WGRAD)
    sync_wgrad_at_backward_end: bool = False   # False: the optimizer joins the weight-gradient stream itself (FusedOptimizer)
    teacher_stream: bool = True            # frozen teacher head on an auxiliary stream (trainloop.TEACHER_STREAM)
    heads_on_two_streams: bool = True      # second TRX head beside the first (ops.HEADS_ON_TWO_STREAMS)
    side_linear_wgrad: bool = True         # Linear / TRX weight-gradient GEMMs on the weight-gradient stream (ops.SIDE_LINEAR_WGRAD)
    # ---- gradient plumbing / fusion
    direct_param_grad: bool = True         # BatchNorm / Linear / TRX parameter gradients added into .grad by the kernels (ops.DIRECT_PARAM_GRAD)
    repack_at_step: bool = True            # all weight packs re-packed by one launch at the optimizer step (trainloop.REPACK_AT_STEP)
    fuse_two_head_linear: bool = True      # fc1 / fc2 of both trunk calls as one autograd node (ops.FUSE_TWO_HEAD_LINEAR)
    dgrad_bn_stats: bool = True            # BatchNorm-backward sums in the data gradient's epilogue (ops.DGRAD_BN_STATS)
    stem_pooled_bwd: bool = True           # the stem's BatchNorm backward from the pooled side (ops.STEM_POOLED_BWD)
    pre_in_plane_modes: bool = True        # inner BatchNorm + ReLU in the consumers' loaders, three-plane modes (ops.PRE_IN_PLANE_MODES)
    fuse_pre_all_modes: bool = False       # ... forced in every mode (ops.FUSE_PRE_ALL_MODES)
    trx_proj_on_conv: bool = False         # TRX projections as 1x1 convolutions (ops.TRX_PROJ_ON_CONV)
    gemm_split_k: bool = False             # split-K for the head's small GEMMs (ops.GEMM_SPLIT_K)

    @classmethod
    def bench(cls, **over):
        """the schedule bench.py times.  Round 4: ONE launch per layer for both trunk calls + cross-episode pipelining - three streams
        busy throughout (forward of episode i + 1 | backward chain of episode i | its weight gradients) with half the launches:
        37.5 episodes/s against 37.0 for round 3's schedule (two trunk calls on two streams + weight-gradient stream, no pipelining =
        Schedule.two_call()), 36.2 merged without pipelining, 35.0 two-call with it (same box, profiles/r04_pipe_ab.txt)"""
        kw = dict(merge_trunk_calls=True, pipeline_episodes=True)
        kw.update(over)
        return cls(**kw)

    @classmethod
    def two_call(cls, **over):
        """round 3's benchmark schedule: the two trunk calls on two streams, weight gradients on a third, no pipelining (the fastest
        schedule where the arithmetic has no two-segment kernels: the native fp32 MFMA mode)"""
        return cls(**over)

    @classmethod
    def serial(cls, **over):
        """everything on the caller's stream, every parameter gradient through autograd: the reference schedule of the parity tests"""
        kw = dict(overlap_trunk_calls=False, merge_trunk_calls=False, side_wgrad=False, sync_wgrad_at_backward_end=True, teacher_stream=False,
                  heads_on_two_streams=False, side_linear_wgrad=False, direct_param_grad=False, repack_at_step=False, pipeline_episodes=False)
        kw.update(over)
        return cls(**kw)

    @classmethod
    def from_env(cls, env=None, **over):
        """bench.py's LMKD_* tuning variables (1 / 0) on top of the defaults"""
        env = os.environ if env is None else env
        names = {"overlap_trunk_calls": "LMKD_OVERLAP", "merge_trunk_calls": "LMKD_MERGE", "pipeline_episodes": "LMKD_PIPELINE", "side_wgrad": "LMKD_SIDE_WGRAD",
                 "sync_wgrad_at_backward_end": "LMKD_SYNC_WG", "teacher_stream": "LMKD_TEACHER_STREAM", "heads_on_two_streams": "LMKD_HEADS2",
                 "side_linear_wgrad": "LMKD_SIDE_LINEAR", "direct_param_grad": "LMKD_DIRECT_GRAD", "repack_at_step": "LMKD_REPACK",
                 "fuse_two_head_linear": "LMKD_FC_FUSED", "dgrad_bn_stats": "LMKD_DGRAD_BN", "stem_pooled_bwd": "LMKD_STEM_POOLED",
                 "pre_in_plane_modes": "LMKD_PRE_X3", "fuse_pre_all_modes": "LMKD_FUSE_PRE", "trx_proj_on_conv": "LMKD_TRX_CONV",
                 "gemm_split_k": "LMKD_GEMM_SPLIT"}
        s = cls.bench(**over)
        for field, var in names.items():
            if var in env and field not in over:
                setattr(s, field, env[var] == "1")
        return s

    # field -> (module path, attribute)
    _WHERE = {
        "overlap_trunk_calls": ("model.backbone.resnet", "OVERLAP_TRUNK_CALLS"), "merge_trunk_calls": ("model.backbone.resnet", "MERGE_TRUNK_CALLS"),
        "side_wgrad": ("ops", "SIDE_WGRAD"), "sync_wgrad_at_backward_end": ("ops", "SYNC_WGRAD_AT_BACKWARD_END"),
        "teacher_stream": ("trainloop", "TEACHER_STREAM"), "heads_on_two_streams": ("ops", "HEADS_ON_TWO_STREAMS"),
        "side_linear_wgrad": ("ops", "SIDE_LINEAR_WGRAD"), "direct_param_grad": ("ops", "DIRECT_PARAM_GRAD"),
        "repack_at_step": ("trainloop", "REPACK_AT_STEP"), "fuse_two_head_linear": ("ops", "FUSE_TWO_HEAD_LINEAR"),
        "dgrad_bn_stats": ("ops", "DGRAD_BN_STATS"), "stem_pooled_bwd": ("ops", "STEM_POOLED_BWD"),
        "pre_in_plane_modes": ("ops", "PRE_IN_PLANE_MODES"), "fuse_pre_all_modes": ("ops", "FUSE_PRE_ALL_MODES"),
        "trx_proj_on_conv": ("ops", "TRX_PROJ_ON_CONV"), "gemm_split_k": ("ops", "GEMM_SPLIT_K"),
    }

    def _targets(self):
        import importlib
        pkg = __name__.rsplit(".", 1)[0]
        for field, (mod, attr) in self._WHERE.items():
            yield field, importlib.import_module(pkg + "." + mod), attr

    def apply(self, arithmetic=True):
        """install this schedule process-wide (arithmetic=False: the switches only, the arithmetic mode stays as it is)"""
        from . import ops
        for field, mod, attr in self._targets():
            setattr(mod, attr, bool(getattr(self, field)))
        _PIPELINE[0] = bool(self.pipeline_episodes)
        if arithmetic:
            ops.set_conv_compute_dtype(self.conv_dtype)
            ops.set_activation_dtype(self.act_dtype)
        return self

    @classmethod
    def current(cls):
        """the schedule that is installed right now"""
        from . import ops
        s = cls(conv_dtype=ops.get_conv_compute_dtype(), act_dtype=ops.get_activation_dtype(), pipeline_episodes=_PIPELINE[0])
        for field, mod, attr in s._targets():
            setattr(s, field, bool(getattr(mod, attr)))
        return s

    @contextlib.contextmanager
    def applied(self):
        from . import ops
        prev = Schedule.current()
        ops.join_all_streams()
        self.apply()
        try:
            yield self
        finally:
            ops.join_all_streams()
            prev.apply()
