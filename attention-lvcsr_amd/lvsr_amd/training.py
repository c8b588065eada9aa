"""Thin trainer around SpeechRecognizer: one data-parallel SGD step of the reference's recipe
(lvsr/main.py:340-345 cost = cost_matrix.sum()/batch_size; :480-519 step rules; GradientDescent.process_batch,
libs/blocks/blocks/algorithms/__init__.py:284-287).

Data parallelism (new; the reference is single-device): every rank holds a full replica, processes its shard of
the utterances (rank r takes utterances r::world), gradients of the summed cost are all-reduced (sum) in ONE
RCCL collective over the flat gradient buffer, then divided by the GLOBAL batch size inside the fused optimiser
kernel; all ranks apply identical updates.
"""
import ctypes
import logging
import os

import numpy
import torch

from .native import ptr


def _is_weight(name):
    return name.endswith(".W") or name.endswith("state_to_state") or name.endswith("state_to_gates")


logger = logging.getLogger(__name__)


def _is_sync_word(key):
    """Workspace names of the cluster kernels' scratch (abort word first): decoder, encoder layers (and their passes)."""
    return key[0] in ("gen.sync", "gen.sync_bwd") or (key[0].startswith("enc") and key[0].endswith(".sync"))


class Trainer(object):
    # What recover() does about an aborted cluster launch (a work-group of a cluster was not resident: something else holds CUs).
    # First abort: the cluster kernels stay, `cluster_reserve` CUs are left free (the knob of csrc/runtime.hip; cluster shapes that no
    # longer fit fall to the next smaller shape: +9 % per WSJ-base step) — for REARM_STEPS clean steps, then the reserve is given back.
    # Another abort within REARM_STEPS steps: encoder and decoder move to the step kernels — for REARM_STEPS clean steps, after which
    # the cluster kernels are armed again (with the reserve, which is given back REARM_STEPS clean steps later).
    RECOVER_RESERVE = 32
    REARM_STEPS = 200
    # CUs the cluster launches leave free when an RCCL kernel can be co-resident with them: only with overlap_allreduce (without it
    # the collective is stream-ordered BETWEEN the backward pass and the optimiser: it never shares the device with a cluster launch)
    OVERLAP_RESERVE = 32

    def __init__(self, recognizer, gradient_threshold=None, rules=("momentum",), scale=0.1, momentum=0.0,
                 decay_rate=0.95, epsilon=1e-8, max_norm=0.0, max_norm_exclude_lookup=False, nonfinite_scaler=0.0,
                 burn_in_steps=0, adaptive_clipping=None, process_group=None, distributed=None, dp_region=True,
                 overlap_allreduce=False):
        """Keywords = `training:` / `regularization:` keys of the reference's config (lvsr/main.py:480-519).
        `adaptive_clipping`: None, True or dict(decay_rate=0.998, burnin_period=500) — the AdaptiveClipping extension the
        reference's `train()` always installs on top of `gradient_threshold` (lvsr/main.py:616-619)."""
        self.rec = recognizer
        self._token = recognizer.lib.unique_token()        # names this trainer's buffers in graph-region keys
        st = recognizer.store
        dev = st.device
        self.conf = dict(clip_threshold=float(gradient_threshold or 0.0), use_momentum=int("momentum" in rules),
                         use_adadelta=int("adadelta" in rules), learning_rate=float(scale), momentum=float(momentum),
                         decay_rate=float(decay_rate), epsilon=float(epsilon), max_norm=float(max_norm or 0.0),
                         remove_not_finite=1, nonfinite_scaler=float(nonfinite_scaler))
        n = st.flat.numel()
        z = lambda: torch.zeros(n, dtype=torch.float32, device=dev)
        self.velocity = z() if self.conf["use_momentum"] else None
        self.ms_step = z() if self.conf["use_adadelta"] else None
        self.ms_dx = z() if self.conf["use_adadelta"] else None
        self.step_buf = z()
        seg, max_cols = [], 1
        for name, (off, cnt) in st.offsets.items():
            shape = st.shapes[name]
            rows, cols = (int(shape[0]), int(numpy.prod(shape[1:]))) if len(shape) >= 2 else (1, int(cnt))
            flag = int(self.conf["max_norm"] > 0 and _is_weight(name) and len(shape) == 2
                       and not (max_norm_exclude_lookup and "lookuptable" in name))
            if flag:
                max_cols = max(max_cols, cols)
            seg.append([off, rows, cols, flag])
        self.max_cols = max_cols
        self.segments = torch.tensor(seg, dtype=torch.int64, device=dev)
        self.segflag = torch.zeros(len(seg), dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(2 + 256, dtype=torch.float32, device=dev)
        self.clip_state = None
        if adaptive_clipping or burn_in_steps:
            ac = dict(decay_rate=0.998, burnin_period=500)
            if isinstance(adaptive_clipping, dict):
                ac.update(adaptive_clipping)
            if adaptive_clipping and not self.conf["clip_threshold"] > 0:
                raise ValueError("adaptive clipping needs gradient_threshold (its initial threshold)")
            self.clip_state = torch.tensor([self.conf["clip_threshold"], 0.0, 0.0, 0.0, float(burn_in_steps or 0), 0, 0, 0],
                                           dtype=torch.float64, device=dev)
            self.conf.update(adaptive_clipping=int(bool(adaptive_clipping)), adaptive_burnin=int(ac["burnin_period"]),
                             adaptive_decay=float(ac["decay_rate"]))
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.distributed = distributed
        self.group = process_group
        self.world = torch.distributed.get_world_size(process_group) if distributed else 1
        self.rank = torch.distributed.get_rank(process_group) if distributed else 0
        # data parallel: forward + backward of a minibatch shape replay as ONE hipGraph launch, the all-reduce and the fused
        # optimiser follow eagerly on the same stream (dp_region=False: per-layer graphs + eager launches instead)
        self.dp_region = bool(dp_region)
        # overlap_allreduce: two buckets instead of one — the decoder's gradients (the tail of the flat buffer, final when the decoder's
        # reverse walk is over) are reduced on the communication stream WHILE the encoder's BPTT runs, the encoder's after it
        # (SURVEY.md 2.3 C1).  Off by default: it cannot be measured on the one-GPU boxes this was developed on, the exchange is
        # ~2 % of a step, and a collective's work-groups share the CUs with latency-bound cluster kernels whose work-groups must
        # all be resident (what concurrent GEMMs did to them is in bricks/__init__.py: slower, not faster).
        self.overlap_allreduce = bool(overlap_allreduce)
        self._comm = None
        # (the knob is process-global state of the library: what this trainer changed is given back by close())
        self._overlap_reserve_before = None
        if self.overlap_allreduce and self.world > 1 and recognizer.lib.get_knob("cluster_reserve") == 0:
            self._overlap_reserve_before = 0
            recognizer.lib.set_knob("cluster_reserve", self.OVERLAP_RESERVE)
        # host mirror of scratch[3] ("the optimiser skipped the last step"), copied behind every optimiser step: train_step looks at
        # it (no synchronisation) before it enqueues the next step and refuses to go on over a skipped step nobody recovered from
        self._skip_host = torch.zeros(1, dtype=torch.float32).pin_memory() if dev.type == "cuda" else None
        self.aborts = 0               # cluster launches that gave up so far (recover() calls)
        self._fallback = None         # while on the step kernels: dict(clean=steps since, saved=kernel choices to restore)
        self._reserve_before = None   # while a recovery's reserve is in force: the value of the knob to give back
        self._last_abort_step = None
        self.steps_done = 0

    @classmethod
    def from_config(cls, recognizer, training, regularization=None, adaptive_clipping=True, **kw):
        """Build the step rules the way lvsr/main.py:480-519 reads `config['training']` / `config['regularization']`."""
        reg = regularization or {}
        return cls(recognizer, gradient_threshold=training.get("gradient_threshold"), rules=tuple(training.get("rules", ["momentum"])),
                   scale=training.get("scale", 0.1), momentum=training.get("momentum", 0.0),
                   decay_rate=training.get("decay_rate", 0.95), epsilon=training.get("epsilon", 1e-8),
                   max_norm=reg.get("max_norm", 0.0) or 0.0, max_norm_exclude_lookup=reg.get("max_norm_exclude_lookup", False),
                   burn_in_steps=training.get("burn_in_steps", 0),
                   adaptive_clipping=adaptive_clipping and bool(training.get("gradient_threshold")), **kw)

    # ---- what a restart needs besides the parameters (the reference pickles the whole main loop, serialization.py:145-262) --
    def state_dict(self):
        """AdaDelta / momentum accumulators (flat, in parameter order), the adaptive-clipping statistics and burn-in counter."""
        out = {}
        for name in ("velocity", "ms_step", "ms_dx"):
            t = getattr(self, name)
            if t is not None:
                out[name] = t.detach().cpu().numpy()
        if self.clip_state is not None:
            out["clip_state"] = self.clip_state.detach().cpu().numpy()
        out["layout"] = numpy.array(["%s:%d:%d" % (n, o, c) for n, (o, c) in self.rec.store.offsets.items()])
        return out

    def load_state_dict(self, state):
        layout = ["%s:%d:%d" % (n, o, c) for n, (o, c) in self.rec.store.offsets.items()]
        if "layout" in state and [str(x) for x in state["layout"]] != layout:
            raise ValueError("training state was saved for another parameter layout")
        for name in ("velocity", "ms_step", "ms_dx", "clip_state"):
            t = getattr(self, name)
            if t is not None and name in state:
                t.copy_(torch.as_tensor(numpy.asarray(state[name])).to(t.dtype))

    def gradient_threshold(self):
        """The StepClipping threshold in force for the next step (moves when adaptive clipping is on)."""
        return float(self.clip_state[0]) if self.clip_state is not None else self.conf["clip_threshold"]

    def gradient_norm(self):
        """L2 norm of the (scaled, all-reduced) gradient of the last step: the reference's `total_gradient_norm`."""
        return float(self.scratch[0])

    def _enqueue_guard(self):
        """Behind the backward pass: store.guard[0] = number of persistent cluster launches of this step that gave up waiting (the
        abort words in front of their workspaces).  The optimiser skips the step on the device when it is non-zero; under data
        parallelism the word is the first element of the gradient bucket, so every rank sees the sum and skips together."""
        rec, st, lib = self.rec, self.rec.store, self.rec.lib
        words = [t for k, t in rec.ws._bufs.items() if _is_sync_word(k)]
        assert len(words) <= 64, "more cluster workspaces than lvsr_guard_collect takes"
        arr = (ctypes.c_void_p * 64)(*[t.data_ptr() for t in words])
        lib.call("lvsr_guard_collect", lib.stream_for(st.flat), arr, len(words), ptr(st.guard))

    def _enqueue_optimizer(self, global_batch_size):
        st, lib = self.rec.store, self.rec.lib
        a = lib.make("lvsr_opt_args", param=st.flat, grad=st.grad, velocity=self.velocity, ms_step=self.ms_step,
                     ms_dx=self.ms_dx, step=self.step_buf, segments=self.segments, segflag=self.segflag,
                     scratch=self.scratch, n=st.flat.numel(), nseg=int(self.segments.shape[0]), max_cols=self.max_cols,
                     grad_scale=1.0 / float(global_batch_size), clip_state=self.clip_state, guard=st.guard, **self.conf)
        lib.call("lvsr_opt_step", lib.stream_for(st.flat), ctypes.byref(a))
        if self._skip_host is not None:
            self._skip_host.copy_(self.scratch[3:4], non_blocking=True)

    def step_was_skipped(self):
        """After a step (synchronises): did the optimiser skip it because a persistent cluster kernel gave up (lvsr_opt_args.guard)?"""
        return float(self.scratch[3]) != 0.0

    def recover(self):
        """Call when step_was_skipped(): clear the abort words, change what caused the abort (class comment: first leave
        `RECOVER_RESERVE` CUs free and keep the cluster kernels; on a second abort within REARM_STEPS steps run on the step kernels
        for REARM_STEPS clean steps, then arm the cluster kernels again), drop the captured graphs and tell the caller to run the
        batch again.  Parameters and rule state are untouched by the skipped step.  -> dict describing what was done (also logged)."""
        rec, lib = self.rec, self.rec.lib
        for k, t in rec.ws._bufs.items():
            if _is_sync_word(k):
                t[:16].zero_()
        self.aborts += 1
        repeated = self._last_abort_step is not None and self.steps_done - self._last_abort_step < self.REARM_STEPS
        self._last_abort_step = self.steps_done
        action = dict(aborts=self.aborts, step=self.steps_done)
        if not repeated and self._fallback is None and self._reserve_before is None:
            # RECOVER_RESERVE more CUs than were already left free (the overlapped all-reduce keeps OVERLAP_RESERVE of its own)
            self._reserve_before = lib.get_knob("cluster_reserve")
            reserve = self._reserve_before + self.RECOVER_RESERVE
            lib.set_knob("cluster_reserve", reserve)
            action.update(action="cluster_reserve", cluster_reserve=reserve)
            logger.warning("a persistent cluster kernel gave up waiting for its partners (step %d): the step was skipped on the device; "
                           "cluster launches now leave %d CUs free; run the batch again", self.steps_done, reserve)
        else:
            if self._fallback is None:
                self._fallback = dict(saved=(rec.encoder.use_persistent, rec.encoder.persist_auto, rec.generator.use_persistent,
                                             getattr(rec.generator, "use_persistent_stack", None)))
            self._fallback["clean"] = 0
            rec.encoder.use_persistent, rec.encoder.persist_auto = False, False
            rec.generator.use_persistent = False
            if hasattr(rec.generator, "use_persistent_stack"):
                rec.generator.use_persistent_stack = False
            action.update(action="step_kernels", rearm_after=self.REARM_STEPS)
            logger.warning("a persistent cluster kernel gave up again (step %d, abort %d): encoder and decoder run on the STEP KERNELS "
                           "(several times slower) for the next %d steps, then the cluster kernels are armed again",
                           self.steps_done, self.aborts, self.REARM_STEPS)
        self._forget_graphs()
        if self._skip_host is not None:
            torch.cuda.synchronize(rec.store.device)
            self._skip_host.zero_()
        return action

    def close(self):
        """Give back the process-global knobs this trainer changed (the overlap reserve, a recovery's reserve still in force): other
        recognizers / trainers of the process see the library as this one found it."""
        lib = self.rec.lib
        if self._reserve_before is not None:
            lib.set_knob("cluster_reserve", self._reserve_before)
            self._reserve_before = None
        if self._overlap_reserve_before is not None:
            lib.set_knob("cluster_reserve", self._overlap_reserve_before)
            self._overlap_reserve_before = None
        self._forget_graphs()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _forget_graphs(self):
        """The captured whole-step graphs replay the launches of the old kernel choice: forget them and every region's counters (the
        next step of every shape is enqueued eagerly — workspaces of the new choice come into being — the one after captured)."""
        rec = self.rec
        for owner in (rec, self, rec.encoder, rec.generator, rec.bottom):
            getattr(owner, "_regions", {}).clear()
        rec.lib._lvsr_graph_clear()

    def _rearm(self):
        """REARM_STEPS clean steps on the step kernels: back to the cluster kernels (the reserve stays)."""
        rec = self.rec
        enc_p, enc_auto, gen_p, gen_stack = self._fallback["saved"]
        rec.encoder.use_persistent, rec.encoder.persist_auto = enc_p, enc_auto
        rec.generator.use_persistent = gen_p
        if gen_stack is not None:
            rec.generator.use_persistent_stack = gen_stack
        self._fallback = None
        self._last_abort_step = self.steps_done          # (an abort within REARM_STEPS of the re-arming counts as a repeated one)
        self._forget_graphs()
        logger.warning("step %d: %d clean steps on the step kernels; the persistent cluster kernels are armed again", self.steps_done,
                       self.REARM_STEPS)

    def _before_step(self):
        if self._skip_host is not None and float(self._skip_host[0]) != 0.0:
            raise RuntimeError("the previous training step was skipped on the device (a persistent cluster kernel gave up waiting); "
                               "call Trainer.recover() and run that batch again (lvsr_amd.main.train does)")
        if self._fallback is not None:
            self._fallback["clean"] += 1
            if self._fallback["clean"] > self.REARM_STEPS:
                self._rearm()
        elif self._reserve_before is not None and self.steps_done - self._last_abort_step > self.REARM_STEPS:
            # REARM_STEPS clean steps on the cluster kernels behind the last recovery (or behind the re-arming): the CUs are given back
            self.rec.lib.set_knob("cluster_reserve", self._reserve_before)
            logger.warning("step %d: no cluster launch gave up for %d steps; cluster_reserve back to %d", self.steps_done,
                           self.steps_done - self._last_abort_step, self._reserve_before)
            self._reserve_before = None
            self._forget_graphs()
        self.steps_done += 1

    def _all_reduce_gradients(self):
        """ONE collective per step over the flat gradient bucket (sum); RCCL over xGMI when the tensors are on GPUs.
        On a GPU it is issued on a communication stream of its own, ordered after / before the compute stream with events:
        ProcessGroupNCCL's watchdog thread polls the collective's end event, and HIP refuses hipEventQuery on an event whose
        stream has meanwhile entered hipGraph capture (hipErrorCapturedEvent kills the process: tools/probes/
        nccl_capture_probe.py) — the compute stream captures the next minibatch shape's graph, the communication stream
        never captures."""
        self._all_reduce(self.rec.store.grad_bucket)          # [guard | gradients]

    def _all_reduce(self, g, wait=True):
        """Sum all-reduce of (a slice of) the flat gradient buffer; wait=False leaves the compute stream un-ordered behind it
        (the caller joins later with `_join_comm`): the exchange then overlaps whatever the compute stream does next."""
        if not g.is_cuda:
            torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, group=self.group)
            return
        cur = torch.cuda.current_stream(g.device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(g.device)
        self._comm.wait_stream(cur)
        with torch.cuda.stream(self._comm):
            torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, group=self.group)
        if wait:
            cur.wait_stream(self._comm)

    def _join_comm(self, ref):
        if ref.is_cuda and self._comm is not None:
            torch.cuda.current_stream(ref.device).wait_stream(self._comm)

    def apply_gradients(self, global_batch_size):
        rec, st = self.rec, self.rec.store
        with rec._on_stream():
            self._enqueue_guard()
            if self.distributed:
                self._all_reduce_gradients()
            self._enqueue_optimizer(global_batch_size)
            st.version += 1

    def train_step(self, batch, global_batch_size=None):
        """batch = this rank's shard (reference layout).  Returns the local cost matrix as a device tensor.
        Single process: forward, backward and the optimiser step are one graph region (one launch per step); with data
        parallelism the region ends before the all-reduce and the optimiser follows it.  `global_batch_size` = number of
        utterances of the whole (all ranks) minibatch, the divisor of the summed cost (lvsr/main.py:340-345); when omitted
        under data parallelism it is obtained by reducing the shard sizes."""
        B_local = int(batch["labels"].shape[1])
        self._before_step()
        if self.distributed:
            if global_batch_size is None:
                # shards may differ in size (a global batch that does not divide over the ranks): the divisor of the summed
                # gradient must be the same number on every rank, so it is reduced too (one 8-byte collective)
                n = torch.tensor([B_local], dtype=torch.int64, device=self.rec.store.grad.device)
                torch.distributed.all_reduce(n, op=torch.distributed.ReduceOp.SUM, group=self.group)
                global_batch_size = int(n[0])
            if self.overlap_allreduce:
                rec, st = self.rec, self.rec.store
                off, cnt = rec.decoder_bucket()

                def reduce_decoder_bucket():
                    with rec._on_stream():
                        self._all_reduce(st.grad[off: off + cnt], wait=False)
                cm = rec.cost_and_gradients(batch, region=self.dp_region, between=reduce_decoder_bucket)
                with rec._on_stream():
                    self._enqueue_guard()
                    self._all_reduce(st.grad_bucket[: 4 + off], wait=False)          # [guard | encoder gradients]
                    self._join_comm(st.grad)
                    self._enqueue_optimizer(global_batch_size)
                    st.version += 1
                return cm
            cm = self.rec.cost_and_gradients(batch, region=self.dp_region)
            self.apply_gradients(global_batch_size)
            return cm
        gbs = global_batch_size if global_batch_size is not None else B_local
        tail_key = ("opt", self._token, float(gbs), tuple(sorted(self.conf.items())))
        def tail():
            self._enqueue_guard()
            self._enqueue_optimizer(gbs)
        cm = self.rec.cost_and_gradients(batch, tail=tail, tail_key=tail_key)
        self.rec.store.version += 1
        return cm
