"""Thin trainer around SpeechRecognizer: one data-parallel SGD step of the reference's recipe
(lvsr/main.py:340-345 cost = cost_matrix.sum()/batch_size; :480-519 step rules; GradientDescent.process_batch,
libs/blocks/blocks/algorithms/__init__.py:284-287).

Data parallelism (new; the reference is single-device): every rank holds a full replica, processes its shard of
the utterances (rank r takes utterances r::world), gradients of the summed cost are all-reduced (sum) in ONE
RCCL collective over the flat gradient buffer, then divided by the GLOBAL batch size inside the fused optimiser
kernel; all ranks apply identical updates.
"""
import ctypes
import os

import numpy
import torch

from .native import ptr


def _is_weight(name):
    return name.endswith(".W") or name.endswith("state_to_state") or name.endswith("state_to_gates")


class Trainer(object):
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
        words = [t for k, t in rec.ws._bufs.items() if k[0] in ("gen.sync", "gen.sync_bwd") or (k[0].startswith("enc") and k[0].endswith(".sync"))]
        arr = (ctypes.c_void_p * 16)(*[t.data_ptr() for t in words[:16]])
        lib.call("lvsr_guard_collect", lib.stream_for(st.flat), arr, min(len(words), 16), ptr(st.guard))

    def _enqueue_optimizer(self, global_batch_size):
        st, lib = self.rec.store, self.rec.lib
        a = lib.make("lvsr_opt_args", param=st.flat, grad=st.grad, velocity=self.velocity, ms_step=self.ms_step,
                     ms_dx=self.ms_dx, step=self.step_buf, segments=self.segments, segflag=self.segflag,
                     scratch=self.scratch, n=st.flat.numel(), nseg=int(self.segments.shape[0]), max_cols=self.max_cols,
                     grad_scale=1.0 / float(global_batch_size), clip_state=self.clip_state, guard=st.guard, **self.conf)
        lib.call("lvsr_opt_step", lib.stream_for(st.flat), ctypes.byref(a))

    def step_was_skipped(self):
        """After a step (synchronises): did the optimiser skip it because a persistent cluster kernel gave up (lvsr_opt_args.guard)?"""
        return float(self.scratch[3]) != 0.0

    def recover(self):
        """Call when step_was_skipped(): clear the abort words, put encoder and decoder on their step kernels for the rest of the run
        (a cluster launch needs all its work-groups resident at once — something else is using the device's CUs) and tell the
        caller to run the batch again.  Parameters and rule state are untouched by the skipped step."""
        rec = self.rec
        for k, t in rec.ws._bufs.items():
            if k[0] in ("gen.sync", "gen.sync_bwd") or (k[0].startswith("enc") and k[0].endswith(".sync")):
                t[:16].zero_()
        rec.encoder.use_persistent, rec.encoder.persist_auto = False, False
        rec.generator.use_persistent = False
        if hasattr(rec.generator, "use_persistent_stack"):
            rec.generator.use_persistent_stack = False
        # the captured whole-step graphs replay the cluster launches: forget them (the next step of every shape is enqueued eagerly,
        # the one after captured again — on the step kernels)
        getattr(rec, "_regions", {}).clear()
        getattr(self, "_regions", {}).clear()
        rec.lib._lvsr_graph_clear()

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
