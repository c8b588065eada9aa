"""Host-side mirror of `lvsr.bricks.recognizer.SpeechRecognizer` (lvsr/bricks/recognizer.py:159-562): same
constructor keywords (the `net:` section of the reference's YAML), same method names
(`cost`, `analyze`, `init_beam_search`, `beam_search`, `load_params`), parameters under the reference's
Blocks names — with every tensor operation executed by the HIP library on an MI355X.

What Theano derived symbolically is explicit here: `cost(...)` is the forward pass (cost matrix (L,B)),
`backward()` the gradient of `cost.sum()` wrt all parameters (lvsr/main.py:340-345 divides by the batch
size afterwards; `cost_and_gradients` does the same).
"""
import contextlib
import os

import numpy
import torch

from .. import spec
from ..params import ParameterStore, Workspace
from . import Encoder, SpeechBottom
from .generator import SequenceGenerator


class SpeechRecognizer(object):
    def __init__(self, device="cuda:0", params=None, lib=None, use_graph=True, use_persistent=None, net_config=None,
                 use_persistent_decoder=None, **net_kwargs):
        """`net_kwargs` = the reference's constructor keywords (recognizer.py:176-204), or pass an
        already-normalised `net_config` (lvsr_amd.spec).  `params` = dict name -> ndarray (Blocks names).
        `use_persistent` / `use_persistent_decoder`: force (True) or forbid (False) the persistent cluster kernels of the
        encoder / of the decoder's teacher-forced label loop; None = use them where they are available and faster."""
        from .. import native
        lm_config = dict(net_kwargs.get("lm") or {})
        cfg = net_config if net_config is not None else spec.from_reference_kwargs(**net_kwargs)
        self.d = spec.Dims(cfg)
        self.cfg = self.d.cfg
        self.device = torch.device(device)
        self.lib = lib if lib is not None else native.get()
        if self.device.type != "cuda" and not self.lib.is_emulator:
            raise native.NativeError("SpeechRecognizer runs on an MI355X (cuda:N) device only; no CPU fallback")
        self.eos_label = self.cfg["eos_label"]
        self.data_prepend_eos = self.cfg["data_prepend_eos"]
        self.max_decoded_length_scale = self.cfg["max_decoded_length_scale"]
        self.store = ParameterStore(cfg, self.device, params)
        self.ws = Workspace(self.device)
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self.bottom = SpeechBottom(self.d, self.store, self.lib, self.ws)
        self.encoder = Encoder(self.d, self.store, self.lib, self.ws, use_graph=self.use_graph, use_persistent=use_persistent)
        if self.d.n_dec > 1:         # RecurrentStack decoder (recognizer.py:250-262)
            from .generator_stack import StackedSequenceGenerator as generator_class
        else:
            generator_class = SequenceGenerator
        self.generator = generator_class(self.d, self.store, self.lib, self.ws, use_graph=self.use_graph,
                                         use_persistent=use_persistent_decoder)
        # hipGraph capture cannot run on the legacy null stream: the hot path owns a side stream
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.beam_size = None
        if lm_config.get("path"):                                  # recognizer.py:322-337
            from ..lm import language_model_from_config
            self.set_language_model(language_model_from_config(lm_config, net_kwargs.get("character_map"), self.device,
                                                               self.lib))

    # ---- plumbing ----------------------------------------------------------------------------------
    @contextlib.contextmanager
    def _on_stream(self):
        if self.stream is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        if cur == self.stream:                                      # re-entrant
            yield
            return
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield
        cur.wait_stream(self.stream)

    def _t(self, x, dtype, name):
        if x is None:
            return None
        if not torch.is_tensor(x):
            x = torch.from_numpy(numpy.ascontiguousarray(x))
        x = x.to(dtype)
        buf = self.ws.get("in." + name, tuple(x.shape), dtype)
        buf.copy_(x, non_blocking=True)
        return buf

    def get_parameter_values(self):
        return self.store.get_values()

    def set_parameter_values(self, values):
        self.store.set_values(values)

    def load_params(self, path):
        """recognizer.py:408-412: load a Blocks checkpoint (tar with `_parameters`) or an .npz by parameter name."""
        from ..checkpoint import load_parameters
        self.store.set_values(load_parameters(path))

    def save_params(self, path, extra=None):
        from ..checkpoint import save_parameters
        save_parameters(path, self.store.get_values(), extra=extra)

    def initialize(self, initialization, seed=1):
        """Apply the reference's `initialization:` config section (lvsr/main.py:225-232): brick paths mapped to
        {weights_init, biases_init, rec_weights_init, initial_states_init}, applied shallow-to-deep; recurrent weights
        follow GatedRecurrent._initialize (libs/blocks/blocks/bricks/recurrent.py:567-580: state_to_gates =
        hstack[init, init]).  The RNG stream is numpy.random.RandomState(seed) per parameter name — it does NOT
        reproduce Blocks' per-brick seeding (parity tests always load explicit parameter values)."""
        import zlib
        chosen = {}
        for path in sorted(initialization, key=lambda s: s.count("/")):
            for name in self.store.shapes:
                if name.startswith(path.rstrip("/") + "/") or name == path:
                    chosen.setdefault(name, {}).update(initialization[path])
        values = {}
        for name, shape in self.store.shapes.items():
            conf = chosen.get(name, {})
            rng = numpy.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 31))
            leaf = name.rsplit("/", 1)[-1]
            rec = conf.get("rec_weights_init", conf.get("weights_init"))
            if leaf.endswith("state_to_state") and rec is not None:
                v = rec.generate(rng, shape)
            elif leaf.endswith("state_to_gates") and rec is not None:
                H = shape[0]
                v = numpy.hstack([rec.generate(rng, (H, H)), rec.generate(rng, (H, H))])
            elif leaf.endswith("initial_state"):
                init = conf.get("initial_states_init")
                v = init.generate(rng, shape) if init is not None else numpy.zeros(shape, numpy.float32)
            elif leaf.endswith(".b"):
                init = conf.get("biases_init")
                v = init.generate(rng, shape) if init is not None else numpy.zeros(shape, numpy.float32)
            else:
                init = conf.get("weights_init")
                if init is None:
                    raise ValueError("no weights_init configured for %s" % name)
                v = init.generate(rng, shape)
            values[name] = numpy.asarray(v, numpy.float32)
        self.store.set_values(values)

    # ---- training cost (recognizer.py:375-390) -------------------------------------------------------
    def cost(self, recordings=None, inputs_mask=None, labels=None, labels_mask=None, save_for_backward=True, **kw):
        """-> cost matrix (L,B) on the device.  `recordings` (T,B,F), `inputs_mask` (T,B) or None,
        `labels` (L,B) int64, `labels_mask` (L,B) or None."""
        if "recordings_mask" in kw:
            inputs_mask = kw.pop("recordings_mask")
        if kw:
            raise TypeError("unknown inputs: %s" % sorted(kw))
        with self._on_stream():
            x, xm, y, ym = self._stage(recordings, inputs_mask, labels, labels_mask)
            cm = self._forward(x, xm, y, ym, save_for_backward)
        return cm

    def _stage(self, recordings, inputs_mask, labels, labels_mask):
        return (self._t(recordings, torch.float32, "recordings"), self._t(inputs_mask, torch.float32, "recordings_mask"),
                self._t(labels, torch.int64, "labels"), self._t(labels_mask, torch.float32, "labels_mask"))

    def _forward(self, x, xm, y, ym, save_for_backward=True):
        encoded, encoded_mask = self.encoder.apply(self.bottom.apply(x, save_for_backward), xm,
                                                   save_for_backward=save_for_backward)
        self.encoded, self.encoded_mask = encoded, encoded_mask
        return self.generator.cost_matrix(y, ym, attended=encoded, attended_mask=encoded_mask,
                                          save_for_backward=save_for_backward)

    # ---- free-running generation (recognizer.py:393-406, 535-547) ------------------------------------------------------
    def generate(self, n_steps=None, inputs_mask=None, recordings=None, uniforms=None, seed=None, **kw):
        """SpeechRecognizer.generate: encoder, then SequenceGenerator.generate for `n_steps` steps on the whole batch.
        -> dict(states, outputs, weighted_averages, weights, energies, costs) of device tensors, time-major."""
        if "recordings_mask" in kw:
            inputs_mask = kw.pop("recordings_mask")
        if kw:
            raise TypeError("unknown inputs: %s" % sorted(kw))
        with self._on_stream():
            x = self._t(recordings, torch.float32, "recordings")
            xm = self._t(inputs_mask, torch.float32, "recordings_mask")
            encoded, encoded_mask = self.encoder.apply(self.bottom.apply(x, False), xm, save_for_backward=False)
            return self.generator.generate(n_steps=n_steps, batch_size=int(encoded.shape[1]), attended=encoded,
                                           attended_mask=encoded_mask, uniforms=uniforms, seed=seed)

    def sample(self, inputs, n_steps=None, uniforms=None, seed=None):
        """SpeechRecognizer.sample (recognizer.py:540-547): one utterance, no input mask, n_steps defaults to
        frames / max_decoded_length_scale; -> the sampled label sequence (n_steps, 1) as a numpy array."""
        x = numpy.asarray(dict(inputs)["recordings"], dtype=numpy.float32)
        if n_steps is None:
            n_steps = int(x.shape[0] / self.max_decoded_length_scale)
        out = self.generate(n_steps=n_steps, inputs_mask=None, recordings=x[:, None, :],
                            uniforms=None if uniforms is None else numpy.asarray(uniforms, numpy.float32).reshape(n_steps, 1), seed=seed)
        return out["outputs"].cpu().numpy()

    def backward(self):
        """Gradient of cost.sum() wrt all parameters -> self.store.grad (flat) / self.store.g (named views)."""
        with self._on_stream():
            # the small weight-gradient products of decoder and encoder (recurrent matrices, readout, ...) are collected and
            # run as ONE grouped launch at the end: alone none of them fills the chip
            self.lib.begin_group()
            d_encoded = self.generator.backward()
            self._backward_encoder(d_encoded)

    def _backward_decoder(self):
        """The decoder's half of backward() with a grouped launch of its own: afterwards every gradient under
        /recognizer/generator is final (the tail of the flat gradient buffer, `decoder_bucket()`).  -> d_encoded."""
        with self._on_stream():
            self.lib.begin_group()
            d_encoded = self.generator.backward()
            self.lib.flush_group(self.ws.get("gemm_ws.grouped", (1 << 26,)))
            return d_encoded

    def _backward_encoder(self, d_encoded):
        """Encoder (and bottom) half of backward(); flushes the pending grouped launch (opens one if none is pending)."""
        with self._on_stream():
            if getattr(self.lib, "_group", None) is None:
                self.lib.begin_group()
            d_bottom = self.encoder.backward(d_encoded, need_input_grad=bool(self.d.bottom_dims))
            self.lib.flush_group(self.ws.get("gemm_ws.grouped", (1 << 26,)))
            self.encoder.finish_backward()
            if self.d.bottom_dims:
                self.bottom.backward(d_bottom)

    def decoder_bucket(self):
        """(offset, count) of the decoder's gradients in the flat buffers: the parameters under /recognizer/generator are laid out
        behind everything else (spec.parameter_shapes), so they form one contiguous tail."""
        offs = self.store.offsets
        first = min(o for k, (o, n) in offs.items() if k.startswith("/recognizer/generator"))
        assert all(k.startswith("/recognizer/generator") == (o >= first) for k, (o, n) in offs.items()), "decoder parameters are not a tail"
        return first, self.store.flat.numel() - first

    def cost_and_gradients(self, batch, tail=None, tail_key=None, region=True, between=None):
        """One training forward+backward on a batch dict in the reference's layout.
        Returns the cost matrix (L,B) on the device; gradients of its sum are in self.store.grad.
        `tail` (optional callable, described by the hashable `tail_key`) enqueues more work behind the backward pass — the
        optimiser step — inside the same graph region: the whole step is then ONE hipGraph launch per minibatch shape.
        `between` (optional callable, data parallelism with overlapped exchange): called — eagerly, outside any graph region — when
        the decoder's gradients are final and before the encoder's backward pass is enqueued; the step is then TWO graph regions
        (forward + decoder backward | encoder backward [+ tail])."""
        with self._on_stream():
            x, xm, y, ym = self._stage(batch["recordings"], batch.get("recordings_mask"), batch["labels"],
                                       batch.get("labels_mask"))
            shape_key = (tuple(x.shape), tuple(y.shape), xm is None, ym is None)
            volatile = (x.data_ptr(), y.data_ptr(), 0 if xm is None else xm.data_ptr(), 0 if ym is None else ym.data_ptr(),
                        self.ws.generation, self.store.flat.data_ptr(), self.store.grad.data_ptr())
            plain = region and self.use_graph and not self.encoder.overlap
            if between is not None:
                def first_half():
                    cm = self._forward(x, xm, y, ym)
                    return cm, self._backward_decoder()

                cm, d_encoded = self.lib.region(self, ("train_step_fwd_dec",) + shape_key, x, enabled=plain, volatile=volatile).run(first_half)
                between()

                def second_half():
                    self._backward_encoder(d_encoded)
                    if tail is not None:
                        tail()
                    return True
                vol2 = volatile + (d_encoded.data_ptr(), self.ws.generation)
                self.lib.region(self, ("train_step_enc",) + shape_key + (tail_key,), x, enabled=plain, volatile=vol2).run(second_half)
                return cm

            def enqueue():
                cm = self._forward(x, xm, y, ym)
                self.backward()
                if tail is not None:
                    tail()
                return cm
            key = ("train_step",) + shape_key + (tail_key,)
            return self.lib.region(self, key, x, enabled=plain, volatile=volatile).run(enqueue)

    # ---- analyze (recognizer.py:452-494) -----------------------------------------------------------
    def analyze(self, inputs, groundtruth, prediction=None):
        """Single utterance: -> [cost (L,), weights (L,T'), energies (L,T')] as numpy arrays."""
        x = numpy.asarray(dict(inputs)["recordings"], dtype=numpy.float32)
        y = numpy.asarray(prediction if prediction is not None else groundtruth, dtype=numpy.int64)
        cm = self.cost(recordings=x[:, None, :], inputs_mask=None, labels=y[:, None], labels_mask=None,
                       save_for_backward=False)
        last = self.generator.last
        torch.cuda.synchronize() if self.device.type == "cuda" else None
        return [cm[:, 0].cpu().numpy(), last["weights"][:, 0, :].cpu().numpy(), last["energies"][:, 0, :].cpu().numpy()]

    # ---- decoding (recognizer.py:496-533) ------------------------------------------------------------
    def compute_contexts(self, recordings):
        """Encoder at batch 1 with NO input mask (init_beam_search builds the graph with use_mask=False,
        recognizer.py:506; Encoder.apply then returns an all-ones mask, lvsr/bricks/__init__.py:78)."""
        x = numpy.asarray(recordings, dtype=numpy.float32)
        if x.ndim == 2:
            x = x[:, None, :]
        xb = self._t(x, torch.float32, "recordings")
        encoded, encoded_mask = self.encoder.apply(self.bottom.apply(xb, False), None, save_for_backward=False)
        self.generator.init_generation(encoded, encoded_mask)

    def compute_contexts_batch(self, recordings, pad_to=64):
        """Encoder pass of several utterances at once (`recordings`: list of (T_i, F) arrays) for `BeamSearch.search_batch`: padded
        to a common length (a multiple of `pad_to` frames, so that few distinct shapes — and captured graphs — occur) under an input
        mask.  A masked recurrent step keeps its state (recurrent.py:297-300), so every utterance's contexts are those of its own
        unmasked batch-1 pass (`compute_contexts`) up to its own attended length."""
        recs = [numpy.asarray(r, dtype=numpy.float32).reshape(len(r), -1) for r in recordings]
        T = max(len(r) for r in recs)
        T = (T + pad_to - 1) // pad_to * pad_to
        x = numpy.zeros((T, len(recs), recs[0].shape[1]), numpy.float32)
        m = numpy.zeros((T, len(recs)), numpy.float32)
        for i, r in enumerate(recs):
            x[: len(r), i] = r
            m[: len(r), i] = 1.0
        xb, mb = self._t(x, torch.float32, "recordings"), self._t(m, torch.float32, "recordings_mask")
        encoded, encoded_mask = self.encoder.apply(self.bottom.apply(xb, False), mb, save_for_backward=False)
        self.generator.init_generation(encoded, encoded_mask)

    def beam_search_batch(self, recordings, **kwargs):
        """`beam_search` for a list of utterances in one set of launches -> list of (outputs, costs) or of the exception the
        single search would have raised (CandidateNotFoundError, ...)."""
        self.init_beam_search(self.beam_size)
        recs = [numpy.asarray(r, dtype=numpy.float32) for r in recordings]
        limits = [int(r.shape[0] / self.max_decoded_length_scale) for r in recs]
        results = self._beam_search.search_batch(recs, self.eos_label, limits, ignore_first_eol=self.data_prepend_eos, **kwargs)
        return [r if isinstance(r, Exception) else ([[int(t) for t in o] for o in r[0]], [float(c) for c in r[1]]) for r in results]

    def init_beam_search(self, beam_size):
        from ..search import BeamSearch
        if getattr(self, "_beam_search", None) is not None and self.beam_size == beam_size:
            return
        self.beam_size = beam_size
        self._beam_search = BeamSearch(beam_size, self)

    def beam_search(self, inputs, **kwargs):
        """-> (outputs: list of label lists, costs: list of floats), best first."""
        self.init_beam_search(self.beam_size)
        inputs = dict(inputs)
        rec = numpy.asarray(inputs.pop("recordings"), dtype=numpy.float32)
        if inputs:
            raise Exception("Unknown inputs passed to beam search: {}".format(inputs.keys()))     # recognizer.py:525-528
        max_length = int(rec.shape[0] / self.max_decoded_length_scale)                             # :519-520
        outputs, search_costs = self._beam_search.search(
            {"recordings": rec[:, None, :]}, self.eos_label, max_length, ignore_first_eol=self.data_prepend_eos, **kwargs)
        return [[int(t) for t in o] for o in outputs], [float(c) for c in search_costs]

    def set_language_model(self, language_model):
        """Attach (or detach with None) an `lvsr_amd.lm.FSTLanguageModel` for shallow-fusion decoding — the `lm:`
        sub-section of the reference's net config (recognizer.py:322-343)."""
        if language_model is not None and language_model.out_dim != self.d.V:
            raise ValueError("language model covers %d characters, the recognizer %d" % (language_model.out_dim, self.d.V))
        self.generator.language_model = language_model

    def lm_initial_states(self, n):
        return self.generator.language_model.initial_states(n)
