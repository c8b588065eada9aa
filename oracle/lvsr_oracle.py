"""CPU ORACLE for the attention-LVCSR hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It is a
plain restatement (torch CPU tensors, autograd for gradients) of the reference's algorithm; every
function cites the reference file:line it follows (paths relative to /root/reference).

Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
  * fixtures in tests/golden/*.npz produced by running the reference's own SpeechRecognizer (Theano
    Python linker, float32) — cost matrices, alignments, encoder outputs, all parameter gradients,
    beam-search hypotheses, analyze() — generator script: oracle/theano_harness/gen_golden.py;
  * the reference's own known-answer tests: tests/test_conv1d.py:6-13 (conv flip),
    libs/blocks/tests/test_search.py:65-69 (_smallest),
    libs/blocks/tests/bricks/test_recurrent.py:432-495 (GRU step, masked sequence), :498-535 (bidirectional).
Exception: mel-filterbank extraction (Kaldi; not under /root/reference, no version pinned) — pinned to an independent
Kaldi-compatible implementation only, see oracle/fbank_oracle.py.
"""
import math
from collections import OrderedDict

import numpy
import torch

import sys, os
_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "attention-lvcsr_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from lvsr_amd.spec import Dims, parameter_shapes, decoder_layer_names  # shape / name tables only (pure python)


# ----------------------------------------------------------------------------------------------
# small pieces pinned by the reference's own unit tests
# ----------------------------------------------------------------------------------------------
def conv1d_full(sequences, filters):
    """lvsr/expressions.py:28-54 with border_mode='full': TRUE convolution (filter flipped).

    sequences (B, N), filters (K, W) -> (B, K, N + W - 1);  out[b,k,p] = sum_j f[k,j] * s[b,p-j].
    """
    s = torch.as_tensor(sequences)
    f = torch.as_tensor(filters).to(s.dtype)
    W = f.shape[1]
    return torch.nn.functional.conv1d(s[:, None, :], torch.flip(f, [1])[:, None, :], padding=W - 1)


def conv1d_valid(sequences, filters):
    """lvsr/expressions.py:28-54 with the default border_mode='valid'."""
    s = torch.as_tensor(sequences)
    f = torch.as_tensor(filters).to(s.dtype)
    return torch.nn.functional.conv1d(s[:, None, :], torch.flip(f, [1])[:, None, :])


def smallest(matrix, k):
    """libs/blocks/blocks/search.py:221-242 (BeamSearch._smallest), verbatim semantics incl. numpy tie order."""
    flatten = matrix.flatten()
    if flatten.shape[0] > k:
        args = numpy.argpartition(flatten, k)[:k]
    else:
        args = numpy.arange(flatten.shape[0])
    args = args[numpy.argsort(flatten[args])]
    return numpy.unravel_index(args, matrix.shape), flatten[args]


def gru_step(h, x_in, g_in, W_hh, W_hg, mask=None, gate_act=torch.sigmoid, act=torch.tanh):
    """libs/blocks/blocks/bricks/recurrent.py:608-620.  Gate columns: [:H] update, [H:] reset."""
    H = h.shape[1]
    g = gate_act(h @ W_hg + g_in)
    u, r = g[:, :H], g[:, H:]
    cand = act((h * r) @ W_hh + x_in)
    nh = cand * u + h * (1 - u)
    if mask is not None:
        nh = mask[:, None] * nh + (1 - mask[:, None]) * h
    return nh


def gru_sequence(x_in, g_in, mask, W_hh, W_hg, h0, reverse=False, **kw):
    """recurrent.py:178-231 (scan, go_backwards) + :622-624 (initial state tiled over the batch).
    reverse=True iterates t = T-1..0 WITH THE SAME MASK and returns outputs re-reversed (Bidirectional, :655-663)."""
    T, B = x_in.shape[0], x_in.shape[1]
    h = h0[None, :].expand(B, -1)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h = gru_step(h, x_in[t], g_in[t], W_hh, W_hg, None if mask is None else mask[t], **kw)
        out[t] = h
    return torch.stack(out, 0)


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
def _f32(x):
    return float(numpy.float32(x))


class OracleRecognizer(object):
    """Restatement of SpeechRecognizer's log-likelihood path (lvsr/bricks/recognizer.py:159-562)."""

    def __init__(self, cfg, params, dtype=torch.float32):
        self.d = Dims(cfg)
        self.cfg = self.d.cfg
        self.dtype = dtype
        shapes = parameter_shapes(cfg)
        assert set(shapes) == set(params), sorted(set(shapes) ^ set(params))
        self.p = OrderedDict()
        for k in shapes:
            v = numpy.asarray(params[k])
            assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
            self.p[k] = torch.tensor(v, dtype=dtype, requires_grad=True)

    # -- encoder: lvsr/bricks/__init__.py:54-78 ------------------------------------------------
    def bottom(self, x):
        """SpeechBottom.apply (lvsr/bricks/recognizer.py:105-137): MLP([activation]*len(dims), [F]+dims) or Identity."""
        d, p = self.d, self.p
        act = {"rectifier": torch.relu, "tanh": torch.tanh, "identity": lambda v: v}[d.bottom_act]
        for j in range(len(d.bottom_dims)):
            x = act(x @ p["/recognizer/bottom/bottom/linear_%d.W" % j] + p["/recognizer/bottom/bottom/linear_%d.b" % j])
        return x

    def encode(self, x, x_mask):
        p, d = self.p, self.d
        h = self.bottom(x)
        m = x_mask
        for i, s in enumerate(d.subsample):
            outs = []
            for direction in ("forward", "backward"):
                base = "/recognizer/encoder/bidir%d/%s" % (i, direction)
                # RecurrentWithFork: lvsr/bricks/__init__.py:28-43 (Fork of two Linear bricks with bias)
                x_in = h @ p[base + "/fork/fork_inputs.W"] + p[base + "/fork/fork_inputs.b"]
                g_in = h @ p[base + "/fork/fork_gate_inputs.W"] + p[base + "/fork/fork_gate_inputs.b"]
                outs.append(gru_sequence(
                    x_in, g_in, m, p[base + "/gatedrecurrent.state_to_state"],
                    p[base + "/gatedrecurrent.state_to_gates"], p[base + "/gatedrecurrent.initial_state"],
                    reverse=(direction == "backward")))
            h = torch.cat(outs, dim=2)                      # recurrent.py:662
            h = h[::s]                                      # lvsr/bricks/__init__.py:75
            if m is not None:
                m = m[::s]
        if m is None:
            m = torch.ones(h.shape[:2], dtype=self.dtype)   # :78
        return h, m

    # -- attention ----------------------------------------------------------------------------
    def _att(self, name):
        return "/recognizer/generator/att_trans/%s/%s" % ("conv_att" if self.d.conv else "cont_att", name)

    def _dec(self, l):
        return decoder_layer_names(self.d, l)

    def _initial_state(self):
        return torch.cat([self.p[self._dec(l)["h0"]] for l in range(self.d.n_dec)], 0)

    def _window(self, step0, alpha_prev, Tp):
        """lvsr/bricks/attention.py:123-168: returns begin, end (python ints) and extra mask (Tc,B) or None."""
        pr = self.cfg["prior"]
        kind = pr.get("type", "expanding")
        if kind == "expanding":
            # int64 step * floatX python constant -> float64 arithmetic on the float32-rounded constant
            begin = pr["initial_begin"] + step0 * _f32(pr["min_speed"])
            end = pr["initial_end"] + step0 * _f32(pr["max_speed"])
            begin = max(0, min(Tp - 1, begin))
            end = max(0, min(Tp, end))
            return int(math.floor(begin)), int(math.ceil(end)), None
        a = alpha_prev.detach()
        if kind == "window_around_mean":
            pos = (a * torch.arange(Tp, dtype=a.dtype)[None, :]).sum(dim=1)          # :135-137
        elif kind == "window_around_median":
            c = torch.cumsum(a, dim=1) - 0.5                                          # :139
            ge = (c >= 0).to(torch.int8)
            diff = ge[:, 1:] - ge[:, :-1]
            if diff.shape[1] == 0:
                pos = torch.zeros(a.shape[0], dtype=a.dtype)
            else:
                # argmax = first index of the maximum
                pos = torch.tensor([int(numpy.argmax(diff[b].numpy())) for b in range(a.shape[0])], dtype=a.dtype)
        else:
            raise Exception("Unknown prior type: %s" % kind)
        before = torch.tensor(pr["before"], dtype=a.dtype)
        after = torch.tensor(pr["after"], dtype=a.dtype)
        begins = torch.floor(pos - before)
        ends = torch.ceil(pos + after)
        begin = int(max(0, float(begins.min())))
        end = int(min(Tp, float(ends.max())))
        posc = torch.arange(begin, end, dtype=a.dtype)[None, :]
        extra = ((posc > begins[:, None]) & (posc < ends[:, None])).to(a.dtype)       # (B, Tc)
        return begin, end, extra.T

    def take_glimpses(self, A, PA, Am, s, alpha_prev, step0):
        """One attention step.  Returns wa (B,E), alpha (B,T'), energies (B,T').
        content: libs/blocks/blocks/bricks/attention.py:346-388; content_and_conv: lvsr/bricks/attention.py:98-183."""
        p, d = self.p, self.d
        Tp, B = A.shape[0], A.shape[1]
        # state transformers of every state of the transition, summed (blocks attention.py:281-283, 346-353); `s` holds the
        # states of a stacked decoder side by side
        sW = s @ torch.cat([p[self._dec(l)["Ws"]] for l in range(d.n_dec)], 0)           # (B,M)
        w_e = p[self._att("energy_comp/linear.W")][:, 0]
        b_e = p[self._att("energy_comp/linear.b")][0] if d.energy_bias else 0.0      # blocks attention.py:417-431
        if not d.conv:
            match = PA + sW[None]
            e = torch.tanh(match) @ w_e                                               # (T',B)
            alpha = self._weights(e, Am)
            wa = (alpha[:, :, None] * A).sum(0)
            return wa, alpha.T, e.T
        begin, end, extra = self._window(step0, alpha_prev, Tp)
        a_cut = alpha_prev[:, begin:end]
        c = d.c
        cv = conv1d_full(a_cut, p[self._att("conv1d.filters")])[:, :, c:a_cut.shape[1] + c]   # (B,K,Tc)
        match = PA[begin:end] + sW[None] + (cv.permute(2, 0, 1) @ p[self._att("handler.W")])
        e_cut = torch.tanh(match) @ w_e + b_e                                         # (Tc,B)
        m_cut = Am[begin:end] * (extra if extra is not None else 1)
        a_new = self._weights(e_cut, m_cut, d.normalizer)
        wa = (a_new[:, :, None] * A[begin:end]).sum(0)
        alpha = torch.zeros(Tp, B, dtype=self.dtype)
        en = torch.zeros(Tp, B, dtype=self.dtype)
        alpha = torch.cat([alpha[:begin], a_new, alpha[end:]], 0)                     # paste, :177-181
        en = torch.cat([en[:begin], e_cut, en[end:]], 0)
        return wa, alpha.T, en.T

    @staticmethod
    def _weights(e, mask, normalizer="softmax"):
        """lvsr/bricks/attention.py:191-213 (softmax | logistic | relu normaliser); softmax == blocks attention.py:202-233."""
        if e.shape[0] == 0:
            return e
        if normalizer == "softmax":
            e = e - e.max(dim=0, keepdim=True)[0]
            u = torch.exp(e) * mask
        elif normalizer == "logistic":
            u = torch.sigmoid(e) * mask
        elif normalizer == "relu":
            u = torch.clamp(e / 1000.0, min=0.0) * mask
        else:
            raise Exception("Unknown energey_normalizer: {}".format(normalizer))
        norm = u.sum(0) + (1 - mask).min(dim=0)[0].clamp(min=0).floor()   # all(1-mask) as 0/1
        return u / norm

    # -- generator pieces ---------------------------------------------------------------------
    def feedback(self, y):
        """LookupFeedback (sequence_generators.py:839-842) / OneOfNFeedback (lvsr/bricks/__init__.py:97-104)."""
        d = self.d
        if d.embed:
            return self.p["/recognizer/generator/readout/lookupfeedback/lookuptable.W"][y]
        return torch.nn.functional.one_hot(y, d.V + 1).to(self.dtype)

    def decoder_gru(self, s, fb, wa, mask=None):
        """AttentionRecurrent.compute_states (blocks attention.py:625-662) = Distribute + GatedRecurrent step."""
        p, D = self.p, self.d.D
        new, below = [], None
        for l in range(self.d.n_dec):
            # RecurrentStack.do_apply, one step (recurrent.py:903-950): every layer gets its own fork of the feedback and its own
            # share of the distributed glimpse (skip connections); layer l > 0 adds a bias-free fork of the NEW state of layer l - 1
            n = self._dec(l)
            x_in = fb @ p[n["Wfi"]] + p[n["bfi"]] + wa @ p[n["Wdi"]]
            g_in = fb @ p[n["Wfg"]] + p[n["bfg"]] + wa @ p[n["Wdg"]]
            if l > 0:
                x_in = x_in + below @ p[n["Fi"]]
                g_in = g_in + below @ p[n["Fg"]]
            below = gru_step(s[..., l * D:(l + 1) * D], x_in, g_in, p[n["Whh"]], p[n["Whg"]], mask)
            new.append(below)
        return new[0] if len(new) == 1 else torch.cat(new, -1)

    def readout(self, s_prev, wa):
        """Readout.readout (sequence_generators.py:614-619) + post-merge (recognizer.py:298-320); feedback is not a source."""
        p, d = self.p, self.d
        g = "/recognizer/generator/readout"
        r = wa @ p[g + "/merge/transform_weighted_averages.W"]
        if d.use_states_for_readout:
            r = r + s_prev @ torch.cat([p[self._dec(l)["Wms"]] for l in range(d.n_dec)], 0)
        if not d.post_merge:
            return r + p[g + "/bias.b"]
        r = r + p[g + "/post_merge/bias.b"]
        if d.act == "maxout2":
            r = r.reshape(r.shape[:-1] + (d.P // 2, 2)).max(-1)[0]                    # simple.py:161-181
        elif d.act == "rectifier":
            r = torch.relu(r)
        elif d.act == "tanh":
            r = torch.tanh(r)
        # MLP([act] * (n - 1) + [Identity()], [d // pieces for d in post_merge_dims] + [V]) (recognizer.py:309-317)
        for j in range(len(d.pm_hidden)):
            r = r @ p[g + "/post_merge/mlp/linear_%d.W" % j] + p[g + "/post_merge/mlp/linear_%d.b" % j]
            r = torch.relu(r) if d.act == "rectifier" else torch.tanh(r) if d.act == "tanh" else r
        j = len(d.pm_hidden)
        return r @ p[g + "/post_merge/mlp/linear_%d.W" % j] + p[g + "/post_merge/mlp/linear_%d.b" % j]

    def initial_glimpses(self, B, Tp):
        """content: blocks attention.py:390-393 (zeros); conv: lvsr/bricks/attention.py:215-222 (one-hot at 0)."""
        wa = torch.zeros(B, self.d.E, dtype=self.dtype)
        alpha = torch.zeros(B, Tp, dtype=self.dtype)
        if self.d.conv:
            alpha[:, 0] = 1.0
        return wa, alpha

    # -- teacher forced cost: sequence_generators.py:254-311 -----------------------------------
    def cost(self, x, x_mask, labels, labels_mask):
        t = lambda a, dt=None: torch.as_tensor(numpy.asarray(a), dtype=dt or self.dtype)
        x, x_mask = t(x), (None if x_mask is None else t(x_mask))
        labels = torch.as_tensor(numpy.asarray(labels), dtype=torch.int64)
        ym = None if labels_mask is None else t(labels_mask)
        p, d = self.p, self.d
        A, Am = self.encode(x, x_mask)
        Tp, B = A.shape[0], A.shape[1]
        L = labels.shape[0]
        PA = A @ p[self._att("preprocess.W")] + p[self._att("preprocess.b")]
        s = self._initial_state()[None, :].expand(B, -1)
        wa, alpha = self.initial_glimpses(B, Tp)
        states, was, alphas, ens = [], [], [], []
        for i in range(L):
            wa, alpha, en = self.take_glimpses(A, PA, Am, s, alpha, i)
            states.append(s)                     # readout uses the OLD state with the NEW glimpse (:276-277)
            was.append(wa)
            alphas.append(alpha)
            ens.append(en)
            s = self.decoder_gru(s, self.feedback(labels[i]), wa, None if ym is None else ym[i])
        S = torch.stack(states, 0)
        WA = torch.stack(was, 0)
        r = self.readout(S, WA)                                                       # (L,B,V)
        logp = torch.log_softmax(r, dim=-1)                                           # simple.py:315-337
        cm = -logp.gather(2, labels[:, :, None])[:, :, 0]
        if ym is not None:
            cm = cm * ym
        return dict(cost_matrix=cm, weights=torch.stack(alphas, 0), energies=torch.stack(ens, 0),
                    encoded=A, encoded_mask=Am, states=S, weighted_averages=WA, readouts=r)

    def cost_and_grads(self, batch):
        """cost = cost_matrix.sum() (gen_golden convention; lvsr/main.py:340-345 divides by batch_size afterwards)."""
        for v in self.p.values():
            v.grad = None
        out = self.cost(batch["recordings"], batch["recordings_mask"], batch["labels"], batch["labels_mask"])
        total = out["cost_matrix"].sum()
        total.backward()
        grads = OrderedDict((k, v.grad.detach().numpy().copy() if v.grad is not None else
                             numpy.zeros(tuple(v.shape), numpy.float32)) for k, v in self.p.items())
        return out, grads

    # -- generation-mode step functions (what BeamSearch compiles: search.py:97-142) ------------
    def contexts(self, x):
        """context_computer: encoder at batch 1 with NO input mask (recognizer.py:506; lvsr/bricks/__init__.py:78)."""
        with torch.no_grad():
            x = torch.as_tensor(numpy.asarray(x), dtype=self.dtype)
            A, Am = self.encode(x[:, None, :], None)
            PA = A @ self.p[self._att("preprocess.W")] + self.p[self._att("preprocess.b")]
        return A, Am, PA

    def initial_states(self, n, Tp):
        s = self._initial_state().detach()[None, :].repeat(n, 1)
        wa, alpha = self.initial_glimpses(n, Tp)
        return dict(states=s, outputs=numpy.full((n,), self.d.V, dtype=numpy.int64),       # initial_output=V (recognizer.py:286)
                    weighted_averages=wa, weights=alpha, step=numpy.zeros((n,), numpy.int64))

    def _tile(self, ctx, n):
        A, Am, PA = ctx
        return A.expand(-1, n, -1), Am.expand(-1, n), PA.expand(-1, n, -1)

    def logprobs(self, ctx, st, lm=None, lm_vecs=None):
        """logprobs_computer: take_glimpses + readout + costs=-log_softmax (search.py:126-134).  With a language model:
        ShallowFusionReadout.readout (lvsr/bricks/language_models.py:92-104) and LMEmitter.costs = -readout (:166-168)."""
        with torch.no_grad():
            n = st["states"].shape[0]
            A, Am, PA = self._tile(ctx, n)
            wa, alpha, en = self.take_glimpses(A, PA, Am, st["states"], st["weights"], int(st["step"][0]) if n else 0)
            r = self.readout(st["states"], wa)
            if lm is not None:
                from oracle import lm_oracle as LO
                add = numpy.stack([lm["dense"].costs(v, lm["remap"], lm["no_transition_cost"]) for v in lm_vecs])
                return -LO.shallow_fusion(r.numpy(), add, lm["weight"], lm.get("am_beta", 1.0), *lm.get("norms", (True, False, False)))
            return (-torch.log_softmax(r, dim=-1)).numpy()

    def next_states(self, ctx, st, outputs):
        """next_state_computer: take_glimpses AGAIN + compute_states (search.py:112-124)."""
        with torch.no_grad():
            n = st["states"].shape[0]
            A, Am, PA = self._tile(ctx, n)
            wa, alpha, en = self.take_glimpses(A, PA, Am, st["states"], st["weights"], int(st["step"][0]) if n else 0)
            y = torch.as_tensor(outputs, dtype=torch.int64)
            s = self.decoder_gru(st["states"], self.feedback(y), wa)
            return dict(states=s, outputs=numpy.asarray(outputs), weighted_averages=wa, weights=alpha,
                        step=st["step"] + 1)

    # -- beam search: libs/blocks/blocks/search.py:244-407 as driven by recognizer.py:513-533 ----
    def beam_search(self, x, beam_size, char_discount=0, round_to_inf=1e9, stop_on="patience",
                    validate_solution_function=None, lm=None):
        x = numpy.asarray(x)
        max_length = int(x.shape[0] / self.cfg["max_decoded_length_scale"])
        ignore_first_eol = self.cfg["data_prepend_eos"]
        eol = self.cfg["eos_label"]
        ctx = self.contexts(x)
        Tp = ctx[0].shape[0]
        states = self.initial_states(1, Tp)
        lm_vecs = None
        if lm is not None:
            states["outputs"] = numpy.zeros((1,), numpy.int64)      # LMEmitter.initial_outputs (language_models.py:172-175)
            lm_vecs = [lm["dense"].initial()]
        all_outputs = states["outputs"][None, :]
        all_costs = numpy.zeros_like(all_outputs, dtype=numpy.float32)
        done = []
        min_cost = 1000
        take = lambda v, idx: (v[torch.as_tensor(idx, dtype=torch.int64)] if torch.is_tensor(v)
                               else numpy.take(v, idx, axis=0))
        for i in range(max_length):
            if states["states"].numel() == 0:
                break
            if stop_on == "patience":
                done = sorted(done, key=lambda z: float(z[1][-1]) - float(char_discount) * len(z[1]))
                done = done[:beam_size]
                if done:
                    cur = float(done[0][1][-1]) - float(char_discount) * len(done[0][1])
                    if cur < min_cost:
                        min_cost = cur
                        patience = 30
                    else:
                        patience -= 1
                        if patience == 0:
                            break
            elif stop_on == "optimistic_future_cost":
                if len(done) >= beam_size:
                    optimistic = float(all_costs[-1, :].min()) - float(char_discount) * max_length
                    last = done[beam_size - 1][1]
                    if float(last[-1]) - float(char_discount) * len(last) < optimistic:
                        break
            else:
                raise ValueError("Unknown stopping criterion {}".format(stop_on))
            logprobs = self.logprobs(ctx, states, lm, lm_vecs).astype(numpy.float32)
            assert numpy.isfinite(logprobs).all()
            next_costs = all_costs[-1, :, None] + logprobs
            (indexes, outputs), chosen = smallest(next_costs, beam_size)
            states = {k: take(v, indexes) for k, v in states.items()}
            all_outputs = numpy.take(all_outputs, indexes, axis=1)
            all_costs = numpy.take(all_costs, indexes, axis=1)
            states = self.next_states(ctx, states, outputs)
            if lm is not None:
                lm_vecs = [lm["dense"].step(lm_vecs[j], lm["remap"][int(o)]) for j, o in zip(indexes, outputs)]
            all_outputs = numpy.vstack([all_outputs, outputs[None, :]])
            all_costs = numpy.vstack([all_costs, chosen[None, :]])
            mask = outputs != eol
            if ignore_first_eol and i == 0:
                mask[:] = 1
            for idx in numpy.where((all_outputs[-1] == eol) &
                                   (all_costs[-1] - all_costs[-2] < round_to_inf))[0]:
                if validate_solution_function is None or validate_solution_function(x, all_outputs[:, idx]):
                    done.append((all_outputs[:, idx], all_costs[:, idx]))
            unfinished = numpy.where(mask == 1)[0]
            if lm is not None:
                lm_vecs = [lm_vecs[j] for j in unfinished]
            states = {k: take(v, unfinished) for k, v in states.items()}
            all_outputs = numpy.take(all_outputs, unfinished, axis=1)
            all_costs = numpy.take(all_costs, unfinished, axis=1)
        if not done:
            raise LookupError("CandidateNotFoundError")
        done = sorted(done, key=lambda z: float(z[1][-1]) - float(char_discount) * len(z[1]))
        outs = [[int(t) for t in seq[1:]] for seq, _ in done]
        # search.py:384-407: per-step cost differences of the padded float64 arrays, summed per hypothesis
        costs = [float(numpy.sum(numpy.diff(c.astype(numpy.float64)))) for _, c in done]
        return outs, costs

    def analyze(self, x, labels):
        """recognizer.py:452-494: single-utterance cost vector and alignment (no masks)."""
        with torch.no_grad():
            out = self.cost(numpy.asarray(x)[:, None, :], None, numpy.asarray(labels)[:, None], None)
        return out["cost_matrix"][:, 0].numpy(), out["weights"][:, 0, :].numpy()
