"""Host-side mirror of the attention sequence generator: Blocks `SequenceGenerator`
(libs/blocks/blocks/bricks/sequence_generators.py:910-949) with `AttentionRecurrent` transition
(libs/blocks/blocks/bricks/attention.py:479-772), the lvsr attention bricks (lvsr/bricks/attention.py:42-237),
`Readout` + post-merge (sequence_generators.py:531-707, lvsr/bricks/recognizer.py:298-320) and
`SoftmaxEmitter` (:751-800), as wired by `SpeechRecognizer.__init__` (lvsr/bricks/recognizer.py:250-343).

`cost_matrix(outputs, mask, attended=, attended_mask=)` keeps the reference signature
(sequence_generators.py:317-326); `backward()` is the counterpart of theano.grad through it.  All arithmetic
is in the C-ABI library; torch only owns buffers/views.
"""
import ctypes
import os

import numpy
import torch

ACT_KIND = {"identity": 0, "maxout2": 1, "rectifier": 2, "tanh": 3}
NORMALIZER_KIND = {"softmax": 0, "logistic": 1, "relu": 2}
PRIOR_KIND = {"expanding": 0, "window_around_mean": 1, "window_around_median": 2}
ATT_MS = 32     # match-dim slice per work-group in the energy kernels (csrc/decoder.h)


def _f32(x):
    return float(numpy.float32(x))


class SequenceGenerator(object):
    def __init__(self, dims, store, lib, workspace, use_graph=True, use_persistent=None):
        self.d = dims
        self.store = store
        self.lib = lib
        self.ws = workspace
        self.use_graph = use_graph
        self.use_persistent = use_persistent      # None: auto (persistent label loop and reverse walk when available), True / False: forced on / off
        self.use_persistent_bwd = None            # None: follows use_persistent; False: step kernels for the reverse walk only (probes)
        self._packs = None
        self._pack_cache = {}
        self._gen_cache = {}
        self._saved = None
        g = "/recognizer/generator"
        att = g + "/att_trans/" + ("conv_att" if dims.conv else "cont_att")
        self.n = dict(
            Wpre=att + "/preprocess.W", bpre=att + "/preprocess.b", Ws=att + "/state_trans/transform_states.W",
            we=att + "/energy_comp/linear.W", eb=att + "/energy_comp/linear.b", filters=att + "/conv1d.filters", handler=att + "/handler.W",
            Wdi=g + "/att_trans/distribute/fork_inputs.W", Wdg=g + "/att_trans/distribute/fork_gate_inputs.W",
            Whh=g + "/att_trans/transition.state_to_state", Whg=g + "/att_trans/transition.state_to_gates",
            h0=g + "/att_trans/transition.initial_state",
            Wfi=g + "/fork/fork_inputs.W", bfi=g + "/fork/fork_inputs.b",
            Wfg=g + "/fork/fork_gate_inputs.W", bfg=g + "/fork/fork_gate_inputs.b",
            Wms=g + "/readout/merge/transform_states.W", Wmw=g + "/readout/merge/transform_weighted_averages.W",
            bpm=g + "/readout/post_merge/bias.b", Wout=g + "/readout/post_merge/mlp/linear_%d.W" % len(dims.pm_hidden),
            bout=g + "/readout/post_merge/mlp/linear_%d.b" % len(dims.pm_hidden), bro=g + "/readout/bias.b",
            table=g + "/readout/lookupfeedback/lookuptable.W")
        # further post-merge layers (recognizer.py:309-317): linear_0 .. linear_{n-2} with the activation behind each
        self.pm_hidden = [(g + "/readout/post_merge/mlp/linear_%d.W" % j, g + "/readout/post_merge/mlp/linear_%d.b" % j, w)
                          for j, w in enumerate(dims.pm_hidden)]

    # ---- what a stacked decoder (bricks/generator_stack.py) replaces ----------------------------------------------
    def _state_width(self):
        """Width of the `states` the attention and the readout see."""
        return self.d.D

    def _merge_states_weight(self):
        """readout/merge/transform_states.W as one (state width, P) matrix."""
        return self.store.p[self.n["Wms"]]

    def _initial_state(self):
        return self.store.p[self.n["h0"]]

    def _AW(self, Tp, B):
        """Buffer of AW = attended @ [fork_inputs.W | fork_gate_inputs.W]: (T'*B, 3D) with the rows a multiple of 4 floats apart (the
        persistent backward reads them 16 bytes at a time; D = 250 of the wsj_paper configs gives 750 columns).  -> (view, row stride);
        the padding columns are never written and never contribute (the vector they meet is zero there)."""
        ld = (3 * self.d.D + 3) // 4 * 4
        return self.ws.get("gen.AW", (Tp * B, ld))[:, : 3 * self.d.D], ld

    def _merge_states_backward(self, S2, dR1, gws):
        """Gradient of the readout's state source: weight gradient into the store, -> dS_r (rows, state width)."""
        d, p, g, n, lib, ws = self.d, self.store.p, self.store.g, self.n, self.lib, self.ws
        lib.sgemm(S2, dR1, g[n["Wms"]], transA=True, ws=gws, group=True)
        dS_r = ws.get("gen.dS_r", (S2.shape[0], d.D))
        lib.sgemm(dR1, p[n["Wms"]], dS_r, transB=True)
        return dS_r

    # ---- packed operand copies ------------------------------------------------------------------
    def _packed(self, packs=True):
        """The operand copies of the current parameters.  packs=False: only allocate the packed copies (and make the plain
        concatenations) — the persistent kernels of the teacher-forced pass read the plain weights, so a training step that stays on
        them never pays the nine pack launches; `_ensure_packs` makes them when a step kernel is about to run."""
        if self._packs is not None and self._packs["version"] == self.store.version and not self.lib.capturing:
            if packs:
                self._ensure_packs(self._packs)
            return self._packs
        p, lib, ws, n, d = self.store.p, self.lib, self.ws, self.n, self.d
        ent = dict(version=self.store.version)
        jobs = []

        def pack(key, W, trans=False):
            K, N = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
            buf = ws.get("gen.%s_p" % key, (lib.pack_size(K, N),))
            jobs.append((W, buf, trans))
            ent[key] = buf
        pack("Ws", p[n["Ws"]]); pack("WsT", p[n["Ws"]], True)
        pack("Whg", p[n["Whg"]]); pack("WhgT", p[n["Whg"]], True)
        pack("Whh", p[n["Whh"]]); pack("WhhT", p[n["Whh"]], True)
        pack("Wdi", p[n["Wdi"]]); pack("Wdg", p[n["Wdg"]])
        wd = ws.get("gen.Wd_cat", (d.E, 3 * d.D))
        lib.copy_many([(p[n["Wdi"]], wd[:, : d.D]), (p[n["Wdg"]], wd[:, d.D:])])
        pack("WdT", wd, True)
        ent["_jobs"], ent["packed"] = jobs, False
        if packs:
            self._ensure_packs(ent)
        self._packs = ent
        return ent

    MERGE_ROWS = 64          # rows of a beam step from which the readout's merge products run as 16-row tiles (lvsr_readout_merge)

    def _readout_packs(self):
        """Packed copies of the merge weights for lvsr_readout_merge (many-row readout of the batched beam search), per parameter version."""
        hit = getattr(self, "_ro_packs", None)
        if hit is not None and hit["version"] == self.store.version:
            return hit
        d, p, n, lib, ws = self.d, self.store.p, self.n, self.lib, self.ws
        ent = dict(version=self.store.version, Wms=None)
        Wmw = p[n["Wmw"]]
        ent["Wmw"] = ws.get("gen.Wmw_p", (lib.pack_size(int(Wmw.shape[0]), int(Wmw.shape[1])),))
        jobs = [(Wmw, ent["Wmw"], False)]
        if d.use_states_for_readout:
            Wms = self._merge_states_weight()
            ent["Wms"] = ws.get("gen.Wms_p", (lib.pack_size(int(Wms.shape[0]), int(Wms.shape[1])),))
            jobs.append((Wms.contiguous(), ent["Wms"], False))
        lib.pack_many(jobs)
        self._ro_packs = ent
        return ent

    def _ensure_packs(self, ent):
        if not ent.get("packed", True):
            self.lib.pack_many(ent["_jobs"], use_graph=self.use_graph, cache=self._pack_cache)
            ent["packed"] = True

    def _prior(self):
        pr = self.d.cfg["prior"]
        if not self.d.conv:
            return 0, (0.0, 0.0, 0.0, 0.0)
        kind = PRIOR_KIND.get(pr.get("type", "expanding"))
        if kind is None:
            raise Exception("Unknown prior type: %s" % pr.get("type"))       # lvsr/bricks/attention.py:158-159
        if kind == 0:
            return 0, (float(pr["initial_begin"]), float(pr["initial_end"]), _f32(pr["min_speed"]), _f32(pr["max_speed"]))
        return kind, (float(pr["before"]), float(pr["after"]), 0.0, 0.0)

    # ---- argument block shared by training and generation ------------------------------------------
    def _attdec_fields(self, pk, A, PA, Am, L, B, bufs, phases, step0, broadcast, groups=0, group_Tp=None):
        d, p, n = self.d, self.store.p, self.n
        Tp = int(A.shape[0])
        kind, pp = self._prior()
        if groups:         # batched beam search: rows [g B/groups, (g+1) B/groups) read utterance g (lvsr_attdec_args.group_rows)
            strides = dict(A_ts=groups * d.E, A_bs=d.E, PA_ts=groups * d.M, PA_bs=d.M, Am_ts=groups, Am_bs=1,
                           group_rows=B // groups, step_stride=16, group_Tp=group_Tp)
        elif broadcast:    # one utterance shared by every hypothesis (beam search)
            strides = dict(A_ts=d.E, A_bs=0, PA_ts=d.M, PA_bs=0, Am_ts=1, Am_bs=0)
        else:
            strides = dict(A_ts=B * d.E, A_bs=d.E, PA_ts=B * d.M, PA_bs=d.M, Am_ts=B, Am_bs=1)
        f = dict(Tp=Tp, B=B, L=L, E=d.E, D=d.D, M=d.M, K=d.K, c=d.c, prior_type=kind, step0=step0, phases=phases,
                 p0=pp[0], p1=pp[1], p2=pp[2], p3=pp[3], A=A, PA=PA, Am=Am, Ws_p=pk["Ws"], w_e=p[n["we"]],
                 normalizer=NORMALIZER_KIND[d.normalizer], e_bias=p[n["eb"]] if d.energy_bias else None,
                 filters=p[n["filters"]] if d.conv else None, handler=p[n["handler"]] if d.conv else None,
                 Whg_p=pk["Whg"], Whh_p=pk["Whh"], Wdi_p=pk["Wdi"], Wdg_p=pk["Wdg"])
        f.update(strides)
        f.update(bufs)
        return f

    def _feedback_fork(self, labels_flat, nrows, xg, fb_buf=None):
        """xg (nrows,3D) = fork(feedback(labels)): LookupFeedback / OneOfNFeedback then Fork of two Linear bricks
        (sequence_generators.py:263-264, 839-842; lvsr/bricks/__init__.py:97-104)."""
        d, p, n, lib = self.d, self.store.p, self.n, self.lib
        st = lib.stream_for(xg)
        if d.embed:
            lib.call("lvsr_gather_rows", st, lib_ptr(p[n["table"]]), d.FB, lib_ptr(labels_flat), nrows, d.V + 1, d.FB,
                     None, lib_ptr(fb_buf), d.FB)
            lib.sgemm(fb_buf, p[n["Wfi"]], xg[:, : d.D], bias=p[n["bfi"]])
            lib.sgemm(fb_buf, p[n["Wfg"]], xg[:, d.D:], bias=p[n["bfg"]])
        else:
            lib.call("lvsr_gather_rows", st, lib_ptr(p[n["Wfi"]]), d.D, lib_ptr(labels_flat), nrows, d.FB, d.D,
                     lib_ptr(p[n["bfi"]]), lib_ptr(xg), 3 * d.D)
            lib.call("lvsr_gather_rows", st, lib_ptr(p[n["Wfg"]]), 2 * d.D, lib_ptr(labels_flat), nrows, d.FB, 2 * d.D,
                     lib_ptr(p[n["bfg"]]), lib_ptr(xg[:, d.D:]), 3 * d.D)

    def preprocess(self, attended):
        """attention.preprocess (lvsr/bricks/attention.py:228-230): PA = attended @ W + b."""
        d, p, n = self.d, self.store.p, self.n
        Tp, B = int(attended.shape[0]), int(attended.shape[1])
        PA = self.ws.get("gen.PA", (Tp, B, d.M))
        self.lib.sgemm(attended.view(Tp * B, d.E), p[n["Wpre"]], PA.view(Tp * B, d.M), bias=p[n["bpre"]])
        return PA

    def _readout(self, S2, WA2, nrows, tag):
        """Readout.readout (sequence_generators.py:614-619) + post-merge (recognizer.py:298-320) -> logits."""
        d, p, n, lib, ws = self.d, self.store.p, self.n, self.lib, self.ws
        R1 = ws.get("gen.R1" + tag, (nrows, d.P))
        self._pm_acts = []                       # (input, pre-activation) of every further post-merge layer, for the backward pass
        bias = p[n["bpm"]] if d.post_merge else p[n["bro"]]
        lib.sgemm(WA2, p[n["Wmw"]], R1, bias=bias)
        if d.use_states_for_readout:
            lib.sgemm(S2, self._merge_states_weight(), R1, beta=1.0)
        if not d.post_merge:
            return R1, None, R1
        R2 = ws.get("gen.R2" + tag, (nrows, d.Pout))
        lib.call("lvsr_act_fwd", lib.stream_for(R2), ACT_KIND[d.act], lib_ptr(R1), d.P, nrows, d.P, lib_ptr(R2), d.Pout)
        for j, (wn, bn, width) in enumerate(self.pm_hidden):
            pre = ws.get("gen.pm_pre%d" % j + tag, (nrows, width))
            lib.sgemm(R2, p[wn], pre, bias=p[bn])
            post = ws.get("gen.pm_post%d" % j + tag, (nrows, width))
            lib.call("lvsr_act_fwd", lib.stream_for(post), ACT_KIND[d.act], lib_ptr(pre), width, nrows, width, lib_ptr(post), width)
            self._pm_acts.append((R2, pre))
            R2 = post
        logits = ws.get("gen.logits" + tag, (nrows, d.V))
        lib.sgemm(R2, p[n["Wout"]], logits, bias=p[n["bout"]])
        return R1, R2, logits

    # ---- persistent label loop -------------------------------------------------------------------
    def _persistent_ws(self, fields):
        """Workspace of the persistent decoder kernel for this argument block, or None when the step kernels run: the
        configuration is outside the kernel's limits (lvsr_attdec_persist_ws_bytes == 0), `use_persistent=False`, or — on the
        CPU emulator — its work-groups are not run concurrently."""
        mode = "auto" if self.use_persistent is None else ("1" if self.use_persistent else "0")
        if mode == "0":
            return None
        import ctypes as _ct
        a = self.lib.make("lvsr_attdec_args", **fields)
        nbytes = int(self.lib._lvsr_attdec_persist_ws_bytes(_ct.byref(a)))
        if self.lib.is_emulator and not self.lib.emulates_concurrency():
            nbytes = 0
        if nbytes == 0:
            if mode == "1":
                raise ValueError("persistent decoder kernel requested but not available for this configuration")
            return None
        return self.ws.get("gen.sync", ((nbytes + 3) // 4,), torch.int32)

    def _persistent_bwd_ws(self, fwd_args):
        """Workspace of the persistent backward kernel, or None (`use_persistent` / `use_persistent_bwd` False, outside its limits, or — on the CPU
        emulator — work-groups not run concurrently)."""
        # Default since round 3: 33.4 us per label against 41.5 for the four step kernels (profiles/r03_decoder_bwd_persist_probe.txt),
        # WSJ-base step 18.3 -> 17.4 ms.
        if self.use_persistent is False or self.use_persistent_bwd is False:
            return None
        import ctypes as _ct
        nbytes = int(self.lib._lvsr_attdec_bwd_persist_ws_bytes(_ct.byref(fwd_args)))
        if self.lib.is_emulator and not self.lib.emulates_concurrency():
            nbytes = 0
        if nbytes == 0:
            return None
        return self.ws.get("gen.sync_bwd", ((nbytes + 3) // 4,), torch.int32)

    def check_persistent(self):
        """After a synchronisation point: raise if the persistent decoder kernel gave up waiting for its cluster."""
        for k, buf in self.ws._bufs.items():
            if k[0] in ("gen.sync", "gen.sync_bwd") and int(buf[0]) != 0:
                buf[:16].zero_()          # sticky abort word (no launch clears it): cleared once reported
                raise RuntimeError("persistent decoder kernel aborted (a work-group of a cluster was not scheduled); results since "
                                   "the last check are invalid")

    # ---- teacher-forced cost ---------------------------------------------------------------------
    def cost_matrix(self, outputs, mask=None, attended=None, attended_mask=None, save_for_backward=True):
        """outputs (L,B) int64 labels, mask (L,B) or None, attended (T',B,E), attended_mask (T',B) -> costs (L,B).
        Also keeps `self.last` = dict(weights, energies, states, weighted_averages) (the auxiliary variables
        `SpeechRecognizer.analyze` extracts, recognizer.py:452-494)."""
        d, p, n, lib, ws = self.d, self.store.p, self.n, self.lib, self.ws
        L, B = int(outputs.shape[0]), int(outputs.shape[1])
        Tp = int(attended.shape[0])
        pk = self._packed(packs=False)
        A = attended.contiguous()
        Am = attended_mask.contiguous()
        labels = outputs.contiguous()
        ym = None if mask is None else mask.contiguous()
        PA = self.preprocess(A)
        rc = self._forward_recurrent(pk, A, PA, Am, labels, ym, L, B, Tp)
        bufs, S, W, WA = rc["bufs"], rc["bufs"]["S"], rc["bufs"]["W"], rc["bufs"]["WA"]
        SW = self._state_width()
        S2, WA2 = S[:L].view(L * B, SW), WA.view(L * B, d.E)
        R1, R2, logits = self._readout(S2, WA2, L * B, "")
        cost = ws.get("gen.cost", (L, B))
        dlogits = ws.get("gen.dlogits", (L * B, d.V))
        lib.call("lvsr_softmax_nll", lib.stream_for(cost), lib_ptr(logits), d.V, lib_ptr(labels), lib_ptr(ym), L * B, d.V,
                 lib_ptr(cost), lib_ptr(dlogits), d.V, 1.0, None, 0)
        lm = self.language_model
        self._cost_has_lm = lm is not None
        if lm is not None:
            # SequenceGenerator.evaluate with a language model (sequence_generators.py:286-296): the readout of label i is fused
            # (ShallowFusionReadout) with the look-ahead costs of the FST state set reached by labels[:i], the emitter is LMEmitter:
            # cost = -fused[label].  This is what `analyze` reports in the decode driver when `net.lm` is configured.
            lm_add = self._lm_lookahead(lm, labels, ym, L, B)
            fused = ws.get("gen.fused", (L * B, d.V))
            lib.call("lvsr_shallow_fusion", lib.stream_for(cost), lib_ptr(logits), d.V, lib_ptr(lm_add), L * B, d.V, float(lm.am_beta),
                     float(lm.lm_weight), int(lm.norm[0]), int(lm.norm[1]), int(lm.norm[2]), 1.0, lib_ptr(fused))
            lib.call("lvsr_select_cost", lib.stream_for(cost), lib_ptr(fused), d.V, lib_ptr(labels), lib_ptr(ym), L * B, d.V, -1.0,
                     lib_ptr(cost))
        self.last = dict(weights=W[1:], energies=bufs["EN"], states=S[:L], weighted_averages=WA)
        if save_for_backward:
            self._saved = dict(L=L, B=B, Tp=Tp, A=A, Am=Am, PA=PA, labels=labels, ym=ym, bufs=bufs,
                               R1=R1, R2=R2, dlogits=dlogits, pk=pk, pm_acts=list(self._pm_acts))
            self._saved.update(rc["saved"])
        return cost

    def _forward_recurrent(self, pk, A, PA, Am, labels, ym, L, B, Tp):
        """The label loop of the teacher-forced pass: fork(feedback(labels)) and the AttentionRecurrent scan
        (sequence_generators.py:263-277).  -> dict(bufs = the state slots / per-label tensors (S, W, WA, EN, ...), saved =
        what `_backward_recurrent` needs beyond them)."""
        d, p, n, lib, ws = self.d, self.store.p, self.n, self.lib, self.ws
        xg = ws.get("gen.xg", (L * B, 3 * d.D))
        fb = ws.get("gen.fb", (L * B, d.FB)) if d.embed else None
        self._feedback_fork(labels.view(-1), L * B, xg, fb)
        S = ws.get("gen.S", (L + 1, B, d.D))
        W = ws.get("gen.W", (L + 1, B, Tp))
        S[0].copy_(p[n["h0"]].unsqueeze(0).expand(B, d.D))     # initial_state tiled (recurrent.py:622-624)
        W[0].zero_()
        if d.conv:
            W[0, :, 0] = 1.0                                   # initial_glimpses (lvsr/bricks/attention.py:215-222)
        Kc = max(d.K, 1)
        bufs = dict(xg=xg, ymask=ym, S=S, W=W,
                    pos=ws.get("gen.pos", (L + 1, B)) if (d.conv and self._prior()[0] != 0) else None,
                    WA=ws.get("gen.WA", (L, B, d.E)), EN=ws.get("gen.EN", (L, B, Tp)), ZB=ws.get("gen.ZB", (L, B)),
                    sW=ws.get("gen.sW", (L, B, d.M)),
                    CV=ws.get("gen.CV", (L, B, Kc, Tp)) if d.conv else None,
                    U=ws.get("gen.U", (L, B, d.D)), R=ws.get("gen.R", (L, B, d.D)), C=ws.get("gen.C", (L, B, d.D)),
                    RH=ws.get("gen.RH", (L, B, d.D)), sg=ws.get("gen.sg", (B, 2 * d.D)), xin=ws.get("gen.xin", (B, d.D)),
                    ep=ws.get("gen.ep", (B, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
        fields = self._attdec_fields(pk, A, PA, Am, L, B, bufs, phases=3, step0=0, broadcast=False)
        sync = self._persistent_ws(fields)
        if sync is not None:
            # one persistent launch for the whole label loop (csrc/decoder_persist.hip): a memset and a kernel, no graph needed
            import ctypes as _ct
            fwd_args = lib.make("lvsr_attdec_args", **fields)
            # gate inputs of the glimpse, reassociated: AW = attended @ [fork_inputs.W | fork_gate_inputs.W] once per batch
            wd = ws.get("gen.Wd_cat", (d.E, 3 * d.D))
            AW, AW_ld = self._AW(Tp, B)
            lib.sgemm(A.view(Tp * B, d.E), wd, AW)
            plain = lib.make("lvsr_attdec_plain", Ws=p[n["Ws"]], Whg=p[n["Whg"]], Whh=p[n["Whh"]], AW=AW, AW_ld=AW_ld)
            lib.call("lvsr_attdec_fwd_persistent", lib.stream_for(S), _ct.byref(fwd_args), _ct.byref(plain), lib_ptr(sync), 0)
            lib.call("lvsr_attdec_glimpses", lib.stream_for(S), _ct.byref(fwd_args))
        else:
            self._ensure_packs(pk)
            fwd_args = lib.run("lvsr_attdec_fwd", "lvsr_attdec_args", S, self.use_graph, **fields)
        return dict(bufs=bufs, saved=dict(xg=xg, fb=fb, fields=fields, AW_valid=sync is not None))

    def _lm_lookahead(self, lm, labels, ym, L, B):
        """(L*B, V) look-ahead costs `lm_add` of the teacher-forced label sequences: row (i, b) = FSTCostsOp of the state set after
        labels[:i, b] (LanguageModel.evaluate -> FSTTransition.apply, lvsr/bricks/language_models.py:34-50; masked steps keep the
        previous state set).  One FST step per label for the whole batch — on the device with a DeviceFSTLanguageModel."""
        dev = labels.device
        out = self.ws.get("gen.lm_add", (L, B, self.d.V))
        st = lm.initial_states(B)
        on_dev = getattr(lm, "on_device", False)
        lab = labels if on_dev else labels.cpu().numpy()
        msk = None if ym is None else (ym if on_dev else ym.cpu().numpy())
        for i in range(L):
            out[i].copy_(lm.stage(st, dev))
            if i + 1 == L:
                break
            new = lm.transition(st, lab[i])
            if msk is not None:
                if on_dev:
                    keep = msk[i] > 0
                    st = {k: torch.where(keep.view(-1, *([1] * (v.dim() - 1))), new[k], st[k]) for k, v in new.items()}
                else:
                    keep = msk[i] > 0
                    st = {k: numpy.where(keep.reshape((-1,) + (1,) * (v.ndim - 1)), new[k], st[k]) for k, v in new.items()}
            else:
                st = new
        return out.view(L * B, self.d.V)

    def backward(self):
        """Gradient of sum(cost_matrix) wrt every generator parameter (written to store.g) and wrt `attended`
        (returned, (T',B,E))."""
        d, p, g, n, lib, ws = self.d, self.store.p, self.store.g, self.n, self.lib, self.ws
        sv = self._saved
        assert sv is not None, "cost_matrix() must run first"
        if getattr(self, "_cost_has_lm", False):
            lm = self.language_model
            # with the acoustic readout normalised alone and am_beta = 1 the language-model term is a constant of the parameters
            # and the gradient is the plain one (dlogits of the softmax); the other fusion settings are decode-time options
            if lm is None or tuple(lm.norm) != (True, False, False) or float(lm.am_beta) != 1.0:
                raise NotImplementedError("gradient through ShallowFusionReadout is built for normalize_am_weights only (am_beta = 1)")
        L, B, Tp = sv["L"], sv["B"], sv["Tp"]
        bufs, pk = sv["bufs"], sv["pk"]
        nrows = L * B
        gws = ws.get("gemm_ws", (1 << 22,))
        S2, WA2 = bufs["S"][:L].view(nrows, self._state_width()), bufs["WA"].view(nrows, d.E)
        dlogits, R1, R2 = sv["dlogits"], sv["R1"], sv["R2"]
        # ---- readout backward
        if d.post_merge:
            lib.sgemm(R2, dlogits, g[n["Wout"]], transA=True, ws=gws, group=True)
            lib.colsum(dlogits, g[n["bout"]], ws=gws)
            dR2 = ws.get("gen.dR2", (nrows, R2.shape[1]))
            lib.sgemm(dlogits, p[n["Wout"]], dR2, transB=True)
            for j in range(len(self.pm_hidden) - 1, -1, -1):          # the further post-merge layers, last first
                wn, bn, width = self.pm_hidden[j]
                xin, pre = sv["pm_acts"][j]
                dpre = ws.get("gen.pm_dpre%d" % j, (nrows, width))
                lib.call("lvsr_act_bwd", lib.stream_for(dpre), ACT_KIND[d.act], lib_ptr(pre), width, lib_ptr(dR2), width, nrows, width,
                         lib_ptr(dpre), width)
                lib.sgemm(xin, dpre, g[wn], transA=True, ws=gws, group=True)
                lib.colsum(dpre, g[bn], ws=gws)
                dR2 = ws.get("gen.pm_dx%d" % j, (nrows, xin.shape[1]))
                lib.sgemm(dpre, p[wn], dR2, transB=True)
            dR1 = ws.get("gen.dR1", (nrows, d.P))
            lib.call("lvsr_act_bwd", lib.stream_for(dR1), ACT_KIND[d.act], lib_ptr(R1), d.P, lib_ptr(dR2), d.Pout, nrows, d.P,
                     lib_ptr(dR1), d.P)
            lib.colsum(dR1, g[n["bpm"]], ws=gws)
        else:
            dR1 = dlogits
            lib.colsum(dR1, g[n["bro"]], ws=gws)
        lib.sgemm(WA2, dR1, g[n["Wmw"]], transA=True, ws=gws, group=True)
        dWA_r = ws.get("gen.dWA_r", (nrows, d.E))
        lib.sgemm(dR1, p[n["Wmw"]], dWA_r, transB=True)
        dS_r = None
        if d.use_states_for_readout:
            dS_r = self._merge_states_backward(S2, dR1, gws)
        rb = self._backward_recurrent(sv, dWA_r, dS_r, gws)
        accH, accWe, accEb, DCV, dPA, DWA = rb["accH"], rb["accWe"], rb["accEb"], rb["DCV"], rb["dPA"], rb["DWA"]
        import ctypes
        st = lib.stream_for(dPA)
        lib.colsum(accWe, g[n["we"]].view(-1), ws=gws)
        if d.energy_bias:
            lib.colsum(accEb, g[n["eb"]], ws=gws)
        if d.conv:
            lib.colsum(accH, g[n["handler"]].view(-1), ws=gws)
            # partial sums per chunk of (label, utterance) rows: at most one chunk per row (large per-GPU batches outgrow gemm_ws)
            fws = ws.get("gen.filter_ws", (max(1 << 20, L * B * d.K * (2 * d.c + 1)),))
            lib.call("lvsr_attdec_filter_grad", st, ctypes.byref(rb["fwd_args"]), lib_ptr(DCV), lib_ptr(g[n["filters"]]), lib_ptr(fws),
                     fws.numel() * 4)
        # ---- attended: preprocess backward + glimpse backward
        A2, dPA2 = sv["A"].view(Tp * B, d.E), dPA.view(Tp * B, d.M)
        lib.sgemm(A2, dPA2, g[n["Wpre"]], transA=True, ws=gws, group=True)
        lib.colsum(dPA2, g[n["bpre"]], ws=gws)
        dA = ws.get("gen.dA", (Tp, B, d.E))
        lib.sgemm(dPA2, p[n["Wpre"]], dA.view(Tp * B, d.E), transB=True)
        W = bufs["W"]
        # dA[:, b, :] += alpha_b^T (T',L) @ dwa_b (L,E) for every utterance b: one batched launch
        lib.call("lvsr_sgemm_batched", lib.stream_for(dA), 1, 0, Tp, d.E, L, 1.0, lib_ptr(W[1:]), B * Tp, Tp,
                 lib_ptr(DWA), B * d.E, d.E, 1.0, lib_ptr(dA), B * d.E, d.E, B)
        return dA

    def _backward_recurrent(self, sv, dWA_r, dS_r, gws):
        """Reverse walk over the labels and the weight gradients of the transition, the glimpse distribution, the state
        transformer and the feedback fork.  -> dict(accH, accWe, accEb: partial sums of the handler / energy vector / energy
        bias gradients (folded by the caller), DCV, dPA, DWA, fwd_args: the forward argument block of the attention)."""
        d, p, g, n, lib, ws = self.d, self.store.p, self.store.g, self.n, self.lib, self.ws
        L, B, Tp = sv["L"], sv["B"], sv["Tp"]
        bufs, pk = sv["bufs"], sv["pk"]
        nrows = L * B
        S2, WA2 = bufs["S"][:L].view(nrows, d.D), bufs["WA"].view(nrows, d.E)
        nslice = (d.M + ATT_MS - 1) // ATT_MS
        ntile = (Tp + 63) // 64
        Kc = max(d.K, 1)
        import ctypes
        DXG = ws.get("gen.DXG", (nrows, 3 * d.D))
        # The glimpse contraction is reassociated (as in the persistent forward): q = DXG . AW + QR — the kernels need no dwa and do
        # not write DWA; the total gradient wrt the weighted averages is formed after the walk, accumulated onto the readout's share
        # where it lies: DWA = dWA_r + DXG @ [Wdi | Wdg]^T
        DWA = dWA_r.view(L, B, d.E)
        DSW = ws.get("gen.DSW", (nrows, d.M))
        DCV = ws.get("gen.DCV", (L, B, Kc, Tp)) if d.conv else None
        dPA = ws.get("gen.dPA", (Tp, B, d.M), zero=True)
        ds = ws.get("gen.ds", (B, d.D), zero=True)
        wd = ws.get("gen.Wd_cat", (d.E, 3 * d.D))
        AW, AW_ld = self._AW(Tp, B)
        if not sv.get("AW_valid"):
            lib.sgemm(sv["A"].view(Tp * B, d.E), wd, AW)
        QR = ws.get("gen.QR", (L, B, Tp))
        lib.call("lvsr_sgemm_batched", lib.stream_for(QR), 0, 1, L, Tp, d.E, 1.0, lib_ptr(dWA_r), B * d.E, d.E,
                 lib_ptr(sv["A"]), B * d.E, d.E, 0.0, lib_ptr(QR), B * Tp, Tp, B)
        fwd_args = lib.make("lvsr_attdec_args", **sv["fields"])
        psync = self._persistent_bwd_ws(fwd_args)
        if psync is not None:
            # the whole reverse walk as one persistent launch (csrc/decoder_persist_bwd.hip); its handler / energy-vector / bias
            # gradient partials come one row per work-group — written, not accumulated: nothing but dPA and ds to clear
            P = int(lib._lvsr_attdec_bwd_persist_clusters(ctypes.byref(fwd_args)))        # work-groups per utterance of the launch
            accH = ws.get("gen.accH_p", (B * P, Kc * d.M))
            accWe = ws.get("gen.accWe_p", (B * P, d.M))
            accEb = ws.get("gen.accEb_p", (B * P, 1))
            bw = lib.make("lvsr_attdec_bwd_args", dS_r=dS_r, DXG=DXG, DSW=DSW, DCV=DCV, dPA=dPA, accH=accH, accWe=accWe, accEb=accEb,
                          ds=ds, AW=AW, QR=QR, AW_ld=AW_ld)
            bw.f = fwd_args
            plain = lib.make("lvsr_attdec_plain", Ws=p[n["Ws"]], Whg=p[n["Whg"]], Whh=p[n["Whh"]], AW=AW, AW_ld=AW_ld)
            lib.call("lvsr_attdec_bwd_persistent", lib.stream_for(ds), ctypes.byref(bw), ctypes.byref(plain), lib_ptr(psync))
        else:
            self._ensure_packs(pk)
            accH = ws.get("gen.accH", (B * ntile, Kc * d.M), zero=True)
            accWe = ws.get("gen.accWe", (B * ntile, d.M), zero=True)
            accEb = ws.get("gen.accEb", (B * ntile, 1), zero=True)
            bw = lib.make("lvsr_attdec_bwd_args", WhhT_p=pk["WhhT"], WhgT_p=pk["WhgT"], WdT_p=pk["WdT"], WsT_p=pk["WsT"],
                          dWA_r=dWA_r, dS_r=dS_r, DXG=DXG, DWA=DWA, DSW=DSW, DCV=DCV, dPA=dPA, accH=accH, accWe=accWe, accEb=accEb,
                          ds=ds, dalp=ws.get("gen.dalp", (B, Kc, Tp), zero=True), dspart=ws.get("gen.dspart", (B, d.D)),
                          dsacc=ws.get("gen.dsacc", (B, d.D)), Q=ws.get("gen.Q", (B, Tp)),
                          dcvp=ws.get("gen.dcvp", (B, nslice, Kc, Tp)) if d.conv else None,
                          dswp=ws.get("gen.dswp", (B, ntile, d.M)), AW=AW, QR=QR, AW_ld=AW_ld)
            bw.f = fwd_args
            lib.call("lvsr_attdec_bwd", lib.stream_for(ds), ctypes.byref(bw), int(self.use_graph))
        lib.sgemm(DXG, wd, dWA_r, transB=True, beta=1.0)
        # ---- weight gradients as batched GEMMs over all steps
        dpc, dg = DXG[:, : d.D], DXG[:, d.D:]
        RH2 = bufs["RH"].view(nrows, d.D)
        lib.sgemm(RH2, dpc, g[n["Whh"]], transA=True, ws=gws, group=True)
        lib.sgemm(S2, dg, g[n["Whg"]], transA=True, ws=gws, group=True)
        lib.sgemm(WA2, dpc, g[n["Wdi"]], transA=True, ws=gws, group=True)
        lib.sgemm(WA2, dg, g[n["Wdg"]], transA=True, ws=gws, group=True)
        lib.sgemm(S2, DSW, g[n["Ws"]], transA=True, ws=gws, group=True)
        lib.colsum(ds, g[n["h0"]], ws=gws)
        lib.colsum(dpc, g[n["bfi"]], ws=gws)
        lib.colsum(dg, g[n["bfg"]], ws=gws)
        st = lib.stream_for(ds)
        labels_flat = sv["labels"].view(-1)
        if d.embed:
            fb = sv["fb"]
            lib.sgemm(fb, dpc, g[n["Wfi"]], transA=True, ws=gws, group=True)
            lib.sgemm(fb, dg, g[n["Wfg"]], transA=True, ws=gws, group=True)
            dfb = ws.get("gen.dfb", (nrows, d.FB))
            lib.sgemm(dpc, p[n["Wfi"]], dfb, transB=True)
            lib.sgemm(dg, p[n["Wfg"]], dfb, transB=True, beta=1.0)
            lib.call("lvsr_scatter_add_rows", st, lib_ptr(dfb), d.FB, lib_ptr(labels_flat), nrows, d.V + 1, d.FB,
                     lib_ptr(g[n["table"]]), d.FB, 0.0)
        else:
            lib.call("lvsr_scatter_add_rows", st, lib_ptr(dpc), 3 * d.D, lib_ptr(labels_flat), nrows, d.FB, d.D,
                     lib_ptr(g[n["Wfi"]]), d.D, 0.0)
            lib.call("lvsr_scatter_add_rows", st, lib_ptr(dg), 3 * d.D, lib_ptr(labels_flat), nrows, d.FB, 2 * d.D,
                     lib_ptr(g[n["Wfg"]]), 2 * d.D, 0.0)
        return dict(accH=accH, accWe=accWe, accEb=accEb, DCV=DCV, dPA=dPA, DWA=DWA, fwd_args=fwd_args)


def lib_ptr(t):
    from ..native import ptr
    return ptr(t)


# ---- generation mode (what BeamSearch drives: libs/blocks/blocks/search.py:97-142) -----------------------
def _generation_methods():
    def init_generation(self, attended, attended_mask):
        """context_computer (search.py:97-104): keep the single utterance's contexts; every hypothesis of the
        beam reads them with a zero batch stride (no tiling as in search.py:336-338)."""
        d = self.d
        Tp, N = int(attended.shape[0]), int(attended.shape[1])
        A = attended.contiguous()
        PA = self.preprocess(A)
        if N == 1:
            self._gen = dict(Tp=Tp, A=A.view(Tp, d.E), Am=attended_mask.contiguous().view(Tp), PA=PA.view(Tp, d.M))
        else:
            # several utterances at once (BeamSearch.search_batch): the beams of all of them advance in the same launches, every
            # row reading the contexts of its own utterance (`beam_begin` with groups); `lengths` = attended length of each
            Am = attended_mask.contiguous()
            self._gen = dict(Tp=Tp, A=A, Am=Am, PA=PA, N=N, lengths=self.ws.get("gen.glen", (N,), torch.int32))
            self._gen["lengths"].copy_(Am.sum(dim=0).to(torch.int32))

    def generation_initial_states(self, n=1):
        """initial_state_computer (search.py:106-110): states = tiled initial_state (recurrent.py:622-624),
        weights = initial glimpses (lvsr/bricks/attention.py:215-222), outputs = num_phonemes
        (recognizer.py:286, sequence_generators.py:793-795), step = 0."""
        d, p, n_ = self.d, self.store.p, self.n
        Tp = self._gen["Tp"]
        S = p[n_["h0"]].unsqueeze(0).expand(n, d.D).clone()
        W = torch.zeros(n, Tp, dtype=torch.float32, device=S.device)
        if d.conv:
            W[:, 0] = 1.0
        # initial outputs: SoftmaxEmitter(initial_output=num_phonemes) (recognizer.py:286); with a language model the
        # emitter is LMEmitter whose initial_outputs are zeros (language_models.py:172-175)
        first = 0 if self.language_model is not None else d.V
        return dict(states=S, weights=W, step=0, outputs=numpy.full((n,), first, dtype=numpy.int64))

    def _gen_run(self, S, W, step0, phases, outputs=None):
        d, lib, ws, g = self.d, self.lib, self.ws, self._gen
        n, Tp = int(S.shape[0]), g["Tp"]
        pk = self._packed()
        # the buffer views and the argument block of a (beam width, phases) pair are built once per set of contexts: the
        # search loop calls this twice per emitted character and the Python bookkeeping was ~0.4 ms of each step
        key = (n, phases, g["A"].data_ptr(), g["PA"].data_ptr(), g["Am"].data_ptr(), Tp, id(pk))
        ent = self._gen_cache.get(key)
        if ent is None or ent["generation"] != ws.generation:
            tag = ".n%d" % n
            Sb = ws.get("gs.S" + tag, (2, n, d.D))
            Wb = ws.get("gs.W" + tag, (2, n, Tp))
            xg = y = fb = None
            if phases & 2:
                xg = ws.get("gs.xg" + tag, (n, 3 * d.D))
                y = ws.get("gs.y" + tag, (n,), torch.int64)
                fb = ws.get("gs.fb" + tag, (n, d.FB)) if d.embed else None
            Kc = max(d.K, 1)
            bufs = dict(xg=xg, ymask=None, S=Sb, W=Wb,
                        pos=ws.get("gs.pos" + tag, (2, n)) if (d.conv and self._prior()[0] != 0) else None,
                        WA=ws.get("gs.WA" + tag, (1, n, d.E)), EN=ws.get("gs.EN" + tag, (1, n, Tp)), ZB=None,
                        sW=ws.get("gs.sW" + tag, (1, n, d.M)), CV=ws.get("gs.CV" + tag, (1, n, Kc, Tp)) if d.conv else None,
                        U=ws.get("gs.U" + tag, (1, n, d.D)), R=ws.get("gs.R" + tag, (1, n, d.D)),
                        C=ws.get("gs.C" + tag, (1, n, d.D)), RH=ws.get("gs.RH" + tag, (1, n, d.D)),
                        sg=ws.get("gs.sg" + tag, (n, 2 * d.D)), xin=ws.get("gs.xin" + tag, (n, d.D)),
                        ep=ws.get("gs.ep" + tag, (n, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
            fields = self._attdec_fields(pk, g["A"], g["PA"], g["Am"], 1, n, bufs, phases=phases, step0=int(step0), broadcast=True)
            ent = dict(generation=ws.generation, bufs=bufs, args=lib.make("lvsr_attdec_args", **fields), y=y, fb=fb, xg=xg,
                       S0=Sb[0], W0=Wb[0])
            if len(self._gen_cache) > 64:
                self._gen_cache.clear()
            self._gen_cache[key] = ent
        ent["S0"].copy_(S)
        ent["W0"].copy_(W)
        if phases & 2:
            ent["y"].copy_(torch.as_tensor(outputs, dtype=torch.int64), non_blocking=False)
            self._feedback_fork(ent["y"], n, ent["xg"], ent["fb"])
        ent["args"].step0 = int(step0)
        lib.call("lvsr_attdec_fwd", lib.stream_for(ent["S0"]), ctypes.byref(ent["args"]), 0)
        return ent["bufs"]

    def generation_logprobs(self, S, W, step0):
        """logprobs_computer (search.py:126-134): take_glimpses -> readout -> -log_softmax; (n,V) device tensor."""
        d, lib, ws = self.d, self.lib, self.ws
        n = int(S.shape[0])
        bufs = self._gen_run(S, W, step0, phases=1)
        _, _, logits = self._readout(bufs["S"][0], bufs["WA"][0], n, ".gen%d" % n)
        nl = ws.get("gs.neglogp.n%d" % n, (n, d.V))
        lm = self.language_model
        if lm is not None:
            # ShallowFusionReadout + LMEmitter: the fused readout IS the log-probability; costs = -readout
            add = ws.get("gs.lm_add.n%d" % n, (n, d.V))
            add.copy_(lm.device_add)
            lib.call("lvsr_shallow_fusion", lib.stream_for(nl), lib_ptr(logits), d.V, lib_ptr(add), n, d.V, lm.am_beta,
                     lm.lm_weight, int(lm.norm[0]), int(lm.norm[1]), int(lm.norm[2]), -1.0, lib_ptr(nl))
            return nl
        lib.call("lvsr_softmax_nll", lib.stream_for(nl), lib_ptr(logits), d.V, None, None, n, d.V, None, None, 0, 1.0,
                 lib_ptr(nl), d.V)
        return nl

    def generation_next_states(self, S, W, step0, outputs):
        """next_state_computer (search.py:112-124): take_glimpses AGAIN on the re-arranged hypotheses (the window
        of the location prior depends on the batch it is computed for) + compute_states with the chosen outputs."""
        bufs = self._gen_run(S, W, step0, phases=3, outputs=outputs)
        # hand the new states out in buffers no later generation call writes to (the step buffers are reused by the very
        # next generation_logprobs, which would otherwise overwrite the alignments the caller still holds)
        n = int(S.shape[0])
        So = self.ws.get("gs.Sout.n%d" % n, tuple(bufs["S"][1].shape))
        Wo = self.ws.get("gs.Wout.n%d" % n, tuple(bufs["W"][1].shape))
        So.copy_(bufs["S"][1])
        Wo.copy_(bufs["W"][1])
        return dict(states=So, weights=Wo, step=int(step0) + 1,
                    weighted_averages=bufs["WA"][0], outputs=numpy.asarray(outputs))

    return dict(init_generation=init_generation, generation_initial_states=generation_initial_states, _gen_run=_gen_run,
                generation_logprobs=generation_logprobs, generation_next_states=generation_next_states)


for _k, _v in _generation_methods().items():
    setattr(SequenceGenerator, _k, _v)
SequenceGenerator.language_model = None


# ---- free-running generation: SequenceGenerator.generate / initial_states, SoftmaxEmitter.emit ---------------------------
def _sampling_methods():
    def initial_states(self, batch_size, attended=None, attended_mask=None):
        """BaseSequenceGenerator.initial_states (sequence_generators.py:404-422): initial values of the `generate` states:
        states = tiled initial_state (recurrent.py:622-624), outputs = SoftmaxEmitter.initial_outputs (num_phonemes,
        recognizer.py:286), glimpses = initial_glimpses (lvsr/bricks/attention.py:215-222)."""
        d = self.d
        h0 = self._initial_state()
        dev = h0.device
        B = int(batch_size)
        Tp = int(attended.shape[0]) if attended is not None else 0
        out = dict(states=h0.unsqueeze(0).expand(B, self._state_width()).clone(),
                   outputs=torch.full((B,), d.V, dtype=torch.int64, device=dev),
                   weighted_averages=torch.zeros(B, d.E, device=dev), weights=torch.zeros(B, Tp, device=dev))
        if d.conv:
            if Tp:
                out["weights"][:, 0] = 1.0
            out["energies"] = out["weights"].clone()
            out["step"] = torch.zeros(B, dtype=torch.int64, device=dev)
        return out

    def emit(self, readouts, uniforms=None, seed=None):
        """SoftmaxEmitter.emit (sequence_generators.py:770-776) on readouts (n,V): one class per row by inverse CDF of the
        softmax at a uniform number per row.  `uniforms` (n) in [0,1) or None = drawn from torch's Philox generator
        (`seed`; the reference draws from Theano's MRG31k3p: another stream of the same distribution).  -> (outputs, costs)."""
        d, lib = self.d, self.lib
        x = readouts.contiguous()
        n = int(x.shape[0])
        u = self._uniforms((n,), uniforms, seed, x.device)
        out = torch.empty(n, dtype=torch.int64, device=x.device)
        cost = torch.empty(n, dtype=torch.float32, device=x.device)
        lib.call("lvsr_softmax_emit", lib.stream_for(x), lib_ptr(x), int(x.stride(0)), lib_ptr(u), n, int(x.shape[1]), lib_ptr(out),
                 lib_ptr(cost))
        return out, cost

    def _uniforms(self, shape, uniforms, seed, dev):
        if uniforms is not None:
            u = torch.as_tensor(numpy.ascontiguousarray(uniforms, dtype=numpy.float32)) if not torch.is_tensor(uniforms) else uniforms
            u = u.to(dev).to(torch.float32).contiguous()
            assert tuple(u.shape) == tuple(shape), "uniforms must have shape %s" % (tuple(shape),)
            return u
        if dev.type == "cuda":
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(seed if seed is not None else 1))
            return torch.rand(shape, generator=gen, device=dev, dtype=torch.float32)
        gen = torch.Generator()
        gen.manual_seed(int(seed if seed is not None else 1))
        return torch.rand(shape, generator=gen, dtype=torch.float32)

    def generate(self, n_steps=None, batch_size=None, attended=None, attended_mask=None, uniforms=None, seed=None):
        """BaseSequenceGenerator.generate under its `recurrent` wrapper (sequence_generators.py:328-377): per step
        take_glimpses(previous states and glimpses) -> readout -> emit -> cost -> feedback/fork -> compute_states, all on the
        device with no host synchronisation.  attended (T',B,E), attended_mask (T',B); `uniforms` (n_steps,B) fixes the
        draws (see `emit`).  -> dict(states (n,B,D), outputs (n,B) int64, weighted_averages (n,B,E), weights (n,B,T'),
        energies (n,B,T'), costs (n,B))."""
        d, p, n, lib, ws = self.d, self.store.p, self.n, self.lib, self.ws
        if self.language_model is not None:
            # With a language model the reference's emitter is LMEmitter, whose `emit` returns zeros "that should never be used"
            # (lvsr/bricks/language_models.py:160-163): free-running generation is not a path of the reference there — beam search
            # (which picks the outputs itself) and the teacher-forced cost (cost_matrix / analyze, built) are.
            raise NotImplementedError("generate() with a language model: the reference's LMEmitter.emit returns zeros (not a "
                                      "sampling path); use beam_search, or cost / analyze for teacher-forced costs")
        N = int(n_steps)
        Tp, B = int(attended.shape[0]), int(attended.shape[1])
        assert batch_size is None or int(batch_size) == B
        pk = self._packed()
        A, Am = attended.contiguous(), attended_mask.contiguous()
        PA = self.preprocess(A)
        dev = A.device
        u = self._uniforms((N, B), uniforms, seed, dev)
        Kc = max(d.K, 1)
        pos_needed = d.conv and self._prior()[0] != 0
        S = ws.get("sg.S", (N + 1, B, d.D))
        W = ws.get("sg.W", (N + 1, B, Tp))
        first = self.initial_states(B, attended=A)
        S[0].copy_(first["states"])
        W[0].copy_(first["weights"])
        full = dict(xg=ws.get("sg.xg", (N, B, 3 * d.D)), S=S, W=W, pos=ws.get("sg.pos", (N + 1, B)) if pos_needed else None,
                    WA=ws.get("sg.WA", (N, B, d.E)), EN=ws.get("sg.EN", (N, B, Tp)), sW=ws.get("sg.sW", (N, B, d.M)),
                    CV=ws.get("sg.CV", (N, B, Kc, Tp)) if d.conv else None, U=ws.get("sg.U", (N, B, d.D)),
                    R=ws.get("sg.R", (N, B, d.D)), C=ws.get("sg.C", (N, B, d.D)), RH=ws.get("sg.RH", (N, B, d.D)))
        shared = dict(ymask=None, ZB=None, sg=ws.get("sg.sg", (B, 2 * d.D)), xin=ws.get("sg.xin", (B, d.D)),
                      ep=ws.get("sg.ep", (B, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
        outputs = ws.get("sg.outputs", (N, B), torch.int64)
        costs = ws.get("sg.costs", (N, B))
        fb = ws.get("sg.fb", (B, d.FB)) if d.embed else None
        if pos_needed:
            full["pos"][0].zero_()
        st = lib.stream_for(S)
        for t in range(N):
            bufs = {k: (None if v is None else v[t:]) for k, v in full.items()}
            bufs.update(shared)
            skip = 4 if pos_needed else 0
            fa = self._attdec_fields(pk, A, PA, Am, 1, B, bufs, phases=1 | skip, step0=t, broadcast=False)
            lib.call("lvsr_attdec_fwd", st, ctypes.byref(lib.make("lvsr_attdec_args", **fa)), 0)
            ra = self._readout_step_args(S[t], full["WA"][t], B, uniforms=u[t], outputs=outputs[t], costs=costs[t])
            lib.call("lvsr_readout_step", st, ctypes.byref(ra))
            self._feedback_fork(outputs[t], B, full["xg"][t], fb)
            fg = self._attdec_fields(pk, A, PA, Am, 1, B, bufs, phases=2, step0=t, broadcast=False)
            lib.call("lvsr_attdec_fwd", st, ctypes.byref(lib.make("lvsr_attdec_args", **fg)), 0)
        return dict(states=S[1:], outputs=outputs, weighted_averages=full["WA"], weights=W[1:], energies=full["EN"], costs=costs)

    return dict(initial_states=initial_states, emit=emit, _uniforms=_uniforms, generate=generate)


for _k, _v in _sampling_methods().items():
    setattr(SequenceGenerator, _k, _v)


# ---- device-resident beam search (csrc/beam.hip): state of one search + the launches of one position ------------------
CTL = dict(nlive=0, pos=1, done=2, nfin=3, patience=4, nsel=5, err=6, steps=7)


def _beam_methods():
    def beam_begin(self, K, eol, max_length, ignore_first_eol=False, char_discount=0.0, round_to_inf=1e9, stop_on="patience",
                   force_merge=False):
        """Allocate (once per (K, T', max_length)) and reset the device state of a beam search over the contexts set by
        `init_generation`: hypothesis 0 = initial state / initial glimpses (search.py:287-299), every other row a copy.
        After `init_generation` on a batch of N utterances: N searches side by side (`max_length` = one limit per utterance),
        search g in rows [g K, g K + K) of every state buffer and block g of the bookkeeping buffers (lvsr_beam_args.groups)."""
        d, p, n_, lib, ws, g = self.d, self.store.p, self.n, self.lib, self.ws, self._gen
        Tp, dev = g["Tp"], g["A"].device
        lm = self.language_model
        on_dev_lm = lm is not None and getattr(lm, "on_device", False)
        G = int(g.get("N", 1))
        if G > 1:
            limits = [int(m) for m in max_length]
            assert len(limits) == G
            assert lm is None or on_dev_lm, "batched search: language model on the device"
            # capacity of the history buffers (and part of the step graph's key): the largest limit, rounded up so that batches of
            # slightly different lengths share buffers and graph; every search stops at its own limit (ctl word 8)
            max_length = (max(limits) + 31) // 32 * 32
            return self._beam_begin(K * G, K, G, limits, eol, max_length, ignore_first_eol, char_discount, round_to_inf, stop_on, True)
        return self._beam_begin(K, K, 1, None, eol, max_length, ignore_first_eol, char_discount, round_to_inf, stop_on, force_merge)

    def _beam_begin(self, R, K, G, limits, eol, max_length, ignore_first_eol, char_discount, round_to_inf, stop_on, force_merge=False):
        """R = G * K rows: G searches of beam K.  `force_merge`: the readout's merge products through lvsr_readout_merge whatever
        the number of rows (every batched decode, so that an utterance's hypotheses do not depend on how many others share its
        launches: a row of that kernel is summed in the same order in any batch)."""
        d, p, n_, lib, ws, g = self.d, self.store.p, self.n, self.lib, self.ws, self._gen
        Tp, dev = g["Tp"], g["A"].device
        lm = self.language_model
        on_dev_lm = lm is not None and getattr(lm, "on_device", False)
        tag = ".K%d" % K if G == 1 else ".K%dx%d" % (K, G)
        K_one, K = K, R            # below, K counts ROWS of the state buffers; K_one is the beam size
        i32, i64, f64 = torch.int32, torch.int64, torch.float64
        fin_cap = 2 * K_one if stop_on == "patience" else K_one * (max_length + 1)
        Kc = max(d.K, 1)
        pos_needed = d.conv and self._prior()[0] != 0
        SW = self._state_width()
        stacked = d.n_dec > 1          # bricks/generator_stack.py: pass B is the attention block and one GRU block per layer

        def attbufs(which, phases):
            t = tag + which
            return dict(xg=ws.get("bs.xg" + t, (K, 3 * d.D)) if phases & 2 else None, ymask=None,
                        S=ws.get("bs.S" + t, (2, K, SW)), W=ws.get("bs.W" + t, (2, K, Tp)),
                        pos=ws.get("bs.pos" + t, (2, K)) if pos_needed else None,
                        WA=ws.get("bs.WA" + t, (1, K, d.E)), EN=ws.get("bs.EN" + t, (1, K, Tp)), ZB=None,
                        sW=ws.get("bs.sW" + t, (1, K, d.M)), CV=ws.get("bs.CV" + t, (1, K, Kc, Tp)) if d.conv else None,
                        U=ws.get("bs.U" + t, (1, K, d.D)), R=ws.get("bs.R" + t, (1, K, d.D)),
                        C=ws.get("bs.C" + t, (1, K, d.D)), RH=ws.get("bs.RH" + t, (1, K, d.D)),
                        sg=ws.get("bs.sg" + t, (K, 2 * d.D)), xin=ws.get("bs.xin" + t, (K, d.D)),
                        ep=ws.get("bs.ep" + t, (K, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
        A_, B_ = attbufs("a", 1), attbufs("b", 3)
        gshape = (lambda *dims: dims) if G == 1 else (lambda *dims: (G,) + dims)
        st = dict(K=K_one, rows=K, groups=G, Tp=Tp, max_length=int(max_length), limits=limits, fin_cap=fin_cap, A=A_, B=B_,
                  ctl=ws.get("bs.ctl" + tag, gshape(16), i32), fctl=ws.get("bs.fctl" + tag, gshape(4)),
                  neglogp=ws.get("bs.neglogp" + tag, (K, d.V)), running=ws.get("bs.running" + tag, (K,)),
                  live_col=ws.get("bs.live_col" + tag, (K,), i32),
                  hist_parent=ws.get("bs.hist_parent" + tag, gshape(max_length, K_one), i32),
                  hist_char=ws.get("bs.hist_char" + tag, gshape(max_length, K_one), i32),
                  hist_cost=ws.get("bs.hist_cost" + tag, gshape(max_length, K_one)),
                  fin_pos=ws.get("bs.fin_pos" + tag, gshape(fin_cap), i32), fin_col=ws.get("bs.fin_col" + tag, gshape(fin_cap), i32),
                  fin_cost=ws.get("bs.fin_cost" + tag, gshape(fin_cap)), fin_score=ws.get("bs.fin_score" + tag, gshape(fin_cap)),
                  keep=ws.get("bs.keep" + tag, (K,), i32), chars=ws.get("bs.chars" + tag, (K,), i64),
                  parents=ws.get("bs.parents" + tag, (K,), i32),
                  fb=ws.get("bs.fb" + tag, (K, d.FB)) if d.embed else None, lm=None)
        if lm is not None:
            st["lm"] = dict(add_live=ws.get("bs.lm_add_live" + tag, (K, d.V)))
            if on_dev_lm:
                st["lm"].update(states_live=ws.get("bs.lm_sl" + tag, (K, 7), i64), weights_live=ws.get("bs.lm_wl" + tag, (K, 7), f64),
                                states_sel=ws.get("bs.lm_ss" + tag, (K, 7), i64), weights_sel=ws.get("bs.lm_ws" + tag, (K, 7), f64),
                                states_new=ws.get("bs.lm_sn" + tag, (K, 7), i64), weights_new=ws.get("bs.lm_wn" + tag, (K, 7), f64),
                                add_new=ws.get("bs.lm_add_new" + tag, (K, d.V)))
        pk = self._packed()
        # window centres travel with the rows (select / compact move them), so neither pass recomputes slot 0: phases bit 2
        skip_pos = 4 if pos_needed else 0
        grp = dict(groups=G, group_Tp=g["lengths"]) if G > 1 else {}
        fa = self._attdec_fields(pk, g["A"], g["PA"], g["Am"], 1, K, A_, phases=1 | skip_pos, step0=0, broadcast=True, **grp)
        pos_word = st["ctl"].view(-1)[CTL["pos"]:]
        st["argsA"] = lib.make("lvsr_attdec_args", step_dev=pos_word, **fa)
        if stacked:
            st["stepB"] = self._beam_step_blocks(pk, g, K, B_, skip_pos, pos_word, tag, **grp)
        else:
            fb_ = self._attdec_fields(pk, g["A"], g["PA"], g["Am"], 1, K, B_, phases=3 | skip_pos, step0=-1, broadcast=True, **grp)
            # the select launch raises word 9 of a search's control block when pass B's glimpses would repeat pass A's row by row
            # (and copies them): lvsr_beam_args.WA_live
            st["argsB"] = lib.make("lvsr_attdec_args", step_dev=pos_word, skip=st["ctl"].view(-1)[9:], skip_stride=16, **fb_)     # runs after the select kernel moved on: step0 = -1
        L = st["lm"] or {}
        host_fork = d.embed or stacked      # the fork of the chosen characters as separate launches of pass B
        st["args"] = lib.make(
            "lvsr_beam_args", K=K_one, groups=G, V=d.V, eol=int(eol), ignore_first_eol=int(bool(ignore_first_eol)),
            stop_on={"patience": 0, "optimistic_future_cost": 1}[stop_on], max_length=int(max_length), fin_cap=fin_cap, D=SW, Tp=Tp,
            round_to_inf=float(numpy.float32(min(float(round_to_inf), 3.0e38))), char_discount=float(char_discount),
            ctl=st["ctl"], fctl=st["fctl"], neglogp=st["neglogp"], running=st["running"], live_col=st["live_col"],
            hist_parent=st["hist_parent"], hist_char=st["hist_char"], hist_cost=st["hist_cost"], fin_pos=st["fin_pos"],
            fin_col=st["fin_col"], fin_cost=st["fin_cost"], fin_score=st["fin_score"], keep=st["keep"], chars=st["chars"],
            parents=st["parents"], S_live=A_["S"][0], W_live=A_["W"][0], S_sel=B_["S"][0], W_sel=B_["W"][0],
            lm_states_live=L.get("states_live"), lm_weights_live=L.get("weights_live"), lm_states_sel=L.get("states_sel"),
            lm_weights_sel=L.get("weights_sel"), S_new=B_["S"][1], W_new=B_["W"][1], S_live_out=A_["S"][0], W_live_out=A_["W"][0],
            lm_states_new=L.get("states_new"), lm_weights_new=L.get("weights_new"), lm_add_new=L.get("add_new"),
            lm_states_live_out=L.get("states_live"), lm_weights_live_out=L.get("weights_live"),
            lm_add_live_out=L.get("add_live") if on_dev_lm else None,
            pos_live=A_["pos"][0] if pos_needed else None, pos_sel=B_["pos"][0] if pos_needed else None,
            pos_new=B_["pos"][1] if pos_needed else None, pos_live_out=A_["pos"][0] if pos_needed else None,
            # one-hot feedback: the fork of the chosen characters is a row gather the select launch does itself
            fork_xg=None if host_fork else B_["xg"], fork_Wi=None if host_fork else p[n_["Wfi"]],
            fork_Wg=None if host_fork else p[n_["Wfg"]], fork_bi=None if host_fork else p[n_["bfi"]],
            fork_bg=None if host_fork else p[n_["bfg"]], fork_rows=0 if host_fork else d.FB,
            **({} if stacked else dict(WA_live=A_["WA"][0], WA_sel=B_["WA"][0], W1_live=A_["W"][1], W1_sel=B_["W"][1], E=d.E,
                                       pos1_live=A_["pos"][1] if pos_needed else None, pos1_sel=B_["pos"][1] if pos_needed else None)))
        # many rows (batched search): the merge products as 16-row tiles in a launch of their own (lvsr_readout_merge)
        R1 = ws.get("bs.R1" + tag, (K, d.P)) if (K >= self.MERGE_ROWS or force_merge) else None
        if R1 is not None:
            ro = self._readout_packs()
            st["merge"] = (lib_ptr(A_["S"][0]), int(A_["S"][0].stride(0)), lib_ptr(A_["WA"][0]), int(A_["WA"][0].stride(0)), K, SW, d.E, d.P,
                           lib_ptr(ro["Wms"]), lib_ptr(ro["Wmw"]), lib_ptr(p[n_["bpm"]] if d.post_merge else p[n_["bro"]]), lib_ptr(R1), int(R1.stride(0)))
        st["readout"] = self._readout_step_args(A_["S"][0], A_["WA"][0], K, neglogp=st["neglogp"],
                                                lm_add=L.get("add_live") if lm is not None else None, R1=R1)
        # ---- reset: one live hypothesis, replicated over the K rows
        ctl0 = numpy.zeros((G, 16), numpy.int32)
        ctl0[:, CTL["nlive"]], ctl0[:, CTL["patience"]] = 1, -1
        if G > 1:
            ctl0[:, 8] = limits            # every search's own position limit
        st["ctl"].copy_(torch.from_numpy(ctl0.reshape(st["ctl"].shape)))
        st["fctl"].copy_(torch.tensor([1000.0, 0.0, 0.0, 0.0]).repeat(G).view(st["fctl"].shape))
        st["running"].zero_()
        st["live_col"].zero_()
        A_["S"][0].copy_(self._initial_state().unsqueeze(0).expand(K, SW))
        A_["W"][0].zero_()
        if d.conv:
            A_["W"][0][:, 0] = 1.0
        if pos_needed:
            A_["pos"].zero_()         # centre of the initial glimpses [1, 0, ...]: mean 0, no median crossing -> 0
        if on_dev_lm:
            first = lm.initial_states(K)
            L["states_live"].copy_(first["states"])
            L["weights_live"].copy_(first["weights"])
            L["add_live"].copy_(first["add"])
        st["on_dev_lm"] = on_dev_lm
        st["stop_on"] = stop_on
        # everything the captured launches depend on belongs to the key: the language model's weighting and normalisation flags are
        # kernel arguments of the readout, its tables and error word are pointers inside the FST walk's argument block
        lm_key = None if lm is None else (float(lm.lm_weight), float(lm.am_beta), tuple(bool(v) for v in lm.norm), float(getattr(lm, "no_transition_cost", 0.0)))
        lm_ptrs = () if not on_dev_lm else tuple(sorted((k, t.data_ptr()) for k, t in lm._dev.items())) + (lm._err.data_ptr(),)
        st["key"] = ("beam_step", K_one, G, Tp, int(max_length), stop_on, int(bool(ignore_first_eol)), int(eol), float(char_discount),
                     float(round_to_inf), lm is not None, on_dev_lm, lm_key, R1 is not None)
        st["volatile"] = (g["A"].data_ptr(), g["PA"].data_ptr(), g["Am"].data_ptr(), ws.generation, id(pk), self.store.version, lm_ptrs)
        self._beam = st
        return st

    def _readout_step_args(self, S2, WA2, n, neglogp=None, lm_add=None, uniforms=None, outputs=None, costs=None, logits=None, R1=None):
        """Argument block of the fused generation-time readout + emitter (lvsr_readout_step)."""
        d, p, n_ = self.d, self.store.p, self.n
        lm = self.language_model
        return self.lib.make(
            "lvsr_readout_step_args", S=S2, WA=WA2, lds=int(S2.stride(0)), ldwa=int(WA2.stride(0)), n=n, D=self._state_width(), E=d.E,
            P=d.P, V=d.V, act=ACT_KIND[d.act] if d.post_merge else 0,
            Wms=self._merge_states_weight() if d.use_states_for_readout else None, Wmw=p[n_["Wmw"]],
            bias1=p[n_["bpm"]] if d.post_merge else p[n_["bro"]], Wout=p[n_["Wout"]] if d.post_merge else None,
            bout=p[n_["bout"]] if d.post_merge else None, lm_add=lm_add,
            am_beta=lm.am_beta if lm_add is not None else 1.0, lm_weight=lm.lm_weight if lm_add is not None else 0.0,
            norm_am=int(lm.norm[0]) if lm_add is not None else 1, norm_lm=int(lm.norm[1]) if lm_add is not None else 0,
            norm_tot=int(lm.norm[2]) if lm_add is not None else 0, neglogp=neglogp, logits=logits, uniforms=uniforms,
            R1=R1, ldr1=0 if R1 is None else int(R1.stride(0)),
            outputs=outputs, costs=costs, n_hidden=len(self.pm_hidden), Wh=[p[w] for w, _, _ in self.pm_hidden],
            bh=[p[b] for _, b, _ in self.pm_hidden], dimh=[w for _, _, w in self.pm_hidden])

    def beam_costs(self):
        """Pass A of a position: glimpses of the live hypotheses -> readout -> step costs `neglogp` (K,V)
        (logprobs_computer, search.py:126-134; with a language model ShallowFusionReadout + LMEmitter)."""
        d, lib, st = self.d, self.lib, self._beam
        K, A_ = st["K"], st["A"]
        lib.call("lvsr_attdec_fwd", lib.stream_for(A_["S"]), ctypes.byref(st["argsA"]), 0)
        if "merge" in st:
            lib.call("lvsr_readout_merge", lib.stream_for(st["neglogp"]), *st["merge"])
        lib.call("lvsr_readout_step", lib.stream_for(st["neglogp"]), ctypes.byref(st["readout"]))

    def beam_select(self):
        """Stopping rules, the beam_size best continuations, finished hypotheses, back-pointers; gathers the rows of pass B."""
        st = self._beam
        self.lib.call("lvsr_beam_select", self.lib.stream_for(st["ctl"]), ctypes.byref(st["args"]))

    def beam_advance(self):
        """Pass B: glimpses AGAIN on the re-arranged hypotheses (the window of the location prior depends on the batch it is
        computed for) + compute_states with the chosen characters (next_state_computer, search.py:112-124), language-model
        transition, then the surviving rows become the new beam."""
        d, lib, st = self.d, self.lib, self._beam
        K, B_ = st["rows"], st["B"]
        # (The language-model walk needs the chosen characters and the gathered state sets, nothing of pass B — and pass B nothing of it.
        # Run on a second stream beside pass B — fork / join through events, two branches of the step's graph — it was measured in round 6:
        # 0.94 -> 1.12 ms per utterance at 64 x 4, 0.92 -> 1.18 at 64 x 8.  A fork and a join per position cost more than the 18 us of
        # pointer chasing they take off the chain; serial.)
        if "stepB" in st:
            self._beam_step_run(st)
        else:
            if d.embed:          # lookup feedback needs the fork's GEMMs; one-hot feedback was gathered by the select launch
                self._feedback_fork(st["chars"], K, B_["xg"], st["fb"])
            lib.call("lvsr_attdec_fwd", lib.stream_for(B_["S"]), ctypes.byref(st["argsB"]), 0)
        if st["on_dev_lm"]:
            self._beam_lm_step(st, K)
        lib.call("lvsr_beam_compact", lib.stream_for(st["ctl"]), ctypes.byref(st["args"]))

    def _beam_lm_step(self, st, K):
        """Language-model transition on the chosen characters + look-ahead costs of the new state sets (lvsr_fst_lm_step*)."""
        lib, L, lm = self.lib, st["lm"], self.language_model
        if st["groups"] > 1:       # the rows of finished searches are skipped (their characters are stale)
            lib.call("lvsr_fst_lm_step_groups", lib.stream_for(L["states_sel"]), ctypes.byref(lm._fst), lib_ptr(L["states_sel"]),
                     lib_ptr(L["weights_sel"]), lib_ptr(st["chars"]), K, lib_ptr(L["states_new"]), lib_ptr(L["weights_new"]),
                     lib_ptr(L["add_new"]), lib_ptr(lm._err), lib_ptr(st["ctl"]), st["K"])
        else:
            lib.call("lvsr_fst_lm_step", lib.stream_for(L["states_sel"]), ctypes.byref(lm._fst), lib_ptr(L["states_sel"]),
                     lib_ptr(L["weights_sel"]), lib_ptr(st["chars"]), K, lib_ptr(L["states_new"]), lib_ptr(L["weights_new"]),
                     lib_ptr(L["add_new"]), lib_ptr(lm._err))

    def beam_step(self):
        """All launches of one position as one replayed hipGraph (captured on the second step of a search shape)."""
        st = self._beam

        def enqueue():
            self.beam_costs()
            self.beam_select()
            self.beam_advance()
        self.lib.region(self, st["key"], st["ctl"], enabled=self.use_graph, volatile=st["volatile"], drain=False).run(enqueue)

    def beam_steps(self, n):
        """n consecutive positions as ONE replayed hipGraph (the position counter lives on the device, so the graph does not
        depend on where the search stands): the driver looks at the control block every n positions anyway."""
        st = self._beam
        if n == 1:
            return self.beam_step()

        def enqueue():
            for _ in range(n):
                self.beam_costs()
                self.beam_select()
                self.beam_advance()
        self.lib.region(self, (st["key"], "x%d" % n), st["ctl"], enabled=self.use_graph, volatile=st["volatile"], drain=False).run(enqueue)

    return dict(beam_begin=beam_begin, _beam_begin=_beam_begin, beam_costs=beam_costs, beam_select=beam_select, beam_advance=beam_advance,
                _beam_lm_step=_beam_lm_step, beam_step=beam_step, beam_steps=beam_steps, _readout_step_args=_readout_step_args)


for _k, _v in _beam_methods().items():
    setattr(SequenceGenerator, _k, _v)
