"""`dec_stack > 1`: the sequence generator whose transition is a `RecurrentStack` of GatedRecurrent layers with skip
connections (lvsr/bricks/recognizer.py:250-262; libs/blocks/blocks/bricks/recurrent.py:677-950), as four shipped WSJ configs
ask for (wsj_jan_debug, wsj_jan_wsj13v2 / 14v2 / 15v2).

What changes against the one-layer generator (reference behaviour, verified by running it: oracle/theano_harness):
  - `transition.apply.states` = [states, states#1, ...]: the attention has one state transformer per layer
    (`state_trans/transform_states#l.W`, summed: libs/blocks/blocks/bricks/attention.py:281-283) and the readout's Merge one
    source per layer (`merge/transform_states#l.W`);
  - `transition.apply.sequences` = [inputs, gate_inputs, inputs#1, gate_inputs#1, ...]: the generator's Fork of the feedback
    and the Distribute of the glimpse feed every layer (`fork/fork_inputs#l`, `distribute/fork_inputs#l`);
  - inside a step, layer l > 0 adds a bias-free Fork (`recurrentstack/fork_l`) of the NEW state of layer l - 1
    (recurrent.py:936-944), so the layers of one step run one after the other.

Layout here: the states of all layers side by side, S (L+1, B, n*D) — for the attention block (energies over the concatenated
state with the row-concatenated transformers) and the readout this is the one-layer problem with D' = n*D; every GRU layer runs
the one-layer GRU kernels with its own state slots and, for l > 0, the "glimpse" [weighted_averages | new state of layer l-1]
against the row-concatenated [distribute ; fork_l] weights.  The label loop is driven from here, one label at a time
(`label0` / `parts` of the argument blocks, include/lvsr_hip.h), a handful of launches per label and layer; inside the training
step's graph region the loop costs host time only at capture.  A layer's state slots and running state gradient are
column blocks of the concatenated arrays (`S_ld` / `ds_ld`), worked on in place.  The persistent one-launch kernels cover one
layer only.
Built: cost_matrix / backward (training), analyze, the device beam search, generate / sample.
"""
import ctypes

import torch

from ..spec import decoder_layer_names
from .generator import SequenceGenerator, ATT_MS, NORMALIZER_KIND, lib_ptr

_LAYER_KEYS = ("Ws", "Wdi", "Wdg", "Whh", "Whg", "h0", "Wfi", "bfi", "Wfg", "bfg", "Wms")


class StackedSequenceGenerator(SequenceGenerator):
    def __init__(self, dims, store, lib, workspace, use_graph=True, use_persistent=None):
        super(StackedSequenceGenerator, self).__init__(dims, store, lib, workspace, use_graph=use_graph, use_persistent=False)
        assert dims.n_dec > 1
        # the teacher-forced label loop of a TWO-layer stack as one persistent launch (csrc/decoder_persist.hip, two clusters per
        # utterance): None = when the configuration fits, True / False = forced on / off.  The reverse walk stays on the step kernels.
        self.use_persistent_stack = use_persistent
        self.nl = [decoder_layer_names(dims, l) for l in range(dims.n_dec)]
        for k in _LAYER_KEYS:          # the one-layer names do not exist here: nothing may fall back to them
            self.n.pop(k, None)

    def _E(self, l):
        """Input width of layer l's distribution: the glimpse, for l > 0 followed by the state of the layer below."""
        return self.d.E + (self.d.D if l > 0 else 0)

    # ---- hooks of the base class -----------------------------------------------------------------------------------
    def _state_width(self):
        return self.d.D_tot

    def _cats(self):
        """The concatenated copies of the current parameters (made by `_packed`; not re-made inside a capture that already has)."""
        if self._packs is not None and self._packs["version"] == self.store.version:
            return self._packs
        return self._packed()

    def _merge_states_weight(self):
        return self._cats()["Wms_cat"]

    def _initial_state(self):
        return self._cats()["h0_cat"]

    def _merge_states_backward(self, S2, dR1, gws):
        d, g, lib, ws = self.d, self.store.g, self.lib, self.ws
        for l, n in enumerate(self.nl):
            lib.sgemm(S2[:, l * d.D:(l + 1) * d.D], dR1, g[n["Wms"]], transA=True, ws=gws, group=True)
        dS_r = ws.get("gen.dS_r", (S2.shape[0], d.D_tot))
        lib.sgemm(dR1, self._merge_states_weight(), dS_r, transB=True)
        return dS_r

    def _packed(self, packs=True):
        if self._packs is not None and self._packs["version"] == self.store.version and not self.lib.capturing:
            return self._packs
        p, lib, ws, d = self.store.p, self.lib, self.ws, self.d
        D, E = d.D, d.E
        ent = dict(version=self.store.version)
        jobs, copies = [], []

        def pack(key, W, trans=False):
            K, N = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
            buf = ws.get("gen.%s_p" % key, (lib.pack_size(K, N),))
            jobs.append((W, buf, trans))
            ent[key] = buf
        Ws_cat = ws.get("gen.Ws_cat", (d.D_tot, d.M))
        Wms_cat = ws.get("gen.Wms_cat", (d.D_tot, d.P))
        h0_cat = ws.get("gen.h0_cat", (d.D_tot,))
        cats = []
        for l, n in enumerate(self.nl):
            rows = slice(l * D, (l + 1) * D)
            copies += [(p[n["Ws"]], Ws_cat[rows]), (p[n["h0"]], h0_cat[rows])]
            if d.use_states_for_readout:
                copies.append((p[n["Wms"]], Wms_cat[rows]))
            El = self._E(l)
            wdi, wdg = ws.get("gen.Wdi_cat%d" % l, (El, D)), ws.get("gen.Wdg_cat%d" % l, (El, 2 * D))
            wd = ws.get("gen.Wd_cat%d" % l, (El, 3 * D))
            copies += [(p[n["Wdi"]], wdi[:E]), (p[n["Wdg"]], wdg[:E]), (p[n["Wdi"]], wd[:E, :D]), (p[n["Wdg"]], wd[:E, D:])]
            if l > 0:
                copies += [(p[n["Fi"]], wdi[E:]), (p[n["Fg"]], wdg[E:]), (p[n["Fi"]], wd[E:, :D]), (p[n["Fg"]], wd[E:, D:])]
            cats.append((wdi, wdg, wd))
        lib.copy_many(copies)
        pack("Ws", Ws_cat); pack("WsT", Ws_cat, True)
        for l, n in enumerate(self.nl):
            wdi, wdg, wd = cats[l]
            pack("Whg%d" % l, p[n["Whg"]]); pack("WhgT%d" % l, p[n["Whg"]], True)
            pack("Whh%d" % l, p[n["Whh"]]); pack("WhhT%d" % l, p[n["Whh"]], True)
            pack("Wdi%d" % l, wdi); pack("Wdg%d" % l, wdg); pack("WdT%d" % l, wd, True)
        lib.pack_many(jobs, use_graph=self.use_graph, cache=self._pack_cache)
        ent.update(Wms_cat=Wms_cat, h0_cat=h0_cat)
        self._packs = ent
        return ent

    # ---- argument blocks ---------------------------------------------------------------------------------------------
    def _strides(self, B, broadcast):
        d = self.d
        if broadcast:
            return dict(A_ts=d.E, A_bs=0, PA_ts=d.M, PA_bs=0, Am_ts=1, Am_bs=0)
        return dict(A_ts=B * d.E, A_bs=d.E, PA_ts=B * d.M, PA_bs=d.M, Am_ts=B, Am_bs=1)

    def _att_fields(self, pk, A, PA, Am, L, B, bufs, phases, step0, broadcast, groups=0, group_Tp=None):
        """The attention over the concatenated states: the one-layer block with D = n*D and no GRU part.  groups: batched beam
        search (rows [g B/groups, ...) read utterance g: lvsr_attdec_args.group_rows)."""
        d, p, n = self.d, self.store.p, self.n
        kind, pp = self._prior()
        f = dict(Tp=int(A.shape[0]), B=B, L=L, E=d.E, D=d.D_tot, M=d.M, K=d.K, c=d.c, prior_type=kind, step0=step0,
                 phases=phases, p0=pp[0], p1=pp[1], p2=pp[2], p3=pp[3], A=A, PA=PA, Am=Am, Ws_p=pk["Ws"], w_e=p[n["we"]],
                 normalizer=NORMALIZER_KIND[d.normalizer], e_bias=p[n["eb"]] if d.energy_bias else None,
                 filters=p[n["filters"]] if d.conv else None, handler=p[n["handler"]] if d.conv else None)
        if groups:
            f.update(A_ts=groups * d.E, A_bs=d.E, PA_ts=groups * d.M, PA_bs=d.M, Am_ts=groups, Am_bs=1,
                     group_rows=B // groups, step_stride=16, group_Tp=group_Tp)
        else:
            f.update(self._strides(B, broadcast))
        f.update(bufs)
        return f

    def _attdec_fields(self, pk, A, PA, Am, L, B, bufs, phases, step0, broadcast, groups=0, group_Tp=None):
        assert not phases & 2, "a stacked decoder has no one-block GRU step"
        return self._att_fields(pk, A, PA, Am, L, B, {k: v for k, v in bufs.items() if k not in ("U", "R", "C", "RH", "sg", "xin", "xg")},
                                phases, step0, broadcast, groups=groups, group_Tp=group_Tp)

    def _layer_fields(self, l, pk, A, PA, Am, L, B, bufs, broadcast):
        """GRU layer l: the one-layer block with the GRU part only, E = width of [glimpse | state below], no location prior."""
        d = self.d
        f = dict(Tp=int(A.shape[0]), B=B, L=L, E=self._E(l), D=d.D, M=d.M, K=0, c=0, prior_type=0, step0=0, phases=2,
                 p0=0.0, p1=0.0, p2=0.0, p3=0.0, A=A, PA=PA, Am=Am, normalizer=0,
                 Whg_p=pk["Whg%d" % l], Whh_p=pk["Whh%d" % l], Wdi_p=pk["Wdi%d" % l], Wdg_p=pk["Wdg%d" % l])
        f.update(self._strides(B, broadcast))
        f.update(bufs)
        return f

    def _feedback_forks(self, labels_flat, nrows, xgs, fb_buf=None):
        """xgs[l] (nrows,3D) = fork#l(feedback(labels)) (sequence_generators.py:263-264 with the stack's sequence names)."""
        d, p, lib = self.d, self.store.p, self.lib
        st = lib.stream_for(xgs[0])
        if d.embed:
            lib.call("lvsr_gather_rows", st, lib_ptr(p[self.n["table"]]), d.FB, lib_ptr(labels_flat), nrows, d.V + 1, d.FB,
                     None, lib_ptr(fb_buf), d.FB)
        for n, xg in zip(self.nl, xgs):
            if d.embed:
                lib.sgemm(fb_buf, p[n["Wfi"]], xg[:, : d.D], bias=p[n["bfi"]])
                lib.sgemm(fb_buf, p[n["Wfg"]], xg[:, d.D:], bias=p[n["bfg"]])
            else:
                lib.call("lvsr_gather_rows", st, lib_ptr(p[n["Wfi"]]), d.D, lib_ptr(labels_flat), nrows, d.FB, d.D,
                         lib_ptr(p[n["bfi"]]), lib_ptr(xg), 3 * d.D)
                lib.call("lvsr_gather_rows", st, lib_ptr(p[n["Wfg"]]), 2 * d.D, lib_ptr(labels_flat), nrows, d.FB, 2 * d.D,
                         lib_ptr(p[n["bfg"]]), lib_ptr(xg[:, d.D:]), 3 * d.D)

    def _step_blocks(self, pk, A, PA, Am, L, B, tag, att_bufs, ym, att_phases, step0, broadcast, step_dev=None, groups=0, group_Tp=None):
        """Buffers and argument blocks of `L` label steps: att_bufs = the attention block's slots (S (L+1,B,n*D), W, pos, WA,
        EN, ...); per layer its own state slots, gate tensors and distribution input."""
        d, lib, ws = self.d, self.lib, self.ws
        D = d.D
        layers = []
        for l in range(d.n_dec):
            t = "%s.l%d" % (tag, l)
            WA_l = att_bufs["WA"] if l == 0 else ws.get(t + ".WA", (L, B, self._E(l)))
            # the layer's states are a column block of the attention block's (L+1, B, n*D) slots: S_ld = n*D in its argument block
            bufs = dict(xg=ws.get(t + ".xg", (L * B, 3 * D)), ymask=ym, S=att_bufs["S"][:, :, l * D:(l + 1) * D], WA=WA_l,
                        U=ws.get(t + ".U", (L, B, D)), R=ws.get(t + ".R", (L, B, D)), C=ws.get(t + ".C", (L, B, D)),
                        RH=ws.get(t + ".RH", (L, B, D)), sg=ws.get(t + ".sg", (B, 2 * D)), xin=ws.get(t + ".xin", (B, D)))
            fields = self._layer_fields(l, pk, A, PA, Am, L, B, bufs, broadcast)
            fields["S_ld"] = d.D_tot
            layers.append(dict(bufs=bufs, fields=fields, args=lib.make("lvsr_attdec_args", **fields)))
        fields = self._att_fields(pk, A, PA, Am, L, B, att_bufs, att_phases, step0, broadcast, groups=groups, group_Tp=group_Tp)
        extra = {} if step_dev is None else dict(step_dev=step_dev)
        return dict(L=L, att=dict(bufs=att_bufs, fields=fields, args=lib.make("lvsr_attdec_args", **dict(fields, **extra))),
                    layers=layers)

    def _run_step(self, blk, i, stream):
        """Label step i of a block set: glimpses from the concatenated state slot i, then the layers bottom-up, each writing its
        column block of slot i + 1."""
        self._run_attention(blk, i, stream)
        self._run_layers(blk, i, stream)

    def _run_attention(self, blk, i, stream):
        att = blk["att"]
        att["args"].label0, att["args"].L = i, i + 1
        self.lib.call("lvsr_attdec_fwd", stream, ctypes.byref(att["args"]), 0)

    def _run_layers(self, blk, i, stream):
        d, lib = self.d, self.lib
        E = d.E
        att, layers = blk["att"], blk["layers"]
        for l, lay in enumerate(layers):
            if l > 0:
                wa_l = lay["bufs"]["WA"][i]
                lib.copy_many([(att["bufs"]["WA"][i], wa_l[:, :E]), (layers[l - 1]["bufs"]["S"][i + 1], wa_l[:, E:])])
            lay["args"].label0, lay["args"].L = i, i + 1
            lib.call("lvsr_attdec_fwd", stream, ctypes.byref(lay["args"]), 0)

    # ---- teacher-forced pass ---------------------------------------------------------------------------------------------
    def _forward_recurrent(self, pk, A, PA, Am, labels, ym, L, B, Tp):
        d, lib, ws = self.d, self.lib, self.ws
        Kc = max(d.K, 1)
        S = ws.get("gen.S", (L + 1, B, d.D_tot))
        W = ws.get("gen.W", (L + 1, B, Tp))
        att_bufs = dict(S=S, W=W, pos=ws.get("gen.pos", (L + 1, B)) if (d.conv and self._prior()[0] != 0) else None,
                        WA=ws.get("gen.WA", (L, B, d.E)), EN=ws.get("gen.EN", (L, B, Tp)), ZB=ws.get("gen.ZB", (L, B)),
                        sW=ws.get("gen.sW", (L, B, d.M)), CV=ws.get("gen.CV", (L, B, Kc, Tp)) if d.conv else None,
                        ep=ws.get("gen.ep", (B, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
        blk = self._step_blocks(pk, A, PA, Am, L, B, "gen", att_bufs, ym, att_phases=1, step0=0, broadcast=False)
        fb = ws.get("gen.fb", (L * B, d.FB)) if d.embed else None
        self._feedback_forks(labels.view(-1), L * B, [lay["bufs"]["xg"] for lay in blk["layers"]], fb)
        S[0].copy_(self._initial_state().unsqueeze(0).expand(B, d.D_tot))       # initial_state of every layer, tiled
        W[0].zero_()
        if d.conv:
            W[0, :, 0] = 1.0
        stream = lib.stream_for(S)
        st2 = self._forward_persistent_stack2(blk, att_bufs, A, PA, Am, L, B, Tp, stream)
        if st2 is None:
            for i in range(L):
                self._run_step(blk, i, stream)
        return dict(bufs=att_bufs, saved=dict(blk=blk, fb=fb, stack2=st2))

    def _forward_persistent_stack2(self, blk, att_bufs, A, PA, Am, L, B, Tp, stream):
        """The label loop of a two-layer stack as ONE persistent launch (lvsr_attdec_fwd_persistent_stack2: per utterance a cluster
        for the attention + layer 0 and one for layer 1), writing everything the step-kernel reverse walk reads: state slots of
        both layers, alignments, energies, transformed states, convolution features, the gates of both layers; the glimpses and
        layer 1's distribution input [glimpse | new state of layer 0] are filled in behind it.  -> None when it does not apply,
        else what the persistent reverse walk re-uses (argument fields, AW0 / AW1, the concatenated distribution weights)."""
        d, p, lib, ws = self.d, self.store.p, self.lib, self.ws
        if d.n_dec != 2 or self.use_persistent_stack is False or (lib.is_emulator and not lib.emulates_concurrency()):
            if self.use_persistent_stack:
                raise ValueError("persistent stacked decoder requested but not available (two layers, concurrent work-groups)")
            return None
        D, E = d.D, d.E
        l0, l1 = blk["layers"]
        n0, n1 = self.nl
        kind, pp = self._prior()
        f = dict(Tp=Tp, B=B, L=L, E=E, D=D, M=d.M, K=d.K, c=d.c, prior_type=kind, step0=0, phases=3, p0=pp[0], p1=pp[1], p2=pp[2],
                 p3=pp[3], A=A, PA=PA, Am=Am, w_e=p[self.n["we"]], normalizer=NORMALIZER_KIND[d.normalizer],
                 e_bias=p[self.n["eb"]] if d.energy_bias else None, filters=p[self.n["filters"]] if d.conv else None,
                 handler=p[self.n["handler"]] if d.conv else None, S_ld=d.D_tot)
        f.update(self._strides(B, False))
        f.update({k: v for k, v in att_bufs.items() if k != "ep"})
        f.update(xg=l0["bufs"]["xg"], ymask=l0["bufs"]["ymask"], U=l0["bufs"]["U"], R=l0["bufs"]["R"], C=l0["bufs"]["C"], RH=l0["bufs"]["RH"])
        args = lib.make("lvsr_attdec_args", **f)
        nbytes = int(lib._lvsr_attdec_stack2_persist_ws_bytes(ctypes.byref(args)))
        if nbytes == 0:
            if self.use_persistent_stack:
                raise ValueError("persistent stacked decoder requested but the configuration is outside the kernel's limits")
            return None
        sync = ws.get("gen.sync", ((nbytes + 3) // 4,), torch.int32)
        cats = self._cats()
        wd0, wd1 = ws.get("gen.Wd_cat0", (E, 3 * D)), ws.get("gen.Wd_cat1", (E + D, 3 * D))
        AW0, AW1 = ws.get("gen.AW0", (Tp * B, 3 * D)), ws.get("gen.AW1", (Tp * B, 3 * D))
        A2 = A.view(Tp * B, E)
        lib.sgemm(A2, wd0, AW0)
        lib.sgemm(A2, wd1[:E], AW1)
        plain = lib.make("lvsr_attdec_plain", Ws=p[n0["Ws"]], Whg=p[n0["Whg"]], Whh=p[n0["Whh"]], AW=AW0, AW_ld=0)
        b1 = l1["bufs"]
        st2 = lib.make("lvsr_attdec_stack2", Whg1=p[n1["Whg"]], Whh1=p[n1["Whh"]], Ws1=p[n1["Ws"]], F1=wd1[E:], AW1=AW1, xg1=b1["xg"],
                       U1=b1["U"], R1=b1["R"], C1=b1["C"], RH1=b1["RH"], F1_ld=0, AW1_ld=0)
        lib.call("lvsr_attdec_fwd_persistent_stack2", stream, ctypes.byref(args), ctypes.byref(plain), ctypes.byref(st2), lib_ptr(sync), 0)
        lib.call("lvsr_attdec_glimpses", stream, ctypes.byref(args))
        WA1 = b1["WA"].view(L * B, E + D)
        lib.copy_many([(att_bufs["WA"].view(L * B, E), WA1[:, :E]), (att_bufs["S"][1:].reshape(L * B, d.D_tot)[:, :D], WA1[:, E:])])
        return dict(fields=f, AW0=AW0, AW1=AW1, wd0=wd0, wd1=wd1)

    def _backward_recurrent(self, sv, dWA_r, dS_r, gws):
        d, p, g, lib, ws = self.d, self.store.p, self.store.g, self.lib, self.ws
        L, B, Tp, pk, blk = sv["L"], sv["B"], sv["Tp"], sv["pk"], sv["blk"]
        D, E, DT, nrows = d.D, d.E, d.D_tot, sv["L"] * sv["B"]
        att, layers = blk["att"], blk["layers"]
        nslice, ntile, Kc = (d.M + ATT_MS - 1) // ATT_MS, (Tp + 63) // 64, max(d.K, 1)
        DWA = ws.get("gen.DWA", (L, B, E))
        DSW = ws.get("gen.DSW", (nrows, d.M))
        DCV = ws.get("gen.DCV", (L, B, Kc, Tp)) if d.conv else None
        dPA = ws.get("gen.dPA", (Tp, B, d.M), zero=True)
        accH = ws.get("gen.accH", (B * ntile, Kc * d.M), zero=True)
        accWe = ws.get("gen.accWe", (B * ntile, d.M), zero=True)
        accEb = ws.get("gen.accEb", (B * ntile, 1), zero=True)
        ds = ws.get("gen.ds", (B, DT), zero=True)
        dsacc = ws.get("gen.dsacc", (B, DT))
        dalp = ws.get("gen.dalp", (B, Kc, Tp), zero=True)
        bw_att = lib.make("lvsr_attdec_bwd_args", WsT_p=pk["WsT"], dS_r=dS_r, DWA=DWA, DSW=DSW, DCV=DCV, dPA=dPA, accH=accH,
                          accWe=accWe, accEb=accEb, ds=ds, dalp=dalp, dsacc=dsacc, Q=ws.get("gen.Q", (B, Tp)),
                          dcvp=ws.get("gen.dcvp", (B, nslice, Kc, Tp)) if d.conv else None,
                          dswp=ws.get("gen.dswp", (B, ntile, d.M)), parts=2)
        bw_att.f = lib.make("lvsr_attdec_args", **att["fields"])
        bws = []
        for l, lay in enumerate(layers):
            t = "gen.l%d" % l
            # running state gradient and its recurrent part: column blocks of the attention block's (B, n*D) arrays (ds_ld = n*D)
            ent = dict(DXG=ws.get(t + ".DXG", (nrows, 3 * D)), DWA=DWA if l == 0 else ws.get(t + ".DWA", (L, B, self._E(l))),
                       ds=ds[:, l * D:(l + 1) * D], dsacc=dsacc[:, l * D:(l + 1) * D])
            a = lib.make("lvsr_attdec_bwd_args", WhhT_p=pk["WhhT%d" % l], WhgT_p=pk["WhgT%d" % l], WdT_p=pk["WdT%d" % l],
                         dWA_r=dWA_r if l == 0 else None, DXG=ent["DXG"], DWA=ent["DWA"], ds=ent["ds"],
                         dspart=ws.get(t + ".dspart", (B, D)), dsacc=ent["dsacc"], parts=1, ds_ld=DT)
            a.f = lib.make("lvsr_attdec_args", **lay["fields"])
            ent["args"] = a
            bws.append(ent)
        stream = lib.stream_for(ds)
        st2 = sv.get("stack2")
        psync = None
        if st2 is not None and self.use_persistent_stack is not False:
            fargs = lib.make("lvsr_attdec_args", **st2["fields"])
            nbytes = int(lib._lvsr_attdec_stack2_bwd_persist_ws_bytes(ctypes.byref(fargs)))
            if nbytes > 0:
                psync = ws.get("gen.sync_bwd", ((nbytes + 3) // 4,), torch.int32)
        if psync is not None:
            # the whole reverse walk of both layers as one persistent launch (csrc/decoder_persist_bwd.hip, two clusters per utterance)
            n0, n1 = self.nl
            P = int(lib._lvsr_attdec_stack2_bwd_persist_clusters(ctypes.byref(fargs)))
            accH = ws.get("gen.accH_p", (B * P, Kc * d.M))
            accWe = ws.get("gen.accWe_p", (B * P, d.M))
            accEb = ws.get("gen.accEb_p", (B * P, 1))
            QR = ws.get("gen.QR", (L, B, Tp))
            lib.call("lvsr_sgemm_batched", stream, 0, 1, L, Tp, E, 1.0, lib_ptr(dWA_r), B * E, E,
                     lib_ptr(sv["A"]), B * E, E, 0.0, lib_ptr(QR), B * Tp, Tp, B)
            bw = lib.make("lvsr_attdec_bwd_args", dS_r=dS_r, DXG=bws[0]["DXG"], DSW=DSW, DCV=DCV, dPA=dPA, accH=accH, accWe=accWe,
                          accEb=accEb, ds=ds, AW=st2["AW0"], QR=QR, AW_ld=0, ds_ld=DT)
            bw.f = fargs
            plain = lib.make("lvsr_attdec_plain", Ws=p[n0["Ws"]], Whg=p[n0["Whg"]], Whh=p[n0["Whh"]], AW=st2["AW0"], AW_ld=0)
            b1 = layers[1]["bufs"]
            s2 = lib.make("lvsr_attdec_stack2", Whg1=p[n1["Whg"]], Whh1=p[n1["Whh"]], Ws1=p[n1["Ws"]], F1=st2["wd1"][E:], AW1=st2["AW1"],
                          xg1=b1["xg"], U1=b1["U"], R1=b1["R"], C1=b1["C"], RH1=b1["RH"], F1_ld=0, AW1_ld=0, DXG1=bws[1]["DXG"])
            lib.call("lvsr_attdec_bwd_persistent_stack2", stream, ctypes.byref(bw), ctypes.byref(plain), ctypes.byref(s2), lib_ptr(psync))
            # total gradient wrt the glimpses (the kernel does not form it): the readout's share + both layers' distribution inputs
            DWA2 = DWA.view(nrows, E)
            lib.copy_many([(dWA_r.view(nrows, E), DWA2)])
            lib.sgemm(bws[0]["DXG"], st2["wd0"], DWA2, transB=True, beta=1.0)
            lib.sgemm(bws[1]["DXG"], st2["wd1"][:E], DWA2, transB=True, beta=1.0)
        for i in (range(L - 1, -1, -1) if psync is None else ()):
            # bws[l]["ds"] = gradient wrt the state layer l wrote at this label (slot i + 1), from everything later
            for l in range(d.n_dec - 1, -1, -1):
                a = bws[l]["args"]
                a.f.label0, a.f.L = i, i + 1
                lib.call("lvsr_attdec_bwd", stream, ctypes.byref(a), 0)
                if l > 0:       # the fork of the layer below: the tail of this layer's distribution-input gradient
                    lib.copy_many([(bws[l]["DWA"][i][:, E:], bws[l - 1]["ds"], 1.0)])
            # total glimpse gradient: layer 0 wrote its share, incl. the readout's, into DWA (one launch per accumulation: the
            # descriptors of a launch run concurrently)
            for l in range(1, d.n_dec):
                lib.copy_many([(bws[l]["DWA"][i][:, :E], DWA[i], 1.0)])
            bw_att.f.label0, bw_att.f.L = i, i + 1
            lib.call("lvsr_attdec_bwd", stream, ctypes.byref(bw_att), 0)
        # ---- weight gradients as batched GEMMs over all labels
        Scat2 = att["bufs"]["S"][:L].view(nrows, DT)
        labels_flat = sv["labels"].view(-1)
        fb = sv["fb"]
        dfb = ws.get("gen.dfb", (nrows, d.FB)) if d.embed else None
        for l, (n, lay, bwl) in enumerate(zip(self.nl, layers, bws)):
            dpc, dg = bwl["DXG"][:, :D], bwl["DXG"][:, D:]
            lb = lay["bufs"]
            lib.sgemm(lb["RH"].view(nrows, D), dpc, g[n["Whh"]], transA=True, ws=gws, group=True)
            lib.sgemm(Scat2[:, l * D:(l + 1) * D], dg, g[n["Whg"]], transA=True, ws=gws, group=True)
            WA2 = lb["WA"].view(nrows, self._E(l))
            lib.sgemm(WA2[:, :E], dpc, g[n["Wdi"]], transA=True, ws=gws, group=True)
            lib.sgemm(WA2[:, :E], dg, g[n["Wdg"]], transA=True, ws=gws, group=True)
            if l > 0:
                lib.sgemm(WA2[:, E:], dpc, g[n["Fi"]], transA=True, ws=gws, group=True)
                lib.sgemm(WA2[:, E:], dg, g[n["Fg"]], transA=True, ws=gws, group=True)
            lib.sgemm(Scat2[:, l * D:(l + 1) * D], DSW, g[n["Ws"]], transA=True, ws=gws, group=True)
            lib.colsum(bwl["ds"], g[n["h0"]], ws=gws)
            lib.colsum(dpc, g[n["bfi"]], ws=gws)
            lib.colsum(dg, g[n["bfg"]], ws=gws)
            if d.embed:
                lib.sgemm(fb, dpc, g[n["Wfi"]], transA=True, ws=gws, group=True)
                lib.sgemm(fb, dg, g[n["Wfg"]], transA=True, ws=gws, group=True)
                lib.sgemm(dpc, p[n["Wfi"]], dfb, transB=True, beta=0.0 if l == 0 else 1.0)
                lib.sgemm(dg, p[n["Wfg"]], dfb, transB=True, beta=1.0)
            else:
                lib.call("lvsr_scatter_add_rows", stream, lib_ptr(dpc), 3 * D, lib_ptr(labels_flat), nrows, d.FB, D,
                         lib_ptr(g[n["Wfi"]]), D, 0.0)
                lib.call("lvsr_scatter_add_rows", stream, lib_ptr(dg), 3 * D, lib_ptr(labels_flat), nrows, d.FB, 2 * D,
                         lib_ptr(g[n["Wfg"]]), 2 * D, 0.0)
        if d.embed:
            lib.call("lvsr_scatter_add_rows", stream, lib_ptr(dfb), d.FB, lib_ptr(labels_flat), nrows, d.V + 1, d.FB,
                     lib_ptr(g[self.n["table"]]), d.FB, 0.0)
        return dict(accH=accH, accWe=accWe, accEb=accEb, DCV=DCV, dPA=dPA, DWA=DWA,
                    fwd_args=lib.make("lvsr_attdec_args", **att["fields"]))

    # ---- device beam search: pass B (csrc/beam.hip, generator.beam_begin / beam_advance) ---------------------------------
    def _beam_step_blocks(self, pk, g, K, B_, skip_pos, pos_word, tag, groups=0, group_Tp=None):
        att_bufs = {k: B_[k] for k in ("S", "W", "pos", "WA", "EN", "ZB", "sW", "CV", "ep")}
        # the select kernel has moved the position counter on: step0 = -1
        return self._step_blocks(pk, g["A"], g["PA"], g["Am"], 1, K, "bs" + tag, att_bufs, None, att_phases=1 | skip_pos, step0=-1,
                                 broadcast=True, step_dev=pos_word, groups=groups, group_Tp=group_Tp)

    def _beam_step_run(self, st):
        lib, blk = self.lib, st["stepB"]
        S = blk["att"]["bufs"]["S"]
        layers = blk["layers"]
        self._feedback_forks(st["chars"], st["rows"], [lay["bufs"]["xg"] for lay in layers], st["fb"])
        self._run_step(blk, 0, lib.stream_for(S))

    # ---- free-running generation -----------------------------------------------------------------------------------------
    def generate(self, n_steps=None, batch_size=None, attended=None, attended_mask=None, uniforms=None, seed=None):
        """BaseSequenceGenerator.generate (sequence_generators.py:328-377) with the stacked transition; `states` comes back with the
        layers side by side, (n, B, dec_stack * D)."""
        d, lib, ws = self.d, self.lib, self.ws
        if self.language_model is not None:
            raise NotImplementedError("generate() with a language model: the reference's LMEmitter.emit returns zeros (not a "
                                      "sampling path); use beam_search, or cost / analyze for teacher-forced costs")
        N = int(n_steps)
        Tp, B = int(attended.shape[0]), int(attended.shape[1])
        assert batch_size is None or int(batch_size) == B
        pk = self._packed()
        A, Am = attended.contiguous(), attended_mask.contiguous()
        PA = self.preprocess(A)
        u = self._uniforms((N, B), uniforms, seed, A.device)
        Kc = max(d.K, 1)
        pos_needed = d.conv and self._prior()[0] != 0
        S = ws.get("sg.S", (N + 1, B, d.D_tot))
        W = ws.get("sg.W", (N + 1, B, Tp))
        att_bufs = dict(S=S, W=W, pos=ws.get("sg.pos", (N + 1, B)) if pos_needed else None, WA=ws.get("sg.WA", (N, B, d.E)),
                        EN=ws.get("sg.EN", (N, B, Tp)), ZB=None, sW=ws.get("sg.sW", (N, B, d.M)),
                        CV=ws.get("sg.CV", (N, B, Kc, Tp)) if d.conv else None,
                        ep=ws.get("sg.ep", (B, (d.M + ATT_MS - 1) // ATT_MS, Tp)))
        blk = self._step_blocks(pk, A, PA, Am, N, B, "sg", att_bufs, None, att_phases=1 | (4 if pos_needed else 0), step0=0,
                                broadcast=False)
        first = self.initial_states(B, attended=A)
        S[0].copy_(first["states"])
        W[0].copy_(first["weights"])
        if pos_needed:
            att_bufs["pos"][0].zero_()
        outputs = ws.get("sg.outputs", (N, B), torch.int64)
        costs = ws.get("sg.costs", (N, B))
        fb = ws.get("sg.fb", (B, d.FB)) if d.embed else None
        st = lib.stream_for(S)
        for t in range(N):
            self._run_attention(blk, t, st)
            ra = self._readout_step_args(S[t], att_bufs["WA"][t], B, uniforms=u[t], outputs=outputs[t], costs=costs[t])
            lib.call("lvsr_readout_step", st, ctypes.byref(ra))
            self._feedback_forks(outputs[t], B, [lay["bufs"]["xg"][t * B:(t + 1) * B] for lay in blk["layers"]], fb)
            self._run_layers(blk, t, st)
        return dict(states=S[1:], outputs=outputs, weighted_averages=att_bufs["WA"], weights=W[1:], energies=att_bufs["EN"],
                    costs=costs)

    # ---- not built for a stack ----------------------------------------------------------------------------------------------
    def generation_initial_states(self, n=1):
        raise NotImplementedError("the step-wise generation helpers are one-layer only; dec_stack > 1 decodes through beam_begin / "
                                  "beam_step")

    generation_logprobs = generation_next_states = _gen_run = generation_initial_states
