"""Host-side mirror of lvsr.bricks: Encoder (lvsr/bricks/__init__.py:54-83) driving the HIP kernels.

`Encoder.apply(input_, mask)` keeps the reference signature and returns `(encoded, encoded_mask)`;
`Encoder.backward(d_encoded)` is the BPTT counterpart the reference gets from theano.grad.
All arithmetic happens in the C-ABI library (lvsr_sgemm / lvsr_bigru_fwd / lvsr_bigru_bwd / lvsr_colsum);
torch is used for buffers and views only.
"""
import contextlib

import torch


class SpeechBottom(object):
    """lvsr.bricks.recognizer.SpeechBottom (recognizer.py:105-157): Identity, or an MLP applied to every frame."""
    ACT = {"identity": 0, "rectifier": 2, "tanh": 3}

    def __init__(self, dims, store, lib, workspace):
        self.d, self.store, self.lib, self.ws = dims, store, lib, workspace
        self._saved = None

    def apply(self, recordings, save_for_backward=True):
        d, p, lib, ws = self.d, self.store.p, self.lib, self.ws
        if not d.bottom_dims:
            return recordings
        T, B = int(recordings.shape[0]), int(recordings.shape[1])
        x = recordings.contiguous().view(T * B, d.F)
        saved = []
        for j, o in enumerate(d.bottom_dims):
            z = ws.get("bottom%d.z" % j, (T * B, o))
            lib.sgemm(x, p["/recognizer/bottom/bottom/linear_%d.W" % j], z, bias=p["/recognizer/bottom/bottom/linear_%d.b" % j])
            a = ws.get("bottom%d.a" % j, (T * B, o))
            lib.call("lvsr_act_fwd", lib.stream_for(a), self.ACT[d.bottom_act], native_ptr(z), o, T * B, o, native_ptr(a), o)
            saved.append((x, z))
            x = a
        if save_for_backward:
            self._saved = saved
        return x.view(T, B, d.bottom_dims[-1])

    def backward(self, d_out):
        d, p, g, lib, ws = self.d, self.store.p, self.store.g, self.lib, self.ws
        if not d.bottom_dims:
            return
        gws = ws.get("gemm_ws", (1 << 22,))
        da = d_out.contiguous().view(-1, d.bottom_dims[-1])
        n = int(da.shape[0])
        for j in reversed(range(len(d.bottom_dims))):
            x, z = self._saved[j]
            o = d.bottom_dims[j]
            dz = ws.get("bottom%d.dz" % j, (n, o))
            lib.call("lvsr_act_bwd", lib.stream_for(dz), self.ACT[d.bottom_act], native_ptr(z), o, native_ptr(da), o, n, o,
                     native_ptr(dz), o)
            W = "/recognizer/bottom/bottom/linear_%d.W" % j
            lib.sgemm(x, dz, g[W], transA=True, ws=gws)
            lib.colsum(dz, g["/recognizer/bottom/bottom/linear_%d.b" % j], ws=gws)
            if j > 0:
                dx = ws.get("bottom%d.dx" % j, (n, int(x.shape[1])))
                lib.sgemm(dz, p[W], dx, transB=True)
                da = dx


def native_ptr(t):
    from ..native import ptr
    return ptr(t)


class Encoder(object):
    # the persistent cluster kernels win while a cluster serves at most this many utterances (measured on MI355X, H = 256,
    # T = 800: 2.0 / 2.7 / 7.5 us per forward step at 1 / 2 / 4 utterances per cluster against 6.3 for the step kernels)
    PERSIST_MAX_ROWS = 2

    def __init__(self, dims, store, lib, workspace, use_graph=True, use_persistent=None):
        """use_persistent: True / False force the persistent cluster kernels (csrc/encoder_persist.hip) on or off (where
        the shape allows them); None (default) = on the GPU whenever a cluster serves at most PERSIST_MAX_ROWS utterances,
        otherwise two step kernels per time step."""
        # needs co-resident work-groups: the GPU, or the emulator with concurrent work-groups switched on (tests)
        can = not lib.is_emulator or lib.emulates_concurrency()
        self.persist_auto = use_persistent is None
        self.use_persistent = can and (not lib.is_emulator if use_persistent is None else bool(use_persistent))
        self.d = dims
        self.store = store
        self.lib = lib
        self.ws = workspace
        self.use_graph = use_graph
        self._saved = None
        self._packs = {}
        self._cats = {}
        self._pack_cache = {}
        # weight-gradient GEMMs on a second stream: measured SLOWER on MI355X (70.0 vs 66.4 ms per WSJ-base step: the
        # concurrent GEMM work-groups delay the latency-bound step kernels more than the overlap saves), so off by default
        # (round 3, persistent cluster kernels, groups flushed per layer on the second stream inside the step graph: 16.36 vs
        # 15.64 ms — the GEMM work-groups sharing the CUs slow the polling clusters by more than the 1.4 ms they hide)
        self.overlap = False        # measured and rejected (comment above); the attribute keeps the second-stream code reachable for probes
        self._side = None
        self._side_pending = False

    def _names(self, i, direction):
        base = "/recognizer/encoder/bidir%d/%s" % (i, direction)
        return dict(Wi=base + "/fork/fork_inputs.W", bi=base + "/fork/fork_inputs.b",
                    Wg=base + "/fork/fork_gate_inputs.W", bg=base + "/fork/fork_gate_inputs.b",
                    Whh=base + "/gatedrecurrent.state_to_state", Whg=base + "/gatedrecurrent.state_to_gates",
                    h0=base + "/gatedrecurrent.initial_state")

    def _packed(self, i):
        """Packed (MFMA operand order) copies of the recurrent weights of layer i.  All layers are refreshed together, in one
        library call, the first time a layer is asked for after the parameters changed (store.version)."""
        if self._packs.get("version") != self.store.version or self.lib.capturing:
            p, lib, ws = self.store.p, self.lib, self.ws
            packs, jobs = dict(version=self.store.version), []
            for li in range(self.d.n_layers):
                H = self.d.Hs[li]
                ent = dict(Whh=[], Whg=[], WhhT=[], WhgT=[])
                for di, direction in enumerate(("forward", "backward")):
                    n = self._names(li, direction)
                    for key, W, trans in (("Whh", p[n["Whh"]], False), ("Whg", p[n["Whg"]], False), ("WhhT", p[n["Whh"]], True)):
                        K, N = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
                        buf = ws.get("enc%d.%d.%s_p" % (li, di, key), (lib.pack_size(K, N),))
                        jobs.append((W, buf, trans))
                        ent[key].append(buf)
                    # state_to_gates^T as two (K=H,N=H) blocks: update rows, then reset rows
                    sz = lib.pack_size(H, H)
                    buf = ws.get("enc%d.%d.WhgT_p" % (li, di), (2 * sz,))
                    Wg = p[n["Whg"]]
                    jobs.append((Wg[:, :H], buf[:sz], True))
                    jobs.append((Wg[:, H:], buf[sz:], True))
                    ent["WhgT"].append(buf)
                packs[li] = ent
            lib.pack_many(jobs, use_graph=self.use_graph, cache=self._pack_cache)
            self._packs = packs
        return self._packs[i]

    def _fork_cat(self, i):
        """The four input-projection matrices of layer i side by side, (I, 6H) = [Wi_f | Wg_f | Wi_b | Wg_b] in the column order
        of `xg`, and their biases (6H): ONE GEMM per layer produces all input projections of both directions (and one its
        input gradient) instead of four.  Refreshed when the parameters changed (copies, no arithmetic)."""
        ent = self._cats.get(i)
        if ent is None or ent["version"] != self.store.version or self.lib.capturing:
            p, ws = self.store.p, self.ws
            H, I = self.d.Hs[i], self.d.layer_input_dim(i)
            W = ws.get("enc%d.Wcat" % i, (I, 6 * H))
            b = ws.get("enc%d.bcat" % i, (6 * H,))
            pairs = []
            for di, direction in enumerate(("forward", "backward")):
                n = self._names(i, direction)
                o = di * 3 * H
                pairs += [(p[n["Wi"]], W[:, o: o + H]), (p[n["Wg"]], W[:, o + H: o + 3 * H]),
                          (p[n["bi"]], b[o: o + H]), (p[n["bg"]], b[o + H: o + 3 * H])]
            self.lib.copy_many(pairs)                      # one launch instead of eight copy kernels
            ent = dict(version=self.store.version, W=W, b=b)
            self._cats[i] = ent
        return ent["W"], ent["b"]

    @contextlib.contextmanager
    def _side_stream(self):
        """Run the enclosed launches on the encoder's second stream (GPU only), ordered after everything already queued
        on the current stream; `join_side_stream()` orders the current stream after them.  Yields the split-K workspace to
        use inside (the two streams must not share one)."""
        if not self.overlap or self.lib.is_emulator or not torch.cuda.is_available():
            yield self.ws.get("gemm_ws", (1 << 22,))
            return
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(cur.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            yield self.ws.get("gemm_ws.side", (1 << 22,))
        self._side_pending = True

    def join_side_stream(self):
        if self._side is not None and self._side_pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_pending = False

    def _sync_ws(self, i, B, H):
        """Scratch of the persistent cluster kernel (granule planes + abort word), or None when the layer runs as
        per-step kernels (emulator, or a cluster that does not fit the chip)."""
        if not self.use_persistent:
            return None
        nbytes = int(self.lib._lvsr_bigru_persist_ws_bytes(int(B), int(H)))
        if nbytes <= 0:
            return None
        if self.persist_auto and int(self.lib._lvsr_bigru_persist_rows(int(B), int(H))) > self.PERSIST_MAX_ROWS:
            return None
        return self.ws.get("enc%d.sync" % i, ((nbytes + 3) // 4,), torch.int32)

    def check_persistent(self):
        """After a synchronisation point: raise if a persistent kernel gave up waiting for its cluster.  The abort word is sticky
        (no launch clears it: a forward cluster that gave up is still reported after the backward pass and after replayed steps, and
        every later launch on the workspace leaves at once); it is cleared here, when the failure has been reported."""
        for k, t in self.ws._bufs.items():
            if k[0].startswith("enc") and k[0].endswith(".sync") and int(t[0]) != 0:
                t[:64].zero_()
                raise RuntimeError("persistent BiGRU kernel aborted (a work-group of the cluster was not scheduled); results since "
                                   "the last check are invalid")

    def apply(self, input_, mask=None, save_for_backward=True):
        """input_ (T,B,F) fp32, mask (T,B) fp32 or None -> encoded (T',B,2H_last), encoded_mask (T',B)."""
        d, p, lib, ws = self.d, self.store.p, self.lib, self.ws
        T, B = int(input_.shape[0]), int(input_.shape[1])
        x = input_.contiguous()
        m = None if mask is None else mask.contiguous()
        saved = []
        gemm_ws = ws.get("gemm_ws", (1 << 22,))
        for i, (H, s) in enumerate(zip(d.Hs, d.subsample)):
            I = d.layer_input_dim(i)
            Ts = (T + s - 1) // s
            xg = ws.get("enc%d.xg" % i, (T, B, 6 * H))
            y = ws.get("enc%d.y" % i, (T, B, 2 * H))
            ysub = y if s == 1 else ws.get("enc%d.ysub" % i, (Ts, B, 2 * H))
            u = ws.get("enc%d.u" % i, (T, B, 2 * H))
            r = ws.get("enc%d.r" % i, (T, B, 2 * H))
            c = ws.get("enc%d.c" % i, (T, B, 2 * H))
            rh = ws.get("enc%d.rh" % i, (T, B, 2 * H))
            x2, xg2 = x.view(T * B, I), xg.view(T * B, 6 * H)
            sync = self._sync_ws(i, B, H)
            pk = self._packed(i) if sync is None else None     # the persistent kernels read the plain weights
            Wcat, bcat = self._fork_cat(i)
            lib.sgemm(x2, Wcat, xg2, bias=bcat)          # all input projections of the layer, both directions
            h0s = [p[self._names(i, direction)["h0"]] for direction in ("forward", "backward")]
            if sync is not None:      # persistent cluster kernel: reads the plain weights and shards them into LDS itself
                nf, nb = self._names(i, "forward"), self._names(i, "backward")
                Whh, Whg = [p[nf["Whh"]], p[nb["Whh"]]], [p[nf["Whg"]], p[nb["Whg"]]]
            else:
                Whh, Whg = [pk["Whh"][0], pk["Whh"][1]], [pk["Whg"][0], pk["Whg"][1]]
            # (a persistent layer is a memset + one launch: no graph to build, and no host wait for a time-loop graph to drain)
            lib.run("lvsr_bigru_fwd", "lvsr_bigru_fwd_args", y, self.use_graph and sync is None, xg=xg, mask=m, Whh_p=Whh, Whg_p=Whg, h0=h0s,
                    y=y, ysub=(ysub if s > 1 else None), u=u, r=r, c=c, rh=rh, sub=s, T=T, B=B, H=H,
                    persistent=int(sync is not None), sync_ws=sync)
            saved.append(dict(x=x, mask=m, T=T, y=y, u=u, r=r, c=c, rh=rh))
            x = ysub
            if m is not None and s > 1:
                mm = ws.get("enc%d.msub" % i, (Ts, B))
                mm.copy_(m[::s])
                m = mm
            T = Ts
        if m is None:
            m = ws.get("enc.ones_mask", (T, B))
            m.fill_(1.0)
        if save_for_backward:
            self._saved = saved
        return x, m

    def backward(self, d_encoded, need_input_grad=False):
        """d_encoded (T',B,2H_last): gradient wrt `encoded`.  Writes the encoder parameter gradients; returns the gradient
        wrt the encoder input when `need_input_grad` (a bottom MLP sits in front), else None."""
        d, p, g, lib, ws = self.d, self.store.p, self.store.g, self.lib, self.ws
        assert self._saved is not None, "apply() must run first"
        gemm_ws = ws.get("gemm_ws", (1 << 22,))
        self._scatter = []
        dy = d_encoded.contiguous()
        B = int(dy.shape[1])
        for i in reversed(range(d.n_layers)):
            H, s, I = d.Hs[i], d.subsample[i], d.layer_input_dim(i)
            sv = self._saved[i]
            T = sv["T"]
            dxg = ws.get("enc%d.dxg" % i, (T, B, 6 * H))
            Bp = (B + 15) // 16 * 16
            dh_ws = ws.get("enc%d.dh" % i, (12 * Bp * H,))
            nf, nb = self._names(i, "forward"), self._names(i, "backward")
            sync = self._sync_ws(i, B, H)
            pk = self._packed(i) if sync is None else None
            if sync is not None:
                WhhT, WhgT = [p[nf["Whh"]], p[nb["Whh"]]], [p[nf["Whg"]], p[nb["Whg"]]]
            else:
                WhhT, WhgT = [pk["WhhT"][0], pk["WhhT"][1]], [pk["WhgT"][0], pk["WhgT"][1]]
            lib.run("lvsr_bigru_bwd", "lvsr_bigru_bwd_args", dxg, self.use_graph and sync is None, mask=sv["mask"], y=sv["y"],
                    u=sv["u"], r=sv["r"], c=sv["c"], WhhT_p=WhhT, WhgT_p=WhgT, h0=[p[nf["h0"]], p[nb["h0"]]], dy=dy,
                    dxg=dxg, dh_ws=dh_ws, dh0=[g[nf["h0"]], g[nb["h0"]]], sub=s, T=T, B=B, H=H,
                    persistent=int(sync is not None), sync_ws=sync)
            x2 = sv["x"].view(T * B, I)
            dxg2 = dxg.view(T * B, 6 * H)
            y2 = sv["y"].view(T * B, 2 * H)
            rh2 = sv["rh"].view(T * B, 2 * H)
            dx = None
            if i > 0 or need_input_grad:
                dx = ws.get("enc%d.dx" % i, (T, B, I))
            # critical path first: the gradient wrt this layer's input is what the next (lower) layer's recurrence waits for
            if dx is not None:      # the concatenated copy the forward pass of this step made (parameters have not moved since)
                lib.sgemm(dxg2, self._cats[i]["W"], dx.view(T * B, I), transB=True)
            # weight gradients are off the critical path; with `self.overlap = True` they go to a second stream and overlap the next
            # layer's recurrence (see __init__ for why this is not the default)
            with self._side_stream() as side_ws:
                for di, direction in enumerate(("forward", "backward")):
                    n = self._names(i, direction)
                    dc = dxg2[:, di * 3 * H: di * 3 * H + H]
                    dg = dxg2[:, di * 3 * H + H: di * 3 * H + 3 * H]
                    hcol = slice(di * H, (di + 1) * H)
                    lib.sgemm(rh2[:, hcol], dc, g[n["Whh"]], transA=True, ws=side_ws, group=True)
                    if T > 1:
                        if di == 0:     # h_{t-1} = y[t-1]
                            lib.sgemm(y2[: (T - 1) * B, hcol], dg[B:], g[n["Whg"]], transA=True, ws=side_ws, group=True)
                        else:           # backward direction: previous state in scan order is y[t+1]
                            lib.sgemm(y2[B:, hcol], dg[: (T - 1) * B], g[n["Whg"]], transA=True, ws=side_ws, group=True)
                        beta = 1.0
                    else:
                        beta = 0.0
                    # the first scan step starts from the (broadcast) initial state: rank-B update with lda = 0
                    first = dg[:B] if di == 0 else dg[(T - 1) * B:]
                    lib.sgemm(p[n["h0"]], first, g[n["Whg"]], transA=True, beta=beta, M=H, K=B, lda=0, group=True)
                # fork gradients of both directions: one (I, 6H) product and one column sum, scattered into the four matrices
                gW = ws.get("enc%d.gWcat" % i, (I, 6 * H))
                gb = ws.get("enc%d.gbcat" % i, (6 * H,))
                # (not a member of the grouped launch: with 48 output tiles it fills the chip alone, and the group's one-size k-chunks
                # would cost it 13 instead of 5 partial copies of its 3 MB output — measured slower)
                lib.sgemm(x2, dxg2, gW, transA=True, ws=side_ws)
                lib.colsum(dxg2, gb, ws=side_ws)
                for di, direction in enumerate(("forward", "backward")):
                    n = self._names(i, direction)
                    o = di * 3 * H
                    self._scatter += [(gW[:, o: o + H], g[n["Wi"]]), (gW[:, o + H: o + 3 * H], g[n["Wg"]]),
                                      (gb[o: o + H], g[n["bi"]]), (gb[o + H: o + 3 * H], g[n["bg"]])]
            dy = dx
        self.join_side_stream()
        if getattr(lib, "_group", None) is None:      # no grouped launch pending: the fork gradients are there
            self.finish_backward()
        return dy

    def finish_backward(self):
        """Scatter the concatenated fork gradients of all layers into the four parameters each (one launch per 32 pieces).
        Call after the grouped weight-gradient products have been flushed (the fork products are members of the group)."""
        self.lib.copy_many(self._scatter)
        self._scatter = []
