"""Beam search driver over the device-resident beam step (csrc/beam.hip, `SequenceGenerator.beam_*`).

The reference's `BeamSearch.search` (libs/blocks/blocks/search.py:244-407, as modified by lvsr) is a host loop around two
compiled functions; here the whole position — step costs of every continuation, choice of the `beam_size` best, finished
hypotheses, stopping rules, next states, language-model transition — is device work replayed as one hipGraph, and the
host's part shrinks to: start the search, replay steps, look at the `done` word every few steps, and at the end follow the
back-pointers of the finished hypotheses.  The RESULT (hypotheses, costs, the CandidateNotFoundError contract) is what has
to agree with the reference, bit for bit on its golden fixtures; the rules that produce it are listed in csrc/beam.hip.

Two things need the host inside the loop and switch the driver to one synchronisation per position: a
`validate_solution_function` (a Python callback that may veto a finished hypothesis, search.py:372-374) and a language
model whose walk runs on the host (`FSTLanguageModel` without the device tables).
"""
import os

import numpy
import torch

from .bricks.generator import CTL

POLL_EVERY = 8          # positions between two looks at the `done` word (a look is the only host<->device round trip)


class CandidateNotFoundError(Exception):
    """search.py:15-16"""
    pass


class BeamSearch(object):
    def __init__(self, beam_size, recognizer):
        self.beam_size = beam_size
        self.rec = recognizer
        self.last_stats = {}

    # ---- the reference's `_smallest` on the device (kept for callers / tests of the selection rule) -----------------
    def _smallest(self, matrix, k):
        """Indices (row, column) and values of the k smallest entries of a 2-D array, ascending, equal values in flat-index
        order (lvsr_topk_smallest)."""
        from .native import ptr
        lib, dev = self.rec.lib, self.rec.device
        m = torch.as_tensor(numpy.ascontiguousarray(matrix, dtype=numpy.float32)).to(dev).contiguous()
        n, k = int(m.numel()), int(min(k, m.numel()))
        idx = torch.empty(k, dtype=torch.int64, device=dev)
        val = torch.empty(k, dtype=torch.float32, device=dev)
        lib.call("lvsr_topk_smallest", lib.stream_for(m), ptr(m), n, k, ptr(idx), ptr(val))
        return numpy.unravel_index(idx.cpu().numpy(), matrix.shape), val.cpu().numpy()

    # ---- driver ---------------------------------------------------------------------------------------------------
    def search(self, input_values, eol_symbol, max_length, ignore_first_eol=False, as_arrays=False, char_discount=0,
               round_to_inf=1e9, stop_on="patience", validate_solution_function=None):
        """`input_values` = {'recordings': (T,F) ndarray}.  Returns (outputs, costs) lists, best first, or the padded
        (outputs, mask, step costs) arrays with `as_arrays`."""
        if stop_on not in ("patience", "optimistic_future_cost"):
            raise ValueError("Unknown stopping criterion {}".format(stop_on))
        rec, gen = self.rec, self.rec.generator
        lm = gen.language_model
        host_lm = lm is not None and not getattr(lm, "on_device", False)
        stepping = host_lm or validate_solution_function is not None
        max_length = int(max_length)
        first_token = 0 if lm is not None else gen.d.V            # LMEmitter / SoftmaxEmitter initial outputs
        if max_length <= 0:
            raise CandidateNotFoundError()
        if stepping:
            with rec._on_stream():
                rec.compute_contexts(input_values["recordings"])
                st = gen.beam_begin(self.beam_size, eol_symbol, max_length, ignore_first_eol, char_discount, round_to_inf, stop_on)
                ctl = self._search_stepping(st, input_values, lm if host_lm else None, validate_solution_function, first_token)
            return self._finish(st, ctl, first_token, char_discount, as_arrays)
        run = self.begin(input_values, eol_symbol, max_length, ignore_first_eol=ignore_first_eol, char_discount=char_discount,
                         round_to_inf=round_to_inf, stop_on=stop_on)
        while not run["done"]:
            self.advance(run, POLL_EVERY, wait=True)
        return self.finish(run, as_arrays=as_arrays)

    # ---- the same search in pieces that never block the host: several searches (one recognizer + stream each) can be kept in
    #      flight from one thread (tools/bench_decode.py: decoding is "replicas only", also within a GPU) -----------------------
    def begin(self, input_values, eol_symbol, max_length, ignore_first_eol=False, char_discount=0, round_to_inf=1e9,
              stop_on="patience", force_merge=False):
        """Enqueue the encoder pass and the reset of the beam state; returns the handle of the running search."""
        rec, gen = self.rec, self.rec.generator
        if stop_on not in ("patience", "optimistic_future_cost"):
            raise ValueError("Unknown stopping criterion {}".format(stop_on))
        lm = gen.language_model
        assert lm is None or getattr(lm, "on_device", False), "the free-running search needs the device language model"
        with rec._on_stream():
            rec.compute_contexts(input_values["recordings"])
            st = gen.beam_begin(self.beam_size, eol_symbol, int(max_length), ignore_first_eol, char_discount, round_to_inf, stop_on,
                                force_merge=force_merge)
        host = None
        if rec.device.type == "cuda":
            host = torch.empty(16, dtype=torch.int32).pin_memory()
        return dict(st=st, positions=0, max_length=int(max_length), done=False, ctl=None, host=host, event=None,
                    first_token=0 if lm is not None else gen.d.V, char_discount=char_discount)

    # ---- several utterances in one set of launches ---------------------------------------------------------------------
    def search_batch(self, recordings, eol_symbol, max_lengths, ignore_first_eol=False, char_discount=0, round_to_inf=1e9,
                     stop_on="patience", as_arrays=False, validate_solution_function=None):
        """The searches of N utterances side by side: one encoder pass over the padded batch, then every launch of a position
        serves the N beams at once (rows [g K, g K + K) of the state buffers belong to utterance g; windows, position counters,
        stopping rules and finished lists stay per utterance: lvsr_attdec_args.group_rows, lvsr_beam_args.groups).  A beam step
        is dispatch bound for one utterance (13 launches of 5-14 us per position); here the launches are shared.
        -> list of N results, each what `search` returns for that utterance alone — or the exception it would raise.
        A language model walked on the host or a `validate_solution_function` put the host inside every position (module
        docstring): there is nothing to share then, and the utterances are searched one after the other."""
        lm = self.rec.generator.language_model
        if validate_solution_function is not None or (lm is not None and not getattr(lm, "on_device", False)):
            out = []
            for x, limit in zip(recordings, max_lengths):
                try:
                    out.append(self.search({"recordings": numpy.asarray(x, numpy.float32)[:, None, :]}, eol_symbol, limit,
                                           ignore_first_eol=ignore_first_eol, as_arrays=as_arrays, char_discount=char_discount,
                                           round_to_inf=round_to_inf, stop_on=stop_on, validate_solution_function=validate_solution_function))
                except (CandidateNotFoundError, AssertionError, RuntimeError, UnboundLocalError) as e:
                    out.append(e)
            return out
        run = self.begin_batch(recordings, eol_symbol, max_lengths, ignore_first_eol=ignore_first_eol, char_discount=char_discount,
                               round_to_inf=round_to_inf, stop_on=stop_on)
        while not run["done"]:
            self.advance(run, POLL_EVERY, wait=True)
        return self.finish_batch(run, as_arrays=as_arrays)

    def begin_batch(self, recordings, eol_symbol, max_lengths, ignore_first_eol=False, char_discount=0, round_to_inf=1e9,
                    stop_on="patience"):
        rec, gen = self.rec, self.rec.generator
        if stop_on not in ("patience", "optimistic_future_cost"):
            raise ValueError("Unknown stopping criterion {}".format(stop_on))
        if len(recordings) == 1:
            run = self.begin({"recordings": numpy.asarray(recordings[0], numpy.float32)}, eol_symbol, max_lengths[0],
                             ignore_first_eol=ignore_first_eol, char_discount=char_discount, round_to_inf=round_to_inf, stop_on=stop_on,
                             force_merge=True)       # the kernels of the batched search: the same hypotheses in any batch
            run["single"] = True
            return run
        lm = gen.language_model
        assert lm is None or getattr(lm, "on_device", False), "the batched search needs the device language model"
        limits = [int(m) for m in max_lengths]
        with rec._on_stream():
            rec.compute_contexts_batch(recordings)
            st = gen.beam_begin(self.beam_size, eol_symbol, [max(m, 1) for m in limits], ignore_first_eol, char_discount, round_to_inf, stop_on)
        host = None
        if rec.device.type == "cuda":
            host = torch.empty(tuple(st["ctl"].shape), dtype=torch.int32).pin_memory()
        return dict(st=st, positions=0, max_length=max(max(limits), 1), limits=limits, done=False, ctl=None, host=host, event=None,
                    first_token=0 if lm is not None else gen.d.V, char_discount=char_discount)

    def finish_batch(self, run, as_arrays=False):
        """-> one entry per utterance: the result, or the exception the single search raises in its place."""
        if run.get("single"):
            try:
                return [self.finish(run, as_arrays=as_arrays)]
            except (CandidateNotFoundError, AssertionError, RuntimeError, UnboundLocalError) as e:
                return [e]
        rec, gen, st = self.rec, self.rec.generator, run["st"]
        assert run["done"]
        lm = gen.language_model
        out, stats = [], []
        with rec._on_stream():
            if lm is not None and getattr(lm, "on_device", False):
                lm.check_error()
            rec.encoder.check_persistent()
            ctl = st["ctl"].cpu().numpy()
            host = {k: st[k].cpu().numpy() for k in ("fin_pos", "fin_col", "hist_parent", "hist_char", "hist_cost")}
            for g in range(st["groups"]):
                view = {k: torch.from_numpy(v[g]) for k, v in host.items()}
                try:
                    if run["limits"][g] <= 0:
                        raise CandidateNotFoundError()
                    self._raise_device_errors(ctl[g])
                    out.append(self._collect(view, ctl[g], run["first_token"], run["char_discount"], as_arrays))
                except (CandidateNotFoundError, AssertionError, RuntimeError, UnboundLocalError) as e:
                    out.append(e)
                stats.append(dict(positions=int(ctl[g][CTL["steps"]]), finished=int(ctl[g][CTL["nfin"]]), done=int(ctl[g][CTL["done"]]),
                                  reused=int(ctl[g][10])))
        self.last_stats = dict(positions=max(s["positions"] for s in stats), reused=sum(s["reused"] for s in stats), per_utterance=stats)
        return out

    def advance(self, run, positions=POLL_EVERY, wait=False):
        """Enqueue up to `positions` more positions and a look at the control block.  wait=False returns at once; the look is
        picked up by the next call (or by `ready`)."""
        rec, gen, st = self.rec, self.rec.generator, run["st"]
        if run["done"]:
            return True
        if run["event"] is not None:                       # a look is in flight
            if not wait and not run["event"].query():
                return False
            run["event"].synchronize()
            run["event"] = None
            run["ctl"] = run["host"].numpy().copy()
            if self._all_done(run["ctl"]) or run["positions"] >= run["max_length"]:
                run["done"] = True
                return True
        if positions <= 0:
            return False
        with rec._on_stream():
            n = min(int(positions), run["max_length"] - run["positions"])
            if n == POLL_EVERY:
                gen.beam_steps(n)                          # one graph launch for the whole stretch between two looks
            else:
                for _ in range(n):
                    gen.beam_step()
            run["positions"] += n
            if run["host"] is None:                        # CPU emulator: plain synchronous look
                run["ctl"] = st["ctl"].cpu().numpy()
                run["done"] = self._all_done(run["ctl"]) or run["positions"] >= run["max_length"]
                return run["done"]
            run["host"].copy_(st["ctl"], non_blocking=True)
            run["event"] = torch.cuda.Event()
            run["event"].record(torch.cuda.current_stream(rec.device))
        if wait:
            return self.advance(run, 0, wait=True) if run["event"] is not None else run["done"]
        return False

    @staticmethod
    def _all_done(ctl):
        """ctl: the control block of one search (16 words) or of a batch of searches (N, 16)."""
        return bool(numpy.all(numpy.asarray(ctl).reshape(-1, 16)[:, CTL["done"]] != 0))

    def finish(self, run, as_arrays=False):
        """Collect the result of a search whose `advance` reported done."""
        rec, gen = self.rec, self.rec.generator
        assert run["done"]
        lm = gen.language_model
        with rec._on_stream():
            if lm is not None and getattr(lm, "on_device", False):
                lm.check_error()
            ctl = run["st"]["ctl"].cpu().numpy()
            return self._finish(run["st"], ctl, run["first_token"], run["char_discount"], as_arrays)

    def _finish(self, st, ctl, first_token, char_discount, as_arrays):
        with self.rec._on_stream():
            self.rec.encoder.check_persistent()
            self._raise_device_errors(ctl)
            self.last_stats = dict(positions=int(ctl[CTL["steps"]]), finished=int(ctl[CTL["nfin"]]), done=int(ctl[CTL["done"]]),
                                   reused=int(ctl[10]))        # positions whose second attention pass was the first one's results
            return self._collect(st, ctl, first_token, char_discount, as_arrays)

    @staticmethod
    def _raise_device_errors(ctl):
        err = int(ctl[CTL["err"]])
        if err == 1:
            raise AssertionError("non-finite step costs in beam search")             # search.py:345 `assert numpy.isfinite`
        if err == 2:
            raise RuntimeError("beam search: the finished-hypothesis list overflowed its device buffer")
        if err == 3:       # the reference decrements `patience` before it ever assigns it (first finished score >= 1000)
            raise UnboundLocalError("local variable 'patience' referenced before assignment")

    def _search_stepping(self, st, input_values, host_lm, validate, first_token):
        """One synchronisation per position: the host language-model walk feeds the fusion costs of the live hypotheses,
        and/or a Python callback vetoes finished hypotheses (search.py:372-374)."""
        gen, dev = self.rec.generator, self.rec.device
        K = st["K"]
        lm_states = host_lm.initial_states(1) if host_lm is not None else None
        hist_p, hist_c = [], []
        ctl = st["ctl"].cpu().numpy()
        for position in range(st["max_length"]):
            if host_lm is not None:
                n = int(ctl[CTL["nlive"]])
                add = numpy.zeros((K, gen.d.V), numpy.float32)
                if n:
                    add[:n] = lm_states["add"][:n]
                    add[n:] = add[0]
                st["lm"]["add_live"].copy_(torch.from_numpy(add))
            # where this position's finished hypotheses will land: in patience mode the kernel first sorts the list and cuts it to K
            nfin_before = int(ctl[CTL["nfin"]])
            if st["stop_on"] == "patience" and int(ctl[CTL["nlive"]]) > 0:
                nfin_before = min(nfin_before, K)
            gen.beam_costs()
            gen.beam_select()
            ctl = st["ctl"].cpu().numpy()
            if ctl[CTL["done"]] in (1, 2):
                break                                         # a stopping rule fired: nothing was selected at this position
            nsel, nlive = int(ctl[CTL["nsel"]]), int(ctl[CTL["nlive"]])
            parents = st["parents"].cpu().numpy()[:nsel]
            chars = st["chars"].cpu().numpy()[:nsel]
            keep = st["keep"].cpu().numpy()[:nlive]
            if validate is not None:
                hist_p.append(st["hist_parent"][position].cpu().numpy().copy())
                hist_c.append(st["hist_char"][position].cpu().numpy().copy())
                nfin = int(ctl[CTL["nfin"]])
                if nfin > nfin_before:
                    cols = st["fin_col"].cpu().numpy()
                    ok = [i for i in range(nfin_before, nfin)
                          if validate(input_values, self._tokens_of(hist_p, hist_c, position, int(cols[i]), first_token))]
                    if len(ok) != nfin - nfin_before:
                        for name in ("fin_pos", "fin_col", "fin_cost", "fin_score"):
                            t = st[name]
                            kept = t[torch.as_tensor(ok, dtype=torch.int64, device=dev)].clone() if ok else t[:0]
                            t[nfin_before: nfin_before + len(ok)] = kept
                        st["ctl"][CTL["nfin"]] = nfin_before + len(ok)
                        ctl[CTL["nfin"]] = nfin_before + len(ok)
            if host_lm is not None:
                lm_states = host_lm.take(lm_states, parents)
                lm_states = host_lm.transition(lm_states, chars)
                lm_states = host_lm.take(lm_states, keep)
            gen.beam_advance()
            if ctl[CTL["done"]]:
                break
        return st["ctl"].cpu().numpy()

    @staticmethod
    def _tokens_of(hist_p, hist_c, position, col, first_token):
        toks = []
        for p in range(position, -1, -1):
            toks.append(int(hist_c[p][col]))
            col = int(hist_p[p][col])
        return numpy.array([first_token] + toks[::-1])

    def _collect(self, st, ctl, first_token, char_discount, as_arrays):
        """Follow the back-pointers of the finished hypotheses and rank them (search.py:378-407)."""
        nfin, npos = int(ctl[CTL["nfin"]]), int(ctl[CTL["steps"]])
        if nfin == 0:
            raise CandidateNotFoundError()
        fin_pos = st["fin_pos"][:nfin].cpu().numpy()
        fin_col = st["fin_col"][:nfin].cpu().numpy()
        hp = st["hist_parent"][:npos].cpu().numpy()
        hc = st["hist_char"][:npos].cpu().numpy()
        hcost = st["hist_cost"][:npos].cpu().numpy()
        hyps = []
        for p, col in zip(fin_pos, fin_col):
            toks, run = [], []
            for q in range(int(p), -1, -1):
                toks.append(hc[q, col])
                run.append(hcost[q, col])
                col = hp[q, col]
            tokens = numpy.array([first_token] + toks[::-1], dtype=numpy.int64)
            running = numpy.array([numpy.float32(0)] + run[::-1], dtype=numpy.float32)
            hyps.append((tokens, running))
        # final ranking: cumulative cost minus the discount per row of the cost column; Python's sort is stable, so equal
        # scores keep the order of completion (patience mode: of the truncated, sorted list)
        # (float64, as numpy.float32 - Python float was under the numpy the reference was written for, and as csrc/beam.hip ranks)
        hyps.sort(key=lambda h: float(h[1][-1]) - float(char_discount) * len(h[1]))
        longest = max(len(t) for t, _ in hyps)
        out_tokens = numpy.zeros((longest, len(hyps)))
        out_mask = numpy.zeros((longest, len(hyps)))
        out_running = numpy.zeros((longest, len(hyps)))
        for col, (tokens, running) in enumerate(hyps):
            n = len(tokens)
            out_tokens[:n, col] = tokens
            out_mask[:n, col] = 1
            out_running[:n, col] = running
            out_running[n:, col] = running[-1]
        # drop the initial pseudo-token; step costs = differences of the running costs
        result = out_tokens[1:], out_mask[1:], out_running[1:] - out_running[:-1]
        return result if as_arrays else self.result_to_lists(result)

    @staticmethod
    def result_to_lists(result):
        """(outputs, mask, step costs) arrays -> ([tokens of hypothesis k], [total cost of hypothesis k])."""
        tokens, mask, step_costs = result
        lengths = mask.sum(axis=0).astype(int)
        outputs = [list(tokens[:n, k]) for k, n in enumerate(lengths)]
        return outputs, list(step_costs.sum(axis=0))
