"""Beam search over the HIP generation step.

What must agree with the reference's (modified) `blocks.search.BeamSearch` (libs/blocks/blocks/search.py:19-407) is the
RESULT — the hypotheses and their costs, bit for bit — so the selection rules are kept exactly:
  * candidates = cumulative cost + step cost over (live hypothesis, character); the `beam_size` smallest are taken with
    `argpartition` then `argsort` (numpy's tie order is part of parity, search.py:221-242);
  * a hypothesis ending in `<eol>` is finished when its last step cost is below `round_to_inf` (and the optional
    `validate_solution_function` accepts it); finished hypotheses leave the beam, so the beam shrinks;
  * ranking of finished hypotheses: final cumulative cost minus `char_discount` per emitted position;
  * `stop_on='patience'`: stop after 30 consecutive steps without a better best finished hypothesis;
    `stop_on='optimistic_future_cost'`: stop once the `beam_size`-th finished hypothesis (in order of completion — the
    reference does not sort in this mode) beats min(live cumulative cost) - `char_discount` * `max_length`;
  * no finished hypothesis at all -> CandidateNotFoundError.
The two compiled Theano functions the reference calls per step are `SequenceGenerator.generation_logprobs` /
`generation_next_states` on the device; everything else here is host bookkeeping in float32 like the reference's.
"""
import numpy
import torch


class CandidateNotFoundError(Exception):
    """search.py:15-16"""
    pass


class _Finished(object):
    """A completed hypothesis: the token column and the running-cost column of the beam at the step it ended."""
    __slots__ = ("tokens", "running")

    def __init__(self, tokens, running):
        self.tokens, self.running = tokens, running

    def score(self, char_discount):
        return self.running[-1] - char_discount * len(self.running)


class BeamSearch(object):
    PATIENCE = 30

    def __init__(self, beam_size, recognizer):
        self.beam_size = beam_size
        self.rec = recognizer

    @staticmethod
    def _smallest(matrix, k):
        """Indices (row, column) and values of the k smallest entries, ascending; ties as numpy breaks them."""
        flat = matrix.reshape(-1)
        pick = numpy.argpartition(flat, k)[:k] if flat.size > k else numpy.arange(flat.size)
        pick = pick[numpy.argsort(flat[pick])]
        return numpy.unravel_index(pick, matrix.shape), flat[pick]

    def search(self, input_values, eol_symbol, max_length, ignore_first_eol=False, as_arrays=False, char_discount=0,
               round_to_inf=1e9, stop_on="patience", validate_solution_function=None):
        """`input_values` = {'recordings': (T,F) ndarray}.  Returns (outputs, costs) lists, best first, or the padded
        (outputs, mask, step costs) arrays with `as_arrays`."""
        if stop_on not in ("patience", "optimistic_future_cost"):
            raise ValueError("Unknown stopping criterion {}".format(stop_on))
        rec, gen = self.rec, self.rec.generator
        lm = gen.language_model
        dev = rec.device
        lm_on_device = getattr(lm, "on_device", False)

        def to_dev(idx):
            return torch.as_tensor(numpy.ascontiguousarray(idx), dtype=torch.int64).to(dev)

        with rec._on_stream():
            rec.compute_contexts(input_values["recordings"])
            start = gen.generation_initial_states(1)
            lm_states = rec.lm_initial_states(1) if lm is not None else None
        S, W, step = start["states"], start["weights"], start["step"]
        tokens = start["outputs"][None, :]                          # (positions so far + 1, live hypotheses)
        running = numpy.zeros(tokens.shape, dtype=numpy.float32)     # cumulative costs, same layout
        finished = []
        best_seen, patience = 1000, None

        for position in range(max_length):
            if S.shape[0] == 0:
                break
            # ---- stopping rules
            if stop_on == "patience":
                finished.sort(key=lambda f: f.score(char_discount))
                del finished[self.beam_size:]
                if finished:
                    leader = finished[0].score(char_discount)
                    if leader < best_seen:
                        best_seen, patience = leader, self.PATIENCE
                    else:
                        if patience is None:      # the reference decrements before it ever assigns (first cost >= 1000)
                            raise UnboundLocalError("local variable 'patience' referenced before assignment")
                        patience -= 1
                        if patience == 0:
                            break
            elif len(finished) >= self.beam_size:
                bound = running[-1].min() - char_discount * max_length
                if finished[self.beam_size - 1].score(char_discount) < bound:
                    break
            # ---- expand: cost of every continuation of every live hypothesis
            with rec._on_stream():
                if lm_states is not None:
                    lm.stage(lm_states, dev)
                neglogp = gen.generation_logprobs(S, W, step)
            step_costs = neglogp.cpu().numpy().astype(numpy.float32)
            assert numpy.isfinite(step_costs).all()
            (parents, chars), chosen = self._smallest(running[-1][:, None] + step_costs, self.beam_size)
            # ---- re-arrange the beam along the chosen parents and advance it by the chosen characters
            with rec._on_stream():
                parents_t = to_dev(parents)
                S, W = S.index_select(0, parents_t), W.index_select(0, parents_t)
                if lm_states is not None:
                    lm_states = lm.take(lm_states, parents_t if lm_on_device else parents)
                nxt = gen.generation_next_states(S, W, step, chars)
                if lm_states is not None:
                    lm_states = lm.transition(lm_states, chars)
            S, W, step = nxt["states"], nxt["weights"], nxt["step"]
            tokens = numpy.concatenate([tokens[:, parents], chars[None, :]], axis=0)
            running = numpy.concatenate([running[:, parents], chosen[None, :]], axis=0)
            # ---- hypotheses that just emitted <eol> finish (unless the step was "infinitely" expensive) and leave the beam
            ended = chars == eol_symbol
            affordable = (running[-1] - running[-2]) < round_to_inf
            for col in numpy.flatnonzero(ended & affordable):
                if validate_solution_function is None or validate_solution_function(input_values, tokens[:, col]):
                    finished.append(_Finished(tokens[:, col], running[:, col]))
            alive = numpy.ones_like(ended) if (ignore_first_eol and position == 0) else ~ended
            if not alive.all():
                keep = numpy.flatnonzero(alive)
                with rec._on_stream():
                    keep_t = to_dev(keep)
                    S, W = S.index_select(0, keep_t), W.index_select(0, keep_t)
                    if lm_states is not None:
                        lm_states = lm.take(lm_states, keep_t if lm_on_device else keep)
                tokens, running = tokens[:, keep], running[:, keep]

        if not finished:
            raise CandidateNotFoundError()
        finished.sort(key=lambda f: f.score(char_discount))
        longest = max(len(f.tokens) for f in finished)
        out_tokens = numpy.zeros((longest, len(finished)))
        out_mask = numpy.zeros((longest, len(finished)))
        out_running = numpy.zeros((longest, len(finished)))
        for col, f in enumerate(finished):
            n = len(f.tokens)
            out_tokens[:n, col] = f.tokens
            out_mask[:n, col] = 1
            out_running[:n, col] = f.running
            out_running[n:, col] = f.running[-1]
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
