"""Beam search over the HIP generation step: host-side bookkeeping identical to the reference's (modified)
`blocks.search.BeamSearch` (libs/blocks/blocks/search.py:19-407: `char_discount`, `round_to_inf`, `stop_on`,
shrinking beam, numpy tie order in `_smallest`), with the two compiled Theano functions it calls per step
replaced by `SequenceGenerator.generation_logprobs` / `generation_next_states` on the device.
"""
import numpy
import torch


class CandidateNotFoundError(Exception):
    """search.py:15-16"""
    pass


class BeamSearch(object):
    def __init__(self, beam_size, recognizer):
        self.beam_size = beam_size
        self.rec = recognizer

    @staticmethod
    def _smallest(matrix, k):
        """search.py:221-242, verbatim semantics (argpartition then argsort: numpy's tie order is part of parity)."""
        flatten = matrix.flatten()
        if flatten.shape[0] > k:
            args = numpy.argpartition(flatten, k)[:k]
        else:
            args = numpy.arange(flatten.shape[0])
        args = args[numpy.argsort(flatten[args])]
        return numpy.unravel_index(args, matrix.shape), flatten[args]

    def search(self, input_values, eol_symbol, max_length, ignore_first_eol=False, as_arrays=False, char_discount=0,
               round_to_inf=1e9, stop_on="patience", validate_solution_function=None):
        """search.py:244-399.  `input_values` = {'recordings': (T,1,F) ndarray}."""
        rec, gen = self.rec, self.rec.generator
        dev = rec.device
        with rec._on_stream():
            rec.compute_contexts(input_values["recordings"])
            st = gen.generation_initial_states(1)
            lm_states = rec.lm_initial_states(1) if gen.language_model is not None else None
        S, W, step = st["states"], st["weights"], st["step"]
        all_outputs = st["outputs"][None, :]
        all_costs = numpy.zeros_like(all_outputs, dtype=numpy.float32)
        done = []
        min_cost = 1000
        # one host->device copy of an index vector per use site, shared by the decoder state, the alignments and the LM state
        to_dev = lambda idx: torch.as_tensor(numpy.ascontiguousarray(idx), dtype=torch.int64).to(dev)
        lm_take = lambda st, idx, idx_t: gen.language_model.take(st, idx_t if getattr(gen.language_model, "on_device", False) else idx)
        for i in range(max_length):
            if S.shape[0] == 0:
                break
            if stop_on == "patience":
                done = sorted(done, key=lambda x: x[1][-1] - char_discount * len(x[1]))
                done = done[:self.beam_size]
                if done:
                    current_best_cost = done[0][1][-1] - char_discount * len(done[0][1])
                    if current_best_cost < min_cost:
                        min_cost = current_best_cost
                        patience = 30
                    else:
                        patience -= 1
                        if patience == 0:
                            break
            elif stop_on == "optimistic_future_cost":
                if len(done) >= self.beam_size:
                    optimistic_future_cost = all_costs[-1, :].min() - char_discount * max_length
                    last_in_done = done[self.beam_size - 1][1]
                    last_in_done_cost = last_in_done[-1] - char_discount * len(last_in_done)
                    if last_in_done_cost < optimistic_future_cost:
                        break
            else:
                raise ValueError("Unknown stopping criterion {}".format(stop_on))
            with rec._on_stream():
                if lm_states is not None:
                    gen.language_model.stage(lm_states, dev)
                nl = gen.generation_logprobs(S, W, step)
            logprobs = nl.cpu().numpy().astype(numpy.float32)
            assert numpy.isfinite(logprobs).all()
            next_costs = all_costs[-1, :, None] + logprobs
            (indexes, outputs), chosen_costs = self._smallest(next_costs, self.beam_size)
            # Rearrange everything
            with rec._on_stream():
                idx_t = to_dev(indexes)
                S, W = S.index_select(0, idx_t), W.index_select(0, idx_t)
                if lm_states is not None:
                    lm_states = lm_take(lm_states, indexes, idx_t)
                all_outputs = numpy.take(all_outputs, indexes, axis=1)
                all_costs = numpy.take(all_costs, indexes, axis=1)
                # Record chosen output and compute new states
                st = gen.generation_next_states(S, W, step, outputs)
                if lm_states is not None:
                    lm_states = gen.language_model.transition(lm_states, outputs)
            S, W, step = st["states"], st["weights"], st["step"]
            all_outputs = numpy.vstack([all_outputs, outputs[None, :]])
            all_costs = numpy.vstack([all_costs, chosen_costs[None, :]])
            mask = outputs != eol_symbol
            if ignore_first_eol and i == 0:
                mask[:] = 1
            for idx in numpy.where((all_outputs[-1] == eol_symbol) & (all_costs[-1] - all_costs[-2] < round_to_inf))[0]:
                if validate_solution_function is None or validate_solution_function(input_values, all_outputs[:, idx]):
                    done.append((all_outputs[:, idx], all_costs[:, idx]))
            unfinished = numpy.where(mask == 1)[0]
            if len(unfinished) != len(mask):                # nothing to drop in most steps: keep the tensors as they are
                with rec._on_stream():
                    idx_t = to_dev(unfinished)
                    S, W = S.index_select(0, idx_t), W.index_select(0, idx_t)
                    if lm_states is not None:
                        lm_states = lm_take(lm_states, unfinished, idx_t)
                all_outputs = numpy.take(all_outputs, unfinished, axis=1)
                all_costs = numpy.take(all_costs, unfinished, axis=1)
        if not done:
            raise CandidateNotFoundError()
        done = sorted(done, key=lambda x: x[1][-1] - char_discount * len(x[1]))
        max_len = max((seq[0].shape[0] for seq in done))
        all_outputs = numpy.zeros((max_len, len(done)))
        all_masks = numpy.zeros((max_len, len(done)))
        all_costs = numpy.zeros((max_len, len(done)))
        for i, (seq, cost) in enumerate(done):
            all_outputs[:len(seq), i] = seq
            all_masks[:len(seq), i] = 1
            all_costs[:len(cost), i] = cost
            all_costs[len(cost):, i] = cost[-1]
        all_outputs = all_outputs[1:]
        all_masks = all_masks[1:]
        all_costs = all_costs[1:] - all_costs[:-1]
        result = all_outputs, all_masks, all_costs
        if as_arrays:
            return result
        return self.result_to_lists(result)

    @staticmethod
    def result_to_lists(result):
        """search.py:401-407"""
        outputs, masks, costs = [array.T for array in result]
        outputs = [list(output[:int(mask.sum())]) for output, mask in zip(outputs, masks)]
        costs = list(costs.T.sum(axis=0))
        return outputs, costs
