"""Decode driver: what `lvsr.main.search` reports per utterance (lvsr/main.py:705-864) — beam search, negative
log-likelihood of the ground truth and of the best hypothesis through `analyze`, character/word error counts — around
`SpeechRecognizer.beam_search`.  Dataset access is the caller's business: utterances come in as
(recordings (T,F) ndarray, groundtruth label list) pairs."""
import numpy

from .error_rate import edit_distance, weights_std, monotonicity_penalty
from .search import CandidateNotFoundError


def _searched(recognizer, utterances, batch, kw):
    """(number, recordings, groundtruth, result of the beam search or the exception it raised), in order; `batch` > 1: the searches
    of that many utterances side by side in one set of launches (SpeechRecognizer.beam_search_batch)."""
    chunk = []
    for number, (recordings, groundtruth) in enumerate(utterances):
        if batch <= 1:
            try:
                yield number, recordings, groundtruth, recognizer.beam_search({"recordings": recordings}, **kw)
            except CandidateNotFoundError as e:
                yield number, recordings, groundtruth, e
            continue
        chunk.append((number, recordings, groundtruth))
        if len(chunk) == batch:
            for item, res in zip(chunk, recognizer.beam_search_batch([c[1] for c in chunk], **kw)):
                yield item + (res,)
            chunk = []
    if chunk:
        for item, res in zip(chunk, recognizer.beam_search_batch([c[1] for c in chunk], **kw)):
            yield item + (res,)


def search(recognizer, utterances, beam_size=10, char_discount=0.0, round_to_inf=1e9, stop_on="patience", to_words=None,
           report=None, batch=1):
    """-> dict(per_utterance=[...], cer=, wer=, nll_groundtruth=, nll_recognized=).  `to_words(labels) -> list[str]` turns a
    label sequence into words for WER (the reference decodes characters with its character map, lvsr/main.py:788-800).
    `batch` (an addition): utterances decoded side by side, several times the throughput on a GPU.  An utterance's hypotheses do not
    depend on the batch it is decoded in for any batch > 1 (trailing chunks and one-utterance chunks included: the readout's merge
    products always run through lvsr_readout_merge there); against batch = 1 (per-row VALU products below 64 rows: another float32
    summation order) costs agree to ~1e-5 relative and candidates closer than that may swap ranks.  A host-side language model
    or a validator decode one utterance at a time whatever `batch` says (BeamSearch.search_batch)."""
    recognizer.init_beam_search(beam_size)
    rows, tot_err, tot_len, tot_werr, tot_wlen = [], 0, 0, 0, 0
    kw = dict(char_discount=char_discount, round_to_inf=round_to_inf, stop_on=stop_on)
    for number, recordings, groundtruth, result in _searched(recognizer, utterances, int(batch), kw):
        groundtruth = [int(t) for t in groundtruth]
        inputs = {"recordings": recordings}
        row = dict(number=number, groundtruth=groundtruth)
        if isinstance(result, CandidateNotFoundError):                       # lvsr/main.py:808-813
            recognized = []
            row.update(recognized=[], search_cost=float("nan"), error="CandidateNotFoundError")
        elif isinstance(result, Exception):
            raise result
        else:
            outputs, search_costs = result
            recognized = outputs[0]
            row.update(recognized=recognized, search_cost=search_costs[0])
        gt_cost, gt_weights, _ = recognizer.analyze(inputs, numpy.asarray(groundtruth))
        row.update(groundtruth_cost=float(gt_cost.sum()),
                   weights_std=float(weights_std(gt_weights[:, None, :])),
                   monotonicity_penalty=float(monotonicity_penalty(gt_weights[:, None, :])))
        if recognized:
            rec_cost, _, _ = recognizer.analyze(inputs, numpy.asarray(groundtruth), numpy.asarray(recognized))
            row["recognized_cost"] = float(rec_cost.sum())
        err = int(edit_distance(groundtruth, recognized))
        row.update(char_errors=err, cer=err / float(len(groundtruth)))
        tot_err += err
        tot_len += len(groundtruth)
        if to_words is not None:
            gw, rw = to_words(groundtruth), to_words(recognized)
            werr = int(edit_distance(gw, rw))
            row.update(word_errors=werr, wer=werr / float(max(1, len(gw))))
            tot_werr += werr
            tot_wlen += len(gw)
        rows.append(row)
        if report is not None:
            report(row)
    out = dict(per_utterance=rows, cer=tot_err / float(max(1, tot_len)))
    if to_words is not None:
        out["wer"] = tot_werr / float(max(1, tot_wlen))
    return out
