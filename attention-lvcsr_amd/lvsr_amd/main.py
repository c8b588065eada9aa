"""Thin stage driver: what `lvsr.main.train` / `train_multistage` do with the path
(lvsr/main.py:140-703, 896-922), without the Blocks main loop, its extensions, plotting or logging back-ends:

  * one stage = build the recognizer from the stage's `net:` section, load parameters (`params`, or the previous stage's
    checkpoint `<save_path>/<previous stage><restart_from>.zip`), build the step rules from `training:` / `regularization:`
    (+ AdaptiveClipping, lvsr/main.py:616-619), iterate `data.get_stream("train")` until `num_batches` / `num_epochs`
    (FinishAfter, :624-626), validate after every epoch, keep `<stage>.zip` (last) and `<stage>_best_ll.zip` (best
    validation cost; the reference's TrackTheBest + Checkpoint conditions :606-660), stop on `patience`
    (lvsr/extensions.py `Patience`: no new best for max(min_epochs, patience_factor x epoch-of-best) epochs);
  * stages run in `number` order, each starting from its predecessor (train_multistage :896-922).

The character-error-rate monitor (`search_every_epochs`, `<stage>_best.zip`) runs through `lvsr_amd.decode.search` when
`monitoring.search_every_epochs` is set.  Returns a log (list of dict rows), the reference's `main_loop.log` in spirit.
"""
import os

import numpy

from .bricks.recognizer import SpeechRecognizer
from .training import Trainer


def _dist_state(distributed):
    """(rank, world, barrier) of the data-parallel job this process belongs to (a single process: 0, 1, no-op)."""
    import torch
    if distributed is None:
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    if not distributed:
        return 0, 1, (lambda: None)
    return torch.distributed.get_rank(), torch.distributed.get_world_size(), torch.distributed.barrier


def _save_atomically(recognizer, path, trainer=None, counters=None):
    """Checkpoints are written to a temporary name and renamed: a reader (the next stage, another rank) never sees a
    half-written tar.  With a trainer the optimiser state and the loop counters ride along as `_training_state`."""
    root, ext = os.path.splitext(path)
    tmp = "%s.tmp%d%s" % (root, os.getpid(), ext)
    extra = None
    if trainer is not None:
        state = trainer.state_dict()
        state.update({k: numpy.asarray(v) for k, v in (counters or {}).items()})
        extra = {"_training_state": state}
    recognizer.save_params(tmp, extra=extra)
    os.replace(tmp, path)


def validate(recognizer, data, part="valid"):
    """Mean cost per utterance over a data part (validation monitor of `cost`, lvsr/main.py:556-570)."""
    total, count = 0.0, 0
    for batch in data.get_stream(part, shuffle=False):
        cm = recognizer.cost(recordings=batch["recordings"], inputs_mask=batch["recordings_mask"], labels=batch["labels"],
                             labels_mask=batch["labels_mask"], save_for_backward=False)
        total += float(cm.sum())                      # (synchronises)
        # a persistent cluster that gave up leaves garbage costs, and the next launch's memset would erase its abort word:
        # check per batch — valid_cost drives the _best_ll checkpoint and the patience stop
        recognizer.encoder.check_persistent()
        recognizer.generator.check_persistent()
        count += int(batch["labels"].shape[1])
    return total / max(1, count)


def train(config, data, save_path, params=None, device="cuda:0", lib=None, log=None, distributed=None, search_subset=10,
          resume=False, stage=None):
    """One stage.  `config`: a (stage) configuration mapping with `net`, `training`, optional `regularization`,
    `monitoring`, `initialization`; `data`: lvsr_amd.data.Data.  Returns (recognizer, log).
    `resume=True`: `params` is a checkpoint of THIS stage written by an earlier call: besides the parameters, the optimiser
    accumulators, the adaptive-clipping statistics and the epoch / iteration / best-cost counters are restored from its
    `_training_state` member, so the recipe continues where it stopped (the reference resumes from its pickled main loop)."""
    from .config import Configuration
    log = [] if log is None else log
    reg = dict(config.get("regularization") or {})
    unbuilt = [k for k in ("dropout", "noise", "adaptive_noise") if reg.get(k)] + \
              [k for k in ("penalty_coof", "decay") if (reg.get(k) or 0) > 0]
    if unbuilt:        # lvsr/main.py:395-470 rewrites the Theano graph for these; silently ignoring them would change the recipe
        raise NotImplementedError("regularization.%s is not built (only max_norm is)" % ", regularization.".join(unbuilt))
    train_conf, mon = dict(config.get("training", {})), dict(config.get("monitoring", {}))
    net = dict(config["net"])
    kw = Configuration.net_kwargs(config, data.num_features(), data.num_labels, eos_label=data.eos_label) \
        if isinstance(config, Configuration) else dict(net, input_dims={"recordings": data.num_features()},
                                                       num_phonemes=data.num_labels, eos_label=data.eos_label)
    kw.setdefault("data_prepend_eos", False)
    rec = SpeechRecognizer(device=device, lib=lib, **kw)
    if not config.get("initialization") and not params:
        # the reference fails on config['initialization'] (lvsr/main.py:225); training from all-zero parameters is never meant
        raise KeyError("neither an `initialization` section nor `params` to start from")
    if config.get("initialization"):
        rec.initialize(config["initialization"])
    if params:
        rec.load_params(params)
    rank, world, barrier = _dist_state(distributed)
    trainer = Trainer.from_config(rec, train_conf, config.get("regularization"), distributed=world > 1 or bool(distributed))
    root, ext = os.path.splitext(save_path)
    best_ll, best_per, best_epoch = float("inf"), float("inf"), 0
    num_batches, num_epochs = train_conf.get("num_batches"), train_conf.get("num_epochs")
    patience = train_conf.get("patience")
    iterations, epoch, done = 0, 0, False
    if resume:
        from .checkpoint import load_member
        state = load_member(params, "_training_state") if params else None
        if state is None:
            raise ValueError("resume=True needs a checkpoint with a `_training_state` member (written by this driver)")
        if stage is not None and "stage" in state and str(state["stage"]) != str(stage):
            raise ValueError("resume=True: the checkpoint holds the training state of stage %r, not of %r" % (str(state["stage"]), stage))
        trainer.load_state_dict(state)
        iterations, epoch = int(state["iterations_done"]), int(state["epochs_done"])
        if num_batches and iterations >= num_batches:        # the stage had already finished on its batch budget: nothing left to do
            return rec, log
        best_ll, best_per, best_epoch = float(state["best_ll"]), float(state["best_per"]), int(state["best_epoch"])
    has_valid = "valid" in data.datasets
    while not done:
        costs = []
        # data parallel: every rank walks the same seeded stream and keeps utterances rank::world of each global minibatch
        for batch in data.get_stream("train", shuffle=True, seed=epoch, rank=rank, world=world):
            if batch is None:                       # a global minibatch smaller than the world leaves this rank without utterances
                continue
            gbs = batch.pop("global_batch_size", None)
            cm = trainer.train_step(batch, global_batch_size=gbs)
            for attempt in range(3):
                if not trainer.step_was_skipped():
                    break
                # a persistent cluster kernel gave up waiting for its partners (its work-groups were not all resident: the device is
                # shared): the optimiser skipped the step on the device — on every rank, the flag travels with the gradients.
                # Trainer.recover() first leaves CUs free for the other tenant, then falls back to the step kernels for a while
                # (and arms the cluster kernels again after Trainer.REARM_STEPS clean steps); run the batch again.
                action = trainer.recover()
                log.append(dict(iterations_done=iterations, persistent_kernels_aborted=True, **action))
                cm = trainer.train_step(batch, global_batch_size=gbs)
            if trainer.step_was_skipped():
                # the last attempt (already on the step kernels) was skipped as well: nothing was updated — the iteration is neither
                # counted nor logged with the cost of a step that did not happen, and training does not go on over it
                log.append(dict(iterations_done=iterations, persistent_kernels_aborted=True, action="gave_up",
                                training_finish_requested="a step was skipped on the device %d times in a row" % 4))
                trainer.close()
                raise RuntimeError("training step %d was skipped on the device 4 times in a row (cluster kernels, a reserve of CUs and the "
                                   "step kernels all gave up): the device is not usable for this run" % iterations)
            iterations += 1
            row = dict(iterations_done=iterations, epochs_done=epoch, train_cost=float(cm.sum()) / int(batch["labels"].shape[1]),
                       total_gradient_norm=trainer.gradient_norm(), gradient_norm_threshold=trainer.gradient_threshold())
            costs.append(row["train_cost"])
            log.append(row)
            if not numpy.isfinite(row["total_gradient_norm"]):        # FinishAfter(...).add_condition(_gradient_norm_is_none)
                row["training_finish_requested"] = "gradient norm is not finite"
                done = True
                break
            if num_batches and iterations >= num_batches:
                done = True
                break
        epoch += 1
        row = dict(iterations_done=iterations, epochs_done=epoch, average_train_cost=float(numpy.mean(costs)) if costs else None)
        if has_valid:
            # replicas are identical: every rank computes the same validation cost (keeps the stopping rules in step without
            # another collective); only rank 0 writes files
            row["valid_cost"] = validate(rec, data, "valid")
            if row["valid_cost"] < best_ll:
                best_ll, best_epoch = row["valid_cost"], epoch
                row["best_valid_cost_so_far"] = True
                if rank == 0:
                    _save_atomically(rec, root + "_best_ll" + ext)
            every = mon.get("search_every_epochs")
            if every and epoch % every == 0:
                from .decode import search
                ds = data.datasets["valid"]
                utts = [(ds.recordings[i], list(ds.labels[i]) + [data.eos_label]) for i in range(min(search_subset, ds.num_examples))]
                res = search(rec, utts, beam_size=mon.get("search", {}).get("beam_size", 10))
                row["valid_per"] = res["cer"]
                if res["cer"] < best_per:
                    best_per, best_epoch = res["cer"], epoch
                    row["best_valid_per_so_far"] = True
                    if rank == 0:
                        _save_atomically(rec, root + "_best" + ext)
        if rank == 0:
            _save_atomically(rec, save_path, trainer, dict(iterations_done=iterations, epochs_done=epoch, best_ll=best_ll,
                                                           best_per=best_per, best_epoch=best_epoch,
                                                           **({} if stage is None else {"stage": str(stage)})))
        barrier()                 # the next stage (any rank) may read the checkpoint as soon as its training returns
        log.append(row)
        if num_epochs and epoch >= num_epochs:
            done = True
        if patience and has_valid:
            allowed = max(patience.get("min_epochs", 0), patience.get("patience_factor", 1.5) * best_epoch)
            if epoch > allowed:
                row["patience_epoch"] = True
                done = True
        if not costs:
            done = True
    trainer.close()
    return rec, log


def train_multistage(config, data, save_path, params=None, start_stage=None, resume=False, **kwargs):
    """lvsr/main.py:896-922: stages in `number` order; stage k > 0 restarts from `<save_path>/<stage k-1><restart_from>.zip`.
    `resume=True` applies to the FIRST stage run only (its `params` is that stage's own checkpoint); later stages start fresh
    from their predecessor's parameters."""
    if not getattr(config, "multi_stage", False):
        return train(config, data, save_path, params, resume=resume, **kwargs)
    rank, _, barrier = _dist_state(kwargs.get("distributed"))
    if not start_stage and rank == 0:
        os.makedirs(save_path, exist_ok=True)
    barrier()
    stages = list(config.ordered_stages.items())
    first = list(config.ordered_stages).index(start_stage) if start_stage else 0
    rec, log = None, []
    for number in range(first, len(stages)):
        name, stage_config = stages[number]
        stage_save_path = "{}/{}.zip".format(save_path, name)
        if number and not params:
            stage_params = "{}/{}{}.zip".format(save_path, stages[number - 1][0], stage_config["training"].get("restart_from", ""))
        else:
            stage_params, params = params, None
        log.append(dict(stage=name))
        rec, log = train(stage_config, data, stage_save_path, stage_params, log=log, resume=resume and number == first, stage=name, **kwargs)
    return rec, log
