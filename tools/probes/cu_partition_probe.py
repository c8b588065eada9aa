"""GPU probe (round-5 verdict, ask 5): weight-gradient products on a CU-masked side stream beside the next layer's BPTT clusters.

hipExtStreamCreateWithCUMask gives a stream whose kernels run on a subset of the CUs.  The probe builds two such streams — G CUs for the
products (side), the remaining 256 - G for everything else (main: clusters, critical-path products) — switches the encoder's second
stream on (bricks.Encoder.overlap: per-layer grouped launches on the side stream) and times the WSJ-base step with EAGER launches (a
hipGraph kernel node carries no CU mask: hipLaunchAttributeID has no such member, so inside the whole-step graph the partition
cannot be expressed).  A/B against the eager step without overlap, and the replayed whole-step graph for reference.

    python tools/probes/cu_partition_probe.py [G ...]
"""
import ctypes
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "attention-lvcsr_amd")]
import torch

from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
TRAIN_CONF = dict(gradient_threshold=100.0, rules=("momentum", "adadelta"), scale=0.1, decay_rate=0.95, epsilon=1e-8, max_norm=1.0)


def masked_stream(bits):
    """CU mask as a bit vector of 256 bits; bit i is CU i as the runtime numbers them (consecutive bits alternate over the XCDs)."""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask failed: %d" % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def timed(trainer, batches, n, warm=3):
    for k in range(warm):
        trainer.train_step(batches[k % len(batches)], global_batch_size=16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        trainer.train_step(batches[k % len(batches)], global_batch_size=16)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def build(use_graph, main=None, side=None, knobs=()):
    cfg = spec.wsj_base()
    rec = SpeechRecognizer(device=dev, params=synthetic.make_params(cfg, seed=10), net_config=cfg, use_graph=use_graph)
    rec.lib.set_knobs(list(knobs))
    if main is not None:
        rec.stream = main
    if side is not None:
        rec.encoder.overlap = True
        rec.encoder._side = side
    tr = Trainer(rec, distributed=False, **TRAIN_CONF)
    batches = [{k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(cfg, 16, 800, 100, seed=1234 + s).items()} for s in range(2)]
    return rec, tr, batches


def main():
    gs = [int(a) for a in sys.argv[1:]] or [64, 96, 128]
    print("| configuration | ms per WSJ-base step |")
    print("|---|---|")
    rec, tr, b = build(True)
    print("| whole-step hipGraph (the product), all 256 CUs | %.2f |" % timed(tr, b, 20, warm=5))
    rec, tr, b = build(False)
    base = timed(tr, b, 20)
    print("| eager launches, no second stream, all 256 CUs | %.2f |" % base)
    rec, tr, b = build(False, side=torch.cuda.Stream(dev))
    print("| eager, products on an UNMASKED second stream (round 3's experiment) | %.2f |" % timed(tr, b, 20))
    for G in gs:
        for flags, what in ((0, "clusters as the library picks them for %d CUs"), (64, "clusters of 4 for %d CUs")):
            try:
                main_s, side_s = masked_stream(range(G, 256)), masked_stream(range(0, G))
                rec, tr, b = build(False, main=main_s, side=side_s, knobs=["max_cluster_wgs=%d" % (256 - G), "persist_flags=%d" % flags])
                ms = timed(tr, b, 20)
                rec.encoder.check_persistent(); rec.generator.check_persistent()
                print("| eager, products on %d masked CUs, everything else on the other %d; %s | %.2f |" % (G, 256 - G, what % (256 - G), ms), flush=True)
            except Exception as e:                                     # a cluster launch that does not fit its partition aborts: report, go on
                print("| eager, products on %d masked CUs; %s | failed: %s |" % (G, what % (256 - G), str(e)[:120]), flush=True)
            finally:
                rec.lib.set_knobs([])
    # the clusters alone on a partition: what does shrinking their share of the chip cost without any second stream
    for G in gs:
        main_s = masked_stream(range(G, 256))
        rec, tr, b = build(False, main=main_s, knobs=["max_cluster_wgs=%d" % (256 - G)])
        try:
            print("| eager, no second stream, everything on %d CUs | %.2f |" % (256 - G, timed(tr, b, 20)), flush=True)
        except Exception as e:
            print("| eager, no second stream, everything on %d CUs | failed: %s |" % (256 - G, str(e)[:120]), flush=True)
        rec.lib.set_knobs([])


if __name__ == "__main__":
    main()
