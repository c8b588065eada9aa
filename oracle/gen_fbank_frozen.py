import sys, numpy
sys.path[:0] = ['/root/repo']
from oracle import fbank_oracle as FO
rng = numpy.random.RandomState(7)
n = 400 + 160 * 11
t = numpy.arange(n) / 16000.0
wav = numpy.clip(2500 * numpy.sin(2 * numpy.pi * 700 * t) + 900 * numpy.sin(2 * numpy.pi * 3100 * t + 0.5) + rng.normal(0, 200, n) - 75, -32768, 32767).astype(numpy.int16)
f = FO.fbank(wav)
numpy.savez_compressed('/root/repo/tests/golden/fbank_frozen.npz', wav=wav, fbank=f.astype(numpy.float64), full=FO.add_deltas(f).astype(numpy.float64))
print(f.shape, f[0, :5])
