"""Stand-ins for the Blocks / lvsr classes that the reference's YAML configs name through
`!!python/name:` / `!!python/object/apply:` tags (lvsr/configs/prototype_speech.yaml:2-29,
exp/wsj/configs/wsj_jan_new.yaml:25-72).  They carry configuration only — the arithmetic is in the HIP library —
except the initialisation schemes, which generate numpy arrays exactly as blocks/initialization.py does.
"""
import numpy


# ---- brick markers (net section) ---------------------------------------------------------------------
class _Marker(object):
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(map(repr, self.args)))


class GatedRecurrent(_Marker):          # blocks.bricks.recurrent.GatedRecurrent (recurrent.py:486-624)
    pass


class SimpleRecurrent(_Marker):         # not built: selecting it raises in spec.from_reference_kwargs
    pass


class LSTM(_Marker):
    pass


class Tanh(_Marker):
    pass


class Rectifier(_Marker):
    pass


class Identity(_Marker):
    pass


class Logistic(_Marker):
    pass


class Maxout(_Marker):                  # blocks.bricks.Maxout(num_pieces) (simple.py:134-181)
    def __init__(self, num_pieces=2, **kwargs):
        _Marker.__init__(self, num_pieces, **kwargs)
        self.num_pieces = num_pieces


class SpeechBottom(_Marker):            # lvsr.bricks.recognizer.SpeechBottom (recognizer.py:105-157)
    pass


class LookupBottom(_Marker):
    pass


class H5PYAudioDatasetTimit(_Marker):   # lvsr.datasets.h5py.H5PYAudioDatasetTimit (exp/timit configs: `data.dataset_class`)
    pass


class H5PYAudioDataset(_Marker):        # lvsr.datasets.h5py.H5PYAudioDataset — data layer is out of scope
    pass


# ---- initialisation schemes (libs/blocks/blocks/initialization.py:80-209) --------------------------------
class NdarrayInitialization(object):
    def generate(self, rng, shape):
        raise NotImplementedError


class Constant(NdarrayInitialization):
    """initialization.py:50-77"""
    def __init__(self, constant=0.0):
        self._constant = numpy.asarray(constant)

    def generate(self, rng, shape):
        dest = numpy.empty(shape, dtype=numpy.float32)
        dest[...] = self._constant
        return dest


class IsotropicGaussian(NdarrayInitialization):
    """initialization.py:80-103"""
    def __init__(self, std=1, mean=0):
        self._mean, self._std = mean, std

    def generate(self, rng, shape):
        return rng.normal(self._mean, self._std, size=shape).astype(numpy.float32)


class Uniform(NdarrayInitialization):
    """initialization.py:105-139"""
    def __init__(self, mean=0., width=None, std=None):
        if (width is not None) == (std is not None):
            raise ValueError("must specify width or std, but not both")
        self._mean = mean
        self._width = width if width is not None else numpy.sqrt(12) * std

    def generate(self, rng, shape):
        w = self._width / 2
        return rng.uniform(self._mean - w, self._mean + w, size=shape).astype(numpy.float32)


def _random_rotation(rng, n):
    """Orthogonal factor of an (n, n) standard-normal draw, columns sign-fixed by the diagonal of R (unique QR)."""
    q, r = numpy.linalg.qr(rng.randn(n, n).astype(numpy.float32))
    return q * numpy.sign(numpy.diag(r))


class Orthogonal(NdarrayInitialization):
    """initialization.py:163-209.  Square: one random rotation.  Rectangular (rows, cols): the product of the leading
    min(rows, cols) columns of a (rows x rows) rotation with the leading rows of a (cols x cols) rotation, drawn in that
    order from the same generator (the order of the two draws is what makes the values reproducible)."""
    def __init__(self, scale=1):
        self.scale = scale

    def generate(self, rng, shape):
        if len(shape) != 2:
            raise ValueError
        rows, cols = shape
        if rows == cols:
            out = _random_rotation(rng, rows)
        else:
            left = _random_rotation(rng, rows)
            right = _random_rotation(rng, cols)
            k = min(rows, cols)
            out = left[:, :k] @ right[:k, :]
        return (out * self.scale).astype(numpy.float32)


# python path (as written in the reference's YAML tags) -> class
REGISTRY = {
    "blocks.bricks.recurrent.GatedRecurrent": GatedRecurrent,
    "blocks.bricks.recurrent.SimpleRecurrent": SimpleRecurrent,
    "blocks.bricks.recurrent.LSTM": LSTM,
    "blocks.bricks.Tanh": Tanh, "blocks.bricks.Rectifier": Rectifier, "blocks.bricks.Identity": Identity,
    "blocks.bricks.Logistic": Logistic, "blocks.bricks.Maxout": Maxout,
    "lvsr.bricks.recognizer.SpeechBottom": SpeechBottom, "lvsr.bricks.recognizer.LookupBottom": LookupBottom,
    "lvsr.datasets.h5py.H5PYAudioDataset": H5PYAudioDataset,
    "lvsr.datasets.h5py.H5PYAudioDatasetTimit": H5PYAudioDatasetTimit,
    "blocks.initialization.Constant": Constant, "blocks.initialization.IsotropicGaussian": IsotropicGaussian,
    "blocks.initialization.Uniform": Uniform, "blocks.initialization.Orthogonal": Orthogonal,
}
