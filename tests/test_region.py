"""State machine of `lvsr_amd.native.Region` (whole-step graph regions) against a recording stand-in for the library: eager
first pass with inner graphs suppressed, capture on the second pass, replays afterwards, re-capture when the volatile part
of the key changes, and the allocation guard (a capture during which device memory was handed out is dropped and the
key stays eager).  The real capture/replay runs in tests/test_gpu_kernels.py::test_whole_step_graph_region_gpu."""
from lvsr_amd import native


class FakeLib(object):
    is_emulator = False
    step_graph = True              # (native.Lib constructor arguments)
    sync_after_graph = True

    def __init__(self):
        self.capturing = False
        self.calls = []
        self.cached = set()
        self.not_capturable = False

    unique_token = native.Lib.unique_token
    _uid = native.Lib._uid

    def stream_for(self, t):
        return 0

    def after_graph(self, ref, steps):
        self.calls.append("sync")

    def last_error(self):
        return ""

    def _lvsr_graph_suppress(self, on):
        self.calls.append("suppress%d" % on)

    def _lvsr_region_begin(self, stream, key, n):
        self._key = key
        if key in self.cached:
            self.calls.append("launch")
            return 1
        if self.not_capturable:
            return 2
        self.calls.append("begin")
        return 0

    def _lvsr_region_end(self, stream, keep):
        self.calls.append("end%d" % keep)
        if keep:
            self.cached.add(self._key)
        return 0


class FakeTensor(object):
    is_cuda = True
    device = "cuda:0"


class Owner(object):
    pass


def make(lib, owner, allocs, key=("k",), volatile=(1,)):
    r = native.Region(lib, owner, key, FakeTensor(), True, volatile)
    r._alloc_count = lambda: allocs[0]
    return r


def test_eager_capture_replay_and_recapture():
    lib, owner, allocs, ran = FakeLib(), Owner(), [0], []
    fn = lambda: (ran.append(1), "out%d" % len(ran))[1]
    assert make(lib, owner, allocs).run(fn) == "out1"
    assert lib.calls == ["suppress1", "suppress0"] and not lib.capturing
    lib.calls.clear()
    assert make(lib, owner, allocs).run(fn) == "out2"                    # captured (body ran once more, recorded)
    assert lib.calls == ["begin", "end1", "sync"] and not lib.capturing
    lib.calls.clear()
    assert make(lib, owner, allocs).run(fn) == "out2" and len(ran) == 2  # replay: body skipped, cached result
    assert lib.calls == ["launch", "sync"]
    lib.calls.clear()
    assert make(lib, owner, allocs, volatile=(2,)).run(fn) == "out3"     # buffers moved: straight to a new capture
    assert lib.calls == ["begin", "end1", "sync"]
    other = Owner()                                                      # another owner never sees these graphs
    lib.calls.clear()
    make(lib, other, allocs).run(fn)
    assert lib.calls == ["suppress1", "suppress0"] and owner._region_token != other._region_token


def test_allocation_inside_a_capture_drops_it():
    lib, owner, allocs, ran = FakeLib(), Owner(), [0], []

    def fn():
        ran.append(1)
        if len(ran) == 2:
            allocs[0] += 1                                               # the caching allocator handed out memory
        return len(ran)
    make(lib, owner, allocs).run(fn)
    lib.calls.clear()
    assert make(lib, owner, allocs).run(fn) == 3                         # capture dropped, body enqueued again eagerly
    assert lib.calls == ["begin", "end0"]
    lib.calls.clear()
    assert make(lib, owner, allocs).run(fn) == 4 and lib.calls == []     # stays eager


def test_uncapturable_stream_and_disabled_regions():
    lib, owner, allocs = FakeLib(), Owner(), [0]
    lib.not_capturable = True
    make(lib, owner, allocs).run(lambda: 1)
    assert make(lib, owner, allocs).run(lambda: 2) == 2
    lib.calls.clear()
    assert make(lib, owner, allocs).run(lambda: 3) == 3 and lib.calls == []
    r = native.Region(lib, Owner(), ("k",), FakeTensor(), False)
    assert r.run(lambda: 5) == 5 and lib.calls == []
    lib.capturing = True                                                 # no nesting
    assert not native.Region(lib, Owner(), ("k",), FakeTensor(), True).enabled
