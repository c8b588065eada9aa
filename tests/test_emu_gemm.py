"""Kernel-source logic checks on the CPU emulator (tests/hipemu): GEMM family."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from emu import emu_lib


@pytest.mark.parametrize("transA,transB", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(70, 37, 29), (16, 16, 4), (65, 130, 33), (130, 141, 37), (96, 128, 16), (300, 390, 70), (256, 128, 80)])
def test_sgemm_layouts(transA, transB, M, N, K):
    lib = emu_lib()
    rng = numpy.random.RandomState(0)
    A = torch.tensor(rng.normal(size=(K, M) if transA else (M, K)), dtype=torch.float32)
    B = torch.tensor(rng.normal(size=(N, K) if transB else (K, N)), dtype=torch.float32)
    C0 = torch.tensor(rng.normal(size=(M, N)), dtype=torch.float32)
    bias = torch.tensor(rng.normal(size=(N,)), dtype=torch.float32)
    C = C0.clone()
    lib.sgemm(A, B, C, transA=transA, transB=transB, alpha=0.5, beta=2.0, bias=bias)
    ref = 0.5 * ((A.T if transA else A).double() @ (B.T if transB else B).double()) + 2.0 * C0.double() + bias.double()
    assert_allclose(C.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_sgemm_strided_and_splitk():
    lib = emu_lib()
    rng = numpy.random.RandomState(1)
    M, N, K = 20, 24, 1100
    big = torch.tensor(rng.normal(size=(K, 50)), dtype=torch.float32)
    A = big[:, 5:5 + M]            # (K, M) view with row stride 50 -> transA
    Bm = torch.tensor(rng.normal(size=(K, 40)), dtype=torch.float32)[:, 3:3 + N]
    C = torch.zeros(M, 64)[:, :N]
    ws = torch.empty(8 * M * N)
    lib.sgemm(A, Bm, C, transA=True, ws=ws)
    assert_allclose(C.numpy(), (A.double().T @ Bm.double()).numpy(), rtol=1e-4, atol=1e-4)


def test_sgemm128_strided_and_splitk():
    lib = emu_lib()
    rng = numpy.random.RandomState(2)
    M, N, K = 100, 132, 1060
    big = torch.tensor(rng.normal(size=(K, 120)), dtype=torch.float32)
    A = big[:, 8:8 + M]            # (K, M) view, row stride 120 -> transA, 16-B aligned base
    Bm = torch.tensor(rng.normal(size=(K, 140)), dtype=torch.float32)[:, 3:3 + N]      # unaligned base -> scalar path
    C = torch.zeros(M, 200)[:, :N]
    ws = torch.empty(6 * M * N)
    lib.sgemm(A, Bm, C, transA=True, ws=ws)
    assert_allclose(C.numpy(), (A.double().T @ Bm.double()).numpy(), rtol=1e-4, atol=1e-4)


def test_colsum_and_transpose():
    lib = emu_lib()
    rng = numpy.random.RandomState(2)
    X = torch.tensor(rng.normal(size=(37, 70)), dtype=torch.float32)
    out = torch.ones(70)
    lib.colsum(X, out, beta=1.0)
    assert_allclose(out.numpy(), 1 + X.double().sum(0).numpy(), rtol=1e-5, atol=1e-5)
    Y = torch.empty(70, 37)
    lib.transpose(X, Y)
    assert (Y == X.T).all()


@pytest.mark.parametrize("M,N,K,batch", [(20, 24, 9, 3), (130, 128, 40, 5), (200, 256, 36, 4)])
def test_sgemm_batched(M, N, K, batch):
    """lvsr_sgemm_batched on interleaved (time, utterance, feature) tensors, as the generator's backward uses it."""
    from lvsr_amd.native import ptr
    lib = emu_lib()
    rng = numpy.random.RandomState(3)
    A = torch.tensor(rng.normal(size=(K, batch, M)), dtype=torch.float32)        # problem b: A[:, b, :]^T  (M x K)
    Bm = torch.tensor(rng.normal(size=(K, batch, N)), dtype=torch.float32)       # problem b: B[:, b, :]    (K x N)
    C0 = torch.tensor(rng.normal(size=(M, batch, N)), dtype=torch.float32)
    C = C0.clone()
    lib.call("lvsr_sgemm_batched", lib.stream_for(C), 1, 0, M, N, K, 0.5, ptr(A), batch * M, M, ptr(Bm), batch * N, N, 1.0,
             ptr(C), batch * N, N, batch)
    ref = 0.5 * torch.einsum("kbm,kbn->mbn", A.double(), Bm.double()) + C0.double()
    assert_allclose(C.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_grouped_transposed_products_and_deferral():
    """lvsr_sgemm_tn_grouped: members of different shapes, strided operand views, one misaligned member (separate launch inside
    the call), beta = 1, with and without a workspace (k-chunk partials + fixed-order fold vs one chunk per tile) — and the
    host-side collection (`begin_group` / `sgemm(group=True)` / `flush_group`) incl. a dependent follow-up kept in order."""
    lib = emu_lib()
    rng = numpy.random.RandomState(4)
    t = lambda *s: torch.tensor(rng.normal(size=s), dtype=torch.float32)
    K1, K2 = 2100, 1100
    big = t(K1, 300)
    jobs = [(big[:, 4:4 + 130], t(K1, 70), torch.zeros(130, 70), 0.0),             # strided A view, aligned
            (t(K2, 40), t(K2, 260)[:, 8:8 + 132], t(40, 132), 1.0),                 # M < one tile, beta = 1
            (t(K2, 64), t(K2, 33), torch.zeros(64, 33), 0.0),                        # ldb = 33: the unaligned pass
            (t(50, 16), t(50, 16), torch.zeros(16, 16), 0.0)]                        # K below one chunk: no split
    for ws in (torch.empty(1 << 20), None):
        outs = [c.clone() for _, _, c, _ in jobs]
        if ws is None:
            cls = lib.structs["lvsr_gemm_desc"]
            arr = (cls * len(jobs))()
            for d, (A, B, _, beta), C in zip(arr, jobs, outs):
                d.A, d.B, d.C, d.M, d.N, d.K = A.data_ptr(), B.data_ptr(), C.data_ptr(), A.shape[1], B.shape[1], A.shape[0]
                d.lda, d.ldb, d.ldc, d.beta = A.stride(0), B.stride(0), C.stride(0), beta
            lib.call("lvsr_sgemm_tn_grouped", lib.stream_for(outs[0]), arr, len(jobs), None, 0)
        else:
            lib.begin_group()
            for (A, B, _, beta), C in zip(jobs, outs):
                lib.sgemm(A, B, C, transA=True, beta=beta, group=True)
            # a product that reads a collected product's output must run after the grouped launch: queued behind it
            dep = torch.zeros(16, 16)
            lib.sgemm(outs[3], outs[3], dep, transA=True, M=16, K=16, group=True)
            assert (outs[0] == 0).all(), "collected products must not run before flush_group"
            lib.flush_group(ws)
            assert_allclose(dep.numpy(), (outs[3].double().T @ outs[3].double()).numpy(), rtol=1e-4, atol=1e-4)
        for (A, B, C0, beta), C in zip(jobs, outs):
            ref = A.double().T @ B.double() + beta * C0.double()
            assert_allclose(C.numpy(), ref.numpy(), rtol=2e-4, atol=2e-4)


def test_copy_many():
    lib = emu_lib()
    rng = numpy.random.RandomState(5)
    src = torch.tensor(rng.normal(size=(37, 90)), dtype=torch.float32)
    dst = torch.zeros(40, 200)
    vec_s, vec_d = torch.arange(50, dtype=torch.float32), torch.zeros(64)
    pairs = [(src[:, 3:3 + 40], dst[:37, 100:140]), (src[:, 48:48 + 32], dst[:37, 4:36]), (vec_s, vec_d[7:57])]
    pairs += [(src[i:i + 1, :8], dst[38:39, 8 * i:8 * i + 8]) for i in range(20)]            # more than one launch's worth with the above
    pairs += [(src[:5, 60:70], dst[30:35, 180:190])] * 12
    lib.copy_many(pairs)
    for s_, d_ in pairs:
        assert (s_ == d_).all()
    assert float(dst[39].abs().sum()) == 0.0 and float(vec_d[:7].abs().sum()) == 0.0


def run_tile_shape_independence(device, lib):
    """What a row of a product rounds to must not depend on the tile shape its size selects (round 6: the 128 x 128 tiles read their
    operands from a [row][k] LDS image with permuted columns, the 64 x 64 tiles from the [k][row] image of rounds 1-5 — both walk k in
    ascending order, two per MFMA): the same product through 64 x 64 tiles (default for few tiles) and through 128 x 128 tiles
    (knob gemm_mid_tiles = 1: never the small tiles) is BIT-IDENTICAL, in all three layouts the step uses, with and without split-K."""
    rng = numpy.random.RandomState(7)
    for transA, transB, (M, N, K) in ((False, False, (384, 256, 96)), (False, True, (256, 384, 160)), (True, False, (256, 256, 1312))):
        A = torch.tensor(rng.normal(size=(K, M) if transA else (M, K)), dtype=torch.float32, device=device)
        B = torch.tensor(rng.normal(size=(N, K) if transB else (K, N)), dtype=torch.float32, device=device)
        ws = torch.empty(1 << 22, device=device) if transA else None
        out = {}
        for mid in (0, 1):
            lib.set_knob("gemm_mid_tiles", mid)
            C = torch.zeros(M, N, device=device)
            lib.sgemm(A, B, C, transA=transA, transB=transB, ws=ws)
            out[mid] = C.cpu().numpy().copy()
        lib.set_knob("gemm_mid_tiles", 0)
        ref = ((A.T if transA else A).double() @ (B.T if transB else B).double()).cpu().numpy()
        assert_allclose(out[0], ref, rtol=2e-5, atol=2e-4)
        assert (out[0] == out[1]).all(), (transA, transB, float(numpy.abs(out[0] - out[1]).max()))


def test_tile_shape_independence_emulated():
    run_tile_shape_independence("cpu", emu_lib())


def run_colsum_many(device, lib):
    """lvsr_colsum_many (the column sums of a backward pass in one launch at flush_group) == lvsr_colsum member by member, BIT for bit:
    row splits, a strided input, beta = 1 onto an existing output, a one-row input, 33 members (more than one launch of 32 without splits)."""
    rng = numpy.random.RandomState(11)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    shapes = [(1600, 768), (700, 33), (1, 5), (513, 64), (12, 512), (1600, 256)]
    xs = [t(rng.normal(size=s)) for s in shapes]
    xs[3] = t(rng.normal(size=(513, 100)))[:, 7:71]                    # strided rows
    ws = torch.empty(1 << 20, device=device)
    want = []
    for i, x in enumerate(xs):
        o = t(rng.normal(size=(x.shape[1],)))
        ref = o.clone()
        lib.colsum(x, ref, beta=1.0 if i == 1 else 0.0, ws=ws)
        want.append((o, ref))
    big = torch.empty(1 << 22, device=device)
    lib.begin_group()
    outs = []
    for i, x in enumerate(xs):
        o = want[i][0].clone()
        lib.colsum(x, o, beta=1.0 if i == 1 else 0.0, ws=ws)
        outs.append(o)
    assert all((o == w[0]).all() for o, w in zip(outs, want)), "a collected column sum must not run before the flush"
    lib.flush_group(big)
    for o, (_, ref) in zip(outs, want):
        assert (o.cpu().numpy() == ref.cpu().numpy()).all()
    assert_allclose(outs[0].cpu().numpy(), xs[0].double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    # more members than one launch holds (no row splits: small inputs)
    small = [t(rng.normal(size=(40, 17))) for _ in range(33)]
    lib.begin_group()
    so = [torch.zeros(17, device=device) for _ in small]
    for x, o in zip(small, so):
        lib.colsum(x, o)
    lib.flush_group(big)
    for x, o in zip(small, so):
        ref = torch.zeros(17, device=device)
        lib.colsum(x, ref)
        assert (o.cpu().numpy() == ref.cpu().numpy()).all()


def test_colsum_many_emulated():
    run_colsum_many("cpu", emu_lib())
