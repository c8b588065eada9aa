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
