"""
GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the
C ABI (sparch_amd._capi), against (a) the golden fixtures the real reference produced and
(b) the CPU oracle run live on the same seeded inputs.

Stated tolerances (fp32):
  * non-recurrent cells given identical Wx: spikes BIT-EXACT (same op order, no FMA);
  * recurrent cells with dyadic V (sums exact in fp32 in any order): spikes BIT-EXACT;
  * recurrent cells with real-valued V: the MFMA k-order differs from the CPU sgemm order,
    so a membrane potential within 1 ulp of the threshold may flip a spike and the
    trajectories then diverge: spike mismatch fraction <= 2e-3 on the fixtures;
  * gradients: max-abs error <= 2e-4 of the tensor's max-abs (+1e-6) when spikes agree;
  * GEMM: |err| <= 2e-6 * sum_k |a||b|  (exact fp32 fmaf chain vs fp64 reference).
"""
import numpy as np
import pytest
import torch

from oracle import snn_oracle as orc
from tests.golden_io import CELL_KINDS, DYADIC_CASES, DYADIC_LONG, SNN_CASES, layer_spikes, load, snn_case

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def sp():
    import sparch_amd
    return sparch_amd


def _Fn():
    from sparch_amd import functional
    return functional


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-6))


# ------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=["split6", "fp32"])
def dense_impl(request, monkeypatch):
    """Both dense GEMM implementations: exact 6-term bf16 split (default) and the fp32-input MFMA."""
    monkeypatch.setattr(_Fn(), "DENSE_GEMM", request.param)
    return request.param


@pytest.mark.parametrize("M,N,K", [(300, 70, 700), (257, 35, 1024), (128, 128, 32), (1000, 129, 41), (64, 1024, 700)])
def test_gemm_nt_bias_and_colstats(M, N, K, dense_impl):
    Fn = _Fn()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double().T + bias.double()
    bound = (A.abs().double() @ B.abs().double().T + bias.abs().double()) * 2e-6 + 1e-6
    C, ws = Fn.gemm_nt(A.to(DEV), B.to(DEV), bias.to(DEV), colstat=True)
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())
    nt = (M + 127) // 128
    ws = ws.cpu().double().view(2, nt, N)
    np.testing.assert_allclose(ws[0].sum(0).numpy(), ref.sum(0).numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(ws[1].sum(0).numpy(), (ref * ref).sum(0).numpy(), rtol=1e-4, atol=1e-2)
    C2, _ = Fn.gemm_nt(A.to(DEV), B.to(DEV))
    err2 = (C2.cpu().double() - (ref - bias.double())).abs()
    assert bool((err2 <= bound).all())


@pytest.mark.parametrize("M,N,K", [(300, 700, 128), (257, 1024, 35), (64, 44, 1000)])
def test_gemm_nn(M, N, K, dense_impl):
    Fn = _Fn()
    g = torch.Generator().manual_seed(7 + M)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    bound = (A.abs().double() @ B.abs().double()) * 2e-6 + 1e-6
    C = Fn.gemm_nn(A.to(DEV), B.to(DEV))
    assert bool(((C.cpu().double() - ref).abs() <= bound).all())


@pytest.mark.parametrize("M,N,K,zd", [(128, 700, 3000, False), (96, 96, 5000, True), (35, 1024, 2048, False),
                                      (1024, 1024, 4096, True)])
def test_gemm_tn_splitk_deterministic(M, N, K, zd, dense_impl):
    Fn = _Fn()
    g = torch.Generator().manual_seed(11 + K)
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double().T @ B.double()
    if zd:
        ref.fill_diagonal_(0)
    bound = (A.abs().double().T @ B.abs().double()) * 2e-6 + 1e-6
    Ad, Bd = A.to(DEV), B.to(DEV)
    C = Fn.gemm_tn(Ad, Bd, zero_diag=zd)
    assert bool(((C.cpu().double() - ref).abs() <= bound).all())
    C2 = Fn.gemm_tn(Ad, Bd, zero_diag=zd)
    assert torch.equal(C, C2), "split-K reduction must be bitwise reproducible"


@pytest.mark.parametrize("M,N,K", [(300, 70, 700), (257, 35, 1024), (1000, 1024, 96), (64, 129, 44),
                                   (1536, 256, 512), (520, 384, 1000)])
def test_gemm_spike_nt_exact_split(M, N, K):
    """A in {0, c}: C = c * (A != 0) @ B^T with B split exactly into three bf16 planes."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(M * 3 + N)
    c = 1.0 / 0.9
    A = (torch.rand(M, K, generator=g) < 0.1).float() * c
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double().T + bias.double()
    bound = (A.abs().double() @ B.abs().double().T + bias.abs().double()) * 2e-6 + 1e-6
    C, ws = Fn.gemm_nt(A.to(DEV), B.to(DEV), bias.to(DEV), colstat=True, spike_scale=c)
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())
    nt = (M + 127) // 128
    np.testing.assert_allclose(ws.cpu().double().view(2, nt, N)[0].sum(0).numpy(), ref.sum(0).numpy(), rtol=1e-4, atol=1e-2)
    # dyadic B: every partial sum is exact -> the result must equal the fp64 reference bit for bit
    Bd = torch.randint(-64, 65, (N, K), generator=g).float() / 128.0
    A1 = (A != 0).float()
    C1, _ = Fn.gemm_nt(A1.to(DEV), Bd.to(DEV), spike_scale=1.0)
    assert torch.equal(C1.cpu().double(), A1.double() @ Bd.double().T)
    # the same spikes given as a bf16 0/1 plane: identical products, so identical results bit for bit
    # (a plane with rows of K elements: 16-byte loadable only when K % 8 == 0, element-wise otherwise)
    a16 = (A != 0).to(torch.bfloat16).to(DEV)
    C16, ws16 = Fn.gemm_nt(A.to(DEV), B.to(DEV), bias.to(DEV), colstat=True, spike_scale=c, a16=a16)
    assert torch.equal(C16, C) and torch.equal(ws16, ws)


@pytest.mark.parametrize("M,N,K,side,zd", [(128, 700, 3000, 0, False), (96, 96, 5000, 0, True),
                                           (1024, 260, 2048, 1, False), (35, 1024, 4096, 1, False),
                                           (700, 35, 1111, 0, False), (512, 520, 8192, 0, True),
                                           (264, 512, 9000, 1, False)])
def test_gemm_spike_tn_exact_split(M, N, K, side, zd):
    Fn = _Fn()
    g = torch.Generator().manual_seed(K + side)
    c = 1.25
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g)
    if side == 0:
        A = (torch.rand(K, M, generator=g) < 0.1).float() * c
    else:
        B = (torch.rand(K, N, generator=g) < 0.1).float() * c
    ref = A.double().T @ B.double()
    if zd:
        ref.fill_diagonal_(0)
    bound = (A.abs().double().T @ B.abs().double()) * 2e-6 + 1e-6
    C = Fn.gemm_tn(A.to(DEV), B.to(DEV), zero_diag=zd, spike_side=side, spike_scale=c)
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())
    C2 = Fn.gemm_tn(A.to(DEV), B.to(DEV), zero_diag=zd, spike_side=side, spike_scale=c)
    assert torch.equal(C, C2)
    # the spike operand as a bf16 0/1 plane: bit-identical result
    A16 = (A != 0).to(torch.bfloat16).to(DEV) if side == 0 else A.to(DEV)
    B16 = (B != 0).to(torch.bfloat16).to(DEV) if side == 1 else B.to(DEV)
    C3 = Fn.gemm_tn(A16, B16, zero_diag=zd, spike_side=side, spike_scale=c, spike16=True)
    assert torch.equal(C, C3)
    # accumulate mode of the fp32 TN GEMM (used for the t = 0 term of dV)
    acc = C.clone()
    Fn.gemm_tn(A[:64].to(DEV), B[:64].to(DEV), zero_diag=zd, out=acc)
    ref2 = ref + (lambda r: (r.fill_diagonal_(0) if zd else r))(A[:64].double().T @ B[:64].double())
    assert bool(((acc.cpu().double() - ref2).abs() <= 2 * bound).all())


@pytest.fixture
def bf16_mode():
    """The bf16 operand mode (sparch_set_operand_precision) for one test; fp32 restored afterwards."""
    Fn = _Fn()
    prev = Fn.set_compute_dtype("bf16")
    yield Fn
    Fn.set_compute_dtype(prev)


def _rb(x):
    """x rounded to bf16 (nearest-even), as fp64: what the bf16 operand mode multiplies."""
    return x.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(300, 70, 700), (1000, 1024, 1024), (257, 35, 1024), (2048, 520, 1000),
                                   (64, 129, 44), (512, 256, 288)])
def test_bf16_operand_mode_gemms(M, N, K, bf16_mode):
    """Every GEMM entry point in the bf16 operand mode against fp64 products of the bf16-ROUNDED operands
    (one nearest-even rounding per operand element, fp32 accumulation: |err| <= 2e-6 sum|a||b|), on shapes
    that take the pipelined kernels and on shapes that take the general ones; spike operands lose nothing."""
    Fn = bf16_mode
    assert Fn.compute_dtype() == "bf16"
    g = torch.Generator().manual_seed(5 * M + N + K)
    c = 1.0 / 0.9
    S = (torch.rand(M, K, generator=g) < 0.1).float() * c          # spike operand (M,K)
    s16 = (S != 0).to(torch.bfloat16).to(DEV)
    A = torch.randn(M, K, generator=g)                               # dense operands
    W = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)

    def ok(C, ref, bound, what):
        err = (C.cpu().double() - ref).abs()
        assert bool((err <= bound).all()), (what, float((err / bound).max()))

    # NT: spikes x W^T (fp32 spikes, bf16 plane), dense x W^T, device-gated choice
    ref = S.double() @ _rb(W).T + bias.double()
    bound = (S.double() @ _rb(W).abs().T + bias.abs().double()) * 2e-6 + 1e-6
    C, ws = Fn.gemm_nt(S.to(DEV), W.to(DEV), bias.to(DEV), colstat=True, spike_scale=c)
    ok(C, ref, bound, "spike_nt")
    nt = (M + 127) // 128
    np.testing.assert_allclose(ws.cpu().double().view(2, nt, N)[0].sum(0).numpy(), ref.sum(0).numpy(), rtol=1e-4, atol=1e-2)
    C16, ws16 = Fn.gemm_nt(S.to(DEV), W.to(DEV), bias.to(DEV), colstat=True, spike_scale=c, a16=s16)
    assert torch.equal(C16, C) and torch.equal(ws16, ws)
    Cp, _ = Fn.gemm_nt(S.to(DEV), W.to(DEV), bias.to(DEV), spike_scale=c, a16=s16, b_planes=Fn.split_planes(W.to(DEV)))
    assert torch.equal(Cp, C), "the pre-split planes are not used in the bf16 mode: same kernel, same result"
    refd = _rb(A) @ _rb(W).T + bias.double()
    boundd = (_rb(A).abs() @ _rb(W).abs().T + bias.abs().double()) * 2e-6 + 1e-6
    Cd, _ = Fn.gemm_nt(A.to(DEV), W.to(DEV), bias.to(DEV))
    ok(Cd, refd, boundd, "dense_nt")
    for X, r, b in ((A, refd, boundd), ((S != 0).float(), (S != 0).double() @ _rb(W).T + bias.double(), bound)):
        Ca, _ = Fn.gemm_nt(X.to(DEV), W.to(DEV), bias.to(DEV), a_exact_flag=Fn.flag_bf16_exact(X.to(DEV)))
        ok(Ca, r, b, "auto_nt")
    # NN: dense (M,N') x W (N',K'): dx = G W
    G = torch.randn(M, N, generator=g)
    ok(Fn.gemm_nn(G.to(DEV), W.to(DEV)), _rb(G) @ _rb(W), (_rb(G).abs() @ _rb(W).abs()) * 2e-6 + 1e-6, "dense_nn")
    # TN over the long axis: spikes^T x dense, dense^T x spikes, dense^T x dense
    D = torch.randn(M, N, generator=g)
    St = (torch.rand(M, K, generator=g) < 0.1).float() * c
    reft = St.double().T @ _rb(D)
    boundt = (St.double().T @ _rb(D).abs()) * 2e-6 + 1e-6
    Ct = Fn.gemm_tn(St.to(DEV), D.to(DEV), spike_side=0, spike_scale=c)
    ok(Ct, reft, boundt, "spike_tn side 0")
    Ct16 = Fn.gemm_tn((St != 0).to(torch.bfloat16).to(DEV), D.to(DEV), spike_side=0, spike_scale=c, spike16=True)
    assert torch.equal(Ct16, Ct)
    ok(Fn.gemm_tn(D.to(DEV), St.to(DEV), spike_side=1, spike_scale=c), reft.T, boundt.T, "spike_tn side 1")
    ok(Fn.gemm_tn(A.to(DEV), D.to(DEV)), _rb(A).T @ _rb(D), (_rb(A).abs().T @ _rb(D).abs()) * 2e-6 + 1e-6, "dense_tn")
    # bf16-exact operands: the mode changes nothing — bit-identical to the exact-split kernels
    Wd = (torch.randint(-64, 65, (N, K), generator=g).float() / 128.0).to(DEV)
    C_b, _ = Fn.gemm_nt(S.to(DEV), Wd, spike_scale=c, a16=s16)
    Fn.set_compute_dtype("fp32")
    C_f, _ = Fn.gemm_nt(S.to(DEV), Wd, spike_scale=c, a16=s16)
    Fn.set_compute_dtype("bf16")
    assert torch.equal(C_b, C_f)


@pytest.mark.parametrize("M,N,K", [(1000, 1024, 1024), (640, 256, 512), (300, 136, 96), (512, 1024, 700),
                                   (257, 70, 64), (2048, 520, 1000)])
def test_gemm_presplit_weight_planes_are_bit_identical(M, N, K):
    """sparch_split3 + the _wp GEMMs (weights split into bf16 planes once) against the kernels that convert
    the fp32 weights on the fly: the planes sum back to W exactly and both products match bit for bit — on
    shapes the pipelined plane kernel takes and on shapes where the _wp entries fall back."""
    Fn = _Fn()
    if not Fn.USE_PRESPLIT:
        pytest.skip("SPARCH_PRESPLIT=0")
    g = torch.Generator().manual_seed(M + 7 * N + K)
    W = (torch.randn(N, K, generator=g) * torch.exp(4 * torch.randn(N, K, generator=g))).to(DEV)
    planes = Fn.split_planes(W)
    if K % 8:  # plane rows must be 16-byte loadable: no planes, the layers then convert on the fly
        assert planes is None
        return
    assert planes.shape == (3, N, K)
    p = planes.float()
    assert torch.equal((p[0] + p[1]) + p[2], W) and torch.equal(p[0].view(torch.int32), W.view(torch.int32) & -65536)
    a16 = (torch.rand(M, K, generator=g) < 0.15).to(torch.bfloat16).to(DEV)
    A = a16.float()
    bias = torch.randn(N, generator=g).to(DEV)
    for colstat, b in ((True, bias), (False, bias), (False, None)):
        C0, ws0 = Fn.gemm_nt(A, W, b, colstat=colstat, spike_scale=1.0, a16=a16)
        C1, ws1 = Fn.gemm_nt(A, W, b, colstat=colstat, spike_scale=1.0, a16=a16, b_planes=planes)
        assert torch.equal(C0, C1)
        assert ws0 is None or torch.equal(ws0, ws1)
    G = torch.randn(M, N, generator=g).to(DEV)  # dx = G W: W is the (K_gemm = N) x (N_gemm = K) operand
    D0 = Fn.gemm_nn(G, W)
    D1 = Fn.gemm_nn(G, W, b_planes=planes)
    assert torch.equal(D0, D1)
    # (heavy-tailed W: a few products dominate each sum, so the fp32 accumulation error sits nearer its
    # worst case than on the gaussian operands of the tests above — bound 1e-5 of sum|a||b|)
    ratio = (D1.double() - G.double() @ W.double()).abs() / (G.abs().double() @ W.abs().double() + 1e-6)
    assert float(ratio.max()) <= 1e-5, float(ratio.max())


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("M,N,K", [(1000, 260, 700), (512, 1024, 40), (300, 64, 33), (2048, 256, 704)])
def test_gemm_auto_with_the_input_plane(exact, M, N, K):
    """sparch_plane_bf16_exact + the _auto16_ GEMMs: the check writes the bf16 plane of the network input in the
    pass that computes the flag; with integer counts (flag 1) the GEMMs read the plane and must give the SAME bits
    as the fp32-operand kernels, with real-valued input (flag 0) they take the six-term kernels as before."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(M + N + K + int(exact))
    A = torch.poisson(torch.full((M, K), 0.2), generator=g) if exact else torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    G = torch.randn(M, N, generator=g)
    bias = torch.randn(N, generator=g)
    Ad, Wd, Gd, bd = A.to(DEV), W.to(DEV), G.to(DEV), bias.to(DEV)
    plane, flag = Fn.plane_bf16_exact(Ad)
    assert int(flag[0].item()) == (1 if exact else 0)
    assert plane.shape == (M, (K + 7) // 8 * 8)
    if exact:
        assert torch.equal(plane[:, :K].float(), Ad) and float(plane[:, K:].abs().sum()) == 0.0
    flag0 = Fn.flag_bf16_exact(Ad)
    assert torch.equal(flag0[:1], flag[:1])
    C0, ws0 = Fn.gemm_nt(Ad, Wd, bd, colstat=(M % 256 == 0), a_exact_flag=flag0)
    C1, ws1 = Fn.gemm_nt(Ad, Wd, bd, colstat=(M % 256 == 0), a_exact_flag=flag, a_plane=plane)
    assert torch.equal(C0, C1) and (ws0 is None or torch.equal(ws0, ws1))
    ref = A.double() @ W.double().T + bias.double()
    bound = (A.abs().double() @ W.abs().double().T + bias.abs().double()) * 2e-6 + 1e-6
    assert bool(((C1.cpu().double() - ref).abs() <= bound).all())
    D0 = Fn.gemm_tn(Gd, Ad, b_exact_flag=flag0)             # dW = G^T A: (N, K)
    D1 = Fn.gemm_tn(Gd, Ad, b_exact_flag=flag, b_plane=plane)
    assert torch.equal(D0, D1)
    refd = G.double().T @ A.double()
    assert bool(((D1.cpu().double() - refd).abs() <= (G.abs().double().T @ A.abs().double()) * 2e-6 + 1e-6).all())


@pytest.mark.parametrize("exact", [True, False])
def test_gemm_auto_device_gated_paths(exact):
    """Network-input GEMMs: the bf16-exactness flag is computed on the device and gates which of the two
    enqueued kernels runs.  Integer spike counts take the single-plane path, real-valued input the 6-term one;
    both must meet the fp32 GEMM bound, and the flag must be right."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(31)
    M, N, K = 1000, 260, 700
    if exact:
        A = torch.poisson(torch.full((M, K), 0.2), generator=g)   # counts 0,1,2,...
    else:
        A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    G = torch.randn(M, N, generator=g)
    Ad, Bd, Gd = A.to(DEV), B.to(DEV), G.to(DEV)
    flag = Fn.flag_bf16_exact(Ad)
    assert int(flag[0].item()) == (1 if exact else 0)
    C, _ = Fn.gemm_nt(Ad, Bd, a_exact_flag=flag)
    ref = A.double() @ B.double().T
    bound = (A.abs().double() @ B.abs().double().T) * 2e-6 + 1e-6
    assert bool(((C.cpu().double() - ref).abs() <= bound).all())
    dW = Fn.gemm_tn(Gd, Ad, b_exact_flag=flag)                      # (N,K) = G^T A
    ref2 = G.double().T @ A.double()
    bound2 = (G.abs().double().T @ A.abs().double()) * 2e-6 + 1e-6
    assert bool(((dW.cpu().double() - ref2).abs() <= bound2).all())


# ------------------------------------------------------------------------------------ cells
def _cell_inputs(z, kind):
    p = {k: dev(z[k]).requires_grad_(True) for k in ("alpha", "beta", "a", "b", "V") if k in z}
    Wx = dev(z["Wx"]).requires_grad_(True)
    u0, s0 = dev(z["u0"]), dev(z["s0"])
    w0 = dev(z["w0"]) if "w0" in z else None
    return Wx, p, u0, w0, s0


@pytest.mark.parametrize("spl", [1, None])
@pytest.mark.parametrize("kind", CELL_KINDS)
def test_cell_forward_backward_vs_reference_golden(kind, spl, record_property):
    Fn = _Fn()
    if spl == 1 and kind in ("LIF", "adLIF"):
        pytest.skip("steps_per_launch only applies to recurrent kinds")
    z = load(f"cell_{kind}")
    Wx, p, u0, w0, s0 = _cell_inputs(z, kind)
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wx, p["alpha"], p.get("beta"), p.get("a"), p.get("b"), p.get("V"),
                               u0, w0, s0, spl)
    Fn.check_status()
    s_np = s.detach().cpu().numpy()
    mism = float((s_np != z["s"]).mean())
    record_property("spike_mismatch_fraction", mism)
    print(f"cell_{kind} spl={spl}: spike mismatch fraction vs the reference fixture = {mism:.3e}")
    if kind in ("LIF", "adLIF"):
        assert mism == 0.0, "non-recurrent cell spikes must be bit-identical to the reference"
    else:
        assert mism <= 2e-3, mism
    (s * dev(z["g_s"])).sum().backward()
    Fn.check_status()
    # rows whose whole trajectory agrees with the reference: their dWx rows must agree too (a row's dWx
    # depends on that row's trajectory only); parameter gradients sum over all rows, so they are compared
    # when every row agrees (the fully dyadic network fixtures pin them for the recurrent kinds otherwise)
    same = (s_np == z["s"]).reshape(s_np.shape[0], -1).all(1)
    record_property("rows_agreeing", int(same.sum()))
    assert same.sum() >= 1
    assert relmax(Wx.grad.cpu().numpy()[same], z["dWx"][same]) <= 2e-4
    if mism == 0.0:
        for k in p:
            assert relmax(p[k].grad.cpu().numpy(), z["d" + k]) <= 2e-4, k
    # clamp gating (raw parameter outside its range -> exactly zero grad)
    assert p["alpha"].grad[0].item() == 0 and p["alpha"].grad[1].item() == 0
    if "V" in p:
        assert float(torch.diag(p["V"].grad).abs().max()) == 0.0


@pytest.mark.parametrize("spl", [1, 7, None])
@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
@pytest.mark.parametrize("Bp,T,H", [(5, 33, 64), (40, 21, 132), (64, 50, 256)])
def test_recurrent_cell_bit_exact_with_dyadic_V(kind, spl, Bp, T, H):
    """V on a 2^-6 grid and binary s0: every partial sum of s@V is exact in fp32, so the result
    cannot depend on summation order and the spikes must equal the oracle's bit for bit."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(H + T)
    V = torch.randint(-24, 25, (H, H), generator=g).float() / 64.0
    Wx = torch.randn(Bp, T, H, generator=g) * 1.5 + 0.4
    p = {"alpha": torch.rand(H, generator=g) * 0.2 + 0.78, "V": V}
    if kind == "RadLIF":
        p.update(beta=torch.rand(H, generator=g) * 0.05 + 0.95, a=torch.rand(H, generator=g) * 2.4 - 1.2,
                 b=torch.rand(H, generator=g) * 2.4 - 0.2)
    u0 = torch.rand(Bp, H, generator=g)
    w0 = torch.rand(Bp, H, generator=g) if kind == "RadLIF" else None
    s0 = (torch.rand(Bp, H, generator=g) < 0.3).float()
    with torch.no_grad():
        ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
    pd = {k: v.to(DEV) for k, v in p.items()}
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wx.to(DEV), pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"),
                               pd["V"], u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), spl)
    Fn.check_status()
    assert ref.sum() > 0
    assert torch.equal(s.cpu(), ref), float((s.cpu() != ref).float().mean())


def _dyadic_cell_case(kind, Bp, T, H, seed):
    g = torch.Generator().manual_seed(seed)
    V = torch.randint(-24, 25, (H, H), generator=g).float() / 64.0
    Wx = torch.randn(Bp, T, H, generator=g) * 1.5 + 0.4
    p = {"alpha": torch.rand(H, generator=g) * 0.2 + 0.78, "V": V}
    if kind == "RadLIF":
        p.update(beta=torch.rand(H, generator=g) * 0.05 + 0.95, a=torch.rand(H, generator=g) * 2.4 - 1.2,
                 b=torch.rand(H, generator=g) * 2.4 - 0.2)
    u0 = torch.rand(Bp, H, generator=g)
    w0 = torch.rand(Bp, H, generator=g) if kind == "RadLIF" else None
    s0 = (torch.rand(Bp, H, generator=g) < 0.3).float()
    g_s = torch.randn(Bp, T, H, generator=g)
    return Wx, p, u0, w0, s0, g_s


def _run_cell(kind, Wx, p, u0, w0, s0, g_s):
    Fn = _Fn()
    pd = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
    Wxd = Wx.to(DEV).requires_grad_(True)
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), pd["V"],
                               u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), None)
    (s * g_s.to(DEV)).sum().backward()
    Fn.check_status()
    return s.detach().cpu(), Wxd.grad.cpu(), {k: v.grad.cpu() for k, v in pd.items()}


@pytest.mark.parametrize("kind,Bp,T,H", [("adLIF", 6, 40, 64), ("LIF", 9, 33, 128), ("RadLIF", 40, 50, 256),
                                         ("RLIF", 5, 33, 64)])
def test_bf16_saved_states_keep_every_discrete_decision(kind, Bp, T, H, monkeypatch):
    """SPARCH_SAVE_DTYPE=bf16 (BASELINE configs[4] is the bf16 long-sequence case): u and w are kept in bf16 for
    the backward pass, rounded so that the spike and box-car decisions stay exactly the fp32 ones.  Stated
    bars against the fp32-saved path on the same inputs: forward spikes identical (the saves are not read by
    the forward); dWx and dV IDENTICAL bit for bit (they depend on the saved states only through those
    decisions — flip rate 0); dalpha / dbeta / da / db within 2e-2 of max-abs (2^-9 relative rounding of the
    u / w factors; measured 1e-3 .. 5e-3)."""
    Fn = _Fn()
    if kind in ("RLIF", "RadLIF"):
        case = _dyadic_cell_case(kind, Bp, T, H, 31)
    else:
        Wx, p, u0, w0, s0, g_s = _dyadic_cell_case("RadLIF" if kind == "adLIF" else "RLIF", Bp, T, H, 31)
        p = {k: v for k, v in p.items() if k != "V"}
        case = (Wx, p, u0, w0, s0, g_s)

    def run():
        Wx, p, u0, w0, s0, g_s = case
        pd = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
        Wxd = Wx.to(DEV).requires_grad_(True)
        s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), pd.get("V"),
                                   u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), None)
        (s * g_s.to(DEV)).sum().backward()
        Fn.check_status()
        return s.detach().cpu(), Wxd.grad.cpu(), {k: v.grad.cpu() for k, v in pd.items()}

    s_a, dwx_a, g_a = run()
    monkeypatch.setattr(Fn, "SAVE_BF16", True)
    s_b, dwx_b, g_b = run()
    assert s_a.sum() > 0 and torch.equal(s_a, s_b)
    assert torch.equal(dwx_a, dwx_b), float((dwx_a - dwx_b).abs().max())
    worst = 0.0
    for k in g_a:
        if k == "V":
            assert torch.equal(g_a[k], g_b[k])
        else:
            e = relmax(g_b[k].numpy(), g_a[k].numpy())
            worst = max(worst, e)
            assert e <= 2e-2, (k, e)
    print(f"bf16 saved states, {kind}: worst neuron-parameter gradient deviation {worst:.2e} of max-abs")


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
def test_recurrent_step_path_equals_persistent_kernels(kind, compute, monkeypatch, request):
    """The step path (one launch per time step, recurrent product between the steps on the exact split
    GEMMs: what hidden sizes above 1024 use) against the persistent kernels at a size both handle, dyadic V:
    identical spikes, dWx and gradients to fp32 rounding (the two differ only in summation order) — also in
    the bf16 operand mode, where both paths multiply the same rounded dWx and V."""
    if compute == "bf16":
        request.getfixturevalue("bf16_mode")
    case = _dyadic_cell_case(kind, 40, 21, 132, 5)
    s_a, dwx_a, g_a = _run_cell(kind, *case)
    monkeypatch.setenv("SPARCH_REC_STEP_PATH", "1")
    s_b, dwx_b, g_b = _run_cell(kind, *case)
    assert s_a.sum() > 0 and torch.equal(s_a, s_b)
    assert relmax(dwx_b.numpy(), dwx_a.numpy()) <= 1e-5
    for k in g_a:
        assert relmax(g_b[k].numpy(), g_a[k].numpy()) <= 1e-5, k


@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
def test_recurrent_cell_hidden_size_above_1024_vs_oracle(kind):
    """The reference accepts any nb_hiddens (snns.py:608-661).  H = 1536 (V slice no longer register-resident:
    step path), dyadic V: spikes bit-equal to the oracle, gradients (oracle autograd) to 2e-4 of max-abs."""
    Wx, p, u0, w0, s0, g_s = _dyadic_cell_case(kind, 5, 12, 1536, 9)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Wxr = Wx.clone().requires_grad_(True)
    ref = orc.spiking_cell(kind, Wxr, pr, u0, w0, s0)
    (ref * g_s).sum().backward()
    s, dwx, grads = _run_cell(kind, Wx, p, u0, w0, s0, g_s)
    assert ref.sum() > 0 and torch.equal(s, ref.detach())
    assert relmax(dwx.numpy(), Wxr.grad.numpy()) <= 2e-4
    for k in grads:
        assert relmax(grads[k].numpy(), pr[k].grad.numpy()) <= 2e-4, k
    assert float(torch.diag(grads["V"]).abs().max()) == 0.0


def test_snn_hidden_size_2048_and_200_classes_runs(sp):
    """--nb_hiddens 2048 (used to fail with 'invalid argument') and a 200-class readout: one
    training step of RadLIF [2048, 2048, 200]; invariants of the softmax-sum readout and finite gradients."""
    B, T, C = 8, 10, 40
    torch.manual_seed(3)
    net = sp.SNN((B, None, C), [2048, 2048, 200], neuron_type="RadLIF", dropout=0.1).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, 200, (B,), generator=g).to(DEV)
    out, rates = net(x)
    torch.nn.functional.cross_entropy(out, y).backward()
    _Fn().check_status()
    np.testing.assert_allclose(out.detach().sum(1).cpu().numpy(), np.full(B, T, np.float32), rtol=1e-4)
    assert tuple(rates.shape) == (4096,) and float(rates.detach().mean()) > 0
    for k, v in net.named_parameters():
        assert bool(torch.isfinite(v.grad).all()), k
    assert float(net.snn[0].V.weight.grad.abs().max()) > 0


@pytest.mark.parametrize("C", [100, 200])
def test_readout_cell_more_than_64_classes_vs_oracle(C):
    Fn = _Fn()
    B, T = 6, 150
    g = torch.Generator().manual_seed(C)
    Wx = torch.randn(B, T, C, generator=g) * 2.0
    alpha = torch.rand(C, generator=g) * 0.2 + 0.78
    u0 = torch.rand(B, C, generator=g)
    g_out = torch.randn(B, C, generator=g)
    Wr, ar = Wx.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref = orc.readout_cell(Wr, ar, u0)
    (ref * g_out).sum().backward()
    Wd, ad = Wx.to(DEV).requires_grad_(True), alpha.to(DEV).requires_grad_(True)
    out = Fn.ReadoutCellFn.apply(Wd, ad, u0.to(DEV))
    (out * g_out.to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    assert relmax(Wd.grad.cpu().numpy(), Wr.grad.numpy()) <= 2e-4
    assert relmax(ad.grad.cpu().numpy(), ar.grad.numpy()) <= 2e-4


@pytest.mark.parametrize("kind,B,T,H", [("adLIF", 5, 17, 64), ("LIF", 33, 9, 128), ("RadLIF", 5, 33, 64),
                                        ("RLIF", 40, 21, 132), ("RadLIF", 48, 40, 1024)])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_bidirectional_cell_addressing_bit_exact(kind, B, T, H, p_drop):
    """dirs=2: virtual rows b' >= B read Wx time-flipped and write their spikes un-flipped into features
    [H, 2H) (snns.py:252-254, 272-275), never materialising the flipped copy.  With dyadic V the spikes must
    equal the oracle's explicit flip/cat/chunk/flip/cat bit for bit; with dropout every kept spike is
    scaled and the kept pattern is a subset; the backward must match autograd through the oracle."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(B * 7 + H)
    adaptive, recurrent = kind in ("adLIF", "RadLIF"), kind in ("RLIF", "RadLIF")
    Wx = (torch.randn(B, T, H, generator=g) * 1.5 + 0.4).requires_grad_(True)
    p = {"alpha": (torch.rand(H, generator=g) * 0.2 + 0.78).requires_grad_(True)}
    if adaptive:
        p.update(beta=(torch.rand(H, generator=g) * 0.05 + 0.95).requires_grad_(True),
                 a=(torch.rand(H, generator=g) * 2.4 - 1.2).requires_grad_(True),
                 b=(torch.rand(H, generator=g) * 2.4 - 0.2).requires_grad_(True))
    if recurrent:
        p["V"] = (torch.randint(-24, 25, (H, H), generator=g).float() / 64.0).requires_grad_(True)
    u0 = torch.rand(2 * B, H, generator=g)
    w0 = torch.rand(2 * B, H, generator=g) if adaptive else None
    s0 = (torch.rand(2 * B, H, generator=g) < 0.3).float()
    gs = torch.randn(B, T, 2 * H, generator=g)
    # oracle: explicit glue around the cell
    Wcat = torch.cat([Wx, Wx.flip(1)], dim=0)
    s_cat = orc.spiking_cell(kind, Wcat, p, u0, w0, s0)
    s_f, s_b = s_cat.chunk(2, dim=0)
    ref = torch.cat([s_f, s_b.flip(1)], dim=2)
    (ref * gs).sum().backward()
    pd = {k: v.detach().to(DEV) for k, v in p.items()}
    Wxd = Wx.detach().to(DEV)
    seed = 1234567
    s_out, count, saved, s16 = Fn.cell_forward(kind, Wxd, None, None, pd, u0.to(DEV), None if w0 is None else w0.to(DEV),
                                          s0.to(DEV), B=B, dirs=2, theta=1.0, p_drop=p_drop, seed=seed)
    Fn.check_status()
    out = s_out.cpu()
    assert torch.equal(s16.float().cpu(), (out != 0).float())  # the bf16 plane is exactly (s_out != 0)
    if p_drop == 0.0:
        assert torch.equal(out, ref.detach())
        dWx, pg = Fn.cell_backward(kind, gs.to(DEV), None, pd, u0.to(DEV), None if w0 is None else w0.to(DEV),
                                   s0.to(DEV), saved, B=B, dirs=2, T=T, H=H, theta=1.0, p_drop=0.0, seed=seed)
        Fn.check_status()
        dsum = (dWx[:B] + dWx[B:]).cpu().numpy()          # both directions share the projection rows
        assert relmax(dsum, Wx.grad.numpy()) <= 2e-4
        for k in p:
            assert relmax(pg[k].cpu().numpy(), p[k].grad.numpy()) <= 2e-4, k
    else:
        keep = 1.0 / (1.0 - p_drop)
        fired = ref.detach() > 0
        assert bool(((out == 0) | ((out - keep).abs() < 1e-6)).all())
        assert not bool((out > 0)[~fired].any())
        assert abs(float((out > 0)[fired].float().mean()) - (1 - p_drop)) < 0.03
    np.testing.assert_array_equal(count.cpu().numpy(), (out > 0).sum(dim=(0, 1)).numpy())


@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
def test_recurrent_backward_vs_oracle_autograd(kind):
    """Backward on a trajectory the HIP forward and the oracle agree on exactly (dyadic V)."""
    Fn = _Fn()
    Bp, T, H = 37, 29, 96
    g = torch.Generator().manual_seed(5)
    V = (torch.randint(-24, 25, (H, H), generator=g).float() / 64.0).requires_grad_(True)
    Wx = (torch.randn(Bp, T, H, generator=g) * 1.5 + 0.4).requires_grad_(True)
    p = {"alpha": (torch.rand(H, generator=g) * 0.2 + 0.78).requires_grad_(True), "V": V}
    if kind == "RadLIF":
        p.update(beta=(torch.rand(H, generator=g) * 0.05 + 0.95).requires_grad_(True),
                 a=(torch.rand(H, generator=g) * 2.4 - 1.2).requires_grad_(True),
                 b=(torch.rand(H, generator=g) * 2.4 - 0.2).requires_grad_(True))
    u0 = torch.rand(Bp, H, generator=g)
    w0 = torch.rand(Bp, H, generator=g) if kind == "RadLIF" else None
    s0 = (torch.rand(Bp, H, generator=g) < 0.3).float()
    gs = torch.randn(Bp, T, H, generator=g)
    ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
    (ref * gs).sum().backward()
    for spl in (1, 5, None):
        pd = {k: v.detach().to(DEV).requires_grad_(True) for k, v in p.items()}
        Wxd = Wx.detach().to(DEV).requires_grad_(True)
        s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), pd["V"],
                                   u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), spl)
        assert torch.equal(s.detach().cpu(), ref.detach())
        (s * gs.to(DEV)).sum().backward()
        Fn.check_status()
        assert relmax(Wxd.grad.cpu().numpy(), Wx.grad.numpy()) <= 2e-4, spl
        for k in p:
            assert relmax(pd[k].grad.cpu().numpy(), p[k].grad.numpy()) <= 2e-4, (k, spl)
        assert float(torch.diag(pd["V"].grad).abs().max()) == 0.0


@pytest.mark.parametrize("kind,H", [("RLIF", 66), ("RadLIF", 130), ("RadLIF", 7), ("RLIF", 1030)])
def test_recurrent_cell_any_hidden_size(kind, H):
    """The reference takes any nb_hiddens (snns.py:608-661); the recurrent kernels own 4 columns per thread, so a
    width that is not a multiple of 4 runs zero-padded (functional.cell_forward): dyadic V, spikes bit-equal to the
    oracle, every gradient (oracle autograd) to 2e-4 of its max-abs, shapes as the caller's."""
    Wx, p, u0, w0, s0, gs = _dyadic_cell_case(kind, 9, 14, H, 40 + H)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    Wx.requires_grad_(True)
    ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
    (ref * gs).sum().backward()
    s, dwx, g = _run_cell(kind, Wx.detach(), {k: v.detach() for k, v in p.items()}, u0, w0, s0, gs)
    assert tuple(s.shape) == tuple(ref.shape) and ref.sum() > 0
    assert torch.equal(s, ref.detach())
    assert bool(torch.isfinite(Wx.grad).all()), "oracle gradient not finite"
    assert bool(torch.isfinite(dwx).all()), ("HIP dWx not finite", torch.isnan(dwx).nonzero()[:8].tolist())
    assert relmax(dwx.numpy(), Wx.grad.numpy()) <= 2e-4
    for k in p:
        assert tuple(g[k].shape) == tuple(p[k].shape)
        assert relmax(g[k].numpy(), p[k].grad.numpy()) <= 2e-4, k
    assert float(torch.diag(g["V"]).abs().max()) == 0.0


def test_snn_with_hidden_sizes_not_multiples_of_four(sp):
    """Whole networks at widths the kernels do not take natively (130 recurrent units, 30 LIF units, 20 classes):
    the HIP path against the CPU oracle on a dyadic network — per-neuron spike counts equal, loss and every
    parameter gradient to fp32 rounding; bidirectional, so that the per-direction slicing of the padded outputs is
    exercised."""
    B, T, C = 6, 20, 44
    for kind, sizes, bidir in (("RadLIF", [130, 66, 20], True), ("LIF", [30, 30, 20], False)):
        torch.manual_seed(11)
        net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.0, normalization="none", bidirectional=bidir)
        with torch.no_grad():
            for lay in net.snn:
                lay.W.weight.copy_(torch.round(lay.W.weight * 4 * 64) / 64)
                if hasattr(lay, "V"):
                    lay.V.weight.copy_(torch.round(lay.V.weight * 64) / 64)
        params = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(5)
        x = (torch.rand(B, T, C, generator=g) < 0.3).float()
        y = torch.randint(0, sizes[-1], (B,), generator=g)
        torch.manual_seed(7)
        init = orc.draw_init_states(B, sizes, kind, bidirectional=bidir)
        init = [{k: torch.floor(v * 16) / 16 for k, v in st.items()} for st in init]
        order = iter([st[k] for st in init for k in ("u0", "w0", "s0") if k in st])
        from sparch_amd import snns as snn_mod
        old = snn_mod._rand_to
        snn_mod._rand_to = lambda rows, cols, device: next(order).to(device)
        try:
            net = net.to(DEV).train()
            out, rates = net(x.to(DEV))
            loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
            loss.backward()
            _Fn().check_status()
        finally:
            snn_mod._rand_to = old
        po = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in params.items()}
        out_o, rates_o = orc.snn_forward(x, po, neuron_type=kind, num_layers=3, init_states=init,
                                         normalization="none", training=True, stats={}, bidirectional=bidir)
        loss_o = torch.nn.functional.cross_entropy(out_o, y)
        loss_o.backward()
        n = B * T
        assert float(rates_o.sum()) > 0
        assert torch.equal(torch.round(rates.detach().cpu() * n).long(), torch.round(rates_o.detach() * n).long()), kind
        assert abs(float(loss.detach()) - float(loss_o.detach())) <= 1e-5 * max(1.0, abs(float(loss_o.detach())))
        for k, v in net.named_parameters():
            assert relmax(v.grad.cpu().numpy(), po[k].grad.numpy()) <= 2e-4, (kind, k)


@pytest.mark.parametrize("kind,Bp,T,H,spl", [("RadLIF", 288, 4, 1024, None), ("RLIF", 520, 3, 1024, None),
                                             ("RadLIF", 300, 5, 512, None), ("RadLIF", 288, 4, 1024, 2),
                                             ("RLIF", 260, 3, 992, None), ("RadLIF", 70, 6, 1000, 3),
                                             ("RLIF", 1030, 2, 96, None), ("RadLIF", 97, 7, 96, 3)])
def test_recurrent_cell_many_row_tiles_and_chunked_launches(kind, Bp, T, H, spl):
    """More row tiles than one persistent launch holds (9 and 17 row tiles of 32 column tiles on 256 CUs: two and
    three launch groups), an odd number of workgroups, partial last tiles, chunked launches: dyadic V, spikes
    bit-equal to the oracle, dWx and every parameter gradient to 2e-4 of max-abs."""
    Wx, p, u0, w0, s0, gs = _dyadic_cell_case(kind, Bp, T, H, 3 * H + Bp)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    Wx.requires_grad_(True)
    ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
    (ref * gs).sum().backward()
    Fn = _Fn()
    pd = {k: v.detach().to(DEV).requires_grad_(True) for k, v in p.items()}
    Wxd = Wx.detach().to(DEV).requires_grad_(True)
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), pd["V"],
                               u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), spl)
    (s * gs.to(DEV)).sum().backward()
    Fn.check_status()
    assert ref.sum() > 0 and torch.equal(s.detach().cpu(), ref.detach())
    assert relmax(Wxd.grad.cpu().numpy(), Wx.grad.numpy()) <= 2e-4
    for k in p:
        assert relmax(pd[k].grad.cpu().numpy(), p[k].grad.numpy()) <= 2e-4, k


_HEADLINE_ORACLE = {}


def _headline_oracle(kind, Bp):
    """Oracle forward + autograd backward of the dyadic-V cell case at the headline launch geometry (cached per
    test session: it is the expensive half, 134 GF forward per 256 rows)."""
    key = (kind, Bp)
    if key not in _HEADLINE_ORACLE:
        Wx, p, u0, w0, s0, gs = _dyadic_cell_case(kind, Bp, 250, 1024, 7 * Bp + len(kind))
        p = {k: v.requires_grad_(True) for k, v in p.items()}
        Wx.requires_grad_(True)
        ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
        (ref * gs).sum().backward()
        _HEADLINE_ORACLE[key] = (Wx.detach(), {k: v.detach() for k, v in p.items()}, u0, w0, s0, gs, ref.detach(),
                                 Wx.grad, {k: v.grad for k, v in p.items()})
    return _HEADLINE_ORACLE[key]


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("kind,Bp", [("RadLIF", 256), ("RLIF", 256), ("RadLIF", 512)])
def test_recurrent_cell_headline_launch_geometry_vs_oracle(kind, Bp, compute, request):
    """The EXACT launch of BASELINE configs[2] / configs[4] against the oracle, not through properties: (Bp, T, H) =
    (256, 250, 1024) — 8 row tiles x 32 column tiles, one workgroup per CU, whole-sequence launch, XCD-local
    hand-off stores, the depth-4 sentinel ring wrapping 62 times, the backward's bulk stores one step late — and
    (512, 250, 1024), the 16 row tiles of a bidirectional B = 256 layer.  Dyadic V (every partial sum of s @ V
    exact in fp32 in any order): spikes torch.equal to oracle.spiking_cell over all 250 steps, so a stale or torn
    hand-off tile anywhere in the sequence flips spikes; dWx and every parameter gradient within 2e-4 of max-abs
    (fp32) / 2e-2 (bf16 operand mode, one rounding of dWx per product; V itself is bf16-exact here, so the
    forward is bit-exact in that mode too).  Reference: snns.py:554-578, 696-727."""
    Fn = _Fn()
    if compute == "bf16":
        request.getfixturevalue("bf16_mode")
    Wx, p, u0, w0, s0, gs, ref, dwx_ref, g_ref = _headline_oracle(kind, Bp)
    s, dwx, g = _run_cell(kind, Wx, p, u0, w0, s0, gs)
    assert ref.sum() > 0
    assert torch.equal(s, ref), float((s != ref).float().mean())
    tol = 2e-4 if compute == "fp32" else 2e-2
    assert bool(torch.isfinite(dwx_ref).all()) and bool(torch.isfinite(dwx).all())
    assert relmax(dwx.numpy(), dwx_ref.numpy()) <= tol
    for k in g_ref:
        assert relmax(g[k].numpy(), g_ref[k].numpy()) <= tol, k
    assert float(torch.diag(g["V"]).abs().max()) == 0.0


def test_cell_kernels_shape_fuzz_vs_oracle():
    """All four cells over a grid of awkward shapes — 1 / 2 / 31 / 33 / 65 rows, 1 / 2 / 5 steps, 1 / 3 / 4 / 5 / 31 /
    33 / 63 / 96 / 100 units, whole-sequence and one-step launches — one after the other in one process (so that what
    a launch leaves behind meets the next shape's buffers): dyadic V, spikes bit-equal to the oracle, dWx and every
    parameter gradient to 2e-4 of max-abs."""
    import itertools
    Fn = _Fn()
    bad = []
    for kind in ("RadLIF", "RLIF", "adLIF", "LIF"):
        rec = kind in ("RLIF", "RadLIF")
        adapt = kind in ("adLIF", "RadLIF")
        for Bp, T, H in itertools.product((1, 2, 31, 33, 65), (1, 2, 5), (1, 3, 4, 5, 31, 33, 63, 96, 100)):
            if (Bp + T + H) % 2 and not rec:   # (half of the non-recurrent grid: it has no hand-off machinery)
                continue
            for spl in ((None, 1) if rec else (None,)):
                Wx, p, u0, w0, s0, gs = _dyadic_cell_case("RadLIF" if adapt else "RLIF", Bp, T, H, 7 * H + Bp + T)
                if not rec:
                    p.pop("V")
                p = {k: v.requires_grad_(True) for k, v in p.items()}
                Wx.requires_grad_(True)
                ref = orc.spiking_cell(kind, Wx, p, u0, w0 if adapt else None, s0)
                (ref * gs).sum().backward()
                pd = {k: v.detach().to(DEV).requires_grad_(True) for k, v in p.items()}
                Wxd = Wx.detach().to(DEV).requires_grad_(True)
                s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"),
                                           pd.get("V"), u0.to(DEV), w0.to(DEV) if adapt else None, s0.to(DEV), spl)
                (s * gs.to(DEV)).sum().backward()
                e = relmax(Wxd.grad.cpu().numpy(), Wx.grad.numpy())
                eg = max(relmax(pd[k].grad.cpu().numpy(), p[k].grad.numpy()) for k in p)
                if not torch.equal(s.detach().cpu(), ref.detach()) or not e <= 2e-4 or not eg <= 2e-4:
                    bad.append((kind, Bp, T, H, spl, e, eg))
    Fn.check_status()
    assert not bad, bad[:10]


@pytest.mark.parametrize("save16", [False, True])
def test_scan_kernels_around_the_prefetch_ring_depths(save16, monkeypatch):
    """LIF / adLIF scan kernels (round 3: a register ring of the next D steps' inputs, D = 8 ... 16 by variant —
    csrc/cell.hip): sequence lengths on both sides of every ring depth and of its multiples (the loop's main trips,
    the tail without refills, cell step 0 inside the tail), one neuron per thread and — 512 x 1024 neurons — four per
    thread; fp32 and bf16 saved states.  Spikes bit-equal to the oracle, dWx to 2e-4 of max-abs in both modes, the
    neuron-parameter gradients to 2e-4 (fp32 saves) / 5e-2 (bf16 saves); evaluation forwards (no saved states:
    another kernel variant) give the same spikes."""
    Fn = _Fn()
    monkeypatch.setattr(Fn, "SAVE_BF16", save16)
    bad = []
    cases = [(3, T, 5) for T in (6, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 49)]
    cases += [(512, T, 1024) for T in (7, 8, 9, 17)]  # 524288 neurons: the four-neurons-per-thread variant (D = 8)
    for kind in ("adLIF", "LIF"):
        adapt = kind == "adLIF"
        for Bp, T, H in cases:
            Wx, p, u0, w0, s0, gs = _dyadic_cell_case("RadLIF" if adapt else "RLIF", Bp, T, H, 11 * T + Bp)
            p.pop("V")
            p = {k: v.requires_grad_(True) for k, v in p.items()}
            Wx.requires_grad_(True)
            ref = orc.spiking_cell(kind, Wx, p, u0, w0 if adapt else None, s0)
            (ref * gs).sum().backward()
            pd = {k: v.detach().to(DEV).requires_grad_(True) for k, v in p.items()}
            Wxd = Wx.detach().to(DEV).requires_grad_(True)
            s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), None,
                                       u0.to(DEV), w0.to(DEV) if adapt else None, s0.to(DEV), None)
            (s * gs.to(DEV)).sum().backward()
            e = relmax(Wxd.grad.cpu().numpy(), Wx.grad.numpy())
            eg = max(relmax(pd[k].grad.cpu().numpy(), p[k].grad.numpy()) for k in p)
            # bf16 saves: dWx depends on the saved states only through the discrete decisions (same bar); the neuron
            # parameters' sums see u / w rounded to 2^-9 — 5e-2 of max-abs here, where 7-9 steps leave little to average
            tol_p = 5e-2 if save16 else 2e-4
            if not torch.equal(s.detach().cpu(), ref.detach()) or not e <= 2e-4 or not eg <= tol_p:
                bad.append((kind, Bp, T, H, e, eg))
            # evaluation mode: no saved states (another kernel variant), same spikes
            with torch.no_grad():
                s_eval = Fn.SpikingCellFn.apply(kind, 1.0, Wxd.detach(), pd["alpha"].detach(),
                                                None if not adapt else pd["beta"].detach(),
                                                None if not adapt else pd["a"].detach(),
                                                None if not adapt else pd["b"].detach(), None, u0.to(DEV),
                                                w0.to(DEV) if adapt else None, s0.to(DEV), None)
            if not torch.equal(s_eval.cpu(), ref.detach()):
                bad.append((kind, Bp, T, H, "eval"))
    Fn.check_status()
    assert not bad, bad[:10]


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_snn_random_configurations_vs_oracle(sp, compute, request):
    """(fp32: the bars below; bf16 operand mode: spike counts still equal — the dyadic weights are bf16-exact — and
    gradients to 2e-2.)  Forty whole networks drawn at random — LIF / adLIF / RLIF / RadLIF, 1-33 rows, 1-19 steps, 1-70 input channels,
    hidden widths 1-100 (most not multiples of 4), 1-130 classes, with and without bias, bidirectional, readout —
    on dyadic weights against the CPU oracle: per-neuron spike counts equal, outputs to 1e-4, every parameter
    gradient to 5e-4 of its max-abs."""
    from sparch_amd import snns as snn_mod
    if compute == "bf16":
        request.getfixturevalue("bf16_mode")
    tol_g = 5e-4 if compute == "fp32" else 2e-2
    rng = np.random.default_rng(5)
    bad = []
    for it in range(40):
        kind = ["LIF", "adLIF", "RLIF", "RadLIF"][it % 4]
        B, T, C = int(rng.choice([1, 2, 5, 17, 33])), int(rng.choice([1, 2, 7, 19])), int(rng.choice([1, 3, 20, 33, 70]))
        nl = int(rng.choice([2, 3]))
        sizes = [int(rng.choice([1, 3, 5, 12, 30, 33, 64, 100])) for _ in range(nl - 1)] + [int(rng.choice([1, 3, 20, 35, 130]))]
        bidir, bias, readout = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(4) > 0)
        torch.manual_seed(100 + it)
        net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.0, normalization="none", bidirectional=bidir,
                     use_bias=bias, use_readout_layer=readout)
        with torch.no_grad():
            for lay in net.snn:
                lay.W.weight.copy_(torch.round(lay.W.weight * 4 * 64) / 64)
                if lay.W.bias is not None:
                    lay.W.bias.copy_(torch.round(lay.W.bias * 64) / 64)
                if hasattr(lay, "V"):
                    lay.V.weight.copy_(torch.round(lay.V.weight * 64) / 64)
        params = {k: v.clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(it)
        x = (torch.rand(B, T, C, generator=g) < 0.4).float()
        torch.manual_seed(7 + it)
        init = orc.draw_init_states(B, sizes, kind, bidirectional=bidir, use_readout_layer=readout)
        init = [{k: torch.floor(v * 16) / 16 for k, v in st.items()} for st in init]
        order = iter([st[k] for st in init for k in ("u0", "w0", "s0") if k in st])
        old = snn_mod._rand_to
        snn_mod._rand_to = lambda rows, cols, device, order=order: next(order).to(device)
        try:
            net = net.to(DEV).train()
            out, rates = net(x.to(DEV))
            gout = torch.randn(out.shape, generator=g)
            (out * gout.to(DEV)).sum().backward()
            _Fn().check_status()
        finally:
            snn_mod._rand_to = old
        po = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in params.items()}
        out_o, rates_o = orc.snn_forward(x, po, neuron_type=kind, num_layers=nl, init_states=init, normalization="none",
                                         training=True, stats={}, bidirectional=bidir, use_readout_layer=readout)
        (out_o * gout).sum().backward()
        same = torch.equal(torch.round(rates.detach().cpu() * B * T).long(), torch.round(rates_o.detach() * B * T).long())
        eo = relmax(out.detach().cpu().numpy(), out_o.detach().numpy())
        # (bf16 mode: the neuron parameters' gradients are sums over (b, t) that largely cancel — one of them, 3
        # entries of size 1e-3 beside weight gradients of 1e-1, moves by 6 % of ITS max-abs under the 2^-9
        # rounding of dWx: five times the bar of the matrices)
        eg = max(relmax(v.grad.cpu().numpy(), po[k].grad.numpy()) /
                 (5.0 if compute == "bf16" and k.split(".")[-1] in ("alpha", "beta", "a", "b") else 1.0)
                 for k, v in net.named_parameters())
        if not same or not eo <= 1e-4 or not eg <= tol_g:
            bad.append((kind, (B, T, C), sizes, bidir, bias, readout, same, eo, eg))
    for b_ in bad:
        print("random configuration off:", b_)
    assert not bad, bad[:5]


@pytest.mark.parametrize("spl", [1, None])
@pytest.mark.parametrize("kind,Bp,T,H", [("RLIF", 48, 60, 128), ("RadLIF", 96, 80, 256), ("RadLIF", 33, 40, 1024)])
def test_recurrent_one_step_ahead_vs_oracle_trajectory(kind, Bp, T, H, spl):
    """Real-valued orthogonal V (the reference's init).  The MFMA's k-order differs from a CPU
    sgemm's, so `s@V` differs in the last bits; the reference's own subthreshold (u,w) map is
    expansive for a -> -1 (eigenvalue ~1.4 per step), so free-running trajectories separate
    exponentially whatever the implementation.  The rigorous check is therefore teacher-forced:
    every (sample, t) of a long ORACLE trajectory becomes an independent 2-step problem started
    from the oracle's exact state (u,w,s)_{t-1}.  Step 0 exercises the rec0 GEMM path, step 1
    the in-kernel MFMA + spike hand-off.  A spike may differ from the oracle's only where the
    oracle's own membrane potential is within 1e-4 of the threshold."""
    from oracle import bptt_numpy as bp
    Fn = _Fn()
    g = torch.Generator().manual_seed(17 + H)
    V = torch.nn.init.orthogonal_(torch.empty(H, H), generator=g)
    Wx = torch.randn(Bp, T, H, generator=g) * 1.5 + 0.5
    p = {"alpha": torch.rand(H, generator=g) * 0.14 + 0.82, "V": V}
    if kind == "RadLIF":
        p.update(beta=torch.rand(H, generator=g) * 0.024 + 0.967, a=torch.rand(H, generator=g) * 2 - 1,
                 b=torch.rand(H, generator=g) * 2)
    u0, s0 = torch.rand(Bp, H, generator=g), torch.rand(Bp, H, generator=g)
    w0 = torch.rand(Bp, H, generator=g) if kind == "RadLIF" else None
    pn = {k: v.numpy() for k, v in p.items()}
    S, U, W = bp.cell_forward(kind, Wx.numpy(), pn, u0.numpy(), None if w0 is None else w0.numpy(), s0.numpy())
    assert S.mean() > 0.003
    # pseudo-samples: state after step t-1, inputs of steps t and t+1, for t = 1 .. T-2
    ts = np.arange(1, T - 1)
    u_in = torch.from_numpy(U[:, ts - 1].reshape(-1, H).copy())
    s_in = torch.from_numpy(S[:, ts - 1].reshape(-1, H).copy())
    w_in = torch.from_numpy(W[:, ts - 1].reshape(-1, H).copy()) if W is not None else None
    Wx2 = torch.from_numpy(np.stack([Wx.numpy()[:, ts], Wx.numpy()[:, ts + 1]], axis=2).reshape(-1, 2, H).copy())
    ref_s = np.stack([S[:, ts], S[:, ts + 1]], axis=2).reshape(-1, 2, H)
    ref_u = np.stack([U[:, ts], U[:, ts + 1]], axis=2).reshape(-1, 2, H)
    pd = {k: v.to(DEV) for k, v in p.items()}
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wx2.to(DEV), pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"),
                               pd["V"], u_in.to(DEV), None if w_in is None else w_in.to(DEV), s_in.to(DEV), spl)
    Fn.check_status()
    s = s.cpu().numpy()
    diff = s != ref_s
    n0, n1 = int(diff[:, 0].sum()), int(diff[:, 1].sum())
    if diff[:, 0].any():
        assert np.abs(ref_u[:, 0][diff[:, 0]] - 1.0).max() <= 1e-4
    # step 1 inherits step 0's (legitimate) flips: only judge rows whose step 0 agreed
    ok_rows = ~diff[:, 0].any(axis=1)
    d1 = diff[:, 1] & ok_rows[:, None]
    if d1.any():
        assert np.abs(ref_u[:, 1][d1] - 1.0).max() <= 1e-4
    print(f"{kind} H={H}: {n0}+{n1} near-threshold flips in {ref_s.size} spikes")
    assert (n0 + n1) <= 1e-4 * ref_s.size + 2


@pytest.mark.parametrize("M,H", [(7, 64), (1001, 1024), (4099, 128), (32000, 1024), (5, 30)])
def test_bn_backward_apply_in_place_any_row_count(M, H):
    """sparch_bn_bwd_apply (dx written over dy, four rows per trip + a ragged tail, scalar kernel for
    H % 4 != 0) against the batch-norm backward formula in fp64."""
    from sparch_amd._capi import check, lib, ptr
    g = torch.Generator().manual_seed(M + H)
    dy, x = torch.randn(M, H, generator=g), torch.randn(M, H, generator=g) * 2 + 0.5
    gamma = torch.rand(H, generator=g) + 0.5
    mean, var = x.double().mean(0), x.double().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    xh = (x.double() - mean) * invstd
    dbeta, dgamma = dy.double().sum(0), (dy.double() * xh).sum(0)
    ref = gamma.double() * invstd * (dy.double() - dbeta / M - xh * dgamma / M)
    d = dy.to(DEV)
    args = [t.float().to(DEV) for t in (x, mean, invstd, gamma, dgamma, dbeta)]
    check(lib.sparch_bn_bwd_apply(M, H, ptr(d), *[ptr(t) for t in args], ptr(d), None), "sparch_bn_bwd_apply")
    assert relmax(d.cpu().numpy(), ref.numpy()) <= 2e-5


@pytest.mark.parametrize("norm", ["batchnorm", "layernorm"])
def test_projection_and_normalisation_vs_oracle(norm):
    """G1+G2 alone (no spikes, no chaos): x@W^T (+bias) -> BatchNorm/LayerNorm vs torch CPU fp32."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(2)
    M, K, H = 3000, 700, 260
    x = (torch.rand(M, K, generator=g) < 0.05).float()
    W = torch.randn(H, K, generator=g) * 0.1
    bias = torch.randn(H, generator=g) * 0.1
    gamma, beta = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.1
    rm, rv = torch.zeros(H), torch.ones(H)
    lin = torch.nn.functional.linear(x, W, bias)
    if norm == "batchnorm":
        ref = torch.nn.functional.batch_norm(lin, rm, rv, gamma, beta, True, 0.05, 1e-5)
    else:
        ref = torch.nn.functional.layer_norm(lin, (H,), gamma, beta, 1e-5)
    Wx_raw, ws = Fn.gemm_nt(x.to(DEV), W.to(DEV), bias.to(DEV), colstat=(norm == "batchnorm"))
    rmd, rvd = torch.zeros(H, device=DEV), torch.ones(H, device=DEV)
    y, scale, shift, _ = Fn._Norm.forward(norm, Wx_raw, ws, gamma.to(DEV), beta.to(DEV), rmd, rvd, True, 1)
    if scale is not None:
        y = y * scale + shift
        np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_readout_cell_vs_reference_golden():
    Fn = _Fn()
    z = load("cell_readout")
    Wx = dev(z["Wx"]).requires_grad_(True)
    alpha = dev(z["alpha"]).requires_grad_(True)
    out = Fn.ReadoutCellFn.apply(Wx, alpha, dev(z["u0"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=2e-5, atol=2e-5)
    (out * dev(z["g_out"])).sum().backward()
    assert relmax(Wx.grad.cpu().numpy(), z["dWx"]) <= 2e-4
    assert relmax(alpha.grad.cpu().numpy(), z["dalpha"]) <= 2e-4
    assert alpha.grad[0].item() == 0 and alpha.grad[1].item() == 0


# ------------------------------------------------------------------------------------ whole SNN
def _build(sp, cfg, params):
    net = sp.SNN((cfg["B"], None, cfg["C"]), cfg["layer_sizes"], neuron_type=cfg["neuron_type"], dropout=0.0,
                 normalization=cfg["normalization"], use_bias=cfg["use_bias"], bidirectional=cfg["bidirectional"],
                 use_readout_layer=cfg["use_readout_layer"])
    missing = net.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.to(DEV)


@pytest.mark.parametrize("name", SNN_CASES)
def test_snn_train_step_vs_reference_golden(sp, name):
    """Same parameters, same input, same torch.manual_seed as the reference run:
    out / firing_rates / loss / all parameter gradients / BN running stats / eval-mode output."""
    Fn = _Fn()
    cfg, x, y, params, init, z = snn_case(name)
    net = _build(sp, cfg, params)
    net.train()
    torch.manual_seed(cfg["fwd_seed"])  # the layers draw u0,[w0],s0 from the CPU generator like the reference
    out, rates = net(x.to(DEV))
    Fn.check_status()
    if cfg["use_readout_layer"]:
        loss = orc.train_step_loss(out, rates, y.to(DEV), use_regularizers=cfg["use_regularizers"])
    else:
        loss = (out * out).mean()
    loss.backward()
    Fn.check_status()
    T = cfg["T"]
    out_np, rates_np = out.detach().cpu().numpy(), rates.detach().cpu().numpy()
    recurrent = cfg["neuron_type"] in ("RLIF", "RadLIF")
    if not recurrent:
        # A threshold-rounding flip stays local to one neuron (reset only): tight bounds.
        if cfg["use_readout_layer"]:
            # softmax-sum over T steps: entries are O(T/classes)
            assert np.abs(out_np - z["out"]).max() <= 2e-3 * T, np.abs(out_np - z["out"]).max()
        else:
            assert float((out_np != z["out"]).mean()) <= 2e-3
        assert np.abs(rates_np - z["rates"]).max() <= 2.5 / (cfg["B"] * T) + 1e-6
        assert abs(float(loss.detach()) - float(z["loss"])) <= 2e-3 * max(1.0, abs(float(z["loss"])))
        for k, v in net.named_parameters():
            e = relmax(v.grad.cpu().numpy(), z["grad." + k])
            assert e <= 5e-2, (k, e)  # loose cap: a flipped spike perturbs downstream grads
    else:
        # Through V one flipped spike reaches every neuron at the next step and the
        # trajectories separate (the recurrence is chaotic), so per-sample closeness is not
        # a meaningful bar; exactness of the recurrent kernels is established by the
        # dyadic-V bit-exact tests and test_recurrent_first_divergence_is_at_threshold.
        # Here: population statistics of the same network on the same input.
        assert np.abs(rates_np - z["rates"]).mean() <= 0.02, np.abs(rates_np - z["rates"]).mean()
        assert np.abs(out_np - z["out"]).mean() <= 0.03 * T / cfg["layer_sizes"][-1] + 0.02
        assert abs(float(loss.detach()) - float(z["loss"])) <= 0.05 * max(1.0, abs(float(z["loss"])))
        for k, v in net.named_parameters():
            assert bool(torch.isfinite(v.grad).all()), k
    for k, v in net.state_dict().items():
        if "running" in k:  # layers fed by a recurrent layer see a (legitimately) different spike train
            tol = 5e-3 if (recurrent and not k.startswith("snn.0.")) else 1e-5
            np.testing.assert_allclose(v.cpu().numpy(), z["after." + k], rtol=1e-4, atol=tol, err_msg=k)
    net.eval()
    with torch.no_grad():
        torch.manual_seed(cfg["fwd_seed"])
        out_e, rates_e = net(x.to(DEV))
    if not recurrent:
        if cfg["use_readout_layer"]:
            assert np.abs(out_e.cpu().numpy() - z["out_eval"]).max() <= 2e-3 * T
        assert np.abs(rates_e.cpu().numpy() - z["rates_eval"]).max() <= 2.5 / (cfg["B"] * T) + 1e-6
    else:
        assert np.abs(rates_e.cpu().numpy() - z["rates_eval"]).mean() <= 0.02


def _run_dyadic(sp, name, monkeypatch):
    """One training step of a dyadic fixture's network on the HIP path, fed the fixture's initial states in
    the reference's draw order.  Returns (cfg, z, per-layer spikes, out, loss, net)."""
    Fn = _Fn()
    cfg, x, y, params, init, z = snn_case(name)
    net = _build(sp, cfg, params).train()
    order = []
    for st in init:
        order += [st[k] for k in ("u0", "w0", "s0") if k in st]
    states = iter(order)

    def next_state(rows, cols, device):
        t = next(states)
        assert tuple(t.shape) == (rows, cols)
        return t.to(device)

    from sparch_amd import snns as snn_mod
    monkeypatch.setattr(snn_mod, "_rand_to", next_state)
    rec = {}
    for k, lay in enumerate(list(net.snn)[:-1]):
        def wrapped(inp, states=None, orig=lay.forward_with_rate, k=k, **kw):
            s, r = orig(inp, states=states, **kw)
            rec[k] = snn_mod.materialize_spikes(s).detach()  # between the network's layers: the bf16 plane
            return s, r
        lay.forward_with_rate = wrapped
    out, rates = net(x.to(DEV))
    Fn.check_status()
    loss = orc.train_step_loss(out, rates, y.to(DEV))
    loss.backward()
    Fn.check_status()
    return cfg, z, rec, out, loss, net


@pytest.mark.parametrize("name", DYADIC_CASES)
def test_snn_dyadic_network_bit_equal_spikes_and_gradients(sp, name, monkeypatch):
    """Whole RLIF / RadLIF networks against the REFERENCE on fully dyadic weights (tools/gen_golden.py
    gen_snn_dyadic): every W x, s @ V and s0 @ V sum is exact in any order, so the MFMA path must reproduce
    the reference's spikes BIT FOR BIT through every layer (guaranteed by construction without
    normalisation; with batchnorm the last bit of the batch variance is order-dependent, `a` >= 0 keeps
    the map contracting and equality is observed), and then every parameter gradient — V.weight, layer-0 W,
    the neuron parameters, norm affine — to 2e-4 of its max-abs."""
    cfg, z, rec, out, loss, net = _run_dyadic(sp, name, monkeypatch)
    for k in sorted(rec):
        ref = layer_spikes(z, k)
        got = rec[k].cpu().numpy()
        assert ref.sum() > 0
        assert np.array_equal(got, ref), (k, float((got != ref).mean()))
    T = cfg["T"]
    assert np.abs(out.detach().cpu().numpy() - z["out"]).max() <= 2e-5 * T
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    for k, v in net.named_parameters():
        e = relmax(v.grad.cpu().numpy(), z["grad." + k])
        assert e <= 2e-4, (k, e)
        if k.endswith("V.weight"):
            assert float(torch.diag(v.grad).abs().max()) == 0.0


@pytest.mark.parametrize("name", DYADIC_CASES)
def test_bf16_operand_mode_dyadic_network(sp, name, monkeypatch, bf16_mode):
    """The bf16 operand mode (BASELINE configs[4]; the reference is fp32-only, SURVEY §7-6: compared with the fp32
    reference under a stated tolerance) on the fully dyadic fixtures.  Their W / V (multiples of 2^-6, |k| < 256)
    are bf16-representable and spikes are 0 / 1, so the rounding of the mode changes NOTHING in the forward pass:
    spikes bit-equal to the reference's through every layer, output and loss as in fp32.  The backward pass
    rounds dWx once per matrix product (2^-9 relative per operand element, fp32 accumulation): every parameter
    gradient within 2e-2 of its max-abs of the reference's."""
    assert bf16_mode.compute_dtype() == "bf16"
    cfg, z, rec, out, loss, net = _run_dyadic(sp, name, monkeypatch)
    for k in sorted(rec):
        ref = layer_spikes(z, k)
        got = rec[k].cpu().numpy()
        assert np.array_equal(got, ref), (k, float((got != ref).mean()))
    assert np.abs(out.detach().cpu().numpy() - z["out"]).max() <= 2e-5 * cfg["T"]
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    worst = 0.0
    for k, v in net.named_parameters():
        e = relmax(v.grad.cpu().numpy(), z["grad." + k])
        worst = max(worst, e)
        assert e <= 2e-2, (k, e)
        if k.endswith("V.weight"):
            assert float(torch.diag(v.grad).abs().max()) == 0.0
    print(f"{name}: worst gradient relmax in the bf16 operand mode {worst:.2e}")


@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
@pytest.mark.parametrize("spl", [1, None])
def test_bf16_operand_mode_recurrent_cell(kind, spl, bf16_mode):
    """The recurrent cell kernels of the bf16 operand mode (whole-sequence persistent launch and one launch per
    step).  With a dyadic (bf16-exact) V the forward is the fp32 one bit for bit; the backward's hand-off tiles
    carry bf16(dWx) (one nearest-even rounding per element per step), so its gradients are compared with oracle
    autograd at 2e-2 of max-abs; dV's diagonal stays exactly zero."""
    Fn = bf16_mode
    Wx, p, u0, w0, s0, gs = _dyadic_cell_case(kind, 40, 29, 132, 5)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    Wx.requires_grad_(True)
    ref = orc.spiking_cell(kind, Wx, p, u0, w0, s0)
    (ref * gs).sum().backward()
    pd = {k: v.detach().to(DEV).requires_grad_(True) for k, v in p.items()}
    Wxd = Wx.detach().to(DEV).requires_grad_(True)
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wxd, pd["alpha"], pd.get("beta"), pd.get("a"), pd.get("b"), pd["V"],
                               u0.to(DEV), None if w0 is None else w0.to(DEV), s0.to(DEV), spl)
    assert torch.equal(s.detach().cpu(), ref.detach())
    (s * gs.to(DEV)).sum().backward()
    Fn.check_status()
    e = relmax(Wxd.grad.cpu().numpy(), Wx.grad.numpy())
    assert 0.0 < e <= 2e-2, e  # (0 would mean the exact kernels ran)
    for k in p:
        assert relmax(pd[k].grad.cpu().numpy(), p[k].grad.numpy()) <= 2e-2, k
    assert float(torch.diag(pd["V"].grad).abs().max()) == 0.0


def test_snn_long_sequence_non_finite_gradient_mask_matches_reference(sp, monkeypatch):
    """T = 1000 (BASELINE configs[4] length) on a dyadic RadLIF net: neurons whose subthreshold (u, w) map is
    unstable overflow fp32 in the reference itself and leave some of ITS alpha / beta / a gradients
    non-finite.  The fixture holds the reference's gradients; the HIP path must leave exactly the same
    entries non-finite, parameter by parameter, and agree on the finite ones."""
    cfg, z, rec, out, loss, net = _run_dyadic(sp, DYADIC_LONG, monkeypatch)
    for k in sorted(rec):
        assert np.array_equal(rec[k].cpu().numpy(), layer_spikes(z, k)), k
    n_bad = 0
    for k, v in net.named_parameters():
        g, g_ref = v.grad.cpu().numpy(), z["grad." + k]
        ok = np.isfinite(g_ref)
        assert np.array_equal(np.isfinite(g), ok), (k, int((~np.isfinite(g)).sum()), int((~ok).sum()))
        n_bad += int((~ok).sum())
        if ok.any():
            assert relmax(g[ok], g_ref[ok]) <= 2e-4, k
    assert n_bad > 0


@pytest.mark.parametrize("kind,bidir", [("RadLIF", False), ("adLIF", True), ("RLIF", True)])
def test_inner_layers_without_fp32_spikes_give_identical_results(sp, kind, bidir):
    """Between two layers of an SNN the spikes travel as a bf16 plane only; the fp32 tensor of the reference
    (snns.py:278 -> 261 of the next layer) is not written (its autograd edge is a placeholder).  A forward hook on
    a layer makes its output observable again, so the same network with hooks takes the fp32-writing path: both
    must give identical outputs, firing rates and parameter gradients, and the hook must see exactly the plane's
    spikes times the dropout scale."""
    Fn = _Fn()
    from sparch_amd import snns as snn_mod
    B, T, C, sizes = 12, 30, 40, [64, 96, 10]
    torch.manual_seed(3)
    net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.25, bidirectional=bidir).to(DEV).train()
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)

    def run(with_hooks):
        seen, hooks = {}, []
        if with_hooks:
            for k, lay in enumerate(list(net.snn)[:-1]):
                hooks.append(lay.register_forward_hook(lambda m, i, o, k=k: seen.__setitem__(k, o.detach().clone())))
        for lay in net.snn:
            lay._calls = 0  # same dropout masks in both runs
        net.zero_grad()
        torch.manual_seed(77)
        out, rates = net(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        Fn.check_status()
        for h in hooks:
            h.remove()
        return out.detach().clone(), rates.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters()}, seen

    planes = {}
    for k, lay in enumerate(list(net.snn)[:-1]):
        def wrapped(inp, orig=lay.forward_with_rate, k=k, **kw):
            s, r = orig(inp, **kw)
            planes[k] = (bool(s._sparch_spike_tag[4]), snn_mod.materialize_spikes(s).detach().clone())
            return s, r
        lay.forward_with_rate = wrapped
    out_a, rates_a, g_a, _ = run(False)
    assert all(planes[k][0] for k in planes), "inner layers should have returned placeholders"
    planes_a = {k: v[1] for k, v in planes.items()}
    out_b, rates_b, g_b, seen = run(True)
    assert not any(planes[k][0] for k in planes), "hooked layers must write their fp32 output"
    assert torch.equal(out_a, out_b) and torch.equal(rates_a, rates_b) and float(rates_a.sum()) > 0
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k
    for k in seen:
        assert torch.equal(seen[k], planes_a[k]), k


def test_inplace_edit_between_layers_invalidates_the_spike_fast_path(sp):
    """the spike fast path (bf16 plane + scale handed from layer to layer) is keyed on the tensor's
    version counter: after an in-place edit of a layer's output the next layer must treat it as an ordinary
    real-valued input, not read the stale plane."""
    cfg, x, y, params, init, z = snn_case("snn_adLIF_bn")
    net = _build(sp, cfg, params).eval()
    from sparch_amd import snns as snn_mod
    with torch.no_grad():
        torch.manual_seed(1)
        s0, _ = net.snn[0].forward_with_rate(x.to(DEV))
        assert snn_mod._spike_tag(s0)[0] == 1.0 and snn_mod._spike_tag(s0)[1] is not None
        s0.mul_(0.5)
        s0[:, ::2, :] = 0.25
        assert snn_mod._spike_tag(s0) == (None, None)
        torch.manual_seed(2)
        s1, _ = net.snn[1].forward_with_rate(s0)
        # oracle: layer 1 on the edited tensor
        torch.manual_seed(2)
        st = {"u0": torch.rand(cfg["B"], cfg["layer_sizes"][1]), "w0": torch.rand(cfg["B"], cfg["layer_sizes"][1]),
              "s0": torch.rand(cfg["B"], cfg["layer_sizes"][1])}
        ref = orc.hidden_layer("adLIF", s0.cpu(), params, "snn.1.", st, normalization="batchnorm", training=False)
    assert float((s1.cpu() != ref).float().mean()) <= 2e-3
    assert float(ref.sum()) > 0


def test_eval_mode_batchnorm_gradients_vs_oracle(sp):
    """net.eval() with autograd on (fixed running statistics): gradients of a non-recurrent net must match
    the oracle's autograd through F.batch_norm(training=False)."""
    cfg, x, y, params, init, z = snn_case("snn_adLIF_bn")
    net = _build(sp, cfg, params).eval()
    torch.manual_seed(cfg["fwd_seed"])
    out, rates = net(x.to(DEV))
    loss = orc.train_step_loss(out, rates, y.to(DEV), use_regularizers=True)
    loss.backward()
    p = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in params.items()}
    out_o, rates_o = orc.snn_forward(x, p, neuron_type=cfg["neuron_type"], num_layers=len(cfg["layer_sizes"]),
                                     init_states=init, normalization="batchnorm", training=False)
    orc.train_step_loss(out_o, rates_o, y, use_regularizers=True).backward()
    assert np.abs(out.detach().cpu().numpy() - out_o.detach().numpy()).max() <= 2e-3 * cfg["T"]
    for k, v in net.named_parameters():
        assert relmax(v.grad.cpu().numpy(), p[k].grad.numpy()) <= 5e-2, k


def test_reference_written_checkpoint_runs_on_hip(sp):
    """f-1: a checkpoint pickled by the reference loads through the shim and its eval forward on the HIP
    path reproduces the reference's eval output (RadLIF: population statistics)."""
    import os
    from tests.golden_io import GOLDEN
    net = torch.load(os.path.join(GOLDEN, "ref_checkpoint_RadLIF.pth"), weights_only=False).to(DEV)
    z = load("ref_checkpoint_RadLIF_io")
    x = torch.from_numpy(z["x"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        torch.manual_seed(2)
        out, rates = net(x)
    _Fn().check_status()
    assert np.abs(rates.cpu().numpy() - z["rates_eval"]).mean() <= 0.02
    assert np.abs(out.cpu().numpy() - z["out_eval"]).mean() <= 0.03 * 30 / 20 + 0.02
    np.testing.assert_allclose(out.sum(1).cpu().numpy(), 30.0, rtol=1e-4)
    net.train()                       # and it trains: dropout seed bookkeeping absent from the pickle
    out, rates = net(x)
    out.sum().backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def test_bidirectional_recurrent_multi_launch_full_width(sp):
    """B' = 2*B = 512 virtual rows at H = 1024: 16 batch tiles, i.e. two persistent launches of 8 tiles x 32
    workgroups per layer and direction-straddling state; invariants + determinism + linearity of the backward."""
    Fn = _Fn()
    torch.manual_seed(7)
    B, T, C = 256, 40, 96
    net = sp.SNN((B, None, C), [1024, 1024, 35], neuron_type="RadLIF", bidirectional=True).to(DEV).train()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(B, T, C, generator=g) < 0.1).float().to(DEV)
    y = torch.randint(0, 35, (B,), generator=g).to(DEV)

    def step(scale):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        out, rates = net(x)
        (torch.nn.functional.cross_entropy(out, y) * scale).backward()
        Fn.check_status()
        return out.detach(), rates.detach(), {k: v.grad.clone() for k, v in net.named_parameters()}

    o1, r1, g1 = step(1.0)
    assert r1.numel() == 2 * 2048 and float(r1.min()) >= 0 and float(r1.max()) <= 1 and float(r1.mean()) > 1e-3
    np.testing.assert_allclose(o1.sum(1).cpu().numpy(), float(T), rtol=1e-4)
    o2, r2, g2 = step(1.0)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and all(torch.equal(g1[k], g2[k]) for k in g1)
    _, _, g3 = step(2.0)
    assert all(torch.allclose(g3[k], 2.0 * g1[k], rtol=1e-5, atol=1e-9) for k in g1)
    # flipping the input in time swaps the two directions' roles: rates of the two halves swap
    torch.manual_seed(5)
    lay = net.snn[0]
    lay.eval()
    with torch.no_grad():
        torch.manual_seed(11); s_a, _ = lay.forward_with_rate(x)
    assert s_a.shape == (B, T, 2048)


def test_cpu_tensors_raise_no_fallback(sp):
    net = sp.SNN((2, None, 16), [8, 8, 4], neuron_type="LIF")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(2, 5, 16))


# ------------------------------------------------------------------------------------ dropout / rates
def test_dropout_statistics_and_backward_mask_consistency(sp):
    torch.manual_seed(3)
    B, T, C, H = 16, 40, 64, 128
    net = sp.SNN((B, None, C), [H, H, 20], neuron_type="adLIF", dropout=0.25).to(DEV)
    net.train()
    x = (torch.rand(B, T, C) < 0.2).float().to(DEV)
    lay = net.snn[0]
    torch.manual_seed(10)
    s_drop, rate = lay.forward_with_rate(x)
    lay.dropout = 0.0  # same train-mode batch statistics, mask off
    torch.manual_seed(10)
    s_full, _ = lay.forward_with_rate(x)
    lay.dropout = 0.25
    s_drop, rate = s_drop.detach(), rate.detach()
    vals = torch.unique(s_drop)
    assert all(min(abs(v), abs(v - 1 / 0.75)) < 1e-6 for v in vals.cpu().tolist()), vals
    fired = s_full > 0
    kept = (s_drop > 0)[fired].float().mean().item()
    assert abs(kept - 0.75) < 0.02, kept
    assert not bool((s_drop > 0)[~fired].any())
    np.testing.assert_allclose(rate.cpu().numpy(), s_drop.mean(dim=(0, 1)).cpu().numpy(), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------ mel front-end
def test_fbank_vs_numpy_restatement_and_known_answers(sp):
    """G9.  PARITY UNPINNED against torchaudio (absent, third-party): checked against the independent
    float64 NumPy restatement in oracle/fbank_numpy.py and analytic known answers."""
    from oracle import fbank_numpy as fo
    g = torch.Generator().manual_seed(0)
    n = 16000
    t = torch.arange(n) / 16000.0
    waves = torch.stack([
        0.1 * (torch.rand(n, generator=g) * 2 - 1) + 0.3 * torch.sin(2 * np.pi * 440.0 * t),   # BASELINE cfg4 shape
        0.5 * torch.sin(2 * np.pi * 1000.0 * t),                                               # pure tone
        torch.full((n,), 0.25),                                                                # DC only
        torch.randn(n, generator=g) * 0.05,
    ])
    out = sp.fbank(waves.to(DEV), num_mel_bins=40).cpu().numpy()
    assert out.shape == (4, 98, 40)
    for i in (0, 1, 3):
        ref = fo.fbank(waves[i].numpy())
        err = np.abs(out[i] - ref)
        # log-mel: 2e-3 wherever the bin carries signal; bins > e^-12 below the frame's peak hold only
        # fp32 FFT rounding leakage (pure tone), which the log amplifies: 2e-2 there
        strong = ref >= ref.max(axis=1, keepdims=True) - 12.0
        assert err[strong].max() <= 2e-3, (i, err[strong].max())
        assert err.max() <= 2e-2, (i, err.max())
    # known answers: a DC signal has no energy after per-frame DC removal -> log(FLT_EPSILON) floor
    assert np.allclose(out[2], np.log(np.finfo(np.float32).eps), atol=1e-3)
    # a 1 kHz tone peaks in the mel bin whose triangle contains 1 kHz
    centers = 700.0 * (np.exp((fo.mel(20.0) + (np.arange(40) + 1) * (fo.mel(8000.0) - fo.mel(20.0)) / 41) / 1127.0) - 1)
    assert abs(int(out[1].mean(0).argmax()) - int(np.abs(centers - 1000.0).argmin())) <= 1


def test_experiment_cli_synthetic_end_to_end(tmp_path):
    """run_exp.py plumbing on the GPU: BASELINE configs[0] shape (LIF [128,128,20], B=4, T=100, C=700)
    on synthetic SHD-shaped spikes, one epoch, checkpoint written, then reload + test-only run."""
    import run_exp
    folder = str(tmp_path / "exp_lif")
    # (the reference saves a checkpoint only when the validation accuracy beats the best so far, which starts at
    # 0: exp.py:263-277.  Seeded, and with enough steps / validation samples of the learnable synthetic task that
    # this does not hinge on one lucky hit among 12 samples.)
    torch.manual_seed(20)
    run_exp.main(["--model_type", "LIF", "--nb_layers", "3", "--nb_hiddens", "128", "--dataset_name", "shd",
                  "--batch_size", "4", "--nb_epochs", "2", "--synthetic", "1", "--synthetic_batches", "24",
                  "--new_exp_folder", folder, "--use_regularizers", "1"])
    import os
    assert os.path.exists(folder + "/checkpoints/best_model.pth")
    # the same command line in the bf16 operand mode (--compute_dtype, added by this build)
    from sparch_amd import functional as Fn
    try:
        run_exp.main(["--model_type", "RadLIF", "--nb_layers", "3", "--nb_hiddens", "64", "--dataset_name", "shd",
                      "--batch_size", "8", "--nb_epochs", "1", "--synthetic", "1", "--synthetic_batches", "3",
                      "--save_best", "0", "--compute_dtype", "bf16", "--new_exp_folder", str(tmp_path / "exp_bf16")])
        assert Fn.compute_dtype() == "bf16"
    finally:
        Fn.set_compute_dtype("fp32")
    run_exp.main(["--use_pretrained_model", "1", "--only_do_testing", "1", "--load_exp_folder", folder,
                  "--dataset_name", "shd", "--batch_size", "4", "--synthetic", "1", "--synthetic_batches", "2"])
    # raw-audio path (sc): waveform -> HIP fbank -> RadLIF
    run_exp.main(["--model_type", "RadLIF", "--nb_hiddens", "64", "--dataset_name", "sc", "--batch_size", "8",
                  "--nb_epochs", "1", "--synthetic", "1", "--synthetic_batches", "2", "--save_best", "0",
                  "--new_exp_folder", str(tmp_path / "exp_sc")])


def test_bin_events_vs_reference_binning(sp):
    """f-3: device event binning == the reference's per-sample np.digitize + sparse->dense (bit-exact counts),
    incl. edge cases: t = 0, t on an edge, t just below max_time, duplicates, an empty sample, rejected events."""
    from oracle import events_numpy as ev
    rng = np.random.default_rng(3)
    edges = np.linspace(0, 1.4, 100)
    samples = []
    for n in (0, 1, 57, 4000, 12000):
        t = rng.uniform(0, 1.4, n).astype(np.float16).astype(np.float32)   # SHD stores float16 times
        u = rng.integers(0, 700, n)
        samples.append((t, u))
    # hand-made edge cases
    t = np.array([0.0, edges[1], np.nextafter(np.float32(edges[1]), np.float32(0)), edges[50], 1.3999, 1.4, 1.5, -0.1,
                  0.7, 0.7, 0.7], np.float32)
    u = np.array([0, 1, 2, 3, 699, 5, 6, 7, 10, 10, 700])
    samples.append((t, u))
    x, dropped = sp.bin_events([s[0] for s in samples], [s[1] for s in samples])
    assert x.shape == (len(samples), 100, 700)
    total_drop = 0
    for i, (t, u) in enumerate(samples):
        ref, nd = ev.bin_sample(t, u)
        total_drop += nd
        assert np.array_equal(x[i].cpu().numpy(), ref), i
    assert int(dropped.item()) == total_drop == 4
    assert float(x[:, 0].sum()) == 0.0      # np.digitize is 1-based: row 0 is never used (reference quirk)
    assert float(x[5, 50, 10]) == 2.0       # duplicates add up


# ------------------------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("neuron_type,sizes,B,T,C", [("adLIF", [512, 512, 20], 128, 250, 700),
                                                     ("RadLIF", [1024, 1024, 35], 256, 250, 700)])
def test_full_size_properties(sp, neuron_type, sizes, B, T, C):
    """BASELINE.json configs[1] and configs[2] at full size: size-independent invariants."""
    Fn = _Fn()
    torch.manual_seed(1234)
    net = sp.SNN((B, None, C), sizes, neuron_type=neuron_type, dropout=0.0).to(DEV)
    net.train()
    g = torch.Generator().manual_seed(4321)
    x = (torch.rand(B, T, C, generator=g) < 0.05).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)

    def step(seed, scale=1.0):
        net.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        out, rates = net(x)
        loss = torch.nn.functional.cross_entropy(out, y) * scale
        loss.backward()
        Fn.check_status()
        return out.detach(), rates.detach(), {k: v.grad.clone() for k, v in net.named_parameters()}

    out1, r1, g1 = step(99)
    # (1) softmax-sum rows add up to T (each step adds a probability vector)
    np.testing.assert_allclose(out1.sum(1).cpu().numpy(), np.full(B, T, np.float32), rtol=1e-4)
    # (2) rates are means of 0/1 spikes
    assert float(r1.min()) >= 0.0 and float(r1.max()) <= 1.0 and float(r1.mean()) > 1e-4
    # (3) determinism: identical seeds -> bitwise identical outputs and gradients
    for lay in net.snn:  # undo the running-stat update so the second pass sees the same state
        pass
    out2, r2, g2 = step(99)
    assert torch.equal(out1, out2) and torch.equal(r1, r2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    # (4) backward is linear in the upstream gradient: 2x loss -> 2x grads (exact in fp32)
    _, _, g3 = step(99, scale=2.0)
    for k in g1:
        assert torch.allclose(g3[k], 2.0 * g1[k], rtol=1e-5, atol=1e-9), k
    # (5) recurrent weights: zero gradient on the masked diagonal
    for k, v in g1.items():
        if k.endswith("V.weight"):
            assert float(torch.diag(v).abs().max()) == 0.0


# ------------------------------------------------------------------ f-2: optimizer step on the device
@pytest.mark.gpu
@pytest.mark.parametrize("kind,bidir,bias", [("RadLIF", False, False), ("adLIF", True, False), ("RLIF", False, True)])
def test_dx_planes_path_is_bit_identical(sp, kind, bidir, bias, monkeypatch):
    """The gradient of a BatchNorm'd projection as bf16 planes made once by the BatchNorm pass
    (sparch_bn_bwd_apply_planes, with the bidirectional halves added in the same pass) and read by the dW / dX products
    (sparch_gemm_spike16_tn_ap, sparch_gemm6_nn_pp) against the fp32-operand path of rounds 1-2 (SPARCH_DX_PLANES=0):
    the truncation split is the same one, so every parameter gradient must be IDENTICAL bit for bit."""
    Fn = _Fn()
    B, T, C, sizes = 4, 64, 40, [256, 256, 10]
    torch.manual_seed(8)
    net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.1, bidirectional=bidir, use_bias=bias).to(DEV).train()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)
    res = []
    for planes in (True, False):
        monkeypatch.setattr(Fn, "USE_DX_PLANES", planes)
        for lay in net.snn:
            lay._calls = 0
        net.zero_grad()
        torch.manual_seed(5)
        Fn.timer.reset()
        Fn.timer.enabled = True
        out, rates = net(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        Fn.timer.enabled = False
        names = set(Fn.timer.collect())
        Fn.check_status()
        assert any(n.startswith("bn_bwd_apply_planes") for n in names) == planes
        res.append({k: v.grad.clone() for k, v in net.named_parameters()})
    assert float(sum(v.abs().sum() for v in res[0].values())) > 0
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,K", [("RadLIF", 700), ("adLIF", 40), ("LIF", 33)])
def test_input_uploaded_as_bytes_equals_the_float_batch(sp, kind, K):
    """The train step's batch as one byte per element (functional.input_from_counts; the reference moves the dense
    float batch over PCIe every step, exp.py:355-356): counts 0..6 — more than one spike per bin included — must
    give, bit for bit, the plane `plane_bf16_exact` makes from the float batch (flag 1), and the network fed the
    bytes the same output, firing rates and parameter gradients as the network fed the floats."""
    Fn = _Fn()
    B, T = 6, 25
    g = torch.Generator().manual_seed(K)
    counts = torch.poisson(torch.full((B, T, K), 0.3), generator=g).clamp_(max=6).to(torch.uint8)
    assert int(counts.max()) > 1
    xf = counts.float().to(DEV)
    xb = Fn.input_from_counts(counts.to(DEV))
    plane_ref, flag_ref = Fn.plane_bf16_exact(xf.view(B * T, K))
    plane, flag = Fn.input_plane_of(xb)
    assert int(flag_ref[0]) == 1 and int(flag[0]) == 1
    assert tuple(plane.shape) == tuple(plane_ref.shape)
    assert torch.equal(plane.view(torch.int16), plane_ref.view(torch.int16))
    torch.manual_seed(2)
    net = sp.SNN((B, None, K), [64, 48, 10], neuron_type=kind, dropout=0.0).to(DEV).train()
    y = torch.randint(0, 10, (B,), generator=g).to(DEV)
    res = []
    for x in (xf, xb):
        net.zero_grad()
        torch.manual_seed(5)
        out, rates = net(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        Fn.check_status()
        res.append((out.detach().clone(), rates.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert kind == "LIF" or float(res[0][1].sum()) > 0  # (the small LIF net stays silent on this input)
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("B,C", [(1, 2), (37, 35), (256, 35), (700, 256), (300, 20)])
def test_cross_entropy_kernel_vs_torch(B, C):
    """The train step's loss (exp.py:100, 362: nn.CrossEntropyLoss(), mean over the batch) and its gradient in one
    launch (sparch_ce_loss) against torch's own cross_entropy on the CPU: loss to 2e-6 relative, d loss / d logits
    to 1e-7 absolute (entries are <= 1 / B), also under an upstream factor (the regulariser adds to the loss)."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(B + C)
    x = (torch.randn(B, C, generator=g) * 3.0).requires_grad_(True)
    y = torch.randint(0, C, (B,), generator=g)
    ref = torch.nn.functional.cross_entropy(x, y)
    (ref * 1.5).backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    got = Fn.CrossEntropyLoss()(xd, y.to(DEV))
    (got * 1.5).backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert float((xd.grad.cpu() - x.grad).abs().max()) <= 1e-7 * 1.5 + 1e-6 * float(x.grad.abs().max())


@pytest.mark.gpu
def test_adam_step_matches_torch_adam():
    """sparch_amd.optim.Adam vs torch.optim.Adam on the CPU (the reference's optimizer, exp.py:89) on the same
    gradients: same operation order; the host's vectorised kernels contract some multiply-adds and its
    sqrt / division round independently, so after 6 steps the O(1) parameters agree to a few ulp
    (tolerance: 2e-6 absolute + 1e-6 relative), over odd sizes and > 24 tensors per launch; the state dicts
    are interchangeable."""
    from sparch_amd.optim import Adam

    g = torch.Generator().manual_seed(5)
    # > 24 tensors per launch with EMPTY ones among them (they are skipped inside a batch: ADVICE r1, each
    # tensor must still be stepped exactly once)
    shapes = [(1024, 700), (1024,), (0,), (35, 1024), (3,), (1,), (4097,)] + [(17, 5)] * 12 + [(0, 4)] + [(17, 5)] * 14
    p_cpu = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    p_gpu = [p.detach().clone().cuda().requires_grad_(True) for p in p_cpu]
    o_cpu = torch.optim.Adam(p_cpu, lr=1e-2)
    o_gpu = Adam(p_gpu, lr=1e-2)
    for step in range(6):
        for pc, pg in zip(p_cpu, p_gpu):
            gr = torch.randn(pc.shape, generator=g) * (10.0 ** (step - 3))
            pc.grad = gr.clone()
            pg.grad = gr.cuda()
        if step == 3:  # ReduceLROnPlateau-style change of lr between steps
            for grp in o_cpu.param_groups + o_gpu.param_groups:
                grp["lr"] *= 0.7
        o_cpu.step()
        o_gpu.step()
    for pc, pg in zip(p_cpu, p_gpu):
        torch.testing.assert_close(pg.detach().cpu(), pc.detach(), rtol=1e-6, atol=2e-6)
    sd = o_gpu.state_dict()
    o_new = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in p_cpu], lr=1e-2)
    o_new.load_state_dict({"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                     for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]})
    assert float(o_new.state_dict()["state"][0]["step"]) == 6.0
    with pytest.raises(NotImplementedError):
        Adam(p_gpu, amsgrad=True)
    # the step is guarded by the recurrent kernels' status word: raised -> parameters and moments untouched
    from sparch_amd import functional as Fn
    before = [p.detach().clone() for p in p_gpu]
    Fn.status_word("cuda")[0] = 1
    try:
        o_gpu.step()
        torch.cuda.synchronize()
    finally:
        Fn.status_word("cuda").zero_()
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, p_gpu))
    o_gpu.step()
    assert not torch.equal(before[0], p_gpu[0].detach())


# ------------------------------------------------------------------ f-4: non-spiking baselines (anns.py)
def _ann_from_fixture(name):
    import json

    from sparch_amd.anns import ANN
    from tests.golden_io import load

    z = load(name)
    cfg = json.loads(str(z["cfg"]))
    net = ANN(input_shape=(cfg["B"], None, cfg["C"]), layer_sizes=cfg["layer_sizes"], ann_type=cfg["ann_type"],
              dropout=0.0, normalization=cfg["normalization"], use_bias=cfg["use_bias"],
              bidirectional=cfg["bidirectional"], use_readout_layer=cfg["use_readout_layer"])
    net.load_state_dict({k[len("param."):]: torch.from_numpy(v) for k, v in z.items() if k.startswith("param.")})
    return cfg, z, net.to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ann_MLP_bn", "ann_MLP_ln_bias_noreadout", "ann_RNN_bn", "ann_RNN_bidir",
                                  "ann_LiGRU_bn", "ann_GRU_bn"])
def test_mlp_ann_matches_reference_fixture(name):
    """The MLP baseline + ANN readout on the HIP path against the real reference's outputs, loss, gradients,
    running statistics and eval-mode output (tests/golden/ann_MLP_*.npz).  fp32 tolerance, relative to each
    tensor's largest entry: 2e-4 without the readout.  With it 1e-3 (outputs) / 3e-3 (gradients): the ANN
    readout sums softmaxes of sigmoid outputs — nearly uniform, so W y varies little over the batch — and
    its BatchNorm over the fixture's batch of 6 divides by that small deviation, which magnifies the
    last-bit differences of expf and summation order about a thousandfold (re-ordering the time sum in
    the CPU oracle alone moves the reference's own output by 2e-5)."""
    cfg, z, net = _ann_from_fixture(name)
    tol_out, tol_grad = (1e-3, 3e-3) if cfg["use_readout_layer"] else (2e-4, 5e-4)
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    net.train()
    out, none = net(x)
    assert none is None
    assert relmax(out.detach().cpu().numpy(), z["out"]) <= tol_out
    loss = torch.nn.functional.cross_entropy(out, y) if cfg["use_readout_layer"] else (out * out).mean()
    assert abs(float(loss.detach()) - float(z["loss"])) <= tol_out * max(1.0, abs(float(z["loss"])))
    loss.backward()
    for k, p in net.named_parameters():
        assert relmax(p.grad.cpu().numpy(), z["grad." + k]) <= tol_grad, k
    sd = net.state_dict()
    for k in z:
        if k.startswith("after."):
            np.testing.assert_allclose(sd[k[len("after."):]].cpu().numpy(), z[k], rtol=1e-4, atol=1e-6, err_msg=k)
    net.eval()
    with torch.no_grad():
        out_e, _ = net(x)
    assert relmax(out_e.cpu().numpy(), z["out_eval"]) <= tol_out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sigmoid", "relu", "tanh"])
def test_act_kernels_vs_torch(kind):
    """sparch_act_fwd/bwd (folded affine + activation + dropout) against torch fp32 on the CPU; the dropout
    mask is the kernel's own (regenerated identically in the backward), checked through the outputs."""
    from sparch_amd._capi import check, lib, ptr

    Fn = _Fn()
    g = torch.Generator().manual_seed(3)
    M, H = 37, 52
    z = torch.randn(M, H, generator=g) * 2
    sc, sh = torch.rand(H, generator=g) + 0.5, torch.randn(H, generator=g) * 0.3
    dy = torch.randn(M, H, generator=g)
    f = {"sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}[kind]
    zz = (z * sc + sh).requires_grad_(True)
    ref = f(zz)
    ref.backward(dy)
    # tanh: the device's tanhf and the host's differ by a few ulp, and 1 - a^2 cancels near saturation
    ry, ay, rg, ag = (5e-6, 5e-7, 2e-4, 2e-6) if kind == "tanh" else (2e-6, 2e-7, 2e-5, 2e-7)
    for p_drop in (0.0, 0.25):
        y = torch.empty(M, H, device=DEV)
        dz = torch.empty(M, H, device=DEV)
        zd, scd, shd, dyd = z.to(DEV), sc.to(DEV), sh.to(DEV), dy.to(DEV)
        k = Fn.ACT_KIND[kind]
        check(lib.sparch_act_fwd(k, M * H, H, ptr(zd), ptr(scd), ptr(shd), p_drop, 99, ptr(y), Fn._stream()), "act_fwd")
        check(lib.sparch_act_bwd(k, M * H, H, ptr(zd), ptr(scd), ptr(shd), ptr(dyd), p_drop, 99, ptr(dz), Fn._stream()), "act_bwd")
        y, dz = y.cpu(), dz.cpu()
        keep = 1.0 / (1.0 - p_drop)
        if p_drop == 0.0:
            torch.testing.assert_close(y, ref.detach(), rtol=ry, atol=ay)
            torch.testing.assert_close(dz, zz.grad, rtol=rg, atol=ag)
        else:
            kept = (dz != 0) | (y != 0)
            frac = float(kept.float().mean())
            assert abs(frac - (1 - p_drop)) < 0.06 or kind == "relu"   # relu is zero on half its inputs anyway
            torch.testing.assert_close(y[kept], (ref.detach() * keep)[kept], rtol=ry, atol=ay)
            torch.testing.assert_close(dz[kept], (zz.grad * keep)[kept], rtol=rg, atol=ag)
            assert bool((y[~kept] == 0).all()) and bool((dz[~kept] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,K", [(3, 17, 48), (5, 40, 1024), (2, 9, 2052)])
def test_softmax_sum_kernels_vs_torch(B, T, K):
    from sparch_amd._capi import check, lib, ptr

    Fn = _Fn()
    g = torch.Generator().manual_seed(B * 100 + T)
    x = (torch.randn(B, T, K, generator=g) * 3).requires_grad_(True)
    gy = torch.randn(B, K, generator=g)
    ref = 0
    for t in range(T):
        ref = ref + torch.softmax(x[:, t, :], dim=-1)
    ref.backward(gy)
    xd = x.detach().to(DEV)
    out = torch.empty(B, K, device=DEV)
    dx = torch.empty(B, T, K, device=DEV)
    check(lib.sparch_softmax_sum_fwd(B, T, K, ptr(xd), ptr(out), Fn._stream()), "softmax_sum_fwd")
    check(lib.sparch_softmax_sum_bwd(B, T, K, ptr(xd), ptr(gy.to(DEV)), ptr(dx), Fn._stream()), "softmax_sum_bwd")
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("bidir,H", [(False, 256), (True, 128)])
def test_rnn_layer_vs_oracle_larger_and_chunked(bidir, H, monkeypatch):
    """RNN baseline layer on shapes with several row tiles and k-groups per wave, against the CPU oracle
    (oracle/ann_oracle.py, pinned to the reference); one persistent launch, 7-step chunks and one launch per
    step must agree BIT FOR BIT (same arithmetic, only the launch boundaries move)."""
    from oracle import ann_oracle as ao
    from sparch_amd.anns import RNNLayer

    B, T, C = 40, 23, 64
    torch.manual_seed(17)
    layer = RNNLayer(C, H, B, dropout=0.0, normalization="batchnorm", use_bias=True, bidirectional=bidir)
    with torch.no_grad():
        layer.norm.weight.uniform_(0.7, 1.3)
        layer.norm.bias.uniform_(-0.2, 0.2)
    g = torch.Generator().manual_seed(18)
    x = torch.randn(B, T, C, generator=g)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g)
    p = {"ann.0." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in layer.state_dict().items() if "num_batches" not in k}
    xr = x.clone().requires_grad_(True)
    ref = ao.hidden_layer("RNN", xr, p, "ann.0", "batchnorm", bidir, training=True, running=None)
    (ref * gy).sum().backward()
    layer = layer.to(DEV).train()
    outs = []
    for spl in ("", "7", "1"):
        monkeypatch.setenv("SPARCH_REC_STEPS_PER_LAUNCH", spl)
        layer.zero_grad()
        xd = x.to(DEV).requires_grad_(True)
        y = layer(xd)
        (y * gy.to(DEV)).sum().backward()
        _Fn().check_status()
        outs.append((y.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu() for k, v in layer.named_parameters()}))
    y0, dx0, g0 = outs[0]
    assert relmax(y0.numpy(), ref.detach().numpy()) <= 2e-5
    assert relmax(dx0.numpy(), xr.grad.numpy()) <= 2e-4
    for k, v in g0.items():
        if k == "W.bias":  # BatchNorm removes the column mean: this gradient is exactly zero in real arithmetic
            assert np.abs(v.numpy() - p["ann.0." + k].grad.numpy()).max() <= 1e-4 * float(g0["W.weight"].abs().max())
            continue
        assert relmax(v.numpy(), p["ann.0." + k].grad.numpy()) <= 2e-4, k
    for y1, dx1, g1 in outs[1:]:
        assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
        for k in g0:
            assert torch.equal(g1[k], g0[k]), k


@pytest.mark.gpu
def test_rnn_layer_hidden_size_above_1024_vs_oracle(monkeypatch):
    """RNN baseline with H = 1536 (step path) against the oracle, and the step path against the persistent kernel
    at H = 128 (same arithmetic up to summation order)."""
    from oracle import ann_oracle as ao
    from sparch_amd.anns import RNNLayer

    def run(H, B, T, C, bidir, step):
        torch.manual_seed(21)
        layer = RNNLayer(C, H, B, dropout=0.0, normalization="batchnorm", use_bias=False, bidirectional=bidir)
        g = torch.Generator().manual_seed(22)
        x = torch.randn(B, T, C, generator=g)
        gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g)
        p = {"ann.0." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
             for k, v in layer.state_dict().items() if "num_batches" not in k}
        xr = x.clone().requires_grad_(True)
        ref = ao.hidden_layer("RNN", xr, p, "ann.0", "batchnorm", bidir, training=True, running=None)
        (ref * gy).sum().backward()
        monkeypatch.setenv("SPARCH_REC_STEP_PATH", "1" if step else "0")
        layer = layer.to(DEV).train()
        xd = x.to(DEV).requires_grad_(True)
        y = layer(xd)
        (y * gy.to(DEV)).sum().backward()
        _Fn().check_status()
        got = {k: v.grad.cpu() for k, v in layer.named_parameters()}
        return y.detach().cpu(), xd.grad.cpu(), got, ref.detach(), xr.grad, {k[6:]: v.grad for k, v in p.items() if v.grad is not None}

    y, dx, g, ref, dxr, gr = run(1536, 6, 9, 40, False, False)
    assert relmax(y.numpy(), ref.numpy()) <= 2e-5 and relmax(dx.numpy(), dxr.numpy()) <= 2e-4
    for k in g:
        assert relmax(g[k].numpy(), gr[k].numpy()) <= 2e-4, k
    ya, dxa, ga, *_ = run(128, 40, 11, 32, True, False)
    yb, dxb, gb, *_ = run(128, 40, 11, 32, True, True)
    assert relmax(yb.numpy(), ya.numpy()) <= 1e-5 and relmax(dxb.numpy(), dxa.numpy()) <= 1e-5
    for k in ga:
        assert relmax(gb[k].numpy(), ga[k].numpy()) <= 2e-5, k


@pytest.mark.gpu
def test_rnn_layer_dropout_is_applied_after_the_cell():
    """anns.py:323-324: dropout acts on the layer output only; the recurrent state stays un-dropped, so the
    kept entries equal the no-dropout output scaled by 1/(1-p)."""
    from sparch_amd.anns import RNNLayer

    torch.manual_seed(5)
    layer = RNNLayer(32, 64, 8, dropout=0.3, normalization="none").to(DEV)
    x = torch.randn(8, 15, 32, generator=torch.Generator().manual_seed(6)).to(DEV)
    layer.eval()
    with torch.no_grad():
        y_ref = layer(x).cpu()
    layer.train()
    with torch.no_grad():
        y = layer(x).cpu()
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.7) < 0.03
    torch.testing.assert_close(y[kept], (y_ref / 0.7)[kept], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_full_size_cfg_audio_frontend_to_radlif(sp):
    """BASELINE.json configs[3] at full size on one GPU: raw audio (256 x 16000) -> HIP mel filterbank
    (98 frames x 40 bins) -> RadLIF [1024,1024,35]; size-independent invariants."""
    Fn = _Fn()
    B = 256
    g = torch.Generator().manual_seed(7)
    t = torch.arange(16000, dtype=torch.float32) / 16000.0
    wave = (torch.rand(B, 16000, generator=g) * 2 - 1) * 0.1 + 0.3 * torch.sin(2 * np.pi * 440.0 * t)
    feats = sp.fbank(wave.to(DEV), num_mel_bins=40)
    assert tuple(feats.shape) == (B, 98, 40) and bool(torch.isfinite(feats).all())
    torch.manual_seed(1234)
    net = sp.SNN((B, None, 40), [1024, 1024, 35], neuron_type="RadLIF", dropout=0.1).to(DEV).train()
    y = torch.randint(0, 35, (B,), generator=g).to(DEV)
    outs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        for lay in net.snn:
            lay._calls = 0  # same dropout seeds in both passes
        torch.manual_seed(5)
        out, rates = net(feats)
        torch.nn.functional.cross_entropy(out, y).backward()
        Fn.check_status()
        outs.append((out.detach().clone(), rates.detach().clone(), net.snn[0].W.weight.grad.clone()))
    out, rates, gW = outs[0]
    np.testing.assert_allclose(out.sum(1).cpu().numpy(), np.full(B, 98, np.float32), rtol=1e-4)
    assert float(rates.min()) >= 0.0 and bool(torch.isfinite(gW).all()) and float(gW.abs().max()) > 0
    assert torch.equal(outs[1][0], out) and torch.equal(outs[1][2], gW)  # bitwise reproducible


@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_full_size_cfg_bidirectional_long_sequence(sp, compute, request, monkeypatch):
    """BASELINE.json configs[4] at full size on one GPU: bidirectional RadLIF [1024,1024,1024,35], B=256,
    T=1000 (512 virtual rows: two groups of row tiles per launch sequence); in fp32 (>= the config's bf16) and
    as the configuration names it: the bf16 operand mode with bf16 saved states."""
    Fn = _Fn()
    if compute == "bf16":
        request.getfixturevalue("bf16_mode")
        monkeypatch.setattr(Fn, "SAVE_BF16", True)
    B, T, C = 256, 1000, 700
    torch.manual_seed(1234)
    net = sp.SNN((B, None, C), [1024, 1024, 1024, 35], neuron_type="RadLIF", dropout=0.0, bidirectional=True).to(DEV)
    net.train()
    g = torch.Generator().manual_seed(4321)
    x = (torch.rand(B, T, C, generator=g) < 0.05).float().to(DEV)
    y = torch.randint(0, 35, (B,), generator=g).to(DEV)
    torch.manual_seed(99)
    out, rates = net(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    Fn.check_status()
    np.testing.assert_allclose(out.detach().sum(1).cpu().numpy(), np.full(B, T, np.float32), rtol=1e-4)
    assert tuple(rates.shape) == (3 * 2048,)
    assert float(rates.min()) >= 0.0 and float(rates.max()) <= 1.0 and float(rates.mean()) > 1e-4
    # The reference's own subthreshold (u, w) map is unstable for a < ~0 (eigenvalue up to 1.4 per step, see
    # DESIGN.md §2): over 1000 steps the membrane state of such neurons overflows in fp32 — in the reference's
    # arithmetic as in ours — and their alpha / beta / a gradients (du * (inf - inf)) are NaN.  Everything
    # that does not multiply those states stays finite.  WHICH entries are non-finite is pinned against the
    # reference at T=1000, H=64 by test_snn_long_sequence_non_finite_gradient_mask_matches_reference; at this
    # size no reference run exists (trajectories are not reproducible with real-valued V), so here only the
    # parameters whose gradients never touch those states are required to be finite.
    for k, v in net.named_parameters():
        assert v.grad is not None, k
        if k.split(".")[-1] not in ("alpha", "beta", "a"):
            assert bool(torch.isfinite(v.grad).all()), k
        if k.endswith("V.weight"):
            assert float(torch.diag(v.grad).abs().max()) == 0.0
    assert bool(torch.isfinite(net.snn[3].alpha.grad).all())  # the readout has no adaptation state
    assert tuple(net.snn[1].W.weight.shape) == (1024, 2048)  # hidden input doubles (snns.py:140)


@pytest.mark.gpu
def test_shd_ssc_loader_batches_binned_on_device():
    """sparch_amd.dataloaders.load_shd_or_ssc on an in-memory stand-in for the h5 file: every batch equals the
    reference's per-sample np.digitize + sparse->dense (oracle/events_numpy.py) bit for bit, and keeps the
    (x, xlens, y) collate contract (spiking_datasets.py:80-87)."""
    from oracle import events_numpy as ev
    from sparch_amd.dataloaders.spiking_datasets import load_shd_or_ssc
    from tests.test_host import _fake_h5

    h5 = _fake_h5(n=11)
    loader = load_shd_or_ssc("shd", "/unused", "valid", batch_size=4, shuffle=False, h5_file=h5, device=DEV)
    seen = 0
    for xs, xlens, ys in loader:
        b = xs.shape[0]
        assert xs.device.type == "cuda" and tuple(xs.shape[1:]) == (100, 700) and xs.dtype == torch.float32
        assert xlens.tolist() == [100] * b and ys.dtype == torch.int64
        for j in range(b):
            ref, _ = ev.bin_sample(h5["spikes"]["times"][seen + j], h5["spikes"]["units"][seen + j])
            np.testing.assert_array_equal(xs[j].cpu().numpy(), ref)
            assert int(ys[j]) == int(h5["labels"][seen + j])
        seen += b
    assert seen == 11


@pytest.mark.parametrize("neuron_type,compute", [("LIF", "fp32"), ("RadLIF", "fp32"), ("RadLIF", "bf16"),
                                                 ("adLIF", "bf16")])
def test_training_learns_class_conditional_synthetic_task(sp, neuron_type, compute, request):
    """End to end: forward + CE on the softmax-sum + backward + one-launch Adam (exp.py:359-377) on the
    synthetic SHD-shaped task whose labels select a band of input channels.  Gradient parity is checked
    elsewhere; this checks that the pieces train: held-out accuracy far above chance (1/20) after 60 steps —
    in the exact fp32 mode and in the bf16 operand mode (same bar)."""
    from sparch_amd.exp import _SyntheticLoader
    from sparch_amd.optim import Adam

    if compute == "bf16":
        request.getfixturevalue("bf16_mode")
    B, T = 64, 50
    torch.manual_seed(3)
    net = sp.SNN((B, None, 700), [128, 128, 20], neuron_type=neuron_type, dropout=0.1).to(DEV).train()
    opt = Adam(net.parameters(), 1e-2)
    first = None
    for x, _, y in _SyntheticLoader("spiking", B, 60, 20, T, seed=11):
        out, rates = net(x.to(DEV))
        loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
        first = float(loss.detach()) if first is None else first
        opt.zero_grad()
        loss.backward()
        opt.step()
    _Fn().check_status()
    net.eval()
    hits = n = 0
    with torch.no_grad():
        for x, _, y in _SyntheticLoader("spiking", B, 4, 20, T, seed=12):
            out, _ = net(x.to(DEV))
            hits += int((out.argmax(1).cpu() == y).sum())
            n += B
    assert float(loss.detach()) < 0.7 * first, (first, float(loss.detach()))
    assert hits / n > 0.6, hits / n


@pytest.mark.parametrize("kind", ["MLP", "RNN", "LiGRU", "GRU"])
def test_bf16_operand_mode_baseline_layers_vs_oracle(kind, bf16_mode):
    """The non-spiking baseline layers in the bf16 operand mode against the fp32 CPU oracle: smooth functions of
    their weights, so the stated tolerance is the operands' rounding: MLP / RNN / GRU 2e-2 of each tensor's largest
    entry and 1e-2 relative rms (output and every gradient), LiGRU (ReLU derivative flips, see below) 0.2 / 6e-2,
    on the persistent recurrent kernels where the layer has them."""
    from oracle import ann_oracle as ao
    from sparch_amd import anns

    B, T, C, H = 12, 21, 40, 64
    torch.manual_seed(29)
    layer = getattr(anns, kind + "Layer")(C, H, B, dropout=0.0, normalization="batchnorm", use_bias=False)
    g = torch.Generator().manual_seed(30)
    x = torch.randn(B, T, C, generator=g)
    gy = torch.randn(B, T, H, generator=g)
    p = {"ann.0." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in layer.state_dict().items() if "num_batches" not in k}
    xr = x.clone().requires_grad_(True)
    ref = ao.hidden_layer(kind, xr, p, "ann.0", "batchnorm", False, training=True, running=None)
    (ref * gy).sum().backward()
    layer = layer.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    y = layer(xd)
    (y * gy.to(DEV)).sum().backward()
    bf16_mode.check_status()

    def relrms(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.sqrt(((a - b) ** 2).sum() / ((b ** 2).sum() + 1e-30)))

    # LiGRU's candidate is a ReLU: a pre-activation within the forward's 2e-3 rounding of zero flips its
    # derivative (0 / 1), a discrete change no operand precision bounds — measured 0.10 of max-abs / 3.3e-2 rms for
    # its dx while its output agrees to 2.5e-3; the smooth cells stay at the operands' rounding level
    tol_max, tol_rms = (0.2, 6e-2) if kind == "LiGRU" else (2e-2, 1e-2)
    pairs = [("y", y.detach().cpu().numpy(), ref.detach().numpy()), ("dx", xd.grad.cpu().numpy(), xr.grad.numpy())]
    pairs += [(k, v.grad.cpu().numpy(), p["ann.0." + k].grad.numpy()) for k, v in layer.named_parameters()]
    assert relmax(pairs[0][1], pairs[0][2]) > 0.0, "the exact kernels ran"
    for name, got, want in pairs:
        em, er = relmax(got, want), relrms(got, want)
        print(f"bf16 operand mode, {kind} {name}: relmax {em:.2e} relrms {er:.2e}")
        assert em <= tol_max and er <= tol_rms, (name, em, er)


@pytest.mark.parametrize("kind", ["adLIF", "RLIF", "RadLIF"])
def test_bf16_operand_mode_equals_exact_mode_on_rounded_weights(sp, kind, bf16_mode):
    """Real-valued weights (the reference's initialisation: not bf16-representable).  What the mode does is
    defined operationally: its forward pass must equal, BIT FOR BIT, the exact fp32 path run on the same network
    with W and V replaced by their bf16-rounded values (adding the zero mid / lo planes changes no sum) — through
    the projections, BatchNorm, the recurrent s @ V and the readout.  On that identical spike trajectory the
    backward pass differs only by the bf16 rounding of dWx in its matrix products: every parameter gradient
    within 2e-2 of its max-abs.  (`a` is kept >= 0 so that the subthreshold (u, w) map contracts: with a -> -1 the
    reference's own gradients grow like 1.4^T and amplify any rounding; see the T = 1000 fixture.)"""
    Fn = bf16_mode
    B, T, C, sizes = 24, 30, 96, [128, 128, 20]
    g = torch.Generator().manual_seed(78)
    x = (torch.rand(B, T, C, generator=g) < 0.15).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)
    torch.manual_seed(5)
    net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.0, normalization="batchnorm").to(DEV).train()
    with torch.no_grad():
        for lay in net.snn:
            if hasattr(lay, "a"):
                lay.a.uniform_(0.0, 1.0)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def run(mode, rounded):
        Fn.set_compute_dtype(mode)
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        if rounded:
            with torch.no_grad():
                for k, v in net.named_parameters():
                    if k.endswith("W.weight") or k.endswith("V.weight"):
                        v.copy_(v.to(torch.bfloat16).float())
        torch.manual_seed(6)
        out, rates = net(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        Fn.check_status()
        return out.detach().clone(), rates.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters()}

    out_b, rates_b, g_b = run("bf16", rounded=False)
    out_f, rates_f, g_f = run("fp32", rounded=True)
    out_x, rates_x, _ = run("fp32", rounded=False)
    Fn.set_compute_dtype("bf16")
    assert float(rates_b.sum()) > 0
    assert torch.equal(rates_b, rates_f) and torch.equal(out_b, out_f), "forward of the bf16 mode != exact path on rounded weights"
    assert not torch.equal(out_b, out_x), "the weights were bf16-exact: the test would prove nothing"
    worst = 0.0
    for k in g_b:
        e = relmax(g_b[k].cpu().numpy(), g_f[k].cpu().numpy())
        worst = max(worst, e)
        assert e <= 2e-2, (k, e)
    print(f"{kind}: worst gradient relmax, bf16 mode vs exact path on rounded weights: {worst:.2e}")


@pytest.mark.parametrize("kind", ["adLIF", "RadLIF"])
def test_bf16_operand_mode_real_valued_network_statistics(sp, kind, bf16_mode):
    """A network with the reference's real-valued initial weights in both modes on the same draws: rounding the
    weights moves membrane potentials by ~2^-9 relative, a few potentials cross the threshold the other way and
    the spike trains then differ, so (as for the real-valued recurrent fixtures) the comparison with the fp32 path
    is statistical — stated bars: mean |firing-rate difference| <= 0.01, loss within 5 % (measured 5e-4, 0.6 %).
    Gradients are NOT compared here: at the reference's initialisation (a in [-1, 1]) they are dominated by the
    expanding subthreshold modes and change sign under a handful of spike flips in either mode; the rigorous
    gradient check of the mode is test_bf16_operand_mode_equals_exact_mode_on_rounded_weights."""
    Fn = bf16_mode
    B, T, C, sizes = 32, 60, 120, [128, 128, 20]
    g = torch.Generator().manual_seed(77)
    x = (torch.rand(B, T, C, generator=g) < 0.1).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        Fn.set_compute_dtype(mode)
        torch.manual_seed(5)
        net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.0, normalization="batchnorm").to(DEV).train()
        torch.manual_seed(6)
        out, rates = net(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        Fn.check_status()
        res[mode] = (rates.detach().cpu(), float(loss.detach()), {k: v.grad.cpu() for k, v in net.named_parameters()})
    Fn.set_compute_dtype("bf16")
    (r0, l0, g0), (r1, l1, g1) = res["fp32"], res["bf16"]
    assert float(r0.sum()) > 0 and not torch.equal(r0, r1), "the two modes ran the same kernels"
    assert float((r0 - r1).abs().mean()) <= 0.01
    assert abs(l0 - l1) <= 0.05 * abs(l0)
    for k in g1:
        assert bool(torch.isfinite(g1[k]).all()), k


def test_ann_random_configurations_vs_oracle():
    """Forty-eight non-spiking baseline networks drawn at random (MLP / RNN / LiGRU / GRU + readout; hidden widths
    5-66, most not multiples of 4 — they run zero-padded, sparch_amd/anns.py — 3-35 classes, with and without bias,
    bidirectional, batchnorm or none; the last sixteen with layernorm, whose kernels normalise over the true width
    inside the padded rows) against the CPU oracle: outputs and every parameter gradient (fp32 bars: 2e-4 / 2e-3
    without batchnorm, 2e-3 / 1e-2 with batchnorm over these small batches, cf. the fixture test)."""
    from oracle import ann_oracle as ao
    from sparch_amd.anns import ANN
    rng = np.random.default_rng(3)
    bad = []
    for it in range(48):
        kind = ["MLP", "RNN", "LiGRU", "GRU"][it % 4]
        B, T, C = int(rng.choice([2, 6, 17])), int(rng.choice([3, 9, 21])), int(rng.choice([3, 12, 37]))
        sizes = [int(rng.choice([5, 30, 33, 66])), int(rng.choice([7, 30, 64])), int(rng.choice([3, 20, 35]))]
        bidir = bool(rng.integers(2)) and kind != "MLP"
        norm = ["batchnorm", "none"][int(rng.integers(2))]
        if it >= 32:
            norm = "layernorm"
        bias = bool(rng.integers(2))
        torch.manual_seed(50 + it)
        net = ANN((B, None, C), sizes, ann_type=kind, dropout=0.0, normalization=norm, use_bias=bias, bidirectional=bidir)
        p = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
             for k, v in net.state_dict().items() if "num_batches" not in k}
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, T, C, generator=g)
        y = torch.randint(0, sizes[-1], (B,), generator=g)
        cfg = dict(ann_type=kind, layer_sizes=sizes, use_readout_layer=True, normalization=norm, bidirectional=bidir)
        torch.nn.functional.cross_entropy(ao.ann_forward(cfg, p, x, training=True, running=None), y).backward()
        out_o = ao.ann_forward(cfg, {k: v.detach() for k, v in p.items()}, x, training=True, running=None)
        net = net.to(DEV).train()
        out, _ = net(x.to(DEV))
        torch.nn.functional.cross_entropy(out, y.to(DEV)).backward()
        _Fn().check_status()
        eo = relmax(out.detach().cpu().numpy(), out_o.numpy())
        eg = max(relmax(v.grad.cpu().numpy(), p[k].grad.numpy()) for k, v in net.named_parameters()
                 if not (k.endswith(".bias") and "norm" not in k and norm == "batchnorm"))  # W*.bias: zero in real arithmetic
        tol_o, tol_g = (2e-3, 1e-2) if norm == "batchnorm" else (2e-4, 2e-3)
        if not (eo <= tol_o and eg <= tol_g):
            bad.append((kind, (B, T, C), sizes, bidir, norm, bias, eo, eg))
    assert not bad, bad[:5]


@pytest.mark.parametrize("kind,bidir,norm", [("LiGRU", True, "batchnorm"), ("GRU", True, "layernorm"),
                                             ("GRU", False, "none")])
def test_gated_baseline_layers_vs_oracle(kind, bidir, norm):
    """LiGRU / GRU baseline layers (launch-per-step path) against the CPU oracle (pinned to the reference) on
    shapes beyond the fixtures: bidirectional, bias, every normalisation; fp32 tolerance 2e-4 of each tensor's
    largest entry (5e-5 on the output)."""
    from oracle import ann_oracle as ao
    from sparch_amd import anns

    B, T, C, H = 10, 19, 36, 64
    torch.manual_seed(23)
    layer = getattr(anns, kind + "Layer")(C, H, B, dropout=0.0, normalization=norm, use_bias=True, bidirectional=bidir)
    with torch.no_grad():
        for n in ("norm", "normz", "normr"):
            if hasattr(layer, n):
                getattr(layer, n).weight.uniform_(0.7, 1.3)
                getattr(layer, n).bias.uniform_(-0.2, 0.2)
    g = torch.Generator().manual_seed(24)
    x = torch.randn(B, T, C, generator=g)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g)
    p = {"ann.0." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in layer.state_dict().items() if "num_batches" not in k}
    xr = x.clone().requires_grad_(True)
    ref = ao.hidden_layer(kind, xr, p, "ann.0", norm, bidir, training=True, running=None)
    (ref * gy).sum().backward()
    layer = layer.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    y = layer(xd)
    (y * gy.to(DEV)).sum().backward()
    assert relmax(y.detach().cpu().numpy(), ref.detach().numpy()) <= 5e-5
    assert relmax(xd.grad.cpu().numpy(), xr.grad.numpy()) <= 2e-4
    wmax = float(layer.W.weight.grad.abs().max())
    for k, v in layer.named_parameters():
        r = p["ann.0." + k].grad.numpy()
        if k.endswith(".bias") and k[0] == "W" and norm != "none":  # removed by the normalisation: exactly zero in real arithmetic
            assert np.abs(v.grad.cpu().numpy() - r).max() <= 1e-4 * wmax, k
            continue
        assert relmax(v.grad.cpu().numpy(), r) <= 2e-4, k


@pytest.mark.parametrize("bidir,B,T,H", [(False, 40, 23, 128), (True, 36, 17, 256)])
def test_ligru_persistent_kernels_vs_launch_per_step_and_chunked(bidir, B, T, H, monkeypatch):
    """LiGRU on the persistent kernels (csrc/gatedcell.hip: 16 hidden units per workgroup, forward on the
    32x32x16 MFMA with interleaved [z | c] columns, backward contracting the stacked [dz_pre | dc_pre] on the
    16x16x32 MFMA) against the launch-per-step path (same arithmetic, different summation order: 1e-5 of
    max-abs), and chunked / per-step launches of the persistent kernels against the whole-sequence launch
    (bit for bit)."""
    from sparch_amd.anns import LiGRULayer

    C = 48
    torch.manual_seed(31)
    layer = LiGRULayer(C, H, B, dropout=0.0, normalization="batchnorm", use_bias=False, bidirectional=bidir).to(DEV).train()
    g = torch.Generator().manual_seed(32)
    x = torch.randn(B, T, C, generator=g).to(DEV)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g).to(DEV)

    def run(persistent, spl):
        monkeypatch.setenv("SPARCH_LIGRU_PERSISTENT", "1" if persistent else "0")
        monkeypatch.setenv("SPARCH_REC_STEPS_PER_LAUNCH", spl)
        layer.zero_grad()
        xd = x.clone().requires_grad_(True)
        y = layer(xd)
        (y * gy).sum().backward()
        _Fn().check_status()
        return y.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu().clone() for k, v in layer.named_parameters()}

    y0, dx0, g0 = run(True, "")
    y1, dx1, g1 = run(False, "")
    assert float(y0.abs().max()) > 0
    assert relmax(y0.numpy(), y1.numpy()) <= 1e-5 and relmax(dx0.numpy(), dx1.numpy()) <= 2e-5
    for k in g0:
        assert relmax(g0[k].numpy(), g1[k].numpy()) <= 5e-5, k
    for spl in ("5", "1"):
        y2, dx2, g2 = run(True, spl)
        assert torch.equal(y2, y0) and torch.equal(dx2, dx0), spl
        for k in g0:
            assert torch.equal(g2[k], g0[k]), (spl, k)


def test_gemm_shape_sweep_pipelined_and_general_paths():
    """Shape sweep across the split-GEMM variants (pipelined kernel with shifted edge tiles and a peeled K tail,
    general bounds-checked kernel, split-K counts that do not divide K, bf16 spike planes, odd leading sizes):
    every result within 2e-6 * sum|a||b| of the fp64 product."""
    Fn = _Fn()
    rng = np.random.default_rng(123)
    shapes = [(260, 132, 264), (516, 388, 1000), (1028, 260, 4100), (300, 1000, 520), (256, 128, 256),
              (772, 644, 300), (2052, 140, 772), (264, 520, 8200)]
    for (M, N, K) in shapes:
        g = torch.Generator().manual_seed(M + N + K)
        D1 = torch.randn(M, K, generator=g)
        D2 = torch.randn(N, K, generator=g)
        S1 = (torch.rand(M, K, generator=g) < 0.1).float()
        S16 = S1.to(torch.bfloat16)

        def chk(C, ref, bound, what):
            err = (C.cpu().double() - ref).abs()
            assert bool((err <= bound).all()), (what, M, N, K, float((err / bound).max()))

        # NT: spike x dense (fp32 spikes and bf16 plane), dense x dense
        ref = S1.double() @ D2.double().T
        bnd = (S1.double() @ D2.abs().double().T) * 2e-6 + 1e-6
        C_a, _ = Fn.gemm_nt(S1.to(DEV), D2.to(DEV), spike_scale=1.0)
        C_b, _ = Fn.gemm_nt(S1.to(DEV), D2.to(DEV), spike_scale=1.0, a16=S16.to(DEV))
        chk(C_a, ref, bnd, "spike_nt")
        assert torch.equal(C_a, C_b)
        refd = D1.double() @ D2.double().T
        bndd = (D1.abs().double() @ D2.abs().double().T) * 2e-6 + 1e-6
        chk(Fn.gemm_nt(D1.to(DEV), D2.to(DEV))[0], refd, bndd, "dense_nt")
        # NN: dense (M,K) x dense (K,N)
        chk(Fn.gemm_nn(D1.to(DEV), D2.T.contiguous().to(DEV)), refd, bndd, "dense_nn")
        # TN over the long axis: operands (K', M') with K' = M here
        A_t, B_t = D1[:, :min(K, 520)].contiguous(), S1[:, :min(K, 388)].contiguous()   # (M, m2), (M, n2)
        ref_t = A_t.double().T @ B_t.double()
        bnd_t = (A_t.abs().double().T @ B_t.double()) * 2e-6 + 1e-6
        C_t = Fn.gemm_tn(A_t.to(DEV), B_t.to(DEV), spike_side=1, spike_scale=1.0)
        chk(C_t, ref_t, bnd_t, "spike_tn side 1")
        C_t16 = Fn.gemm_tn(A_t.to(DEV), B_t.to(torch.bfloat16).to(DEV), spike_side=1, spike_scale=1.0, spike16=True)
        assert torch.equal(C_t, C_t16)
        C_t0 = Fn.gemm_tn(B_t.to(DEV), A_t.to(DEV), spike_side=0, spike_scale=1.0)
        chk(C_t0, ref_t.T, bnd_t.T, "spike_tn side 0")
        chk(Fn.gemm_tn(A_t.to(DEV), D1[:, :min(K, 260)].contiguous().to(DEV)),
            A_t.double().T @ D1[:, :min(K, 260)].double(),
            (A_t.abs().double().T @ D1[:, :min(K, 260)].abs().double()) * 2e-6 + 1e-6, "dense_tn")


# ------------------------------------------------------------------ the training step as one HIP graph
@pytest.mark.gpu
@pytest.mark.parametrize("kind,compute", [("adLIF", "fp32"), ("RadLIF", "fp32"), ("RadLIF", "bf16")])
def test_graphed_train_step_matches_eager_steps(sp, kind, compute, request):
    """sparch_amd.graph.GraphedTrainStep (zero_grad -> forward -> CE -> backward -> Adam as ONE captured HIP
    graph; dropout seeds, Adam's per-step factors and the random initial states come from device memory)
    against the same steps run eagerly: same initial parameters, same CPU generator seed (so the same initial
    states in the same order), pdrop = 0.  Non-recurrent net: parameters after 5 steps agree to fp32 rounding
    (the device evaluates Adam's bias-correction powers itself); recurrent net: losses agree step by step while
    the trajectories coincide (dyadic V is not kept by training, so only the first steps are compared)."""
    from sparch_amd.graph import GraphedTrainStep
    from sparch_amd.optim import Adam

    if compute == "bf16":  # the operand mode is read when a launch is enqueued: captured with the graph
        request.getfixturevalue("bf16_mode")
    B, T, C, sizes = 16, 25, 40, [64, 64, 20]
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, sizes[-1], (B,), generator=g).to(DEV)

    def make():
        torch.manual_seed(11)
        net = sp.SNN((B, None, C), sizes, neuron_type=kind, dropout=0.0).to(DEV).train()
        return net, Adam(net.parameters(), 1e-2)

    loss_fn = torch.nn.CrossEntropyLoss()
    steps = 5
    net_c, opt_c = make()
    torch.manual_seed(77)
    gs = GraphedTrainStep(net_c, opt_c, loss_fn, x, y, warmup=0)
    losses_g = [float(gs.step()) for _ in range(steps)]
    _Fn().check_status()
    # eager twin: construction of the graphed step consumed two forwards' worth of draws from the CPU generator
    # (the static state buffers, and the refill before capture; capture itself runs no kernel)
    net_d, opt_d = make()
    torch.manual_seed(77)
    net_d.draw_states(B, torch.device(DEV))
    net_d.draw_states(B, torch.device(DEV))
    losses_d = []
    for _ in range(steps):
        opt_d.zero_grad(set_to_none=True)
        out, rates = net_d(x)
        loss = loss_fn(out, y)
        loss.backward()
        opt_d.step()
        losses_d.append(float(loss.detach()))
    _Fn().check_status()
    assert float(opt_c.state[next(iter(net_c.parameters()))]["step"]) == steps
    n_cmp = steps if kind == "adLIF" else 2
    np.testing.assert_allclose(losses_g[:n_cmp], losses_d[:n_cmp], rtol=2e-4)
    if kind == "adLIF":
        for (k, pa), pb in zip(net_d.named_parameters(), net_c.parameters()):
            assert relmax(pb.detach().cpu().numpy(), pa.detach().cpu().numpy()) <= 2e-4, k
    assert losses_g[-1] < losses_g[0]
    gs.close()


@pytest.mark.gpu
def test_timeout_degrades_to_one_launch_per_step_and_guards_the_optimizer(sp, monkeypatch):
    """After an in-kernel timeout (simulated by raising the status word, as a recurrent kernel would): the
    optimizer step and the BatchNorm running statistics are no-ops ON THE DEVICE while the word is raised,
    `check_status(on_timeout="degrade")` reports it once and switches this process to one launch per time step,
    and a training step in that mode gives bit-identical results to the persistent launch."""
    Fn = _Fn()
    from sparch_amd.optim import Adam

    B, T, C = 8, 20, 40
    torch.manual_seed(5)
    net = sp.SNN((B, None, C), [64, 64, 20], neuron_type="RadLIF", dropout=0.0).to(DEV).train()
    opt = Adam(net.parameters(), 1e-2)
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, 20, (B,), generator=g).to(DEV)

    def step(apply):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(9)
        out, rates = net(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        if apply:
            opt.step()
        return out.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters()}

    out_a, g_a = step(False)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    Fn.status_word(DEV)[0] = 1                      # "a wait timed out"
    step(True)                                      # forward updates BN statistics, Adam steps: both must skip
    torch.cuda.synchronize()
    after = net.state_dict()
    for k, v in before.items():
        if "num_batches" not in k:
            assert torch.equal(v, after[k]), k
    monkeypatch.setattr(Fn, "_degraded", set())
    assert Fn.check_status(DEV, on_timeout="degrade") is True
    assert Fn.check_status(DEV, on_timeout="degrade") is False   # reported once, word cleared
    assert Fn.rec_steps_per_launch(T) == 1
    out_b, g_b = step(False)                        # one launch per time step now
    assert torch.equal(out_a, out_b)
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k
    step(True)
    torch.cuda.synchronize()
    assert not torch.equal(before["snn.0.W.weight"], net.state_dict()["snn.0.W.weight"])


# ------------------------------------------------------------------ shape sweeps over the round-2 code paths
@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H,bidir", [(3, 7, 32, False), (33, 11, 96, True), (70, 5, 160, False), (9, 6, 1024, True)])
def test_ligru_persistent_shape_sweep(B, T, H, bidir, monkeypatch):
    """Ragged batches (row tiles with padding rows, several row-tile groups), hidden sizes with partial k-group
    coverage per wave, the largest supported size, both directions: persistent LiGRU == launch-per-step LiGRU."""
    from sparch_amd.anns import LiGRULayer

    C = 20
    torch.manual_seed(B + H)
    layer = LiGRULayer(C, H, B, dropout=0.0, normalization="layernorm", use_bias=True, bidirectional=bidir).to(DEV).train()
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, C, generator=g).to(DEV)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g).to(DEV)

    def run(persistent):
        monkeypatch.setenv("SPARCH_LIGRU_PERSISTENT", "1" if persistent else "0")
        layer.zero_grad()
        xd = x.clone().requires_grad_(True)
        y = layer(xd)
        (y * gy).sum().backward()
        _Fn().check_status()
        return y.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu().clone() for k, v in layer.named_parameters()}

    y0, dx0, g0 = run(True)
    y1, dx1, g1 = run(False)
    assert relmax(y0.numpy(), y1.numpy()) <= 2e-5 and relmax(dx0.numpy(), dx1.numpy()) <= 5e-5
    for k in g0:
        assert relmax(g0[k].numpy(), g1[k].numpy()) <= 1e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,Bp,T,H", [("RadLIF", 3, 9, 36), ("RLIF", 70, 6, 100), ("RadLIF", 33, 5, 1028)])
def test_recurrent_step_path_shape_sweep(kind, Bp, T, H, monkeypatch):
    """The step path on hidden sizes that are not multiples of 32 (partial column tiles), ragged row tiles, and
    just above the persistent kernels' limit, against the oracle (dyadic V: spikes bit-equal)."""
    Wx, p, u0, w0, s0, g_s = _dyadic_cell_case(kind, Bp, T, H, 100 + H)
    monkeypatch.setenv("SPARCH_REC_STEP_PATH", "1")
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Wxr = Wx.clone().requires_grad_(True)
    ref = orc.spiking_cell(kind, Wxr, pr, u0, w0, s0)
    (ref * g_s).sum().backward()
    s, dwx, grads = _run_cell(kind, Wx, p, u0, w0, s0, g_s)
    assert torch.equal(s, ref.detach())
    assert relmax(dwx.numpy(), Wxr.grad.numpy()) <= 2e-4
    for k in grads:
        assert relmax(grads[k].numpy(), pr[k].grad.numpy()) <= 2e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,bidir", [("adLIF", True), ("RadLIF", True), ("RLIF", False)])
def test_bf16_saved_states_whole_network(sp, kind, bidir, monkeypatch):
    """SPARCH_SAVE_DTYPE=bf16 through whole networks (BatchNorm sums folded into the backward kernels, both
    directions): output identical, every weight / V / norm gradient identical to the fp32-saved run, neuron
    parameters within 2e-2."""
    Fn = _Fn()
    B, T, C = 10, 30, 40
    torch.manual_seed(4)
    net = sp.SNN((B, None, C), [64, 64, 20], neuron_type=kind, dropout=0.0, bidirectional=bidir).to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, T, C, generator=g) < 0.2).float().to(DEV)
    y = torch.randint(0, 20, (B,), generator=g).to(DEV)

    def run():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(6)
        out, rates = net(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        Fn.check_status()
        return out.detach().cpu(), {k: v.grad.cpu().clone() for k, v in net.named_parameters()}

    out_a, g_a = run()
    monkeypatch.setattr(Fn, "SAVE_BF16", True)
    out_b, g_b = run()
    assert torch.equal(out_a, out_b)
    for k in g_a:
        if k.split(".")[-1] in ("alpha", "beta", "a", "b") and not k.startswith("snn.2."):
            assert relmax(g_b[k].numpy(), g_a[k].numpy()) <= 2e-2, k
        else:
            assert torch.equal(g_a[k], g_b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("bidir,B,T,H", [(False, 40, 23, 128), (True, 36, 17, 256)])
def test_gru_persistent_kernels_vs_launch_per_step_and_chunked(bidir, B, T, H, monkeypatch):
    """GRU on the persistent kernels (csrc/gatedcell.hip: 16 hidden units per workgroup, two hand-offs per step —
    q = r y before the candidate's product, [dz_pre | dr_pre] after dq = dc_pre V in the backward) against the
    launch-per-step path (same arithmetic, different summation order), with dropout, and chunked launches of the
    persistent kernels against the whole-sequence launch (bit for bit)."""
    from sparch_amd.anns import GRULayer

    C = 48
    torch.manual_seed(41)
    layer = GRULayer(C, H, B, dropout=0.1, normalization="batchnorm", use_bias=False, bidirectional=bidir).to(DEV).train()
    g = torch.Generator().manual_seed(42)
    x = torch.randn(B, T, C, generator=g).to(DEV)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g).to(DEV)

    def run(persistent, spl):
        monkeypatch.setenv("SPARCH_GRU_PERSISTENT", "1" if persistent else "0")
        monkeypatch.setenv("SPARCH_REC_STEPS_PER_LAUNCH", spl)
        torch.manual_seed(7)
        layer._calls = 0  # same dropout seed (initial seed mixed with the layer's call count) in every run
        layer.zero_grad()
        xd = x.clone().requires_grad_(True)
        y = layer(xd)
        (y * gy).sum().backward()
        _Fn().check_status()
        return y.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu().clone() for k, v in layer.named_parameters()}

    y0, dx0, g0 = run(True, "")
    y1, dx1, g1 = run(False, "")
    assert float(y0.abs().max()) > 0
    assert relmax(y0.numpy(), y1.numpy()) <= 1e-5 and relmax(dx0.numpy(), dx1.numpy()) <= 5e-5
    for k in g0:
        assert relmax(g0[k].numpy(), g1[k].numpy()) <= 1e-4, k
    for spl in ("5", "1"):
        y2, dx2, g2 = run(True, spl)
        assert torch.equal(y2, y0) and torch.equal(dx2, dx0), spl
        for k in g0:
            assert torch.equal(g2[k], g0[k]), (spl, k)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H,bidir", [(3, 7, 32, False), (33, 11, 96, True), (70, 5, 160, False), (9, 6, 1024, True)])
def test_gru_persistent_shape_sweep_vs_oracle(B, T, H, bidir, monkeypatch):
    """Ragged batches, partial k-group coverage per wave, the largest supported size, both directions: the
    persistent GRU against the CPU oracle (oracle/ann_oracle.py, pinned by tests/golden/ann_GRU*.npz) and against
    the launch-per-step path."""
    from oracle import ann_oracle as ao
    from sparch_amd.anns import GRULayer

    C = 20
    torch.manual_seed(B + H)
    layer = GRULayer(C, H, B, dropout=0.0, normalization="layernorm", use_bias=True, bidirectional=bidir).train()
    p = {"ann.0." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(DEV)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, C, generator=g)
    gy = torch.randn(B, T, H * (2 if bidir else 1), generator=g)

    def run(persistent):
        monkeypatch.setenv("SPARCH_GRU_PERSISTENT", "1" if persistent else "0")
        layer.zero_grad()
        xd = x.to(DEV).requires_grad_(True)
        y = layer(xd)
        (y * gy.to(DEV)).sum().backward()
        _Fn().check_status()
        return y.detach().cpu(), xd.grad.cpu(), {k: v.grad.cpu().clone() for k, v in layer.named_parameters()}

    y0, dx0, g0 = run(True)
    y1, dx1, g1 = run(False)
    assert relmax(y0.numpy(), y1.numpy()) <= 2e-5 and relmax(dx0.numpy(), dx1.numpy()) <= 5e-5
    for k in g0:
        assert relmax(g0[k].numpy(), g1[k].numpy()) <= 1e-4, k
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    ref = ao.hidden_layer("GRU", xr, pr, "ann.0", "layernorm", bidir, training=True, running=None)
    (ref * gy).sum().backward()
    assert relmax(y0.numpy(), ref.detach().numpy()) <= 2e-5
    assert relmax(dx0.numpy(), xr.grad.numpy()) <= 2e-4
    for k, v in g0.items():
        assert relmax(v.numpy(), pr["ann.0." + k].grad.numpy()) <= 2e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RLIF", "RadLIF"])
def test_xcd_local_handoff_stores_give_identical_results(kind):
    """The recurrent kernels' XCD-local hand-off (plain stores among the workgroups of a row tile once they have
    established, through agent-scope accesses, that they share an XCD) against the agent-scope write-through
    stores: same spikes, same gradients, bit for bit — whole-sequence launches with several row-tile groups and
    real-valued V, so that any stale or missed tile would change the result."""
    from sparch_amd._capi import lib

    Fn = _Fn()
    Bp, T, H = 300, 40, 1024  # 10 row tiles x 32 column tiles: two co-resident launches on 256 CUs
    g = torch.Generator().manual_seed(17)
    Wx = (torch.randn(Bp, T, H, generator=g) * 2.0 + 0.6).to(DEV)
    V = torch.nn.init.orthogonal_(torch.empty(H, H), generator=g).to(DEV)
    adaptive = kind == "RadLIF"
    p = dict(alpha=torch.rand(H, generator=g) * 0.14 + 0.82, beta=torch.rand(H, generator=g) * 0.024 + 0.967,
             a=torch.rand(H, generator=g) * 2 - 1, b=torch.rand(H, generator=g) * 2)
    p = {k: v.to(DEV) for k, v in p.items()}
    u0, w0, s0 = (torch.rand(Bp, H, generator=g).to(DEV) for _ in range(3))
    gs = torch.randn(Bp, T, H, generator=g).to(DEV)

    def run(on):
        lib.sparch_set_xcd_local(on)
        try:
            Wr = Wx.clone().requires_grad_(True)
            Vr = V.clone().requires_grad_(True)
            s = Fn.SpikingCellFn.apply(kind, 1.0, Wr, p["alpha"], p["beta"] if adaptive else None,
                                       p["a"] if adaptive else None, p["b"] if adaptive else None, Vr, u0,
                                       w0 if adaptive else None, s0, None)
            (s * gs).sum().backward()
            Fn.check_status()
            return s.detach().clone(), Wr.grad.clone(), Vr.grad.clone()
        finally:
            lib.sparch_set_xcd_local(-1)  # back to the environment's setting

    s1, dw1, dv1 = run(1)
    s0_, dw0, dv0 = run(0)
    assert float(s1.mean()) > 0.005
    assert torch.equal(s1, s0_) and torch.equal(dw1, dw0) and torch.equal(dv1, dv0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C", [(3, 1, 5), (5, 257, 7), (2, 1000, 35), (4, 300, 256), (7, 513, 1), (9, 64, 64)])
def test_readout_cell_shape_sweep_vs_oracle(B, T, C):
    """The staged readout kernels across their chunking rules: sequences longer than one 256-step chunk, chunk
    lengths bounded by LDS at many classes, a single step, a single class (odd / even row strides, ragged last
    staging sweep), against the oracle; the value is bit-exact in u (serial recurrence, same order) so the output
    may differ only by the softmax's rounding."""
    Fn = _Fn()
    g = torch.Generator().manual_seed(B * 1000 + T + C)
    Wx = torch.randn(B, T, C, generator=g) * 1.5
    alpha = torch.rand(C, generator=g) * 0.3 + 0.72
    u0 = torch.rand(B, C, generator=g)
    g_out = torch.randn(B, C, generator=g)
    Wr, ar = Wx.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref = orc.readout_cell(Wr, ar, u0)
    (ref * g_out).sum().backward()
    Wd, ad = Wx.to(DEV).requires_grad_(True), alpha.to(DEV).requires_grad_(True)
    out = Fn.ReadoutCellFn.apply(Wd, ad, u0.to(DEV))
    (out * g_out.to(DEV)).sum().backward()
    Fn.check_status()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=3e-5, atol=3e-5 * max(1, T / 100))
    assert relmax(Wd.grad.cpu().numpy(), Wr.grad.numpy()) <= 2e-4
    assert relmax(ad.grad.cpu().numpy(), ar.grad.numpy()) <= 5e-4
