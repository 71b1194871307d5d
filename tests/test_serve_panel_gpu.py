"""Row-panel GEMM of the serving prefill (csrc/serve_panel.hip) against the generic path it replaces: lap_layernorm_fwd /
lap_rmsnorm_fwd + lap_gemm_bf16_ex.  Same MFMA, same k order, same epilogue arithmetic: the unsplit form must be BITWISE equal;
the split form (f32 partials + lap_fused_reduce_norm) is held to the f32 product at bf16 output rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(*shape, dtype=torch.bfloat16, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_err(a, b):
    a = a.float(); b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


# SigLIP So400m block at 2 x 256 tokens (qkv, out, fc1 with the engine's padded MLP width), Gemma-2B prefix rows (560: ragged last panel)
@pytest.mark.parametrize("M,N,K", [(512, 3456, 1152), (512, 1152, 1152), (512, 4352, 1152), (560, 2560, 2048), (560, 2048, 2048),
                                   (50, 1024, 1024), (520, 144, 64), (256, 3456, 1152), (768, 1152, 1152)])   # one / three camera images
@pytest.mark.parametrize("nt", [0, 1, 2, 3, 4])
def test_panel_plain_bitwise(hip, M, N, K, nt):
    x = rnd(M, K); w = rnd(N, K, seed=1, scale=0.05)
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    ref = hip.linear_fwd(x, w, tile=6, ksplit=1)       # the unsplit 128 x 128 HIP tile (tile id 6)
    out = hip.panel_linear(x, wp, N, nt=nt)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K", [(512, 3456, 1152), (512, 4352, 1152)])
def test_panel_layernorm_bias_gelu_bitwise(hip, M, N, K):
    x = rnd(M, K, scale=2.0); w = rnd(N, K, seed=1, scale=0.05)
    g = rnd(K, dtype=torch.float32, seed=2) * 0.2 + 1.0
    b = rnd(K, dtype=torch.float32, seed=3) * 0.1
    bias = rnd(N, dtype=torch.float32, seed=4) * 0.3
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    y, _, _ = hip.layernorm_fwd(x, g, b)
    for gelu in (False, "bf16"):
        ref = hip.linear_fwd(y, w, bias=bias, gelu=gelu, tile=6, ksplit=1)
        out = hip.panel_linear(x, wp, N, bias=bias, norm=2, gamma=g, beta=b, gelu=gelu)
        assert torch.equal(out, ref), gelu


def test_panel_bias_residual_bitwise(hip):
    M, N, K = 512, 1152, 1152
    x = rnd(M, K); w = rnd(N, K, seed=1, scale=0.05); r = rnd(M, N, seed=5)
    bias = rnd(N, dtype=torch.float32, seed=4) * 0.3
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    ref = hip.linear_fwd(x, w, bias=bias, residual=r, tile=6, ksplit=1)
    out = hip.panel_linear(x, wp, N, bias=bias, residual=r)
    assert torch.equal(out, ref)


def test_panel_rmsnorm_bitwise(hip):
    M, N, K = 560, 2560, 2048
    x = rnd(M, K, scale=3.0); w = rnd(N, K, seed=1, scale=0.05)
    sc = rnd(K, dtype=torch.float32, seed=2) * 0.2
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    y, _ = hip.rmsnorm_fwd(x, scale=sc, save_rstd=False)
    ref = hip.linear_fwd(y, w, tile=6, ksplit=1)
    out = hip.panel_linear(x, wp, N, norm=1, gamma=sc)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K,ks", [(512, 1152, 4352, 4), (560, 2048, 16384, 8), (560, 2048, 2048, 2)])
def test_panel_partials_with_the_fused_consumer(hip, M, N, K, ks):
    x = rnd(M, K); w = rnd(N, K, seed=1, scale=0.05); r = rnd(M, N, seed=5)
    bias = rnd(N, dtype=torch.float32, seed=4) * 0.3
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    scratch = torch.empty(ks * M * N, dtype=torch.float32, device=DEV)
    part, k2 = hip.panel_partials(x, wp, N, scratch, ks)
    assert k2 == ks
    kk = K // ks
    for s in range(ks):     # every slab is the f32 product of its K slice (f32 accumulation: summation order only)
        ref = x[:, s * kk:(s + 1) * kk].float() @ w[:, s * kk:(s + 1) * kk].float().t()
        assert rel_err(part[s], ref) < 2e-5
    xn, _ = hip.fused_reduce_norm(part, ks, M, N, bias=bias, residual=r, norm=0)
    ref = (x.float() @ w.float().t() + bias + r.float())
    assert rel_err(xn, ref) < 4e-3     # one bf16 rounding of the output


def test_panel_rejects_what_it_does_not_serve(hip):
    assert not hip.panel_gemm_ok(512, 1152, 4352, 1)      # K too long for one panel
    assert hip.panel_gemm_ok(512, 1152, 4352, 4)
    assert not hip.panel_gemm_ok(512, 1150, 1152, 1)      # N % 16
    assert not hip.panel_gemm_ok(512, 1152, 1120, 1)      # K % 64
    x = rnd(64, 4352); wp = torch.empty(1152 * 4352, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(hip.LapHipError):
        hip.panel_linear(x, wp, 1152)


def test_panel_prefetch_wave_changes_nothing(hip):
    """The fifth wave that reads the next launch's weights (and discards them) must leave the product bit for bit alone —
    also when its range is ragged or larger than one pass."""
    M, N, K = 512, 3456, 1152
    x = rnd(M, K); w = rnd(N, K, seed=1, scale=0.05)
    bias = rnd(N, dtype=torch.float32, seed=4) * 0.3
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    ref = hip.panel_linear(x, wp, N, bias=bias, nt=2)
    for nbytes in (16, 1000 * 16, 10 * 1024 * 1024 + 48, 67 * 1024 * 1024):
        nxt = torch.randn(nbytes // 2, device=DEV).to(torch.bfloat16)
        keep = nxt.clone()
        out = hip.panel_linear(x, wp, N, bias=bias, nt=2, prefetch=nxt)
        assert torch.equal(out, ref), nbytes
        assert torch.equal(nxt, keep)
    scratch = torch.empty(4 * M * 1152, dtype=torch.float32, device=DEV)
    x2 = rnd(M, 4352); w2 = rnd(1152, 4352, seed=2, scale=0.05)
    wp2 = hip.serve_pack_weight(w2, hip.PACK_PLAIN)
    p0 = hip.panel_partials(x2, wp2, 1152, scratch, 4, nt=3)[0].clone()
    p1 = hip.panel_partials(x2, wp2, 1152, scratch, 4, nt=3, prefetch=wp)[0]
    assert torch.equal(p0, p1)


def test_panel_exp2_gelu_is_the_training_kernels_gelu(hip):
    """gelu="exp2": the sigmoid form through v_exp / v_rcp.  Bitwise the activation lap_gemm_asm_bias_gelu (the training step's SigLIP
    fc1) stores for the same pre-activation, and within one bf16 ulp of the tanhf form the generic path uses."""
    M, N, K = 512, 4352, 1152
    x = rnd(M, K); w = rnd(N, K, seed=1, scale=0.05)
    bias = rnd(N, dtype=torch.float32, seed=4) * 0.3
    wp = hip.serve_pack_weight(w, hip.PACK_PLAIN)
    out = hip.panel_linear(x, wp, N, bias=bias, gelu="exp2")
    tanh_form = hip.panel_linear(x, wp, N, bias=bias, gelu="bf16")
    # one bf16 ulp = 2^-8 relative; near zero the absolute difference is what counts
    d = (out.float() - tanh_form.float()).abs()
    assert bool((d <= tanh_form.float().abs() * 2 ** -7 + 1e-6).all())
    assert (out != tanh_form).float().mean().item() < 0.02
    if hip.linear_bias_gelu_train_ok(x, w, bias):
        h, a = hip.linear_bias_gelu_train(x, w, bias)
        assert torch.equal(h, hip.panel_linear(x, wp, N, bias=bias))
        assert torch.equal(a, out)
    else:
        pytest.skip("assembly bias + GELU kernel not available for this shape")
