"""GPU parity of the three core kernels (GEMM, LayerNorm, attention) against fp32 references.

GEMM / attention take fp16 operands and accumulate in fp32, so the reference is the fp32 op evaluated
on the SAME fp16-rounded operands; tolerances below are the fp16 output rounding (2^-11 relative) plus
accumulation-order noise.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1297, 768, 768), (300, 256, 640), (4096, 1280, 1280)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm(dev, M, N, K, epi):
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 1).half()
    w = _rand((N, K), dev, 0.05, 2).half()  # asymmetric, transposition-detecting
    bias = _rand((N,), dev, 0.5, 3)
    ref = a.float() @ w.float().t() + bias
    if epi == 0:
        out = ops.gemm(a, w, bias, epilogue=ops.EPI_F16)
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
    elif epi == 1:
        out = ops.gemm(a, w, bias, epilogue=ops.EPI_GELU_F16)
        torch.testing.assert_close(out.float(), torch.nn.functional.gelu(ref), rtol=2e-3, atol=2e-3)
    else:
        resid = _rand((M, N), dev, 1.0, 4)
        gamma = _rand((N,), dev, 1.0, 5)
        out = ops.gemm(a, w, bias, epilogue=ops.EPI_F32, resid=resid, gamma=gamma)
        torch.testing.assert_close(out, resid + gamma * ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("tile", [1, 11, 15, 16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1297, 768, 768), (4096, 1280, 1280), (1000, 512, 192)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_large_tiles(dev, tile, M, N, K, epi):
    """Every GEMM kernel forced through psam_gemm_set_tile: 1 / 11 HIP, 15 / 16 assembly (a tile that cannot take a shape falls back)."""
    from protosam_amd import ops
    if K % 64:
        pytest.skip("the C ABI requires K % 64 == 0 for every tile")
    a = _rand((M, K), dev, 1.0, 11).half()
    w = _rand((N, K), dev, 0.05, 12).half()
    bias = _rand((N,), dev, 0.5, 13)
    ref = a.float() @ w.float().t() + bias
    ops.gemm_set_tile(tile)
    try:
        if epi == 0:
            out = ops.gemm(a, w, bias, epilogue=ops.EPI_F16)
            torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
        elif epi == 1:
            out = ops.gemm(a, w, bias, epilogue=ops.EPI_GELU_F16)
            torch.testing.assert_close(out.float(), torch.nn.functional.gelu(ref), rtol=2e-3, atol=2e-3)
        else:
            resid = _rand((M, N), dev, 1.0, 14)
            gamma = _rand((N,), dev, 1.0, 15)
            out = ops.gemm(a, w, bias, epilogue=ops.EPI_F32, resid=resid, gamma=gamma)
            torch.testing.assert_close(out, resid + gamma * ref, rtol=1e-4, atol=2e-4)
    finally:
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("M,N,K", [(1297, 768, 768), (1297, 768, 3072), (1297, 2304, 768), (1297, 3072, 768), (4096, 768, 3072), (4096, 256, 2304),
                                   (130, 128, 256), (5330, 768, 64), (5330, 768, 128), (300, 256, 192)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_deep_ring(dev, M, N, K, epi):
    """Tiles 12 and 13 (round 5). Tile 12 ( the 128x128 kernel with a four-deep K-tile ring and counted waits, taken automatically where a launch leaves
    at most one workgroup per CU - one DINOv2 slice, one SAM ViT-B image) is BIT-IDENTICAL to tile 1 (same MFMAs, k order and epilogue
    code) and right against the fp32 arithmetic: K of 1 ... 48 K-tiles (the ring's fill and drain paths), ragged M, in-place residual."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 61).half()
    w = _rand((N, K), dev, 0.05, 62).half()
    bias = _rand((N,), dev, 0.5, 63)
    gamma = _rand((N,), dev, 1.0, 65)
    ref = a.float() @ w.float().t() + bias
    outs = []
    for tile in (12, 1, 0, 13):
        ops.gemm_set_tile(tile)
        try:
            if epi == 2:
                x = _rand((M, N), dev, 1.0, 64)
                outs.append(ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, gamma=gamma))     # x += gamma * (a w^T + b)
            else:
                outs.append(ops.gemm(a, w, bias, epilogue=epi))
        finally:
            ops.gemm_set_tile(0)
    assert torch.equal(outs[0], outs[1])
    if epi == 2:
        torch.testing.assert_close(outs[0], _rand((M, N), dev, 1.0, 64) + gamma * ref, rtol=1e-4, atol=2e-4)
    else:
        torch.testing.assert_close(outs[0].float(), ref if epi == 0 else torch.nn.functional.gelu(ref), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(outs[2].float(), outs[0].float(), rtol=2e-3, atol=2e-3)      # (auto may pick another family)
    # tile 13 (64x64 tiles for launches that would leave most CUs idle): same k order, its own epilogue code
    torch.testing.assert_close(outs[3].float(), outs[0].float(), rtol=1e-3 if epi != 2 else 1e-5, atol=1e-3 if epi != 2 else 1e-5)


@pytest.mark.parametrize("M,N,K", [(20000, 1536, 128), (70001, 768, 64), (33000, 2304, 192)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_persistent_tile_walks_many_tiles(dev, M, N, K, epi):
    """Tiles 11 / 15 (one persistent workgroup per CU) with more 256x256 tiles than CUs and a ragged last row of tiles: the same
    arithmetic in the same order => bit-identical outputs, and correct against fp32 matmul."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 21).half()
    w = _rand((N, K), dev, 0.05, 22).half()
    bias = _rand((N,), dev, 0.5, 23)
    e = (ops.EPI_F16, ops.EPI_GELU_F16, ops.EPI_F32)[epi]
    resid = _rand((M, N), dev, 1.0, 24) if epi == 2 else None
    gamma = _rand((N,), dev, 1.0, 25) if epi == 2 else None
    outs = []
    for tile in (11, 15):   # 15 = the assembly kernels (csrc/gemm_asm_gen.py): same MFMA, same k order, same epilogue arithmetic
        ops.gemm_set_tile(tile)
        try:
            if epi == 2:
                x = resid.clone()                                   # in place, as the encoder blocks call it
                outs.append(ops.gemm(a, w, bias, out=x, epilogue=e, resid=x, gamma=gamma))
            else:
                outs.append(ops.gemm(a, w, bias, epilogue=e))
        finally:
            ops.gemm_set_tile(0)
    if epi == 1:
        # (round 5: the assembly kernel's GELU is x / (1 + exp2(x Q(x^2))), within 3.4e-6 of the erf form the HIP kernels evaluate - an
        # fp16 result differs in its last place where that shift crosses a rounding boundary: ~7 % of the outputs, whose ulp around
        # |y| = 0.1 is 6e-5)
        d = (outs[0].float() - outs[1].float()).abs()
        assert (d <= outs[0].float().abs() * 2.0 ** -10 + 5e-6).all() and (d > 0).float().mean().item() < 0.15
    else:
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = a.float() @ w.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        torch.testing.assert_close(outs[1], resid + gamma * ref, rtol=1e-4, atol=2e-4)
    else:
        torch.testing.assert_close(outs[1].float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(300, 384, 1344), (4096, 1280, 1280), (2000, 640, 5120)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_half_tile_pingpong(dev, M, N, K, epi):
    """Tile 16 (assembly, 256x128 half-tiles, epilogue scheduled under the next half-tile's MFMAs, bias through the accumulator
    initialisation, N in blocks of 128): against the fp32 product and within rounding of tile 15."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 61).half()
    w = _rand((N, K), dev, 0.05, 62).half()
    bias = _rand((N,), dev, 0.5, 63)
    resid = _rand((M, N), dev, 1.0, 64) if epi == 2 else None
    gamma = _rand((N,), dev, 1.0, 65) if epi == 2 else None
    e = (ops.EPI_F16, ops.EPI_GELU_F16, ops.EPI_F32)[epi]
    outs = []
    for tile in (16, 15 if N % 256 == 0 else 1):
        ops.gemm_set_tile(tile)
        try:
            x = resid.clone() if epi == 2 else None
            outs.append(ops.gemm(a, w, bias, out=x, epilogue=e, resid=x, gamma=gamma))
        finally:
            ops.gemm_set_tile(0)
    ref = a.float() @ w.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        torch.testing.assert_close(outs[0], resid + gamma * ref, rtol=1e-4, atol=3e-4)
        torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=5e-5)
    else:
        torch.testing.assert_close(outs[0].float(), ref, rtol=2e-3, atol=2e-3)
        assert (outs[0].float() - outs[1].float()).abs().max().item() <= 2 ** -7   # one fp16 ulp at |x| < 16


def test_gemm_row_remap_and_resid_mod(dev):
    """patch-embed style: rows of batch b land at b*stride + off + p, resid (pos-embed) indexed by p."""
    from protosam_amd import ops
    B, P, N, K = 3, 100, 128, 64
    a = _rand((B * P, K), dev, 1.0, 1).half()
    w = _rand((N, K), dev, 0.1, 2).half()
    bias = _rand((N,), dev, 0.5, 3)
    pos = _rand((P, N), dev, 1.0, 4)
    out = torch.zeros((B, P + 1, N), device=dev)
    ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=pos, resid_mod=P, out_seg=P, out_seg_stride=P + 1,
             out_seg_off=1)
    ref = (a.float() @ w.float().t() + bias).view(B, P, N) + pos
    torch.testing.assert_close(out[:, 1:], ref, rtol=1e-4, atol=2e-4)
    assert float(out[:, 0].abs().max()) == 0.0


def test_gemm_inplace_residual(dev):
    from protosam_amd import ops
    M, N, K = 257, 256, 128
    a = _rand((M, K), dev, 1.0, 1).half()
    w = _rand((N, K), dev, 0.1, 2).half()
    x = _rand((M, N), dev, 1.0, 3)
    ref = x + a.float() @ w.float().t()
    ops.gemm(a, w, None, out=x, epilogue=ops.EPI_F32, resid=x)
    torch.testing.assert_close(x, ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("M,N,K,ks", [(4096, 1280, 5120, 3), (4000, 1280, 5120, 3), (2048, 2048, 6144, 4), (8192, 256, 16384, 8),
                                      (1297, 768, 3072, 1), (4096, 1280, 1280, 1)])
def test_gemm_splitk_residual(dev, request, M, N, K, ks):
    """The residual update of few 256-tiles over a long K (one slice through fc2) takes the split-K form of the persistent
    kernel: several workgroups per tile write partial sums, a reduce pass applies bias, LayerScale and the residual in a fixed
    order. Against a float64 product and against the single-pass 128x128 kernel (forced tile 1: no split).
    `ks` = the split the dispatch rule picks for the shape (1: stays on the single-pass kernel)."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 1).half()
    w = _rand((N, K), dev, 0.05, 2).half()
    bias = _rand((N,), dev, 1.0, 4)
    gamma = _rand((N,), dev, 1.0, 5)
    x0 = _rand((M, N), dev, 1.0, 3)
    ref = (x0.double() + gamma.double() * (a.double() @ w.double().t() + bias.double())).float()
    ops.gemm_set_option("half_tiles", 0)     # (the half-tile assembly kernels would take these shapes first)
    request.addfinalizer(lambda: ops.gemm_set_option("half_tiles", 1))
    x = x0.clone()
    ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, gamma=gamma)
    y = x0.clone()
    ops.gemm_set_tile(1)
    try:
        ops.gemm(a, w, bias, out=y, epilogue=ops.EPI_F32, resid=y, gamma=gamma)
    finally:
        ops.gemm_set_tile(0)
    tol = 2e-4 * (K / 1280) ** 0.5
    e_split, e_plain = (x - ref).abs().max().item(), (y - ref).abs().max().item()
    print(f"{M}x{N}x{K}: max err vs float64: auto {e_split:.2e}, 128-tile {e_plain:.2e}; auto vs 128-tile {(x - y).abs().max().item():.2e}")
    assert e_split < tol and e_plain < tol
    # deterministic: a second run is bit-identical; so is the out-of-place form (resid != out) and a row-periodic residual
    x2 = x0.clone()
    ops.gemm(a, w, bias, out=x2, epilogue=ops.EPI_F32, resid=x2, gamma=gamma)
    assert torch.equal(x2, x)
    z = torch.empty_like(x0)
    ops.gemm(a, w, bias, out=z, epilogue=ops.EPI_F32, resid=x0, gamma=gamma)
    assert torch.equal(z, x)
    per = 100
    zp = torch.empty_like(x0)
    ops.gemm(a, w, None, out=zp, epilogue=ops.EPI_F32, resid=x0[:per].contiguous(), resid_mod=per)
    refp = (x0[:per].double().repeat((M + per - 1) // per, 1)[:M] + a.double() @ w.double().t()).float()
    assert (zp - refp).abs().max().item() < tol


def test_gemm_splitk_two_streams_share_the_workspace(dev, request):
    """Two streams taking the split-K path at the same time (ProtoSAM.overlap_streams runs the two encoders on two streams): the
    library orders the users of the per-device workspace through an event, so both results equal the single-stream ones."""
    from protosam_amd import ops
    ops.gemm_set_option("half_tiles", 0)
    request.addfinalizer(lambda: ops.gemm_set_option("half_tiles", 1))
    M, N, K = 4096, 1280, 5120
    a1, a2 = _rand((M, K), dev, 1.0, 71).half(), _rand((M, K), dev, 1.0, 72).half()
    w = _rand((N, K), dev, 0.05, 73).half()
    x0 = _rand((M, N), dev, 1.0, 74)
    r1, r2 = x0.clone(), x0.clone()
    ops.gemm(a1, w, None, out=r1, epilogue=ops.EPI_F32, resid=r1)
    ops.gemm(a2, w, None, out=r2, epilogue=ops.EPI_F32, resid=r2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        y1, y2 = x0.clone(), x0.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            ops.gemm(a1, w, None, out=y1, epilogue=ops.EPI_F32, resid=y1)
        with torch.cuda.stream(s2):
            ops.gemm(a2, w, None, out=y2, epilogue=ops.EPI_F32, resid=y2)
        torch.cuda.synchronize()
        assert torch.equal(y1, r1) and torch.equal(y2, r2)


def test_gemm_splitk_needs_a_registered_workspace(dev, request):
    """Without a workspace (psam_gemm_set_workspace(NULL)) the fc2-per-slice shape stays on the single-pass kernel: the result is
    bit-identical to the forced 128-tile kernel; registering a workspace again switches the split-K path back on."""
    from protosam_amd import _lib, ops
    M, N, K = 4096, 1280, 5120
    a = _rand((M, K), dev, 1.0, 1).half()
    w = _rand((N, K), dev, 0.05, 2).half()
    x0 = _rand((M, N), dev, 1.0, 3)
    ops.gemm_set_option("half_tiles", 0)
    request.addfinalizer(lambda: ops.gemm_set_option("half_tiles", 1))
    ops.gemm(a, w, None, out=x0.clone(), epilogue=ops.EPI_F32, resid=x0.clone())       # (registers the default workspace)
    y = x0.clone()
    ops.gemm_set_tile(1)
    try:
        ops.gemm(a, w, None, out=y, epilogue=ops.EPI_F32, resid=y)
    finally:
        ops.gemm_set_tile(0)
    ws = ops._GEMM_WS[x0.device.index]
    try:
        _lib.check(_lib.lib().psam_gemm_set_workspace(0, 0), "psam_gemm_set_workspace")
        x = x0.clone()
        ops.gemm(a, w, None, out=x, epilogue=ops.EPI_F32, resid=x)
        assert torch.equal(x, y)
        assert _lib.lib().psam_gemm_set_workspace(ws.data_ptr() + 4, 1024) == 1      # misaligned: bad argument
    finally:
        _lib.check(_lib.lib().psam_gemm_set_workspace(ws.data_ptr(), ws.numel()), "psam_gemm_set_workspace")
    x = x0.clone()
    ops.gemm(a, w, None, out=x, epilogue=ops.EPI_F32, resid=x)
    assert not torch.equal(x, y) and (x - y).abs().max().item() < 2e-4                # split-K: another summation order


@pytest.mark.parametrize("M,D", [(5, 256), (1297, 768), (4096, 1280), (33, 1024)])
@pytest.mark.parametrize("eps", [1e-6, 1e-5])
def test_layernorm(dev, M, D, eps):
    from protosam_amd import ops
    x = _rand((M, D), dev, 3.0, 1) + 0.7
    w = _rand((D,), dev, 1.0, 2)
    b = _rand((D,), dev, 1.0, 3)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, eps)
    y32 = ops.layernorm(x, w, b, eps, out_dtype=torch.float32)
    torch.testing.assert_close(y32, ref, rtol=1e-5, atol=1e-5)
    y2 = torch.empty_like(x)
    y16 = ops.layernorm(x, w, b, eps, out_dtype=torch.float16, out2=y2, zero_tail_rows=1)
    assert y16.shape == (M + 1, D)
    torch.testing.assert_close(y16[:M].float(), ref, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(y2, ref, rtol=1e-5, atol=1e-5)
    assert float(y16[M].abs().max()) == 0.0


def _ref_attn_global(qkv, B, N, H, hd, scale, rel=None):
    q, k, v = qkv.float().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-2, -1)
    if rel is not None:
        att = att + rel
    return (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * hd)


@pytest.mark.parametrize("N,H,hd,B", [(1297, 12, 64, 2), (64, 2, 64, 1), (200, 3, 80, 1)])
def test_attention_global(dev, N, H, hd, B):
    from protosam_amd import ops
    qkv = _rand((B, N, 3, H, hd), dev, 1.0, 7).half()
    scale = hd ** -0.5
    out = ops.attention(qkv, B, N, H, hd, scale)
    ref = _ref_attn_global(qkv, B, N, H, hd, scale)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_attention_global_spiky_rows(dev):
    """forces the online-softmax rescale path: one key dominates late in the sequence."""
    from protosam_amd import ops
    B, N, H, hd = 1, 512, 1, 64
    qkv = _rand((B, N, 3, H, hd), dev, 0.3, 9)
    qkv[0, 400, 1, 0] = qkv[0, 17, 0, 0] * 40.0  # key 400 aligned with query 17
    qkv = qkv.half()
    out = ops.attention(qkv, B, N, H, hd, hd ** -0.5)
    ref = _ref_attn_global(qkv, B, N, H, hd, hd ** -0.5)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("H,hd", [(2, 64), (2, 80)])
def test_attention_global_relpos(dev, H, hd):
    from oracle.sam_image_encoder import decomposed_rel_pos_terms
    from protosam_amd import ops
    B, g = 1, 64
    N = g * g
    qkv = _rand((B, N, 3, H, hd), dev, 1.0, 11).half()
    Rh = _rand((2 * g - 1, hd), dev, 0.3, 12)
    Rw = _rand((2 * g - 1, hd), dev, 0.3, 13)
    scale = hd ** -0.5
    rel_h, rel_w = ops.relpos(qkv, ops.pack_rel_tables(Rh, Rw, False, hd), B, N, H, hd, g, g, False, scale)
    q = qkv.float().view(B, N, 3, H, hd)[:, :, 0].permute(0, 2, 1, 3).reshape(B * H, N, hd)
    rh_ref, rw_ref = decomposed_rel_pos_terms(q.cpu(), Rh.cpu(), Rw.cpu(), (g, g))
    torch.testing.assert_close(rel_h.cpu().view(B * H, g, g, g), rh_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rel_w.cpu().view(B * H, g, g, g), rw_ref, rtol=1e-4, atol=1e-4)
    out = ops.attention(qkv, B, N, H, hd, scale, mode=1, rel_h=rel_h, rel_w=rel_w, gh=g, gw=g)
    bias = (rh_ref[..., :, None] + rw_ref[..., None, :]).reshape(B, H, N, N).to(dev)
    ref = _ref_attn_global(qkv, B, N, H, hd, scale, rel=bias)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,g,H,hd", [(2, 64, 2, 64), (2, 64, 2, 80), (3, 32, 3, 80), (1, 20, 1, 64), (5, 30, 16, 80),
                                      (1, 24, 2, 80),    # (24 x 24 = 576 tokens: psam_relpos' one-tile-per-wave form)
                                      (3, 64, 16, 80)])  # (1200 items: several per workgroup of the persistent kernels, every edge class)
def test_attention_window_relpos(dev, B, g, H, hd):
    """14x14 windows over a g x g map, zero-padded tokens carry the qkv bias (image_encoder.py:267-271): the two-kernel path
    (psam_relpos -> relq) and the window kernels of the fused path (attn_kernel, wattn_kernel, the assembly kernel of
    csrc/wattn_asm_gen.py - hd = 80 on the 64 x 64 map -, the persistent wattn_p_kernel) against the reference arithmetic. Shapes: the 64 x 64 map of a 1024 input; 32 x 32 (nine windows, three of
    them ragged on two sides); a map smaller than two windows; item counts below / not a multiple of the XCD and CU counts."""
    from oracle.sam_image_encoder import decomposed_rel_pos_terms, window_partition, window_unpartition
    from protosam_amd import ops
    ws = 14
    N, C = g * g, H * hd
    qkv = _rand((B, N, 3, H, hd), dev, 1.0, 21)
    # rows whose maximum jumps late in the key order (the lazy-rescale branch, guide rule 26): the LAST key of window 0 (the
    # 16-key tail tile) aligned with query 0, and a key in the middle of the window aligned with query 17
    last = min(13, g - 1) * g + min(13, g - 1)
    qkv[0, last, 1, 0] = qkv[0, 0, 0, 0] * 25.0
    qkv[0, 5 * g + 6, 1, 0] = qkv[0, g + 3, 0, 0] * 40.0
    qkv = qkv.half()
    pad = _rand((3, H, hd), dev, 1.0, 22).half()
    Rh = _rand((2 * ws - 1, hd), dev, 0.3, 23)
    Rw = _rand((2 * ws - 1, hd), dev, 0.3, 24)
    scale = hd ** -0.5
    rp = ops.pack_rel_tables(Rh, Rw, True, hd)
    outs = {}
    if N % 64 == 0:                                   # psam_relpos works on 64-token blocks
        relq = ops.relpos(qkv, rp, B, N, H, hd, g, ws, True, scale)
        outs["relq"] = ops.attention(qkv, B, N, H, hd, scale, mode=2, relq=relq, pad_row=pad, gh=g, gw=g, ws=ws)
    try:
        for v in (1, 3, 5, 7):
            ops.attention_set_variant(v)
            o = torch.full((B, N, C), float("nan"), device=dev, dtype=torch.float16)
            outs[f"variant{v}"] = ops.attention(qkv, B, N, H, hd, scale, out=o, mode=2, rpack=rp, pad_row=pad, gh=g, gw=g, ws=ws)
    finally:
        ops.attention_set_variant(5)

    # reference on CPU: pad the qkv map with the pad row, partition, attend per window, unpartition
    m = qkv.float().cpu().view(B, g, g, 3 * C) - pad.float().cpu().view(1, 1, 1, 3 * C)
    wins, pad_hw = window_partition(m, ws)          # zero padding == pad row after the shift below
    wins = wins + pad.float().cpu().view(1, 1, 1, 3 * C)
    nW = wins.shape[0]
    qkv_w = wins.view(nW, ws * ws, 3, H, hd).permute(2, 0, 3, 1, 4).reshape(3, nW * H, ws * ws, hd)
    q, k, v = qkv_w[0], qkv_w[1], qkv_w[2]
    att = (q * scale) @ k.transpose(-2, -1)
    rh, rw = decomposed_rel_pos_terms(q, Rh.cpu(), Rw.cpu(), (ws, ws))
    att = (att.view(-1, ws, ws, ws, ws) + rh[..., :, None] + rw[..., None, :]).view(-1, ws * ws, ws * ws)
    o = (att.softmax(-1) @ v).view(nW, H, ws, ws, hd).permute(0, 2, 3, 1, 4).reshape(nW, ws, ws, C)
    ref = window_unpartition(o, ws, pad_hw, (g, g)).reshape(B, N, C)
    for name, out in outs.items():
        err = (out.float().cpu() - ref).abs().max().item()
        assert torch.isfinite(out.float()).all(), name
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3, msg=f"{name}: max err {err:.2e}")


def test_head_major_qkv_layout(dev):
    """psam_gemm_f16_heads writes the packed qkv projection as [3,H,B*N,hd]; attention / relpos with head_major=1 give
    bit-identical results to the token-major [B,N,3,H,hd] path (same arithmetic, different addresses)."""
    from protosam_amd import ops
    B, g, H, hd, ws = 1, 64, 16, 80, 14
    N, D = g * g, H * hd
    x = _rand((B * N, D), dev, 1.0, 41).half()
    w = _rand((3 * D, D), dev, 0.05, 42).half()
    bias = _rand((3 * D,), dev, 0.5, 43)
    tokm = ops.gemm(x, w, bias, epilogue=ops.EPI_F16)                              # [B*N, 3*H*hd]
    for tile in (1, 11, 15):   # (15 cannot write head-major: it falls back to 11)
        ops.gemm_set_tile(tile)
        try:
            hm = ops.gemm_heads(x, w, bias, hd)                                    # [3*H, B*N, hd]
        finally:
            ops.gemm_set_tile(0)
        assert torch.equal(hm, tokm.view(B * N, 3 * H, hd).permute(1, 0, 2).contiguous())
    scale = hd ** -0.5
    Rh, Rw = _rand((2 * ws - 1, hd), dev, 0.3, 44), _rand((2 * ws - 1, hd), dev, 0.3, 45)
    pad = bias.half().view(3, H, hd).contiguous()
    rp = ops.pack_rel_tables(Rh, Rw, True, hd)
    a = ops.attention(tokm, B, N, H, hd, scale, mode=2, pad_row=pad, gh=g, gw=g, ws=ws,
                      relq=ops.relpos(tokm, rp, B, N, H, hd, g, ws, True, scale))
    b = ops.attention(hm, B, N, H, hd, scale, mode=2, pad_row=pad, gh=g, gw=g, ws=ws, head_major=True,
                      relq=ops.relpos(hm, rp, B, N, H, hd, g, ws, True, scale, head_major=True))
    assert torch.equal(a, b)
    Gh, Gw = _rand((2 * g - 1, hd), dev, 0.3, 46), _rand((2 * g - 1, hd), dev, 0.3, 47)
    rpg = ops.pack_rel_tables(Gh, Gw, False, hd)
    rh, rw = ops.relpos(tokm, rpg, B, N, H, hd, g, g, False, scale)
    rh2, rw2 = ops.relpos(hm, rpg, B, N, H, hd, g, g, False, scale, head_major=True)
    assert torch.equal(rh, rh2) and torch.equal(rw, rw2)
    a = ops.attention(tokm, B, N, H, hd, scale, mode=1, rel_h=rh, rel_w=rw, gh=g, gw=g)
    b = ops.attention(hm, B, N, H, hd, scale, mode=1, rel_h=rh, rel_w=rw, gh=g, gw=g, head_major=True)
    assert torch.equal(a, b)
    assert torch.equal(ops.attention(tokm, B, N, H, hd, scale), ops.attention(hm, B, N, H, hd, scale, head_major=True))


@pytest.mark.parametrize("M", [200, 512, 1297])
@pytest.mark.parametrize("tile", [0, 13, 12, 1])
def test_head_major_qkv_small_launches(dev, M, tile):
    """psam_gemm_f16_heads on launches small enough for the 64x64 / deep-ring kernels (a low-resolution DINOv2 call): the automatic
    choice and a forced tile 13 / 12 must still write [3*H, M, hd] planes (tile 13's epilogue has no plane addressing: the dispatch keeps
    such launches on the 128-tile kernel)."""
    from protosam_amd import ops
    H, hd = 12, 64
    D = H * hd
    x = _rand((M, D), dev, 1.0, 141).half()
    w = _rand((3 * D, D), dev, 0.05, 142).half()
    bias = _rand((3 * D,), dev, 0.5, 143)
    ops.gemm_set_tile(1)
    try:
        tokm = ops.gemm(x, w, bias, epilogue=ops.EPI_F16)
    finally:
        ops.gemm_set_tile(0)
    ops.gemm_set_tile(tile)
    try:
        hm = ops.gemm_heads(x, w, bias, hd)
    finally:
        ops.gemm_set_tile(0)
    assert torch.equal(hm, tokm.view(M, 3 * H, hd).permute(1, 0, 2).contiguous())


def test_window_attention_fused_relpos(dev):
    """mode 2 with the rel-pos query terms computed in-kernel (rpack) equals the two-kernel path (psam_relpos -> relq)."""
    from protosam_amd import ops
    for (H, hd) in ((16, 80), (12, 64)):
        B, g, ws = 2, 64, 14
        N = g * g
        qkv = _rand((B, N, 3, H, hd), dev, 1.0, 51).half()
        pad = _rand((3, H, hd), dev, 0.5, 52).half()
        Rh, Rw = _rand((2 * ws - 1, hd), dev, 0.3, 53), _rand((2 * ws - 1, hd), dev, 0.3, 54)
        rp = ops.pack_rel_tables(Rh, Rw, True, hd)
        scale = hd ** -0.5
        relq = ops.relpos(qkv, rp, B, N, H, hd, g, ws, True, scale)
        a = ops.attention(qkv, B, N, H, hd, scale, mode=2, relq=relq, pad_row=pad, gh=g, gw=g, ws=ws)
        b = ops.attention(qkv, B, N, H, hd, scale, mode=2, rpack=rp, pad_row=pad, gh=g, gw=g, ws=ws)
        err = (a.float() - b.float()).abs().max().item()
        print(f"H{H} hd{hd}: fused vs two-kernel max abs diff {err:.2e}")
        # (outputs reach |4|: one fp16 ulp there is 3.9e-3 - the two kernels take their rescale decisions over different key chunks)
        assert torch.isfinite(b.float()).all()
        torch.testing.assert_close(b.float(), a.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("mode,N,H,hd,B", [(0, 1297, 12, 64, 2), (0, 200, 3, 80, 1), (1, 4096, 2, 80, 1), (1, 4096, 2, 64, 1), (1, 4096, 8, 80, 1),
                                           (1, 4096, 16, 80, 1), (1, 1024, 8, 80, 2), (1, 512, 8, 80, 1),
                                           (1, 4096, 8, 64, 1), (1, 1024, 12, 64, 2), (1, 4096, 12, 64, 2),    # hd = 64: the ViT-B kernel; 12 heads
                                           (1, 4096, 12, 64, 1), (1, 1024, 3, 80, 3)])   # B * H = 12 / 9: surplus workgroups of the last eight leave
def test_attention_softmax_variants_agree(dev, mode, N, H, hd, B):
    """V2 (tree reductions, one rescale decision for both query tiles, row sums on the matrix pipe) against the round-1 serial
    form and the fp32 reference, including rows whose maximum jumps late in the key sequence (the lazy-rescale branch: a key
    aligned with a query far beyond the 2^8 threshold, guide rule 26) and jumps that hit only ONE of a wave's two query tiles."""
    from protosam_amd import ops
    qkv = _rand((B, N, 3, H, hd), dev, 0.5, 31)
    for (qi, ki, gain) in ((17, N - 70, 40.0), (40, 130, 25.0), (N - 3, N // 2, 60.0)):   # rows 17 / 40: tiles 0 / 1 of wave 0
        qkv[0, ki, 1, 0] = qkv[0, qi, 0, 0] * gain
    qkv = qkv.half()
    scale = hd ** -0.5
    kw, rel = {}, None
    if mode == 1:
        g, gh = 64, N // 64                                                                # (gw is 64 in this mode; gh rows of keys)
        rel_h = _rand((B, H, N, g), dev, 0.7, 32)
        rel_w = _rand((B, H, N, g), dev, 0.7, 33)
        rel_h[0, 0, 100, min(50, gh - 1)] = 30.0                                           # a spike in the bias itself
        kw = dict(mode=1, rel_h=rel_h, rel_w=rel_w, gh=gh, gw=g)
        rel = (rel_h[..., :gh].reshape(B, H, N, gh, 1) + rel_w.view(B, H, N, 1, g)).reshape(B, H, N, N)
    outs = []
    # round-1 serial softmax, V2 on the register-staged kernel, V2 on the DMA-fed HIP kernel (gattn_kernel), the default: the assembly
    # kernels of csrc/gattn_asm_gen.py where they apply (rel-pos, hd = 80 / 64, B * H a multiple of 8, N a multiple of 256)
    for v in (0, 9, 17, 1):
        ops.attention_set_variant(v)
        try:
            outs.append(ops.attention(qkv, B, N, H, hd, scale, **kw).float())
        finally:
            ops.attention_set_variant(5)
    ref = _ref_attn_global(qkv, B, N, H, hd, scale, rel=rel)
    for o in outs:
        assert torch.isfinite(o).all()
        torch.testing.assert_close(o, ref, rtol=2e-3, atol=2e-3)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-3


@pytest.mark.parametrize("N,H,B", [(1297, 12, 2), (1301, 16, 1), (2594, 12, 1), (5330, 12, 1), (128, 2, 1), (192, 3, 3), (320, 1, 9), (449, 5, 2)])
def test_attention_global_asm_any_token_count(dev, N, H, B):
    """psam_gattn_asm_64_norel (csrc/gattn_asm_gen.py, round 5: DINOv2's attention, models/grid_proto_fewshot.py:88-98) at the token
    counts the configurations run - 1297 = one 504^2 slice (21 key tiles: the odd tail; 17 valid keys in the last), 1301 (ViT-L + 4
    register tokens), 2594, 5330 = 1022^2 (84 tiles, 18 valid keys in the last) - and at the edges of its structure: 128 tokens (two
    tiles, no loop), 192 (three), 320 (five tiles, a full last one), 449 (last tile holds ONE key); rows beyond N of the last query
    block; B * H not a multiple of eight. Against the fp32 arithmetic AND the HIP kernel it replaces, with rows whose maximum
    jumps by far more than the 2^8 lazy-rescale threshold late in the key sequence - once inside the masked last tile."""
    from protosam_amd import ops
    hd = 64
    qkv = _rand((B, N, 3, H, hd), dev, 0.5, 41)
    for (qi, ki, gain) in ((17, N - 1, 40.0), (70, N - 70, 25.0), (N - 3, N // 2, 60.0), (N - 1, 3, 30.0)):
        qkv[0, ki, 1, 0] = qkv[0, qi, 0, 0] * gain
    qkv = qkv.half()
    scale = hd ** -0.5
    out = ops.attention(qkv, B, N, H, hd, scale).float()
    ops.attention_set_variant(5 | 16)                       # the DMA-fed HIP kernel everywhere
    try:
        hip = ops.attention(qkv, B, N, H, hd, scale).float()
    finally:
        ops.attention_set_variant(5)
    ref = _ref_attn_global(qkv, B, N, H, hd, scale)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)
    assert (out - hip).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,H,hd", [(1, 2, 64), (1, 2, 80), (2, 16, 80), (3, 12, 64), (1, 5, 80)])
def test_attention_global_fused_relpos(dev, B, H, hd):
    """psam_gattn_asm_{80,64}_fused (round 5): the decomposed rel-pos terms of the global blocks (image_encoder.py:325-372) computed
    inside the attention kernel from the packed tables - against the fp32 reference arithmetic (oracle's restatement of
    add_decomposed_rel_pos, pinned to the reference) and against the two-kernel path (psam_relpos -> fp32 tables in HBM -> the _rel
    kernel) it replaces. One query is aligned with a late key (the lazy-rescale branch)."""
    from oracle.sam_image_encoder import decomposed_rel_pos_terms
    from protosam_amd import ops
    g = 64
    N = g * g
    assert ops.attention_fused_relpos(B, N, H, hd, g, g)
    qkv = _rand((B, N, 3, H, hd), dev, 1.0, 51)
    qkv[0, N - 100, 1, 0] = qkv[0, 1000, 0, 0] * 6.0
    qkv = qkv.half()
    Rh = _rand((2 * g - 1, hd), dev, 0.3, 52)
    Rw = _rand((2 * g - 1, hd), dev, 0.3, 53)
    scale = hd ** -0.5
    rpack = ops.pack_rel_tables(Rh, Rw, False, hd)
    fused = ops.attention(qkv, B, N, H, hd, scale, mode=1, rpack=rpack, gh=g, gw=g).float()
    rel_h, rel_w = ops.relpos(qkv, rpack, B, N, H, hd, g, g, False, scale)
    two = ops.attention(qkv, B, N, H, hd, scale, mode=1, rel_h=rel_h, rel_w=rel_w, gh=g, gw=g).float()
    assert torch.isfinite(fused).all()
    assert (fused - two).abs().max().item() < 2e-3, (fused - two).abs().max().item()
    if B * H <= 4:      # (the [B H, N, N] fp32 bias of the reference arithmetic: 64 MB per head)
        q = qkv.float().view(B, N, 3, H, hd)[:, :, 0].permute(0, 2, 1, 3).reshape(B * H, N, hd)
        rh_ref, rw_ref = decomposed_rel_pos_terms(q.cpu(), Rh.cpu(), Rw.cpu(), (g, g))
        bias = (rh_ref[..., :, None] + rw_ref[..., None, :]).reshape(B, H, N, N).to(dev)
        ref = _ref_attn_global(qkv, B, N, H, hd, scale, rel=bias)
        torch.testing.assert_close(fused, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("tile", [1, 11, 15])
@pytest.mark.parametrize("M,D,N2,act", [(4096, 1280, 3840, 0), (1297 * 3, 768, 3072, 1), (777, 256, 256, 0)])
def test_gemm_folded_layernorm(dev, tile, M, D, N2, act):
    """psam_gemm_f16_ln: x = resid + gamma * (a w^T + b) emitting half(x) + per-row partial sums, psam_ln_finalize, then the
    consuming GEMM on half(x) with LayerNorm-folded weights  ==  Linear(LayerNorm(x)) (GELU) to fp16-operand accuracy; also
    against the separate LayerNorm-pass path, and the statistics themselves against torch."""
    from protosam_amd import ops
    K1 = 256
    a = _rand((M, K1), dev, 1.0, 51).half()
    w1 = _rand((D, K1), dev, 0.06, 52).half()
    b1 = _rand((D,), dev, 0.3, 53)
    gamma = (_rand((D,), dev, 0.2, 54) + 1.0).contiguous()
    resid = (_rand((M, D), dev, 1.0, 55) + 0.7).contiguous()          # a common-mode offset the LayerNorm removes
    ln_w = (_rand((D,), dev, 0.1, 56) + 1.0).contiguous()
    ln_b = _rand((D,), dev, 0.1, 57)
    w2 = _rand((N2, D), dev, 0.03, 58)
    b2 = _rand((N2,), dev, 0.2, 59)
    eps = 1e-6
    x_ref = resid + gamma * (a.float() @ w1.float().t() + b1)
    y_ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x_ref, (D,), ln_w, ln_b, eps), w2, b2)
    if act:
        y_ref = torch.nn.functional.gelu(y_ref)
    ops.gemm_set_tile(tile)
    try:
        x = resid.clone()
        x16 = torch.empty((M, D), dtype=torch.float16, device=dev)
        stats = torch.full((M, D // 64, 2), float("nan"), device=dev)
        ops.gemm(a, w1, b1, out=x, epilogue=ops.EPI_F32, resid=x, gamma=gamma, out16=x16, stats=stats)
        torch.testing.assert_close(x, x_ref, rtol=1e-4, atol=3e-4)
        assert torch.equal(x16, x.half())
        torch.testing.assert_close(stats[..., 0].sum(1), x.sum(1), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(stats[..., 1].sum(1), (x * x).sum(1), rtol=1e-4, atol=1e-2)
        mr = ops.ln_finalize(stats, M, D, eps)                 # fp32 (mean, rstd) [M][2], then the fp16 fragments of -mean [M][8]
        mr2 = mr[:2 * M].view(M, 2)
        torch.testing.assert_close(mr2[:, 0], x.mean(1), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(mr2[:, 1], torch.rsqrt(x.var(1, unbiased=False) + eps), rtol=1e-4, atol=1e-5)
        frag = mr[2 * M:].view(torch.float16).view(M, 8).float()
        assert torch.equal(frag[:, 0], frag[:, 1]) and int(frag[:, 3:].abs().sum()) == 0
        torch.testing.assert_close(frag[:, 0] + frag[:, 2], -mr2[:, 0], rtol=1e-6, atol=1e-7)
        wf, s_, t_ = ops.fold_layernorm(w2, b2, ln_w, ln_b)
        e = ops.EPI_GELU_F16 if act else ops.EPI_F16
        y = ops.gemm(x16, wf, t_, epilogue=e, ln_mr=mr, ln_s=s_).float()
        # the separate-pass path through the same tile
        y_sep = ops.gemm(ops.layernorm(x, ln_w, ln_b, eps), w2.half(), b2, epilogue=e).float()
    finally:
        ops.gemm_set_tile(0)
    err, err_sep = (y - y_ref).abs().max().item(), (y_sep - y_ref).abs().max().item()
    print(f"tile {tile} M={M} D={D}: folded max err {err:.2e}, separate LayerNorm pass {err_sep:.2e}")
    torch.testing.assert_close(y, y_ref, rtol=3e-3, atol=3e-3)
    assert err < 2.0 * err_sep + 1e-3


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (4096, 1280, 1280), (33000, 1280, 192), (70001, 768, 128), (2000, 1280, 5120)])
@pytest.mark.parametrize("ln", [0, 1])
def test_gemm_fp32_residual_without_gamma(dev, M, N, K, ln):
    """Tile 15's fp32 epilogue as the SAM encoder calls it (no gamma; in place): its residual slabs travel through the VGPR ring AND
    through accumulator registers freed by the parked slabs (gemm_asm_gen.py epilogue_f32) - every slab position must get its own rows.
    Bit-identical to the HIP kernel (tile 11: same arithmetic in the same order), out of place, without a residual, and as the
    folded-LayerNorm producer (fp16 copy + row sums)."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 71).half()
    w = _rand((N, K), dev, 0.05, 72).half()
    bias = _rand((N,), dev, 0.5, 73)
    resid = (_rand((M, N), dev, 3.0, 74) + 0.5).contiguous()
    ref = resid + (a.float() @ w.float().t() + bias)
    outs = []
    try:
        for tile in (15, 11):
            ops.gemm_set_tile(tile)
            x = resid.clone()
            kw = {}
            if ln:
                kw = dict(out16=torch.empty((M, N), dtype=torch.float16, device=dev), stats=torch.full((M, N // 64, 2), float("nan"), device=dev))
            ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=x, **kw)
            outs.append(x)
            torch.testing.assert_close(x, ref, rtol=1e-5, atol=2e-4)
            if ln:
                assert torch.equal(kw["out16"], x.half())
                torch.testing.assert_close(kw["stats"][..., 0].sum(1), x.sum(1), rtol=1e-4, atol=2e-2)
                torch.testing.assert_close(kw["stats"][..., 1].sum(1), (x * x).sum(1), rtol=1e-4, atol=2e-2)
        assert torch.equal(outs[0], outs[1])
        ops.gemm_set_tile(15)
        if not ln:
            y = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(a, w, bias, out=y, epilogue=ops.EPI_F32, resid=resid)              # out of place
            assert torch.equal(y, outs[0])
            z = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(a, w, bias, out=z, epilogue=ops.EPI_F32)                           # no residual
            torch.testing.assert_close(z, a.float() @ w.float().t() + bias, rtol=1e-5, atol=2e-4)
    finally:
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("B,T,Nk,hm,kdt", [(1, 7, 4096, True, torch.float32), (2, 9, 4096, True, torch.float32), (3, 16, 4096, False, torch.float32),
                                            (2, 8, 1024, True, torch.float32), (1, 7, 576, False, torch.float16), (27, 8, 4096, True, torch.float32)])
def test_t2i_attention_key_split(dev, B, T, Nk, hm, kdt):
    """token-to-image attention of the two-way transformer (transformer.py:228-230 on the 4096 image tokens): the kernel against fp64
    softmax(q k^T / 4) v per head, and its key-split form (psam_t2i_attention_split: S workgroups per (prompt set, head), partials
    merged in the order of the split index by a second launch) against the unsplit launch."""
    from protosam_amd import ops
    NH, HD = 8, 16
    q = _rand((B * T, 128), dev, 1.5, 81)
    k = _rand((B, Nk, 128), dev, 1.0, 82)
    v = _rand((B, Nk, 128), dev, 1.0, 83)
    k[0, 5] *= 6.0                                   # a dominant key
    qh = q.double().view(B, T, NH, HD).permute(0, 2, 1, 3)
    kh = k.double().view(B, Nk, NH, HD).permute(0, 2, 1, 3)
    vh = v.double().view(B, Nk, NH, HD).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / 4.0, -1) @ vh).permute(0, 2, 1, 3).reshape(B * T, 128)
    if kdt == torch.float16:
        ref16 = k.half(), v.half()
        kh = ref16[0].double().view(B, Nk, NH, HD).permute(0, 2, 1, 3)
        vh = ref16[1].double().view(B, Nk, NH, HD).permute(0, 2, 1, 3)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) / 4.0, -1) @ vh).permute(0, 2, 1, 3).reshape(B * T, 128)
    lay = (lambda x: x.view(B, Nk, NH, HD).permute(0, 2, 1, 3).contiguous()) if hm else (lambda x: x.contiguous())
    K, V = lay(k.to(kdt)), lay(v.to(kdt))
    base = torch.full((B * T, 128), float("nan"), device=dev)
    ops.t2i_attention(q, K, V, base, B, T, Nk, NH, head_major=hm)
    torch.testing.assert_close(base.double(), ref, rtol=2e-5, atol=2e-5)
    for S in (2, 5, 16, ops.t2i_split(B, NH, T, Nk)):
        if S < 2 or Nk // S < 32:
            continue
        part = torch.full((B * NH * S * T * 18,), float("nan"), device=dev)
        outs = []
        for _ in range(3):
            out = torch.full((B * T, 128), float("nan"), device=dev)
            ops.t2i_attention(q, K, V, out, B, T, Nk, NH, head_major=hm, split=(S, part))
            outs.append(out)
        for out in outs:
            assert torch.equal(out, outs[0])
            torch.testing.assert_close(out, base, rtol=1e-5, atol=2e-6)
    assert ops.t2i_split(1, 8, 7, 4096) == 16 and ops.t2i_split(27, 8, 8, 4096) == 1 and ops.t2i_split(16, 8, 8, 4096) == 2


@pytest.mark.parametrize("M,N,K,mod", [(3 * 512 + 100, 512, 768, 512), (2 * 4096, 1280, 768, 4096), (1024 + 7, 256, 128, 256)])
@pytest.mark.parametrize("ln", [False, True])
def test_gemm_position_table_residual_on_the_assembly_tile(dev, M, N, K, mod, ln):
    """The patch embedding's epilogue (image_encoder.py:107-108: x = patch_embed(img) + pos_embed): residual row = output row % resid_mod, a
    [resid_mod, N] table shared by all images, out of place - on the 256-tile assembly kernel (tile 15, the row of a tile's origin masked:
    resid_mod a power of two >= 256) with and without the folded-LayerNorm producer outputs, against the fp32 product and bit-for-bit
    against the HIP 256-tile kernel (tile 11), ragged last tile included."""
    from protosam_amd import ops
    a = _rand((M, K), dev, 1.0, 91).half()
    w = _rand((N, K), dev, 0.05, 92).half()
    bias = _rand((N,), dev, 0.5, 93)
    pos = (_rand((mod, N), dev, 2.0, 94) + 0.25).contiguous()
    ref = a.float() @ w.float().t() + bias + pos.repeat((M + mod - 1) // mod, 1)[:M]
    outs = []
    try:
        for tile in (15, 11):
            ops.gemm_set_tile(tile)
            x = torch.full((M, N), float("nan"), device=dev)
            kw = {}
            if ln:
                kw = dict(out16=torch.empty((M, N), dtype=torch.float16, device=dev), stats=torch.full((M, N // 64, 2), float("nan"), device=dev))
            ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_F32, resid=pos, resid_mod=mod, **kw)
            torch.testing.assert_close(x, ref, rtol=1e-5, atol=2e-4)
            if ln:
                assert torch.equal(kw["out16"], x.half())
                torch.testing.assert_close(kw["stats"][..., 0].sum(1), x.sum(1), rtol=1e-4, atol=2e-2)
            outs.append(x)
        assert torch.equal(outs[0], outs[1])
    finally:
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("M,N,K,mode", [(2 * 4096, 128, 256, "pe_heads"), (8192 + 37, 256, 128, "resid"), (1000, 64, 32, "plain"),
                                        (3 * 4096, 256, 256, "plain"), (4096, 128, 256, "tiny")])
def test_gemm_f32x3_matches_fp32_accuracy(dev, M, N, K, mode):
    """psam_gemm_f32x3 (three fp16 MFMA products on (hi, lo) halves, fp32 accumulation) against a float64 product of the fp32 operands:
    within a few 2^-22 of the sum of |a| |w| terms - the accuracy class of the exact-fp32 MFMA kernel it replaces on the decoder's image
    side, three orders of magnitude tighter than fp16 operands - with the fused position add, head-major output, in-place residual,
    a ragged last tile, and operands whose lo halves are fp16 subnormals."""
    from protosam_amd import ops
    scale = 1e-3 if mode == "tiny" else 1.0
    a = (_rand((M, K), dev, 2.0, 101) * scale).contiguous()
    w = (_rand((N, K), dev, 0.08, 102) * scale).contiguous()
    bias = _rand((N,), dev, 0.5, 103) * scale * scale
    ws = ops.split_weight_f16(w, 256.0) + (256.0,)
    a2 = _rand((4096, K), dev, 1.0, 104).contiguous() if mode == "pe_heads" else None
    ad = a.double() + (a2.double().repeat(M // 4096, 1) if a2 is not None else 0)
    ref = ad @ w.double().t() + bias.double()
    mag = ad.abs() @ w.double().abs().t() + bias.double().abs()            # sum of the terms' magnitudes: the scale of the rounding error
    if mode == "resid":
        x = _rand((M, N), dev, 1.0, 105).contiguous()
        ref = ref + x.double()
        out = x.clone()
        ops.gemm_f32x3(a, ws, bias, out=out, resid=out)
        exact = ops.gemm_f32(a, w, bias, out=x.clone(), resid=x)
    elif mode == "pe_heads":
        nk, hd = 4096, 16
        out = torch.full((M * N,), float("nan"), device=dev)
        ops.gemm_f32x3(a, ws, bias, out=out, a2=a2, a2_mod=4096, heads=(nk, hd))
        out = out.view(M // nk, N // hd, nk, hd).permute(0, 2, 1, 3).reshape(M, N)
        exact = ops.gemm_f32(a, w, bias, a2=a2, a2_mod=4096)
    else:
        out = torch.full((M, N), float("nan"), device=dev)
        ops.gemm_f32x3(a, ws, bias, out=out)
        exact = ops.gemm_f32(a, w, bias)
    # operand resolution of the split: 2^-22 relative, or 2^-25 absolute where the lo half is an fp16 subnormal (|a| < 1/8; |w| < 2^-11
    # after the weights' scaling by 2^8)
    floor = 2.0 ** -25 * w.double().abs().sum(1)[None, :] + 2.0 ** -33 * ad.abs().sum(1)[:, None]
    assert bool(((out.double() - ref).abs() <= 6 * 2.0 ** -22 * mag + 2 * floor).all())
    err = ((out.double() - ref).abs() / mag).max().item()
    err_exact = ((exact.double() - ref).abs() / mag).max().item()
    err_f16 = (((a.half().float() + (a2.half().float().repeat(M // 4096, 1) if a2 is not None else 0)).double() @ w.half().double().t()
                + bias.double() + (x.double() if mode == "resid" else 0) - ref).abs() / mag).max().item()
    if mode != "tiny":
        assert err < 8 * 2.0 ** -22, (err, err_exact, err_f16)
    assert err < err_f16 / (10 if mode == "tiny" else 100), (err, err_f16)


@pytest.mark.parametrize("M,N,K", [(4096, 1280, 5120), (4096, 1024, 4096), (3000, 1280, 5120)])
@pytest.mark.parametrize("with_ln", [True, False])
def test_gemm_splitk_asm_residual_layernorm(dev, M, N, K, with_ln):
    """Round 6: x += a w^T + b as K ranges of the assembly tile (psam_gemm_asm_f32_sk: items = (tile, range), partial sums to planes of a
    caller-owned workspace) + one pass that sums the ranges and applies the LayerNorm that follows in the block stack. Against the fp32
    arithmetic, against the one-launch GEMM + LayerNorm pass it replaces, run-to-run identical, and identical when replayed from a graph."""
    from protosam_amd import ops
    ks = ops.gemm_splitk_ranges(M, N, K)
    assert ks >= 2, ks
    a = _rand((M, K), dev, 1.0, 71).half()
    w = _rand((N, K), dev, 0.02, 72).half()
    bias = _rand((N,), dev, 0.5, 73)
    x0 = _rand((M, N), dev, 1.0, 74)
    lnw, lnb = (1.0 + _rand((N,), dev, 0.1, 75)), _rand((N,), dev, 0.1, 76)
    ws = torch.empty((ks, ops.splitk_rows(M), N), dtype=torch.float32, device=dev)
    outs = []
    for rep in range(2):
        x = x0.clone()
        o16 = torch.zeros((M, N), dtype=torch.float16, device=dev)
        ws.fill_(float("nan"))                                     # every element the reduce pass reads has to be written by the launch
        ops.gemm_splitk_ln(a, w, bias, x, ks, ws, lnw if with_ln else None, lnb if with_ln else None, 1e-6, out16=o16)
        outs.append((x, o16))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    x, o16 = outs[0]
    ref = x0 + a.float() @ w.float().t() + bias
    torch.testing.assert_close(x, ref, rtol=1e-4, atol=3e-4)
    ref16 = torch.nn.functional.layer_norm(x, (N,), lnw, lnb, 1e-6) if with_ln else x
    torch.testing.assert_close(o16.float(), ref16, rtol=2e-3, atol=2e-3)
    # the path it replaces: one GEMM launch with the residual epilogue, then the LayerNorm pass (same products, another summation order)
    y = ops.gemm(a, w, bias, out=x0.clone(), epilogue=ops.EPI_F32, resid=x0)
    torch.testing.assert_close(x, y, rtol=1e-5, atol=2e-5)
    if with_ln:
        y16 = ops.layernorm(y, lnw, lnb, 1e-6)
        assert (o16.float() - y16.float()).abs().max().item() <= 4e-3
    # graph capture: caller-owned workspace, no library state
    xg = x0.clone()
    og = torch.zeros_like(o16)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.gemm_splitk_ln(a, w, bias, xg, ks, ws, lnw if with_ln else None, lnb if with_ln else None, 1e-6, out16=og)
    xg.copy_(x0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg, x) and torch.equal(og, o16)


def test_gemm_splitk_asm_declines_shapes_that_do_not_pay(dev):
    from protosam_amd import ops
    assert ops.gemm_splitk_ranges(65536, 1280, 5120) == 0          # the CUs are full already
    assert ops.gemm_splitk_ranges(4096, 1280, 1280) == 0           # 20 K-tiles: no range of >= 16
    assert ops.gemm_splitk_ranges(4096, 768, 3072) == 0            # 48 tiles x 48 K-tiles: three ranges of 16 are 144 items, under 3/4 of the CUs
    assert ops.gemm_splitk_ranges(4096, 1280, 5120) == 3
