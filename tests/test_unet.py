"""The torch restatement of the reference UNet (fbs/nn/unet.py): structure, conventions, flat-parameter
order.  Pure torch, runs on CPU.  No JAX checkpoint exists here, so numerical parity with flax is
unpinned; these tests pin every building block against numpy / torch references and the flat layout
against the ravel_pytree rules."""
import math

import numpy as np
import pytest
import torch

from fbs_amd.unet import (UNet, sinusoidal_embedding, pixel_shuffle_nhwc, WeightStandardizedConv, _ChannelLayerNorm,
                          make_st_nn)


def test_sinusoidal_embedding_closed_form():
    t = torch.tensor(3.7)
    e = sinusoidal_embedding(t, out_dim=8).numpy()
    fs = np.exp(-math.log(10000) * np.arange(4) / 3)
    np.testing.assert_allclose(e, np.concatenate([np.sin(3.7 * fs), np.cos(3.7 * fs)]), rtol=1e-5)
    assert sinusoidal_embedding(torch.tensor([1.0, 2.0]), out_dim=64).shape == (2, 64)
    with pytest.raises(NotImplementedError):
        sinusoidal_embedding(t, out_dim=7)


def test_pixel_shuffle_matches_einops_pattern_and_torch():
    """tests/test_nns.py:7-16 of the reference: the flax PixelShuffle equals torch's up to the channel
    ordering convention; here additionally against the einops pattern itself."""
    import einops
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 5, 6, 12)).astype(np.float32)             # b h w (h2 w2 c), scale 2, c = 3
    want = einops.rearrange(x, 'b h w (h2 w2 c) -> b (h h2) (w w2) c', h2=2, w2=2)
    got = pixel_shuffle_nhwc(torch.from_numpy(x).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(got, want)
    # torch.nn.PixelShuffle orders channels (c h2 w2): permute channels and compare
    xc = x.reshape(2, 5, 6, 2, 2, 3).transpose(0, 1, 2, 5, 3, 4).reshape(2, 5, 6, 12)
    tor = torch.nn.PixelShuffle(2)(torch.from_numpy(xc).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(tor, want, atol=1e-6)


def test_weight_standardisation_and_layernorm_conventions():
    torch.manual_seed(0)
    m = WeightStandardizedConv(3, 5)
    x = torch.randn(2, 3, 8, 8)
    w = m.conv.weight.detach().numpy()
    ws = (w - w.mean(axis=(1, 2, 3), keepdims=True)) / np.sqrt(w.var(axis=(1, 2, 3), keepdims=True) + 1e-5)
    want = torch.nn.functional.conv2d(x, torch.from_numpy(ws), m.conv.bias, padding=1)
    np.testing.assert_allclose(m(x).detach().numpy(), want.detach().numpy(), atol=1e-5)
    ln = _ChannelLayerNorm(3)
    y = ln(x).detach().numpy()
    xn = x.numpy()
    ref = (xn - xn.mean(1, keepdims=True)) / np.sqrt(xn.var(1, keepdims=True) + 1e-5)
    np.testing.assert_allclose(y, ref, atol=1e-5)


@pytest.mark.parametrize("upsampling", ["resize", "pixel_shuffle"])
def test_unet_shapes_and_flat_param_round_trip(upsampling):
    torch.manual_seed(1)
    net = UNet(dt=2.0 / 200, dim=16, in_channels=1, upsampling=upsampling, dim_mults=(1, 2, 4)).eval()
    x = torch.randn(3, 28, 28, 1)
    with torch.no_grad():
        out = net(x, 0.5)
        assert out.shape == (3, 28, 28, 1) and torch.isfinite(out).all()
        assert net(x[0], torch.tensor(0.5)).shape == (28, 28, 1)           # unbatched call (unet.py:365-368)
        per_t = net(x, torch.tensor([0.5, 0.5, 0.5]))
    np.testing.assert_allclose(per_t.numpy(), out.numpy(), atol=1e-5)
    # every torch parameter appears exactly once in the flat layout
    spec = net.flat_param_spec()
    assert len({id(p) for _, p, _ in spec}) == len(spec) == len(list(net.parameters()))
    assert net.num_flat_params() == sum(p.numel() for p in net.parameters())
    # ravel_pytree order: sorted keys at every level ('D' < 'R' < 'd' < 'f' < 'i' < 'm' < 'u'; bias < kernel < scale)
    paths = [p for p, _, _ in spec]
    assert paths == sorted(paths, key=lambda s: s.split('/'))
    assert paths[0] == 'params/Dense_0/bias' and paths[1] == 'params/Dense_0/kernel'
    assert any(p.startswith('params/ResnetBlock_5/') for p in paths)
    assert 'params/mid.attenblock_0/Attention_0/to_qkv.conv_0/kernel' in paths
    assert 'params/down_0.attnblock_0/LinearAttention_0/to_out.norm_0/scale' in paths
    assert ('params/up_1.upsample_0/Conv_1/kernel' in paths) == (upsampling == 'pixel_shuffle')
    # round trip through the flat vector, and layout of a conv kernel chunk = (kh, kw, in, out)
    vec = net.export_flat_params()
    net2 = UNet(dt=2.0 / 200, dim=16, in_channels=1, upsampling=upsampling, dim_mults=(1, 2, 4)).eval()
    net2.load_flat_params(vec.numpy())
    with torch.no_grad():
        np.testing.assert_allclose(net2(x, 0.5).numpy(), out.numpy(), atol=1e-6)
    assert torch.equal(net2.export_flat_params(), vec)
    o = 0
    for path, p, kind in spec:
        if path == 'params/init.conv_0/kernel':
            k = vec[o:o + p.numel()].reshape(7, 7, 1, 16)
            assert torch.equal(k.permute(3, 2, 0, 1), p.detach())
        o += p.numel()
    with pytest.raises(ValueError):
        net2.load_flat_params(np.zeros(10, np.float32))


def test_unet_reference_configuration_size():
    """UNet(dt=T/200, dim=64, upsampling='pixel_shuffle') of experiments/imgs/inpainting.py:85 on MNIST."""
    net = UNet(dt=2.0 / 200, dim=64, in_channels=1, upsampling='pixel_shuffle')
    n = net.num_flat_params()
    assert 5_000_000 < n < 20_000_000
    _, _, fwd = make_st_nn(net)
    assert fwd(torch.zeros(2, 28, 28, 1), 0.3).shape == (2, 28, 28, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,dim", [(5, 28, 28, 64), (3, 7, 7, 256), (2, 9, 13, 128), (1, 64, 64, 64)])
def test_fused_linear_attention_matches_the_torch_ops(B, H, W, dim):
    """libfbsmi's one-kernel LinearAttention core (include/fbsmi_nn.h) against the eager restatement of
    fbs/nn/unet.py:209-245 in fbs_amd/unet.py, float32 (tight) and bfloat16 autocast (bf16 tolerance)."""
    from fbs_amd.unet import LinearAttention
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 100 + H)
    attn = LinearAttention(dim).to(dev).eval()
    x = torch.randn(B, dim, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.enable_grad():          # autograd on: the eager path
        want = attn(x).detach()
    with torch.no_grad():              # inference: the fused kernel
        got = attn(x)
    assert got.shape == want.shape
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * max(scale, 1.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.enable_grad():
            want16 = attn(x).detach().float()
        with torch.no_grad():
            got16 = attn(x).float()
    assert (got16 - want16).abs().max().item() <= 3e-2 * max(want16.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,din,dim", [(6, 28, 28, 64, 64), (3, 14, 14, 64, 128), (2, 7, 7, 384, 256), (1, 64, 64, 128, 64)])
def test_fused_resnet_block_matches_the_torch_ops(B, H, W, din, dim):
    """ResnetBlock with libfbsmi's GroupNorm + modulation + SiLU kernel against the eager restatement of
    fbs/nn/unet.py:127-172, float32 and bfloat16 autocast; the same for the channel LayerNorm's fused path."""
    from fbs_amd.unet import ResnetBlock, _ChannelLayerNorm
    dev = torch.device("cuda:0")
    torch.manual_seed(dim + H)
    blk = ResnetBlock(din, dim, 256).to(dev).eval()
    with torch.no_grad():
        for nrm in (blk.norm_0, blk.norm_1):
            nrm.weight.uniform_(0.5, 1.5)
            nrm.bias.uniform_(-0.3, 0.3)
    x = (torch.randn(B, din, H, W, device=dev) * 2 + 0.7).contiguous(memory_format=torch.channels_last)
    emb = torch.randn(B, 256, device=dev)
    with torch.enable_grad():
        want = blk(x, emb).detach()
    with torch.no_grad():
        got = blk(x, emb)
    assert (got - want).abs().max().item() <= 1e-4 * max(want.abs().max().item(), 1.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.enable_grad():
            want16 = blk(x, emb).detach().float()
        with torch.no_grad():
            got16 = blk(x, emb).float()
    assert (got16 - want16).abs().max().item() <= 4e-2 * max(want16.abs().max().item(), 1.0)
    ln = _ChannelLayerNorm(din).to(dev)
    with torch.no_grad():
        ln.scale.uniform_(0.5, 1.5)
    ref = (x - x.mean(1, keepdim=True)) * torch.rsqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5) * ln.scale.view(1, -1, 1, 1)
    assert (ln(x) - ref).abs().max().item() <= 1e-4                 # autograd on: torch's layer_norm
    with torch.no_grad():
        assert (ln(x) - ref).abs().max().item() <= 1e-4             # inference: libfbsmi's kernel
        fused = (din // 8) & (din // 8 - 1) == 0          # the kernel takes C / 8 a power of two; else torch's layer_norm
        got_bf = ln(x.to(torch.bfloat16)).float() if fused else None
    if fused:
        assert (got_bf - ref).abs().max().item() <= 4e-2 * max(ref.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,dim", [(5, 28, 28, 64), (3, 14, 14, 128), (2, 9, 13, 32), (1, 64, 64, 64), (7, 5, 5, 16)])
def test_fused_qkv_linear_attention_on_the_matrix_cores(B, H, W, dim):
    """fbsmi_nn_qkv_linear_attention (to_qkv + LinearAttention core, bfloat16 MFMA) against the float32 torch ops of
    fbs/nn/unet.py:209-245 on the same bfloat16-rounded inputs and weights; the tolerance is bfloat16's (the kernel rounds the
    softmax numerators, v, the context and q to bfloat16 where the autocast reference rounds qkv and the einsum operands)."""
    from fbs_amd import _lib
    from fbs_amd.unet import LinearAttention
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 100 + dim)
    att = LinearAttention(dim).to(dev).eval()
    x = (torch.randn(B, dim, H, W, device=dev) * 1.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        att.to_qkv.weight.mul_(3.0)                       # spread the logits: softmaxes away from uniform
        w16 = att.to_qkv.weight.to(torch.bfloat16).float()
        got = att._fused_qkv_core(x, B, H, W).float()
        n = H * W
        qkv = torch.nn.functional.conv2d(x.float(), w16)
        q, k, v = (t.reshape(B, 4, 32, n).permute(0, 3, 1, 2) for t in qkv.chunk(3, dim=1))
        q = torch.softmax(q, dim=-1) / 32 ** 0.5
        k = torch.softmax(k, dim=-3)
        ctx = torch.einsum('bnhd,bnhe->bhde', k, v / n)
        want = torch.einsum('bhde,bnhd->bhen', ctx, q).reshape(B, 128, H, W)
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * max(want.abs().max().item(), 1e-6), (err, want.abs().max().item())
    # through the module under autocast: fused inference path against the eager path
    xin = torch.randn(B, dim, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.enable_grad():
            ref = att(xin).detach().float()
        with torch.no_grad():
            fused = att(xin.to(torch.bfloat16)).float()
    assert (fused - ref).abs().max().item() <= 6e-2 * max(ref.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 28, 28, 64, 64), (2, 14, 14, 128, 128), (5, 7, 7, 64, 256), (1, 64, 64, 128, 64),
                                           (2, 9, 13, 64, 64), (1, 1, 1, 64, 64), (300, 3, 2, 128, 64)])
def test_conv3x3_on_the_matrix_cores(B, H, W, cin, cout):
    """fbsmi_nn_conv3x3 against torch's convolution in float32 on the same bfloat16-rounded inputs and weights (the kernel
    accumulates in float32 and rounds the result to bfloat16 once: tolerance = one bfloat16 rounding of the result plus
    accumulation-order noise); image borders, tiles that straddle rows and images, a ragged last tile, the bias."""
    from fbs_amd.unet import _conv3x3_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(B + cin + cout)
    x = torch.randn(B, cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    want = torch.nn.functional.conv2d(x.float(), w.float(), bias, padding=1)
    got = _conv3x3_hip(x, w, bias).float()
    assert got.shape == want.shape
    tol = 2.0 ** -8 * want.abs() + 1e-3
    assert bool(((got - want).abs() <= tol).all()), (got - want).abs().max().item()
    got0 = _conv3x3_hip(x, w, None).float()
    want0 = want - bias.view(1, -1, 1, 1)
    assert bool(((got0 - want0).abs() <= 2.0 ** -8 * want0.abs() + 1e-3).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,parts,cout", [(3, 14, 14, (128, 64), 128), (2, 7, 7, (256, 128), 256), (2, 28, 28, (64, 64), 64),
                                             (1, 9, 5, (192,), 64), (2, 6, 6, (256,), 128)])
def test_conv3x3_over_channel_slices_and_unconcatenated_inputs(B, H, W, parts, cout):
    """conv(cat(a, b)) = conv_a(a) + conv_b(b): the kernel takes the inputs' channel slices (128 / 64 wide) in turn and
    accumulates in the output (rounded to bfloat16 between slices: one more rounding per slice in the tolerance)."""
    from fbs_amd.unet import _conv3x3_fusable, _conv3x3_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(parts) + cout)
    xs = tuple(torch.randn(B, c, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c in parts)
    cin = sum(parts)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    with torch.no_grad():
        assert _conv3x3_fusable(xs if len(xs) > 1 else xs[0], w)
        got = _conv3x3_hip(xs if len(xs) > 1 else xs[0], w, bias).float()
    want = torch.nn.functional.conv2d(torch.cat([t.float() for t in xs], dim=1), w.float(), bias, padding=1)
    # every slice but the last leaves a bfloat16-rounded PARTIAL sum, whose size is that of the whole result's scale, not of
    # the element's final value: one rounding (2^-9 relative) of a partial of up to ~the largest output per extra slice
    nslices = sum((c + 127) // 128 for c in parts)
    tol = 2.0 ** -8 * want.abs() + (nslices - 1) * 2.0 ** -9 * 2.0 * want.abs().max() + 4e-3
    assert got.shape == want.shape and bool(((got - want).abs() <= tol).all()), (got - want).abs().max().item()


def test_conv3x3_supported_query_matches_the_kernel_limits():
    """fbsmi_nn_conv3x3_supported (host-only: no GPU work) is what unet.py asks before routing a convolution to the kernel;
    its bound is the LDS footprint of the staged pixel range: W < 248 at 64-channel slices, W <= 100 at 128-channel ones
    (celeba-128 / celeba-256 rows are configurable in the reference: those must fall back, not raise)."""
    from fbs_amd import _lib
    q = _lib.lib().fbsmi_nn_conv3x3_supported
    assert q(28, 28, 64, 64) == 1 and q(64, 64, 128, 64) == 1 and q(128, 128, 64, 64) == 1
    assert q(128, 128, 128, 64) == 0 and q(256, 256, 64, 64) == 0 and q(256, 256, 128, 128) == 0
    assert q(8, 100, 128, 64) == 1 and q(8, 101, 128, 64) == 0
    assert q(8, 247, 64, 64) == 1 and q(8, 248, 64, 64) == 0
    assert q(8, 8, 32, 64) == 0 and q(8, 8, 64, 32) == 0 and q(0, 8, 64, 64) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,cin,cout", [(8, 128, 64, 64), (8, 128, 128, 64), (4, 256, 64, 64), (4, 256, 128, 128)])
def test_wide_rows_fall_back_to_the_library_convolution(H, W, cin, cout):
    """Rows too wide for the kernel's staged range: the dispatch must take the library path instead of raising;
    where the kernel does have a tile shape (W = 128 at 64 channels) it is used and must agree."""
    from fbs_amd.unet import WeightStandardizedConv, _conv3x3_fusable
    dev = torch.device("cuda:0")
    torch.manual_seed(W + cin)
    m = WeightStandardizedConv(cin, cout).to(dev).eval()
    x = torch.randn(2, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.enable_grad():
        want = m(x).detach()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        x16 = x.to(torch.bfloat16)
        fus = _conv3x3_fusable(x16, m._standardised().to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
        got = m(x16).float()
    assert fus == (W == 128 and cin == 64)
    assert (got - want).abs().max().item() <= 4e-2 * max(want.abs().max().item(), 1.0)


@pytest.mark.gpu
def test_conv3x3_many_slices_rounding_is_bounded():
    """A 256 + 256 channel up-path convolution is four slice launches, the partial sum rounded to bfloat16 between them
    (ADVICE r2): the error against one float32 accumulation stays within (slices - 1) roundings of the result's scale."""
    from fbs_amd.unet import _conv3x3_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(512)
    xs = tuple(torch.randn(2, 256, 14, 14, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2))
    w = (torch.randn(128, 512, 3, 3, device=dev) / (3 * 512 ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = _conv3x3_hip(xs, w, None).float()
    want = torch.nn.functional.conv2d(torch.cat([t.float() for t in xs], dim=1), w.float(), None, padding=1)
    tol = 2.0 ** -8 * want.abs() + 3 * 2.0 ** -9 * 2.0 * want.abs().max() + 4e-3
    assert bool(((got - want).abs() <= tol).all()), (got - want).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,chans,ln", [(3, 28, 28, (128,), True), (2, 9, 7, (64, 64), False), (5, 5, 5, (64,), False),
                                           (1, 64, 64, (128,), True), (2, 14, 14, (64, 64), True)])
def test_proj64_with_layernorm_and_residual(B, H, W, chans, ln):
    """fbsmi_nn_proj64 (1x1 projection of one or two inputs to 64 channels [+ channel LayerNorm] [+ residual]) against the
    float32 torch ops on the same bfloat16-rounded operands."""
    from fbs_amd.unet import _ChannelLayerNorm, _proj64, _proj64_fusable
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(chans) + B)
    parts = tuple(torch.randn(B, c, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c in chans)
    conv = torch.nn.Conv2d(sum(chans), 64, 1).to(dev)
    norm = _ChannelLayerNorm(64).to(dev)
    res = torch.randn(B, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        norm.scale.uniform_(0.5, 1.5)
        assert _proj64_fusable(parts, 64)
        got = _proj64(parts, conv, ln=norm if ln else None, residual=res).float()
        w16 = conv.weight.to(torch.bfloat16).float()
        y = torch.nn.functional.conv2d(torch.cat([t.float() for t in parts], dim=1), w16, conv.bias)
        if ln:
            y = (y - y.mean(1, keepdim=True)) * torch.rsqrt(y.var(1, unbiased=False, keepdim=True) + 1e-5) * norm.scale.view(1, -1, 1, 1)
        want = y + res.float()
    assert got.shape == want.shape
    assert bool(((got - want).abs() <= 2.0 ** -7 * want.abs() + 2e-2).all()), (got - want).abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_folds_and_pixel_shuffle_kernels(dtype):
    """fbsmi_nn_bias_add / fbsmi_nn_pixel_shuffle (+ bias) / the xbias of the channel LayerNorm against the torch ops they replace:
    the float32 results are the same bits (one addition per element either way), bfloat16 rounds once after the addition."""
    from fbs_amd import _lib
    from fbs_amd.unet import Upsample, _ChannelLayerNorm, _add_bias, pixel_shuffle_nhwc
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, H, W, c = 3, 7, 5, 16
    y = torch.randn(B, 4 * c, H, W, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(4 * c, device=dev)
    want = (y.float() + bias.view(1, -1, 1, 1)).to(dtype)
    with torch.no_grad():
        got = _add_bias(y.clone(memory_format=torch.preserve_format), bias)
    assert torch.equal(got, want)
    # the shuffle on a given convolution output (the convolution itself is MIOpen's and need not be reproducible call to call)
    want = pixel_shuffle_nhwc((y.float() + bias.view(1, -1, 1, 1)).to(dtype), 2)
    tok = y.permute(0, 2, 3, 1).contiguous()
    out = torch.empty((B, 2 * H, 2 * W, c), dtype=dtype, device=dev)
    _lib.call("fbsmi_nn_pixel_shuffle", tok.data_ptr(), out.data_ptr(), 0 if dtype == torch.float32 else 1, B, H, W, c, 2,
              bias.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out.permute(0, 3, 1, 2), want)
    up = Upsample(2 * c, c, "pixel_shuffle").to(dev).eval()      # and through the module, within the convolution's tolerance
    x = torch.randn(B, 2 * c, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.enable_grad():
        want_up = up(x).detach()
    with torch.no_grad():
        got_up = up(x)
    assert got_up.shape == (B, c, 2 * H, 2 * W) and (got_up - want_up).abs().max().item() <= 1e-4 * max(want_up.abs().max().item(), 1.0)
    ln = _ChannelLayerNorm(4 * c).to(dev)
    with torch.no_grad():
        ln.scale.uniform_(0.5, 1.5)
        res = torch.randn_like(y)
        got = ln(y, residual=res, xbias=bias).float()
        xb = y.float() + bias.view(1, -1, 1, 1)
        ref = (xb - xb.mean(1, keepdim=True)) * torch.rsqrt(xb.var(1, unbiased=False, keepdim=True) + 1e-5) * ln.scale.view(1, -1, 1, 1)
        ref = ref + res.float()
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert (got - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,up", [((3, 28, 28, 1), "pixel_shuffle"), ((2, 32, 32, 3), "resize")])
def test_whole_unet_fused_inference_matches_eager(shape, up):
    """The inference path (libfbsmi kernels for attention / normalisation glue, cached standardised weights) against
    the eager torch restatement, end to end, float32; bfloat16 autocast within bf16 tolerance."""
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = UNet(dt=0.01, dim=64, in_channels=shape[-1], upsampling=up).to(dev).eval()
    x = torch.randn(*shape, device=dev)
    t = torch.tensor(0.37, device=dev)
    with torch.enable_grad():
        want = net(x, t).detach()
    with torch.no_grad():
        got = net(x, t)
    # (no bit-equality between two calls is asked for: MIOpen's float32 convolutions are not reproducible run to
    # run for every shape, in eager mode either)
    assert (got - want).abs().max().item() <= 2e-4 * max(want.abs().max().item(), 1.0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        got16 = net(x, t).float()
    assert (got16 - want).abs().max().item() <= 8e-2 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("shape,dim", [((2, 16, 16, 1), 8), ((1, 12, 12, 3), 8)])
def test_whole_network_against_the_numpy_restatement_of_the_flax_unet(shape, dim):
    """fbs_amd/unet.py against oracle/unet_np.py, an independent float64 numpy restatement of fbs/nn/unet.py that reads
    the parameters from ONE flat vector in ravel_pytree order: the torch network's export_flat_params() must be that
    order (names, nesting, (kh, kw, in, out) kernels), and every convention that the reference leaves to flax defaults
    (GroupNorm epsilon 1e-6, tanh gelu, l2norm over the token axis, softmax axes, pixel-shuffle channel order) must
    agree, or the outputs differ."""
    from oracle import unet_np
    torch.manual_seed(3)
    B, H, W, C = shape
    net = UNet(dt=2.0 / 200, dim=dim, in_channels=C, upsampling="pixel_shuffle", dim_mults=(1, 2, 4)).eval()
    with torch.no_grad():                       # make every parameter matter (flax / torch zero-initialise some biases)
        for p_ in net.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    flat = net.export_flat_params().numpy()
    spec = unet_np.param_spec(dim, C, (1, 2, 4), "pixel_shuffle")
    assert sum(int(np.prod(s)) for _, s in spec) == flat.size == net.num_flat_params()
    # the same leaf names in the same order as the torch module's own spec
    assert ["/".join(p_) for p_, _ in spec] == [name for name, _, _ in net.flat_param_spec()]
    x = torch.randn(*shape)
    for t in (0.37, 1.9):
        with torch.no_grad():
            got = net(x, t).numpy().reshape(shape)
        want = unet_np.forward(flat, x.numpy(), t, 2.0 / 200, dim)
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
    # a different flat vector through load_flat_params gives the restatement's answer for that vector
    rng = np.random.default_rng(0)
    flat2 = (flat + 0.02 * rng.normal(size=flat.size)).astype(np.float32)
    net.load_flat_params(flat2)
    with torch.no_grad():
        got = net(x, 0.5).numpy().reshape(shape)
    np.testing.assert_allclose(got, unet_np.forward(flat2, x.numpy(), 0.5, 2.0 / 200, dim), rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 16, 1), (2, 16, 16, 3)])
def test_whole_network_hip_inference_path_against_the_numpy_restatement(shape):
    """The GPU inference path at the configurations' width (dim 64: the MFMA convolution / qkv-attention / 64-channel
    projection kernels and the normalisation kernels are all engaged) directly against oracle/unet_np.py, the float64 numpy
    restatement of the flax UNet -- not only against the eager torch module.  Tolerances per arithmetic: float32 (the
    reference's precision; MIOpen / hipBLASLt accumulate in another order than numpy) 1e-3 of the output scale; bf16 autocast
    (inputs and weights of every matrix product rounded to 8 significant bits, float32 accumulation, ~60 layers) 6e-2."""
    from oracle import unet_np
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, H, W, C = shape
    dim = 64
    net = UNet(dt=2.0 / 200, dim=dim, in_channels=C, upsampling="pixel_shuffle", dim_mults=(1, 2, 4)).eval()
    with torch.no_grad():
        for p_ in net.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
    flat = net.export_flat_params().numpy()
    x = torch.randn(*shape)
    want = unet_np.forward(flat, x.numpy(), 0.37, 2.0 / 200, dim)
    scale = max(float(np.abs(want).max()), 1.0)
    net = net.to(dev)
    xd = x.to(dev)
    with torch.no_grad():
        got32 = net(xd, 0.37).float().cpu().numpy().reshape(shape)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got16 = net(xd, 0.37).float().cpu().numpy().reshape(shape)
    assert np.abs(got32 - want).max() <= 1e-3 * scale, np.abs(got32 - want).max() / scale
    assert np.abs(got16 - want).max() <= 6e-2 * scale, np.abs(got16 - want).max() / scale
    # the bf16 error is rounding noise, not a convention slip: it is small on average too
    assert np.abs(got16 - want).mean() <= 1.5e-2 * scale
