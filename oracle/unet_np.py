"""numpy (float64) restatement of the reference's flax UNet -- TEST INFRASTRUCTURE, independent of torch and of
fbs_amd/unet.py: it exists so that the torch restatement is checked against something that is not torch.

Follows /root/reference/fbs/nn/unet.py:267-368 (UNet), :127-172 (ResnetBlock), :77-124 (WeightStandardizedConv),
:175-206 (Attention, incl. its l2norm over axis 1 = the token axis, :23-39), :209-245 (LinearAttention), :248-264
(AttnBlock), :42-74 (Down/Upsample), fbs/nn/base.py:44-77 (sinusoidal_embedding), fbs/nn/utils.py:53-57 (PixelShuffle),
and flax.linen's defaults for what the reference leaves implicit: nn.GroupNorm epsilon 1e-6 with biased variance,
nn.LayerNorm as called (epsilon 1e-5, no bias), nn.gelu in its tanh form, nn.swish = x sigmoid(x), nn.Conv with integer
padding on both sides and kernels stored (kh, kw, in, out), nn.Dense kernels (in, out).

Parameters come as ONE flat float vector in jax.flatten_util.ravel_pytree order (fbs/nn/base.py:29-30): the leaves of
the nested parameter dict in sorted-key order at every level.  Module names are the reference's (`name=` arguments) or
flax's automatic ones for unnamed submodules (Dense_0, Dense_1, ResnetBlock_0.., Conv_0, LayerNorm_0,
LinearAttention_0 / Attention_0).  parity unpinned against flax itself (not installable here).
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------
# parameter tree
# ------------------------------------------------------------------------------------------------
def param_spec(dim, in_channels, dim_mults=(1, 2, 4), upsampling="pixel_shuffle", init_dim=None, out_dim=None, heads=4,
               dim_head=32):
    """[(path tuple, shape)] of every parameter leaf, in ravel_pytree order."""
    init_dim = dim if init_dim is None else init_dim
    leaves = []

    def conv(path, kh, cin, cout, bias=True):
        leaves.append((path + ("kernel",), (kh, kh, cin, cout)))
        if bias:
            leaves.append((path + ("bias",), (cout,)))

    def dense(path, cin, cout):
        leaves.append((path + ("kernel",), (cin, cout)))
        leaves.append((path + ("bias",), (cout,)))

    def resblock(path, cin, cout):
        conv(path + ("conv_0",), 3, cin, cout)
        leaves.append((path + ("norm_0", "scale"), (cout,)))
        leaves.append((path + ("norm_0", "bias"), (cout,)))
        dense(path + ("time_mlp.dense_0",), 4 * dim, 2 * cout)
        conv(path + ("conv_1",), 3, cout, cout)
        leaves.append((path + ("norm_1", "scale"), (cout,)))
        leaves.append((path + ("norm_1", "bias"), (cout,)))
        if cin != cout:
            conv(path + ("res_conv_0",), 1, cin, cout)

    def attnblock(path, c, linear):
        leaves.append((path + ("LayerNorm_0", "scale"), (c,)))
        inner = path + (("LinearAttention_0" if linear else "Attention_0"),)
        conv(inner + ("to_qkv.conv_0",), 1, c, 3 * heads * dim_head, bias=False)
        conv(inner + ("to_out.conv_0",), 1, heads * dim_head, c)
        if linear:
            leaves.append((inner + ("to_out.norm_0", "scale"), (c,)))

    P = ("params",)
    conv(P + ("init.conv_0",), 7, in_channels, init_dim)
    dense(P + ("Dense_0",), dim, 4 * dim)
    dense(P + ("Dense_1",), 4 * dim, 4 * dim)
    R = len(dim_mults)
    ch = init_dim
    for ind in range(R):
        resblock(P + (f"ResnetBlock_{2 * ind}",), ch, ch)
        resblock(P + (f"ResnetBlock_{2 * ind + 1}",), ch, ch)
        attnblock(P + (f"down_{ind}.attnblock_0",), ch, True)
        if ind < R - 1:
            conv(P + (f"down_{ind}.downsample_0", "Conv_0"), 4, ch, dim * dim_mults[ind])
            ch = dim * dim_mults[ind]
    mid = dim * dim_mults[-1]
    conv(P + (f"down_{R - 1}.conv_0",), 3, ch, mid)
    resblock(P + ("mid.resblock_0",), mid, mid)
    attnblock(P + ("mid.attenblock_0",), mid, False)
    resblock(P + ("mid.resblock_1",), mid, mid)
    for ind in reversed(range(R)):
        din = dim * dim_mults[ind]
        dout = dim * dim_mults[ind - 1] if ind > 0 else init_dim
        resblock(P + (f"up_{ind}.resblock_0",), din + dout, din)
        resblock(P + (f"up_{ind}.resblock_1",), din + dout, din)
        attnblock(P + (f"up_{ind}.attnblock_0",), din, True)
        if ind > 0:
            if upsampling == "pixel_shuffle":
                conv(P + (f"up_{ind}.upsample_0", "Conv_0"), 3, din, 4 * din)
                conv(P + (f"up_{ind}.upsample_0", "Conv_1"), 3, din, dout)
            else:
                conv(P + (f"up_{ind}.upsample_0", "Conv_0"), 3, din, dout)
    conv(P + ("up_0.conv_0",), 3, dim * dim_mults[0], init_dim)
    resblock(P + ("final.resblock_0",), 2 * init_dim, dim)
    conv(P + ("final.conv_0",), 1, dim, in_channels if out_dim is None else out_dim)
    return sorted(leaves, key=lambda lv: lv[0])          # nested dicts with sorted keys == lexicographic paths


def unravel(flat, spec):
    flat = np.asarray(flat, np.float64).reshape(-1)
    out, o = {}, 0
    for path, shape in spec:
        n = int(np.prod(shape))
        out[path] = flat[o:o + n].reshape(shape)
        o += n
    if o != flat.size:
        raise ValueError(f"the flat vector has {flat.size} entries, the network {o}")
    return out


# ------------------------------------------------------------------------------------------------
# layers (NHWC, float64)
# ------------------------------------------------------------------------------------------------
def conv2d(x, kernel, bias=None, stride=1, pad=0):
    kh, kw, cin, cout = kernel.shape
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))[:, ::stride, ::stride]   # (B,Ho,Wo,C,kh,kw)
    y = np.einsum("bhwcij,ijco->bhwo", win, kernel, optimize=True)
    return y if bias is None else y + bias


def ws_conv(x, kernel, bias):                                              # unet.py:77-124
    mean = kernel.mean(axis=(0, 1, 2), keepdims=True)
    var = kernel.var(axis=(0, 1, 2), keepdims=True)
    return conv2d(x, (kernel - mean) / np.sqrt(var + 1e-5), bias, pad=1)


def group_norm(x, scale, bias, groups=8, eps=1e-6):                        # flax.linen.GroupNorm defaults
    B, H, W, C = x.shape
    g = x.reshape(B, H * W, groups, C // groups)
    mean = g.mean(axis=(1, 3), keepdims=True)
    var = g.var(axis=(1, 3), keepdims=True)
    return ((g - mean) / np.sqrt(var + eps)).reshape(B, H, W, C) * scale + bias


def layer_norm(x, scale, eps=1e-5):
    mean = x.mean(axis=-1, keepdims=True)
    var = x.var(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * scale


def swish(x):
    return x / (1.0 + np.exp(-x))


def gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def sinusoidal_embedding(t, out_dim, max_period=10_000):                  # base.py:44-77
    half = out_dim // 2
    fs = np.exp(-math.log(max_period) * np.arange(half) / (half - 1))
    e = np.asarray(t, np.float64)[..., None] * fs
    return np.concatenate([np.sin(e), np.cos(e)], axis=-1)


def pixel_shuffle(x, s=2):                                                 # 'b h w (h2 w2 c) -> b (h h2) (w w2) c'
    B, H, W, C = x.shape
    c = C // (s * s)
    return x.reshape(B, H, W, s, s, c).transpose(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, c)


def forward(flat, x, time, dt, dim, dim_mults=(1, 2, 4), upsampling="pixel_shuffle", heads=4, dim_head=32):
    """UNet(dt, dim, upsampling).apply(unravel(flat), x, time) for x (B, H, W, C) and a scalar time."""
    x = np.asarray(x, np.float64)
    B, H, W, C = x.shape
    p = unravel(flat, param_spec(dim, C, dim_mults, upsampling, heads=heads, dim_head=dim_head))
    P = ("params",)
    g = lambda *path: p[P + path]

    def resblock(name, h, temb):                                           # unet.py:127-172
        cout = g(name, "conv_0", "bias").shape[0]
        y = ws_conv(h, g(name, "conv_0", "kernel"), g(name, "conv_0", "bias"))
        y = group_norm(y, g(name, "norm_0", "scale"), g(name, "norm_0", "bias"))
        te = swish(temb) @ g(name, "time_mlp.dense_0", "kernel") + g(name, "time_mlp.dense_0", "bias")
        scale, shift = te[:, None, None, :cout], te[:, None, None, cout:]
        y = swish(y * (1 + scale) + shift)
        y = ws_conv(y, g(name, "conv_1", "kernel"), g(name, "conv_1", "bias"))
        y = swish(group_norm(y, g(name, "norm_1", "scale"), g(name, "norm_1", "bias")))
        if h.shape[-1] != cout:
            h = conv2d(h, g(name, "res_conv_0", "kernel"), g(name, "res_conv_0", "bias"))
        return h + y

    def split_heads(t):                                                    # 'b x y (h d) -> b (x y) h d'
        b, hh, ww, _ = t.shape
        return t.reshape(b, hh * ww, heads, dim_head)

    def attnblock(name, h, linear):                                        # unet.py:175-264
        b, hh, ww, c = h.shape
        xn = layer_norm(h, g(name, "LayerNorm_0", "scale"))
        inner = "LinearAttention_0" if linear else "Attention_0"
        qkv = conv2d(xn, g(name, inner, "to_qkv.conv_0", "kernel"))
        hd = heads * dim_head
        q, k, v = (split_heads(qkv[..., i * hd:(i + 1) * hd]) for i in range(3))
        if linear:
            q = softmax(q, -1) / math.sqrt(dim_head)
            k = softmax(k, -3)
            v = v / (hh * ww)
            ctx = np.einsum("bnhd,bnhe->bhde", k, v)
            out = np.einsum("bhde,bnhd->bhen", ctx, q)                     # (b, heads, e, n)
            out = out.transpose(0, 3, 1, 2).reshape(b, hh, ww, hd)         # 'b h e (x y) -> b x y (h e)'
        else:
            l2 = lambda t: t / np.clip(np.linalg.norm(t, axis=1, keepdims=True), 1e-12, None)   # axis 1: the tokens
            q, k = l2(q), l2(k)
            sim = np.einsum("bihd,bjhd->bhij", q, k) * 10
            out = np.einsum("bhij,bjhd->bhid", softmax(sim, -1), v)       # (b, heads, n, d)
            out = out.transpose(0, 2, 1, 3).reshape(b, hh, ww, hd)         # 'b h (x y) d -> b x y (h d)'
        out = conv2d(out, g(name, inner, "to_out.conv_0", "kernel"), g(name, inner, "to_out.conv_0", "bias"))
        if linear:
            out = layer_norm(out, g(name, inner, "to_out.norm_0", "scale"))
        return out + h

    init_dim = dim
    h = conv2d(x, g("init.conv_0", "kernel"), g("init.conv_0", "bias"), pad=3)
    hs = [h]
    temb = np.broadcast_to(sinusoidal_embedding(float(time) / dt, dim), (B, dim))
    temb = temb @ g("Dense_0", "kernel") + g("Dense_0", "bias")
    temb = gelu_tanh(temb) @ g("Dense_1", "kernel") + g("Dense_1", "bias")
    R = len(dim_mults)
    for ind in range(R):
        h = resblock(f"ResnetBlock_{2 * ind}", h, temb)
        hs.append(h)
        h = resblock(f"ResnetBlock_{2 * ind + 1}", h, temb)
        h = attnblock(f"down_{ind}.attnblock_0", h, True)
        hs.append(h)
        if ind < R - 1:
            h = conv2d(h, g(f"down_{ind}.downsample_0", "Conv_0", "kernel"), g(f"down_{ind}.downsample_0", "Conv_0", "bias"),
                       stride=2, pad=1)
    h = conv2d(h, g(f"down_{R - 1}.conv_0", "kernel"), g(f"down_{R - 1}.conv_0", "bias"), pad=1)
    h = resblock("mid.resblock_0", h, temb)
    h = attnblock("mid.attenblock_0", h, False)
    h = resblock("mid.resblock_1", h, temb)
    for ind in reversed(range(R)):
        h = resblock(f"up_{ind}.resblock_0", np.concatenate([h, hs.pop()], -1), temb)
        h = resblock(f"up_{ind}.resblock_1", np.concatenate([h, hs.pop()], -1), temb)
        h = attnblock(f"up_{ind}.attnblock_0", h, True)
        if ind > 0:
            u = f"up_{ind}.upsample_0"
            if upsampling != "pixel_shuffle":
                raise NotImplementedError("only the pixel_shuffle upsampling of the shipped experiments is restated")
            h = conv2d(h, g(u, "Conv_0", "kernel"), g(u, "Conv_0", "bias"), pad=1)
            h = conv2d(pixel_shuffle(h), g(u, "Conv_1", "kernel"), g(u, "Conv_1", "bias"), pad=1)
    h = conv2d(h, g("up_0.conv_0", "kernel"), g("up_0.conv_0", "bias"), pad=1)
    out = resblock("final.resblock_0", np.concatenate([h, hs.pop()], -1), temb)
    return conv2d(out, g("final.conv_0", "kernel"), g("final.conv_0", "bias"))
