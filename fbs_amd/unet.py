"""The score / drift network of the image experiments as a PyTorch (ROCm) module.

A restatement of the reference's flax UNet (fbs/nn/unet.py:267-368 and the blocks :42-264,
fbs/nn/base.py:44-77, fbs/nn/utils.py:53-57) so that its matmuls / convolutions run on the MFMA
units through PyTorch-ROCm, as BASELINE.json's north_star prescribes; nothing here is hand-written
HIP.  Images stay channels-last at the interface ((B, H, W, C), like flax) and are viewed as NCHW
(channels_last strides, zero copy) inside.

Checkpoints of the reference are ONE flat float32 vector in ``jax.flatten_util.ravel_pytree`` order
(fbs/nn/base.py:29-30): nested dicts flattened with keys sorted at every level, each leaf ravelled
in C order.  ``UNet.flat_param_spec()`` reproduces that order (flax's auto-names ``Dense_0``,
``ResnetBlock_3`` ... for unnamed sub-modules, explicit names otherwise), ``load_flat_params``
fills the torch parameters from such a vector (flax conv kernels are (kh, kw, in, out) -> torch
(out, in, kh, kw); Dense kernels (in, out) -> (out, in)) and ``export_flat_params`` is its inverse.
No JAX checkpoint is available in this environment, so the ordering is PARITY UNPINNED against a
real file; it is pinned by construction rules and round-trip tests only.

Numerical conventions that differ between flax and torch defaults and are set explicitly here:
GroupNorm eps 1e-6 (flax default), LayerNorm eps 1e-5 over the channel axis without bias,
``nn.gelu`` = tanh approximation (flax default), weight standardisation with population variance
and eps 1e-5, ``jax.image.resize(..., 'linear')`` = bilinear with half-pixel centres.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# FBSMI_NN_MFMA=0 keeps the convolutions / projections / qkv-attention on the library paths (MIOpen, hipBLASLt, the round-1
# attention kernel): an A/B switch for timing and for chasing numerical differences; the glue kernels stay on.
_MFMA_KERNELS = os.environ.get("FBSMI_NN_MFMA", "1") != "0"


def sinusoidal_embedding(t: torch.Tensor, out_dim: int = 64, max_period: int = 10_000) -> torch.Tensor:
    """fbs/nn/base.py:44-77."""
    if out_dim % 2 == 1:
        raise NotImplementedError(f'out_dim is implemented for even number only, while {out_dim} is given.')
    half = out_dim // 2
    fs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - 1))
    embs = t.unsqueeze(-1) * fs
    return torch.cat([torch.sin(embs), torch.cos(embs)], dim=-1)


def pixel_shuffle_nhwc(x: torch.Tensor, scale: int) -> torch.Tensor:
    """einops 'b h w (h2 w2 c) -> b (h h2) (w w2) c' (fbs/nn/utils.py:53-57) on an NCHW tensor whose
    channel axis is ordered (h2 w2 c)."""
    B, Cc, H, W = x.shape
    c = Cc // (scale * scale)
    x = x.reshape(B, scale, scale, c, H, W)          # (b, h2, w2, c, h, w)
    x = x.permute(0, 3, 4, 1, 5, 2)                  # (b, c, h, h2, w, w2)
    return x.reshape(B, c, H * scale, W * scale)


class _ChannelLayerNorm(nn.Module):
    """flax nn.LayerNorm(epsilon=1e-5, use_bias=False) over the channel axis of an NCHW tensor."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def fusable(self, x) -> bool:
        C = x.shape[1]
        return x.is_cuda and not torch.is_grad_enabled() and C % 8 == 0 and (C // 8) <= 64 and ((C // 8) & (C // 8 - 1)) == 0

    def forward(self, x, residual=None, xbias=None):
        """LayerNorm over channels of x [+ xbias: the bias of the convolution that produced x] [+ residual: the attention
        block's skip connection], all in the same kernel."""
        C = x.shape[1]
        if xbias is not None and not self.fusable(x):
            x, xbias = x + xbias.to(x.dtype).view(1, -1, 1, 1), None
        if self.fusable(x):
            from . import _lib
            dt = {torch.float32: 0, torch.bfloat16: 1}.get(x.dtype)
            if dt is None:
                x, dt = x.float(), 0
            tok = x.permute(0, 2, 3, 1).contiguous()
            res = residual.to(tok.dtype).permute(0, 2, 3, 1).contiguous() if residual is not None else None
            out = torch.empty_like(tok)
            _lib.call("fbsmi_nn_channel_layernorm", tok.data_ptr(), out.data_ptr(), dt, tok.numel() // C, C,
                      self.scale.data_ptr(), float(self.eps), res.data_ptr() if res is not None else None,
                      xbias.data_ptr() if xbias is not None else None, torch.cuda.current_stream().cuda_stream)
            return _nchw_view(out)
        if residual is not None:
            return self.forward(x) + residual
        if x.is_cuda:   # one layer_norm kernel over the channel axis of the channels_last view
            return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.scale, None, self.eps).permute(0, 3, 1, 2)
        mean = x.mean(dim=1, keepdim=True)
        var = x.var(dim=1, unbiased=False, keepdim=True)
        return (x - mean) * torch.rsqrt(var + self.eps) * self.scale.view(1, -1, 1, 1)


class WeightStandardizedConv(nn.Module):
    """fbs/nn/unet.py:77-124."""

    def __init__(self, dim_in: int, features: int, kernel_size: int = 3, padding: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(dim_in, features, kernel_size, padding=padding)

    def _standardised(self):
        w = self.conv.weight                      # (out, in, kh, kw): statistics per output channel
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        return (w - mean) / torch.sqrt(var + 1e-5)

    def forward(self, x, with_bias: bool = True):
        """x: an NCHW tensor, or a tuple of them standing for their concatenation along the channels."""
        bias = self.conv.bias if with_bias else None
        if torch.is_grad_enabled() or self.training:
            return F.conv2d(_joined(x), self._standardised(), bias, padding=self.conv.padding)
        # inference: the weights are constants, standardise them once (per weight version, device and compute dtype)
        w0 = self.conv.weight
        x0 = _as_parts(x)[0]
        dt = torch.get_autocast_dtype("cuda") if (x0.is_cuda and torch.is_autocast_enabled()) else w0.dtype
        tag = (w0._version, w0.data_ptr(), w0.device, dt)
        if getattr(self, "_w_tag", None) != tag:
            with torch.no_grad():
                w = self._standardised().to(dt)
                if x0.is_cuda and dt != torch.float32:
                    w = w.contiguous(memory_format=torch.channels_last)
            self._w_std, self._w_tag = w, tag
        if self._w_std.dtype == torch.bfloat16 and _conv3x3_fusable(x, self._w_std, self.conv.stride, self.conv.padding):
            return _conv3x3_hip(x, self._w_std, bias)
        return F.conv2d(_joined(x), self._w_std, bias, padding=self.conv.padding)


def _nchw_view(tok):
    """(B, H, W, C) token-major -> NCHW.  bfloat16 stays channels_last (MIOpen's NHWC kernels, the fast path);
    float32 is copied to NCHW-contiguous: MIOpen's float32 NHWC convolutions accumulate with atomics and are not
    reproducible run to run, which a sampler with explicit keys must be."""
    out = tok.permute(0, 3, 1, 2)
    return out.contiguous() if tok.dtype == torch.float32 else out


def _as_parts(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x,)


def _joined(x):
    """A tuple of NCHW tensors stands for their concatenation along the channels (the skip connections of the up path)."""
    return torch.cat(tuple(x), dim=1) if isinstance(x, (tuple, list)) else x


def _channel_chunks(C: int):
    """C (a multiple of 64) as slices of 128 and 64 channels: the widths libfbsmi's convolution multiplies at a time."""
    out, c0 = [], 0
    while C - c0 >= 128:
        out.append((c0, 128))
        c0 += 128
    if C - c0 == 64:
        out.append((c0, 64))
    return out


def _conv3x3_fusable(x, weight, stride=(1, 1), padding=(1, 1)) -> bool:
    """libfbsmi's 3 x 3 convolution takes bfloat16 channels_last inference activations whose channels come in slices of 64 or
    128 -- one tensor or several (their concatenation is then never formed)."""
    parts = _as_parts(x)
    x0 = parts[0]
    if not _MFMA_KERNELS:
        return False
    if not (x0.is_cuda and not torch.is_grad_enabled() and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1)
            and tuple(padding) == (1, 1) and weight.shape[0] % 64 == 0 and sum(t.shape[1] for t in parts) == weight.shape[1]):
        return False
    for t in parts:
        if not (t.dtype == torch.bfloat16 and t.shape[1] % 64 == 0 and t.shape[0] == x0.shape[0] and t.shape[2:] == x0.shape[2:]
                and t.is_contiguous(memory_format=torch.channels_last)):
            return False
    # where it beats MIOpen (tools/bench_conv.py): every 64-channel slice (1.6-2.5x), 128-channel slices when the output is
    # 64 channels wide or the rows are short enough for the 8- or 6-wave tile (1.1-2.4x); wider rows x wide outputs stay
    # with MIOpen
    wide = any(t.shape[1] >= 128 for t in parts)
    if wide and not (weight.shape[0] == 64 or x0.shape[3] <= 32):
        return False
    # ... and only where the kernel has a tile shape for rows this wide (its staged pixel range must fit LDS: W < 248 for
    # 64-channel slices, W <= 100 for 128-channel ones); wider images keep the library convolution
    from . import _lib
    H, W = int(x0.shape[2]), int(x0.shape[3])
    widths = {width for t in parts for _, width in _channel_chunks(t.shape[1])}
    return all(_lib.lib().fbsmi_nn_conv3x3_supported(H, W, width, int(weight.shape[0])) == 1 for width in widths)


def _conv3x3_hip(x, w16, bias):
    """fbsmi_nn_conv3x3 over the channel slices of x (one tensor or a tuple standing for their concatenation); w16 is the
    (Cout, Cin, 3, 3) weight in bfloat16, channels_last memory format."""
    from . import _lib
    parts = _as_parts(x)
    B, _, H, W = parts[0].shape
    cout, cin = w16.shape[0], w16.shape[1]
    out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=parts[0].device)
    st = torch.cuda.current_stream().cuda_stream
    first, off = True, 0
    for t in parts:
        C = t.shape[1]
        for c0, width in _channel_chunks(C):
            _lib.call("fbsmi_nn_conv3x3", t.data_ptr() + 2 * c0, C, w16.data_ptr(), cin, off + c0,
                      bias.data_ptr() if (bias is not None and first) else None, out.data_ptr(), 0 if first else 1,
                      B, H, W, width, cout, st)
            first = False
        off += C
    return out.permute(0, 3, 1, 2)


def _w16_of(conv: nn.Conv2d):
    """The convolution's weight in bfloat16 / channels_last, converted once per weight version."""
    w0 = conv.weight
    tag = (w0._version, w0.data_ptr(), w0.device)
    if getattr(conv, "_w16_tag", None) != tag:
        conv._w16 = w0.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        conv._w16_tag = tag
    return conv._w16


def _proj64_fusable(parts, cout) -> bool:
    chans = tuple(t.shape[1] for t in parts)
    return (_MFMA_KERNELS and os.environ.get("FBSMI_NN_PROJ64", "1") != "0" and cout == 64 and chans in ((64,), (128,), (64, 64)) and not torch.is_grad_enabled()
            and all(t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(memory_format=torch.channels_last) for t in parts))


def _proj64(parts, conv: nn.Conv2d, with_bias=True, ln=None, residual=None):
    """fbsmi_nn_proj64: the 1x1 convolution `conv` (to 64 channels) of the concatenation of `parts`, optionally followed by
    the channel LayerNorm `ln` and a residual, in one kernel.  -> NCHW view of a (B, H, W, 64) tensor."""
    from . import _lib
    a = parts[0]
    b = parts[1] if len(parts) > 1 else None
    B, _, H, W = a.shape
    out = torch.empty((B, H, W, 64), dtype=torch.bfloat16, device=a.device)
    res = None
    if residual is not None:
        res = residual if (residual.dtype == torch.bfloat16 and residual.is_contiguous(memory_format=torch.channels_last)) \
            else residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    _lib.call("fbsmi_nn_proj64", a.data_ptr(), a.shape[1], b.data_ptr() if b is not None else None, b.shape[1] if b is not None else 0,
              _w16_of(conv).data_ptr(), conv.bias.data_ptr() if (with_bias and conv.bias is not None) else None,
              ln.scale.data_ptr() if ln is not None else None, float(ln.eps) if ln is not None else 0.0,
              res.data_ptr() if res is not None else None, out.data_ptr(), B * H * W, torch.cuda.current_stream().cuda_stream)
    return out.permute(0, 3, 1, 2)


def _conv1x1(x, weight):
    """A 1x1 convolution without bias.  On channels_last inference activations it is a plain GEMM over the token-major view
    (rows = pixels), which the BLAS library runs near the memory rate; MIOpen's implicit-GEMM kernel takes 2-3x as long for
    the network's 128 -> 64 projections and zero-fills its output first.  A tuple of inputs stands for their concatenation:
    one GEMM per part, accumulated (addmm), the concatenation is never formed."""
    if isinstance(x, (tuple, list)):
        parts = tuple(x)
        if all(t.is_cuda and t.is_contiguous(memory_format=torch.channels_last) and (t.dtype != torch.float32 or torch.is_autocast_enabled())
               for t in parts) and not torch.is_grad_enabled():
            w2 = weight.reshape(weight.shape[0], weight.shape[1])
            out, off = None, 0
            for t in parts:
                C = t.shape[1]
                tok = t.permute(0, 2, 3, 1)
                if out is None:
                    out = F.linear(tok, w2[:, off:off + C])
                else:
                    out = torch.addmm(out.reshape(-1, out.shape[-1]), tok.reshape(-1, C), w2[:, off:off + C].t()).view(out.shape)
                off += C
            return out.permute(0, 3, 1, 2)
        x = torch.cat(parts, dim=1)
    if (x.is_cuda and not torch.is_grad_enabled() and x.is_contiguous(memory_format=torch.channels_last)
            and (x.dtype != torch.float32 or torch.is_autocast_enabled())):
        return F.linear(x.permute(0, 2, 3, 1), weight.reshape(weight.shape[0], weight.shape[1])).permute(0, 3, 1, 2)
    return F.conv2d(x, weight, None)


def _gn_fusable(norm: nn.GroupNorm) -> bool:
    C, g = norm.num_channels, norm.num_groups
    return C % (8 * g) == 0 and g <= 32 and (C // 8) <= 256 and 256 % (C // 8) == 0


def _tokens(x):
    """NCHW (any strides) -> (B, H, W, C) contiguous; free for channels_last activations."""
    return x.permute(0, 2, 3, 1).contiguous()


def _gn_silu(x, norm: nn.GroupNorm, scale, shift, xbias=None, residual=None, rbias=None):
    """silu(GroupNorm(x + xbias) * (1 + scale) + shift) [+ residual + rbias] in one libfbsmi kernel (include/fbsmi_nn.h); x
    and residual are NCHW (any strides)."""
    from . import _lib
    dt = {torch.float32: 0, torch.bfloat16: 1}.get(x.dtype)
    if dt is None:
        x, dt = x.float(), 0
    B, C, H, W = x.shape
    tok = _tokens(x)
    res = _tokens(residual.to(tok.dtype)) if residual is not None else None
    out = torch.empty_like(tok)
    _lib.call("fbsmi_nn_groupnorm_silu", tok.data_ptr(), out.data_ptr(), dt, B, H * W, C, norm.num_groups,
              norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
              scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None,
              xbias.data_ptr() if xbias is not None else None, res.data_ptr() if res is not None else None,
              rbias.data_ptr() if (rbias is not None and res is not None) else None, torch.cuda.current_stream().cuda_stream)
    return _nchw_view(out)


class ResnetBlock(nn.Module):
    """fbs/nn/unet.py:127-172."""

    def __init__(self, dim_in: int, dim: int, time_dim: int, groups: int = 8):
        super().__init__()
        self.conv_0 = WeightStandardizedConv(dim_in, dim)
        self.norm_0 = nn.GroupNorm(groups, dim, eps=1e-6)
        self.time_mlp = nn.Linear(time_dim, 2 * dim)
        self.conv_1 = WeightStandardizedConv(dim, dim)
        self.norm_1 = nn.GroupNorm(groups, dim, eps=1e-6)
        self.res_conv = nn.Conv2d(dim_in, dim, 1) if dim_in != dim else None

    def forward(self, x, time_emb):
        """x: an NCHW tensor, or (h, skip) standing for cat([h, skip], dim=1) -- the up path hands its skip connections
        over unconcatenated so that the convolutions can take the two parts in turn."""
        x0 = _as_parts(x)[0]
        if isinstance(x, (tuple, list)) and (self.res_conv is None or x0.dtype != torch.bfloat16):
            x = _joined(x)          # only the bfloat16 kernels take the parts in turn
        if x0.is_cuda and not torch.is_grad_enabled() and _gn_fusable(self.norm_0):
            te = self.time_mlp(F.silu(time_emb)).float()
            B = x0.shape[0]
            scale, shift = (p.expand(B, p.shape[1]).contiguous() for p in te.chunk(2, dim=1))
            # the convolution biases are added inside the normalisation kernel (one pass less over the activations)
            h = _gn_silu(self.conv_0(x, with_bias=False), self.norm_0, scale, shift, self.conv_0.conv.bias)
            rbias = None
            if self.res_conv is not None and _proj64_fusable(_as_parts(x), self.res_conv.out_channels):
                x = _proj64(_as_parts(x), self.res_conv)          # both parts and the bias in one kernel
            elif self.res_conv is not None:   # its bias rides on the same kernel as the skip connection it feeds
                x, rbias = _conv1x1(x, self.res_conv.weight), self.res_conv.bias
            # ... and so is the skip connection: x + silu(norm_1(conv_1(h)))
            return _gn_silu(self.conv_1(h, with_bias=False), self.norm_1, None, None, self.conv_1.conv.bias, residual=x,
                            rbias=rbias)
        x = _joined(x)
        h = self.norm_0(self.conv_0(x))
        te = self.time_mlp(F.silu(time_emb))[:, :, None, None]
        scale, shift = te.chunk(2, dim=1)
        h = F.silu(h * (1 + scale) + shift)
        h = F.silu(self.norm_1(self.conv_1(h)))
        if self.res_conv is not None:
            x = self.res_conv(x)
        return x + h


def _l2norm(t, dim, eps=1e-12):
    return t / torch.clamp(torch.linalg.norm(t, ord=2, dim=dim, keepdim=True), min=eps)


class Attention(nn.Module):
    """Full attention of the middle block, fbs/nn/unet.py:175-206."""

    def __init__(self, dim: int, heads: int = 4, dim_head: int = 32, scale: int = 10):
        super().__init__()
        self.heads, self.dim_head, self.scale = heads, dim_head, scale
        self.to_qkv = nn.Conv2d(dim, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Conv2d(heads * dim_head, dim, 1)

    def forward(self, x):
        B, C, H, W = x.shape
        q, k, v = self.to_qkv(x).chunk(3, dim=1)
        # 'b x y (h d) -> b (x y) h d' ; the reference's l2norm uses axis=1, i.e. the (x y) axis
        q, k, v = (t.reshape(B, self.heads, self.dim_head, H * W).permute(0, 3, 1, 2) for t in (q, k, v))
        q, k = _l2norm(q, 1), _l2norm(k, 1)
        sim = torch.einsum('bihd,bjhd->bhij', q, k) * self.scale
        attn = torch.softmax(sim, dim=-1)
        out = torch.einsum('bhij,bjhd->bhid', attn, v)                       # (b, h, xy, d)
        out = out.permute(0, 1, 3, 2).reshape(B, self.heads * self.dim_head, H, W)   # channel = (h d)
        return self.to_out(out)


class LinearAttention(nn.Module):
    """fbs/nn/unet.py:209-245."""

    def __init__(self, dim: int, heads: int = 4, dim_head: int = 32):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = nn.Conv2d(dim, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Conv2d(heads * dim_head, dim, 1)
        self.to_out_norm = _ChannelLayerNorm(dim)

    def forward(self, x, residual=None):
        B, C, H, W = x.shape
        if x.is_cuda and self.dim_head == 32 and not torch.is_grad_enabled():
            core = self._fused_qkv_core(x, B, H, W) if self._qkv_fusable(x) else self._fused_core(self.to_qkv(x), B, H, W)
            if _proj64_fusable((core,), self.to_out.out_channels) and (residual is None or residual.shape[1] == 64):
                return _proj64((core,), self.to_out, ln=self.to_out_norm, residual=residual)   # to_out + norm + skip connection
            y = _conv1x1(core, self.to_out.weight)
            return self.to_out_norm(y, residual, xbias=self.to_out.bias)
        if residual is not None:
            return self.forward(x) + residual
        q, k, v = self.to_qkv(x).chunk(3, dim=1)
        q, k, v = (t.reshape(B, self.heads, self.dim_head, H * W).permute(0, 3, 1, 2) for t in (q, k, v))  # b n h d
        q = torch.softmax(q, dim=-1)            # over the embedding axis
        k = torch.softmax(k, dim=-3)            # over the spatial axis
        q = q / math.sqrt(self.dim_head)
        v = v / (H * W)
        context = torch.einsum('bnhd,bnhe->bhde', k, v)
        out = torch.einsum('bhde,bnhd->bhen', context, q)                    # (b, h, e, n)
        out = out.reshape(B, self.heads * self.dim_head, H, W)               # channel = (h e)
        return self.to_out_norm(self.to_out(out))


def _linear_attention_core(self, qkv, B, H, W):
    """softmaxes, scalings and both einsums of LinearAttention in one libfbsmi kernel (inference, on the GPU)."""
    from . import _lib
    dt = {torch.float32: 0, torch.bfloat16: 1}.get(qkv.dtype)
    if dt is None:
        qkv, dt = qkv.float(), 0
    tok = qkv.permute(0, 2, 3, 1).contiguous()          # (B, H, W, 3hd): free for a channels_last convolution output
    out = torch.empty((B, H, W, self.heads * self.dim_head), dtype=tok.dtype, device=tok.device)
    for b0 in range(0, B, 32768):
        nb = min(32768, B - b0)
        _lib.call("fbsmi_nn_linear_attention", tok[b0:b0 + nb].data_ptr(), out[b0:b0 + nb].data_ptr(), dt, nb, H * W,
                  self.heads, self.dim_head, torch.cuda.current_stream().cuda_stream)
    return _nchw_view(out)                             # NCHW view, channel = (head, e)


def _qkv_fusable(self, x) -> bool:
    """bfloat16 activations (the autocast network), C in {16, 32, 64, 128}: to_qkv runs inside the attention kernel."""
    return (_MFMA_KERNELS and x.dtype == torch.bfloat16 and x.shape[1] in (16, 32, 64, 128)
            and x.is_contiguous(memory_format=torch.channels_last))


def _linear_attention_qkv_core(self, x, B, H, W):
    """to_qkv + softmaxes + both einsums in one libfbsmi kernel on the matrix cores (fbsmi_nn_qkv_linear_attention)."""
    from . import _lib
    w0 = self.to_qkv.weight
    tag = (w0._version, w0.data_ptr(), w0.device)
    if getattr(self, "_w16_tag", None) != tag:     # the projection's weight in bfloat16, once per weight version
        self._w16, self._w16_tag = w0.detach().reshape(w0.shape[0], w0.shape[1]).to(torch.bfloat16).contiguous(), tag
    tok = x.permute(0, 2, 3, 1)                    # (B, H, W, C) view of the channels_last activations
    out = torch.empty((B, H, W, self.heads * self.dim_head), dtype=torch.bfloat16, device=x.device)
    _lib.call("fbsmi_nn_qkv_linear_attention", tok.data_ptr(), self._w16.data_ptr(), out.data_ptr(), B, H * W, x.shape[1],
              self.heads, self.dim_head, torch.cuda.current_stream().cuda_stream)
    return _nchw_view(out)


LinearAttention._fused_core = _linear_attention_core
LinearAttention._fused_qkv_core = _linear_attention_qkv_core
LinearAttention._qkv_fusable = _qkv_fusable


class AttnBlock(nn.Module):
    """fbs/nn/unet.py:248-264."""

    def __init__(self, dim: int, use_linear_attention: bool = True):
        super().__init__()
        self.norm = _ChannelLayerNorm(dim)
        self.attn = LinearAttention(dim) if use_linear_attention else Attention(dim)
        self.linear = use_linear_attention

    def forward(self, x):
        if self.linear:
            return self.attn(self.norm(x), residual=x)     # the skip connection rides on the output norm's kernel
        y = self.attn(self.norm(x)) + x
        if x.is_cuda and x.dtype != torch.float32 and not torch.is_grad_enabled():
            y = y.contiguous(memory_format=torch.channels_last)   # (the full attention's reshape leaves NCHW: keep the network NHWC)
        return y


def _kernel_dtype(x):
    return {torch.float32: 0, torch.bfloat16: 1}.get(x.dtype)


def _add_bias(y, bias):
    """y + bias over the channel axis of an NCHW tensor: in place by fbsmi_nn_bias_add (16-byte vectors) when y is a
    channels_last inference activation, torch's broadcasting add otherwise."""
    dt = _kernel_dtype(y)
    if (y.is_cuda and not torch.is_grad_enabled() and dt is not None and y.shape[1] % 8 == 0
            and y.is_contiguous(memory_format=torch.channels_last)):
        from . import _lib
        _lib.call("fbsmi_nn_bias_add", y.data_ptr(), dt, y.numel() // y.shape[1], y.shape[1], bias.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        return y
    return y + bias.to(y.dtype).view(1, -1, 1, 1)


def _conv_bias(conv: nn.Conv2d, x):
    """conv(x) for a convolution whose consumer is not one of libfbsmi's kernels: at inference on the GPU the bias is added
    by fbsmi_nn_bias_add instead of torch's broadcasting elementwise kernel."""
    if not (x.is_cuda and not torch.is_grad_enabled() and conv.bias is not None):
        return conv(x)
    if _conv3x3_fusable(x, conv.weight, conv.stride, conv.padding):
        return _conv3x3_hip(x, _w16_of(conv), conv.bias)
    return _add_bias(F.conv2d(x, conv.weight, None, conv.stride, conv.padding), conv.bias)


def _conv_pixel_shuffle(conv: nn.Conv2d, x, scale: int):
    """pixel_shuffle_nhwc(conv(x), scale), the bias of the convolution added by the shuffle kernel."""
    c = conv.out_channels // (scale * scale)
    if not (x.is_cuda and not torch.is_grad_enabled() and c % 8 == 0):
        return pixel_shuffle_nhwc(conv(x), scale)
    from . import _lib
    if _conv3x3_fusable(x, conv.weight, conv.stride, conv.padding) and x.shape[1] <= 128:   # (two slices x 1024 outputs: MIOpen's)
        y = _conv3x3_hip(x, _w16_of(conv), None)
    else:
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
    dt = _kernel_dtype(y)
    if dt is None:
        return pixel_shuffle_nhwc(_add_bias(y, conv.bias) if conv.bias is not None else y, scale)
    B, _, H, W = y.shape
    tok = _tokens(y)
    out = torch.empty((B, H * scale, W * scale, c), dtype=tok.dtype, device=tok.device)
    _lib.call("fbsmi_nn_pixel_shuffle", tok.data_ptr(), out.data_ptr(), dt, B, H, W, c, scale,
              conv.bias.data_ptr() if conv.bias is not None else None, torch.cuda.current_stream().cuda_stream)
    return _nchw_view(out)


class Downsample(nn.Module):
    def __init__(self, dim_in: int, dim: int):
        super().__init__()
        self.conv = nn.Conv2d(dim_in, dim, 4, stride=2, padding=1)

    def forward(self, x):
        return _conv_bias(self.conv, x)


class Upsample(nn.Module):
    def __init__(self, dim_in: int, dim: int, method: str = 'resize'):
        super().__init__()
        self.method = method
        if method == 'resize':
            self.convs = nn.ModuleList([nn.Conv2d(dim_in, dim, 3, padding=1)])
        elif method == 'pixel_shuffle':
            self.convs = nn.ModuleList([nn.Conv2d(dim_in, dim_in * 4, 3, padding=1), nn.Conv2d(dim_in, dim, 3, padding=1)])
        else:
            raise ValueError(f'Unknown upsampling method: {method}')

    def forward(self, x):
        if self.method == 'resize':
            x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
            return _conv_bias(self.convs[0], x)
        return _conv_bias(self.convs[1], _conv_pixel_shuffle(self.convs[0], x, 2))


class UNet(nn.Module):
    """fbs/nn/unet.py:267-368.  forward(x (B,H,W,C) or (H,W,C), time scalar or (B,)) -> same shape."""

    def __init__(self, dt: float, dim: int, in_channels: int, upsampling: str = 'resize',
                 init_dim: Optional[int] = None, out_dim: Optional[int] = None,
                 dim_mults: Sequence[int] = (1, 2, 4), resnet_block_groups: int = 8, learnt_variance: bool = False):
        super().__init__()
        self.dt, self.dim, self.dim_mults = dt, dim, tuple(dim_mults)
        init_dim = dim if init_dim is None else init_dim
        self.init_dim = init_dim
        g, td = resnet_block_groups, dim * 4
        self.init_conv = nn.Conv2d(in_channels, init_dim, 7, padding=3)
        self.time_dense_0 = nn.Linear(dim, td)
        self.time_dense_1 = nn.Linear(td, td)
        nres = len(self.dim_mults)
        self.down_res_a, self.down_res_b = nn.ModuleList(), nn.ModuleList()
        self.down_attn, self.down_sample = nn.ModuleList(), nn.ModuleList()
        ch = init_dim
        self._skip_dims: List[int] = [init_dim]
        for ind in range(nres):
            self.down_res_a.append(ResnetBlock(ch, ch, td, g))
            self.down_res_b.append(ResnetBlock(ch, ch, td, g))
            self.down_attn.append(AttnBlock(ch))
            self._skip_dims += [ch, ch]
            if ind < nres - 1:
                self.down_sample.append(Downsample(ch, dim * self.dim_mults[ind]))
                ch = dim * self.dim_mults[ind]
        mid_dim = dim * self.dim_mults[-1]
        self.down_last_conv = nn.Conv2d(ch, mid_dim, 3, padding=1)
        self.mid_res_0 = ResnetBlock(mid_dim, mid_dim, td, g)
        self.mid_attn = AttnBlock(mid_dim, use_linear_attention=False)
        self.mid_res_1 = ResnetBlock(mid_dim, mid_dim, td, g)
        self.up_res_0, self.up_res_1 = nn.ModuleDict(), nn.ModuleDict()
        self.up_attn, self.up_sample = nn.ModuleDict(), nn.ModuleDict()
        skips = list(self._skip_dims)
        for ind in reversed(range(nres)):
            dim_in = dim * self.dim_mults[ind]
            dim_o = dim * self.dim_mults[ind - 1] if ind > 0 else init_dim
            self.up_res_0[str(ind)] = ResnetBlock(dim_in + skips.pop(), dim_in, td, g)
            self.up_res_1[str(ind)] = ResnetBlock(dim_in + skips.pop(), dim_in, td, g)
            self.up_attn[str(ind)] = AttnBlock(dim_in)
            if ind > 0:
                self.up_sample[str(ind)] = Upsample(dim_in, dim_o, upsampling)
        self.up_last_conv = nn.Conv2d(dim * self.dim_mults[0], init_dim, 3, padding=1)
        self.final_res = ResnetBlock(init_dim + skips.pop(), dim, td, g)
        default_out = in_channels * (2 if learnt_variance else 1)
        self.final_conv = nn.Conv2d(dim, default_out if out_dim is None else out_dim, 1)

    # ---------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, time) -> torch.Tensor:
        squeeze = x.dim() < 4
        if squeeze:
            x = x.unsqueeze(0)
        B = x.shape[0]
        xin = x.permute(0, 3, 1, 2)                    # NHWC -> NCHW view (channels_last strides)
        if xin.is_cuda and not torch.is_grad_enabled():
            h = F.conv2d(xin, self.init_conv.weight, None, padding=self.init_conv.padding)
            if h.dtype != torch.float32:
                h = h.contiguous(memory_format=torch.channels_last)   # the whole bf16 network then stays NHWC (MIOpen's fast path)
            h = _add_bias(h, self.init_conv.bias)      # after the layout is settled
        else:
            h = self.init_conv(xin)
        hs = [h]
        time = torch.as_tensor(time, dtype=torch.float32, device=x.device)
        emb = sinusoidal_embedding(time / self.dt, out_dim=self.dim)
        emb = emb.expand(B, self.dim) if emb.dim() < 2 else emb
        emb = self.time_dense_1(F.gelu(self.time_dense_0(emb), approximate='tanh'))
        nres = len(self.dim_mults)
        for ind in range(nres):
            h = self.down_res_a[ind](h, emb)
            hs.append(h)
            h = self.down_attn[ind](self.down_res_b[ind](h, emb))
            hs.append(h)
            if ind < nres - 1:
                h = self.down_sample[ind](h)
        h = _conv_bias(self.down_last_conv, h)
        h = self.mid_res_1(self.mid_attn(self.mid_res_0(h, emb)), emb)
        for ind in reversed(range(nres)):
            h = self.up_res_0[str(ind)]((h, hs.pop()), emb)      # (h, skip) = cat([h, skip], dim=1), formed only if needed
            h = self.up_res_1[str(ind)]((h, hs.pop()), emb)
            h = self.up_attn[str(ind)](h)
            if ind > 0:
                h = self.up_sample[str(ind)](h)
        h = _conv_bias(self.up_last_conv, h)
        out = self.final_conv(self.final_res((h, hs.pop()), emb))
        out = out.permute(0, 2, 3, 1)
        return out[0] if (squeeze or B == 1) else out

    # ---------------------------------------------------------------------------------------------
    # flat parameter vector in ravel_pytree order
    # ---------------------------------------------------------------------------------------------
    def flat_param_spec(self) -> List[Tuple[str, nn.Parameter, str]]:
        """[(flax path, torch parameter, kind)] in ravel_pytree order.  kind: 'conv' (kh,kw,in,out),
        'dense' (in,out), 'vec'."""
        tree = {}

        def put(path, param, kind):
            node = tree
            for p in path[:-1]:
                node = node.setdefault(p, {})
            node[path[-1]] = (param, kind)

        def conv(path, m, bias=True):
            put(path + ['kernel'], m.weight, 'conv')
            if bias and m.bias is not None:
                put(path + ['bias'], m.bias, 'vec')

        def dense(path, m):
            put(path + ['kernel'], m.weight, 'dense')
            put(path + ['bias'], m.bias, 'vec')

        def resblock(path, rb):
            conv(path + ['conv_0'], rb.conv_0.conv)
            put(path + ['norm_0', 'scale'], rb.norm_0.weight, 'vec')
            put(path + ['norm_0', 'bias'], rb.norm_0.bias, 'vec')
            dense(path + ['time_mlp.dense_0'], rb.time_mlp)
            conv(path + ['conv_1'], rb.conv_1.conv)
            put(path + ['norm_1', 'scale'], rb.norm_1.weight, 'vec')
            put(path + ['norm_1', 'bias'], rb.norm_1.bias, 'vec')
            if rb.res_conv is not None:
                conv(path + ['res_conv_0'], rb.res_conv)

        def attnblock(path, ab):
            put(path + ['LayerNorm_0', 'scale'], ab.norm.scale, 'vec')
            inner = path + ['LinearAttention_0' if ab.linear else 'Attention_0']
            conv(inner + ['to_qkv.conv_0'], ab.attn.to_qkv, bias=False)
            conv(inner + ['to_out.conv_0'], ab.attn.to_out)
            if ab.linear:
                put(inner + ['to_out.norm_0', 'scale'], ab.attn.to_out_norm.scale, 'vec')

        P = ['params']
        conv(P + ['init.conv_0'], self.init_conv)
        dense(P + ['Dense_0'], self.time_dense_0)
        dense(P + ['Dense_1'], self.time_dense_1)
        nres = len(self.dim_mults)
        for ind in range(nres):                      # unnamed ResnetBlocks are auto-numbered in call order
            resblock(P + [f'ResnetBlock_{2 * ind}'], self.down_res_a[ind])
            resblock(P + [f'ResnetBlock_{2 * ind + 1}'], self.down_res_b[ind])
            attnblock(P + [f'down_{ind}.attnblock_0'], self.down_attn[ind])
            if ind < nres - 1:
                conv(P + [f'down_{ind}.downsample_0', 'Conv_0'], self.down_sample[ind].conv)
        conv(P + [f'down_{nres - 1}.conv_0'], self.down_last_conv)
        resblock(P + ['mid.resblock_0'], self.mid_res_0)
        attnblock(P + ['mid.attenblock_0'], self.mid_attn)
        resblock(P + ['mid.resblock_1'], self.mid_res_1)
        for ind in range(nres):
            resblock(P + [f'up_{ind}.resblock_0'], self.up_res_0[str(ind)])
            resblock(P + [f'up_{ind}.resblock_1'], self.up_res_1[str(ind)])
            attnblock(P + [f'up_{ind}.attnblock_0'], self.up_attn[str(ind)])
            if ind > 0:
                for j, c in enumerate(self.up_sample[str(ind)].convs):
                    conv(P + [f'up_{ind}.upsample_0', f'Conv_{j}'], c)
        conv(P + ['up_0.conv_0'], self.up_last_conv)
        resblock(P + ['final.resblock_0'], self.final_res)
        conv(P + ['final.conv_0'], self.final_conv)

        out = []

        def walk(node, path):
            for key in sorted(node):                 # ravel_pytree: dict keys in sorted order
                val = node[key]
                if isinstance(val, dict):
                    walk(val, path + [key])
                else:
                    out.append(('/'.join(path + [key]), val[0], val[1]))

        walk(tree, [])
        return out

    def num_flat_params(self) -> int:
        return sum(p.numel() for _, p, _ in self.flat_param_spec())

    @torch.no_grad()
    def load_flat_params(self, vec) -> None:
        vec = torch.as_tensor(np.asarray(vec, np.float32) if not isinstance(vec, torch.Tensor) else vec).reshape(-1)
        if vec.numel() != self.num_flat_params():
            raise ValueError(f"flat parameter vector has {vec.numel()} entries, the network needs "
                             f"{self.num_flat_params()}")
        o = 0
        for _, p, kind in self.flat_param_spec():
            n = p.numel()
            chunk = vec[o:o + n].to(p.device, p.dtype)
            if kind == 'conv':
                out_c, in_c, kh, kw = p.shape
                p.copy_(chunk.reshape(kh, kw, in_c, out_c).permute(3, 2, 0, 1))
            elif kind == 'dense':
                out_f, in_f = p.shape
                p.copy_(chunk.reshape(in_f, out_f).t())
            else:
                p.copy_(chunk.reshape(p.shape))
            o += n

    @torch.no_grad()
    def export_flat_params(self) -> torch.Tensor:
        parts = []
        for _, p, kind in self.flat_param_spec():
            if kind == 'conv':
                parts.append(p.permute(2, 3, 1, 0).reshape(-1))
            elif kind == 'dense':
                parts.append(p.t().reshape(-1))
            else:
                parts.append(p.reshape(-1))
        return torch.cat([q.detach().float().cpu() for q in parts])


def make_st_nn(nn_module: UNet, param=None, device=None):
    """fbs/nn/base.py:9-41: returns (flat parameter vector, None, forward_pass(x, t, param)).  The
    network lives on `device`; `forward_pass` reloads the weights only when handed a different
    vector than the one it currently holds."""
    nn_module = nn_module.to(device) if device is not None else nn_module
    nn_module.eval()
    # the vector whose weights the network currently holds: a strong reference (an id() can be recycled once the object is
    # collected) plus what tells an in-place update apart -- a tensor's version counter, a checksum for anything else
    state = {"obj": None, "stamp": None}

    def stamp_of(p):
        if isinstance(p, torch.Tensor):
            return ("v", p._version)
        a = np.asarray(p)
        return ("s", a.shape, float(a.reshape(-1)[:: max(1, a.size // 4096)].astype(np.float64).sum()))

    def load(p):
        nn_module.load_flat_params(p)
        state["obj"], state["stamp"] = p, stamp_of(p)

    if param is not None:
        load(param)

    @torch.no_grad()
    def forward_pass(x, t, p=None):
        if p is not None and (p is not state["obj"] or stamp_of(p) != state["stamp"]):
            load(p)
        return nn_module(x, t)

    return nn_module.export_flat_params(), None, forward_pass
