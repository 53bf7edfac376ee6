/* fbsmi_nn.h -- device kernels for the score network of the image experiments (SURVEY.md section 8 f1:
 * the flax UNet of fbs/nn/unet.py restated in torch, fbs_amd/unet.py).  Not part of the sampler hot
 * path (include/fbsmi.h); same conventions: extern "C", device pointers, int status, void* stream. */
#ifndef FBSMI_NN_H
#define FBSMI_NN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* LinearAttention core of fbs/nn/unet.py:209-245 for dim_head = 32:
 *   q = softmax(q, over the embedding) / sqrt(dim_head);  k = softmax(k, over the tokens);  v = v / n
 *   context[d][e] = sum_n k[n][d] v[n][e];   out[n][e] = sum_d context[d][e] q[n][d]
 * qkv: (B, n, 3 * heads * 32) token-major (a channels_last (B, 3hd, H, W) convolution output viewed as
 * (B, H*W, 3hd)); channel = which * heads * 32 + head * 32 + d, which = 0 q, 1 k, 2 v.
 * out: (B, n, heads * 32), channel = head * 32 + e.  dtype: 0 float32, 1 bfloat16 (computed in float32). */
int fbsmi_nn_linear_attention(const void* qkv, void* out, int dtype, int64_t B, int32_t n, int32_t heads,
                              int32_t dim_head, void* stream);

/* The same with the to_qkv projection in front (a 1x1 convolution without bias, fbs/nn/unet.py:219-221), bfloat16 on the
 * matrix cores: qkv = xn W^T is formed tile by tile inside the kernel and never written.
 * xn: (B, n, C) token-major bfloat16 (the PreNorm output); w: (3 * heads * 32, C) bfloat16, row = which * heads * 32 + head * 32
 * + d (the convolution's weight); out: (B, n, heads * 32) bfloat16.  C in {16, 32, 64, 128}. */
int fbsmi_nn_qkv_linear_attention(const void* xn, const void* w, void* out, int64_t B, int32_t n, int32_t C, int32_t heads,
                                  int32_t dim_head, void* stream);

/* 3 x 3 convolution, stride 1, zero padding 1, bfloat16 on the matrix cores (float32 accumulation), for the network's layers
 * whose input comes in slices of 64 or 128 channels:
 *   y[b, i, j, :] (+)= bias + sum over taps and the Cin channels of this slice of w[:, tap, ci_off + c] x[b, i + di, j + dj, c]
 * x: (B, H, W, .) token-major with `xstride` elements between pixels (a channel slice of a wider tensor: point x at the
 * slice's first channel), w: (Cout, 3, 3, wstride) (a torch weight in channels_last memory format; wstride = its full input
 * width, ci_off = where this slice's channels start), bias (Cout) float32 or NULL, y (B, H, W, Cout); accumulate != 0 adds to
 * y instead of overwriting it (the later slices of a wide or concatenated input: conv(cat(a, b)) = conv_a(a) + conv_b(b), so
 * the concatenation is never formed).  Cin in {64, 128}, Cout a multiple of 64. */
int fbsmi_nn_conv3x3(const void* x, int32_t xstride, const void* w, int32_t wstride, int32_t ci_off, const float* bias, void* y,
                     int accumulate, int64_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream);

/* 1 when fbsmi_nn_conv3x3 has a tile shape for rows of W pixels and slices of Cin channels (the staged pixel range must fit
 * 160 KB of LDS beside the weights: W < 248 at Cin = 64, W <= 100 at Cin = 128), else 0: the caller then keeps the library
 * convolution.  No GPU work. */
int fbsmi_nn_conv3x3_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout);

/* 1x1 projection to 64 channels, bfloat16 on the matrix cores, with its consumer folded in:
 *   y = [LN_c]( a Wa^T [+ b Wb^T] [+ bias] ) [* ln_scale] [+ residual]
 * a (npix, Ca), b (npix, Cb) or NULL: token-major inputs whose concatenation w (64, Ca + Cb) multiplies ((Ca, Cb) in
 * {(64, 0), (128, 0), (64, 64)}); bias (64) float32 or NULL; ln_scale (64) float32 or NULL: when given, the channel
 * LayerNorm (no bias, eps) of LinearAttention's to_out (fbs/nn/unet.py:228-232) is applied to the biased product;
 * residual (npix, 64) or NULL is added last; y (npix, 64). */
int fbsmi_nn_proj64(const void* a, int32_t Ca, const void* b, int32_t Cb, const void* w, const float* bias,
                    const float* ln_scale, float eps, const void* residual, void* y, int64_t npix, void* stream);

/* GroupNorm (biased variance, eps) + per-image channel modulation + SiLU in one pass over the activations --
 * the two normalisation sites of ResnetBlock (fbs/nn/unet.py:127-172):
 *   x' = x + xbias[c]   (the bias of the convolution that produced x, folded in here; NULL: none)
 *   y = silu( ((x' - mean_g) * rsqrt(var_g + eps) * gamma[c] + beta[c]) * (1 + scale[b][c]) + shift[b][c] ) [+ residual + rbias[c]]
 * x, y, residual (NULL: none; same layout and dtype as x; the block's skip connection, rbias (NULL: none) the bias of the 1x1
 * convolution that produced it): (B, n, C) token-major
 * (channels_last); C a multiple of 8 * groups; gamma, beta, xbias: (C) float32;
 * scale, shift: (B, C) float32 or NULL (no modulation).  dtype: 0 float32, 1 bfloat16 (statistics in float32,
 * Welford / Chan merging). */
int fbsmi_nn_groupnorm_silu(const void* x, void* y, int dtype, int64_t B, int32_t n, int32_t C, int32_t groups,
                            const float* gamma, const float* beta, float eps, const float* scale, const float* shift,
                            const float* xbias, const void* residual, const float* rbias, void* stream);

/* LayerNorm over the channel axis without bias (flax nn.LayerNorm(epsilon, use_bias=False), fbs/nn/unet.py:
 * the PreNorm of every attention block and LinearAttention's output norm):
 *   x' = x + xbias[c];  y[r][c] = (x'[r][c] - mean_r) * rsqrt(var_r + eps) * scale[c] [+ residual[r][c]],  biased variance
 * over the C channels of row r.  x, y, residual (NULL: none): (rows, C) with C / 8 a power of two <= 64; xbias (NULL: none):
 * (C) float32, the bias of the convolution that produced x; dtype 0 float32, 1 bfloat16 (statistics in float32). */
int fbsmi_nn_channel_layernorm(const void* x, void* y, int dtype, int64_t rows, int32_t C, const float* scale, float eps,
                               const void* residual, const float* xbias, void* stream);

/* y[r][c] += bias[c] in place, (rows, C) token-major, C a multiple of 8: the bias of a channels_last convolution whose
 * consumer is not one of the kernels above. */
int fbsmi_nn_bias_add(void* y, int dtype, int64_t rows, int32_t C, const float* bias, void* stream);

/* einops 'b h w (h2 w2 c) -> b (h h2) (w w2) c' (fbs/nn/utils.py:53-57; the pixel_shuffle upsampling of fbs/nn/unet.py) with
 * the bias (NULL: none; (s*s*c) float32) of the convolution that produced x added on the way.
 * x: (B, H, W, s*s*c) token-major, y: (B, s*H, s*W, c); c a multiple of 8. */
int fbsmi_nn_pixel_shuffle(const void* x, void* y, int dtype, int64_t B, int32_t H, int32_t W, int32_t c, int32_t s,
                           const float* bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif
