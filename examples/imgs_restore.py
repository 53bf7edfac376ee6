"""Image inpainting / super-resolution with the forward-backward samplers on MI355X.

Counterpart of the reference drivers experiments/imgs/inpainting.py and experiments/imgs/supr.py (they differ by the
mask only; here `--task inpaint` / `--task supr`) and, with `--sb`, of experiments/sb_imgs/supr.py: same command-line
flags, same key schedule, same result arrays (`*-gibbs-eb-ef.npy`, `*-filter.npy`, `*-pmcmc-<delta>.npy` of shape
(nsamples, H, W, C)), written against fbs_amd.  `--method twisted` is experiments/imgs/inpainting_twisted.py (the twisted-SMC
baseline: whole-image particles, the twisting function's gradient through the score network by torch.autograd; result
`*-twisted.npy`), `--method csgm` experiments/imgs/inpainting_csgm.py / supr_csgm.py (conditional score-based sampling: one
reverse-SDE trajectory with the observed pixels re-noised every step; `*-csgm.npy`).  The closures are the bound methods of one fbs_amd.score.ScoreBridge, so
every SMC step is two HIP kernels around one network evaluation (PyTorch-ROCm).

The reference loads a trained checkpoint (`./checkpoints/<dataset>_<sde>_<epoch>.npz`, a flat `param` / `ema_param`
vector in ravel_pytree order) and a dataset (`../datasets/mnist.npz`, `datasets/celeba_hq<res>.npy`); neither exists in
this environment.  If the files are there they are used (fbs_amd.unet.UNet.load_flat_params); otherwise the network is
randomly initialised and the test image is uniform noise -- the sampler runs the same either way (PNG output is left
out: matplotlib is not a dependency).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd import ops  # noqa: E402
from fbs_amd.images import ImageRestore  # noqa: E402
from fbs_amd.samplers import gibbs_init, gibbs_kernel, pmcmc_kernel, stratified  # noqa: E402
from fbs_amd.score import ScoreBridge  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE  # noqa: E402
from fbs_amd.unet import UNet  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description='Inpainting / super-resolution.')
    p.add_argument('--task', type=str, default='inpaint', help='inpaint (inpainting.py) or supr (supr.py).')
    p.add_argument('--sb', action='store_true', help='Schrodinger-bridge model (experiments/sb_imgs/supr.py).')
    p.add_argument('--dataset', type=str, default='mnist', help="'mnist' or 'celeba-64' / 'celeba-128'.")
    p.add_argument('--rect_size', type=int, default=15, help='The w/h of the inpainting rectangle.')
    p.add_argument('--rate', type=int, default=4, help='The rate of super-resolution.')
    p.add_argument('--sde', type=str, default='lin')
    p.add_argument('--test_nsteps', type=int, default=200)
    p.add_argument('--test_epoch', type=int, default=2999)
    p.add_argument('--sb_step', type=int, default=9)
    p.add_argument('--test_ema', action='store_true', default=False)
    p.add_argument('--test_seed', type=int, default=666)
    p.add_argument('--ny0s', type=int, default=10)
    p.add_argument('--start_from', type=int, default=0)
    p.add_argument('--nparticles', type=int, default=100)
    p.add_argument('--nsamples', type=int, default=100)
    p.add_argument('--method', type=str, default='gibbs-eb-ef', help="'filter', 'gibbs[-eb][-ef]', 'pmcmc[-delta]', 'twisted', 'csgm'.")
    p.add_argument('--init_method', type=str, default='filter')
    p.add_argument('--marg', action='store_true', default=False, help='Whether marginalise out the Y path.')
    p.add_argument('--dim', type=int, default=64, help='UNet width (64 in the reference).')
    p.add_argument('--fp32', action='store_true', help='float32 network instead of bf16 autocast.')
    p.add_argument('--chunk', type=int, default=4096, help='Particles per network call (a power of two: MIOpen ships kernels for those batch sizes; bigger calls are faster per particle).')
    p.add_argument('--outdir', type=str, default='./imgs/results')
    p.add_argument('--quiet', action='store_true')
    args = p.parse_args(argv)

    dev = torch.device('cuda:0')
    resolution = 28 if args.dataset == 'mnist' else int(args.dataset.split('-')[-1])
    nchannels = 1 if args.dataset == 'mnist' else 3
    task = f'inpaint-{args.rect_size}' if args.task == 'inpaint' else f'supr-{args.rate}'
    key = ops.PRNGKey(args.test_seed)                                              # inpainting.py:54-55
    key, data_key = ops.split(key)
    T = 0.5 if args.sb else 2.0                                                    # sb_imgs/supr.py:46 / inpainting.py:57
    nsteps = args.test_nsteps
    ts = np.linspace(0, T, nsteps + 1)
    key, subkey = ops.split(key)                                                   # the dataset's key (unused without data)
    ds = ImageRestore(task, (resolution, resolution, nchannels), sr_random=not args.sb, device=dev)
    sde = StationaryConstLinearSDE(a=-0.5, b=1.) if args.sde == 'const' else \
        StationaryLinLinearSDE(beta_min=0.02, beta_max=5., t0=0., T=T)
    key, subkey = ops.split(key)                                                   # the network's init key

    # the trained model(s): flat parameter vectors in ravel_pytree order, if present
    torch.manual_seed(args.test_seed)
    mk = lambda: UNet(dt=T / 200, dim=args.dim, in_channels=nchannels, upsampling='pixel_shuffle').to(dev).eval()
    net, net_fwd = mk(), (mk() if args.sb else None)
    ck = f'./checkpoints/sb_{args.dataset}_{args.sde}_{args.sb_step}.npz' if args.sb else \
        f'./checkpoints/{args.dataset}_{args.sde}_{args.test_epoch}.npz'
    if os.path.exists(ck):
        z = np.load(ck)
        if args.sb:
            net_fwd.load_flat_params(z['param_fwd'])
            net.load_flat_params(z['param_bwd'])
        else:
            net.load_flat_params(z['ema_param' if args.test_ema else 'param'])
    elif not args.quiet:
        print(f'no checkpoint at {ck}: randomly initialised network')

    def run(module, x, t):
        with torch.no_grad():
            if args.fp32:
                return module(x, t)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                return module(x, t)

    sb = ScoreBridge(lambda x, t: run(net, x, t), ds, sde, ts, chunk=args.chunk, mode='drift' if args.sb else 'score',
                     net_input_dtype=torch.float32 if args.fp32 else torch.bfloat16,
                     fwd_drift_fn=(lambda x, t: run(net_fwd, x.unsqueeze(0), t).float().reshape(x.shape)) if args.sb else None)
    x_shape = tuple(ds.unobs_shape)
    delta = float(args.method.split('-')[-1]) if ('pmcmc' in args.method and len(args.method.split('-')) > 1) else None
    eb, ef = ('eb' in args.method, 'ef' in args.method) if 'gibbs' in args.method else (True, True)
    if args.sb:
        eb = ef = True                                                             # sb_imgs/supr.py:169-172
    cl = dict(transition_sampler=sb.transition_sampler, transition_logpdf=sb.transition_logpdf,
              likelihood_logpdf=sb.likelihood_logpdf)
    os.makedirs(args.outdir, exist_ok=True)
    out = None
    for k in range(args.ny0s):
        data_key, subkey = ops.split(data_key)
        if k < args.start_from:
            continue
        k_img, k_mask = ops.split(subkey)                                          # dataset.sampler: an image and a mask
        test_img = ops.uniform(k_img, (resolution, resolution, nchannels), device=dev)
        mask = ds.gen_mask(k_mask)
        _, test_y0 = ds.unpack(test_img, mask)
        head = os.path.join(args.outdir, f'{args.dataset}-{task}-{args.sde}-{args.nparticles}-{k}')
        np.savez(head + '-true', test_img=test_img.cpu().numpy())
        restored = np.zeros((args.nsamples, resolution, resolution, nchannels), np.float32)
        to_img = lambda x0: ds.concat(x0.unsqueeze(0), test_y0, mask)[0].cpu().numpy()
        if args.method == 'filter':
            for i in range(args.nsamples):
                key, subkey = ops.split(key)
                x0, _ = gibbs_init(subkey, test_y0, x_shape, ts, sb.fwd_sampler, sde, sb.unpack, nparticles=args.nparticles,
                                   method='filter', marg_y=args.marg, mask_=mask, **cl)
                restored[i] = to_img(x0)
            np.save(head + f'-filter{"-marg" if args.marg else ""}', restored)
        elif 'gibbs' in args.method:
            key, subkey = ops.split(key)
            x0, us_star = gibbs_init(subkey, test_y0, x_shape, ts, sb.fwd_sampler, sde, sb.unpack,
                                     nparticles=args.nparticles, method=args.init_method, marg_y=args.marg, mask_=mask, **cl)
            bs_star = np.zeros(nsteps + 1, np.int32)
            np.save(head + '-gibbs-init', to_img(x0))
            for i in range(args.nsamples):
                key, subkey = ops.split(key)
                x0, us_star, bs_star, acc = gibbs_kernel(subkey, x0, test_y0, us_star, bs_star, ts, sb.fwd_sampler, sde,
                                                         sb.unpack, args.nparticles, marg_y=args.marg,
                                                         explicit_backward=eb, explicit_final=ef, mask_=mask, **cl)
                restored[i] = to_img(x0)
                if not args.quiet:
                    print(f'{task} | Gibbs | iter: {i}, acc: {float(acc.float().mean()):.3f}')
            np.save(head + f'-gibbs{"-eb" if eb else ""}{"-ef" if ef else ""}{"-marg" if args.marg else ""}', restored)
        elif args.method == 'twisted':                                               # inpainting_twisted.py:148-191
            from fbs_amd.twisted import make_image_twisted

            def score(uv, t):                                                        # differentiable: no no_grad here
                if args.fp32:
                    return net(uv, t).reshape(uv.shape)
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    return net(uv, t).float().reshape(uv.shape)

            tw = make_image_twisted(score, ds, sde, ts, args.nparticles)
            for i in range(args.nsamples):
                key, subkey = ops.split(key)
                sample = tw.conditional_sampler(subkey, test_y0, stratified, mask_=mask)
                restored[i] = sample.cpu().numpy()
                if not args.quiet:
                    print(f'{task} | twisted | iter: {i}')
            np.save(head + '-twisted', restored)
        elif args.method == 'csgm':                                                  # inpainting_csgm.py:147-158
            from fbs_amd.csgm import make_image_csgm
            sampler = make_image_csgm(lambda x, t: run(net, x, t), ds, sde, ts)
            for i in range(args.nsamples):
                key, subkey = ops.split(key)
                restored[i] = to_img(sampler(subkey, test_y0, mask))
                if not args.quiet:
                    print(f'{task} | cSGM | iter: {i}')
            np.save(head + '-csgm', restored)
        elif 'pmcmc' in args.method:
            key, subkey = ops.split(key)
            x0, log_ell, ys = torch.zeros(x_shape, device=dev), 0., sb.fwd_ys_sampler(subkey, test_y0)
            for i in range(args.nsamples):
                key, subkey = ops.split(key)
                x0, log_ell, ys, st = pmcmc_kernel(subkey, x0, log_ell, ys, test_y0, ts, sb.fwd_ys_sampler, sde,
                                                   sb.ref_sampler, sb.transition_sampler, sb.likelihood_logpdf, stratified,
                                                   args.nparticles, delta=delta, mask_=mask)
                restored[i] = to_img(x0)
                if not args.quiet:
                    print(f'{task} | pMCMC {delta} | iter: {i}, acc_prob: {float(st.acceptance_prob):.3f}')
            np.save(head + f'-pmcmc-{delta}', restored)
        else:
            raise ValueError(f'Unknown method {args.method}')
        out = restored
    return out


if __name__ == '__main__':
    main()
