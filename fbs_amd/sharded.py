"""One particle ensemble sharded over the ranks of a process group.

Rank g owns the particle slots [g*n, (g+1)*n), n = N / world.  What is expensive -- propagating
and weighting particles through the model closures (the score network for images) -- runs on the
local slots only.  Per SMC step there are exactly two exchanges:

1. ``all_gather`` of the N unnormalised log-weights (4N bytes).  Every rank then holds the full
   weight vector and computes the normalising constant / ESS, the resampling CDF and the ancestor
   indices of ALL slots redundantly with the single-GPU kernels.  That costs O(N) scalar work per
   rank, removes every further scalar collective, and -- because the canonical summation tree does
   not depend on how the slots are partitioned -- makes the sharded run produce bit-identical
   ancestors, weights and particles to the unsharded one.
2. ``all_to_all`` of ancestor rows: slot m needs row A[m], which may live on another rank.  Since A
   is known everywhere, every rank derives the send/receive plan locally (no count exchange).  The
   conditional killing resampler leaves survivors in place up to one global rotation
   (resamplings.py:85), so the traffic is mostly a contiguous shift between neighbours.

After the last step the forced-move index is drawn from the replicated weights and the selected
particle is broadcast from its owner.  With NCCL (= RCCL on ROCm) the two exchanges are
``ncclAllGather`` / ``ncclAllToAllv`` over xGMI; the same code runs over gloo (CPU tensors) in the
tests.

The local array operations come from a small ``backend`` namespace; the product default is the
libfbsmi-backed GPU backend below (no CPU path), tests may inject another implementation of the same
functions to exercise the exchange logic without a GPU.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


def gpu_backend():
    """The product backend: every local operation is a libfbsmi kernel on the GPU."""
    from . import ops
    from .samplers.csmc.resamplings import killing
    from .samplers.gibbs import force_move
    return SimpleNamespace(split=ops.split, normalise=ops.normalise, exp=lambda x: ops.math_map("exp", x),
                           take_rows=ops.take_rows, set_row=ops.set_row, cond_resampling=killing,
                           force_move=force_move, randint=ops.randint)


class ParticleShards:
    """Slot ownership and the two exchanges."""

    def __init__(self, n_total: int, group=None, dist=None):
        self.dist = dist
        if dist is not None and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.group = group
        if n_total % self.world:
            raise ValueError(f"the ensemble size {n_total} must be a multiple of the world size {self.world}")
        self.N = int(n_total)
        self.n = self.N // self.world
        self.offset = self.rank * self.n

    @property
    def row_slice(self):
        return (self.offset, self.n, self.N)

    def owner(self, idx):
        return idx // self.n

    def owns(self, slot: int) -> bool:
        return self.offset <= int(slot) < self.offset + self.n

    def all_gather_vec(self, x_local: torch.Tensor) -> torch.Tensor:
        """Concatenate every rank's (n, ...) slice into the full (N, ...) array, on every rank."""
        if self.world == 1:
            return x_local
        x_local = x_local.contiguous()
        full = torch.empty((self.N,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        self.dist.all_gather_into_tensor(full, x_local, group=self.group)
        return full

    def exchange_plan(self, A_full: torch.Tensor):
        """Who sends which rows to whom, derived locally from the replicated ancestor array.

        Returns (send_rows, send_splits, recv_splits, recv_positions): the local row numbers this
        rank sends (grouped by destination rank, destination-slot order inside a group), the per-rank
        counts, and for the rows it receives (grouped by source rank) the local slot each one fills."""
        A = A_full.to(torch.int64).reshape(self.world, self.n)        # A[q] = requests of rank q
        own = A // self.n
        mine = own == self.rank                                        # (world, n): rows I must send to q
        send_splits = mine.sum(dim=1).tolist()
        send_rows = (A[mine] - self.offset)                            # row-major: grouped by q, slot order
        my_req_owner = own[self.rank]                                  # (n,): owner of each row I need
        order = torch.argsort(my_req_owner, stable=True)               # grouped by source rank, slot order
        recv_splits = torch.bincount(my_req_owner, minlength=self.world).tolist()
        return send_rows, send_splits, recv_splits, order

    def gather_ancestors(self, us_local: torch.Tensor, A_full: torch.Tensor, take_rows) -> torch.Tensor:
        """us_prev_local[m] = us_full[A_full[offset + m]] without ever materialising us_full."""
        if self.world == 1:
            return take_rows(us_local, A_full)
        send_rows, send_splits, recv_splits, order = self.exchange_plan(A_full)
        rowshape = tuple(us_local.shape[1:])
        send = take_rows(us_local, send_rows.to(torch.int32)).reshape(len(send_rows), -1).contiguous()
        recv = torch.empty((self.n, send.shape[1]), dtype=us_local.dtype, device=us_local.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=recv_splits, input_split_sizes=send_splits,
                                    group=self.group)
        out = torch.empty_like(recv)
        out[order] = recv
        return out.reshape((self.n,) + rowshape)

    def broadcast_row(self, us_local: torch.Tensor, idx: int) -> torch.Tensor:
        """us_full[idx] on every rank."""
        idx = int(idx)
        src = idx // self.n
        row = us_local[idx - self.offset].clone() if src == self.rank else torch.empty_like(us_local[0])
        if self.world > 1:
            self.dist.broadcast(row, src=src if self.group is None else self.dist.get_global_rank(self.group, src),
                                group=self.group)
        return row


def _bs_list(bs_star):
    if isinstance(bs_star, torch.Tensor):
        return [int(b) for b in bs_star.detach().cpu().tolist()]
    return [int(b) for b in np.asarray(bs_star).reshape(-1).tolist()]


def forward_pass(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                 likelihood_logpdf, nrows, shards: ParticleShards, backend=None, **kwargs):
    """csmc.forward_pass (fbs/samplers/csmc/csmc.py:80-164) with conditional killing resampling on a
    sharded ensemble of `nrows` particles.  The closures receive ``row_slice=(offset, count, total)``
    so that per-row randomness can be drawn as a slice of the global draw.

    Returns (log_ws_full (N,) normalised, us_local (n, ...))."""
    be = backend or gpu_backend()
    sh = shards
    assert sh.N == nrows
    nsteps = us_star.shape[0] - 1
    bs = _bs_list(bs_star)
    key_init, key_scan = be.split(key, 2)                                            # csmc.py:150
    us = init_sampler(key_init, nrows, row_slice=sh.row_slice, **kwargs)             # :151 (local rows)
    if sh.owns(bs[0]):
        us = be.set_row(us, bs[0] - sh.offset, us_star[0])                           # :152
    lw = init_likelihood_logpdf(vs[0], us, vs[1], row_slice=sh.row_slice, **kwargs)   # :154 (local)
    log_ws = be.normalise(sh.all_gather_vec(lw), log_space=True)                      # :155 (replicated)
    keys = be.split(key_scan, nsteps)                                                 # :157
    for k in range(nsteps):                                                           # scan_body :132-148
        key_resampling, key_transition = be.split(keys[k], 2)
        v, v_prev, t_prev = vs[k + 1], vs[k], ts[k]
        A = be.cond_resampling(key_resampling, be.exp(log_ws), bs[k], bs[k + 1], True)  # :139 (replicated)
        us_prev = sh.gather_ancestors(us, A, be.take_rows)                            # :140 (exchange)
        us = transition_sampler(us_prev, v_prev, t_prev, key_transition, row_slice=sh.row_slice, **kwargs)  # :142
        if sh.owns(bs[k + 1]):
            us = be.set_row(us, bs[k + 1] - sh.offset, us_star[k + 1])                # :143
        lw = likelihood_logpdf(v, us_prev, v_prev, t_prev, **kwargs)                  # :145 (local)
        log_ws = be.normalise(sh.all_gather_vec(lw), log_space=True)                  # :146 (gather + replicated)
    return log_ws, us


def gibbs_kernel(key, x0, y0, us_star, bs_star, ts, fwd_sampler, sde, unpack, nparticles, transition_sampler,
                 transition_logpdf, likelihood_logpdf, shards: ParticleShards, backend=None, **kwargs):
    """gibbs_kernel (fbs/samplers/gibbs.py:68-168; marg_y=False, explicit_backward=True,
    explicit_final=False) with the particle ensemble sharded over `shards`.  Every rank returns the
    same (x0, us_star, bs_star, acc) as the unsharded kernel."""
    be = backend or gpu_backend()
    sh = shards
    key_fwd, key_csmc, key_bridge = be.split(key, 3)                                  # :126
    path_xy = fwd_sampler(key_fwd, x0, y0, **kwargs)                                  # :127 (replicated, one path)
    path_x, path_y = unpack(path_xy, **kwargs)
    us = torch.flip(path_x, [0])
    vs = torch.flip(path_y, [0])

    def init_sampler(key_, n_, row_slice=None, **kw):                                 # :140-141
        return us[0].unsqueeze(0).expand((sh.n,) + tuple(us.shape[1:])).clone()

    def init_likelihood_logpdf(v0, u0s, v1, row_slice=None, **kw):                    # :143-144
        return torch.full((sh.n,), -math.log(nparticles), dtype=torch.float32, device=us.device)

    bs_np = np.asarray(bs_star.detach().cpu() if isinstance(bs_star, torch.Tensor) else bs_star).reshape(-1)
    k_fwd, k_x0, k_us, k_bs = be.split(key_csmc, 4)                                   # :147
    log_ws_T, us_T = forward_pass(k_fwd, us, bs_np, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                                  likelihood_logpdf, nparticles, sh, be, **kwargs)    # :148
    idx, _ = be.force_move(k_x0, be.exp(log_ws_T), int(bs_np[-1]))                    # :152 (replicated)
    x0_new = sh.broadcast_row(us_T, int(idx))                                         # :154
    us_star_next = torch.flip(unpack(fwd_sampler(k_us, x0_new, y0, **kwargs), **kwargs)[0], [0])  # :155
    bs_star_next = be.randint(k_bs, (us.shape[0],), 0, nparticles, device=us.device)  # :156
    bs_old = torch.as_tensor(bs_np.astype(np.int32), device=bs_star_next.device)
    return us_star_next[-1], us_star_next, bs_star_next, bs_star_next != bs_old       # :167-168
