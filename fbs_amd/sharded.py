"""One particle ensemble sharded over the ranks of a process group.

The R rows of the ensemble (R = nparticles, or nparticles + 1 with explicit_final: gibbs.py:133-134) are cut
into `world` contiguous shards of n = ceil(R / world) slots; rank g owns rows [g n, min((g + 1) n, R)) -- only
the last shard can be short (a split that would leave a rank without rows is refused), its missing slots are padding
that no weight, ancestor or draw ever refers to.
What is expensive -- propagating and weighting particles through the model closures (the score network for
images) -- runs on the local rows only.  Per SMC step there are exactly two exchanges:

1. ``all_gather`` of the R unnormalised log-weights (4 n bytes per rank).  Every rank then holds the full weight
   vector and computes the normalising constant / ESS, the resampling CDF and the ancestor indices of ALL rows
   redundantly with the single-GPU kernels.  That costs O(R) scalar work per rank, removes every further scalar
   collective, and -- because the canonical summation tree does not depend on how the rows are partitioned --
   makes the sharded run produce bit-identical ancestors, weights and particles to the unsharded one.
2. the ancestor rows: slot m needs row A[m], which may live on another rank.  Two forms:
   * ``exchange="all_gather"`` (default): one ``all_gather`` of the particle array; the ancestor gather itself is
     then folded into the kernel that builds the network input (fbsmi_em_concat reads us_full[A[m]]).  No plan, no
     host synchronisation, one collective of fixed shape; it moves n du floats to every rank whatever the
     ancestors are (config 5: 25 MB per rank and step, ~0.1 ms on xGMI against a 170 ms network evaluation).
   * ``exchange="all_to_all"``: the all-to-all of the north-star design.  A is replicated, so every rank derives
     who sends which rows to whom ON THE DEVICE (no count exchange); only the 2 x world split sizes come to the
     host (one small copy per step, overlapped with the gather of the send buffer -- ``all_to_all_single`` takes
     Python split lists, so a variable-size exchange cannot avoid it; a fixed-capacity exchange would have to carry
     a whole shard per pair, i.e. the all_gather's bytes), and only rows whose ancestor lives elsewhere travel: about
     one shard per rank instead of world - 1.  The conditional killing resampler leaves survivors in place up to one
     global rotation (resamplings.py:85), so the traffic is mostly a contiguous shift between two ranks.
   * ``exchange="auto"``: all_to_all for rows of at least 4 KB on more than two ranks (the image configurations),
     all_gather otherwise (the toy's rows are a few bytes: one fixed-shape collective wins).
   * ``exchange="peer"``: device-initiated (libfbsmi_dist, include/fbsmi_dist.h).  Every rank publishes its rows in a
     window of its own HBM that the peers have mapped (hipIpc), and one gather kernel LOADS row A[m] straight from its
     owner over xGMI: only the needed rows move, nothing is packed or counted, the host never waits.  The ordering
     rides on exchange 1: rows are published before the log-weight collective and read after it; the window is
     double buffered.  Runs with several ranks on ONE card too (tests/test_gpu_sharded.py).

``DistContext`` wraps the C-ABI of include/fbsmi_dist.h (SURVEY.md 8(b) ``fbsmi_dist_logsumexp`` /
``fbsmi_dist_resample_exchange``); with ``native_collectives=True`` and the nccl backend, exchange 1 and the all_gather form
of exchange 2 are libfbsmi_dist's own RCCL calls on a communicator created from an id broadcast through the process group.

Every normalisation also yields the step's log-normaliser increment and ESS (fbsmi_normalise_ess) without any
further collective -- the weights are replicated; ``shards.diagnostics`` holds them after a pass.

After the last step the forced-move index is drawn from the replicated weights and the selected particle is
broadcast from its owner.  Per-row noise is drawn as a slice of the global draw (``row_slice = (offset, count,
total)``), so every rank's rows equal the unsharded sweep's.  With NCCL (= RCCL on ROCm) the exchanges are
``ncclAllGather`` / ``ncclAllToAllv`` over xGMI; the same code runs over gloo (CPU tensors) in the tests.

The local array operations come from a small ``backend`` namespace; the product default is the libfbsmi-backed
GPU backend below (no CPU path); tests may inject another implementation of the same functions to exercise the
exchange and slicing logic without a GPU.
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
    from .score import bridge_of

    def fused_step(closures, us_src, A_local, v, v_prev, t_prev, key, pin, row_slice, kwargs):
        """One network evaluation per step when the closures are a ScoreBridge's (fbs_amd/score.py), else None."""
        sb = bridge_of(*closures) if set(kwargs) == {"mask_"} else None
        if sb is None:
            return None
        return sb.fused_step(us_src, A_local, v, v_prev, t_prev, key, kwargs["mask_"], pin=pin, row_slice=row_slice)

    def normalise_diag(lw):
        """(normalised log-weights, logsumexp, ESS) of a step's unnormalised log-weights: fbsmi_normalise_ess."""
        return ops.normalise(lw, log_space=True, return_lse=True, return_ess=True)

    return SimpleNamespace(split=ops.split, normalise=ops.normalise, normalise_diag=normalise_diag,
                           exp=lambda x: ops.math_map("exp", x),
                           take_rows=ops.take_rows, set_row=ops.set_row, cond_resampling=killing,
                           force_move=force_move, randint=ops.randint, normal=ops.normal, fused_step=fused_step)


class DistContext:
    """The C-ABI of include/fbsmi_dist.h for one rank of one ensemble: RCCL collectives and peer windows owned by the
    library.  `dist` is torch.distributed (it only carries the 128-byte communicator id and the 64-byte window handles)."""

    def __init__(self, n_total: int, dist=None, group=None, rccl: bool = False, device=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.dist, self.group = dist, group
        on = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.R = int(n_total)
        ident = None
        if rccl:
            box = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                _lib.dist_call("fbsmi_dist_unique_id", buf)
                box[0] = buf.raw
            if self.world > 1:
                src = 0 if group is None else dist.get_global_rank(group, 0)
                dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.dist_call("fbsmi_dist_create", ident, self.rank, self.world, self.R, C.byref(h))
        self._h = h
        n, off, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.dist_call("fbsmi_dist_shard", self._h, C.byref(n), C.byref(off), C.byref(cnt))
        self.n, self.offset, self.count = n.value, off.value, cnt.value
        self.has_comm = bool(rccl)
        self.window_row = 0

    def close(self):
        """Release the context.  COLLECTIVE when peer windows are open on more than one rank: every rank must call it (a
        barrier makes sure nobody still reads the window this rank is about to free)."""
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            if self.window_row and self.world > 1:
                self.dist.barrier(group=self.group)
            self._lib.dist_lib().fbsmi_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        # Garbage collection is not collective: a context whose window peers may still be reading is left to the process's end
        # rather than freed under them (call close() on every rank to release it earlier).
        try:
            if not (self.window_row and self.world > 1):
                self.close()
        except Exception:
            pass

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def _rows(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.dtype != torch.float32:
            raise TypeError("libfbsmi_dist moves float32 rows on the GPU")
        return x.contiguous()

    def logsumexp(self, lw_local: torch.Tensor, log_space: bool = True):
        """fbsmi_dist_logsumexp -> (normalised (log-)weights of ALL rows, logsumexp, ESS)."""
        lw = self._rows(lw_local).reshape(-1)
        if lw.numel() != self.count:
            raise ValueError(f"{lw.numel()} local log-weights, this rank owns {self.count} rows")
        out = torch.empty(self.R, dtype=torch.float32, device=lw.device)
        diag = torch.empty(2, dtype=torch.float32, device=lw.device)
        self._lib.dist_call("fbsmi_dist_logsumexp", self._h, lw.data_ptr(), 1 if log_space else 0, out.data_ptr(),
                            diag.data_ptr(), diag.data_ptr() + 4, self._stream())
        return out, diag[0], diag[1]

    def exchange(self, A_full: torch.Tensor, rows_local: torch.Tensor = None, mode: str = "all_gather", rowshape=None):
        """fbsmi_dist_resample_exchange -> this rank's (count, ...) ancestor rows."""
        A = A_full.to(torch.int32).contiguous()
        if A.numel() != self.R:
            raise ValueError("the ancestor array must cover the whole ensemble")
        if mode == "peer":
            shape = tuple(rowshape)
            src = 0
        else:
            x = self._rows(rows_local)
            shape, src = tuple(x.shape[1:]), x.data_ptr()
        d = int(np.prod(shape, dtype=np.int64)) if shape else 1
        out = torch.empty((self.count,) + shape, dtype=torch.float32, device=A.device)
        self._lib.dist_call("fbsmi_dist_resample_exchange", self._h, src, A.data_ptr(), d, out.data_ptr(),
                            1 if mode == "peer" else 0, self._stream())
        return out

    def open_windows(self, max_row_floats: int):
        """Allocate this rank's window, swap the IPC handles through the process group, map the peers' windows."""
        C = self._C
        mine = C.create_string_buffer(64)
        with torch.cuda.device(self.device):
            self._lib.dist_call("fbsmi_dist_window_export", self._h, int(max_row_floats), mine)
            if self.world > 1:
                got = [None] * self.world
                self.dist.all_gather_object(got, mine.raw, group=self.group)
                allh = C.create_string_buffer(b"".join(got), 64 * self.world)
            else:
                allh = C.create_string_buffer(mine.raw, 64)
            self._lib.dist_call("fbsmi_dist_window_open", self._h, allh)
        self.window_row = int(max_row_floats)

    def publish(self, rows_local: torch.Tensor):
        x = self._rows(rows_local)
        d = int(np.prod(x.shape[1:], dtype=np.int64)) if x.dim() > 1 else 1
        if x.shape[0] != self.count:
            raise ValueError(f"{x.shape[0]} rows published, this rank owns {self.count}")
        self._lib.dist_call("fbsmi_dist_window_publish", self._h, x.data_ptr(), d, self._stream())

    def read_row(self, idx: int, rowshape) -> torch.Tensor:
        shape = tuple(rowshape)
        d = int(np.prod(shape, dtype=np.int64)) if shape else 1
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        self._lib.dist_call("fbsmi_dist_window_read_row", self._h, int(idx), d, out.data_ptr(), self._stream())
        return out


class ParticleShards:
    """Slot ownership and the exchanges."""

    # rows of at least this many bytes travel by all_to_all under exchange="auto" (only the rows whose ancestor lives elsewhere
    # move: under conditional killing that is ~one shard per rank, against world - 1 shards for the all_gather), smaller rows by
    # all_gather (one collective of fixed shape, no split sizes on the host)
    AUTO_ROW_BYTES = 4096

    def __init__(self, n_total: int, group=None, dist=None, exchange: str = "all_gather", native_collectives: bool = False):
        self.dist = dist
        if dist is not None and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        if exchange not in ("all_gather", "all_to_all", "auto", "peer"):
            raise ValueError(f"unknown exchange {exchange}")
        self.group, self.exchange = group, exchange
        self.R = int(n_total)
        self.n = -(-self.R // self.world)                      # slots per rank
        self.offset = self.rank * self.n
        self.count = max(0, min(self.n, self.R - self.offset))  # rows this rank really owns
        if self.R < self.world or (self.world - 1) * self.n >= self.R:
            # (with shards of ceil(R / world) slots a trailing rank would own no row at all, e.g. 9 rows over 8 ranks)
            raise ValueError(f"an ensemble of {self.R} rows cannot be split over {self.world} ranks in shards of "
                             f"{self.n} slots: rank {self.world - 1} would own no row")
        self.diagnostics = None                                # per-step (lse, ess) of the last forward pass (device tensors)
        self._bytes_moved = 0                                  # payload this rank received in ancestor exchanges
        self._remote_acc = None                                # exchange="peer": the same count kept on the device
        # gloo has no device collectives: GPU tensors are staged through the host (multi-rank rehearsals on one GPU box;
        # the production backend is nccl = RCCL, which takes device pointers)
        self._stage = dist is not None and dist.is_initialized() and dist.get_backend(group) == "gloo"
        # libfbsmi_dist (include/fbsmi_dist.h): the peer windows of exchange="peer", and -- native_collectives, nccl backend
        # only -- its own RCCL communicator for the log-weight all-gather and the all_gather form of the row exchange
        if native_collectives and self.world > 1 and self._stage:
            raise ValueError("native_collectives needs the nccl backend (RCCL takes device pointers; gloo does not)")
        self.native = None
        if exchange == "peer" or native_collectives:
            self.native = DistContext(self.R, dist=dist if self.world > 1 else None, group=group, rccl=native_collectives)
            assert (self.native.n, self.native.offset, self.native.count) == (self.n, self.offset, self.count)
        self.native_collectives = bool(native_collectives)

    def _h(self, x: torch.Tensor) -> torch.Tensor:
        return x.cpu() if (self._stage and x.is_cuda) else x

    # kept for callers of the round-1 interface
    @property
    def N(self):
        return self.R

    @property
    def row_slice(self):
        return (self.offset, self.count, self.R)

    def exchange_for(self, us_local: torch.Tensor) -> str:
        """The ancestor exchange this ensemble uses: the constructor's choice, or by row size under "auto"."""
        if self.exchange != "auto":
            return self.exchange
        row_bytes = int(np.prod(us_local.shape[1:])) * us_local.element_size()
        return "all_to_all" if (self.world > 2 and row_bytes >= self.AUTO_ROW_BYTES) else "all_gather"

    @property
    def bytes_moved(self) -> int:
        return self._bytes_moved + (int(self._remote_acc) if self._remote_acc is not None else 0)

    @bytes_moved.setter
    def bytes_moved(self, v: int):
        self._bytes_moved, self._remote_acc = int(v), None

    def close(self):
        """Release libfbsmi_dist's context (windows, communicator).  Collective on more than one rank with exchange="peer"."""
        if self.native is not None:
            self.native.close()
            self.native = None

    def publish(self, us_local: torch.Tensor):
        """exchange="peer": this rank's rows of the step, into its window (before the step's log-weight collective)."""
        if self.exchange != "peer":
            return
        d = int(np.prod(us_local.shape[1:], dtype=np.int64)) if us_local.dim() > 1 else 1
        if self.native.window_row == 0:
            self.native.open_windows(d)
        elif d > self.native.window_row:
            raise ValueError(f"rows of {d} floats do not fit the window opened for {self.native.window_row}")
        self.native.publish(us_local)

    def peer_ancestors(self, us_local: torch.Tensor, A_full: torch.Tensor) -> torch.Tensor:
        """us_prev_local[m] = us_full[A_full[offset + m]], each row loaded from its owner's window."""
        out = self.native.exchange(A_full, mode="peer", rowshape=tuple(us_local.shape[1:]))
        own = (A_full[self.offset:self.offset + self.count].to(torch.int64) // self.n) == self.rank
        row_bytes = int(np.prod(us_local.shape[1:], dtype=np.int64)) * us_local.element_size()
        moved = (own.numel() - own.sum()) * row_bytes           # a device scalar: counted without a host sync
        self._remote_acc = moved if self._remote_acc is None else self._remote_acc + moved
        return out

    def owns(self, slot: int) -> bool:
        return self.offset <= int(slot) < self.offset + self.count

    def _pad(self, x_local: torch.Tensor) -> torch.Tensor:
        if x_local.shape[0] == self.n:
            return x_local.contiguous()
        pad = torch.zeros((self.n - x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        return torch.cat([x_local, pad], 0)

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """Every rank's (count, ...) rows -> the full (R, ...) array, on every rank."""
        if self.world == 1:
            return x_local
        x = self._h(self._pad(x_local))
        full = torch.empty((self.world * self.n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        self.dist.all_gather_into_tensor(full, x, group=self.group)
        return full[:self.R].to(x_local.device)                 # shards are contiguous: only the tail is padding

    all_gather_vec = all_gather_rows

    def exchange_plan(self, A_full: torch.Tensor):
        """Who sends which rows to whom, derived from the replicated ancestor array on the device.

        Returns (send_rows, splits (2, world) int64 on the device: row 0 = rows this rank sends to each rank, row 1 = rows
        it receives from each rank, order): the local row numbers this rank sends (grouped by destination rank, destination
        slot order inside a group) and, for the rows it receives (grouped by source rank), the local slot each one fills."""
        A = A_full.to(torch.int64).reshape(-1)
        slots = torch.arange(self.R, device=A.device)
        dest = slots // self.n                                   # rank that needs row A[m] for its slot m
        own = A // self.n                                        # rank that holds row A[m]
        mine = own == self.rank
        send_rows = A[mine] - self.offset                        # slot order = grouped by destination rank
        send_splits = torch.bincount(dest[mine], minlength=self.world)
        my_owner = own[self.offset:self.offset + self.count]
        order = torch.argsort(my_owner, stable=True)             # grouped by source rank, slot order inside
        recv_splits = torch.bincount(my_owner, minlength=self.world)
        return send_rows, torch.stack([send_splits, recv_splits]), order

    def gather_ancestors(self, us_local: torch.Tensor, A_full: torch.Tensor, take_rows) -> torch.Tensor:
        """us_prev_local[m] = us_full[A_full[offset + m]] by the all-to-all (us_full is never materialised)."""
        if self.world == 1:
            return take_rows(us_local, A_full)
        send_rows, splits, order = self.exchange_plan(A_full)
        # all_to_all_single takes its split sizes as Python lists: 2 x world integers come to the host, the one host copy of
        # the step.  It is started before the send buffer is gathered and awaited after, so the gather overlaps it.
        if splits.is_cuda:
            sp_host = torch.empty(splits.shape, dtype=splits.dtype, pin_memory=True)
            sp_host.copy_(splits, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            sp_host, ev = splits, None
        rowshape = tuple(us_local.shape[1:])
        rowsize = int(np.prod(rowshape, dtype=np.int64)) if rowshape else 1
        # (explicit row size: a rank none of whose rows is anybody's ancestor -- a collapsed ensemble -- sends zero rows, and
        # reshape(0, -1) is ambiguous)
        send = take_rows(us_local, send_rows.to(torch.int32)).reshape(send_rows.numel(), rowsize).contiguous()
        send = self._h(send)
        if ev is not None:
            ev.synchronize()
        sp = sp_host.tolist()
        recv = torch.empty((self.count, send.shape[1]), dtype=us_local.dtype, device=send.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=sp[1], input_split_sizes=sp[0], group=self.group)
        recv = recv.to(us_local.device)
        self._bytes_moved += (self.count - sp[1][self.rank]) * send.shape[1] * send.element_size()
        out = torch.empty_like(recv)
        out[order] = recv
        return out.reshape((self.count,) + rowshape)

    def broadcast_row(self, us_local: torch.Tensor, idx: int) -> torch.Tensor:
        """us_full[idx] on every rank."""
        idx = int(idx)
        if self.exchange == "peer" and self.world > 1:             # the rows as last published: one load from the owner
            return self.native.read_row(idx, tuple(us_local.shape[1:]))
        src = idx // self.n
        row = us_local[idx - self.offset].clone() if src == self.rank else torch.empty_like(us_local[0])
        if self.world > 1:
            h = self._h(row)
            self.dist.broadcast(h, src=src if self.group is None else self.dist.get_global_rank(self.group, src),
                                group=self.group)
            row = h.to(row.device)
        return row


def _bs_list(bs_star):
    if isinstance(bs_star, torch.Tensor):
        return [int(b) for b in bs_star.detach().cpu().tolist()]
    return [int(b) for b in np.asarray(bs_star).reshape(-1).tolist()]


def forward_pass(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                 likelihood_logpdf, nrows, shards: ParticleShards, backend=None, **kwargs):
    """csmc.forward_pass (fbs/samplers/csmc/csmc.py:80-164) with conditional killing resampling on a sharded
    ensemble of `nrows` rows.  ``init_sampler(key, nrows, row_slice=..., **kw)`` returns the LOCAL rows; the closures
    receive ``row_slice=(offset, count, total)`` so that per-row randomness is drawn as a slice of the global draw.

    Returns (log_ws_full (R,) normalised, us_local (count, ...))."""
    be = backend or gpu_backend()
    sh = shards
    assert sh.R == nrows
    nsteps = us_star.shape[0] - 1
    bs = _bs_list(bs_star)
    pin_of = lambda k: (bs[k] - sh.offset, us_star[k]) if sh.owns(bs[k]) else None
    key_init, key_scan = be.split(key, 2)                                            # csmc.py:150
    us = init_sampler(key_init, nrows, row_slice=sh.row_slice, **kwargs)             # :151 (local rows)
    if sh.owns(bs[0]):
        us = be.set_row(us, bs[0] - sh.offset, us_star[0])                           # :152
    lw = init_likelihood_logpdf(vs[0], us, vs[1], row_slice=sh.row_slice, **kwargs)   # :154 (local)
    diag = []                                      # (logsumexp, ESS) of every normalisation, device scalars (no host sync)

    def normalise(lw_local):
        if sh.native_collectives:                              # fbsmi_dist_logsumexp: ncclAllGather + the same kernel
            out, lse, ess = sh.native.logsumexp(lw_local, log_space=True)
            diag.append(torch.stack([lse.reshape(()), ess.reshape(())]))
            return out
        full = sh.all_gather_rows(lw_local)
        if getattr(be, "normalise_diag", None) is None:
            return be.normalise(full, log_space=True)
        out, lse, ess = be.normalise_diag(full)
        diag.append(torch.stack([lse.reshape(()), ess.reshape(())]))
        return out

    sh.publish(us)                                 # exchange="peer": rows into the window BEFORE the step's collective
    log_ws = normalise(lw)                                                            # :155 (replicated)
    exchange = sh.exchange_for(us)
    keys = be.split(key_scan, nsteps)                                                 # :157
    closures = (transition_sampler, likelihood_logpdf)
    for k in range(nsteps):                                                           # scan_body :132-148
        key_resampling, key_transition = be.split(keys[k], 2)
        v, v_prev, t_prev = vs[k + 1], vs[k], ts[k]
        A = be.cond_resampling(key_resampling, be.exp(log_ws), bs[k], bs[k + 1], True)  # :139 (replicated)
        A_local = A[sh.offset:sh.offset + sh.count]
        if exchange == "peer":                                                        # :140: loaded from the owners
            us_src, A_src = sh.peer_ancestors(us, A), None
        elif exchange == "all_gather" and sh.native_collectives and sh.world > 1:     # :140: fbsmi_dist_resample_exchange
            us_src, A_src = sh.native.exchange(A, us, mode="all_gather"), None
            sh._bytes_moved += (sh.R - sh.count) * int(np.prod(us.shape[1:])) * us.element_size()
        elif exchange == "all_gather" or sh.world == 1:                               # :140: rows by one collective,
            us_src, A_src = sh.all_gather_rows(us), A_local                           # gathered through A by the consumer
            if sh.world > 1:
                sh._bytes_moved += (sh.R - sh.count) * int(np.prod(us.shape[1:])) * us.element_size()
        else:                                                                         # :140: only the rows that move
            us_src, A_src = sh.gather_ancestors(us, A, be.take_rows), None
        fused = be.fused_step(closures, us_src, A_src, v, v_prev, t_prev, key_transition, pin_of(k + 1), sh.row_slice,
                              kwargs) if getattr(be, "fused_step", None) else None
        if fused is not None:                                                         # :140-145 around one evaluation
            us, lw = fused
        else:
            us_prev = us_src if A_src is None else be.take_rows(us_src, A_src)
            us = transition_sampler(us_prev, v_prev, t_prev, key_transition, row_slice=sh.row_slice, **kwargs)  # :142
            if sh.owns(bs[k + 1]):
                us = be.set_row(us, bs[k + 1] - sh.offset, us_star[k + 1])            # :143
            lw = likelihood_logpdf(v, us_prev, v_prev, t_prev, **kwargs)              # :145 (local)
        sh.publish(us)
        log_ws = normalise(lw)                                                        # :146 (gather + replicated)
    # per normalisation (T + 1 of them): the log-normaliser increment logsumexp(lw) (csmc.py:146 / smc.py:145-146 `c`) and
    # the ESS 1 / sum w^2 -- SURVEY.md 8(b) `out_ess`; replicated on every rank, (T + 1, 2) on the device
    sh.diagnostics = torch.stack(diag) if diag else None
    return log_ws, us


def gibbs_kernel(key, x0, y0, us_star, bs_star, ts, fwd_sampler, sde, unpack, nparticles, transition_sampler,
                 transition_logpdf, likelihood_logpdf, shards: ParticleShards, backend=None,
                 explicit_final: bool = False, **kwargs):
    """gibbs_kernel (fbs/samplers/gibbs.py:68-168; marg_y=False, explicit_backward=True) with the particle ensemble
    sharded over `shards` (built for nparticles rows, or nparticles + 1 with explicit_final).  Every rank returns the
    same (x0, us_star, bs_star, acc) as the unsharded kernel."""
    be = backend or gpu_backend()
    sh = shards
    rows = nparticles + 1 if explicit_final else nparticles
    if sh.R != rows:
        raise ValueError(f"the shards describe {sh.R} rows, the sweep has {rows}")
    key_fwd, key_csmc, key_bridge = be.split(key, 3)                                  # :126
    path_xy = fwd_sampler(key_fwd, x0, y0, **kwargs)                                  # :127 (replicated, one path)
    path_x, path_y = unpack(path_xy, **kwargs)
    us = torch.flip(path_x, [0])
    vs = torch.flip(path_y, [0])
    ts0 = ts[0]

    if explicit_final:                                                                # :132-138
        def init_sampler(key_, n_, row_slice=None, **kw):
            return be.normal(key_, (n_,) + tuple(us.shape[1:]), device=us.device, rows=row_slice[:2])

        def init_likelihood_logpdf(v0, u0s, v1, row_slice=None, **kw):
            return likelihood_logpdf(v0, u0s, v1, ts0, **kw)
    else:                                                                             # :139-144
        def init_sampler(key_, n_, row_slice=None, **kw):
            return us[0].unsqueeze(0).expand((sh.count,) + tuple(us.shape[1:])).clone()

        def init_likelihood_logpdf(v0, u0s, v1, row_slice=None, **kw):
            return torch.full((sh.count,), -math.log(nparticles), dtype=torch.float32, device=us.device)

    bs_np = np.asarray(bs_star.detach().cpu() if isinstance(bs_star, torch.Tensor) else bs_star).reshape(-1)
    k_fwd, k_x0, k_us, k_bs = be.split(key_csmc, 4)                                   # :147
    log_ws_T, us_T = forward_pass(k_fwd, us, bs_np, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                                  likelihood_logpdf, rows, sh, be, **kwargs)          # :148
    idx, _ = be.force_move(k_x0, be.exp(log_ws_T), int(bs_np[-1]))                    # :152 (replicated)
    x0_new = sh.broadcast_row(us_T, int(idx))                                         # :154
    us_star_next = torch.flip(unpack(fwd_sampler(k_us, x0_new, y0, **kwargs), **kwargs)[0], [0])  # :155
    bs_star_next = be.randint(k_bs, (us.shape[0],), 0, nparticles, device=us.device)  # :156
    bs_old = torch.as_tensor(bs_np.astype(np.int32), device=bs_star_next.device)
    return us_star_next[-1], us_star_next, bs_star_next, bs_star_next != bs_old       # :167-168
