/* libfbsmi_dist -- the two exchange steps of ONE particle ensemble sharded over the GPUs of a node
 * (SURVEY.md section 8(b) "Multi-GPU: fbsmi_dist_logsumexp, fbsmi_dist_resample_exchange", section 8(e)).
 *
 * A library of its own (fbs_amd/lib/libfbsmi_dist.so, links librccl and libfbsmi) so that single-GPU users never load
 * RCCL.  One process per GPU; rank g owns the contiguous slots [g n, min((g + 1) n, R)) of the R rows, n = ceil(R / world)
 * (the same rule as fbs_amd/sharded.py; only the last shard can be short).
 *
 * What the reference does at this point of the path (one device, one XLA program):
 *   fbs/samplers/csmc/csmc.py:146   log_ws = log_ws - logsumexp(log_ws)            -> fbsmi_dist_logsumexp
 *   fbs/samplers/csmc/csmc.py:139-140   A = cond_resampling(...); us_prev = jnp.take(us, A, axis=0)
 *                                                                                     -> fbsmi_dist_resample_exchange
 *
 * Conventions: as include/fbsmi.h (device pointers owned by the caller, `stream` a hipStream_t, 0 = OK, negative = error with
 * fbsmi_dist_last_error(), no exceptions across the ABI, stream-ordered, no host synchronisation in any per-step entry).
 * Unlike libfbsmi the context OWNS device memory: the gathered log-weights, the normaliser's workspace, and the row
 * windows -- allocated at creation / by fbsmi_dist_window_export, never per step.
 *
 * Bit-identity: every rank normalises the SAME full vector of R log-weights with the single-GPU kernel
 * (fbsmi_normalise_ess), so weights, ancestors and particles of a sharded run equal the unsharded run's bit for bit
 * whatever the number of ranks (the canonical summation tree does not depend on the partition).
 *
 * Two transports for the ancestors' rows:
 *   mode 0  FBSMI_DIST_ALL_GATHER  ncclAllGather of the shards + a local gather.  Fixed shape, world - 1 shards received.
 *   mode 1  FBSMI_DIST_PEER        device-initiated: every rank publishes its rows in a window of its own HBM that the
 *           peers have mapped (hipIpc over xGMI), and the gather kernel LOADS row A[m] straight from its owner.  Only the
 *           rows that are needed move, nobody packs, counts or waits on the host.  Ordering rides on the step's own
 *           collective: rows are published BEFORE fbsmi_dist_logsumexp and read AFTER it, and the window is double
 *           buffered, so a buffer is rewritten only after a later collective that every reader enters after its reads.
 */
#ifndef FBSMI_DIST_H
#define FBSMI_DIST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBSMI_DIST_ABI_VERSION 1
#define FBSMI_DIST_ID_BYTES 128     /* ncclUniqueId */
#define FBSMI_DIST_HANDLE_BYTES 64  /* hipIpcMemHandle_t */
#define FBSMI_DIST_MAX_WORLD 16

#define FBSMI_DIST_ALL_GATHER 0
#define FBSMI_DIST_PEER 1

typedef struct fbsmi_dist_ctx fbsmi_dist_ctx;

int fbsmi_dist_abi_version(void);
const char* fbsmi_dist_last_error(void);

/* Rank 0 fills id (host, FBSMI_DIST_ID_BYTES) and hands it to the other ranks by whatever channel launched them. */
int fbsmi_dist_unique_id(void* id);
/* One context per rank and ensemble of n_total rows, on the calling thread's current device.  id == NULL: no RCCL
 * communicator is created -- only the peer-window entries and fbsmi_dist_shard work (a caller that brings its own
 * collective for the log-weights); otherwise collective over all `world` ranks (ncclCommInitRank). */
int fbsmi_dist_create(const void* id, int rank, int world, int64_t n_total, fbsmi_dist_ctx** out);
int fbsmi_dist_destroy(fbsmi_dist_ctx* ctx);
/* slots per rank n, this rank's first slot, the rows it really owns (any pointer may be NULL) */
int fbsmi_dist_shard(const fbsmi_dist_ctx* ctx, int64_t* n_slots, int64_t* offset, int64_t* count);

/* csmc.py:146 on the sharded ensemble.  lw_local: this rank's `count` unnormalised log-weights.  out_full (n_total, on
 * every rank): lw - logsumexp(lw) (log_space != 0) or exp of it; out_lse / out_ess (nullable, one float each): the step's
 * log-normaliser increment and ESS = 1 / sum w^2.  One ncclAllGather of n floats per rank + fbsmi_normalise_ess. */
int fbsmi_dist_logsumexp(fbsmi_dist_ctx* ctx, const float* lw_local, int log_space, float* out_full, float* out_lse,
                         float* out_ess, void* stream);

/* csmc.py:140 on the sharded ensemble: out_local[m, :] = rows_full[A_full[offset + m], :] for this rank's `count` slots.
 * A_full: the n_total ancestors, replicated (every rank resamples the replicated weights).  Rows of row_floats float32.
 *   mode FBSMI_DIST_ALL_GATHER: rows_local = this rank's rows (count x row_floats); collective.
 *   mode FBSMI_DIST_PEER: rows_local is ignored -- the rows are the ones last given to fbsmi_dist_window_publish on every
 *   rank, and a collective over all ranks (fbsmi_dist_logsumexp, or the caller's own) must lie between those publishes and
 *   this call on every rank's stream.  Not collective itself. */
int fbsmi_dist_resample_exchange(fbsmi_dist_ctx* ctx, const float* rows_local, const int32_t* A_full, int64_t row_floats,
                                 float* out_local, int mode, void* stream);

/* ---- peer windows (mode 1) ---- */
/* Allocate this rank's window: two buffers of n x max_row_floats float32; handle (host, FBSMI_DIST_HANDLE_BYTES) receives
 * the IPC handle to pass to the peers.  Once per context. */
int fbsmi_dist_window_export(fbsmi_dist_ctx* ctx, int64_t max_row_floats, void* handle);
/* handles: world x FBSMI_DIST_HANDLE_BYTES (host), entry g = rank g's handle (the own entry is not opened). */
int fbsmi_dist_window_open(fbsmi_dist_ctx* ctx, const void* handles);
/* Copy this rank's `count` rows into the window's next buffer (the one the following exchange reads). */
int fbsmi_dist_window_publish(fbsmi_dist_ctx* ctx, const float* rows_local, int64_t row_floats, void* stream);
/* One row of the ensemble as last published, to every caller: out[:] = rows_full[idx, :] (gibbs.py:154 after the last
 * step; same ordering rule as mode 1). */
int fbsmi_dist_window_read_row(fbsmi_dist_ctx* ctx, int64_t idx, int64_t row_floats, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
