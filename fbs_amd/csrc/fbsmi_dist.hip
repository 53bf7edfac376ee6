// fbsmi_dist.hip -- libfbsmi_dist: the exchange steps of one particle ensemble sharded over the GPUs of a node
// (include/fbsmi_dist.h; SURVEY.md 8(b) "Multi-GPU", 8(e)).  RCCL for the two fixed-shape collectives, hipIpc windows and a
// gather kernel that loads rows straight from their owner's HBM over xGMI for the ancestor exchange.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/fbsmi.h"
#include "../../include/fbsmi_dist.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define DIST_HIP_TRY(call)                                                                \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) return fail(-2, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)
#define DIST_NCCL_TRY(call)                                                                 \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) return fail(-3, std::string(#call ": ") + ncclGetErrorString(r_)); \
    } while (0)
#define DIST_FBSMI_TRY(call)                                                     \
    do {                                                                         \
        int s_ = (call);                                                         \
        if (s_ != 0) return fail(s_, std::string(#call ": ") + fbsmi_last_error()); \
    } while (0)

struct PeerTab {
    const float* base[FBSMI_DIST_MAX_WORLD];
};

constexpr int kBlock = 256;

// out[m, :] = window_of(owner(a))[a - owner(a) n, :], a = A[m].  LANES (a power of two <= kBlock) threads share a row and
// walk its chunks; many independent rows per workgroup keep enough loads in flight to cover the xGMI round trip.
template <typename V>
__global__ void __launch_bounds__(kBlock) k_peer_gather(PeerTab tab, const int32_t* __restrict__ A, int64_t count,
                                                        int64_t n_slots, int64_t chunks, int lanes, V* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t m = t / lanes;
    const int l = (int)(t % lanes);
    if (m >= count) return;
    const int64_t a = A[m];
    const int owner = (int)(a / n_slots);
    const V* src = reinterpret_cast<const V*>(tab.base[owner]) + (a - owner * n_slots) * chunks;
    V* dst = out + m * chunks;
    for (int64_t c = l; c < chunks; c += lanes) dst[c] = src[c];
}

__global__ void __launch_bounds__(kBlock) k_copy_row(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = src[i];
}

}  // namespace

struct fbsmi_dist_ctx {
    int rank = 0, world = 1, device = 0;
    int64_t R = 0, n = 0, offset = 0, count = 0;
    ncclComm_t comm = nullptr;
    float* lw_full = nullptr;  // world x n
    void* ws = nullptr;        // fbsmi_workspace_bytes(R)
    float* rows_full = nullptr;  // mode 0 staging, world x n x rows_cap floats
    int64_t rows_cap = 0;
    // peer windows
    float* win = nullptr;  // 2 x n x win_row floats
    int64_t win_row = 0;
    int cur = 0;                // buffer the next exchange reads
    int64_t pub_row_floats = 0;  // row size of the last publish
    bool opened = false;
    float* peer[FBSMI_DIST_MAX_WORLD] = {};
};

extern "C" {

int fbsmi_dist_abi_version(void) { return FBSMI_DIST_ABI_VERSION; }
const char* fbsmi_dist_last_error(void) { return g_err.c_str(); }

int fbsmi_dist_unique_id(void* id) {
    static_assert(sizeof(ncclUniqueId) == FBSMI_DIST_ID_BYTES, "ncclUniqueId size");
    if (!id) return fail(-1, "fbsmi_dist_unique_id: null buffer");
    ncclUniqueId u;
    DIST_NCCL_TRY(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return 0;
}

int fbsmi_dist_create(const void* id, int rank, int world, int64_t n_total, fbsmi_dist_ctx** out) {
    if (!out) return fail(-1, "fbsmi_dist_create: null out");
    *out = nullptr;
    if (world < 1 || world > FBSMI_DIST_MAX_WORLD || rank < 0 || rank >= world)
        return fail(-1, "fbsmi_dist_create: rank / world out of range (world <= " + std::to_string(FBSMI_DIST_MAX_WORLD) + ")");
    if (n_total < world) return fail(-1, "fbsmi_dist_create: fewer rows than ranks");
    const int64_t n = (n_total + world - 1) / world;
    if ((int64_t)(world - 1) * n >= n_total)
        return fail(-1, "fbsmi_dist_create: an ensemble of " + std::to_string(n_total) + " rows cannot be split over " +
                            std::to_string(world) + " ranks in shards of " + std::to_string(n) + " slots: the last rank would own no row");
    if (n_total > INT32_MAX) return fail(-1, "fbsmi_dist_create: ancestors are int32");
    fbsmi_dist_ctx* c = new (std::nothrow) fbsmi_dist_ctx;
    if (!c) return fail(-4, "fbsmi_dist_create: out of host memory");
    c->rank = rank;
    c->world = world;
    c->R = n_total;
    c->n = n;
    c->offset = rank * n;
    c->count = (c->offset + n <= n_total) ? n : n_total - c->offset;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipMalloc(&c->lw_full, sizeof(float) * world * n);
    if (e == hipSuccess) e = hipMalloc(&c->ws, fbsmi_workspace_bytes(n_total));
    if (e != hipSuccess) {
        fbsmi_dist_destroy(c);
        return fail(-2, std::string("fbsmi_dist_create: ") + hipGetErrorString(e));
    }
    if (id) {
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) {
            c->comm = nullptr;
            fbsmi_dist_destroy(c);
            return fail(-3, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        }
    }
    *out = c;
    return 0;
}

int fbsmi_dist_destroy(fbsmi_dist_ctx* c) {
    if (!c) return 0;
    if (c->opened)
        for (int g = 0; g < c->world; ++g)
            if (g != c->rank && c->peer[g]) (void)hipIpcCloseMemHandle(c->peer[g]);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->win) (void)hipFree(c->win);
    if (c->rows_full) (void)hipFree(c->rows_full);
    if (c->ws) (void)hipFree(c->ws);
    if (c->lw_full) (void)hipFree(c->lw_full);
    delete c;
    return 0;
}

int fbsmi_dist_shard(const fbsmi_dist_ctx* c, int64_t* n_slots, int64_t* offset, int64_t* count) {
    if (!c) return fail(-1, "fbsmi_dist_shard: null context");
    if (n_slots) *n_slots = c->n;
    if (offset) *offset = c->offset;
    if (count) *count = c->count;
    return 0;
}

int fbsmi_dist_logsumexp(fbsmi_dist_ctx* c, const float* lw_local, int log_space, float* out_full, float* out_lse,
                         float* out_ess, void* stream) {
    if (!c || !lw_local || !out_full) return fail(-1, "fbsmi_dist_logsumexp: null argument");
    if (c->world > 1 && !c->comm) return fail(-1, "fbsmi_dist_logsumexp: the context was created without a communicator");
    hipStream_t st = (hipStream_t)stream;
    // in-place all-gather: rank g's n slots at lw_full + g n (the last shard's padding is never read: the shards are contiguous,
    // so lw_full[0 .. R) is the ensemble)
    DIST_HIP_TRY(hipMemcpyAsync(c->lw_full + c->offset, lw_local, sizeof(float) * c->count, hipMemcpyDeviceToDevice, st));
    if (c->world > 1) DIST_NCCL_TRY(ncclAllGather(c->lw_full + c->offset, c->lw_full, (size_t)c->n, ncclFloat, c->comm, st));
    DIST_FBSMI_TRY(fbsmi_normalise_ess(c->lw_full, c->R, log_space, out_full, out_lse, out_ess, c->ws, st));
    return 0;
}

static int peer_gather(fbsmi_dist_ctx* c, const int32_t* A_local, int64_t row_floats, float* out_local, hipStream_t st) {
    PeerTab tab;
    for (int g = 0; g < FBSMI_DIST_MAX_WORLD; ++g) tab.base[g] = nullptr;
    const int64_t buf = (int64_t)c->cur * c->n * c->win_row;
    for (int g = 0; g < c->world; ++g) tab.base[g] = (g == c->rank ? c->win : c->peer[g]) + buf;
    const bool vec = row_floats % 4 == 0;
    const int64_t chunks = vec ? row_floats / 4 : row_floats;
    int lanes = 1;
    while (lanes < kBlock && lanes < chunks) lanes *= 2;
    const int64_t threads = c->count * lanes;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    if (vec)
        k_peer_gather<float4><<<grid, kBlock, 0, st>>>(tab, A_local, c->count, c->n, chunks, lanes, reinterpret_cast<float4*>(out_local));
    else
        k_peer_gather<float><<<grid, kBlock, 0, st>>>(tab, A_local, c->count, c->n, chunks, lanes, out_local);
    DIST_HIP_TRY(hipGetLastError());
    return 0;
}

int fbsmi_dist_resample_exchange(fbsmi_dist_ctx* c, const float* rows_local, const int32_t* A_full, int64_t row_floats,
                                 float* out_local, int mode, void* stream) {
    if (!c || !A_full || !out_local || row_floats < 1) return fail(-1, "fbsmi_dist_resample_exchange: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int32_t* A_local = A_full + c->offset;
    if (mode == FBSMI_DIST_PEER) {
        if (!c->win || (c->world > 1 && !c->opened)) return fail(-1, "fbsmi_dist_resample_exchange: no peer windows (window_export / window_open)");
        if (row_floats != c->pub_row_floats)
            return fail(-1, "fbsmi_dist_resample_exchange: rows of " + std::to_string(row_floats) + " floats, the last publish had " +
                                std::to_string(c->pub_row_floats));
        return peer_gather(c, A_local, row_floats, out_local, st);
    }
    if (mode != FBSMI_DIST_ALL_GATHER) return fail(-1, "fbsmi_dist_resample_exchange: unknown mode");
    if (!rows_local) return fail(-1, "fbsmi_dist_resample_exchange: null rows");
    if (c->world == 1) {
        DIST_FBSMI_TRY(fbsmi_gather_rows(rows_local, A_local, c->count, row_floats, out_local, st));
        return 0;
    }
    if (!c->comm) return fail(-1, "fbsmi_dist_resample_exchange: the context was created without a communicator");
    if (row_floats > c->rows_cap) {  // first call at this row size: the one allocation (synchronises the device)
        if (c->rows_full) DIST_HIP_TRY(hipFree(c->rows_full));
        c->rows_full = nullptr;
        c->rows_cap = 0;
        DIST_HIP_TRY(hipMalloc(&c->rows_full, sizeof(float) * c->world * c->n * row_floats));
        c->rows_cap = row_floats;
    }
    float* mine = c->rows_full + c->offset * row_floats;
    DIST_HIP_TRY(hipMemcpyAsync(mine, rows_local, sizeof(float) * c->count * row_floats, hipMemcpyDeviceToDevice, st));
    DIST_NCCL_TRY(ncclAllGather(mine, c->rows_full, (size_t)(c->n * row_floats), ncclFloat, c->comm, st));
    DIST_FBSMI_TRY(fbsmi_gather_rows(c->rows_full, A_local, c->count, row_floats, out_local, st));
    return 0;
}

int fbsmi_dist_window_export(fbsmi_dist_ctx* c, int64_t max_row_floats, void* handle) {
    if (!c || !handle || max_row_floats < 1) return fail(-1, "fbsmi_dist_window_export: bad argument");
    if (c->win) return fail(-1, "fbsmi_dist_window_export: the context already has a window");
    DIST_HIP_TRY(hipMalloc(&c->win, sizeof(float) * 2 * c->n * max_row_floats));
    c->win_row = max_row_floats;
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == FBSMI_DIST_HANDLE_BYTES, "hipIpcMemHandle_t size");
    DIST_HIP_TRY(hipIpcGetMemHandle(&h, c->win));
    std::memcpy(handle, &h, sizeof h);
    return 0;
}

int fbsmi_dist_window_open(fbsmi_dist_ctx* c, const void* handles) {
    if (!c || !handles) return fail(-1, "fbsmi_dist_window_open: null argument");
    if (!c->win) return fail(-1, "fbsmi_dist_window_open: export this rank's window first");
    if (c->opened) return fail(-1, "fbsmi_dist_window_open: already open");
    const char* hs = static_cast<const char*>(handles);
    for (int g = 0; g < c->world; ++g) {
        if (g == c->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, hs + (size_t)g * FBSMI_DIST_HANDLE_BYTES, sizeof h);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < g; ++q)
                if (q != c->rank && c->peer[q]) {
                    (void)hipIpcCloseMemHandle(c->peer[q]);
                    c->peer[q] = nullptr;
                }
            return fail(-2, "hipIpcOpenMemHandle(rank " + std::to_string(g) + "): " + hipGetErrorString(e));
        }
        c->peer[g] = static_cast<float*>(p);
    }
    c->opened = true;
    return 0;
}

int fbsmi_dist_window_publish(fbsmi_dist_ctx* c, const float* rows_local, int64_t row_floats, void* stream) {
    if (!c || !rows_local) return fail(-1, "fbsmi_dist_window_publish: null argument");
    if (!c->win) return fail(-1, "fbsmi_dist_window_publish: no window");
    if (row_floats < 1 || row_floats > c->win_row)
        return fail(-1, "fbsmi_dist_window_publish: rows of " + std::to_string(row_floats) + " floats, the window holds " + std::to_string(c->win_row));
    c->cur ^= 1;
    c->pub_row_floats = row_floats;
    float* dst = c->win + (int64_t)c->cur * c->n * c->win_row;
    DIST_HIP_TRY(hipMemcpyAsync(dst, rows_local, sizeof(float) * c->count * row_floats, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int fbsmi_dist_window_read_row(fbsmi_dist_ctx* c, int64_t idx, int64_t row_floats, float* out, void* stream) {
    if (!c || !out) return fail(-1, "fbsmi_dist_window_read_row: null argument");
    if (!c->win || (c->world > 1 && !c->opened)) return fail(-1, "fbsmi_dist_window_read_row: no peer windows");
    if (idx < 0 || idx >= c->R || row_floats != c->pub_row_floats) return fail(-1, "fbsmi_dist_window_read_row: row index / size out of range");
    const int owner = (int)(idx / c->n);
    const float* base = (owner == c->rank ? c->win : c->peer[owner]) + (int64_t)c->cur * c->n * c->win_row;
    const float* src = base + (idx - owner * c->n) * row_floats;
    const unsigned grid = (unsigned)((row_floats + kBlock - 1) / kBlock);
    k_copy_row<<<grid < 64 ? grid : 64, kBlock, 0, (hipStream_t)stream>>>(src, out, row_floats);
    DIST_HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
