// persist_probe -- two questions behind a persistent (one launch per sweep) SMC step on MI355X:
//  (1) what does a kernel boundary cost when the kernel takes a ~600-byte by-value argument structure (as the step
//      kernels of fbsmi_lg.hip do) against a 16-byte one;
//  (2) what do the step's two all-to-all edges cost as data-tagged 8-byte granules polled with sc1 loads inside ONE
//      launch: per iteration every workgroup (1024 threads = four 256-slot tiles, one workgroup per CU) publishes
//      2 granules per tile (edge 1: the tile's (max, sumexp)), waits for all tiles of its chain, stores w and u of its
//      slots write-through (sc1), drains, publishes 3 granules per tile (edge 2: tile sums; they double as the flag
//      of the bulk stores), waits for all of them and gathers w[src], u[src] from a rotated source with sc1 loads.
//      Every word read is checked.  Spins are bounded; a timeout raises a flag every poller watches.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/persist_probe tools/persist_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { void* p[70]; int a[8]; };   // 592 bytes, about sizeof(LgDev)

__global__ void __launch_bounds__(256) k_big(Big b, int s) {
    const float* in = (const float*)b.p[s & 1];
    float* out = (float*)b.p[(s & 1) ^ 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[i] = in[(i + b.a[s & 7]) & 65535] + 1.0f;
}
__global__ void __launch_bounds__(256) k_ptr(const Big* __restrict__ bp, int s) {
    const float* in = (const float*)bp->p[s & 1];
    float* out = (float*)bp->p[(s & 1) ^ 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[i] = in[(i + bp->a[s & 7]) & 65535] + 1.0f;
}
__global__ void __launch_bounds__(256) k_small(const float* in, float* out, int sh) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[i] = in[(i + sh) & 65535] + 1.0f;
}

// ---------------------------------------------------------------------------------------------------
struct PP {
    unsigned long long* g1;   // [C][2 * tiles]
    unsigned long long* g2;   // [C][4 * tiles]  (3 used)
    uint32_t* w;              // [2][C][N]
    uint32_t* u;              // [2][C][N]
    int* bail;
    int* bad;
    int* xcc;                 // [grid] XCC_ID of every workgroup
    unsigned long long* clk;  // [grid][2]
    int iters, N, C, mapping, pack;
};

__device__ __forceinline__ unsigned long long pack2(uint32_t v, uint32_t tag) { return ((unsigned long long)tag << 32) | v; }

__device__ __forceinline__ bool poll(const unsigned long long* p, uint32_t tag, uint32_t& v, int* bail) {
    for (int spin = 0;; ++spin) {
        const unsigned long long x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32) == tag) { v = (uint32_t)x; return true; }
        if ((spin & 255) == 255) {
            if (__hip_atomic_load(bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (spin > (1 << 18)) { __hip_atomic_store(bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ void __launch_bounds__(1024) k_persist(PP a) {
    __shared__ uint32_t s1[1024];
    __shared__ int dead;
    const int b = blockIdx.x, tid = threadIdx.x;
    int c, g;   // chain, workgroup of the chain
    const int G = a.N / 1024;   // workgroups per chain
    if (a.mapping == 0) {
        c = b / G; g = b % G;
        if (c >= a.C) return;
    } else {   // blocks b and b + 8 share an XCD: a chain takes G / 32 XCD slots
        const int xpc = G / 32, x = b & 7, r = b >> 3;
        c = x / xpc; g = r * xpc + (x % xpc);
        if (c >= a.C || r >= 32) return;
    }
    if (tid == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[b] = (int)(id & 15);
        a.clk[2 * b] = __builtin_amdgcn_s_memrealtime();
        dead = 0;
    }
    __syncthreads();
    const int N = a.N, tiles = N / 256, T = g * 4 + (tid >> 8), m = g * 1024 + tid;
    unsigned long long* g1 = a.g1 + (size_t)c * 2 * tiles;
    unsigned long long* g2 = a.g2 + (size_t)c * 4 * tiles;
    int nbad = 0;
    for (int it = 1; it <= a.iters; ++it) {
        const int par = it & 1;
        uint32_t* w = a.w + ((size_t)par * a.C + c) * N;
        uint32_t* u = a.u + ((size_t)par * a.C + c) * N;
        // phase A -> edge 1
        if ((tid & 255) < 2) __hip_atomic_store(g1 + 2 * T + (tid & 255), pack2((uint32_t)(T * 2 + (tid & 255)) ^ (uint32_t)it, it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t v = 0;
        bool ok = true;
        if (tid < 2 * tiles) {
            ok = poll(g1 + tid, it, v, a.bail);
            if (ok && v != ((uint32_t)tid ^ (uint32_t)it)) ++nbad;
        }
        if (!ok) dead = 1;
        s1[tid] = v;
        __syncthreads();
        if (dead) break;
        // phase B: bulk stores, write-through
        const uint32_t wv = ((uint32_t)it << 20) ^ (uint32_t)m ^ s1[(tid * 7) & 511 & (2 * tiles - 1)] ^ ((uint32_t)((tid * 7) & 511 & (2 * tiles - 1)) ^ (uint32_t)it);
        const uint32_t uv = ~wv;
        if (a.pack) {
            const uint32_t w1 = __shfl_down(wv, 1), w2 = __shfl_down(wv, 2), w3 = __shfl_down(wv, 3);
            const uint32_t u1 = __shfl_down(uv, 1), u2 = __shfl_down(uv, 2), u3 = __shfl_down(uv, 3);
            if ((tid & 3) == 0) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 wq = {wv, w1, w2, w3}, uq = {uv, u1, u2, u3};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(w + m), "v"(wq) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(u + m), "v"(uq) : "memory");
            }
        } else {
            __hip_atomic_store(w + m, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(u + m, uv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((tid & 255) < 3) __hip_atomic_store(g2 + 4 * T + (tid & 255), pack2((uint32_t)(T * 4 + (tid & 255)) * 3u + (uint32_t)it, it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // edge 2
        v = 0;
        ok = true;
        if (tid < 4 * tiles && (tid & 3) < 3) {
            ok = poll(g2 + tid, it, v, a.bail);
            if (ok && v != (uint32_t)tid * 3u + (uint32_t)it) ++nbad;
        }
        if (!ok) dead = 1;
        __syncthreads();
        if (dead) break;
        // phase C: rotated gather
        const int shift = (it * 9973 + 31) & (N - 1);
        const int src = (m - shift) & (N - 1);
        const uint32_t gw = __hip_atomic_load(w + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t gu = __hip_atomic_load(u + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t want = ((uint32_t)it << 20) ^ (uint32_t)src;
        if (gw != want || gu != ~want) ++nbad;
    }
    if (nbad) atomicAdd(a.bad, nbad);
    if (tid == 0) a.clk[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
}

template <typename F> double time_graph(hipStream_t st, int reps, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int r = 0; r < reps; ++r) enqueue(r);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); for (int it = 0; it < 5; ++it) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.0 / (5.0 * reps);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    // ---- (1) argument size
    {
        float *a, *b2; CK(hipMalloc(&a, 65536 * 4)); CK(hipMalloc(&b2, 65536 * 4));
        CK(hipMemset(a, 0, 65536 * 4)); CK(hipMemset(b2, 0, 65536 * 4));
        Big hb{}; for (int i = 0; i < 70; ++i) hb.p[i] = (i & 1) ? (void*)b2 : (void*)a;
        for (int i = 0; i < 8; ++i) hb.a[i] = i * 17;
        Big* db; CK(hipMalloc(&db, sizeof(Big))); CK(hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice));
        for (int grid : {64, 256}) {
            const double t0 = time_graph(st, 1000, [&](int r) { (r & 1) ? k_small<<<grid, 256, 0, st>>>(b2, a, r & 63) : k_small<<<grid, 256, 0, st>>>(a, b2, r & 63); });
            const double t1 = time_graph(st, 1000, [&](int r) { k_big<<<grid, 256, 0, st>>>(hb, r); });
            const double t2 = time_graph(st, 1000, [&](int r) { k_ptr<<<grid, 256, 0, st>>>(db, r); });
            printf("boundary, grid %3d x 256: 24-byte args %.2f us, %zu-byte by-value struct %.2f us, pointer to the struct %.2f us per kernel\n",
                   grid, t0, sizeof(Big), t1, t2);
        }
    }
    // ---- (2) the persistent exchange
    const int N = 65536, Cmax = 4;
    PP a{};
    CK(hipMalloc(&a.g1, sizeof(unsigned long long) * Cmax * 2 * 256));
    CK(hipMalloc(&a.g2, sizeof(unsigned long long) * Cmax * 4 * 256));
    CK(hipMalloc(&a.w, 4ull * 2 * Cmax * N)); CK(hipMalloc(&a.u, 4ull * 2 * Cmax * N));
    CK(hipMalloc(&a.bail, 4)); CK(hipMalloc(&a.bad, 4)); CK(hipMalloc(&a.xcc, 4 * 256)); CK(hipMalloc(&a.clk, 16 * 256));
    a.N = N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int C : {1, 4})
        for (int mapping : {0, 1})
            for (int pack : {0, 1}) {
                a.C = C; a.mapping = mapping; a.pack = pack; a.iters = 2000;
                CK(hipMemset(a.g1, 0, sizeof(unsigned long long) * Cmax * 2 * 256));
                CK(hipMemset(a.g2, 0, sizeof(unsigned long long) * Cmax * 4 * 256));
                CK(hipMemset(a.bail, 0, 4)); CK(hipMemset(a.bad, 0, 4)); CK(hipMemset(a.xcc, 0xff, 4 * 256));
                const int grid = mapping == 0 ? C * 64 : 256;
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, st));
                k_persist<<<grid, 1024, 0, st>>>(a);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int bail, bad; std::vector<int> xcc(256); std::vector<unsigned long long> clk(512);
                CK(hipMemcpy(&bail, a.bail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(xcc.data(), a.xcc, 4 * 256, hipMemcpyDeviceToHost));
                CK(hipMemcpy(clk.data(), a.clk, 16 * 256, hipMemcpyDeviceToHost));
                // do blocks b and b + 8 share an XCD?
                int same = 0, tot = 0, distinct[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
                for (int b = 0; b < grid; ++b) {
                    if (xcc[b] < 0) continue;
                    if (distinct[b & 7] < 0) distinct[b & 7] = xcc[b];
                    ++tot; same += xcc[b] == distinct[b & 7];
                }
                printf("persist C=%d mapping=%d pack=%d: %.3f us per iteration (2 edges + stores + gather), bail=%d bad=%d; "
                       "b%%8 classes consistent for %d of %d blocks; XCC of classes:", C, mapping, pack, ms * 1e3 / a.iters, bail, bad, same, tot);
                for (int k = 0; k < 8; ++k) printf(" %d", distinct[k]);
                printf("\n");
                if (bail) { printf("timeout: stopping\n"); return 2; }
            }
    return 0;
}
