// What does a grid-wide barrier cost on MI355X when the grid is one 1024-thread workgroup per CU (the step kernel's shape), and do
// write-through stores + device-scope loads carry data between workgroups of DIFFERENT XCDs inside one launch? (the
// persistent form of the step — K step-batches in one launch, the reduce work between two grid barriers — stands or falls with this.)
//   every workgroup, ROUNDS times:  write a 4 KB payload (sc1 write-through stores) -> barrier -> read the payload of workgroup
//   (b + 37) mod n with device-scope loads and check it -> barrier.
// The barrier is a monotonic counter in global memory: thread 0 of a workgroup adds 1 (release) and polls (acquire) with s_sleep
// between polls. EVERY poll is bounded: a workgroup that runs out raises a flag that makes all later barriers fall through, so the
// grid always drains (a grid that is not co-resident — another process on the GPU — ends with "gave up", not with a hang).
// build: hipcc -O3 --offload-arch=gfx950 -o grid_barrier_probe grid_barrier_probe.hip ; run: ./grid_barrier_probe [workgroups]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int THREADS = 1024, ROUNDS = 200, PAYLOAD = 1024;      // floats per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 18;                         // x s_sleep(4) ~ 64 cycles: ~8 ms

struct Ctl { unsigned ctr; unsigned gave_up; unsigned mismatches; unsigned pad; unsigned xctr[16 * 32]; };   // xctr[32 x]: XCC x's counter, 128 B apart

__device__ __forceinline__ void grid_barrier(Ctl *ctl, unsigned target, unsigned &spins_total) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&ctl->gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(&ctl->gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(4);
        }
        spins_total += spins;
    }
    __syncthreads();
}

// relaxed form: no release / acquire (on gfx950 those are an L2 write-back and an L2 invalidate at device scope — the same work a
// kernel boundary does); the data that crosses workgroups must then be written through (sc1 stores, waited for with vmcnt) and read
// with device-scope loads — which is how the step kernel's slabs travel anyway
__device__ __forceinline__ void grid_barrier_relaxed(Ctl *ctl, unsigned target, unsigned &spins_total) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&ctl->gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(&ctl->gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        spins_total += spins;
    }
    __syncthreads();
}

// two levels: a workgroup adds to its XCC's counter; the arrival that completes the XCC adds to the top counter, everybody polls the top
__device__ __forceinline__ void grid_barrier2(Ctl *ctl, unsigned xid, unsigned n_in_xcc, unsigned n_xcc, unsigned round, unsigned &spins_total) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctl->xctr[32 * xid], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_in_xcc * round) __hip_atomic_fetch_add(&ctl->ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_xcc * round) {
            if (__hip_atomic_load(&ctl->gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(&ctl->gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        spins_total += spins;
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier2_relaxed(Ctl *ctl, unsigned xid, unsigned n_in_xcc, unsigned n_xcc, unsigned round, unsigned &spins_total) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctl->xctr[32 * xid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == n_in_xcc * round) __hip_atomic_fetch_add(&ctl->ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_xcc * round) {
            if (__hip_atomic_load(&ctl->gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(&ctl->gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        spins_total += spins;
    }
    __syncthreads();
}

// mode 0: one counter, no payload; 1: one counter + payload; 2: two-level, no payload; 3: two-level + payload; 4, 5: one counter, RELAXED atomics; 6, 7: two-level, RELAXED
// (two-level: assumes the workgroups are spread evenly over the XCCs — checked by the host from the first launch's xcc[] table)
__global__ __launch_bounds__(THREADS) void probe(Ctl *ctl, float *payload, unsigned long long *cycles, unsigned *xcc, int mode, int n_xcc) {
    extern __shared__ float lds[];                                 // 150 KB requested: one workgroup per CU, like the step kernel
    const int b = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
    const bool with_payload = mode & 1, two_level = mode == 2 || mode == 3 || mode >= 6, relaxed = mode >= 4;
    lds[tid] = (float)tid;
    unsigned spins = 0, bad = 0, xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    xid &= 15u;
    if (tid == 0) xcc[b] = xid;
    unsigned target = 0, round = 0;
    auto bar = [&]() {
        if (two_level && relaxed) { ++round; grid_barrier2_relaxed(ctl, xid, (unsigned)(n / n_xcc), (unsigned)n_xcc, round, spins); }
        else if (two_level) { ++round; grid_barrier2(ctl, xid, (unsigned)(n / n_xcc), (unsigned)n_xcc, round, spins); }
        else if (relaxed) { target += (unsigned)n; grid_barrier_relaxed(ctl, target, spins); }
        else { target += (unsigned)n; grid_barrier(ctl, target, spins); }
    };
    bar();                                                         // everybody is running
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < ROUNDS; ++r) {
        if (with_payload) {
            float *mine = payload + (size_t)b * PAYLOAD;
            const float v = (float)(r * 1000 + b);
            asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 2" : : "v"(mine + tid), "v"(v) : "memory");
        }
        bar();
        if (with_payload) {
            const int src = (b + 37) % n;
            const float got = __hip_atomic_load(payload + (size_t)src * PAYLOAD + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += got != (float)(r * 1000 + src);
        }
        bar();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (bad) atomicAdd(&ctl->mismatches, bad);
    if (tid == 0) { cycles[2 * b] = t1 - t0; cycles[2 * b + 1] = spins; }
}

int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    if (n > p.multiProcessorCount) { printf("asked for %d workgroups, the device has %d CUs: clamped (the grid must be co-resident)\n", n, p.multiProcessorCount); n = p.multiProcessorCount; }
    Ctl *ctl; float *payload; unsigned long long *cycles; unsigned *xcc;
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&payload, (size_t)n * PAYLOAD * 4); hipMalloc(&cycles, (size_t)n * 16); hipMalloc(&xcc, (size_t)n * 4);
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n_xcc = 1;
    for (int mode = 0; mode < 8; ++mode) {
        const int with_payload = mode & 1;
        hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(payload, 0, (size_t)n * PAYLOAD * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(n), dim3(THREADS), lds, 0, ctl, payload, cycles, xcc, mode, n_xcc);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        Ctl h; hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> c(2 * n); hipMemcpy(c.data(), cycles, (size_t)n * 16, hipMemcpyDeviceToHost);
        std::vector<unsigned> x(n); hipMemcpy(x.data(), xcc, (size_t)n * 4, hipMemcpyDeviceToHost);
        unsigned long long cmax = 0, smax = 0; double cmean = 0;
        for (int i = 0; i < n; ++i) { cmax = std::max(cmax, c[2 * i]); cmean += c[2 * i]; smax = std::max(smax, c[2 * i + 1]); }
        int per_xcc[16] = {0}; for (int i = 0; i < n; ++i) per_xcc[x[i] & 15]++;
        printf("%s, %d workgroups x %d threads, %d rounds of two barriers%s: %.1f us per launch by events -> %.2f us per barrier; in-kernel %.0f cycles per barrier "
               "(mean over workgroups; slowest %.0f), polls per barrier of the busiest poller %.1f; gave up %u; payload mismatches %u\n",
               mode >= 6 ? "TWO-LEVEL, RELAXED atomics" : mode >= 4 ? "ONE counter, RELAXED atomics (no L2 write-back / invalidate)" : mode >= 2 ? "TWO-LEVEL (per-XCC counter, then one of 8)" : "ONE counter, release / acquire", n, THREADS, ROUNDS,
               with_payload ? " + a 4 KB write-through payload per workgroup read by another" : "", ms * 1e3, ms * 1e3 / (2 * ROUNDS + 1),
               cmean / n / (2 * ROUNDS), (double)cmax / (2 * ROUNDS), (double)smax / (2 * ROUNDS + 1), h.gave_up, h.mismatches);
        if (mode == 1) {
            printf("workgroups per XCC:"); n_xcc = 0; bool even = true;
            for (int i = 0; i < 16; ++i) if (per_xcc[i]) { printf(" %d:%d", i, per_xcc[i]); ++n_xcc; }
            for (int i = 0; i < 16; ++i) if (per_xcc[i] && per_xcc[i] != n / n_xcc) even = false;
            printf("  (reader and writer 37 workgroups apart)\n");
            if (!even || n % n_xcc) { printf("workgroups are not spread evenly over the XCCs: the two-level form is skipped\n"); break; }
        }
    }
    return 0;
}
