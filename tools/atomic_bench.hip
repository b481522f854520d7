/*
 * atomic_bench.hip -- how fast can MI355X add fp64 values into a table that
 * lives in the XCD L2s?  Decides whether a "x window in LDS, y panel in L2"
 * blocking could beat the "y tile in LDS, x panel in L2" one (panels.hip),
 * whose floor is the 128-byte line each L2 gather moves to L1.
 *   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/atomic_bench
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(c)                                                              \
    do {                                                                      \
        hipError_t e_ = (c);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,                 \
                    hipGetErrorString(e_));                                   \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* MODE 0: agent-scope atomic add, 1: workgroup-scope atomic add (executes in
 * the XCD's L2), 2: plain gather for reference, 3: plain store */
template <int MODE>
__global__ void __launch_bounds__(256)
    k_scatter(double *tab, unsigned mask, int per_lane, double *sink) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    /* every XCD works in its own copy of the table (workgroups are dealt
     * round-robin to the 8 XCDs) */
    double *t = tab + (size_t)(blockIdx.x & 7) * ((size_t)mask + 1);
    uint64_t h = mix(gid);
    double acc = 0.0;
    for (int i = 0; i < per_lane; i += 4) {
        unsigned a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            a[u] = (unsigned)(h >> 33) & mask;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0)
                __hip_atomic_fetch_add(t + a[u], 1.0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1)
                __hip_atomic_fetch_add(t + a[u], 1.0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2)
                acc += t[a[u]];
            else
                t[a[u]] = 1.0;
        }
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

template <int MODE> static void run(const char *name, double *tab, size_t words,
                                    double *sink) {
    const int lanes = 10 * 1000 * 1000, per_lane = 32;
    const unsigned mask = (unsigned)(words - 1);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        CHECK(hipMemset(tab, 0, words * 8 * sizeof(double)));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_scatter<MODE>, dim3(lanes / 256), dim3(256), 0, 0,
                           tab, mask, per_lane, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best)
            best = ms;
    }
    /* check the sum for the atomic modes */
    double total = -1.0;
    if (MODE < 2) {
        double *h = (double *)malloc(words * 8 * sizeof(double));
        CHECK(hipMemcpy(h, tab, words * 8 * sizeof(double), hipMemcpyDeviceToHost));
        total = 0.0;
        for (size_t i = 0; i < words * 8; ++i)
            total += h[i];
        free(h);
    }
    printf("%-28s table %7.2f MB/XCD: %7.3f ms  %7.1f G/s  sum %.0f (want %.0f)\n",
           name, words * 8 / 1048576.0, best,
           (double)(lanes / 256 * 256) * per_lane / best * 1e-6, total,
           (double)(lanes / 256 * 256) * per_lane);
    fflush(stdout);
}

int main(void) {
    double *tab, *sink;
    const size_t maxw = (size_t)1 << 24; /* 128 MB per XCD copy max */
    CHECK(hipMalloc((void **)&tab, maxw * 8 * sizeof(double)));
    CHECK(hipMalloc((void **)&sink, 8));
    for (size_t words = (size_t)1 << 13; words <= maxw; words <<= 3) {
        run<2>("gather (plain load)", tab, words, sink);
        run<3>("scatter (plain store)", tab, words, sink);
        run<1>("atomic add, workgroup scope", tab, words, sink);
        run<0>("atomic add, agent scope", tab, words, sink);
    }
    return 0;
}
