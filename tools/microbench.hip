/*
 * microbench.hip -- MI355X memory-path probes that size the SpMV kernels.
 *   stream   bytes/s of coalesced reads at 4 / 8 / 16 B per lane, plain and
 *            non-temporal
 *   spmvmix  the HLL stream without gathers: int32 + fp64 per slot
 *   gather   random 8-byte gathers from a table of T bytes (L1 / L2 /
 *            Infinity Cache / HBM resident), indices computed in-register
 * Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <typename T, bool NT>
__global__ void k_stream(const T *__restrict__ p, size_t n, int *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    int acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        T a, b, c, d;
        if (NT) {
            a = __builtin_nontemporal_load(p + i);
            b = __builtin_nontemporal_load(p + i + stride);
            c = __builtin_nontemporal_load(p + i + 2 * stride);
            d = __builtin_nontemporal_load(p + i + 3 * stride);
        } else {
            a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
        }
        acc += ((const int *)&a)[0] + ((const int *)&b)[0] +
               ((const int *)&c)[0] + ((const int *)&d)[0];
    }
    if (acc == 0x7fffffff)
        *sink = acc;
}

/* HLL-like stream: lane reads one int and one double per step */
template <bool NT>
__global__ void k_spmvmix(const int *__restrict__ ja, const double *__restrict__ as,
                          size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        int c0, c1, c2, c3; double a0, a1, a2, a3;
        if (NT) {
            c0 = __builtin_nontemporal_load(ja + i); c1 = __builtin_nontemporal_load(ja + i + stride);
            c2 = __builtin_nontemporal_load(ja + i + 2 * stride); c3 = __builtin_nontemporal_load(ja + i + 3 * stride);
            a0 = __builtin_nontemporal_load(as + i); a1 = __builtin_nontemporal_load(as + i + stride);
            a2 = __builtin_nontemporal_load(as + i + 2 * stride); a3 = __builtin_nontemporal_load(as + i + 3 * stride);
        } else {
            c0 = ja[i]; c1 = ja[i + stride]; c2 = ja[i + 2 * stride]; c3 = ja[i + 3 * stride];
            a0 = as[i]; a1 = as[i + stride]; a2 = as[i + 2 * stride]; a3 = as[i + 3 * stride];
        }
        acc += a0 * c0 + a1 * c1 + a2 * c2 + a3 * c3;
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

/* HLL-shaped stream: a wave owns two consecutive 32x32 blocks (col-major):
 * per step a half-wave reads 128 B of JA and 256 B of AS, 32 steps, U loads
 * in flight.  PAIRS consecutive block pairs per wave. */
template <int U, int PAIRS>
__global__ void k_hllshape(const int *__restrict__ ja, const double *__restrict__ as,
                           size_t nblocks, double *sink) {
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    int lane = threadIdx.x & 63, half = lane >> 5, i = lane & 31;
    double acc = 0;
    for (int pr = 0; pr < PAIRS; ++pr) {
        size_t b = (wave * PAIRS + pr) * 2 + half;
        if (b >= nblocks) break;
        const int *cj = ja + b * 1024 + i;
        const double *ca = as + b * 1024 + i;
        for (int j = 0; j < 32; j += U) {
            int c[U]; double a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = __builtin_nontemporal_load(cj + (j + u) * 32);
                a[u] = __builtin_nontemporal_load(ca + (j + u) * 32);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u] * c[u];
        }
    }
    if (acc == 1.2345e300) *sink = acc;
}

/* the same plus, one at a time, what the real kernel adds: F bit 0 = block
 * offsets from an off[] array, bit 1 = y store, bit 2 = dependent x gather
 * (x[c & 1023], L1 resident), bit 3 = two-stage software pipeline */
template <int U, int F>
__global__ void k_hllshape2(const int64_t *__restrict__ off, const int *__restrict__ ja,
                            const double *__restrict__ as, const double *__restrict__ x,
                            double *__restrict__ y, size_t nblocks, double *sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = t / 32;
    int i = (int)(t % 32);
    if (b >= nblocks) return;
    size_t o = (F & 1) ? (size_t)off[b] : b * 1024;
    const int *cj = ja + o + i;
    const double *ca = as + o + i;
    double acc = 0;
    if (F & 8) {
        int c[U]; double a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { c[u] = __builtin_nontemporal_load(cj + u * 32); a[u] = __builtin_nontemporal_load(ca + u * 32); }
        for (int j = 0; j < 32; j += U) {
            double xv[U], av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { xv[u] = (F & 4) ? x[c[u] & 1023] : (double)c[u]; av[u] = a[u]; }
            if (j + U < 32) {
#pragma unroll
                for (int u = 0; u < U; ++u) { c[u] = __builtin_nontemporal_load(cj + (j + U + u) * 32); a[u] = __builtin_nontemporal_load(ca + (j + U + u) * 32); }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += av[u] * xv[u];
        }
    } else {
        for (int j = 0; j < 32; j += U) {
            int c[U]; double a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { c[u] = __builtin_nontemporal_load(cj + (j + u) * 32); a[u] = __builtin_nontemporal_load(ca + (j + u) * 32); }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u] * ((F & 4) ? x[c[u] & 1023] : (double)c[u]);
        }
    }
    if (F & 2) y[b * 32 + i] = acc;
    else if (acc == 1.2345e300) *sink = acc;
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* each lane does `per` random 8-byte gathers inside a window of `win`
 * elements that slides with the lane id (win == n: anywhere) */
__global__ void k_gather(const double *__restrict__ t, size_t n, size_t win,
                         int per, double *sink) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)gridDim.x * blockDim.x;
    size_t base = win >= n ? 0 : (size_t)((double)gid / total * (double)(n - win));
    double acc = 0;
    uint64_t h = mix(gid);
    for (int j = 0; j < per; j += 8) {
        size_t i0 = base + (mix(h + j) % win), i1 = base + (mix(h + j + 1) % win);
        size_t i2 = base + (mix(h + j + 2) % win), i3 = base + (mix(h + j + 3) % win);
        size_t i4 = base + (mix(h + j + 4) % win), i5 = base + (mix(h + j + 5) % win);
        size_t i6 = base + (mix(h + j + 6) % win), i7 = base + (mix(h + j + 7) % win);
        acc += t[i0] + t[i1] + t[i2] + t[i3] + t[i4] + t[i5] + t[i6] + t[i7];
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

/* gathers through a buffer descriptor with explicit cache-policy bits
 * (gfx940+: aux bit0 = sc0, bit1 = nt, bit4 = sc1) */
template <int AUX>
__global__ void k_gather_buf(const double *__restrict__ t, size_t n, int per,
                             double *sink) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)t, 0, (int)(n * 8 > 0xffffffffu ? 0xffffffffu : n * 8), 0x00020000);
    double acc = 0;
    uint64_t h = mix(gid);
    for (int j = 0; j < per; j += 4) {
        unsigned o0 = (unsigned)(mix(h + j) % n) * 8u, o1 = (unsigned)(mix(h + j + 1) % n) * 8u;
        unsigned o2 = (unsigned)(mix(h + j + 2) % n) * 8u, o3 = (unsigned)(mix(h + j + 3) % n) * 8u;
        v2i a = __builtin_amdgcn_raw_buffer_load_b64(rs, o0, 0, AUX);
        v2i b = __builtin_amdgcn_raw_buffer_load_b64(rs, o1, 0, AUX);
        v2i c = __builtin_amdgcn_raw_buffer_load_b64(rs, o2, 0, AUX);
        v2i d = __builtin_amdgcn_raw_buffer_load_b64(rs, o3, 0, AUX);
        acc += __builtin_bit_cast(double, a) + __builtin_bit_cast(double, b) +
               __builtin_bit_cast(double, c) + __builtin_bit_cast(double, d);
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

/* divergent-gather issue rate with cheap index math: lane stride `ls`
 * doubles inside a table of `mask`+1 doubles (L1/L2 resident) */
__global__ void k_gather_cheap(const double *__restrict__ t, unsigned mask,
                               unsigned ls, int per, double *sink) {
    unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b = gid * ls;
    double acc = 0;
    for (int j = 0; j < per; j += 8) {
        unsigned o = b + j * 1031u;
        acc += t[(o) & mask] + t[(o + 1031u) & mask] + t[(o + 2062u) & mask] +
               t[(o + 3093u) & mask] + t[(o + 4124u) & mask] +
               t[(o + 5155u) & mask] + t[(o + 6186u) & mask] +
               t[(o + 7217u) & mask];
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

/* the same out of LDS (16 Ki doubles = 128 KiB), random-ish bank pattern */
__global__ void k_gather_lds(const double *__restrict__ t, unsigned ls, int per,
                             double *sink) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x)
        lds[i] = t[i];
    __syncthreads();
    unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned b = gid * ls;
    double acc = 0;
    for (int j = 0; j < per; j += 8) {
        unsigned o = b + j * 1031u;
        acc += lds[(o) & 16383] + lds[(o + 1031u) & 16383] +
               lds[(o + 2062u) & 16383] + lds[(o + 3093u) & 16383] +
               lds[(o + 4124u) & 16383] + lds[(o + 5155u) & 16383] +
               lds[(o + 6186u) & 16383] + lds[(o + 7217u) & 16383];
    }
    if (acc == 1.2345e300)
        *sink = acc;
}

/* the same index arithmetic without the loads: cost of the hash itself */
__global__ void k_gather_null(size_t n, size_t win, int per, double *sink) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)gridDim.x * blockDim.x;
    size_t base = win >= n ? 0 : (size_t)((double)gid / total * (double)(n - win));
    size_t acc = 0;
    uint64_t h = mix(gid);
    for (int j = 0; j < per; ++j)
        acc += base + (mix(h + j) % win);
    if (acc == 0x7fffffffffffull)
        *sink = (double)acc;
}

template <typename F> static double time_ms(F f, int reps = 7) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<double> v;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms);
    }
    std::sort(v.begin(), v.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[v.size() / 2];
}

/* read-only sweep: evicts a small working set from L2 / Infinity Cache */
__global__ void k_sweep_ro(const double *buf, size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride)
        acc += buf[i];
    if (acc == 1.2345e300)
        *sink = acc;
}

/*
 * `microbench small`: what ONE launch can stream when the whole job is the
 * size of BASELINE config 2 (16M slots: 64 MB of int32 + 128 MB of fp64 = 192
 * MB, read once, non-temporal), each launch after a 1 GiB read-only sweep --
 * the regime of bench.py --config 2.  The ceiling a 1M x 16 SpMV launch can be
 * compared with: ramp-up and tail of a ~35 us launch included.
 */
static int small_stream(void) {
    const size_t n = 16000000, fl = (size_t)1 << 27; /* 1 GiB of doubles */
    int *ja; double *as, *scratch, *sink;
    CK(hipMalloc((void **)&ja, n * 4)); CK(hipMalloc((void **)&as, n * 8));
    CK(hipMalloc((void **)&scratch, fl * 8)); CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(ja, 1, n * 4)); CK(hipMemset(as, 0, n * 8));
    CK(hipMemset(scratch, 0, fl * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("== small stream: 16M slots (192 MB) once per launch, nt loads, 1 GiB read-only sweep before each\n");
    for (int flush = 1; flush >= 0; --flush)
        for (int g : {1024, 2048, 4096, 8192, 16384}) {
            std::vector<float> v;
            for (int it = 0; it < 33; ++it) {
                if (flush)
                    k_sweep_ro<<<2048, 256>>>(scratch, fl, sink);
                CK(hipEventRecord(a));
                k_spmvmix<true><<<g, 256>>>(ja, as, n, sink);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (it >= 3) v.push_back(ms);
            }
            std::sort(v.begin(), v.end());
            printf("%s grid %6d x 256: median %7.2f us  min %7.2f us  -> %6.2f TB/s (median)\n",
                   flush ? "flushed  " : "unflushed", g, v[v.size() / 2] * 1e3, v[0] * 1e3,
                   12.0 * n / v[v.size() / 2] * 1e-9);
        }
    return 0;
}

int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IOLBF, 0);
    if (argc > 1 && !strcmp(argv[1], "small"))
        return small_stream();
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s %s CUs=%d clock=%d MHz L2=%d KB\n", p.name, p.gcnArchName,
           p.multiProcessorCount, p.clockRate / 1000, p.l2CacheSize / 1024);
    const size_t bytes = (size_t)4 << 30;
    void *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    int *isink; CK(hipMalloc((void **)&isink, 64));
    double *dsink = (double *)isink;
    const int grids[] = {2048, 8192, 32768};
    printf("== stream (4 GiB) GB/s\n");
    for (int g : grids) {
        double t;
        t = time_ms([&] { k_stream<int, false><<<g, 256>>>((int *)buf, bytes / 4, isink); });
        printf("grid %6d dword    plain %8.1f", g, bytes / t * 1e-6);
        t = time_ms([&] { k_stream<int, true><<<g, 256>>>((int *)buf, bytes / 4, isink); });
        printf("  nt %8.1f\n", bytes / t * 1e-6);
        t = time_ms([&] { k_stream<v2i, false><<<g, 256>>>((v2i *)buf, bytes / 8, isink); });
        printf("grid %6d dwordx2  plain %8.1f", g, bytes / t * 1e-6);
        t = time_ms([&] { k_stream<v2i, true><<<g, 256>>>((v2i *)buf, bytes / 8, isink); });
        printf("  nt %8.1f\n", bytes / t * 1e-6);
        t = time_ms([&] { k_stream<v4i, false><<<g, 256>>>((v4i *)buf, bytes / 16, isink); });
        printf("grid %6d dwordx4  plain %8.1f", g, bytes / t * 1e-6);
        t = time_ms([&] { k_stream<v4i, true><<<g, 256>>>((v4i *)buf, bytes / 16, isink); });
        printf("  nt %8.1f\n", bytes / t * 1e-6);
    }
    printf("== spmvmix: int32+fp64 per slot, 300M slots (3.6 GB) GB/s\n");
    {
        size_t n = 300000000; /* 1.2 GB of JA at offset 0, 2.4 GB of AS at 1.5 GiB: inside the 4 GiB buffer */
        int *ja = (int *)buf; double *as = (double *)((char *)buf + ((size_t)3 << 29));
        for (int g : grids) {
            double t = time_ms([&] { k_spmvmix<false><<<g, 256>>>(ja, as, n, dsink); });
            printf("grid %6d plain %8.1f", g, 12.0 * n / t * 1e-6);
            t = time_ms([&] { k_spmvmix<true><<<g, 256>>>(ja, as, n, dsink); });
            printf("  nt %8.1f\n", 12.0 * n / t * 1e-6);
        }
    }
    printf("== hll-shaped stream (no gather), 300M slots, nt loads GB/s\n");
    {
        size_t nblocks = 300000000 / 1024;
        int *ja = (int *)buf; double *as = (double *)((char *)buf + ((size_t)3 << 29));
        double bytes_h = 12.0 * nblocks * 1024;
#define HS(U, PAIRS, THREADS) { size_t waves = (nblocks / 2 + PAIRS - 1) / PAIRS; \
        unsigned g = (unsigned)((waves * 64 + THREADS - 1) / THREADS); \
        double t = time_ms([&] { k_hllshape<U, PAIRS><<<g, THREADS>>>(ja, as, nblocks, dsink); }); \
        printf("U=%2d pairs/wave=%d threads=%4d : %8.1f GB/s\n", U, PAIRS, THREADS, bytes_h / t * 1e-6); }
        HS(4, 1, 256) HS(8, 1, 256) HS(8, 1, 512) HS(16, 1, 256) HS(32, 1, 256)
        HS(8, 2, 256) HS(8, 4, 256) HS(8, 8, 256) HS(32, 4, 256)
    }
    printf("== hll-shaped stream + one feature at a time (U=8, 256 threads)\n");
    {
        size_t nblocks = 300000000 / 1024;
        int *ja = (int *)buf; double *as = (double *)((char *)buf + ((size_t)3 << 29));
        int64_t *off; CK(hipMalloc((void **)&off, (nblocks + 1) * 8));
        { std::vector<int64_t> h(nblocks + 1); for (size_t k = 0; k <= nblocks; ++k) h[k] = (int64_t)k * 1024;
          CK(hipMemcpy(off, h.data(), (nblocks + 1) * 8, hipMemcpyHostToDevice)); }
        double *xx, *yy; CK(hipMalloc((void **)&xx, 8 * 1024 * 8)); CK(hipMemset(xx, 0, 8 * 1024 * 8));
        CK(hipMalloc((void **)&yy, nblocks * 32 * 8));
        CK(hipMemset(buf, 0, (size_t)3 << 29)); /* ja = 0: gathers stay in range */
        double bytes_h = 12.0 * nblocks * 1024;
        unsigned g = (unsigned)((nblocks * 32 + 255) / 256);
#define HF(F) { double t = time_ms([&] { k_hllshape2<8, F><<<g, 256>>>(off, ja, as, xx, yy, nblocks, dsink); }); \
        printf("features %2d (off=%d ystore=%d gather=%d pipelined=%d): %8.1f GB/s\n", F, F & 1, (F >> 1) & 1, (F >> 2) & 1, (F >> 3) & 1, bytes_h / t * 1e-6); }
        HF(0) HF(1) HF(2) HF(3) HF(4) HF(7) HF(8) HF(12) HF(15)
        CK(hipMemset(buf, 1, (size_t)3 << 29));
    }
    printf("== gather: 32 random fp64 gathers per lane, 10M lanes (320M gathers)\n");
    printf("   needed for config 3 at 60%% roofline: 384 Ggather/s\n");
    {
        const int per = 32; const int lanes = 10000000; const int g = lanes / 256;
        double tn = time_ms([&] { k_gather_null<<<g, 256>>>((size_t)1 << 20, (size_t)1 << 20, per, dsink); });
        printf("index arithmetic alone: %.3f ms\n", tn);
        const size_t tabs[] = {(size_t)16 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)8 << 20,
                               (size_t)32 << 20, (size_t)80 << 20, (size_t)200 << 20, (size_t)640 << 20,
                               (size_t)2 << 30};
        for (size_t tb : tabs) {
            size_t n = tb / 8;
            double t = time_ms([&] { k_gather<<<g, 256>>>((double *)buf, n, n, per, dsink); });
            printf("table %8.2f MB uniform          : %7.3f ms  %7.1f Ggather/s  (useful %6.1f GB/s)\n",
                   tb / 1048576.0, t, (double)lanes * per / t * 1e-6, 8.0 * lanes * per / t * 1e-6);
        }
        printf("== divergent gather issue rate, cheap indices (320M gathers)\n");
        {
            const unsigned strides[] = {1, 2, 16, 17, 67};
            for (unsigned ls : strides) {
                double t1 = time_ms([&] { k_gather_cheap<<<g, 256>>>((double *)buf, 2047u, ls, per, dsink); });
                double t2 = time_ms([&] { k_gather_cheap<<<g, 256>>>((double *)buf, 262143u, ls, per, dsink); });
                printf("lane stride %3u doubles: 16 KB table %7.3f ms %7.1f G/s | 2 MB table %7.3f ms %7.1f G/s\n",
                       ls, t1, (double)lanes * per / t1 * 1e-6, t2, (double)lanes * per / t2 * 1e-6);
            }
            CK(hipFuncSetAttribute((const void *)k_gather_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            for (unsigned ls : strides) {
                double t1 = time_ms([&] { k_gather_lds<<<g / 4, 1024, 131072>>>((double *)buf, ls, per, dsink); });
                printf("LDS gather lane stride %3u: %7.3f ms %7.1f G/s\n", ls, t1, (double)lanes * per / t1 * 1e-6);
            }
        }
        printf("== gather cache-policy bits (buffer_load_dwordx2), 2 MB and 512 MB tables\n");
        {
            const size_t ts[] = {(size_t)2 << 20, (size_t)512 << 20};
            for (size_t tb : ts) {
                size_t n = tb / 8;
#define GB(AUX) { double t = time_ms([&] { k_gather_buf<AUX><<<g, 256>>>((double *)buf, n, per, dsink); }); \
                  printf("table %7.1f MB aux=%2d : %7.3f ms %7.1f Ggather/s\n", tb / 1048576.0, AUX, t, (double)lanes * per / t * 1e-6); }
                GB(0) GB(1) GB(2) GB(3) GB(16) GB(17) GB(18) GB(19)
            }
        }
        /* sliding windows over an 80 MB table: what XCD-contiguous row ranges see */
        const size_t wins[] = {(size_t)1 << 9, (size_t)1 << 11, (size_t)1 << 14, (size_t)1 << 17,
                               (size_t)1 << 20, (size_t)1 << 22};
        for (size_t w : wins) {
            size_t n = 10000000;
            double t = time_ms([&] { k_gather<<<g, 256>>>((double *)buf, n, w, per, dsink); });
            printf("table 76.29 MB window %8zu cols: %7.3f ms  %7.1f Ggather/s\n", w, t,
                   (double)lanes * per / t * 1e-6);
        }
    }
    return 0;
}
