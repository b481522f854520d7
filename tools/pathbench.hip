/*
 * pathbench.hip -- round-2 probe: can the scalar data path (s_load through
 * the scalar cache) pull a once-read stream into the XCD L2 fast enough to
 * take the entry stream of the blocked SpMV off the CU's vector-memory queue?
 *
 *   scalar   W persistent waves per CU walk a 2 GiB buffer with
 *            s_load_dwordx16 (64 B) or s_load_dword at a stride of 64 or
 *            128 B, up to 8 loads outstanding per wave -> bytes touched / s
 *   vector   the same walk with one dwordx4 per lane (reference rate)
 *
 * Build: hipcc --offload-arch=gfx950 -O3 tools/pathbench.hip -o tools/pathbench
 * Result of the MI355X run: profiles/r02_pathbench_mi355x.txt
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

/* each wave owns a contiguous slice; 8 scalar loads in flight, then a wait */
template <int WIDE>
__global__ void k_scalar_walk(const char *__restrict__ buf, size_t bytes,
                              int stride, int *sink) {
    const size_t waves = (size_t)gridDim.x * (blockDim.x / 64);
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t slice = (bytes / waves) & ~(size_t)1023;
    const char *p = buf + wave * slice;
    const char *e = p + slice;
    unsigned acc = 0;
    for (; p + 8 * (size_t)stride <= e; p += 8 * (size_t)stride) {
        /* readfirstlane returns int: go through unsigned, or a low word with
         * bit 31 set sign-extends into the high word (the round's first GPU
         * call faulted on exactly that) */
        const unsigned a_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)p);
        const unsigned a_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)p >> 32));
        const uint64_t a = (uint64_t)a_lo | ((uint64_t)a_hi << 32);
        unsigned r0, r1, r2, r3, r4, r5, r6, r7;
        if (WIDE) {
            /* 8 x s_load_dwordx16: 16 SGPRs each; keep only one dword live */
            typedef unsigned u16v __attribute__((ext_vector_type(16)));
            u16v v0, v1, v2, v3;
            asm volatile("s_load_dwordx16 %0, %4, 0x0\n\t"
                         "s_load_dwordx16 %1, %5, 0x0\n\t"
                         "s_load_dwordx16 %2, %6, 0x0\n\t"
                         "s_load_dwordx16 %3, %7, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                         : "s"(a), "s"(a + stride), "s"(a + 2 * (uint64_t)stride),
                           "s"(a + 3 * (uint64_t)stride)
                         : "memory");
            r0 = v0[0]; r1 = v1[0]; r2 = v2[0]; r3 = v3[0];
            asm volatile("s_load_dwordx16 %0, %4, 0x0\n\t"
                         "s_load_dwordx16 %1, %5, 0x0\n\t"
                         "s_load_dwordx16 %2, %6, 0x0\n\t"
                         "s_load_dwordx16 %3, %7, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                         : "s"(a + 4 * (uint64_t)stride), "s"(a + 5 * (uint64_t)stride),
                           "s"(a + 6 * (uint64_t)stride), "s"(a + 7 * (uint64_t)stride)
                         : "memory");
            r4 = v0[0]; r5 = v1[0]; r6 = v2[0]; r7 = v3[0];
        } else {
            asm volatile("s_load_dword %0, %8, 0x0\n\t"
                         "s_load_dword %1, %9, 0x0\n\t"
                         "s_load_dword %2, %10, 0x0\n\t"
                         "s_load_dword %3, %11, 0x0\n\t"
                         "s_load_dword %4, %12, 0x0\n\t"
                         "s_load_dword %5, %13, 0x0\n\t"
                         "s_load_dword %6, %14, 0x0\n\t"
                         "s_load_dword %7, %15, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4),
                           "=&s"(r5), "=&s"(r6), "=&s"(r7)
                         : "s"(a), "s"(a + stride), "s"(a + 2 * (uint64_t)stride),
                           "s"(a + 3 * (uint64_t)stride), "s"(a + 4 * (uint64_t)stride),
                           "s"(a + 5 * (uint64_t)stride), "s"(a + 6 * (uint64_t)stride),
                           "s"(a + 7 * (uint64_t)stride)
                         : "memory");
        }
        acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    }
    if (acc == 0x12345678u)
        *sink = (int)acc;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_vector_walk(const char *__restrict__ buf, size_t bytes, int *sink) {
    const size_t waves = (size_t)gridDim.x * (blockDim.x / 64);
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t slice = (bytes / waves) & ~(size_t)4095;
    const char *p = buf + wave * slice + (threadIdx.x & 63) * 16;
    const char *e = buf + (wave + 1) * slice;
    unsigned acc = 0;
    for (; p + 4096 <= e; p += 4096) {
        u32x4 a = __builtin_nontemporal_load((const u32x4 *)p);
        u32x4 b = __builtin_nontemporal_load((const u32x4 *)(p + 1024));
        u32x4 c = __builtin_nontemporal_load((const u32x4 *)(p + 2048));
        u32x4 d = __builtin_nontemporal_load((const u32x4 *)(p + 3072));
        acc += a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    if (acc == 0x12345678u)
        *sink = (int)acc;
}

/*
 * Does a 4-byte scalar load pull the WHOLE 128-byte line into the XCD's L2?
 * Every wave owns a private 32 KiB region (256 lines) of a cold buffer:
 *   mode 0  nothing             (vector read then misses to HBM)
 *   mode 1  s_load_dword per 128-B line
 *   mode 2  s_load_dword per 64 B
 *   mode 3  one vector dword per 128-B line (known to fill the line)
 * then the wave reads its region with 16 B/lane vector loads and reports the
 * cycles (s_memtime) that read took.  mode 1 == mode 3 << mode 0 would mean
 * scalar touches are a usable L2 prefetch.
 */
template <int MODE>
__global__ void k_touch_then_read(const char *__restrict__ buf, size_t region,
                                  unsigned long long *cycles, int *sink) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const char *p = buf + wave * region;
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    if (MODE == 1 || MODE == 2) {
        const int stride = MODE == 1 ? 128 : 64;
        for (size_t o = 0; o < region; o += 8 * (size_t)stride) {
            const char *q = p + o;
            const unsigned a_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)q);
            const unsigned a_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)q >> 32));
            const uint64_t a = (uint64_t)a_lo | ((uint64_t)a_hi << 32);
            unsigned r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("s_load_dword %0, %8, 0x0\n\t"
                         "s_load_dword %1, %9, 0x0\n\t"
                         "s_load_dword %2, %10, 0x0\n\t"
                         "s_load_dword %3, %11, 0x0\n\t"
                         "s_load_dword %4, %12, 0x0\n\t"
                         "s_load_dword %5, %13, 0x0\n\t"
                         "s_load_dword %6, %14, 0x0\n\t"
                         "s_load_dword %7, %15, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4),
                           "=&s"(r5), "=&s"(r6), "=&s"(r7)
                         : "s"(a), "s"(a + stride), "s"(a + 2 * (uint64_t)stride),
                           "s"(a + 3 * (uint64_t)stride), "s"(a + 4 * (uint64_t)stride),
                           "s"(a + 5 * (uint64_t)stride), "s"(a + 6 * (uint64_t)stride),
                           "s"(a + 7 * (uint64_t)stride)
                         : "memory");
            acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
        }
    } else if (MODE == 3) {
        for (size_t o = 0; o < region; o += 64 * 128)
            acc += *(const unsigned *)(p + o + lane * 128);
    }
    /* make sure the touches have landed (and give HBM time in mode 0 too) */
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (size_t o = 0; o < region; o += 4096) {
        u32x4 a = *(const u32x4 *)(p + o + lane * 16);
        u32x4 b = *(const u32x4 *)(p + o + 1024 + lane * 16);
        u32x4 c = *(const u32x4 *)(p + o + 2048 + lane * 16);
        u32x4 d = *(const u32x4 *)(p + o + 3072 + lane * 16);
        acc += a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0)
        cycles[wave] = t1 - t0;
    if (acc == 0x12345678u)
        *sink = (int)acc;
}

/*
 * How much does the memory side care WHERE 256 persistent workgroups read at
 * one time?  Every workgroup (512 lanes, one per CU) streams the same number
 * of bytes in 3-KiB steps per wavefront; only the address pattern differs:
 *   mode 0  tile-major: workgroup b owns one contiguous 1/256 of the buffer
 *   mode 1  panel-major: the buffer is cut into `panels` windows; inside a
 *           window workgroup b owns a contiguous 1/256 (all workgroups are in
 *           the same window at the same time)
 *   mode 2  block-major: consecutive 24-KiB steps of ALL workgroups are
 *           adjacent (the chip reads one dense moving front)
 */
__global__ void __launch_bounds__(512)
    k_where(const char *__restrict__ buf, size_t bytes, int mode, int panels, int *sink) {
    const size_t per_wg = bytes / gridDim.x;            /* bytes per workgroup */
    const size_t step = 512 * 16 * 3;                   /* 24 KiB per workgroup step */
    const size_t nsteps = per_wg / step;
    const size_t steps_per_panel = nsteps / panels;
    unsigned acc = 0;
    for (size_t k = 0; k < nsteps; ++k) {
        size_t off;
        if (mode == 0) {
            off = blockIdx.x * per_wg + k * step;
        } else if (mode == 1) {
            const size_t p = k / steps_per_panel, j = k % steps_per_panel;
            off = p * (steps_per_panel * step * gridDim.x) +
                  blockIdx.x * (steps_per_panel * step) + j * step;
        } else {
            off = (k * gridDim.x + blockIdx.x) * step;
        }
        if (off + step > bytes)
            break;
        const char *q = buf + off + (size_t)threadIdx.x * 16;
        u32x4 a = __builtin_nontemporal_load((const u32x4 *)q);
        u32x4 b = __builtin_nontemporal_load((const u32x4 *)(q + 8192));
        u32x4 c = __builtin_nontemporal_load((const u32x4 *)(q + 16384));
        acc += a[0] ^ b[1] ^ c[2];
    }
    if (acc == 0x12345678u)
        *sink = (int)acc;
}

template <class F> static double time_ms(F f, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / iters;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const size_t bytes = (size_t)2 << 30;
    char *buf;
    int *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s CUs=%d\n", prop.gcnArchName, cus);
    printf("== vector walk (dwordx4 nt, 4 KiB per wave step), 2 GiB\n");
    for (int wpc : {4, 8, 16, 32}) {
        double ms = time_ms([&] {
            hipLaunchKernelGGL(k_vector_walk, dim3(cus * wpc / 4), dim3(256), 0, 0, buf, bytes, sink);
        }, 3);
        printf("waves/CU %2d : %8.3f ms  %8.1f GB/s\n", wpc, ms, bytes / ms / 1e6);
    }
    printf("== scalar walk, 8 loads in flight per wave\n");
    for (int wide : {1, 0})
        for (int stride : {64, 128})
            for (int wpc : {4, 8, 16, 32}) {
                double ms = time_ms([&] {
                    if (wide)
                        hipLaunchKernelGGL(k_scalar_walk<1>, dim3(cus * wpc / 4), dim3(256), 0, 0,
                                           buf, bytes, stride, sink);
                    else
                        hipLaunchKernelGGL(k_scalar_walk<0>, dim3(cus * wpc / 4), dim3(256), 0, 0,
                                           buf, bytes, stride, sink);
                }, 2);
                const double touched = (double)bytes / stride; /* loads */
                printf("%-16s stride %3d waves/CU %2d : %8.3f ms  %7.2f Gload/s  "
                       "(%8.1f GB/s of lines covered)\n",
                       wide ? "s_load_dwordx16" : "s_load_dword", stride, wpc, ms,
                       touched / ms / 1e6, bytes / ms / 1e6);
            }
    {
        printf("== where 256 persistent workgroups read at one time (2 GiB, 512 lanes each)\n");
        const char *names[3] = {"tile-major (256 distant streams)",
                                "panel-major (77 windows)", "block-major (one front)"};
        for (int mode = 0; mode < 3; ++mode) {
            double ms = time_ms([&] {
                hipLaunchKernelGGL(k_where, dim3(cus), dim3(512), 0, 0, buf, bytes, mode, 77, sink);
            }, 3);
            printf("%-36s : %8.3f ms  %8.1f GB/s\n", names[mode], ms, bytes / ms / 1e6);
        }
    }
    {
        printf("== scalar touch, then vector read of the same 32 KiB (cycles of the read, "
               "mean over 2048 waves, cold regions)\n");
        const size_t region = 32 << 10;
        const int waves = 2048;
        unsigned long long *cyc, h[2048];
        CK(hipMalloc(&cyc, waves * sizeof *cyc));
        const char *names[4] = {"nothing (HBM)", "s_load_dword / 128 B", "s_load_dword / 64 B",
                                "vector dword / 128 B"};
        for (int mode = 0; mode < 4; ++mode) {
            /* a fresh 64 MiB slice of the 2 GiB buffer per mode, far apart */
            const char *base = buf + (size_t)(mode + 1) * (256u << 20);
            if (mode == 0) hipLaunchKernelGGL(k_touch_then_read<0>, dim3(waves / 4), dim3(256), 0, 0, base, region, cyc, sink);
            if (mode == 1) hipLaunchKernelGGL(k_touch_then_read<1>, dim3(waves / 4), dim3(256), 0, 0, base, region, cyc, sink);
            if (mode == 2) hipLaunchKernelGGL(k_touch_then_read<2>, dim3(waves / 4), dim3(256), 0, 0, base, region, cyc, sink);
            if (mode == 3) hipLaunchKernelGGL(k_touch_then_read<3>, dim3(waves / 4), dim3(256), 0, 0, base, region, cyc, sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < waves; ++i) sum += (double)h[i];
            printf("%-24s : %10.0f cycles per 32 KiB read\n", names[mode], sum / waves);
        }
        CK(hipFree(cyc));
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
