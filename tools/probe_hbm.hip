// probe_hbm.hip -- measures what random 64-byte block reads of a multi-GB table can reach on
// MI355X, in the access shapes the Bloom kernels could use.  Diagnostic only (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31);
}
// shape A: 8 lanes x 8 B per block, R independent blocks in flight per lane group
template <int R>
__global__ void __launch_bounds__(256) k_a(const uint64_t* t, uint64_t nblocks, uint64_t per_group, uint64_t* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grp = tid >> 3; const int sub = tid & 7;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_group; i += R) {
        uint64_t v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint64_t b = mix64(grp * per_group + i + r) % nblocks; v[r] = t[b * 8 + sub]; }
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= v[r];
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// shape B: 4 lanes x 16 B
template <int R>
__global__ void __launch_bounds__(256) k_b(const uint64_t* t, uint64_t nblocks, uint64_t per_group, uint64_t* sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grp = tid >> 2; const int sub = tid & 3;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_group; i += R) {
        ulonglong2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint64_t b = mix64(grp * per_group + i + r) % nblocks; v[r] = *reinterpret_cast<const ulonglong2*>(t + b * 8 + sub * 2); }
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= v[r].x ^ v[r].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// shape C: 1 lane x 64 B (4 x 16 B)
template <int R>
__global__ void __launch_bounds__(256) k_c(const uint64_t* t, uint64_t nblocks, uint64_t per_group, uint64_t* sink) {
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_group; i += R) {
        ulonglong2 v[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint64_t b = mix64(grp * per_group + i + r) % nblocks; const ulonglong2* p = reinterpret_cast<const ulonglong2*>(t + b * 8);
            v[r][0] = p[0]; v[r][1] = p[1]; v[r][2] = p[2]; v[r][3] = p[3]; }
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= v[r][0].x ^ v[r][1].y ^ v[r][2].x ^ v[r][3].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// 8 lanes x 8 B atomic OR (no return) on random blocks
__global__ void __launch_bounds__(256) k_atomic(uint64_t* t, uint64_t nblocks, uint64_t per_group) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t grp = tid >> 3; const int sub = tid & 7;
    for (uint64_t i = 0; i < per_group; ++i) {
        const uint64_t b = mix64(grp * per_group + i) % nblocks;
        atomicOr((unsigned long long*)&t[b * 8 + sub], 1ull << (i & 63));
    }
}
// shape D: 1 lane x 16 B -- one whole 128-bit block per lane, 64 distinct lines per wave instruction
template <int R, bool WITH_PATTERN>
__global__ void __launch_bounds__(256) k_d(const uint64_t* t, uint64_t nblocks, uint64_t per_group, uint64_t* sink, const uint64_t* pat) {
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_group; i += R) {
        ulonglong2 v[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t h = mix64(grp * per_group + i + r);
            const uint64_t b = (uint64_t)__umul64hi(h, nblocks);
            v[r] = *reinterpret_cast<const ulonglong2*>(t + b * 2);
            if (WITH_PATTERN) q[r] = *reinterpret_cast<const ulonglong2*>(pat + ((h >> 7) & 0xFFFF) * 2);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { acc ^= v[r].x ^ v[r].y; if (WITH_PATTERN) acc ^= q[r].x & q[r].y; }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// 1 lane x 8 B atomic OR on a random 16-byte block (N per block: 1 or 2 words)
template <int N>
__global__ void __launch_bounds__(256) k_atomic16(uint64_t* t, uint64_t nblocks, uint64_t per_group) {
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < per_group; ++i) {
        const uint64_t b = (uint64_t)__umul64hi(mix64(grp * per_group + i), nblocks);
        atomicOr((unsigned long long*)&t[b * 2], 1ull << (i & 63));
        if (N > 1) atomicOr((unsigned long long*)&t[b * 2 + 1], 1ull << (i & 63));
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 40.0;
    const uint64_t nblocks = (uint64_t)(gb * 1e9 / 64);
    uint64_t *t, *sink; CK(hipMalloc(&t, nblocks * 64)); CK(hipMalloc(&sink, 64)); CK(hipMemset(t, 1, nblocks * 64));
    const uint64_t Q = 1ull << 30;   // block reads per measurement
    if (argc > 2) {   // compact-block shapes: the table is nblocks16 x 16 bytes
        const uint64_t nb16 = (uint64_t)(gb * 1e9 / 16);
        uint64_t* pat; CK(hipMalloc(&pat, 1 << 20)); CK(hipMemset(pat, 3, 1 << 20));
        for (int blocks_per_cu : {4, 8}) {
            const int grid = 256 * blocks_per_cu;
            printf("table %.1f GB as 16-byte blocks, grid %d x 256\n", gb, grid);
#define RUND(name, kern) { const uint64_t groups = (uint64_t)grid * 256; const uint64_t per = Q / groups; \
            float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, t, nb16, per, sink, pat); }); \
            printf("  %-36s %8.2f ms  %6.2f Gblk/s\n", name, ms, groups * per / ms / 1e6); }
            RUND("1 lane x 16B, 1 in flight", (k_d<1, false>)) RUND("1 lane x 16B, 2 in flight", (k_d<2, false>)) RUND("1 lane x 16B, 4 in flight", (k_d<4, false>))
            RUND("1 lane x 16B + 1MiB pattern, 1", (k_d<1, true>)) RUND("1 lane x 16B + 1MiB pattern, 2", (k_d<2, true>)) RUND("1 lane x 16B + 1MiB pattern, 4", (k_d<4, true>))
            { const uint64_t groups = (uint64_t)grid * 256; const uint64_t per = (Q / 4) / groups;
              float ms = timeit([&] { hipLaunchKernelGGL(k_atomic16<1>, dim3(grid), dim3(256), 0, 0, t, nb16, per); });
              printf("  %-36s %8.2f ms  %6.2f Gblk/s\n", "1 lane x 8B atomicOr", ms, groups * per / ms / 1e6);
              ms = timeit([&] { hipLaunchKernelGGL(k_atomic16<2>, dim3(grid), dim3(256), 0, 0, t, nb16, per); });
              printf("  %-36s %8.2f ms  %6.2f Gblk/s\n", "1 lane x 2 x 8B atomicOr", ms, groups * per / ms / 1e6); }
        }
        return 0;
    }
    for (int blocks_per_cu : {4, 8}) {
        const int grid = 256 * blocks_per_cu;
        printf("table %.1f GB, grid %d x 256\n", gb, grid);
#define RUN(name, kern, lanes_per_blk) { const uint64_t groups = (uint64_t)grid * 256 / lanes_per_blk; const uint64_t per = Q / groups; \
        float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, t, nblocks, per, sink); }); \
        printf("  %-28s %8.2f ms  %7.1f GB/s  %6.2f Gblk/s\n", name, ms, groups * per * 64 / ms / 1e6, groups * per / ms / 1e6); }
        RUN("8 lanes x 8B, 1 in flight", k_a<1>, 8) RUN("8 lanes x 8B, 4 in flight", k_a<4>, 8) RUN("8 lanes x 8B, 8 in flight", k_a<8>, 8) RUN("8 lanes x 8B, 16 in flight", k_a<16>, 8)
        RUN("4 lanes x 16B, 4 in flight", k_b<4>, 4) RUN("4 lanes x 16B, 8 in flight", k_b<8>, 4)
        RUN("1 lane x 64B, 1 in flight", k_c<1>, 1) RUN("1 lane x 64B, 2 in flight", k_c<2>, 1) RUN("1 lane x 64B, 4 in flight", k_c<4>, 1)
        { const uint64_t groups = (uint64_t)grid * 256 / 8; const uint64_t per = (Q / 4) / groups;
          float ms = timeit([&] { hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, 0, t, nblocks, per); });
          printf("  %-28s %8.2f ms  %7.1f GB/s  %6.2f Gblk/s\n", "8 lanes x 8B atomicOr", ms, groups * per * 64 / ms / 1e6, groups * per / ms / 1e6); }
    }
    return 0;
}
