// probe_req.hip -- does any load flavour make the random 16-byte Bloom lookup cheaper than one 64-byte HBM request?
// One lane x one 16-byte block of a 4 GB table (buffer addressing: 32-bit offsets), cache-policy bits of the
// gfx940+ buffer loads swept (aux: bit0 sc0, bit1 nt, bit4 sc1).  Run under
//   rocprofv3 --pmc TCC_EA_RDREQ_32B_sum TCC_EA_RDREQ_sum --kernel-trace ...
// to see the size of the requests the L2 sends to memory.  Diagnostic only (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31);
}
template <int AUX, int WIDTH>   // WIDTH: bytes per lane (16, 8, 4)
__global__ void __launch_bounds__(256) k_buf(const uint64_t *t, uint32_t nblocks, uint64_t per_lane, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(t), 0, (int)(nblocks * 16u), 0x00020000);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < per_lane; ++i) {
        const uint32_t b = (uint32_t)__umul64hi(mix64(tid * per_lane + i), (uint64_t)nblocks);
        if (WIDTH == 16) { const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(b * 16u), 0, AUX); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
        else if (WIDTH == 8) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(b * 16u), 0, AUX); acc ^= v[0] ^ v[1]; }
        else { acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(b * 16u), 0, AUX); }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
template <int FLAVOUR>   // 0 plain global load, 1 nontemporal builtin
__global__ void __launch_bounds__(256) k_glob(const uint64_t *t, uint32_t nblocks, uint64_t per_lane, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_lane; ++i) {
        const uint64_t b = __umul64hi(mix64(tid * per_lane + i), (uint64_t)nblocks);
        if (FLAVOUR == 0) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(t + b * 2); acc ^= v[0] ^ v[1]; }
        else { acc ^= __builtin_nontemporal_load(t + b * 2) ^ __builtin_nontemporal_load(t + b * 2 + 1); }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
template <int N>
__global__ void __launch_bounds__(256) k_atom(uint64_t *t, uint32_t nblocks, uint64_t per_lane, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < per_lane; ++i) {
        const uint64_t b = __umul64hi(mix64(tid * per_lane + i), (uint64_t)nblocks);
        atomicOr((unsigned long long *)&t[b * 2], 1ull << (i & 63));
        if (N > 1) atomicOr((unsigned long long *)&t[b * 2 + 1], 1ull << (i & 63));
    }
}
template <int N, int SCOPE>   // atomic OR at a given HIP memory scope (1 wavefront, 2 workgroup, 3 agent, 4 system), after a look or blind
__global__ void __launch_bounds__(256) k_atom_scope(uint64_t *t, uint32_t nblocks, uint64_t per_lane, uint64_t *sink, int look) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_lane; ++i) {
        const uint64_t b = __umul64hi(mix64(tid * per_lane + i), (uint64_t)nblocks);
        if (look) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(t + b * 2); acc ^= v.x ^ v.y; }
        __hip_atomic_fetch_or(&t[b * 2], (1ull << (i & 63)) | (acc & 1), __ATOMIC_RELAXED, SCOPE);
        if (N > 1) __hip_atomic_fetch_or(&t[b * 2 + 1], 1ull << (i & 63), __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0x1234567) sink[0] = acc;
}
template <int R>   // R independent loads in flight per lane
__global__ void __launch_bounds__(256) k_globr(const uint64_t *t, uint32_t nblocks, uint64_t per_lane, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_lane; i += R) {
        ulonglong2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const ulonglong2 *>(t + __umul64hi(mix64(tid * per_lane + i + r), (uint64_t)nblocks) * 2);
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= v[r].x ^ v[r].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0 hipMalloc, 1 uncached, 2 fine-grained
    const uint32_t nblocks = 1u << (argc > 3 ? atoi(argv[3]) : 27);      // default 2 GiB of 16-byte blocks (buffer offsets are 32-bit)
    uint64_t *t, *sink;
    if (mode == 0) CK(hipMalloc(&t, (uint64_t)nblocks * 16));
    else CK(hipExtMallocWithFlags((void **)&t, (uint64_t)nblocks * 16, mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    printf("allocation mode %d\n", mode);
    CK(hipMalloc(&sink, 64)); CK(hipMemset(t, 1, (uint64_t)nblocks * 16));
    const int grid = 256 * 8;
    const uint64_t lanes = (uint64_t)grid * 256, per = (1ull << 29) / lanes;
#define RUN(name, kern) { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, t, nblocks, per, sink); }); \
        printf("  %-44s %8.2f ms  %6.2f Gblk/s\n", name, ms, lanes * per / ms / 1e6); }
    RUN("global_load_dwordx4", (k_glob<0>)) RUN("2 x nontemporal 8B", (k_glob<1>))
    RUN("global x4, 4 in flight", (k_globr<4>))
    { const uint64_t pa = per / 4; float ms = timeit([&] { hipLaunchKernelGGL(k_atom<1>, dim3(grid), dim3(256), 0, 0, t, nblocks, pa, sink); }); printf("  %-44s %8.2f ms  %6.2f Gblk/s\n", "atomicOr 8B", ms, lanes * pa / ms / 1e6);
      ms = timeit([&] { hipLaunchKernelGGL(k_atom<2>, dim3(grid), dim3(256), 0, 0, t, nblocks, pa, sink); }); printf("  %-44s %8.2f ms  %6.2f Gblk/s\n", "atomicOr 2 x 8B", ms, lanes * pa / ms / 1e6); }
    if (argc > 2 && argv[2][0] == 'a') {   // atomic scopes
        const uint64_t pa = per / 4;
#define RUNA(name, kern, look) { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, t, nblocks, pa, sink, look); }); \
        printf("  %-44s %8.2f ms  %6.2f Gblk/s\n", name, ms, lanes * pa / ms / 1e6); }
        RUNA("2 x atomicOr wavefront scope, blind", (k_atom_scope<2, __HIP_MEMORY_SCOPE_WAVEFRONT>), 0)
        RUNA("2 x atomicOr workgroup scope, blind", (k_atom_scope<2, __HIP_MEMORY_SCOPE_WORKGROUP>), 0)
        RUNA("2 x atomicOr agent scope, blind", (k_atom_scope<2, __HIP_MEMORY_SCOPE_AGENT>), 0)
        RUNA("2 x atomicOr system scope, blind", (k_atom_scope<2, __HIP_MEMORY_SCOPE_SYSTEM>), 0)
        RUNA("2 x atomicOr wavefront scope, after look", (k_atom_scope<2, __HIP_MEMORY_SCOPE_WAVEFRONT>), 1)
        RUNA("2 x atomicOr workgroup scope, after look", (k_atom_scope<2, __HIP_MEMORY_SCOPE_WORKGROUP>), 1)
        RUNA("2 x atomicOr agent scope, after look", (k_atom_scope<2, __HIP_MEMORY_SCOPE_AGENT>), 1)
        RUNA("1 x atomicOr workgroup scope, blind", (k_atom_scope<1, __HIP_MEMORY_SCOPE_WORKGROUP>), 0)
        RUNA("1 x atomicOr agent scope, blind", (k_atom_scope<1, __HIP_MEMORY_SCOPE_AGENT>), 0)
        return 0;
    }
    if (argc > 2 && argv[2][0] == 's') return 0;
    RUN("buffer b128 aux=0", (k_buf<0, 16>)) RUN("buffer b128 sc0", (k_buf<1, 16>)) RUN("buffer b128 nt", (k_buf<2, 16>)) RUN("buffer b128 sc0 nt", (k_buf<3, 16>))
    RUN("buffer b128 sc1", (k_buf<16, 16>)) RUN("buffer b128 sc1 sc0", (k_buf<17, 16>)) RUN("buffer b128 sc1 nt", (k_buf<18, 16>)) RUN("buffer b128 sc1 sc0 nt", (k_buf<19, 16>))
    RUN("buffer b64 aux=0", (k_buf<0, 8>)) RUN("buffer b64 nt", (k_buf<2, 8>)) RUN("buffer b64 sc1 sc0 nt", (k_buf<19, 8>))
    RUN("buffer b32 aux=0", (k_buf<0, 4>)) RUN("buffer b32 nt", (k_buf<2, 4>)) RUN("buffer b32 sc1 sc0 nt", (k_buf<19, 4>))
    return 0;
}
