// Does vmcnt count buffer loads in issue order when some of them are entirely out of range (returning zeros without a
// memory access)?  Load A (real, far rows: cache misses), then N out-of-range loads, then s_waitcnt vmcnt(N), then read A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N, bool OOB>
__global__ void k(const unsigned char* slab, unsigned bytes, unsigned* bad) {
    const unsigned long long addr = (unsigned long long)slab;
    const i32x4 rs = {(int)(unsigned)addr, (int)(unsigned)(addr >> 32), (int)bytes, 0x00020000};
    const int lane = threadIdx.x & 63;
    unsigned nbad = 0;
    for (int it = 0; it < 64; ++it) {
        u32x4 a = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}, z[N];
        for (int j = 0; j < N; ++j) z[j] = u32x4{7u, 7u, 7u, 7u};
        const unsigned off = ((blockIdx.x * 64 + it) * 1048583u + lane * 4099u) % (bytes / 16) * 16;   // scattered: misses
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(a) : "v"(off), "s"(rs) : "memory");
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const unsigned o2 = OOB ? 0x80000000u : (unsigned)(lane * 16);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(z[j]) : "v"(o2), "s"(rs) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
        if (a[0] == 0xdeadbeefu) ++nbad;     // the slab holds 0x01010101
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int j = 0; j < N; ++j) if (z[j][0] == 12345u) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const unsigned bytes = 1u << 30;
    unsigned char* slab; unsigned* bad;
    hipMalloc(&slab, bytes); hipMemset(slab, 1, bytes); hipMalloc(&bad, 4);
    auto run = [&](auto kern, const char* name) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, slab, bytes, bad);
        unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("%-50s premature reads: %u of %d\n", name, h, 2048 * 256 * 64);
    };
    run(k<4, true>, "A real + 4 out-of-range loads, vmcnt(4)");
    run(k<4, false>, "A real (miss) + 4 real cached loads, vmcnt(4)");
    run(k<1, true>, "A real + 1 out-of-range load, vmcnt(1)");
    return 0;
}
