// vmcnt and PARTIALLY out-of-range gathers: load A has in-range lanes (scattered: cache misses) and out-of-range lanes
// (zeros); N younger loads follow; s_waitcnt vmcnt(N); every lane of A must hold its final value.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N, int PATTERN>
__global__ void k(const unsigned char* slab, unsigned bytes, unsigned* bad) {
    const unsigned long long addr = (unsigned long long)slab;
    const i32x4 rs = {(int)(unsigned)addr, (int)(unsigned)(addr >> 32), (int)bytes, 0x00020000};
    const int lane = threadIdx.x & 63;
    unsigned nbad = 0;
    for (int it = 0; it < 64; ++it) {
        u32x4 a = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}, z[N];
        for (int j = 0; j < N; ++j) z[j] = u32x4{7u, 7u, 7u, 7u};
        unsigned off = ((blockIdx.x * 64 + it) * 1048583u + (lane & 15) * 40961u) % (bytes / 64) * 64 + (lane >> 4) * 16;
        const bool oob = PATTERN == 0 ? (lane & 1) : PATTERN == 1 ? ((lane & 15) < 9) : ((lane & 15) == 3);
        if (oob) off = 0x80000000u;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(a) : "v"(off), "s"(rs) : "memory");
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const unsigned o2 = (unsigned)(((blockIdx.x + j) * 7919u + lane) % (bytes / 16)) * 16;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(z[j]) : "v"(o2), "s"(rs) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
        const unsigned want = oob ? 0u : 0x01010101u;
        if (a[0] != want || a[3] != want) ++nbad;
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
        printf("%-60s wrong lanes: %u of %d\n", name, h, 2048 * 256 * 64);
    };
    run(k<5, 0>, "odd lanes out of range, 5 younger loads, vmcnt(5)");
    run(k<5, 1>, "9 of 16 rows out of range, 5 younger loads, vmcnt(5)");
    run(k<5, 2>, "1 of 16 rows out of range, 5 younger loads, vmcnt(5)");
    run(k<1, 1>, "9 of 16 rows out of range, 1 younger load, vmcnt(1)");
    return 0;
}
