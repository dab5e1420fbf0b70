// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS path's access patterns (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel touches each byte of
// a slab far larger than the 256 MiB Infinity Cache exactly once, so the bytes that must cross the fabric are known:
//   k_stream16   coalesced streaming read, 16 B per lane                       (the guide's case: FETCH_SIZE reports 1/2)
//   k_stream4    coalesced streaming read, 4 B per lane                        (rulebook tiles that are not int4-aligned)
//   k_gather_q   rows of ROW bytes through a random permutation, quad-coalesced: lane 4 r + c reads 16-B chunk c of row r of a
//                16-row fragment, piece by 64-byte piece                       (k_spconv_rs / k_spconv_wave gathers)
//   k_gather_op  the same rows in MFMA operand order: lane 16 g + r reads chunk g of row r   (k_spconv's gathers)
//   k_gather_dma k_gather_q through buffer_load_dwordx4 ... lds                (k_spconv_rsh staging, weight rings)
//   k_write16    coalesced streaming write, 16 B per lane                      (WRITE_SIZE)
//   k_write_rows rows of ROW bytes written in fragment order, 8 B per lane (the epilogue's bf16 stores: 4 channels per lane)
// build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE):
//   ./fetch_calib [row_bytes=192]   prints the bytes every kernel touched; scripts/fetch_calib.sh joins them with the counters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const u32x4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const u32x4 v = p[i]; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_stream4(const unsigned* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345u) *sink = acc;
}
// one wave per 16-row fragment per step; MODE 0 quad-coalesced, 1 operand order, 2 quad-coalesced through the LDS-DMA path
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* __restrict__ slab, unsigned long long slab_bytes, const int* __restrict__ perm,
                                                int n_rows, int row_bytes, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long a = (unsigned long long)slab;
    const i32x4 rs = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)0xffffffffu, 0x00020000};   // (the slab is < 4 GiB)
    const int n_frag = n_rows / 16, pieces = row_bytes / 64;
    unsigned acc = 0;
    for (int f = blockIdx.x * 4 + wave; f < n_frag; f += gridDim.x * 4) {
        const int r = MODE == 1 ? (lane & 15) : (lane >> 2), c = MODE == 1 ? (lane >> 4) : (lane & 3);
        const unsigned row = (unsigned)perm[f * 16 + r];
        for (int p = 0; p < pieces; ++p) {
            const unsigned off = row * (unsigned)row_bytes + (unsigned)p * 64u + (unsigned)c * 16u;
            if (MODE == 2) {
                const unsigned m0 = (unsigned)(size_t)(&lds[wave][0]);
                asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(m0)), "v"(off), "s"(rs) : "memory");
            } else {
                u32x4 v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rs) : "memory");
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            }
        }
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc ^= *reinterpret_cast<unsigned*>(&lds[wave][lane * 16]); }
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write16(u32x4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
}
// the epilogue's store shape: a wave writes a 16-row x 16-channel bf16 tile, lane 16 g + r stores 8 bytes (channels 4 g .. 4 g + 3) of row r
__global__ __launch_bounds__(256) void k_write_rows(unsigned char* __restrict__ slab, int n_rows, int row_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_frag = n_rows / 16, tiles = row_bytes / 32;
    for (int f = blockIdx.x * 4 + wave; f < n_frag; f += gridDim.x * 4)
        for (int t = 0; t < tiles; ++t) {
            const size_t off = (size_t)(f * 16 + (lane & 15)) * row_bytes + t * 32 + (lane >> 4) * 8;
            *reinterpret_cast<uint2*>(slab + off) = make_uint2((unsigned)f, (unsigned)t);
        }
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 192;
    const size_t slab_bytes = (size_t)3 << 30;                      // 3 GiB: 12 x the Infinity Cache
    const int n_rows = (int)(slab_bytes / row_bytes / 16 * 16);
    unsigned char* slab; int* perm; unsigned* sink;
    CK(hipMalloc(&slab, slab_bytes)); CK(hipMalloc(&perm, (size_t)n_rows * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(slab, 1, slab_bytes));
    std::vector<int> h(n_rows); std::iota(h.begin(), h.end(), 0);
    std::mt19937 g(7); std::shuffle(h.begin(), h.end(), g);
    CK(hipMemcpy(perm, h.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice));
    const size_t row_total = (size_t)n_rows * row_bytes;
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(256), 0, 0, (const u32x4*)slab, slab_bytes / 16, sink);
        hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(256), 0, 0, (const unsigned*)slab, slab_bytes / 4, sink);
        hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, slab, (unsigned long long)slab_bytes, perm, n_rows, row_bytes, sink);
        hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, slab, (unsigned long long)slab_bytes, perm, n_rows, row_bytes, sink);
        hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, slab, (unsigned long long)slab_bytes, perm, n_rows, row_bytes, sink);
        hipLaunchKernelGGL(k_write16, dim3(grid), dim3(256), 0, 0, (u32x4*)slab, slab_bytes / 16);
        hipLaunchKernelGGL(k_write_rows, dim3(grid), dim3(256), 0, 0, slab, n_rows, row_bytes);
        CK(hipDeviceSynchronize());
    }
    printf("row_bytes %d n_rows %d\n", row_bytes, n_rows);
    printf("BYTES k_stream16 %zu\nBYTES k_stream4 %zu\nBYTES k_gather<0> %zu\nBYTES k_gather<1> %zu\nBYTES k_gather<2> %zu\nBYTES k_write16 %zu\nBYTES k_write_rows %zu\n",
           slab_bytes, slab_bytes, row_total + (size_t)n_rows * 4, row_total + (size_t)n_rows * 4, row_total + (size_t)n_rows * 4, slab_bytes, row_total);
    return 0;
}
