// Microbenchmark: what does a 1 KiB wave-load cost on the CU's vector-memory path as a function of the lane -> address
// mapping?  8 waves per workgroup, one workgroup per CU, L2-resident slab of rows.
//   mode 0: contiguous 1 KiB (a packed weight fragment)
//   mode 1: MFMA operand layout -- lane (g = l / 16, r = l % 16) reads 16 B at row[r] + 16 g     (64 lines / instruction)
//   mode 2: quad-coalesced      -- lane (r = l / 4, c = l % 4)  reads 16 B at row[r] + 16 c      (16 lines / instruction)
//   mode 3: as mode 1, every lane out of range (returns zeros)
//   mode 4: as mode 2 through the LDS-DMA path (buffer_load_dwordx4 ... lds), no VGPR destination
//   mode 5: MFMA operand layout over a FRAGMENT-BLOCKED slab -- 16 consecutive rows interleaved per 16-byte chunk: chunk c of row
//           r lives at (r / 16) * 16 * row_bytes + c * 256 + (r % 16) * 16 -- with the 16 rows of a fragment in RUNS of `runlen`
//           consecutive rows from random starts (round 4: would consecutive neighbour rows coalesce again?)
//   mode 6: as mode 1 (row-major slab) with the same runs (control: runs alone buy nothing in a row-major slab)
//   mode 7: as mode 1, a row has a neighbour with probability pop % (the others' lanes are out of range) -- round 5: does the cost
//           follow the instructions issued or the lanes that return data?
//   mode 8: as mode 7 with the empty rows' lanes switched off in EXEC instead of out of range
//   mode 9: as mode 2 (quad-coalesced) with the same population
//   mode 10: whole rows contiguous -- lane l reads chunk l % (row_bytes / 16) of row l / (row_bytes / 16)
//   mode 11: as mode 1 with EXEC = 0 for the whole instruction (inline asm), does an empty instruction cost anything?
// build: hipcc --offload-arch=gfx950 -O3 gather_layout.hip -o gather_layout ; run: ./gather_layout [row_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* slab, unsigned slab_bytes, int n_rows, int row_bytes,
                                         const int* rows, int iters, unsigned* sink, unsigned long long* cyc, int runlen, int pop, int deep) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long addr = (unsigned long long)slab;
    const i32x4 rs = {(int)(unsigned)addr, (int)(unsigned)(addr >> 32), (int)slab_bytes, 0x00020000};
    // rows of a "fragment" are generated arithmetically (no index loads in the timed loop): a base row per load + a
    // per-row scatter inside a 48-row window, like the neighbours of a Z-order run
    const unsigned seed = (blockIdx.x * 8 + wave) * 2654435761u;
    auto row_of = [&](int load, int r) -> unsigned {
        unsigned h = seed + (unsigned)load * 0x9E3779B9u;
        h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
        unsigned q = (h + (unsigned)r * 0xC2B2AE35u); q ^= q >> 16; q *= 0x27D4EB2Fu; q ^= q >> 15;
        return (h % (unsigned)(n_rows - 64)) + (q % 48u);
    };
    u32x4 acc = {0, 0, 0, 0};
    // round 5: 16 loads in flight per wave (four batches of four; a batch is waited for when three younger ones are queued behind
    // it), so that the figure is the throughput of the path and not the latency of four loads (PBN_MICRO_DEEP=0: the old form)
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = u32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (deep && (j & 3) == 0) {
                asm volatile("s_waitcnt vmcnt(12)" : "+v"(v[j]), "+v"(v[j + 1]), "+v"(v[j + 2]), "+v"(v[j + 3]));
                acc |= v[j] | v[j + 1] | v[j + 2] | v[j + 3];
            }
            unsigned voff;
            if (MODE == 0) voff = (row_of(i + j, 0) * (unsigned)row_bytes) / 1024u * 1024u + lane * 16;
            else if (MODE == 1 || MODE == 3) voff = row_of(i + j, lane & 15) * (unsigned)row_bytes + (lane >> 4) * 16;
            else if (MODE == 5 || MODE == 6) {
                const int r = lane & 15;
                const unsigned row = row_of(i + j, r / runlen) + (unsigned)(r % runlen);     // runs of runlen consecutive rows
                if (MODE == 5) voff = (row >> 4) * 16u * (unsigned)row_bytes + (unsigned)(lane >> 4) * 256u + (row & 15u) * 16u;
                else voff = row * (unsigned)row_bytes + (lane >> 4) * 16;
            }
            else if (MODE == 7 || MODE == 8 || MODE == 11) {
                const unsigned rw = row_of(i + j, lane & 15);
                voff = rw * (unsigned)row_bytes + (lane >> 4) * 16;
                if (MODE != 11 && (rw * 2654435761u >> 8) % 100u >= (unsigned)pop) voff = 0x80000000u;
            } else if (MODE == 9) {
                const unsigned rw = row_of(i + j, lane >> 2);
                voff = rw * (unsigned)row_bytes + (lane & 3) * 16;
                if ((rw * 2654435761u >> 8) % 100u >= (unsigned)pop) voff = 0x80000000u;
            } else if (MODE == 10) {
                const int cpr = row_bytes / 16;
                voff = row_of(i + j, lane / cpr) * (unsigned)row_bytes + (lane % cpr) * 16;
            }
            else voff = row_of(i + j, lane >> 2) * (unsigned)row_bytes + (lane & 3) * 16;
            if (MODE == 3) voff = 0x80000000u;
            if (MODE == 8) {
                if (voff != 0x80000000u) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(v[j]) : "v"(voff), "s"(rs) : "memory");
                continue;
            }
            if (MODE == 11) {
                asm volatile("s_mov_b64 exec, 0\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_mov_b64 exec, -1" : "+v"(v[j]) : "v"(voff), "s"(rs) : "memory");
                continue;
            }
            if (MODE == 4) {
                const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(lds + wave * 4096 + (j & 3) * 1024));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(base)), "v"(voff), "s"(rs) : "memory");
            } else {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(v[j]) : "v"(voff), "s"(rs) : "memory");
            }
        }
        if (!deep) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                         "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= v[j];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                 "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
    for (int j = 0; j < 16; ++j) acc |= v[j];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (MODE == 4) acc |= *reinterpret_cast<u32x4*>(lds + wave * 4096 + lane * 16);
    if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 512;
    const int n_rows = argc > 2 ? atoi(argv[2]) : 4096, iters = 256, wgs = 256;
    const int deep = getenv("PBN_MICRO_DEEP") ? atoi(getenv("PBN_MICRO_DEEP")) : 1;
    const unsigned slab_bytes = (unsigned)n_rows * row_bytes;
    unsigned char* slab; int* rows; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&slab, slab_bytes); hipMemset(slab, 1, slab_bytes);
    std::vector<int> h((size_t)wgs * 8 * iters * 16);
    srand(1);
    for (size_t i = 0; i < h.size(); i += 16) {               // 16 rows of a "fragment": neighbours of a Z-order run
        const int base = rand() % (n_rows - 64);
        for (int r = 0; r < 16; ++r) h[i + r] = base + (rand() % 48);
    }
    hipMalloc(&rows, h.size() * 4); hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 16); hipMalloc(&cyc, wgs * 8 * 8);
    std::vector<unsigned long long> hc(wgs * 8);
    auto run = [&](auto kern, const char* name, int runlen = 1, int pop = 100) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, slab, slab_bytes, n_rows, row_bytes, rows, iters, sink, cyc, runlen, pop, deep);
        hipDeviceSynchronize();
        hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto c : hc) s += (double)c;
        const double per_wave = s / hc.size();
        printf("%-44s %8.0f cycles per wave for %d loads -> %.1f cycles per wave-load, %.1f per CU-load (8 waves), %.1f B/clk/CU\n", name,
               per_wave, iters, per_wave / iters, per_wave / iters / 8, 1024.0 * 8 * iters / per_wave);
    };
    printf("rows of %d bytes, %d rows (slab %.1f MB)\n", row_bytes, n_rows, slab_bytes / 1e6);
    run(k<0>, "0 contiguous 1 KiB");
    run(k<1>, "1 MFMA layout (16 rows x 4 x 16 B)");
    run(k<2>, "2 quad-coalesced (16 rows x 64 B)");
    run(k<3>, "3 MFMA layout, all lanes out of range");
    run(k<4>, "4 quad-coalesced through LDS-DMA");
    for (int rl : {1, 2, 4, 8, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "5 MFMA layout, blocked slab, runs of %d", rl);
        run(k<5>, nm, rl);
    }
    for (int rl : {4, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "6 MFMA layout, row-major slab, runs of %d", rl);
        run(k<6>, nm, rl);
    }
    for (int pop : {100, 50, 36, 20}) {
        char nm[96];
        snprintf(nm, sizeof nm, "7 MFMA layout, %d %% of rows populated (OOB)", pop);
        run(k<7>, nm, 1, pop);
        snprintf(nm, sizeof nm, "8 MFMA layout, %d %% of rows populated (EXEC)", pop);
        run(k<8>, nm, 1, pop);
        snprintf(nm, sizeof nm, "9 quad-coalesced, %d %% of rows populated (OOB)", pop);
        run(k<9>, nm, 1, pop);
    }
    run(k<10>, "10 whole rows contiguous");
    run(k<11>, "11 MFMA layout, EXEC = 0");
    return 0;
}
