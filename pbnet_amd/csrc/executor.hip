// executor.hip -- whole sub-pipelines behind ONE C call each, so the host issues a coordinate pyramid or a fused
// U-Net forward in microseconds instead of hundreds of interpreter round trips:
//   pbn_coords_build  : de-duplication + 4 strided levels + every kernel map a MinkUNet needs (k=5 @1, k=3 @1..16,
//                       up tables @2..16) into one caller-owned arena; row counts stay on the device.
//   pbn_unet_forward  : executes a static plan (list of fused convolution ops over symbolic buffers) -- the body of
//                       MinkUNetBase.forward (/root/reference/network/Mink.py:291-354) with BatchNorm(eval)/ReLU/residual
//                       folded into the convolution epilogues and skip concatenations written in place.
// Both only sequence the kernels of coords.hip / spconv.hip; no new arithmetic lives here.
#include <cstdlib>
#include <cstring>
#include "pbn_common.h"
#include "spconv_common.h"

using namespace pbn;

static inline size_t a256(size_t x) { return align_up(x, 256); }

extern "C" size_t pbn_coords_arena_bytes(int n, int want_k5, pbn_coords_layout* L) {
    if (n < 0 || !L) return 0;
    const size_t N = (size_t)(n > 0 ? n : 1);
    const int cap = pbn_hash_capacity(n);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = a256(off + bytes); return (int64_t)o; };
    // fill-pattern groups are contiguous so that pbn_coords_build clears a whole pyramid with three memsets:
    //   [counts + status: 0x00] [keys of 5 levels, nbr_down of 4 levels: 0xff] [values of 5 levels: 0x7f]
    L->counts = take(16 * sizeof(int));                      // 5 row counts, [8] = range-error status
    for (int l = 0; l < 5; ++l) { L->capacity[l] = cap; L->keys[l] = take((size_t)cap * 8); }
    for (int l = 0; l < 4; ++l) L->nbr_down[l] = take(N * 8 * 4);
    for (int l = 0; l < 5; ++l) L->vals[l] = take((size_t)cap * 4);
    const int64_t fill_end = (int64_t)off;
    (void)fill_end;
    L->unique_index = take(N * 4);
    L->inverse = take(N * 4);
    for (int l = 0; l < 5; ++l) {
        L->coords[l] = take(N * 16);
        L->k3[l] = take(N * 27 * 4);
    }
    for (int l = 0; l < 4; ++l) {
        L->parent_row[l] = take(N * 4);
        L->child_k[l] = take(N * 4);
        L->up[l] = take(N * 8 * 4);
    }
    L->k5 = want_k5 ? take(N * 125 * 4) : -1;
    L->workspace = take(pbn_coords_workspace_bytes(n));
    L->workspace_bytes = (int64_t)pbn_coords_workspace_bytes(n);
    return off;
}

extern "C" int pbn_coords_build(const int32_t* coords, const int32_t* n_dev, int n, int want_k5, int x_fastest, void* arena,
                                size_t arena_bytes, const pbn_coords_layout* L, pbn_stream_t stream) {
    if (n < 0 || !arena || !L) return PBN_ERR_ARG;
    pbn_coords_layout chk;
    if (pbn_coords_arena_bytes(n, want_k5, &chk) > arena_bytes) return PBN_ERR_WORKSPACE;
    char* A = (char*)arena;
    auto I = [&](int64_t o) { return (int32_t*)(A + o); };
    int32_t* counts = I(L->counts);
    void* ws = A + L->workspace;
    const size_t wsb = (size_t)L->workspace_bytes;
    hipStream_t st = (hipStream_t)stream;
    {
        const FillRange fr[3] = {{A + L->counts, 16 * sizeof(int), 0}, {A + L->keys[0], (size_t)(L->vals[0] - L->keys[0]), 0xff},
                                 {A + L->vals[0], (size_t)(L->unique_index - L->vals[0]), 0x7f}};
        const int frc = fill_ranges(fr, 3, st);
        if (frc != PBN_OK) return frc;
    }
    int rc = coords_unique_impl(coords, n_dev, n, (uint64_t*)(A + L->keys[0]), I(L->vals[0]), L->capacity[0],
                                I(L->unique_index), I(L->inverse), I(L->coords[0]), counts + 0, ws, wsb, counts + 8, false,
                                st);
    if (rc != PBN_OK) return rc;
    return coords_build_upper(n, want_k5, x_fastest, arena, L, st);
}

// levels 1..4, the k=3 maps of all levels, the k=5 map of level 0: level 0 (coords, table, count) must be finished and
// the fill-pattern groups of the arena cleared
int pbn::coords_build_upper(int n, int want_k5, int x_fastest, void* arena, const pbn_coords_layout* L, hipStream_t st) {
    char* A = (char*)arena;
    auto I = [&](int64_t o) { return (int32_t*)(A + o); };
    int32_t* counts = I(L->counts);
    void* ws = A + L->workspace;
    const size_t wsb = (size_t)L->workspace_bytes;
    int rc = PBN_OK;
    for (int l = 0; l < 4; ++l) {
        rc = coords_stride_impl(I(L->coords[l]), counts + l, n, 2 << l, (uint64_t*)(A + L->keys[l + 1]), I(L->vals[l + 1]),
                                L->capacity[l + 1], I(L->coords[l + 1]), I(L->parent_row[l]), I(L->child_k[l]),
                                I(L->nbr_down[l]), I(L->up[l]), counts + l + 1, ws, wsb, counts + 8, false, st);
        if (rc != PBN_OK) return rc;
    }
    {   // the k=3 map of every level and the k=5 map of level 0: one launch
        MapJobs jb;
        jb.n_jobs = 0; jb.n_max = n; jb.x_fastest = x_fastest;
        long long total = 0;
        auto add = [&](int l, int ksize, int32_t* out) {
            const int j = jb.n_jobs++;
            jb.coords[j] = I(L->coords[l]); jb.n_dev[j] = counts + l; jb.keys[j] = (const unsigned long long*)(A + L->keys[l]);
            jb.vals[j] = I(L->vals[l]); jb.nbr[j] = out; jb.mask[j] = (unsigned)L->capacity[l] - 1; jb.ksize[j] = ksize;
            jb.stride[j] = 1 << l;
            total += (long long)n * ksize * ksize * ksize;
        };
        for (int l = 0; l < 5; ++l) add(l, 3, I(L->k3[l]));
        if (want_k5) add(0, 5, I(L->k5));
        rc = kernel_maps_multi(jb, total, st);
        if (rc != PBN_OK) return rc;
    }
    return PBN_OK;
}

// ---- pbn_coords_prepare: de-duplication, Z-order, pyramid and maps of one SparseTensor lineage in ONE call -----------
extern "C" size_t pbn_coords_prepare_bytes(int n, int want_k5, pbn_prepare_layout* P) {
    if (n < 0 || !P) return 0;
    const size_t N = (size_t)(n > 0 ? n : 1);
    const size_t pyr = pbn_coords_arena_bytes(n, want_k5, &P->pyramid);
    if (!pyr) return 0;
    size_t off = a256(pyr);
    auto take = [&](size_t bytes) { size_t o = off; off = a256(off + bytes); return (int64_t)o; };
    const int cap = pbn_hash_capacity(n);
    P->n_unique = take(16 * sizeof(int));               // [0] survivors (or -1), [8] range-error status
    P->tmp_keys = take((size_t)cap * 8);
    P->tmp_vals = take((size_t)cap * 4);
    P->unique_index = take(N * 8);
    P->inverse = take(N * 8);
    P->perm = take(N * 8);
    P->inv_perm = take(N * 8);
    P->ucoords = take(N * 16);
    P->uidx32 = take(N * 4);
    P->inv32 = take(N * 4);
    P->sort_keys = take(2 * N * 8);
    P->sort_vals = take(2 * N * 4);
    {   // scratch of either pipeline: the library sort's temporary storage / the scan states and level keys of pyramid.hip
        const size_t a = sort_pairs_temp_bytes(n), b = pyramid_scratch_bytes(n);
        P->sort_temp_bytes = (int64_t)(a > b ? a : b);
    }
    P->sort_temp = take((size_t)P->sort_temp_bytes);
    return off;
}

static int coords_prepare_impl(const int32_t* coords, const int32_t* n_dev, int n, int want_k5, int x_fastest, void* arena,
                               size_t arena_bytes, const pbn_prepare_layout* P, pbn_stream_t stream, bool force_hash = false) {
    if (n < 0 || !arena || !P) return PBN_ERR_ARG;
    pbn_prepare_layout chk;
    if (pbn_coords_prepare_bytes(n, want_k5, &chk) > arena_bytes) return PBN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // PBN_PREPARE_HASH=1: the round-2 pipeline (a hash table per level + the library radix sort), kept as the cross-check of
    // pyramid.hip (tests/test_pyramid_gpu.py compares every array of the two)
    static const int hash_env = getenv("PBN_PREPARE_HASH") ? atoi(getenv("PBN_PREPARE_HASH")) : 0;
    if (!hash_env && !force_hash) return coords_prepare_sorted(coords, n_dev, n, want_k5, x_fastest, arena, P, st);
    char* A = (char*)arena;
    const pbn_coords_layout* L = &P->pyramid;
    auto I = [&](int64_t o) { return (int32_t*)(A + o); };
    const size_t N = (size_t)(n > 0 ? n : 1);
    // clears: the temporary table + the pyramid's three fill-pattern groups
    {   // ... in ONE launch (six hipMemsetAsync = six launches per pyramid)
        const FillRange fr[6] = {{A + P->n_unique, 16 * sizeof(int), 0},
                                 {A + P->tmp_keys, (size_t)(P->tmp_vals - P->tmp_keys), 0xff},
                                 {A + P->tmp_vals, (size_t)(P->unique_index - P->tmp_vals), 0x7f},
                                 {A + L->counts, 16 * sizeof(int), 0},
                                 {A + L->keys[0], (size_t)(L->vals[0] - L->keys[0]), 0xff},
                                 {A + L->vals[0], (size_t)(L->unique_index - L->vals[0]), 0x7f}};
        const int frc = fill_ranges(fr, 6, st);
        if (frc != PBN_OK) return frc;
    }
    if (n == 0) return PBN_OK;
    if (!coords) return PBN_ERR_ARG;
    int32_t* n_unique = I(P->n_unique);
    // 1. de-duplication in the external (first occurrence, ascending) order
    int rc = coords_unique_impl(coords, n_dev, n, (uint64_t*)(A + P->tmp_keys), I(P->tmp_vals), pbn_hash_capacity(n),
                                I(P->uidx32), I(P->inv32), I(P->ucoords), n_unique, A + L->workspace,
                                (size_t)L->workspace_bytes, n_unique + 8, false, st);
    if (rc != PBN_OK) return rc;
    // 2. Z-order: keys, stable radix sort of (key, row), permutation pair + sorted coordinates
    uint64_t* keys = (uint64_t*)(A + P->sort_keys);
    int32_t* vals = I(P->sort_vals);
    rc = coords_morton_iota(I(P->ucoords), n_unique, n, keys, vals, st);
    if (rc != PBN_OK) return rc;
    rc = sort_pairs_u64_i32(keys, keys + N, vals, vals + N, n, A + P->sort_temp, (size_t)P->sort_temp_bytes, st);
    if (rc != PBN_OK) return rc;
    rc = coords_apply_perm(I(P->ucoords), vals + N, I(P->uidx32), I(P->inv32), n, I(L->coords[0]),
                           (int64_t*)(A + P->perm), (int64_t*)(A + P->inv_perm), (int64_t*)(A + P->unique_index),
                           (int64_t*)(A + P->inverse), st);
    if (rc != PBN_OK) return rc;
    // 3. level 0 of the Z-ordered lineage: rows are unique already -> plain insert, value = row
    rc = coords_insert_identity(I(L->coords[0]), n_unique, n, (uint64_t*)(A + L->keys[0]), I(L->vals[0]), L->capacity[0],
                                I(L->coords[0]), I(L->counts), I(L->counts) + 8, st);
    if (rc != PBN_OK) return rc;
    // 4. levels 1..4 and every map
    return coords_build_upper(n, want_k5, x_fastest, arena, L, st);
}

extern "C" int pbn_coords_prepare(const int32_t* coords, int n, int want_k5, int x_fastest, void* arena, size_t arena_bytes,
                                  const pbn_prepare_layout* P, pbn_stream_t stream) {
    return coords_prepare_impl(coords, nullptr, n, want_k5, x_fastest, arena, arena_bytes, P, stream);
}

// capacity form: the input holds *n_dev (<= n_cap) rows, a count that stays on the device
extern "C" int pbn_coords_prepare_dev(const int32_t* coords, const int32_t* n_dev, int n_cap, int want_k5, int x_fastest,
                                      void* arena, size_t arena_bytes, const pbn_prepare_layout* P, pbn_stream_t stream) {
    if (!n_dev) return PBN_ERR_ARG;
    return coords_prepare_impl(coords, n_dev, n_cap, want_k5, x_fastest, arena, arena_bytes, P, stream);
}

// the round-2 pipeline behind the same layout (n_dev may be null): cross-check of pyramid.hip
extern "C" int pbn_coords_prepare_hash(const int32_t* coords, const int32_t* n_dev, int n_cap, int want_k5, int x_fastest,
                                       void* arena, size_t arena_bytes, const pbn_prepare_layout* P, pbn_stream_t stream) {
    return coords_prepare_impl(coords, n_dev, n_cap, want_k5, x_fastest, arena, arena_bytes, P, stream, true);
}

static inline int esize(int dtype) { return dtype == PBN_F32 ? 4 : 2; }

extern "C" size_t pbn_unet_arena_bytes(const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows, int dtype,
                                       int64_t* buf_offsets) {
    if (!bufs || !n_rows || n_bufs < 1) return 0;
    size_t off = 0;
    for (int b = 0; b < n_bufs; ++b) {
        if (b == 0) { if (buf_offsets) buf_offsets[0] = -1; continue; }  // buffer 0 is the caller's input slab
        const size_t bytes = (size_t)n_rows[bufs[b].level] * bufs[b].width * esize(dtype);
        if (buf_offsets) buf_offsets[b] = (int64_t)off;
        off = a256(off + (bytes ? bytes : 16));
    }
    return off;
}

// Rows expected per level for the NEXT forward of this thread whose row counts are capacities (pbn_unet_forward_dev): the values are
// COPIED (nothing of the caller's is kept) and consumed at the top of every entry point, whatever it returns.
static thread_local int g_unet_rows_hint[6] = {0, 0, 0, 0, 0, 0};      // [5] = armed
extern "C" void pbn_unet_set_rows_hint(const int32_t* rows) {
    for (int l = 0; l < 5; ++l) g_unet_rows_hint[l] = rows ? rows[l] : 0;
    g_unet_rows_hint[5] = rows ? 1 : 0;
}
struct RowsHint { int rows[5]; bool armed; };
static RowsHint take_rows_hint() {
    RowsHint h;
    for (int l = 0; l < 5; ++l) h.rows[l] = g_unet_rows_hint[l];
    h.armed = g_unet_rows_hint[5] != 0;
    g_unet_rows_hint[5] = 0;
    return h;
}

static int unet_forward_impl(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                             const int32_t* n_rows, const void* input, int ld_input, const int32_t* const* k3,
                             const int32_t* k5, const int32_t* const* down, const int32_t* const* up, void* arena,
                             size_t arena_bytes, int dtype, void* splitk_ws, size_t splitk_bytes, pbn_stream_t stream,
                             hipEvent_t* events, const int32_t* n_rows_dev = nullptr, const RowsHint* hint = nullptr) {
    if (!ops || !bufs || !n_rows || !input || !arena || n_ops < 1 || n_bufs < 2 || n_bufs > 512) return PBN_ERR_ARG;
    int64_t offs[512];
    if (pbn_unet_arena_bytes(bufs, n_bufs, n_rows, dtype, offs) > arena_bytes) return PBN_ERR_WORKSPACE;
    const int es = esize(dtype);
    char* A = (char*)arena;
    struct RsGuard { ~RsGuard() { g_rows_hint = 0; } } rs_guard;   // cleared on every return path
    auto base = [&](int b) -> char* { return b == 0 ? (char*)input : A + offs[b]; };
    auto ld = [&](int b) -> int { return b == 0 ? ld_input : bufs[b].width; };
    for (int i = 0; i < n_ops; ++i) {
        const pbn_unet_op& o = ops[i];
        if (o.in_buf < 0 || o.in_buf >= n_bufs || o.out_buf < 1 || o.out_buf >= n_bufs || o.res_buf >= n_bufs ||
            o.level_in < 0 || o.level_in > 4 || o.level_out < 0 || o.level_out > 4)
            return PBN_ERR_ARG;
        const int32_t* nbr = nullptr;
        int K = 1;
        switch (o.map_kind) {
            case 0: break;
            case 1: nbr = k3[o.level_out]; K = 27; break;
            case 2: nbr = k5; K = 125; break;
            case 3: nbr = down[o.level_in]; K = 8; break;   // level_in = fine level
            case 4: nbr = up[o.level_out]; K = 8; break;    // level_out = fine level
            default: return PBN_ERR_ARG;
        }
        if (o.map_kind != 0 && !nbr) return PBN_ERR_ARG;
        const void* in = base(o.in_buf) + (size_t)o.in_col * es;
        void* out = base(o.out_buf) + (size_t)o.out_col * es;
        const void* res = o.res_buf >= 0 ? base(o.res_buf) + (size_t)o.res_col * es : nullptr;
        // the NEXT op's packed weights are touched by this op's workgroups (spconv_common.h: prefetch_next_weights)
        g_next_weights = NextWeights{nullptr, 0, 0, 0, 0};
        static const int pf_env = getenv("PBN_CONV_PREFETCH") ? atoi(getenv("PBN_CONV_PREFETCH")) : 0;   // (off by default: nothing to describe)
        if (pf_env && i + 1 < n_ops) {
            const pbn_unet_op& q = ops[i + 1];
            if (q.in_buf >= 0 && q.in_buf < n_bufs && q.level_in >= 0 && q.level_in <= 4 && q.level_out >= 0 && q.level_out <= 4 && q.w) {
                ConvArgs nx;
                memset(&nx, 0, sizeof(nx));
                nx.K = q.map_kind == 0 ? 1 : (q.map_kind == 1 ? 27 : (q.map_kind == 2 ? 125 : 8));
                nx.vpo = q.vpo; nx.n_steps = q.n_steps; nx.ntiles_total = q.cout_p / 16; nx.n_out = nx.n_sel = n_rows[q.level_out];
                nx.w_bytes = (unsigned)((unsigned long long)q.n_steps * (q.cout_p / 16) * 1024ull);
                nx.in_bytes = (unsigned)((unsigned long long)n_rows[q.level_in] * ld(q.in_buf) * es);
                LaunchDesc d;
                describe_launch(nx, dtype, &d);
                g_next_weights = NextWeights{q.w, q.n_steps, q.cout_p / 16, d.wave_family ? d.nt : 0, d.wmajor ? d.groups : 0};
            }
        }
        if (events) PBN_HIP_CHECK(hipEventRecord(events[2 * i], (hipStream_t)stream));
        g_rows_hint = (hint && hint->armed && n_rows_dev) ? hint->rows[o.level_out] : 0;
        int rc = PBN_ERR_UNSUPPORTED;
        if (o.in2_buf >= 0) {           // a BasicBlock's 1x1 shortcut folded into this convolution's reduction
            if (o.in2_buf >= n_bufs) return PBN_ERR_ARG;
            const void* in2 = base(o.in2_buf) + (size_t)o.in2_col * es;
            rc = pbn_spconv_forward_dual(in, ld(o.in_buf), n_rows[o.level_in], nbr, K,
                                         n_rows_dev ? n_rows_dev + o.level_out : nullptr, n_rows[o.level_out], o.w, o.vpo,
                                         o.n_steps, o.cout_p, o.scale, o.shift, res, o.res_buf >= 0 ? ld(o.res_buf) : 0, o.relu,
                                         out, ld(o.out_buf), dtype, 0, splitk_ws, splitk_bytes, in2, ld(o.in2_buf),
                                         n_rows[o.level_out], o.vpo2, stream);
        } else
            rc = pbn_spconv_forward(in, ld(o.in_buf), n_rows[o.level_in], nbr, K, nullptr,
                                    n_rows_dev ? n_rows_dev + o.level_out : nullptr, n_rows[o.level_out], o.w, o.vpo,
                                    o.n_steps, o.cout_p, o.scale, o.shift, res, o.res_buf >= 0 ? ld(o.res_buf) : 0,
                                    o.relu, out, ld(o.out_buf), dtype, 0, splitk_ws, splitk_bytes, stream);
        g_next_weights = NextWeights{nullptr, 0, 0, 0, 0};
        if (rc != PBN_OK) return rc;
        if (events) PBN_HIP_CHECK(hipEventRecord(events[2 * i + 1], (hipStream_t)stream));
    }
    return PBN_OK;
}

extern "C" int pbn_unet_forward(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                                const int32_t* n_rows, const void* input, int ld_input, const int32_t* const* k3,
                                const int32_t* k5, const int32_t* const* down, const int32_t* const* up, void* arena,
                                size_t arena_bytes, int dtype, void* splitk_ws, size_t splitk_bytes, pbn_stream_t stream) {
    (void)take_rows_hint();
    return unet_forward_impl(ops, n_ops, bufs, n_bufs, n_rows, input, ld_input, k3, k5, down, up, arena, arena_bytes, dtype,
                             splitk_ws, splitk_bytes, stream, nullptr);
}

// capacity form: n_rows are capacities, the rows that exist are n_rows_dev[level] (device); launches are sized by the
// capacities and every kernel bounds itself by the device-side count
extern "C" int pbn_unet_forward_dev(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                                    const int32_t* n_rows_cap, const int32_t* n_rows_dev, const void* input, int ld_input,
                                    const int32_t* const* k3, const int32_t* k5, const int32_t* const* down,
                                    const int32_t* const* up, void* arena, size_t arena_bytes, int dtype, void* splitk_ws,
                                    size_t splitk_bytes, pbn_stream_t stream) {
    const RowsHint hint = take_rows_hint();           // consumed by THIS call whatever it returns
    if (!n_rows_dev) return PBN_ERR_ARG;
    return unet_forward_impl(ops, n_ops, bufs, n_bufs, n_rows_cap, input, ld_input, k3, k5, down, up, arena, arena_bytes, dtype,
                             splitk_ws, splitk_bytes, stream, nullptr, n_rows_dev, &hint);
}

// Measurement variant: brackets every op with HIP events on the launching stream, SYNCHRONISES the stream at the end and
// returns the per-op durations in milliseconds (host array op_ms[n_ops]).  Used by bench.py's roofline probe only.
extern "C" int pbn_unet_forward_timed(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs,
                                      const int32_t* n_rows, const void* input, int ld_input, const int32_t* const* k3,
                                      const int32_t* k5, const int32_t* const* down, const int32_t* const* up,
                                      void* arena, size_t arena_bytes, int dtype, void* splitk_ws, size_t splitk_bytes,
                                      pbn_stream_t stream, float* op_ms) {
    (void)take_rows_hint();
    if (!op_ms || n_ops < 1 || n_ops > 4096) return PBN_ERR_ARG;
    hipEvent_t* ev = new hipEvent_t[2 * (size_t)n_ops];
    int made = 0, rc = PBN_OK;
    for (; made < 2 * n_ops; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) { rc = PBN_ERR_HIP; break; }
    if (rc == PBN_OK)
        rc = unet_forward_impl(ops, n_ops, bufs, n_bufs, n_rows, input, ld_input, k3, k5, down, up, arena, arena_bytes,
                               dtype, splitk_ws, splitk_bytes, stream, ev);
    if (rc == PBN_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = PBN_ERR_HIP;
    if (rc == PBN_OK)
        for (int i = 0; i < n_ops; ++i)
            if (hipEventElapsedTime(&op_ms[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) { rc = PBN_ERR_HIP; break; }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}
