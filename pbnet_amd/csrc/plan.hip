// plan.hip -- the control flow of PBNet.forward between its big kernels, kept ON THE DEVICE (capacity-planned inference):
// the reference decides these things on the host, with one device->host copy each:
//   class gate          network/PBNet.py:151-160   which classes have enough points to be grouped
//   local-scene plan    network/PBNet.py:182-234   per cluster: itself (+ its k nearest clusters when it is large)
//   proposal offsets    network/PBNet.py:330-345   rows kept per local scene -> offsets, surviving scenes renumbered
// Here each is one small launch that reads its sizes from device memory and writes counts for its consumers, so the whole
// forward is a fixed launch sequence over capacity-sized buffers (graph-capturable, no host synchronisation).
// counts layout (int32[PBN_CNT_WORDS]): see include/pbnet_hip.h.
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int MAX_K = 6;          // network/PBNet.py:35 K_max
constexpr int SEG_CLUSTERS = 2048; // clusters of one (class, batch) segment the kNN stage keeps in LDS

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- class gate (one wave) --------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_class_gate(const int* __restrict__ table, const float* __restrict__ thr05, int n_cls,
                                                  int nb, int m_cap, int n_points, int* __restrict__ class_base,
                                                  int* __restrict__ seg_len, int* __restrict__ counts) {
    const int lane = threadIdx.x;
    int tot = 0;
    if (lane < n_cls)
        for (int b = 0; b < nb; ++b) tot += table[lane * nb + b];
    const bool keep = lane >= 2 && lane < n_cls && !((float)tot < thr05[lane]);      // PBNet.py:156
    const int kept = keep ? tot : 0;
    const int incl = wave_incl_scan_i(kept, lane);
    const int m = __shfl(incl, 63, 64);
    int all = tot;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) all += __shfl_xor(all, o, 64);
    int flags = 0;
    if (all != n_points) flags |= PBN_OVF_BATCH;      // a batch index outside [0, nb): PBNet.py:286 asserts
    if (m > m_cap) flags |= PBN_OVF_POINTS;
    const bool drop_all = flags != 0;
    if (lane < n_cls) class_base[lane] = (keep && !drop_all) ? incl - kept : -1;
    if (lane >= 2 && lane < n_cls)
        for (int b = 0; b < nb; ++b) seg_len[(lane - 2) * nb + b] = (keep && !drop_all) ? table[lane * nb + b] : 0;
    if (lane == 0) {
        counts[PBN_CNT_POINTS] = drop_all ? 0 : m;
        if (flags) atomicOr(&counts[PBN_CNT_OVERFLOW], flags);
    }
}

// ---- local-scene plan, stage 1: one wave per cluster = per local scene (task 'test': every cluster is a scene) ------
// scene c: entry 0 = cluster c with weight 1; if size(c) > count_mean * 0.2 and the segment has other clusters, the
// para_k = min(C_b - 1, K_max) nearest clusters of the SAME (class, batch) segment follow with weights
// 0.5 * (para_k + 1 - i) / (para_k + 1)  (PBNet.py:196-221).  Nearest = ascending (d^2, cluster id), d^2 = (dx^2+dy^2)+dz^2 in
// unfused fp32 -- the reference sorts torch.cdist distances, the same order up to ties of the rounded distances; segments of more
// than 25 clusters raise PBN_OVF_CDIST (torch.cdist's matrix-multiply path: the caller falls back to the reference's own call).
__global__ __launch_bounds__(64) void k_plan_scenes(const int* __restrict__ cluster_num, int n_seg, int nb,
                                                   const int* __restrict__ member_start, const float* __restrict__ centers,
                                                   const float* __restrict__ thr02, const int* __restrict__ kmax,
                                                   const int* __restrict__ n_clusters, int c_cap,
                                                   int* __restrict__ scene_n_ent, int* __restrict__ scene_ent,
                                                   float* __restrict__ scene_w, int* __restrict__ counts) {
    __shared__ float s_d2[SEG_CLUSTERS];
    const int lane = threadIdx.x;
    const int C = min(*n_clusters, c_cap);
    const int gid = blockIdx.x;
    if (gid >= C) return;
    // segment of this cluster: prefix over cluster_num
    int seg = -1, g0 = 0, cb = 0;
    {
        int carry = 0;
        for (int base = 0; base < n_seg && seg < 0; base += 64) {
            const int i = base + lane;
            const int v = i < n_seg ? cluster_num[i] : 0;
            const int incl = wave_incl_scan_i(v, lane);
            const int lo = carry + incl - v, hi = carry + incl;
            const unsigned long long hit = __ballot(i < n_seg && gid >= lo && gid < hi);
            if (hit) {
                const int src = __ffsll((long long)hit) - 1;
                seg = base + src;
                g0 = __shfl(lo, src, 64);
                cb = __shfl(v, src, 64);
            }
            carry += __shfl(incl, 63, 64);
        }
    }
    if (seg < 0) return;   // inconsistent table: leaves n_ent = 0 for this scene
    const int cls = 2 + seg / nb;
    const int para_k = min(min(cb - 1, kmax[cls]), MAX_K);
    const int size = member_start[gid + 1] - member_start[gid];
    const bool big = (float)size > thr02[cls] && para_k > 0;     // PBNet.py:209
    int* ent = scene_ent + (size_t)gid * (MAX_K + 1);
    float* wts = scene_w + (size_t)gid * (MAX_K + 1);
    if (lane == 0) { ent[0] = gid; wts[0] = 1.0f; }
    if (!big) {
        if (lane == 0) scene_n_ent[gid] = 1;
        return;
    }
    if (cb > SEG_CLUSTERS) {
        if (lane == 0) { scene_n_ent[gid] = 1; atomicOr(&counts[PBN_CNT_OVERFLOW], PBN_OVF_SEGMENT); }
        return;
    }
    // more than 25 clusters in the segment: torch.cdist (the reference's call, PBNet.py:201) computes these distances through a
    // matrix multiply, not as direct differences -- nearly equidistant clusters can rank differently.  The plan below is still
    // written (same order unless two rounded distances tie), the flag lets the caller take the reference's own call instead
    if (cb > 25 && lane == 0) atomicOr(&counts[PBN_CNT_OVERFLOW], PBN_OVF_CDIST);
    const float cx = centers[3 * gid + 0], cy = centers[3 * gid + 1], cz = centers[3 * gid + 2];
    for (int j = lane; j < cb; j += 64) {
        const int o = g0 + j;
        const float dx = __fsub_rn(centers[3 * o + 0], cx), dy = __fsub_rn(centers[3 * o + 1], cy),
                    dz = __fsub_rn(centers[3 * o + 2], cz);
        s_d2[j] = (o == gid) ? __builtin_inff() : __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    }
    __syncthreads();
    for (int k = 0; k < para_k; ++k) {
        float best = __builtin_inff();
        int bj = 0x7fffffff;
        for (int j = lane; j < cb; j += 64) {
            const float d = s_d2[j];
            if (d < best) { best = d; bj = j; }    // ascending j per lane: ties keep the lower index
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (lane == 0) {
            ent[1 + k] = g0 + bj;
            wts[1 + k] = (float)(0.5 * (double)((para_k + 1) - k) / (double)(para_k + 1));
            s_d2[bj] = __builtin_inff();
        }
        __syncthreads();
    }
    if (lane == 0) scene_n_ent[gid] = 1 + para_k;
}

// ---- local-scene plan, stage 2 (one workgroup): entries in scene order, row offsets, totals -------------------------------
constexpr int PACK_TPB = 256;
__device__ __forceinline__ int block_excl_scan256(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int incl = wave_incl_scan_i(v, lane);
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PACK_TPB / 64; ++w) {
        const int t = s_w[w];
        if (w < wid) off += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return off + incl - v;
}

__global__ __launch_bounds__(PACK_TPB) void k_plan_pack(const int* __restrict__ scene_n_ent, const int* __restrict__ scene_ent,
                                                       const float* __restrict__ scene_w, const int* __restrict__ member_start,
                                                       const int* __restrict__ n_clusters, int c_cap, int e_cap, int r_cap,
                                                       int* __restrict__ scene_base, int* __restrict__ ent_row_start,
                                                       int* __restrict__ ent_member_start, int* __restrict__ ent_scene,
                                                       float* __restrict__ ent_weight, int* __restrict__ counts) {
    __shared__ int s_w[PACK_TPB / 64];
    __shared__ int s_flag;
    const int C = min(*n_clusters, c_cap);
    if (threadIdx.x == 0) s_flag = (*n_clusters > c_cap) ? PBN_OVF_CLUSTERS : 0;
    // pass 1: first entry of every scene
    int carry = 0;
    for (int base = 0; base < C; base += PACK_TPB) {
        const int c = base + threadIdx.x;
        const int v = c < C ? scene_n_ent[c] : 0;
        int tot;
        const int ex = block_excl_scan256(v, s_w, tot);
        if (c < C) scene_base[c] = carry + ex;
        carry += tot;
    }
    const int n_ent = carry;
    __syncthreads();
    if (n_ent > e_cap) {
        if (threadIdx.x == 0) {
            atomicOr(&counts[PBN_CNT_OVERFLOW], PBN_OVF_ENTRIES | s_flag);
            counts[PBN_CNT_ENTRIES] = 0; counts[PBN_CNT_ROWS] = 0; counts[PBN_CNT_SCENES] = 0;
            ent_row_start[0] = 0;
        }
        return;
    }
    // pass 2: entries (scene order), their member ranges and sizes
    for (int c = threadIdx.x; c < C; c += PACK_TPB) {
        const int b = scene_base[c], ne = scene_n_ent[c];
        for (int j = 0; j < ne; ++j) {
            const int cl = scene_ent[(size_t)c * (MAX_K + 1) + j];
            ent_member_start[b + j] = member_start[cl];
            ent_scene[b + j] = c;
            ent_weight[b + j] = scene_w[(size_t)c * (MAX_K + 1) + j];
            ent_row_start[b + j + 1] = member_start[cl + 1] - member_start[cl];   // size, turned into an offset below
        }
    }
    __syncthreads();
    // pass 3: exclusive scan of the sizes in place (ent_row_start[e + 1] holds size(e))
    carry = 0;
    for (int base = 0; base < n_ent; base += PACK_TPB) {
        const int e = base + threadIdx.x;
        const int v = e < n_ent ? ent_row_start[e + 1] : 0;
        int tot;
        const int ex = block_excl_scan256(v, s_w, tot);
        if (e < n_ent) ent_row_start[e + 1] = carry + ex + v;     // inclusive end of entry e = start of entry e + 1
        carry += tot;
    }
    if (threadIdx.x == 0) {
        ent_row_start[0] = 0;
        int flags = s_flag;
        const bool ovf = carry > r_cap;
        if (ovf) flags |= PBN_OVF_ROWS;
        counts[PBN_CNT_ENTRIES] = ovf ? 0 : n_ent;
        counts[PBN_CNT_ROWS] = ovf ? 0 : carry;
        counts[PBN_CNT_SCENES] = ovf ? 0 : C;
        counts[PBN_CNT_CLUSTERS] = C;
        if (flags) atomicOr(&counts[PBN_CNT_OVERFLOW], flags);
    }
}

// ---- proposal offsets (one workgroup): PBNet.py:330-345 -----------------------------------------------------------------
__global__ __launch_bounds__(PACK_TPB) void k_proposal_offsets(const int* __restrict__ per_scene, int s_cap,
                                                              long long* __restrict__ proposals_offset,
                                                              long long* __restrict__ alive_ids, int* __restrict__ dense_of,
                                                              int* __restrict__ counts) {
    __shared__ int s_w[PACK_TPB / 64];
    const int S = min(counts[PBN_CNT_SCENES], s_cap);
    int carry_a = 0, carry_r = 0;
    for (int base = 0; base < S; base += PACK_TPB) {
        const int s = base + threadIdx.x;
        const int rows = s < S ? per_scene[s] : 0;
        const int alive = rows > 0 ? 1 : 0;
        int tot_a, tot_r;
        const int ex_a = block_excl_scan256(alive, s_w, tot_a);
        const int ex_r = block_excl_scan256(rows, s_w, tot_r);
        if (s < S) {
            dense_of[s] = carry_a + ex_a + alive - 1;                 // cumsum(alive) - 1 (PBNet.py:342-345)
            if (alive) {
                alive_ids[carry_a + ex_a] = s;
                proposals_offset[carry_a + ex_a] = carry_r + ex_r;
            }
        }
        carry_a += tot_a;
        carry_r += tot_r;
    }
    if (threadIdx.x == 0) {
        proposals_offset[carry_a] = carry_r;
        counts[PBN_CNT_PROPOSALS] = carry_a;
        counts[PBN_CNT_PROPOSAL_ROWS] = carry_r;
    }
}

// ---- first row of every batch index in a coordinate list sorted by batch (score-branch pooling segments) ---------------
__global__ __launch_bounds__(256) void k_batch_starts(const int* __restrict__ coords, const int* __restrict__ n_dev, int n_cap,
                                                     int n_seg, int* __restrict__ seg_start) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s > n_seg) return;
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    int lo = 0, hi = n;                                    // first row with batch >= s
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (coords[4 * (size_t)mid] < s) lo = mid + 1; else hi = mid;
    }
    seg_start[s] = lo;
}

}  // namespace
}  // namespace pbn

using namespace pbn;

extern "C" int pbn_class_gate(const int32_t* table, const float* thr05, int n_classes, int nb, int m_cap, int n_points,
                              int32_t* class_base, int32_t* seg_len, int32_t* counts, pbn_stream_t stream) {
    if (!table || !thr05 || !class_base || !seg_len || !counts || n_classes < 3 || n_classes > 64 || nb < 1 || m_cap < 0)
        return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_class_gate, dim3(1), dim3(64), 0, (hipStream_t)stream, table, thr05, n_classes, nb, m_cap, n_points,
                       class_base, seg_len, counts);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" size_t pbn_local_plan_workspace_bytes(int c_cap) {
    if (c_cap < 0) return 0;
    const size_t c = (size_t)(c_cap > 0 ? c_cap : 1);
    return align_up(c * 4, 256) * 2 + align_up(c * (MAX_K + 1) * 4, 256) * 2;
}

extern "C" int pbn_local_plan(const int32_t* cluster_num, int n_segments, int nb, const int32_t* member_start,
                              const float* centers, const int32_t* n_clusters, const float* thr02, const int32_t* kmax,
                              int c_cap, int e_cap, int r_cap, int32_t* ent_row_start, int32_t* ent_member_start,
                              int32_t* ent_scene, float* ent_weight, int32_t* counts, void* workspace,
                              size_t workspace_bytes, pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!cluster_num || !member_start || !centers || !n_clusters || !thr02 || !kmax || !ent_row_start || !ent_member_start ||
        !ent_scene || !ent_weight || !counts || !workspace || n_segments < 1 || nb < 1 || c_cap < 1 || e_cap < 1 || r_cap < 1)
        return PBN_ERR_ARG;
    if (workspace_bytes < pbn_local_plan_workspace_bytes(c_cap)) return PBN_ERR_WORKSPACE;
    Carver cv(workspace, workspace_bytes);
    int* scene_n_ent = cv.take<int>(c_cap);
    int* scene_base = cv.take<int>(c_cap);
    int* scene_ent = cv.take<int>((size_t)c_cap * (MAX_K + 1));
    float* scene_w = cv.take<float>((size_t)c_cap * (MAX_K + 1));
    { const int frc_ = fill_bytes(scene_n_ent, 0, sizeof(int) * (size_t)c_cap, stream); if (frc_ != PBN_OK) return frc_; }
    hipLaunchKernelGGL(k_plan_scenes, dim3(c_cap), dim3(64), 0, stream, cluster_num, n_segments, nb, member_start, centers,
                       thr02, kmax, n_clusters, c_cap, scene_n_ent, scene_ent, scene_w, counts);
    hipLaunchKernelGGL(k_plan_pack, dim3(1), dim3(PACK_TPB), 0, stream, scene_n_ent, scene_ent, scene_w, member_start,
                       n_clusters, c_cap, e_cap, r_cap, scene_base, ent_row_start, ent_member_start, ent_scene, ent_weight,
                       counts);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_proposal_offsets(const int32_t* per_scene, int s_cap, int64_t* proposals_offset, int64_t* alive_ids,
                                    int32_t* dense_of, int32_t* counts, pbn_stream_t stream) {
    if (!per_scene || !proposals_offset || !alive_ids || !dense_of || !counts || s_cap < 1) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_proposal_offsets, dim3(1), dim3(PACK_TPB), 0, (hipStream_t)stream, per_scene, s_cap,
                       (long long*)proposals_offset, (long long*)alive_ids, dense_of, counts);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_batch_starts(const int32_t* coords, const int32_t* n_dev, int n_cap, int n_segments, int32_t* seg_start,
                                pbn_stream_t stream) {
    if (!coords || !seg_start || n_cap < 0 || n_segments < 0) return PBN_ERR_ARG;
    hipLaunchKernelGGL(k_batch_starts, dim3(cdiv(n_segments + 1, 256)), dim3(256), 0, (hipStream_t)stream, coords, n_dev, n_cap,
                       n_segments, seg_start);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
