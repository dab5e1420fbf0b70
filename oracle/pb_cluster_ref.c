/*
 * oracle/pb_cluster_ref.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's point-wise binarization + neighbour clustering
 * ("binary_cluster"), step by step.  Citations are relative to /root/reference/.
 *
 *   driver (segment loop, global id accumulation, output resize)  lib/PB_lib/src/pbnet/cluster.cu:57-118
 *   sort_input_by_l1norm                                           lib/PB_lib/src/pbnet/binary.cu:49-69
 *   calc_num_neighbours / k_num_nbs                                binary.cu:71-90, binary_cuda_functions.cu:29-89
 *   calc_start_pos / append_neighbours / k_append_neighbours       binary.cu:92-128, binary_cuda_functions.cu:101-166
 *   identify_HPs / k_identify_HPs                                  binary.cu:130-152, binary_cuda_functions.cu:175-186
 *   identify_clusters / bfs_sem / k_bfs                            binary.cu:154-217, binary_cuda_functions.cu:197-215
 *   filter / shift_con_clt                                         binary.cu:219-268, binary_cuda_functions.cu:249-256
 *   assigned_LPs / noise_id_cluster                                binary.cu:270-358, binary_cuda_functions.cu:258-302
 *   get_clt_center / cal_mean                                      binary.cu:360-415, binary_cuda_functions.cu:217-246
 *   square_dist                                                    binary_cuda_functions.cu:305-308
 *
 * PARITY UNPINNED: the reference is CUDA-only (no CPU path), has no tests and no golden vectors, and cannot
 * be compiled or run in this environment (no nvcc, no NVIDIA GPU).  This file is pinned only against an
 * independent brute-force statement of the same specification (tests/bruteforce_cluster.py).
 *
 * Fixed floating-point form (the reference leaves it to nvcc's -fmad default, which is unobservable here):
 *   d2 = (dx*dx + dy*dy) + dz*dz   evaluated in IEEE binary32, one rounding per operation, NO fused
 *   multiply-add; the radius test is d2 <= r*r with r*r rounded to binary32.  Build with -ffp-contract=off.
 *
 * Restriction: radius[] and min_pts[] must be uniform over the 18 classes, which is what the only caller
 * guarantees (lib/PB_lib/torch_io/pbnet_ops.py:33-36).  The reference kernels index both tables with
 * sem[<position after the l1 sort>] - 2 against the UNSORTED sem array (binary_cuda_functions.cu:35,110 vs
 * binary.cu:67-68,82), so with non-uniform tables its result depends on thrust's unstable sort order and on
 * 32-lane warp composition; that behaviour is not restated.  sem must lie in [2,19] (sem-2 indexes 18-tables).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PBREF_OK 0
#define PBREF_ERR_NONUNIFORM 1
#define PBREF_ERR_SEM_RANGE 2
#define PBREF_ERR_ALLOC 3

/* binary_cuda_functions.cu:305-308, with the fp form fixed as stated in the header. */
static inline float square_dist(float x1, float y1, float z1, float x2, float y2, float z2) {
    float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
    float a = dx * dx;
    float b = dy * dy;
    float c = dz * dz;
    float ab = a + b;
    return ab + c;
}

/* binary.cu:229 -- "mean count from HAIS", indexed by sem-2. */
static const float k_mean_count[18] = {3917.0f, 12056.0f, 2303.0f, 8331.0f, 3948.0f, 3166.0f, 5629.0f, 11719.0f,
                                       1003.0f, 3317.0f,  4912.0f, 10221.0f, 3889.0f, 4136.0f, 2120.0f, 945.0f,
                                       3967.0f, 2589.0f};

typedef struct {
    float key;
    int idx;
} l1_item;

static int cmp_l1(const void* a, const void* b) {
    const l1_item* p = (const l1_item*)a;
    const l1_item* q = (const l1_item*)b;
    if (p->key < q->key) return -1;
    if (p->key > q->key) return 1;
    /* thrust::sort_by_key is unstable; any order of equal keys yields the same neighbour SETS. */
    return (p->idx > q->idx) - (p->idx < q->idx);
}

/* lower_bound / upper_bound over the sorted l1 array (binary_cuda_functions.cu:49-51). */
static int lower_bound_f(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static int upper_bound_f(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* One Solver life-cycle (cluster.cu:78-104) on one batch segment of n points.
 * cluster_ids (in/out, pre-filled -1 by the caller as pbnet_ops.py:40-41 does), den_queue (out).
 * Returns the new cluster_accum through *accum_io; appends centres / kept sems. */
static int solve_segment(const float* x, const float* y, const float* z, const float* xo, const float* yo,
                         const float* zo, const int* sem, int n, float radius, int min_pts, float para_f,
                         int nv_flag, int* cluster_ids, int* den_queue, int* accum_io, float* center,
                         int* n_center_io, int* clt_sem_out, int* n_clt_sem_io, int* cluster_num_out) {
    const int cluster_accum_old = *accum_io;
    int rc = PBREF_OK;
    l1_item* items = (l1_item*)malloc(sizeof(l1_item) * (size_t)n);
    float* sx = (float*)malloc(sizeof(float) * (size_t)n);
    float* sy = (float*)malloc(sizeof(float) * (size_t)n);
    float* sz = (float*)malloc(sizeof(float) * (size_t)n);
    float* sl1 = (float*)malloc(sizeof(float) * (size_t)n);
    int* mapper = (int*)malloc(sizeof(int) * (size_t)n);
    int* num_nbs = (int*)calloc((size_t)n, sizeof(int));
    long long* start_pos = (long long*)malloc(sizeof(long long) * ((size_t)n + 1));
    int* memberships = (int*)malloc(sizeof(int) * (size_t)n);
    unsigned char* visited = (unsigned char*)malloc((size_t)n);
    int* queue = (int*)malloc(sizeof(int) * (size_t)n);
    int* neighbours = NULL;
    if (!items || !sx || !sy || !sz || !sl1 || !mapper || !num_nbs || !start_pos || !memberships || !visited ||
        !queue) {
        rc = PBREF_ERR_ALLOC;
        goto done;
    }

    /* ---- sort_input_by_l1norm (binary.cu:49-69); l1 = |x|+|y|+|z| in fp32 (pbnet_ops.py:19). */
    for (int i = 0; i < n; ++i) {
        float ax = fabsf(x[i]), ay = fabsf(y[i]), az = fabsf(z[i]);
        float s = ax + ay;
        items[i].key = s + az;
        items[i].idx = i; /* index_mapper = arange(n) per segment, pbnet_ops.py:21-25 */
    }
    qsort(items, (size_t)n, sizeof(l1_item), cmp_l1);
    for (int p = 0; p < n; ++p) {
        int i = items[p].idx;
        sx[p] = x[i]; sy[p] = y[i]; sz[p] = z[i];
        sl1[p] = items[p].key;
        mapper[p] = i;
    }

    /* ---- calc_num_neighbours / k_num_nbs (binary_cuda_functions.cu:29-89).
     * The l1 window [l1 - 2r, l1 + 2r] is a superset of the r-ball (|l1(p)-l1(q)| <= sqrt(3)*|p-q|), so the
     * count equals the plain ball count; the reference takes the window per 512-thread block, which only
     * widens it.  num_nbs excludes self by the trailing "- 1" (:88). */
    const float r2 = radius * radius;
    const float two_r = 2 * radius;
    for (int p = 0; p < n; ++p) {
        int lo = lower_bound_f(sl1, n, sl1[p] - two_r);
        int hi = upper_bound_f(sl1, n, sl1[p] + two_r);
        int ans = 0;
        for (int q = lo; q < hi; ++q) ans += square_dist(sx[p], sy[p], sz[p], sx[q], sy[q], sz[q]) <= r2;
        num_nbs[mapper[p]] = ans - 1;
    }
    /* ---- calc_start_pos (binary.cu:92-103): exclusive scan, original index order. */
    start_pos[0] = 0;
    for (int i = 0; i < n; ++i) start_pos[i + 1] = start_pos[i] + (num_nbs[i] > 0 ? num_nbs[i] : 0);
    {
        long long total = start_pos[n];
        neighbours = (int*)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
        if (!neighbours) { rc = PBREF_ERR_ALLOC; goto done; }
    }
    /* ---- append_neighbours / k_append_neighbours (binary_cuda_functions.cu:101-166): CSR adjacency,
     * self excluded by sorted position (:159), original indices stored (:161). */
    for (int p = 0; p < n; ++p) {
        int lo = lower_bound_f(sl1, n, sl1[p] - two_r);
        int hi = upper_bound_f(sl1, n, sl1[p] + two_r);
        long long upos = start_pos[mapper[p]];
        for (int q = lo; q < hi; ++q)
            if (q != p && square_dist(sx[p], sy[p], sz[p], sx[q], sy[q], sz[q]) <= r2) neighbours[upos++] = mapper[q];
    }
    /* ---- identify_HPs (binary_cuda_functions.cu:175-186): 0 = HP (high density), 2 = LP.
     * den_queue = num_nbs (binary.cu:148). */
    for (int i = 0; i < n; ++i) {
        memberships[i] = (num_nbs[i] >= min_pts) ? 0 : 2;
        den_queue[i] = num_nbs[i];
    }
    /* ---- identify_clusters + bfs_sem (binary.cu:154-217).  Seeds in original index order; the BFS
     * expands through HPs of ANY class (k_bfs :209 tests membership only); every visited vertex of the
     * seed's class is (re)labelled, so a border LP ends with the id of the LAST component that reached it. */
    int cluster = cluster_accum_old;
    for (int u = 0; u < n; ++u) {
        if (!(cluster_ids[u] == -1 && memberships[u] == 0)) continue;
        memset(visited, 0, (size_t)n);
        int head = 0, tail = 0;
        queue[tail++] = u;
        visited[u] = 1;
        while (head < tail) {
            int v = queue[head++];
            if (memberships[v] != 0) continue; /* LPs are visited but not expanded */
            for (long long e = start_pos[v]; e < start_pos[v + 1]; ++e) {
                int w = neighbours[e];
                if (!visited[w]) { visited[w] = 1; queue[tail++] = w; }
            }
        }
        const int sem_cur = sem[u];
        for (int m = 0; m < n; ++m) {
            if (visited[m] && sem[m] == sem_cur) {
                cluster_ids[m] = cluster;
                if (memberships[m] != 0) memberships[m] = 1;
            }
        }
        ++cluster;
    }
    /* ---- filter (binary.cu:219-268). */
    int num_cluster = cluster - cluster_accum_old;
    {
        int* clt_semv = (int*)calloc((size_t)(num_cluster > 0 ? num_cluster : 1), sizeof(int));
        int* clt_num = (int*)calloc((size_t)(num_cluster > 0 ? num_cluster : 1), sizeof(int));
        if (!clt_semv || !clt_num) { free(clt_semv); free(clt_num); rc = PBREF_ERR_ALLOC; goto done; }
        for (int i = 0; i < n; ++i) {
            int c = cluster_ids[i];
            if (c != -1) { clt_num[c - cluster_accum_old]++; clt_semv[c - cluster_accum_old] = sem[i]; }
        }
        int reduce_count = 0;
        for (int i = 0; i < num_cluster; ++i) {
            int cur_sem = clt_semv[i] - 2;
            clt_sem_out[(*n_clt_sem_io)++] = clt_semv[i];
            int reduce_clt_idx = i + cluster_accum_old - reduce_count;
            float cur_mean_count = k_mean_count[cur_sem] * para_f;
            if ((float)clt_num[i] < cur_mean_count) {
                /* shift_con_clt (binary_cuda_functions.cu:249-256) */
                for (int m = 0; m < n; ++m) {
                    if (cluster_ids[m] == reduce_clt_idx) cluster_ids[m] = -1;
                    if (cluster_ids[m] > reduce_clt_idx) cluster_ids[m]--;
                }
                reduce_count++;
                (*n_clt_sem_io)--;
            }
        }
        free(clt_semv);
        free(clt_num);
        num_cluster -= reduce_count;
    }
    int cluster_accum = cluster_accum_old + num_cluster;

    /* ---- assigned_LPs + noise_id_cluster (binary.cu:270-358, binary_cuda_functions.cu:258-302).
     * ORIGINAL coordinates; nearest assigned point of the same class; `<=` while scanning assigned points in
     * ascending index keeps the LAST (highest-index) minimum.  No same-class assigned point: the second loop
     * (:287-299) re-tests one fixed point, the last entry of clt_idx, i.e. the highest-index assigned point.
     * No assigned point at all: min_index stays 0 and the id of point 0 (-1) is copied. */
    if (nv_flag) {
        int noise_num = 0;
        for (int j = 0; j < n; ++j) noise_num += (cluster_ids[j] == -1);
        if (noise_num != 0) {
            int un_noise_num = n - noise_num;
            int* noise_idx = (int*)malloc(sizeof(int) * (size_t)noise_num);
            int* clt_idx = (int*)malloc(sizeof(int) * (size_t)(un_noise_num > 0 ? un_noise_num : 1));
            int* new_ids = (int*)malloc(sizeof(int) * (size_t)noise_num);
            if (!noise_idx || !clt_idx || !new_ids) { free(noise_idx); free(clt_idx); free(new_ids); rc = PBREF_ERR_ALLOC; goto done; }
            int nc = 0, uc = 0;
            for (int k = 0; k < n; ++k) {
                if (cluster_ids[k] != -1) clt_idx[uc++] = k; else noise_idx[nc++] = k;
            }
            for (int u = 0; u < noise_num; ++u) {
                int nr = noise_idx[u];
                float nx = xo[nr], ny = yo[nr], nz = zo[nr];
                int sem_noise = sem[nr];
                float min_dist = 0.0f;
                int min_index = 0, clt_real_idx = 0, count_i = 0;
                for (int i = 0; i < un_noise_num; ++i) {
                    clt_real_idx = clt_idx[i];
                    if (sem[clt_real_idx] != sem_noise) continue;
                    float dist = square_dist(nx, ny, nz, xo[clt_real_idx], yo[clt_real_idx], zo[clt_real_idx]);
                    if (count_i == 0) min_dist = dist;
                    count_i++;
                    if (dist <= min_dist) { min_dist = dist; min_index = clt_real_idx; }
                }
                if (count_i == 0) {
                    for (int i = 0; i < un_noise_num; ++i) {
                        float dist = square_dist(nx, ny, nz, xo[clt_real_idx], yo[clt_real_idx], zo[clt_real_idx]);
                        if (count_i == 0) min_dist = dist;
                        count_i++;
                        if (dist <= min_dist) { min_dist = dist; min_index = clt_real_idx; }
                    }
                }
                new_ids[u] = cluster_ids[min_index]; /* min_index is never a noise point unless nothing is assigned */
            }
            for (int u = 0; u < noise_num; ++u) cluster_ids[noise_idx[u]] = new_ids[u];
            free(noise_idx);
            free(clt_idx);
            free(new_ids);
        }
    }
    *cluster_num_out = cluster_accum - cluster_accum_old;
    /* ---- get_clt_center + cal_mean (binary.cu:360-415, binary_cuda_functions.cu:217-246): running mean of the
     * SHIFTED coordinates, index order, fp32, IEEE division; skipped when the segment kept no cluster
     * (cluster.cu:98-103). */
    if (*cluster_num_out != 0) {
        for (int c = 0; c < *cluster_num_out; ++c) {
            int cluster_cur = cluster_accum_old + c;
            int N = 0;
            float Mx = 0, My = 0, Mz = 0;
            for (int i = 0; i < n; ++i) {
                if (cluster_ids[i] == cluster_cur) {
                    N++;
                    Mx = Mx + (x[i] - Mx) / N;
                    My = My + (y[i] - My) / N;
                    Mz = Mz + (z[i] - Mz) / N;
                }
            }
            center[(*n_center_io)++] = Mx;
            center[(*n_center_io)++] = My;
            center[(*n_center_io)++] = Mz;
        }
    }
    *accum_io = cluster_accum;
done:
    free(items); free(sx); free(sy); free(sz); free(sl1); free(mapper); free(num_nbs); free(start_pos);
    free(memberships); free(visited); free(queue); free(neighbours);
    return rc;
}

/* Signature follows PB_lib.binary_cluster (lib/PB_lib/src/pbnet/cluster.h:13-18) with the tensors as raw
 * pointers; l1_norm and index_mapper are derived (they are functions of x,y,z and the segment lengths).
 * center must hold 3*n floats, clt_sem n ints; *n_clusters_out receives the total number of kept clusters,
 * i.e. center is valid for 3*(*n_clusters_out) floats (the reference resizes the tensors, cluster.cu:112-118). */
int pbref_binary_cluster(const float* x, const float* y, const float* z, const float* xo, const float* yo,
                         const float* zo, const int* sem, const int* batch_ind, const float* radius18,
                         const int* min_pts18, int* cluster_idx, int* cluster_num, int* den_queue, float* center,
                         int* clt_sem, int batch_size, float para_f, int nv_flag, int* n_clusters_out) {
    for (int i = 1; i < 18; ++i)
        if (radius18[i] != radius18[0] || min_pts18[i] != min_pts18[0]) return PBREF_ERR_NONUNIFORM;
    int batch_start = 0, cluster_accum = 0, n_center = 0, n_clt_sem = 0;
    for (int b = 0; b < batch_size; ++b) {
        int len = batch_ind[b];
        if (len == 0) continue; /* cluster.cu:59-61; cluster_num[b] keeps its initial 0 */
        for (int i = 0; i < len; ++i) {
            int s = sem[batch_start + i];
            if (s < 2 || s > 19) return PBREF_ERR_SEM_RANGE;
        }
        int rc = solve_segment(x + batch_start, y + batch_start, z + batch_start, xo + batch_start,
                               yo + batch_start, zo + batch_start, sem + batch_start, len, radius18[0],
                               min_pts18[0], para_f, nv_flag, cluster_idx + batch_start, den_queue + batch_start,
                               &cluster_accum, center, &n_center, clt_sem, &n_clt_sem, &cluster_num[b]);
        if (rc != PBREF_OK) return rc;
        batch_start += len;
    }
    *n_clusters_out = n_center / 3;
    return PBREF_OK;
}

/* get_iou (lib/PB_lib/src/iou/get_iou.cu:12-29): the 1e-5 literal is a double, so the denominator is
 * evaluated in double and the quotient rounded to float (:26). */
void pbref_get_iou(int nInstance, int nProposal, const int* proposals_idx, const int* proposals_offset,
                   const long long* instance_labels, const int* instance_pointnum, float* proposals_iou) {
    for (int p = 0; p < nProposal; ++p) {
        int start = proposals_offset[p], end = proposals_offset[p + 1];
        int proposal_total = end - start;
        for (int ins = 0; ins < nInstance; ++ins) {
            int instance_total = instance_pointnum[ins];
            int intersection = 0;
            for (int i = start; i < end; ++i)
                if ((int)instance_labels[proposals_idx[i]] == ins) intersection += 1;
            proposals_iou[(size_t)p * nInstance + ins] =
                (float)((float)intersection / ((float)(proposal_total + instance_total - intersection) + 1e-5));
        }
    }
}
