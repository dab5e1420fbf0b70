/*
 * include/pbnet_hip.h -- C ABI of libpbnet_hip.so, the MI355X (gfx950) hot path of PBNet.
 *
 * Every entry point takes raw DEVICE pointers, sizes and a hipStream_t; nothing here knows about torch.
 * All functions return PBN_OK (0) or a negative error code and never call exit() (the reference does:
 * lib/PB_lib/src/pbnet/binary.cuh:22-27).  No entry point allocates or frees device memory or synchronises the
 * stream unless stated: scratch comes from the caller (`workspace`), sized by the matching *_workspace_bytes().
 * Citations are relative to /root/reference/.
 */
#ifndef PBNET_HIP_H
#define PBNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pbn_stream_t; /* hipStream_t */

enum {
    PBN_OK = 0,
    PBN_ERR_ARG = -1,        /* bad argument (null pointer, negative size, unsupported shape) */
    PBN_ERR_WORKSPACE = -2,  /* workspace too small */
    PBN_ERR_HIP = -3,        /* a HIP runtime call failed; see pbn_last_hip_error() */
    PBN_ERR_RANGE = -4,      /* a value is outside the supported range (coordinates, class ids) */
    PBN_ERR_UNSUPPORTED = -5
};

/* element types of feature slabs */
enum { PBN_F32 = 0, PBN_BF16 = 1, PBN_F16 = 2 };

const char* pbn_version(void);
/* hipError_t of the last failing HIP call on this thread, as int (0 = none). */
int pbn_last_hip_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Grouping: point-wise binarization + neighbour clustering.
 * Replaces PB_lib.binary_cluster (lib/PB_lib/src/PB_lib_api.cpp:7, lib/PB_lib/src/pbnet/cluster.h:13-18,
 * cluster.cu:16-119) and the Solver pipeline behind it (binary.cu:19-415, binary_cuda_functions.cu:29-308).
 *
 *  off_xyz   [n,3] f32  offset-shifted coordinates (x,y,z of pbnet_ops.py:16-18 as one AoS slab)
 *  org_xyz   [n,3] f32  original coordinates (xo,yo,zo of pbnet_ops.py:27-29)
 *  sem       [n]   i32  predicted class per point, each in [2,19]
 *  seg_len   [n_seg] i32 points per batch segment (ins_bp of PBNet.py:172; "batch_index" of cluster.h:16);
 *                       segments are contiguous slices in order, empty segments allowed
 *  radius, min_pts      scalars: the reference broadcasts one value to all 18 classes (pbnet_ops.py:33-36)
 *  para_f, nv_flag      pbnet_ops.py:70-71 (0.05, true)
 *  general_sem          0: every segment holds ONE class (how PBNet.forward calls it, PBNet.py:154,176);
 *                       1: classes may be mixed inside a segment ("component x class" rule, cluster.cu/binary.cu:206)
 *  cluster_id [n]  i32 out  global ids across segments (cluster.cu:91-93,108), -1 = unassigned
 *  cluster_num[n_seg] i32 out
 *  den        [n]  i32 out  neighbour count EXCLUDING self (binary.cu:148); the Python op returns den+1
 *  centers    [3*n] f32 out capacity; first 3*C valid (cluster.cu:112-114 resizes instead)
 *  clt_sem    [n]  i32 out  capacity; first C valid (cluster.cu:116-118)
 *  n_clusters [1]  i32 out  C, in DEVICE memory (no host sync inside)
 *  member_start [n+1] i32 out, optional (NULL to skip): CSR offsets of the members of each final cluster
 *  member_idx   [n]   i32 out, optional: point indices grouped by final cluster id, ascending index inside a
 *                       cluster (= torch.nonzero(cluster_id == c) of PBNet.py:204); unassigned points are absent
 *
 * Arithmetic contract (identical to oracle/pb_cluster_ref.c): d2 = (dx*dx + dy*dy) + dz*dz in unfused binary32,
 * test d2 <= r*r; centre = sequential running mean M += (p-M)/N in index order with IEEE division.
 * Results are bit-exact and independent of scheduling.
 */
size_t pbn_cluster_workspace_bytes(int n_points, int n_segments, int general_sem);

int pbn_binary_cluster(const float* off_xyz, const float* org_xyz, const int32_t* sem, const int32_t* seg_len,
                       int n_points, int n_segments, float radius, int min_pts, float para_f, int nv_flag,
                       int general_sem /* bit 0: general (mixed-class) rule; bit 1: n_points is a CAPACITY -- the points that
                       exist are the first sum(seg_len) rows, a count that stays on the device (capacity-planned forward:
                       no host read-back between class selection and grouping) */, int32_t* cluster_id, int32_t* cluster_num, int32_t* den, float* centers,
                       int32_t* clt_sem, int32_t* n_clusters, int32_t* member_start, int32_t* member_idx,
                       void* workspace, size_t workspace_bytes, pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Proposal x instance IoU.  Replaces PB_lib.get_iou (lib/PB_lib/src/iou/get_iou.h:15, get_iou.cu:12-38).
 *  proposals_idx [S] i32, proposals_offset [P+1] i32, instance_labels [N] i64, instance_pointnum [I] i32,
 *  proposals_iou [P,I] f32 out.  iou = (f32)inter / ((f64)(f32)(len_p + len_i - inter) + 1e-5) rounded to f32.
 */
int pbn_get_iou(const int32_t* proposals_idx, const int32_t* proposals_offset, const int64_t* instance_labels,
                const int32_t* instance_pointnum, float* proposals_iou, int n_instance, int n_proposal,
                pbn_stream_t stream);

/* Replaces PB_lib.cal_iou_and_masklabel (lib/PB_lib/src/cal_iou_and_masklabel/cal_iou_and_masklabel.h:14-16,
 * cal_iou_and_masklabel.cu:15-107).  Exported by the reference but never called by its Python; kept so the
 * module symbol table is identical.  mask_label must be pre-filled with -1 by the caller (pbnet_ops.py:121). */
int pbn_cal_iou_and_masklabel(const int32_t* proposals_idx, const int32_t* proposals_offset,
                              const int64_t* instance_labels, const int32_t* instance_pointnum,
                              float* proposals_iou, int n_instance, int n_proposal,
                              const float* mask_scores_sigmoid, float* mask_label, int mode, pbn_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------
 * Sparse-voxel backbone: coordinate hashing and kernel maps.
 * Re-creates the coordinate-manager behaviour the reference obtains from MinkowskiEngine, an un-vendored,
 * un-pinned third-party dependency (README.md:15-27): ME.SparseTensor construction (network/PBNet.py:117,240-247,
 * 265-271), ME.utils.sparse_quantize (datasets/scannetv2/dataset_preprocess.py:269-272), strided / transposed
 * coordinate maps and kernel maps behind MinkowskiConvolution(Transpose) (network/Mink.py:221-288).
 * Coordinates are int32 rows (batch, x, y, z); batch in [0,65534], x/y/z in [-32768,32767].
 * Hash tables are caller-owned: keys uint64[capacity], vals int32[capacity], capacity = pbn_hash_capacity(n).
 * Data-dependent row counts are written to DEVICE ints; a count of -1 flags an out-of-range coordinate.
 * `n_*_dev` inputs may be NULL (then the *_max value is the exact count).
 */
int pbn_hash_capacity(int n);
size_t pbn_coords_workspace_bytes(int n_max);

/* De-duplicate rows: first occurrence survives, survivors keep ascending original order.
 * unique_index[n_unique] (original row of each survivor), inverse[n] (survivor id of every input row, may be NULL),
 * unique_coords[n_unique,4] (may be NULL).  The table maps coordinate -> survivor id afterwards. */
int pbn_coords_unique(const int32_t* coords, const int32_t* n_dev, int n_max, uint64_t* table_keys, int32_t* table_vals,
                      int capacity, int32_t* unique_index, int32_t* inverse, int32_t* unique_coords, int32_t* n_unique,
                      void* workspace, size_t workspace_bytes, pbn_stream_t stream);

/* Coarser coordinate set floor(c / stride_out) * stride_out (first-occurrence order) of a unique fine set whose
 * tensor stride is stride_out/2.  parent_row[n_fine], child_k[n_fine] in [0,8) (kernel index of the k=2,s=2
 * convolution, x fastest), nbr_down[n_coarse,8] child rows (-1 = absent).  The table maps coarse coordinate -> row. */
int pbn_coords_stride(const int32_t* fine_coords, const int32_t* n_fine_dev, int n_fine_max, int stride_out,
                      uint64_t* table_keys, int32_t* table_vals, int capacity, int32_t* coarse_coords,
                      int32_t* parent_row, int32_t* child_k, int32_t* nbr_down, int32_t* n_coarse, void* workspace,
                      size_t workspace_bytes, pbn_stream_t stream);

/* Output-stationary kernel map: nbr[row, k] = row of (out_coords[row] + offsets[k]) in the table, or -1.
 * offsets int32[n_offsets,3] (already multiplied by the tensor stride). */
int pbn_kernel_map(const int32_t* out_coords, const int32_t* n_out_dev, int n_out_max, const int32_t* offsets,
                   int n_offsets, const uint64_t* table_keys, const int32_t* table_vals, int capacity, int32_t* nbr,
                   pbn_stream_t stream);

/* Kernel map of a K^3 hyper-cube whose offsets are generated on the fly: odd K centred, even K not, offsets scaled by
 * tensor_stride, first spatial dimension fastest when x_fastest != 0 (the MinkowskiEngine convention assumed here). */
int pbn_kernel_map_cube(const int32_t* out_coords, const int32_t* n_out_dev, int n_out_max, int kernel_size,
                        int tensor_stride, int x_fastest, const uint64_t* table_keys, const int32_t* table_vals,
                        int capacity, int32_t* nbr, pbn_stream_t stream);

/* Table of the transposed k=2,s=2 convolution: nbr_up[fine_row, k] = parent row for k == child_k[fine_row], else -1. */
int pbn_up_table(const int32_t* parent_row, const int32_t* child_k, const int32_t* n_fine_dev, int n_fine_max,
                 int32_t* nbr_up, pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sparse convolution forward (implicit GEMM on MFMA), replacing MinkowskiConvolution / MinkowskiConvolutionTranspose /
 * MinkowskiLinear forward as used at network/Mink.py:293-350 and network/PBNet.py:43-82,121-123,249,273-277:
 *     out[o, c] = act( (sum_k sum_ci in[nbr[o,k], ci] * W[k, ci, c]) * scale[c] + shift[c] + residual[o, c] )
 *  in_feat   [n_in, ld_in] T  input slab (pointer already advanced to the first input column); padding columns
 *                            up to vecs_per_offset*(16/sizeof(T)) must be readable and finite.  Rows are gathered
 *                            with 32-bit byte offsets: n_in*ld_in*sizeof(T) must be < 2 GiB, else PBN_ERR_RANGE
 *                            (entries of nbr outside [0, n_in) read zeros, never memory outside the slab)
 *  nbr       [n_out, n_offsets] i32 or NULL (identity: 1x1 convolution / linear, n_offsets must be 1)
 *  row_perm  optional processing order (tile position p computes output row row_perm[p])
 *  w_packed  weights in MFMA fragment order: [n_steps][cout_padded/16][64 lanes][16 bytes] where lane l of
 *            (step s, tile t) holds W_flat[(4*s + l/16)*E + j, 16*t + l%16], j < E = 16/sizeof(T), and W_flat is
 *            W[k, ci, c] flattened over (k, ci) with ci padded to vecs_per_offset*E; zero outside the real extent
 *  scale/shift [cout_padded] f32 or NULL (folded eval-mode BatchNorm / bias)
 *  residual  [*, ld_res] T or NULL;  relu != 0 applies max(.,0);  out_feat [*, ld_out] T, cout_padded columns written
 *  dtype     PBN_F32 (v_mfma_f32_16x16x4_f32, exact fp32: the parity configuration), PBN_BF16, PBN_F16
 *  rows_per_wave 0 = automatic choice of kernel family and tile (production); 16 / 32 = the workgroup-tile family of
 *                csrc/spconv.hip with that many rows per wave; >= 100 = the wave-autonomous family of
 *                csrc/spconv_wave.hip in configuration 1000*ksplit + 100*NF + NT (NF 16-row fragments per wave, NT
 *                16-channel tiles per workgroup; PBN_ERR_UNSUPPORTED for a combination that is not built) -- the explicit
 *                values exist for tests and tuning.
 *  workspace     optional scratch (16-byte aligned) for split-K launches: when the row count is too small to fill the
 *                256 CUs the reduction axis is cut into slices whose fp32 partial sums go through this buffer and are
 *                combined in a fixed order by a second kernel; NULL / too small => single-pass launch.
 *  fp32 accumulation, fixed summation order: results are deterministic (bit-identical run to run).
 */
int pbn_spconv_forward(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets, const int32_t* row_perm,
                       const int32_t* n_out_dev, int n_out, const void* w_packed, int vecs_per_offset, int n_steps,
                       int cout_padded, const float* scale, const float* shift, const void* residual, int ld_res,
                       int relu, void* out_feat, int ld_out, int dtype, int rows_per_wave, void* workspace,
                       size_t workspace_bytes, pbn_stream_t stream);

/* The same convolution with a SECOND SOURCE folded into its reduction (round 4): a BasicBlock's 1x1 shortcut
 * (network/Mink.py:77-87: `downsample` = 1x1 convolution + BatchNorm on the block input, added before the last ReLU)
 * becomes extra reduction steps of the block's second convolution --
 *     out[o, c] = act( (sum_k sum_ci in[nbr[o,k], ci] W[k, ci, c] + sum_cj in2[o, cj] W2[cj, c]) * scale[c] + shift[c] + ... )
 * -- one launch and no residual slab instead of two launches.  w_packed holds the steps of the map followed by the steps of
 * the second source: vecs_second / 4 of them, zero-padded by the caller to a whole number of barrier groups of the first
 * source (largest divisor <= 4 of vecs_per_offset / 4), n_steps = the total.  BatchNorm scales of the two branches differ:
 * the caller folds them into the packed weights (scale NULL) and passes the summed shifts.  Both sources: vectors per row a
 * multiple of 4; in2 has at least n_out rows (row o pairs with output row o). */
int pbn_spconv_forward_dual(const void* in_feat, int ld_in, int n_in, const int32_t* nbr, int n_offsets,
                            const int32_t* n_out_dev, int n_out, const void* w_packed, int vecs_per_offset, int n_steps,
                            int cout_padded, const float* scale, const float* shift, const void* residual, int ld_res,
                            int relu, void* out_feat, int ld_out, int dtype, int rows_per_wave, void* workspace,
                            size_t workspace_bytes, const void* in2_feat, int ld_in2, int n_in2, int vecs_second,
                            pbn_stream_t stream);

/* Which kernel family pbn_spconv_forward's automatic choice (rows_per_wave = 0) gives a launch of this shape: 0 workgroup-tile
 * (csrc/spconv.hip), 1 wave-autonomous (spconv_wave.hip), 2 row-stationary (spconv_rs.hip).  No launch; for reports. */
int pbn_spconv_family(int n_out, int n_offsets, int vecs_per_offset, int n_steps, int cout_padded, int dtype, int has_map);

/* out[i, :] = in[idx[i], :] on 16-byte multiples (voxel -> point gathers, network/PBNet.py:130-134,250); a negative
 * index gives a zero row (padding slots of the compacted weight-gradient operands). */
int pbn_gather_rows(const void* in, int ld_in_bytes, const int64_t* idx, int n, int row_bytes, void* out,
                    int ld_out_bytes, pbn_stream_t stream);

/* Per-batch global max and/or average pooling of a slab whose rows are grouped by batch index
 * (MinkowskiGlobalMaxPooling / MinkowskiGlobalAvgPooling of the score branch, network/PBNet.py:67-68,274-276).
 * seg_start int32[n_seg+1] row offsets; out_max / out_avg f32 [n_seg, channels] (either may be NULL).
 * An empty segment yields -inf / NaN exactly like the reductions it replaces.  Deterministic.
 * workspace (optional, pbn_segment_pool_workspace_bytes): lets long segments be reduced by many workgroups in two
 * fixed-order passes; without it every (segment, 32-channel chunk) is one workgroup. */
size_t pbn_segment_pool_workspace_bytes(int n_seg, int channels);
int pbn_segment_pool(const void* feats, int ld, int channels, int dtype, const int32_t* seg_start, int n_seg,
                     float* out_max, float* out_avg, void* workspace, size_t workspace_bytes, pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Stage glue of PBNet.forward (csrc/stages.hip): each call replaces a run of small tensor ops of the reference by one
 * launch with identical integer results and the same fp32 arithmetic.
 *
 * pbn_local_scene_rows -- rows of every local scene (network/PBNet.py:182-247).  Entry e (one grouping cluster taking
 * part in one local scene) owns rows [ent_row_start[e], ent_row_start[e+1]); row j of it is the point
 * ins_ind[member_idx[ent_member_start[e] + j]].  Per row: point_idx, row_scene = ent_scene[e], the mask-branch voxel
 * coordinate (scene, floor(xyz * inv_voxel)) -- the reference divides a device tensor by a host scalar, which is a
 * multiplication by the fp32 reciprocal -- and the mask-branch input row
 *   [point_feat[p, 0:channels], sem_score[p, sem_pred[p]], ent_weight[e], 0 ... 0]   (ld_out elements of `dtype`);
 * sem_pred NULL = column 0 (sem_score is then the per-point own-class score of pbn_sem_argmax_table).
 *
 * pbn_gather_pad_rows -- out[i, 0:row_bytes] = in[idx[i], 0:row_bytes] (idx NULL = identity), the rest of the output
 * row zero-filled: builds a 16-byte-aligned, zero-padded input slab in one pass.  All byte counts multiples of 4. */
int pbn_local_scene_rows(const int32_t* ent_row_start, const int32_t* ent_member_start, const int32_t* ent_scene,
                         const float* ent_weight, int n_ent, int n_rows, const int32_t* member_idx,
                         const int64_t* ins_ind, const float* xyz, float inv_voxel, const void* point_feat, int ld_feat,
                         int channels, const void* sem_score, int ld_sem, const int64_t* sem_pred, int dtype,
                         int64_t* point_idx, int64_t* row_scene, int32_t* coords, void* feat_out, int ld_out,
                         pbn_stream_t stream);
int pbn_gather_pad_rows(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx, int n, void* out,
                        int ld_out_bytes, pbn_stream_t stream);

/* pbn_pack_weight -- fp32 master weights [n_offsets, dim_a, dim_b] (the reference's `kernel` layout [K, Cin, Cout]) into
 * the `w_packed` fragment order pbn_spconv_forward reads, in one launch.  flip mirrors the offsets (K-1-k), transpose
 * swaps the channel roles: flip + transpose of a centred forward kernel are its input-gradient weights.
 * vecs_per_offset / n_steps / cout_padded as for pbn_spconv_forward with (Cin, Cout) = transpose ? (dim_b, dim_a) :
 * (dim_a, dim_b).  out: n_steps * cout_padded/16 * 1024 bytes. */
int pbn_pack_weight(const float* src, int n_offsets, int dim_a, int dim_b, int flip, int transpose, int dtype,
                    int vecs_per_offset, int n_steps, int cout_padded, void* out, pbn_stream_t stream);

/* pbn_pack_weights_batch -- the same packing for many layers in ONE launch (a training step repacks every layer's forward
 * and input-gradient weights once): `jobs` is a DEVICE array of n_jobs descriptors with pbn_pack_weight's arguments
 * (cin / cout / cin_p = vecs_per_offset * elements-per-16-bytes spelled out), max_vectors = the largest job's
 * n_steps * cout_p/16 * 64 (sizes the grid).  All jobs share `dtype`. */
typedef struct pbn_pack_job {
    const float* src;      /* fp32 master [n_offsets, dim_a, dim_b] */
    void* out;             /* n_steps * cout_p/16 * 1024 bytes */
    int32_t n_offsets, dim_a, dim_b, flip, transpose, cin, cout, cin_p, cout_p, n_steps;
    int32_t reserved[2];
} pbn_pack_job;
int pbn_pack_weights_batch(const pbn_pack_job* jobs, int n_jobs, int max_vectors, int dtype, pbn_stream_t stream);

/* pbn_gather_rulebook_rows -- the gathered operand of the weight gradient (training, BASELINE configs[2]):
 * out[v, j, 0:row_bytes] = in[nbr[v, k0 + j], 0:row_bytes] for j < kc, zeros where nbr is -1; out is dense
 * [n, kc, row_bytes].  dW[k0:k0+kc] is then one dense contraction of this slab with the output gradient. */
int pbn_gather_rulebook_rows(const void* in, int ld_in_bytes, int row_bytes, const int32_t* nbr, int n_offsets, int k0,
                             int kc, int n, void* out, pbn_stream_t stream);

/* Train-mode batch normalisation of a feature slab x[n, channels] (row stride ld_* elements, rows 16-byte aligned;
 * ME.MinkowskiBatchNorm = torch.nn.BatchNorm1d on .F, /root/reference/network/Mink.py:71-73,224; training, configs[2]).
 *   forward : save_mean / save_invstd f32[channels] from the batch (biased variance), running_mean / running_var (NULL =
 *             not tracked) updated with `momentum` and the unbiased variance, y = (x - mean) * invstd * weight + bias
 *             (weight / bias NULL = 1 / 0).  fp32 arithmetic, sums merged in a fixed order (double for the final merge).
 *   backward: dx, dweight = sum dy * xhat, dbias = sum dy (either may be NULL).
 * workspace: pbn_bn_workspace_bytes(channels), 16-byte aligned; its FIRST 16 BYTES must be zero when a workspace is used for the
 * first time (a ticket counter of the fused merge step -- small slabs merge their block sums in the last workgroup to
 * finish instead of a second launch -- which every call leaves at zero again); one workspace per stream.
 * PBN_ERR_UNSUPPORTED when channels is not a multiple of the 16-byte
 * vector width of `dtype` or a row is not 16-byte aligned (the caller then uses the framework's batch norm). */
size_t pbn_bn_workspace_bytes(int channels);
int pbn_bn_train_forward(const void* x, int ld_x, int n, int channels, int dtype, const float* weight, const float* bias,
                         float eps, float momentum, float* running_mean, float* running_var, void* y, int ld_y,
                         float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, pbn_stream_t stream);
int pbn_bn_train_backward(const void* x, int ld_x, const void* dy, int ld_dy, int n, int channels, int dtype,
                          const float* weight, const float* save_mean, const float* save_invstd, void* dx, int ld_dx,
                          float* dweight, float* dbias, void* workspace, size_t workspace_bytes, pbn_stream_t stream);

/* The same passes with the tail of the reference's blocks fused in (Mink.py:293-350, resnet block: conv -> bn -> relu and
 * conv -> bn -> += residual -> relu):
 *   forward : y = act(bn(x) [+ residual]) with act = max(., 0) when relu != 0 (residual NULL = none);
 *   backward: g = dy where y > 0 else 0 (y = the forward output; NULL = no activation), dx / dweight / dbias from g,
 *             dres (NULL = not wanted) = g, the gradient of the residual branch.
 * Statistics, running buffers, workspace, summation order and error codes as pbn_bn_train_forward / _backward. */
int pbn_bn_act_train_forward(const void* x, int ld_x, int n, int channels, int dtype, const float* weight, const float* bias,
                             float eps, float momentum, float* running_mean, float* running_var, const void* residual,
                             int ld_res, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, void* workspace,
                             size_t workspace_bytes, pbn_stream_t stream);
int pbn_bn_act_train_backward(const void* x, int ld_x, const void* dy, int ld_dy, const void* y, int ld_y, int n, int channels,
                              int dtype, const float* weight, const float* save_mean, const float* save_invstd, void* dx,
                              int ld_dx, void* dres, int ld_dres, float* dweight, float* dbias, void* workspace,
                              size_t workspace_bytes, pbn_stream_t stream);

/* Offset-major pair lists of an output-stationary map (training, BASELINE configs[2]): the rule pairs of ME's
 * convolution weight gradient, dW[k] = sum over pairs (i, o) of offset k of x[i]^T g[o].
 *   pbn_rulebook_pair_counts : table int32[pbn_rulebook_pair_blocks(n), K] (per-block prefix counts, kept for the fill)
 *                              and totals int32[K] (pairs per offset) -- the caller reads the totals back, gives every
 *                              offset ceil(total/seg) segments of `seg` pairs and passes their first-segment indices
 *   pbn_rulebook_pair_fill   : in_idx / out_idx int32[n_segments * seg] (-1 = padding; int64 until round 3), seg_offset int64[n_segments]
 * Pairs of an offset keep ascending output-row order; positions are prefix counts (no atomics): deterministic. */
int pbn_rulebook_pair_blocks(int n);
int pbn_rulebook_pair_counts(const int32_t* nbr, int n, int n_offsets, int32_t* table, int32_t* totals, pbn_stream_t stream);
int pbn_rulebook_pair_fill(const int32_t* nbr, int n, int n_offsets, const int32_t* table, const int32_t* seg_start, int seg,
                           int n_segments, int32_t* in_idx, int32_t* out_idx, int64_t* seg_offset, pbn_stream_t stream);

/* pbn_rulebook_pair_fill_dev -- the same lists with NO host knowledge of the pair counts (no read-back between
 * pbn_rulebook_pair_counts and the fill: a training step builds ~40 maps): seg_begin int32[n_offsets + 1] is computed on the
 * device from `totals`, in_idx / out_idx / seg_offset must hold the worst case of n * n_offsets / segment + n_offsets
 * segments, only the segments below seg_begin[n_offsets] are written and may be read.  The tail of an offset's last segment
 * is NOT padded: the consumer bounds every offset by its pair count (pbn_spconv_wgrad: pair_counts = `totals`).  One launch
 * (every block derives seg_begin from the totals itself); n_offsets <= 512. */
int pbn_rulebook_pair_fill_dev(const int32_t* nbr, int n, int n_offsets, const int32_t* table, const int32_t* totals,
                               int segment, int32_t* seg_begin, int32_t* in_idx, int32_t* out_idx, int64_t* seg_offset,
                               pbn_stream_t stream);

/* pbn_rulebook_pairs_multi -- counts + device-side fill (the two calls above) of up to 16 maps in THREE launches: the
 * training executor needs the lists of all 14 maps of a lineage, most of them small (a launch costs more than their work).
 * Per job the buffers of pbn_rulebook_pair_counts / _fill_dev (table int32[max(pbn_rulebook_pair_blocks(n), 1) * n_offsets]). */
typedef struct {
    const int32_t* nbr;
    int32_t n, n_offsets;
    int32_t* table;
    int32_t* totals;
    int32_t* seg_begin;
    int32_t* in_idx;
    int32_t* out_idx;
    int64_t* seg_offset;
} pbn_pair_job;
int pbn_rulebook_pairs_multi(const pbn_pair_job* jobs, int n_jobs, int segment, pbn_stream_t stream);

/* Weight gradient of the sparse convolution on the matrix cores (csrc/wgrad.hip), ME's convolution backward w.r.t. the
 * kernel (reached from train.py:57 loss.backward()):  dw[k, ci, co] = sum over the pairs (i, o) of offset k of
 * x[i, ci] * g[o, co].  in_idx / out_idx / seg_begin: the lists of pbn_rulebook_pair_fill, seg_begin int32[K+1] = first
 * `segment`-pair segment of every offset, read on the DEVICE (with lists n_pairs_total only sizes the pair splits: an estimate
 * will do); pair_counts int32[K] (device, may be NULL) = pairs of every offset: without it a workgroup walks the -1 padding of
 * its offset's last segment as well (a stride-16 level holds ~450 pairs per offset in segments of 4096); all lists NULL = identity pairs (1x1 convolution / linear layer; n_offsets 1, n_pairs_total rows).  x [*, ld_x], g [*, ld_g] of `dtype` (widened exactly), dw f32[K, cin, cout], any cin / cout.
 * fp32 accumulation in a fixed order (deterministic).  workspace: pbn_spconv_wgrad_workspace_bytes (pair splits). */
size_t pbn_spconv_wgrad_workspace_bytes(int n_offsets, int cin, int cout);
int pbn_spconv_wgrad(const void* x, int ld_x, const void* g, int ld_g, int dtype, const int32_t* in_idx,
                     const int32_t* out_idx, const int32_t* seg_begin, const int32_t* pair_counts, int segment,
                     int n_pairs_total, int n_offsets, int cin, int cout, float* dw, void* workspace, size_t workspace_bytes,
                     pbn_stream_t stream);
/* The same with the caller's row counts: PBN_ERR_RANGE when a slab reaches 4 GiB (the kernels address rows with 32-bit byte
 * offsets) or the lists 2^30 pairs, PBN_ERR_ARG for device-built (unpadded) lists without pair_counts unless the caller declares
 * them padded (pbn_rulebook_pair_fill pads with -1), or for more identity pairs than rows.  What the package itself calls. */
int pbn_spconv_wgrad_checked(const void* x, int ld_x, long long n_x_rows, const void* g, int ld_g, long long n_g_rows, int dtype,
                             const int32_t* in_idx, const int32_t* out_idx, const int32_t* seg_begin, const int32_t* pair_counts,
                             int lists_padded, int segment, int n_pairs_total, int n_offsets, int cin, int cout, float* dw,
                             void* workspace, size_t workspace_bytes, pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Capacity-planned inference (csrc/plan.hip): the data-dependent sizes of PBNet.forward stay on the device.  Every buffer
 * is allocated at a CAPACITY chosen by the caller (pbnet_amd/planned.py takes them from a previous forward of a scene of the
 * same size class); kernels read the true counts from `counts` (int32[PBN_CNT_WORDS]) and raise bits of
 * counts[PBN_CNT_OVERFLOW] instead of writing past a capacity (the caller then falls back to the size-exact path).
 * The `_dev` variants of the stage kernels take (capacity, pointer to the device-side count).
 *   pbn_class_gate       network/PBNet.py:151-160,172-173: classes with (float)count < thr05[c] are dropped; class_base[c]
 *                        = first output slot of class c (-1 dropped), seg_len[(c-2)*nb + b] = points of (class, batch)
 *                        (0 for dropped classes: the grouping skips empty segments), counts[PBN_CNT_POINTS].
 *   pbn_local_plan       network/PBNet.py:182-234 for task 'test': one local scene per cluster; entry table in the layout
 *                        pbn_local_scene_rows reads; counts[ENTRIES, ROWS, SCENES, CLUSTERS].  kNN of the cluster centres:
 *                        ascending (d^2, id) with d^2 = (dx^2+dy^2)+dz^2 in unfused fp32.
 *   pbn_proposal_offsets network/PBNet.py:330-345: proposals_offset i64[s_cap+1], surviving scene ids, dense renumbering;
 *                        counts[PROPOSALS, PROPOSAL_ROWS].
 *   pbn_batch_starts     seg_start[s] = first row of a batch-sorted [n,4] coordinate list whose batch index is >= s. */
enum {
    PBN_CNT_POINTS = 0, PBN_CNT_CLUSTERS = 1, PBN_CNT_ENTRIES = 2, PBN_CNT_ROWS = 3, PBN_CNT_SCENES = 4,
    PBN_CNT_PROPOSAL_ROWS = 5, PBN_CNT_PROPOSALS = 6, PBN_CNT_OVERFLOW = 7, PBN_CNT_WORDS = 16
};
enum {
    PBN_OVF_POINTS = 1, PBN_OVF_CLUSTERS = 2, PBN_OVF_ENTRIES = 4, PBN_OVF_ROWS = 8, PBN_OVF_SEGMENT = 16,
    PBN_OVF_BATCH = 32, PBN_OVF_LEVEL = 64,
    PBN_OVF_CDIST = 128     /* pbn_local_plan: a (class, batch) segment with more than 25 clusters needed its neighbours ranked --
                             * torch.cdist (PBNet.py:201) switches to its matrix-multiply distance there, whose rounding can rank near-ties
                             * differently from the direct d^2 of this plan: the caller takes the reference's own call (host plan) */
};
int pbn_class_gate(const int32_t* table, const float* thr05, int n_classes, int nb, int m_cap, int n_points,
                   int32_t* class_base, int32_t* seg_len, int32_t* counts, pbn_stream_t stream);
size_t pbn_local_plan_workspace_bytes(int c_cap);
int pbn_local_plan(const int32_t* cluster_num, int n_segments, int nb, const int32_t* member_start, const float* centers,
                   const int32_t* n_clusters, const float* thr02, const int32_t* kmax, int c_cap, int e_cap, int r_cap,
                   int32_t* ent_row_start, int32_t* ent_member_start, int32_t* ent_scene, float* ent_weight,
                   int32_t* counts, void* workspace, size_t workspace_bytes, pbn_stream_t stream);
int pbn_proposal_offsets(const int32_t* per_scene, int s_cap, int64_t* proposals_offset, int64_t* alive_ids,
                         int32_t* dense_of, int32_t* counts, pbn_stream_t stream);
int pbn_batch_starts(const int32_t* coords, const int32_t* n_dev, int n_cap, int n_segments, int32_t* seg_start,
                     pbn_stream_t stream);
int pbn_local_scene_rows_dev(const int32_t* ent_row_start, const int32_t* ent_member_start, const int32_t* ent_scene,
                             const float* ent_weight, int n_ent_cap, int n_rows_cap, const int32_t* n_ent_dev,
                             const int32_t* n_rows_dev, const int32_t* member_idx, const int64_t* ins_ind, const float* xyz,
                             float inv_voxel, const void* point_feat, int ld_feat, int channels, const void* sem_score,
                             int ld_sem, const int64_t* sem_pred, int dtype, int64_t* point_idx, int64_t* row_scene,
                             int32_t* coords, void* feat_out, int ld_out, pbn_stream_t stream);
int pbn_gather_pad_rows_dev(const void* in, int ld_in_bytes, int row_bytes, const int64_t* idx, const int64_t* idx2,
                            int n_cap, const int32_t* n_dev, void* out, int ld_out_bytes, pbn_stream_t stream);
/* in_rows: the row capacity of `in`; a resolved row index outside [0, in_rows) reads as a row of zeros (an index table of a
 * level that overflowed its capacity names rows that were never computed). */
int pbn_mlp_rows_dev(const void* in, int ld_in, int in_rows, int channels, const int64_t* idx_a, const int64_t* idx_b, int n_cap,
                     const int32_t* n_dev, const float* w1, const float* scale, const float* shift, const float* slope,
                     int hidden, const float* w2, const float* b2, int n_out, int sigmoid, void* out, int ld_out, int dtype,
                     pbn_stream_t stream);
int pbn_mask_count_dev(const void* mask_score, int ld, float thd, const int64_t* row_scene, int n_cap, const int32_t* n_dev,
                       int n_scenes_cap, int dtype, int32_t* per_scene, int32_t* block_cnt, pbn_stream_t stream);
int pbn_proposal_rows_dev(const void* mask_score, int ld, float thd, const int64_t* row_scene, const int64_t* point_idx,
                          int n_cap, const int32_t* n_dev, const int32_t* dense_of, const int32_t* block_cnt,
                          const float* xyz, float scale, float inv_voxel, const void* point_feat, int ld_feat, int channels,
                          int dtype, int64_t* proposals_idx, void* proposals_ms, int32_t* coords, void* feat_out,
                          pbn_stream_t stream);

/* pbn_mlp_rows -- the two-layer heads of network/PBNet.py:43-82 in eval mode, one launch per head:
 *   out[i, 0:n_out] = act( W2 . prelu( (W1 . x) * scale + shift ) + b2 ),   x = in[row(i), 0:channels],
 *   row(i) = idx_b[idx_a[i]] (either index level may be NULL), act = sigmoid or identity.
 * W1 f32[hidden, channels], scale/shift = eval-mode BatchNorm folded (f32[hidden]), slope f32[hidden] (PReLU),
 * W2 f32[n_out, hidden], b2 f32[n_out] or NULL.  fp32 arithmetic, one rounding to `dtype` at the end.
 * channels must be 32 and hidden 16 or 32 (the shapes PBNet builds); anything else returns PBN_ERR_UNSUPPORTED. */
int pbn_mlp_rows(const void* in, int ld_in, int channels, const int64_t* idx_a, const int64_t* idx_b, int n,
                 const float* w1, const float* scale, const float* shift, const float* slope, int hidden, const float* w2,
                 const float* b2, int n_out, int sigmoid, void* out, int ld_out, int dtype, pbn_stream_t stream);

/* pbn_sem_argmax_table -- network/PBNet.py:134,151-163: per point the arg-max class (first maximum), the softmax score
 * of that class (1 / sum exp(s - max), fp32, rounded to `dtype`; may be NULL) and the [n_cls, nb] population table
 * (zeroed here; points whose batch index is outside [0, nb) are not counted).  block_hist int32[pbn_select_blocks(n),
 * n_cls] receives the per-block class histogram pbn_select_points needs.
 *
 * pbn_select_points -- network/PBNet.py:151-170: the points of the kept classes in class-major order, ascending point
 * index inside a class (what a stable sort by class yields), written straight to the grouping inputs:
 *   position(i) = class_base[c] + #(j < i with class c);   class_base[c] < 0 drops the class.
 *   ins_ind[pos] = i, ins_orig[pos] = xyz[i], ins_off[pos] = xyz[i] + float(offset[i]) (fp32 add), ins_sem[pos] = c. */
int pbn_select_blocks(int n);
int pbn_sem_argmax_table(const void* score, int ld, int n_cls, const int32_t* batch, int nb, int n, int dtype,
                         int64_t* sem_pred, void* sem_prob, int32_t* table, int32_t* block_hist, pbn_stream_t stream);
int pbn_select_points(const int64_t* sem_pred, int n, int n_cls, const int32_t* class_base, const int32_t* block_hist,
                      const float* xyz, const void* offset, int ld_off, int dtype, int64_t* ins_ind, float* ins_orig,
                      float* ins_off, int32_t* ins_sem, pbn_stream_t stream);

/* get_proposal (network/PBNet.py:317-347) and the score-branch inputs (:240-252) as an order-preserving compaction:
 *   pbn_mask_count    : rows with mask_score > thd per local scene (per_scene, zeroed here) and per block of rows
 *                       (block_cnt int32[pbn_select_blocks(n)]); the host turns per_scene into the dense renumbering
 *                       of the surviving scenes ("remove null proposals", :342-345) and the proposal offsets;
 *   pbn_proposal_rows : for every kept row, in row order: proposals_idx = (dense_of[row_scene], point_idx),
 *                       proposals_ms = its mask score and, optionally, the score-branch voxel coordinate
 *                       (dense id, floor(xyz[p] * scale * inv_voxel)) and input feature row point_feat[p, 0:channels]. */
int pbn_mask_count(const void* mask_score, int ld, float thd, const int64_t* row_scene, int n, int n_scenes, int dtype,
                   int32_t* per_scene, int32_t* block_cnt, pbn_stream_t stream);
int pbn_proposal_rows(const void* mask_score, int ld, float thd, const int64_t* row_scene, const int64_t* point_idx, int n,
                      const int32_t* dense_of, const int32_t* block_cnt, const float* xyz, float scale, float inv_voxel,
                      const void* point_feat, int ld_feat, int channels, int dtype, int64_t* proposals_idx,
                      void* proposals_ms, int32_t* coords, void* feat_out, pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Evaluation-time post-processing (csrc/post.hip) -- eval_map.py:55-123, the step right after PBNet.forward
 * (SURVEY.md 8f rank 1).  A proposal is a bitset over the n_fold = N/3 folded points (pbn_post_words(n_fold) 32-bit
 * words per row) instead of the reference's dense [P, N/3] int mask; all results are integers or fp32 quotients of
 * exact integers, i.e. identical to the reference's tensors.
 *   pbn_proposal_bitmask  : eval_map.py:67-70 (TTA fold `% (point_num/3)`, mask rows) + row sizes (:80)
 *   pbn_mask_iou          : eval_map.py:90-96 for the rows listed in `rows` (NULL = all): iou f32[n_rows, n_rows]
 *   pbn_superpoint_refine : eval_map.py:104-116 + tools/getins.py:72-98: per-point label of the picked clusters (the
 *                           last one containing the point), per-superpoint label histogram hist[n_sp, n_pick+1] (bucket
 *                           n_pick = unlabelled), first arg-max label per superpoint, refined per-point labels and the
 *                           rebuilt cluster bitsets masks_out[n_pick, words] with their sizes (0 = cluster vanished)
 *   pbn_bitmask_to_dense  : int32[n_rows, n_fold] 0/1 tensor of selected rows (the reference's `clusters`)
 * The greedy NMS between the two (tools/mIOU.py:77-87) runs on the host on a [P, P] matrix, as in the reference. */
int pbn_post_words(int n_fold);
int pbn_proposal_bitmask(const int64_t* proposals_idx, int n_entries, int n_fold, int n_prop, uint32_t* masks,
                         int32_t* counts, pbn_stream_t stream);
int pbn_mask_iou(const uint32_t* masks, const int32_t* rows, int n_rows, int n_fold, const int32_t* counts, float* iou,
                 pbn_stream_t stream);
int pbn_superpoint_refine(const uint32_t* masks, const int32_t* pick, int n_pick, int n_fold, const int64_t* superpoint,
                          int n_sp, int64_t* seg, int32_t* hist, int64_t* sp_label, int64_t* seg_refined,
                          uint32_t* masks_out, int32_t* counts_out, pbn_stream_t stream);
int pbn_bitmask_to_dense(const uint32_t* masks, const int32_t* rows, int n_rows, int n_fold, int32_t* dense,
                         pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * ScanNet AP evaluator, the association step of /root/reference/tools/eval.py:205-250
 * (`assign_instances_for_scan`).  The reference counts, for every (prediction, ground-truth instance) pair, the points
 * both contain with one pass over the scene per pair (:239) plus one for the void overlap (:235); here ONE pass per
 * prediction fills a row of the overlap table: inter[p][gt_index[i]] += 1 for every point i with masks[p][i] != 0.
 *   masks    : int32[n_pred, n_pts], any non-zero value = inside (eval.py:226 `np.not_equal(pred_mask, 0)`)
 *   gt_index : int32[n_pts], the point's ground-truth id as an index into the scene's sorted unique id list
 *              (0 <= index < n_gt; the ids themselves -- class*1000+instance+1, 0 = unannotated,
 *              datasets/scannetv2/get_val_gt.py:26-39 -- stay on the host)
 *   inter    : int32[n_pred, n_gt] out (zeroed here).  Row sums are the predictions' vertex counts (:227), columns of
 *              non-benchmark ids sum to the void intersection (:235), the rest are the `intersection` fields (:239).
 * Matching and the precision/recall integration (eval.py:27-190) are scalar bookkeeping on these counts and run on
 * the host (pbnet_amd/evaluate.py). */
int pbn_instance_overlap(const int32_t* masks, int n_pred, int n_pts, const int32_t* gt_index, int n_gt, int32_t* inter,
                         pbn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * One-call sub-pipelines (csrc/executor.hip): they only sequence the entry points above.
 *
 * pbn_coords_build: everything a MinkUNet needs from one coordinate lineage -- de-duplication, the four coarser
 * levels (tensor strides 2..16), the k=3 maps of all five levels, optionally the k=5 map of level 0, and the four
 * transposed-convolution tables -- laid out in one caller-owned arena.  Every array is sized for n rows (the input
 * size bounds every level); the real row counts are the first five ints at `counts` (device memory, -1 = range error).
 * Layout offsets are bytes from the arena base and are filled by pbn_coords_arena_bytes(). */
typedef struct {
    int64_t counts, unique_index, inverse;
    int64_t keys[5], vals[5], coords[5], k3[5];
    int64_t parent_row[4], child_k[4], nbr_down[4], up[4];
    int64_t k5, workspace, workspace_bytes;
    int32_t capacity[5];
    int32_t _pad;
} pbn_coords_layout;

size_t pbn_coords_arena_bytes(int n, int want_k5, pbn_coords_layout* layout);
int pbn_coords_build(const int32_t* coords, const int32_t* n_dev, int n, int want_k5, int x_fastest, void* arena,
                     size_t arena_bytes, const pbn_coords_layout* layout, pbn_stream_t stream);

/* pbn_coords_prepare: what constructing a SparseTensor and expanding its lineage in Z-order takes, in ONE call
 * (network/PBNet.py:117,240-247,265-271 reach this through ME.SparseTensor): de-duplication of the input rows (first
 * occurrence wins, survivors in ascending input order = the external row order), Z-order sort of the survivors, and the
 * pbn_coords_build expansion of the sorted rows.  Everything lives in one caller-owned arena:
 *   pyramid       the sorted lineage exactly as pbn_coords_build lays it out (counts[0..4] = rows per level, -1 on a
 *                 coordinate range error)
 *   n_unique      int32: survivors (= counts[0])
 *   unique_index  int64[n]: external row -> input row          inverse   int64[n]: input row -> external row
 *   perm          int64[n]: sorted position -> external row    inv_perm  int64[n]: external row -> sorted position
 *   ucoords       int32[n,4]: survivor coordinates in the external order
 * (only the first n_unique entries of the per-survivor arrays are meaningful; the rest of the struct is scratch). */
typedef struct {
    pbn_coords_layout pyramid;
    int64_t n_unique, unique_index, inverse, perm, inv_perm, ucoords;
    int64_t tmp_keys, tmp_vals, uidx32, inv32, sort_keys, sort_vals, sort_temp, sort_temp_bytes;
} pbn_prepare_layout;

size_t pbn_coords_prepare_bytes(int n, int want_k5, pbn_prepare_layout* layout);
int pbn_coords_prepare_dev(const int32_t* coords, const int32_t* n_dev, int n_cap, int want_k5, int x_fastest, void* arena,
                           size_t arena_bytes, const pbn_prepare_layout* layout, pbn_stream_t stream);
int pbn_coords_prepare(const int32_t* coords, int n, int want_k5, int x_fastest, void* arena, size_t arena_bytes,
                       const pbn_prepare_layout* layout, pbn_stream_t stream);
/* The same contract through the hash-table pipeline of pbn_coords_build (one table per level, values renumbered) instead of
 * the sorted, hash-free one: the cross-check of csrc/pyramid.hip (every output array must be equal); n_dev may be null. */
int pbn_coords_prepare_hash(const int32_t* coords, const int32_t* n_dev, int n_cap, int want_k5, int x_fastest, void* arena,
                            size_t arena_bytes, const pbn_prepare_layout* layout, pbn_stream_t stream);

/* Z-order keys (batch-major, then bit-interleaved x, y, z) of coordinate rows; rows at or beyond *n_dev get the largest
 * key.  Sorting rows by this key turns every run of consecutive rows into a compact spatial block: convolution tiles
 * then drop whole offset groups and their gathers stay inside one XCD's L2. */
int pbn_morton_keys(const int32_t* coords, const int32_t* n_dev, int n_max, int64_t* keys, pbn_stream_t stream);

/* pbn_unet_forward: executes a static list of fused convolutions -- MinkUNetBase.forward (network/Mink.py:291-354) with
 * eval-mode BatchNorm, ReLU and residual adds folded into the epilogues and skip concatenations written in place.
 * Buffers are symbolic: buffer 0 is the caller's input slab, buffer b > 0 is [n_rows[level], width] elements inside the
 * arena (offsets from pbn_unet_arena_bytes).  map_kind: 0 identity (1x1 / linear), 1 k=3 at level_out, 2 k=5 (level 0),
 * 3 k=2,s=2 down (level_in = fine level), 4 transposed k=2,s=2 (level_out = fine level). */
typedef struct {
    int32_t map_kind, level_in, level_out;
    int32_t in_buf, in_col, res_buf, res_col, out_buf, out_col;
    int32_t vpo, n_steps, cout_p, relu;
    int32_t in2_buf;            /* >= 0: second source of pbn_spconv_forward_dual (a folded 1x1 shortcut), else -1 */
    int32_t in2_col, vpo2;
    const void* w;
    const float* scale;
    const float* shift;
} pbn_unet_op;

typedef struct {
    int32_t level, width;
} pbn_unet_buf;

size_t pbn_unet_arena_bytes(const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows, int dtype, int64_t* buf_offsets);
/* Round 5: rows EXPECTED per level (5 ints, copied) for the next pbn_unet_forward_dev call of this thread, whose row counts are
 * capacities: kernel families and tile shapes are then chosen for the rows expected (as the size-exact forward would choose them),
 * grids for the capacities.  NULL disarms.  Consumed by the next pbn_unet_forward* call of the thread, whatever it returns. */
void pbn_unet_set_rows_hint(const int32_t* rows);
int pbn_unet_forward_dev(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows_cap,
                         const int32_t* n_rows_dev, const void* input, int ld_input, const int32_t* const* k3,
                         const int32_t* k5, const int32_t* const* down, const int32_t* const* up, void* arena,
                         size_t arena_bytes, int dtype, void* splitk_ws, size_t splitk_bytes, pbn_stream_t stream);
int pbn_unet_forward(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows,
                     const void* input, int ld_input, const int32_t* const* k3, const int32_t* k5,
                     const int32_t* const* down, const int32_t* const* up, void* arena, size_t arena_bytes, int dtype,
                     void* splitk_ws, size_t splitk_bytes, pbn_stream_t stream);

/* Measurement variant of pbn_unet_forward: brackets every op with HIP events on the launching stream, synchronises the
 * stream before returning and fills op_ms[n_ops] (host) with the per-op durations in milliseconds. */
int pbn_unet_forward_timed(const pbn_unet_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows,
                           const void* input, int ld_input, const int32_t* const* k3, const int32_t* k5,
                           const int32_t* const* down, const int32_t* const* up, void* arena, size_t arena_bytes,
                           int dtype, void* splitk_ws, size_t splitk_bytes, pbn_stream_t stream, float* op_ms);

/* ------------------------------------------------------------------------------------------------------------
 * Native TRAINING executor of a MinkUNet body (csrc/train_exec.hip): the train-mode forward of
 * /root/reference/network/Mink.py:291-350 (every convolution but the final 1x1: conv -> bn -> relu, the BasicBlocks with
 * their shortcuts, the skip concatenations written in place) and its backward as ONE call each -- the sequencing that
 * MinkowskiEngine/fused_train.py does from Python in ~330 native calls per network and direction (configs[2] was host-bound).
 * Same kernels, same order, same arithmetic as that path: pbn_spconv_forward (convolutions and input gradients),
 * pbn_bn_act_train_* (batch norm with its tail), pbn_spconv_wgrad.
 *   op       : convolution in (buf, col) -> pre_buf, then y = act(bn(pre) [+ res]) -> (out_buf, out_col).  Buffers are the
 *              pbn_unet_buf list (buffer 0 = the caller's input slab); pbn_unet_arena_bytes lays out BOTH arenas: the
 *              activations (kept for the backward) and, with the same offsets, their gradients.
 *   backward : ops in reverse; the caller has written d(loss)/d(output) into the gradient arena at the output buffer.
 *              dx_accumulate = the gradient of the input view already holds a contribution (a skip, a shortcut): the input
 *              gradient is then added in the convolution's epilogue (residual = out).  The residual gradient of a block is
 *              WRITTEN by the batch-norm backward, so it must be the first contribution of its buffer (the planner checks).
 *   pairs    : the weight gradient's pair lists per map (pbn_rulebook_pair_fill_dev): index 0..4 = k3 of levels 0..4,
 *              5 = k5, 6..9 = down of fine levels 0..3, 10..13 = up of fine levels 0..3; entries of unused maps may be zero.
 *   stats    : f32, per op [mean | invstd] at stat_off; param_grads: f32, per op dW [K, cin, cout] at dw_off, dgamma /
 *              dbeta at their offsets (all in floats). */
typedef struct {
    int32_t map_kind, level_in, level_out;
    int32_t in_buf, in_col, pre_buf, res_buf, res_col, out_buf, out_col;
    int32_t relu, cin, cout;
    int32_t vpo, n_steps, cout_p;            /* forward weights (pbn_pack_weight) */
    int32_t vpo_d, n_steps_d, cout_p_d;      /* input-gradient weights (offsets mirrored for cubes, channel roles swapped) */
    int32_t want_dx, dx_accumulate, _pad;
    const void* w;
    const void* w_d;
    const float* gamma;
    const float* beta;
    float* running_mean;                      /* NULL: statistics not tracked */
    float* running_var;
    float eps, momentum;
    int64_t stat_off, dw_off, dgamma_off, dbeta_off;
} pbn_train_op;

typedef struct {
    const int32_t* in_idx;
    const int32_t* out_idx;
    const int32_t* seg_begin;
    const int32_t* counts;
    int32_t segment, n_pairs_estimate;
} pbn_pair_lists;

int pbn_unet_train_forward(const pbn_train_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows,
                           const void* input, int ld_input, const int32_t* const* k3, const int32_t* k5,
                           const int32_t* const* down, const int32_t* const* up, void* act_arena, size_t arena_bytes,
                           float* stats, int dtype, void* splitk_ws, size_t splitk_bytes, void* bn_ws, size_t bn_ws_bytes,
                           pbn_stream_t stream);
int pbn_unet_train_backward(const pbn_train_op* ops, int n_ops, const pbn_unet_buf* bufs, int n_bufs, const int32_t* n_rows,
                            const void* input, int ld_input, const int32_t* const* k3, const int32_t* k5,
                            const int32_t* const* down, const int32_t* const* up, const pbn_pair_lists* pairs,
                            const void* act_arena, void* grad_arena, size_t arena_bytes, const float* stats,
                            float* param_grads, void* dinput, int ld_dinput, int dtype, void* splitk_ws, size_t splitk_bytes,
                            void* bn_ws, size_t bn_ws_bytes, void* wgrad_ws, size_t wgrad_ws_bytes, pbn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PBNET_HIP_H */
