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
                       int general_sem, int32_t* cluster_id, int32_t* cluster_num, int32_t* den, float* centers,
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

#ifdef __cplusplus
}
#endif
#endif /* PBNET_HIP_H */
