// iou.hip -- proposal x instance IoU (training-time score target) and the HAIS-style mask-label variant.
// Replaces /root/reference/lib/PB_lib/src/iou/get_iou.cu:12-38 and
// /root/reference/lib/PB_lib/src/cal_iou_and_masklabel/cal_iou_and_masklabel.cu:15-107.
//
// Layout: one workgroup per proposal; the proposal's points are streamed once (coalesced index reads, gathered
// 8-byte label reads) into an LDS histogram over instances, instead of the reference's O(P*I*len) rescans.
#include "pbn_common.h"

namespace pbn {
namespace {

constexpr int TPB = 256;

// mode -1: plain get_iou; mode 0/1: cal_iou_and_masklabel semantics (1 = only points with mask score > 0.5)
__global__ __launch_bounds__(TPB) void k_iou(const int* __restrict__ proposals_idx, const int* __restrict__ proposals_offset,
                                            const long long* __restrict__ instance_labels,
                                            const int* __restrict__ instance_pointnum, float* __restrict__ proposals_iou,
                                            int n_instance, int n_proposal, const float* __restrict__ mask_scores,
                                            int mode) {
    extern __shared__ __attribute__((aligned(16))) int hist[];  // n_instance + 1 ints
    for (int p = blockIdx.x; p < n_proposal; p += gridDim.x) {
        for (int k = threadIdx.x; k <= n_instance; k += TPB) hist[k] = 0;
        __syncthreads();
        const int start = proposals_offset[p], end = proposals_offset[p + 1];
        for (int i = start + (int)threadIdx.x; i < end; i += TPB) {
            if (mode == 1 && !(mask_scores[i] > 0.5f)) continue;
            atomicAdd(&hist[n_instance], 1);  // proposal_total for mode 1
            const int lab = (int)instance_labels[proposals_idx[i]];  // (int) cast as get_iou.cu:22
            if (lab >= 0 && lab < n_instance) atomicAdd(&hist[lab], 1);
        }
        __syncthreads();
        const int proposal_total = (mode == 1) ? hist[n_instance] : (end - start);
        for (int k = threadIdx.x; k < n_instance; k += TPB) {
            const int inter = hist[k];
            const int uni = proposal_total + instance_pointnum[k] - inter;
            // get_iou.cu:26: the 1e-5 literal is a double -> double denominator, double quotient, rounded to float
            proposals_iou[(size_t)p * n_instance + k] = (float)((double)(float)inter / ((double)(float)uni + 1e-5));
        }
        __syncthreads();
    }
}

// cal_iou_and_masklabel.cu:62-90
__global__ __launch_bounds__(TPB) void k_mask_label(const int* __restrict__ proposals_idx,
                                                   const int* __restrict__ proposals_offset,
                                                   const long long* __restrict__ instance_labels,
                                                   const float* __restrict__ proposals_iou, int n_instance,
                                                   int n_proposal, float* __restrict__ mask_label) {
    __shared__ int s_ind;
    __shared__ float s_iou;
    for (int p = blockIdx.x; p < n_proposal; p += gridDim.x) {
        if (threadIdx.x == 0) {
            float max_iou = 0.f;
            int max_ind = 0;
            for (int k = 0; k < n_instance; ++k) {
                const float v = proposals_iou[(size_t)p * n_instance + k];
                if (v > max_iou) { max_iou = v; max_ind = k; }
            }
            s_ind = max_ind;
            s_iou = max_iou;
        }
        __syncthreads();
        if (s_iou > 0.5f) {
            const int start = proposals_offset[p], end = proposals_offset[p + 1];
            for (int i = start + (int)threadIdx.x; i < end; i += TPB)
                mask_label[i] = ((int)instance_labels[proposals_idx[i]] == s_ind) ? 1.f : 0.f;
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace pbn

using namespace pbn;

static int launch_iou(const int32_t* proposals_idx, const int32_t* proposals_offset, const int64_t* instance_labels,
                      const int32_t* instance_pointnum, float* proposals_iou, int n_instance, int n_proposal,
                      const float* mask_scores, int mode, hipStream_t stream) {
    if (n_instance < 0 || n_proposal < 0) return PBN_ERR_ARG;
    if (n_instance == 0 || n_proposal == 0) return PBN_OK;
    if (!proposals_idx || !proposals_offset || !instance_labels || !instance_pointnum || !proposals_iou) return PBN_ERR_ARG;
    const size_t lds = sizeof(int) * ((size_t)n_instance + 1);
    if (lds > 160 * 1024) return PBN_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        PBN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iou, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_proposal < 32768 ? n_proposal : 32768;
    hipLaunchKernelGGL(k_iou, dim3(grid), dim3(TPB), lds, stream, proposals_idx, proposals_offset,
                       (const long long*)instance_labels, instance_pointnum, proposals_iou, n_instance, n_proposal,
                       mask_scores, mode);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}

extern "C" int pbn_get_iou(const int32_t* proposals_idx, const int32_t* proposals_offset, const int64_t* instance_labels,
                           const int32_t* instance_pointnum, float* proposals_iou, int n_instance, int n_proposal,
                           pbn_stream_t stream) {
    return launch_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, n_instance,
                      n_proposal, nullptr, -1, (hipStream_t)stream);
}

extern "C" int pbn_cal_iou_and_masklabel(const int32_t* proposals_idx, const int32_t* proposals_offset,
                                         const int64_t* instance_labels, const int32_t* instance_pointnum,
                                         float* proposals_iou, int n_instance, int n_proposal,
                                         const float* mask_scores_sigmoid, float* mask_label, int mode,
                                         pbn_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (mode != 0 && mode != 1) return PBN_OK;  // the reference kernel does nothing for other modes (:17,34)
    if ((mode == 1 && !mask_scores_sigmoid) || !mask_label) return PBN_ERR_ARG;
    int rc = launch_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, n_instance,
                        n_proposal, mask_scores_sigmoid, mode, stream);
    if (rc != PBN_OK || n_instance == 0 || n_proposal == 0) return rc;
    const int grid = n_proposal < 32768 ? n_proposal : 32768;
    hipLaunchKernelGGL(k_mask_label, dim3(grid), dim3(TPB), 0, stream, proposals_idx, proposals_offset,
                       (const long long*)instance_labels, proposals_iou, n_instance, n_proposal, mask_label);
    PBN_LAUNCH_CHECK();
    return PBN_OK;
}
