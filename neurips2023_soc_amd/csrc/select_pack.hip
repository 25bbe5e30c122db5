// K26: query selection + record packing of the inference loop in one launch (reference infer_refytb.py:216-226:
//   pred_scores = pred_cls.sigmoid().mean(0); max over classes; argmax over queries; pred_masks of that query) for every clip
// of a launch group:
//   record[b] = [ idx_b, pred_cls[:, b, :, 0] (T*Q logits), pred_masks[:, b, idx_b] (T*HW logits) ]
//   idx_b = argmax_q max_k mean_t sigmoid(pred_cls[t, b, q, k])          (first maximum, as torch.argmax)
// Replaces sigmoid / mean / max / argmax / index_select / three slice copies per clip (nine short torch launches, 36 per group
// of four on the tail's branch).  Every workgroup of a clip works out idx_b for itself (T*Q*K = 160 values) and then copies
// its share of the selected masks: no second launch, no grid-wide hand-off.
#include "soc_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void select_pack_kernel(
    const float* __restrict__ cls, long cls_st, long cls_sb, long cls_sq, const float* __restrict__ masks,
    float* __restrict__ records, long rec_stride, int T, int B, int Q, int K, long HW) {
    __shared__ float score[256];
    __shared__ int best;
    const int b = blockIdx.y;
    const float* cb = cls + (long)b * cls_sb;
    // one thread per query (Q <= 256): max over classes of the mean over frames of sigmoid(logit), summed in frame order
    float s = -INFINITY;
    if ((int)threadIdx.x < Q) {
        for (int k = 0; k < K; ++k) {
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc += 1.f / (1.f + expf(-cb[(long)t * cls_st + (long)threadIdx.x * cls_sq + k]));
            s = fmaxf(s, acc / (float)T);
        }
    }
    score[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int arg = 0;
        for (int q = 1; q < Q; ++q)
            if (score[q] > score[arg]) arg = q;          // strict: the first maximum wins, as torch.argmax
        best = arg;
    }
    __syncthreads();
    const int idx = best;
    float* rec = records + (long)b * rec_stride;
    const long n_cls = (long)T * Q, n_mask = (long)T * HW;
    const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsize = (long)gridDim.x * blockDim.x;
    if (gtid == 0) rec[0] = (float)idx;
    for (long i = gtid; i < n_cls; i += gsize) {
        const long t = i / Q, q = i - t * Q;
        rec[1 + i] = cb[t * cls_st + q * cls_sq];        // class 0, as the reference's drivers read it ([..., 0])
    }
    // selected masks: pred_masks [T, B, Q, HW] contiguous
    for (long i = gtid; i < n_mask; i += gsize) {
        const long t = i / HW, p = i - t * HW;
        rec[1 + n_cls + i] = masks[(((long)t * B + b) * Q + idx) * HW + p];
    }
}

}  // namespace

extern "C" int soc_select_pack_f32(const float* pred_cls, long cls_stride_t, long cls_stride_b, long cls_stride_q,
                                   const float* pred_masks, float* records, long record_stride, int T, int B, int Q, int K,
                                   long HW, void* stream) {
    if (T <= 0 || B <= 0 || Q <= 0 || K <= 0 || HW <= 0) return SOC_EINVAL;
    if (!pred_cls || !pred_masks || !records) return SOC_EINVAL;
    if (Q > 256 || record_stride < 1 + (long)T * Q + (long)T * HW) return SOC_EUNSUPPORTED;
    const long n = (long)T * HW;
    int blocks = (int)((n + 255) / 256 / 4);             // ~4 elements per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(select_pack_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, pred_cls, cls_stride_t,
                       cls_stride_b, cls_stride_q, pred_masks, records, record_stride, T, B, Q, K, HW);
    return soc_check_launch();
}
