// Slice preparation of the inference driver on the device (reference eval_3d_sagittal_twostage.py:15-30,46-98):
//   slice_components_kernel   8-connected components of (label == vertebra id) per slice, components below min_size dropped, then the
//                             statistics run_model derives from the remaining pixels: count, first / last row, sum of row indices
//   infer_prepare_kernel      bounding rows (40-row window around the mean row when the vertebra is taller than maxheight), the masked band,
//                             uint8 quantisation, row re-stacking, ToTensor / Normalize -> the four planes the generator consumes
//   select_slices_kernel      dst[s] = flag[s] ? src[s] : (keep dst[s] | 0): slices without the vertebra pass through a stage unchanged
// Integer / byte work, bit-exact against the reference (fixture G10).  One workgroup owns one slice.
#include "hv_common.h"

__device__ __forceinline__ int coherent_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Label equivalence (Hawick et al.): L[p] = index of a pixel of the same component with a smaller-or-equal index; "scan" lowers the root of
// a pixel's tree to the smallest neighbouring label (atomicMin), "analysis" points every pixel at its root; repeat until a scan changes
// nothing.  Labels live in global memory (64 K pixels x 4 B exceed the LDS) and are read with device-scope loads: the atomics resolve in L2.
// T = float (label maps of the inference driver) or unsigned char (the uint8 mask planes of the batch assembler).  keep != NULL: the filtered
// mask is written back (value where the pixel's component survived, 0 elsewhere; may be the input plane itself: it is read only by the
// first loop).
template <typename T>
__global__ __launch_bounds__(1024) void slice_components_kernel(const T* __restrict__ label, int H, int W, T value, int min_size,
                                                                int* __restrict__ Lall, int* __restrict__ Call, int* __restrict__ stats,
                                                                T* keep) {
    __shared__ int changed, s_cnt, s_min, s_max, s_sum;
    const int HW = H * W, tid = threadIdx.x;
    const T* lab = label + (long long)blockIdx.x * HW;
    int* L = Lall + (long long)blockIdx.x * HW;
    int* C = Call + (long long)blockIdx.x * HW;
    for (int p = tid; p < HW; p += 1024) {
        L[p] = lab[p] == value ? p : -1;
        C[p] = 0;
    }
    __threadfence();
    __syncthreads();
    for (int iter = 0; iter < 4096; ++iter) {      // bound: a pass that changes nothing ends the loop long before
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int p = tid; p < HW; p += 1024) {
            const int l = coherent_load(L + p);
            if (l < 0) continue;
            const int r = p / W, c = p - r * W;
            int m = l;
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    if (!dr && !dc) continue;
                    const int rr = r + dr, cc = c + dc;
                    if ((unsigned)rr < (unsigned)H && (unsigned)cc < (unsigned)W) {
                        const int lq = coherent_load(L + rr * W + cc);
                        if (lq >= 0 && lq < m) m = lq;
                    }
                }
            if (m < l) {
                atomicMin(L + l, m);
                changed = 1;
            }
        }
        __threadfence();
        __syncthreads();
        for (int p = tid; p < HW; p += 1024) {
            int l = coherent_load(L + p);
            if (l < 0) continue;
            int r = coherent_load(L + l);
            while (r != l) { l = r; r = coherent_load(L + l); }
            L[p] = r;
        }
        __threadfence();
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int p = tid; p < HW; p += 1024) {
        const int l = coherent_load(L + p);
        if (l >= 0) atomicAdd(C + l, 1);
    }
    if (tid == 0) { s_cnt = 0; s_min = 1 << 30; s_max = -1; s_sum = 0; }
    __threadfence();
    __syncthreads();
    int cnt = 0, rmin = 1 << 30, rmax = -1, rsum = 0;
    for (int p = tid; p < HW; p += 1024) {
        const int l = coherent_load(L + p);
        const bool kept = l >= 0 && coherent_load(C + l) >= min_size;
        if (kept) {
            const int r = p / W;
            ++cnt; rsum += r;
            rmin = min(rmin, r); rmax = max(rmax, r);
        }
        if (keep) keep[(long long)blockIdx.x * HW + p] = kept ? value : (T)0;
    }
    if (cnt) { atomicAdd(&s_cnt, cnt); atomicAdd(&s_sum, rsum); atomicMin(&s_min, rmin); atomicMax(&s_max, rmax); }
    __syncthreads();
    if (tid == 0) {
        int* o = stats + blockIdx.x * 4;
        o[0] = s_cnt; o[1] = s_cnt ? s_min : -1; o[2] = s_max; o[3] = s_sum;
    }
}

extern "C" size_t hv_slice_components_workspace_bytes(int S, int H, int W) {
    return (size_t)2 * S * H * W * sizeof(int);
}
extern "C" int hv_slice_components(const float* label, int S, int H, int W, float value, int min_size, int* stats, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!label || !stats || S <= 0 || H <= 0 || W <= 0) return HV_ERR_ARG;
    if ((long long)H * W > (1 << 24)) return HV_ERR_UNSUPPORTED;     // row sums stay below 2^31
    if (!workspace || workspace_bytes < hv_slice_components_workspace_bytes(S, H, W)) return HV_ERR_WORKSPACE;
    int* L = reinterpret_cast<int*>(workspace);
    hipLaunchKernelGGL(slice_components_kernel<float>, dim3(S), dim3(1024), 0, (hipStream_t)stream, label, H, W, value, min_size, L,
                       L + (long long)S * H * W, stats, (float*)nullptr);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_slice_components_u8(const void* plane, int S, int H, int W, int value, int min_size, int* stats, void* filtered,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!plane || !stats || S <= 0 || H <= 0 || W <= 0 || value <= 0 || value > 255) return HV_ERR_ARG;
    if ((long long)H * W > (1 << 24)) return HV_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < hv_slice_components_workspace_bytes(S, H, W)) return HV_ERR_WORKSPACE;
    int* L = reinterpret_cast<int*>(workspace);
    hipLaunchKernelGGL(slice_components_kernel<unsigned char>, dim3(S), dim3(1024), 0, (hipStream_t)stream, (const unsigned char*)plane, H, W,
                       (unsigned char)value, min_size, L, L + (long long)S * H * W, stats, (unsigned char*)filtered);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// count[s] = number of elements of slice s equal to `value` (the reference's `np.sum(label[:, :, z] == neighbour) > 200` gate on the ORIGINAL
// labels, eval_3d_sagittal_twostage.py:208,217): one workgroup per slice, integer sums in a fixed order
__global__ __launch_bounds__(1024) void slice_count_kernel(const float* __restrict__ label, long long per, float value, int* __restrict__ count) {
    __shared__ int red[16];
    const float* lab = label + (long long)blockIdx.x * per;
    int c = 0;
    for (long long i = threadIdx.x; i < per; i += 1024) c += lab[i] == value ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += red[k];
        count[blockIdx.x] = t;
    }
}
extern "C" int hv_slice_count(const float* label, int S, long long per_slice, float value, int* count, void* stream) {
    if (!label || !count || S <= 0 || per_slice <= 0 || per_slice > (1ll << 30)) return HV_ERR_ARG;
    hipLaunchKernelGGL(slice_count_kernel, dim3(S), dim3(1024), 0, (hipStream_t)stream, label, per_slice, value, count);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

struct PrepK {
    const float* ct; const float* cam; const int* stats; const int* selected;
    float* ct_masked; float* ori_ct; float* mask; float* cam_out;
    long long* x1; long long* x2; long long* height; int* valid;
    int H, W, maxheight;
};

__device__ __forceinline__ unsigned quant_u8(float v) { return (unsigned)(int)v & 255u; }   // numpy astype(uint8) of an in-range float: truncation

__global__ __launch_bounds__(256) void infer_prepare_kernel(const PrepK k) {
    const int s = blockIdx.y, H = k.H, W = k.W;
    const int* st = k.stats + s * 4;
    const bool valid = st[0] > 0 && (!k.selected || k.selected[s]);
    int x1 = 0, x2 = 0, height = H, min_x = 0, max_x = 0;
    if (valid) {
        x1 = st[1]; x2 = st[2];
        height = x2 - x1;
        if (height > k.maxheight) {       // 40-row window around the mean row (:58-61); `height` keeps the full extent
            x1 = st[3] / st[0] - 20;
            x2 = x1 + 40;
        }
        const int mask_x = (x1 + x2) >> 1, h2 = k.maxheight;     // floor division like Python's // (x1 + x2 may be negative only when invalid)
        if (mask_x <= h2 / 2) min_x = 0;
        else if (2 * (H - mask_x) <= h2) min_x = H - h2;
        else min_x = mask_x - h2 / 2;
        max_x = min_x + h2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // slices without the vertebra: rows that keep the re-compositing kernels inside the image (their result is discarded)
        k.x1[s] = x1; k.x2[s] = x2; k.height[s] = height; k.valid[s] = valid ? 1 : 0;
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int r = p / W;
    const long long o = (long long)s * H * W + p;
    const float* ct = k.ct + (long long)s * H * W;
    const float* cam = k.cam + (long long)s * H * W;
    k.ori_ct[o] = ((float)quant_u8(ct[p]) / 255.0f - 0.5f) / 0.5f;
    unsigned cq = 0, mq = 0;
    float band = 0.f;
    if (valid) {
        band = (r >= min_x && r <= max_x) ? 1.0f : 0.0f;                // one row more than the data band (:75)
        int src = -1;
        if (r < min_x) src = r + (x1 - min_x);
        else if (r >= max_x) src = x2 + (r - max_x);
        if ((unsigned)src < (unsigned)H) { cq = quant_u8(ct[src * W + (p - r * W)]); mq = quant_u8(cam[src * W + (p - r * W)]); }
    }
    k.ct_masked[o] = ((float)cq / 255.0f - 0.5f) / 0.5f;
    k.mask[o] = band;
    k.cam_out[o] = (float)mq / 255.0f;
}

extern "C" int hv_infer_prepare(const float* ct, const float* cam, const int* stats, const int* selected, int S, int H, int W, int maxheight,
                                float* ct_masked, float* ori_ct, float* mask, float* cam_out, long long* x1, long long* x2, long long* height,
                                int* valid, void* stream) {
    if (!ct || !cam || !stats || !ct_masked || !ori_ct || !mask || !cam_out || !x1 || !x2 || !height || !valid) return HV_ERR_ARG;
    if (S <= 0 || S > 65535 || H <= 0 || W <= 0 || maxheight <= 0 || maxheight > H || (maxheight & 1)) return HV_ERR_ARG;
    PrepK k{ct, cam, stats, selected, ct_masked, ori_ct, mask, cam_out, x1, x2, height, valid, H, W, maxheight};
    hipLaunchKernelGGL(infer_prepare_kernel, dim3(hv_cdiv((long long)H * W, 256), S), dim3(256), 0, (hipStream_t)stream, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

__global__ void select_slices_kernel(const int* __restrict__ flag, const float* __restrict__ src, float* __restrict__ dst, long long per, int keep) {
    const int s = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const long long o = (long long)s * per + i;
    if (flag[s]) dst[o] = src[o];
    else if (!keep) dst[o] = 0.f;
}
extern "C" int hv_select_slices(const int* flag, const float* src, float* dst, int S, long long per_slice, int keep_unflagged, void* stream) {
    if (!flag || !src || !dst || S <= 0 || S > 65535 || per_slice <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(select_slices_kernel, dim3(hv_cdiv(per_slice, 256), S), dim3(256), 0, (hipStream_t)stream, flag, src, dst, per_slice, keep_unflagged);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
