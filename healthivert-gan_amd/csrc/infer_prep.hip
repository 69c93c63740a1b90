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

// ------------------------------------------------------------------------------------------------------------------------------------------
// Volume intake / output of the inference driver on the device (reference eval_3d_sagittal_twostage.py:186-197,:208,:217,:236-239): the float64
// [H][W][Z] volumes (z fastest) are uploaded as they lie -- ONE pinned copy per volume on the host, no scan and no float32 conversion there --
//   volume_scan_kernel     counts[j][z] = number of voxels of slice z equal to ids[j] (the vertebra's z-extent and the two neighbours' > 200-pixel
//                          gates in one pass over the label volume; integer sums: exact in any order)
//   volume_slices_kernel   out[s][p] = (float)vol[p][z0 + s]: cut the z-range, convert (C cast, numpy's astype) and transpose to slices
//   volume_merge_kernel    out[p][z] = (z in range and flag[z - z0]) ? (double)src[z - z0][p] : 0: the output volume as the reference builds it
//                          (np.zeros + per-slice assignment), z fastest, float64
__global__ __launch_bounds__(256) void volume_scan_kernel(const double* __restrict__ label, long long n, int Z, double id0, double id1, double id2,
                                                          int* __restrict__ counts) {
    extern __shared__ int hist[];      // [3][Z]
    for (int i = threadIdx.x; i < 3 * Z; i += 256) hist[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = label[i];
        if (v == 0.0) continue;          // background: no id is 0 (checked on the host side)
        const int z = (int)(i % Z);
        if (v == id0) atomicAdd(hist + z, 1);
        else if (v == id1) atomicAdd(hist + Z + z, 1);
        else if (v == id2) atomicAdd(hist + 2 * Z + z, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Z; i += 256)
        if (hist[i]) atomicAdd(counts + i, hist[i]);
}
extern "C" int hv_volume_scan(const double* label, long long HW, int Z, double id0, double id1, double id2, int* counts, void* stream) {
    if (!label || !counts || HW <= 0 || Z <= 0 || Z > 2048) return HV_ERR_ARG;
    if (id0 == 0.0 || id1 == 0.0 || id2 == 0.0) return HV_ERR_ARG;      // unused ids are passed as a negative number
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)3 * Z * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return -1000 - (int)e;
    const long long n = HW * Z;
    const int grid = (int)(n / (256 * 16) < 1 ? 1 : (n / (256 * 16) > 2048 ? 2048 : n / (256 * 16)));
    hipLaunchKernelGGL(volume_scan_kernel, dim3(grid), dim3(256), (size_t)3 * Z * sizeof(int), (hipStream_t)stream, label, n, Z, id0, id1, id2, counts);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// 64 pixels x 32 slices per workgroup through LDS: 256-byte runs of doubles in, 256-byte runs of floats out
__global__ __launch_bounds__(256) void volume_slices_kernel(const double* __restrict__ vol, long long HW, int Z, int z0, int S, float* __restrict__ out) {
    __shared__ float t[64][33];
    const long long p0 = (long long)blockIdx.x * 64;
    const int s0 = blockIdx.y * 32, zz = threadIdx.x & 31, pr = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long p = p0 + pr + 8 * k;
        t[pr + 8 * k][zz] = (p < HW && s0 + zz < S) ? (float)vol[p * Z + z0 + s0 + zz] : 0.f;
    }
    __syncthreads();
    const int pl = threadIdx.x & 63, sr = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int s = s0 + sr + 4 * k;
        if (s < S && p0 + pl < HW) out[(long long)s * HW + p0 + pl] = t[pl][sr + 4 * k];
    }
}
extern "C" int hv_volume_slices(const double* vol, long long HW, int Z, int z0, int S, float* out, void* stream) {
    if (!vol || !out || HW <= 0 || Z <= 0 || z0 < 0 || S <= 0 || z0 + S > Z || HW > (1ll << 31) || S > 32 * 65535) return HV_ERR_ARG;
    hipLaunchKernelGGL(volume_slices_kernel, dim3((unsigned)hv_cdiv(HW, 64), hv_cdiv(S, 32)), dim3(256), 0, (hipStream_t)stream, vol, HW, Z, z0, S, out);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

__global__ __launch_bounds__(256) void volume_merge_kernel(const float* __restrict__ src, const int* __restrict__ flag, long long HW, int Z, int z0, int S,
                                                           double* __restrict__ out) {
    __shared__ float t[64][33];
    const long long p0 = (long long)blockIdx.x * 64;
    const int zb = blockIdx.y * 32;                 // 32 output slices zb .. zb + 31 of the FULL z range
    const int pl = threadIdx.x & 63, sr = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int z = zb + sr + 4 * k, s = z - z0;
        const bool in = z < Z && s >= 0 && s < S && p0 + pl < HW && flag[s] != 0;
        t[pl][sr + 4 * k] = in ? src[(long long)s * HW + p0 + pl] : 0.f;
    }
    __syncthreads();
    const int zz = threadIdx.x & 31, pr = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long p = p0 + pr + 8 * k;
        if (p < HW && zb + zz < Z) out[p * Z + zb + zz] = (double)t[pr + 8 * k][zz];
    }
}
extern "C" int hv_volume_merge(const float* src, const int* flag, long long HW, int Z, int z0, int S, double* out, void* stream) {
    if (!src || !flag || !out || HW <= 0 || Z <= 0 || z0 < 0 || S <= 0 || z0 + S > Z || HW > (1ll << 31)) return HV_ERR_ARG;
    hipLaunchKernelGGL(volume_merge_kernel, dim3((unsigned)hv_cdiv(HW, 64), hv_cdiv(Z, 32)), dim3(256), 0, (hipStream_t)stream, src, flag, HW, Z, z0, S, out);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
