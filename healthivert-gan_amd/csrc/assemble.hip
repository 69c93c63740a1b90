// Batch assembly on the device: the per-item arithmetic of AlignedDataset.__getitem__
// (reference data/aligned_dataset.py:204-280) on uint8 planes that already live in HBM -- row re-stacking around the masked band,
// ToTensor (/255) and Normalize((0.5,), (0.5,)) in float32.  Pure byte traffic: 4 B read, 24 B written per pixel; one lane = 4 pixels.
#include "hv_common.h"

__device__ __forceinline__ float u8_unit(unsigned v) { return (float)v / 255.0f; }             // ToTensor
__device__ __forceinline__ float u8_norm(unsigned v) { return ((float)v / 255.0f - 0.5f) / 0.5f; }   // + Normalize

template <int V>   // pixels per lane: 4 (W % 4 == 0, 4-byte loads / 16-byte stores) or 1
__global__ __launch_bounds__(256) void assemble_kernel(const hv_assemble_item* __restrict__ items, int H, int W, float* __restrict__ A,
                                                       float* __restrict__ Bi, float* __restrict__ Am, float* __restrict__ Mk,
                                                       float* __restrict__ Nv, float* __restrict__ Cm) {
    const hv_assemble_item it = items[blockIdx.y];
    const int per_row = W / V;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= H * per_row) return;
    const int r = e / per_row, c = (e - r * per_row) * V;
    const bool band = r >= it.min_x && r < it.max_x;
    const int src = r < it.min_x ? r + (it.x1 - it.min_x) : it.x2 + (r - it.max_x);    // re-stacked source row (unused inside the band)
    const long long o = ((long long)blockIdx.y * H + r) * W + c;
    const int here = r * W + c, there = (band ? r : src) * W + c;
    unsigned ct0[V], vt[V], ct1[V], nv[V], cm[V];
    if (V == 4) {
        const uchar4 a = *reinterpret_cast<const uchar4*>(it.ct + here), b = *reinterpret_cast<const uchar4*>(it.vert + here);
        const uchar4 p = *reinterpret_cast<const uchar4*>(it.ct + there), q = *reinterpret_cast<const uchar4*>(it.normal + there);
        const uchar4 s = *reinterpret_cast<const uchar4*>(it.cam + there);
        ct0[0] = a.x; ct0[1 % V] = a.y; ct0[2 % V] = a.z; ct0[3 % V] = a.w;
        vt[0] = b.x; vt[1 % V] = b.y; vt[2 % V] = b.z; vt[3 % V] = b.w;
        ct1[0] = p.x; ct1[1 % V] = p.y; ct1[2 % V] = p.z; ct1[3 % V] = p.w;
        nv[0] = q.x; nv[1 % V] = q.y; nv[2 % V] = q.z; nv[3 % V] = q.w;
        cm[0] = s.x; cm[1 % V] = s.y; cm[2 % V] = s.z; cm[3 % V] = s.w;
    } else {
        ct0[0] = it.ct[here]; vt[0] = it.vert[here]; ct1[0] = it.ct[there]; nv[0] = it.normal[there]; cm[0] = it.cam[there];
    }
    float oa[V], ob[V], om[V], ok[V], on[V], oc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        oa[i] = u8_norm(ct0[i]);
        ob[i] = u8_norm(band ? 0u : ct1[i]);
        om[i] = u8_unit(vt[i]);
        ok[i] = band ? 1.0f : 0.0f;
        on[i] = u8_unit(band ? 0u : nv[i]);
        oc[i] = u8_unit(band ? 0u : cm[i]);
    }
    if (V == 4) {
        *reinterpret_cast<float4*>(A + o) = make_float4(oa[0], oa[1 % V], oa[2 % V], oa[3 % V]);
        *reinterpret_cast<float4*>(Bi + o) = make_float4(ob[0], ob[1 % V], ob[2 % V], ob[3 % V]);
        *reinterpret_cast<float4*>(Am + o) = make_float4(om[0], om[1 % V], om[2 % V], om[3 % V]);
        *reinterpret_cast<float4*>(Mk + o) = make_float4(ok[0], ok[1 % V], ok[2 % V], ok[3 % V]);
        *reinterpret_cast<float4*>(Nv + o) = make_float4(on[0], on[1 % V], on[2 % V], on[3 % V]);
        *reinterpret_cast<float4*>(Cm + o) = make_float4(oc[0], oc[1 % V], oc[2 % V], oc[3 % V]);
    } else {
        A[o] = oa[0]; Bi[o] = ob[0]; Am[o] = om[0]; Mk[o] = ok[0]; Nv[o] = on[0]; Cm[o] = oc[0];
    }
}

extern "C" int hv_assemble_batch(const hv_assemble_item* d_items, int B, int H, int W, float* A, float* Bimg, float* A_mask, float* mask,
                                 float* normal_vert, float* CAM, void* stream) {
    if (!d_items || !A || !Bimg || !A_mask || !mask || !normal_vert || !CAM || B <= 0 || H <= 0 || W <= 0 || B > 65535) return HV_ERR_ARG;
    if ((long long)H * W >= (1ll << 30)) return HV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (W & 3) == 0 && !(((uintptr_t)A | (uintptr_t)Bimg | (uintptr_t)A_mask | (uintptr_t)mask | (uintptr_t)normal_vert | (uintptr_t)CAM) & 15);
    if (vec) hipLaunchKernelGGL((assemble_kernel<4>), dim3(hv_cdiv((long long)H * (W / 4), 256), B), dim3(256), 0, s, d_items, H, W, A, Bimg, A_mask, mask, normal_vert, CAM);
    else hipLaunchKernelGGL((assemble_kernel<1>), dim3(hv_cdiv((long long)H * W, 256), B), dim3(256), 0, s, d_items, H, W, A, Bimg, A_mask, mask, normal_vert, CAM);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
