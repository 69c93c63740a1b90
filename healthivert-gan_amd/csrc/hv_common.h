// Shared device/host helpers for libhvgan (gfx950 only: wave = 64 lanes, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hvgan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define HV_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return -1000 - (int)e__;     \
    } while (0)

extern thread_local int hv_path_note;   // set by the launcher that actually launched (hv_last_kernel_path)
// hv_set_kernel_timing: events recorded right around the MAIN kernel of the next weight-gradient call (not its slab reduction)
extern thread_local hipEvent_t hv_ev_start, hv_ev_stop;
#define HV_TIMING_BEGIN(s) do { if (hv_ev_start) (void)hipEventRecord(hv_ev_start, (s)); } while (0)
#define HV_TIMING_END(s) do { if (hv_ev_stop) (void)hipEventRecord(hv_ev_stop, (s)); hv_ev_start = hv_ev_stop = nullptr; } while (0)

static inline int hv_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float hv_act(float v, int act) {
    switch (act) {
        case HV_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case HV_ACT_RELU: return v > 0.f ? v : 0.f;
        case HV_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case HV_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case HV_ACT_CLAMP: return fminf(fmaxf(v, -1.f), 1.f);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float hv_act_grad_from_out(float y, int act) {
    switch (act) {
        case HV_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case HV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case HV_ACT_LRELU: return y > 0.f ? 1.f : 0.2f;
        case HV_ACT_SIGMOID: return y * (1.f - y);
        case HV_ACT_CLAMP: return (y > -1.f && y < 1.f) ? 1.f : 0.f;
        default: return 1.f;
    }
}

__device__ __forceinline__ float hv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float hv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x multiple of 64 (<= 1024); result valid in all threads
__device__ __forceinline__ float hv_block_sum(float v, float* red /* >= 17 floats of LDS */) {
    v = hv_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
