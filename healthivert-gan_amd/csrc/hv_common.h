// Shared device/host helpers for libhvgan (gfx950 only: wave = 64 lanes, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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
// which prepared weight tables the convolution launched last read (hv_last_weight_tables): 1 = fp32 (hv_conv_desc.w; also w1), 2 = fp16 plain rows (w_f16),
// 4 = fp16 in MFMA-fragment order (w_f16_tiled).  A launcher may over-report, never under-report: the layout pass skips what no conv of a layer reads
extern thread_local int hv_wtable_used;
#define HV_WUSE(bits) (hv_wtable_used |= (bits))
extern thread_local int hv_probe_only;  // hv_conv2d_supported: the dispatch runs without launching (the launch sites of the forms it asks about return HV_OK early)
// name of the kernel instantiation the last launcher launched, as rocprofv3 lists it (hv_last_kernel_name)
extern thread_local char hv_kname[192];
#define HV_KNAME(...) snprintf(hv_kname, sizeof(hv_kname), __VA_ARGS__)
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

// The same activations for the kernels whose operands are fp16 anyway (HV_F16 mode; results mostly stored as fp16, relative precision 4.9e-4): hardware exp2 / rcp instead of the correctly rounded expm1f
// / expf / division (~35 instructions per element: 3 us of the 17 us a 64 -> 64 channel 3x3 layer at 64x64, bs 16 took).  ELU near zero: v + v^2/2
// (|v| < 2^-8: error v^3/6 < 3e-6 relative), elsewhere exp(v) - 1 with an absolute error of one fp32 ulp of 1 (< 1.6e-5 relative there).
__device__ __forceinline__ float hv_act_fast(float v, int act) {
    switch (act) {
        case HV_ACT_ELU: return v > 0.f ? v : (v > -0.00390625f ? v + 0.5f * v * v : __builtin_amdgcn_exp2f(v * 1.44269504f) - 1.f);
        case HV_ACT_RELU: return v > 0.f ? v : 0.f;
        case HV_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case HV_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504f));
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

// ---- storage type of activation / gradient tensors: fp32, or fp16 in the HV_F16 storage mode (hv_conv_desc.x_f16 / y_f16 ...).
// MFMA kernels of the fp16 mode round their operands to fp16 when they stage them, so a tensor that is only consumed that way holds the
// same operand values either way; fp16 storage halves its bytes in HBM / L2 and drops the conversions from the staging code.
typedef unsigned int hv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int hv_u32x2 __attribute__((ext_vector_type(2)));
template <bool H> struct HvSt;           // H: elements are fp16
template <> struct HvSt<false> {
    typedef hv_u32x4 R;                  // 4 consecutive channels in flight
    static constexpr unsigned B = 4u;    // bytes per element
    static __device__ __forceinline__ R ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
    static __device__ __forceinline__ f16x4 h4(R v) {
        return (f16x4){(_Float16)__uint_as_float(v.x), (_Float16)__uint_as_float(v.y), (_Float16)__uint_as_float(v.z), (_Float16)__uint_as_float(v.w)};
    }
    static __device__ __forceinline__ float4 f4(R v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
};
template <> struct HvSt<true> {
    typedef hv_u32x2 R;
    static constexpr unsigned B = 2u;
    static __device__ __forceinline__ R ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); }
    static __device__ __forceinline__ f16x4 h4(R v) { return __builtin_bit_cast(f16x4, v); }
    static __device__ __forceinline__ float4 f4(R v) {
        const f16x4 h = __builtin_bit_cast(f16x4, v);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
};
// plain-pointer forms for the pointwise kernels (runtime flag, wave-uniform)
__device__ __forceinline__ float hv_ld1(const void* p, long long i, int half) {
    return half ? (float)reinterpret_cast<const _Float16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void hv_st1(void* p, long long i, float v, int half) {
    if (half) reinterpret_cast<_Float16*>(p)[i] = (_Float16)v; else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ float4 hv_ld4(const void* p, long long i, int half) {     // i % 4 == 0 and the base 16-B (8-B) aligned
    if (half) {
        const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p) + i);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ void hv_st4(void* p, long long i, float4 v, int half) {
    if (half) *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(p) + i) = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = v;
}

// Shared conv epilogue for one lane's 4 consecutive output channels (ch0 .. ch0+3) of one output pixel:
//   t = acc*alpha [*scale] [+bias] [+y if accumulate == 2];  v = act(t) [* act'(mul)];  y = v  or  y += v (accumulate == 1)
// yp / mp point at the pixel's channel 0 (mp = NULL: no multiplier); scale / bias are per-channel arrays or NULL.
// y_half / mul_half: the output / the multiplier tensor is stored as fp16 (yp / mp then point at _Float16 elements).
struct HvEpi {
    float alpha; int act, accumulate, vec_store, Cout;
    const float* bias; const float* scale;
    int mul_act, mul_vec;
    int y_half, mul_half;
};
// The activation / its derivative over a lane's channel QUAD with ONE wave-uniform switch (round 5: with the switch -- and the scale / bias tests -- taken per
// element, a thin layer's epilogue was hundreds of scalar branches per wave: conv_s2t_kernel issued ~820 scalar + ~680 vector instructions around 18 MFMAs).
template <bool FAST>
__device__ __forceinline__ void hv_act4(float (&t)[4], int act) {
    switch (act) {
        case HV_ACT_NONE: break;
        case HV_ACT_ELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = FAST ? hv_act_fast(t[r], HV_ACT_ELU) : hv_act(t[r], HV_ACT_ELU);
            break;
        case HV_ACT_LRELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = t[r] > 0.f ? t[r] : 0.2f * t[r];
            break;
        case HV_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = t[r] > 0.f ? t[r] : 0.f;
            break;
        default:
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = FAST ? hv_act_fast(t[r], act) : hv_act(t[r], act);
            break;
    }
}
__device__ __forceinline__ void hv_act_grad4(float (&m)[4], int act) {      // m: the producer's outputs -> its act'
    switch (act) {
        case HV_ACT_ELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = m[r] > 0.f ? 1.f : m[r] + 1.f;
            break;
        case HV_ACT_LRELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = m[r] > 0.f ? 1.f : 0.2f;
            break;
        case HV_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = m[r] > 0.f ? 1.f : 0.f;
            break;
        default:
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = hv_act_grad_from_out(m[r], act);
            break;
    }
}
// t = acc * alpha [* scale] [+ bias] over the quad (channels past Cout read the last valid parameter and are dropped by the callers)
__device__ __forceinline__ void hv_conv_affine4(const HvEpi& e, const f32x4& a, int ch0, float (&t)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = a[r] * e.alpha;
    if (e.scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] *= e.scale[min(ch0 + r, e.Cout - 1)];
    }
    if (e.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += e.bias[min(ch0 + r, e.Cout - 1)];
    }
}
// the act' factors of the quad from the multiplier tensor (1 where the channel is past Cout)
__device__ __forceinline__ void hv_conv_mul4(const HvEpi& e, int ch0, const void* __restrict__ mp, float (&f)[4]) {
    if (e.mul_vec && ch0 + 3 < e.Cout) {
        const float4 m4 = hv_ld4(mp, ch0, e.mul_half);
        f[0] = m4.x; f[1] = m4.y; f[2] = m4.z; f[3] = m4.w;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = ch0 + r < e.Cout ? hv_ld1(mp, ch0 + r, e.mul_half) : 0.f;
    }
    hv_act_grad4(f, e.mul_act);
}
// FAST: hv_act_fast (the fp16-operand kernels; chosen at compile time -- both activation bodies in one epilogue spilled the accumulators to scratch)
template <bool FAST = false>
__device__ __forceinline__ void hv_conv_epilogue4(const HvEpi& e, const f32x4& a, int ch0, void* __restrict__ yp, const void* __restrict__ mp) {
    if (ch0 >= e.Cout) return;
    float v[4];
    hv_conv_affine4(e, a, ch0, v);
    if (e.accumulate == 2) {   // pre-activation accumulate (split-K over concatenated inputs)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ch0 + r < e.Cout) v[r] += hv_ld1(yp, ch0 + r, e.y_half);
    }
    hv_act4<FAST>(v, e.act);
    if (mp) {   // hand the producer layer its pre-activation gradient: multiply by act'(its output)
        float f[4];
        hv_conv_mul4(e, ch0, mp, f);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= f[r];
    }
    if (e.vec_store && ch0 + 3 < e.Cout) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (e.accumulate == 1) {
            const float4 old = hv_ld4(yp, ch0, e.y_half);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        hv_st4(yp, ch0, o, e.y_half);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ch0 + r < e.Cout) hv_st1(yp, ch0 + r, e.accumulate == 1 ? hv_ld1(yp, ch0 + r, e.y_half) + v[r] : v[r], e.y_half);
    }
}
// The values of hv_conv_epilogue4 without its store, for epilogues that hand the tile to LDS first (accumulate 0 or 1; 2 adds y BEFORE the activation):
//   v = act(acc*alpha [*scale] [+bias]) [* act'(mul)]; channels >= Cout give 0
template <bool FAST>
__device__ __forceinline__ f32x4 hv_conv_value4(const HvEpi& e, const f32x4& a, int ch0, const void* __restrict__ mp) {
    float t[4];
    hv_conv_affine4(e, a, ch0, t);
    hv_act4<FAST>(t, e.act);
    if (mp) {
        float f[4];
        hv_conv_mul4(e, ch0, mp, f);
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] *= f[r];
    }
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ch0 + r < e.Cout ? t[r] : 0.f;
    return v;
}
// byte-wise element address of a tensor whose element size is 2 (half != 0) or 4 bytes
__device__ __forceinline__ void* hv_eptr(void* base, long long elem, int half) { return reinterpret_cast<char*>(base) + elem * (half ? 2 : 4); }
__device__ __forceinline__ const void* hv_eptr(const void* base, long long elem, int half) { return reinterpret_cast<const char*>(base) + elem * (half ? 2 : 4); }

// ---- "the last workgroup to arrive folds the partial results" -- for kernels of a FEW HUNDRED workgroups at most, and without agent-scope release fences.
// On this multi-XCD part a release fence is a write-back of the XCD's whole L2 (buffer_wbl2 sc1): with thousands of workgroups each issuing one beside tens of
// MB of ordinary stores the fences cost far more than the launch they save (round 5: the attention backward's coef fold +0.4 ms per step, the heads' bias sums
// +0.16 ms).  Agent-scope ATOMIC stores are written through to the device's coherence point themselves (sc1), so the partial results go out with hv_publish(),
// the writer waits for its stores (vmcnt counts stores on gfx9) and takes a ticket; the ONE last arriver calls hv_acquire_once() (one invalidate per kernel) and
// reads the partial results with ordinary loads (hv_collect(), an agent-scope atomic load, for single values: hipcc issues those one round trip at a time,
// 1 024 of them took 15 us).  What stays is the ticket itself: the atomics of all workgroups on one address serialise at the memory side, ~15 ns each -- 1 024
// workgroups: 8 -> 22 us for head_seed_kernel, 8 448 on 16 addresses: 85 -> 124 us for the attention kernel, both slower than the 5-us launch they replaced
// and removed again; at 113-256 workgroups (loss head, gradient check) the pattern pays.  Only values stored with hv_publish() are covered; what the last
// arriver writes is read by LATER kernels.
__device__ __forceinline__ void hv_publish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float hv_collect(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hv_acquire_once() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void hv_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool hv_take_ticket_is_last(unsigned* ticket, unsigned n) {
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1u;
}
__device__ __forceinline__ void hv_ticket_reset(unsigned* ticket) { __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// ---- carried slab folds (hv_wgrad_desc.carry): the fold of the PREVIOUS weight gradient's split-K slabs runs as extra workgroups of this weight gradient's
// launch (blockIdx.x >= the main grid's x extent, for every (y, z)) instead of a launch of its own between the two -- a dependent ~5-us node less per layer
// on the backward's chains.  Same arithmetic as wgrad_reduce_kernel<G> (conv_igemm.hip), G by slab count: bit-identical sums.
struct HvFold {
    const float* slabs; float* dw; long long n; int splits, accumulate;
    const float* bslabs; float* dbias; int nb, bias_accumulate;
};
template <int G>
__device__ __forceinline__ void hv_fold_blocks_g(const HvFold& F, int fb, int nfb, float4* sh /* >= 256 float4 of LDS */) {
    constexpr int Q = 256 / G;
    const int tid = threadIdx.x;
    const bool worker = tid < 256;                       // (512-thread kernels: the upper half only keeps the barriers company)
    const int e = tid % Q, grp = (tid & 255) / Q;
    const long long wblocks = (F.n / 4 + Q - 1) / Q;
    const long long total = wblocks + (F.dbias ? (F.nb + Q - 1) / Q : 0);
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    for (long long blk = fb; blk < total; blk += nfb) {
        const bool bias_block = blk >= wblocks;          // block-uniform
        const long long i = bias_block ? (blk - wblocks) * Q + e : (blk * Q + e) * 4;
        const bool live = worker && i < (bias_block ? (long long)F.nb : F.n);
        float4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            int k = grp;
            if (bias_block) {
                for (; k + 3 * G < F.splits; k += 4 * G)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u].x += F.bslabs[(long long)(k + u * G) * F.nb + i];
                for (; k < F.splits; k += G) acc[0].x += F.bslabs[(long long)k * F.nb + i];
            } else {
                for (; k + 3 * G < F.splits; k += 4 * G)
#pragma unroll
                    for (int u = 0; u < 4; ++u) add(acc[u], *reinterpret_cast<const float4*>(F.slabs + (long long)(k + u * G) * F.n + i));
                for (; k < F.splits; k += G) add(acc[0], *reinterpret_cast<const float4*>(F.slabs + (long long)k * F.n + i));
            }
        }
        add(acc[0], acc[1]); add(acc[2], acc[3]); add(acc[0], acc[2]);
        if (worker) sh[grp * Q + e] = acc[0];
        __syncthreads();
#pragma unroll
        for (int h = G / 2; h >= 1; h >>= 1) {
            if (worker && grp < h) add(sh[grp * Q + e], sh[(grp + h) * Q + e]);
            __syncthreads();
        }
        if (worker && grp == 0 && live) {
            float4 t = sh[e];
            if (bias_block) F.dbias[i] = F.bias_accumulate ? F.dbias[i] + t.x : t.x;
            else {
                if (F.accumulate) add(t, *reinterpret_cast<const float4*>(F.dw + i));
                *reinterpret_cast<float4*>(F.dw + i) = t;
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void hv_fold_blocks(const HvFold& F, int fb, int nfb, void* lds) {
    float4* sh = reinterpret_cast<float4*>(lds);
    if (F.splits <= 16) hv_fold_blocks_g<4>(F, fb, nfb, sh);
    else if (F.splits <= 64) hv_fold_blocks_g<16>(F, fb, nfb, sh);
    else hv_fold_blocks_g<32>(F, fb, nfb, sh);
}
// host side: the carried fold of the weight-gradient call being dispatched (hv_wgrad_desc.carry; splits == 0: none) and whether a launcher took it along
extern thread_local HvFold hv_carry;
extern thread_local int hv_carry_taken;
// x-blocks to append per (y, z) plane of a grid with `planes` planes for the carried fold (0: nothing carried)
static inline int hv_carry_blocks(int planes) {
    if (hv_carry.splits <= 0) return 0;
    const int Q = 256 / (hv_carry.splits <= 16 ? 4 : hv_carry.splits <= 64 ? 16 : 32);
    long long total = (hv_carry.n / 4 + Q - 1) / Q + (hv_carry.dbias ? (hv_carry.nb + Q - 1) / Q : 0);
    if (total > 192) total = 192;                        // (fold blocks beside the main grid: the rest strides)
    if (planes < 1) planes = 1;
    return (int)((total + planes - 1) / planes);
}
