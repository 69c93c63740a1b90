// Convolutions with ONE output channel and many input channels, fp16 mode: the PatchGAN logits (512 -> 1, 4x4) and the data
// gradient of the discriminators' 1-channel stem (64 -> 1, 4x4, stride 2, transposed).
// A 16-wide MFMA output tile wastes 15/16 of its columns on Cout = 1 and the halo-tiled kernels re-read the input patch from LDS
// once per tap; both layers are bound by reading their input once.  Here the TAPS are the GEMM's second dimension:
//   head_gemm_kernel    P[pixel][tap] = sum_c x[pixel][c] * w[tap][c]          (M = taps <= 16, N = input pixels, K = Cin; MFMA)
//   head_tapsum_kernel  y[n,ho,wo]    = epilogue( sum_{r,s} P[n, hi(ho,r), wi(wo,s)][r*KW+s] )   (index rule of hv_conv_desc)
// x is streamed exactly once, straight from global memory into MFMA operand registers (a lane's two 16-byte loads per 32-channel
// step are converted to fp16 in registers; the same channel permutation is applied to the filter fragment, so no LDS transpose
// is needed); P (64 B per input pixel) goes through the caller's workspace.
#include "hv_common.h"

struct HeadK {
    const void* x; const _Float16* w; float* P;        // x: fp32, or fp16 elements with the XH instantiations
    int M, tiles, x_ld, x_coff, K, ntaps, wstride;
    // XN instantiations (hv_conv_desc.xn_*): x is the raw input of a normalisation + activation; the operand is made from it in registers
    const float* xn_stats; const float* xn_gamma; const float* xn_beta; _Float16* xn_out;
    int xn_groups, xn_ppg, xn_act, xn_out_ld, xn_out_coff;      // xn_ppg: pixels per statistics group
};

// CH = 32-channel MFMA steps per pipeline item (one item's loads are in flight while the previous one is multiplied)
// XN (fp16 x only): the operand of pixel n, channel c is act(((x - mean) * rstd) * gamma + beta) rounded to fp16 -- norm_apply_kernel's arithmetic in its
// order -- with the four per-channel constants read from an LDS table [group][channel] (the 16 lanes of a channel quad share an address: broadcast
// reads); the lane's 8-byte pieces of the normalised map go out to xn_out as they are made (the separate normalisation pass read x and wrote that map,
// this kernel then read it back: one read and one launch per forward are gone)
template <int CH, bool XH, int XN = 0>      // XN: 0 plain, 1 normalisation without affine parameters (InstanceNorm), 2 with (BatchNorm); LeakyReLU(0.2) behind it
__global__ __launch_bounds__(256) void head_gemm_kernel(const HeadK p) {
    typedef typename HvSt<XH>::R XR;               // 4 consecutive channels as loaded (16 B of fp32 or 8 B of fp16)
    static_assert(XN == 0 || XH, "the fused normalisation reads fp16 x");
    extern __shared__ __attribute__((aligned(16))) _Float16 wl[];   // [16 taps][wstride], rows >= ntaps zero; XN: + [groups][K] float4 {mean, rstd, gamma, beta}
    const int tid = threadIdx.x;
    {
        const int k4 = p.K >> 2;
#pragma unroll 4
        for (int e = tid; e < 16 * k4; e += 256) {
            const int t = e / k4, c = (e - t * k4) * 4;
            f16x4 v = {0, 0, 0, 0};
            if (t < p.ntaps) v = *reinterpret_cast<const f16x4*>(p.w + (long long)t * p.K + c);
            *reinterpret_cast<f16x4*>(wl + t * p.wstride + c) = v;
        }
    }
    float4* pn = reinterpret_cast<float4*>(wl + 16 * p.wstride);
    if constexpr (XN != 0) {      // four entries per thread and round, every load of a round in flight before the first LDS store
        const int ne = p.xn_groups * p.K;
        for (int e0 = 0; e0 < ne; e0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = min(e0 + tid + 256 * i, ne - 1), gq = e / p.K, c = e - gq * p.K;
                v[i] = make_float4(p.xn_stats[(long long)gq * 2 * p.K + c], p.xn_stats[(long long)gq * 2 * p.K + p.K + c], XN == 2 ? p.xn_gamma[c] : 1.f, XN == 2 ? p.xn_beta[c] : 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (e0 + tid + 256 * i < ne) pn[e0 + tid + 256 * i] = v[i];
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int col = lane & 15, g = lane >> 4;
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int tpw = (p.tiles + nwaves - 1) / nwaves;
    const int t0 = gw * tpw, t1 = min(t0 + tpw, p.tiles);
    if (t0 >= t1) return;
    const int chunks = p.K / (32 * CH);
    const int nitems = (t1 - t0) * chunks;
    const _Float16* wrow = wl + col * p.wstride + g * 4;

    XR bufA[CH * 2], bufB[CH * 2];
    auto issue = [&](XR (&buf)[CH * 2], int it) {
        const int t = t0 + it / chunks, c = it - (it / chunks) * chunks;
        const int n = min(t * 16 + col, p.M - 1);
        const char* xp = reinterpret_cast<const char*>(p.x) + ((long long)n * p.x_ld + p.x_coff + c * (32 * CH) + g * 4) * HvSt<XH>::B;
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            buf[2 * ks] = *reinterpret_cast<const XR*>(xp + (ks * 32) * HvSt<XH>::B);
            buf[2 * ks + 1] = *reinterpret_cast<const XR*>(xp + (ks * 32 + 16) * HvSt<XH>::B);
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // XN: the normalised map leaves through buffer stores whose out-of-range lanes (pixels past the end, or no xn_out at all) are dropped by the range check:
    // a store under a branch makes the compiler wait for ALL memory operations at the join -- the next item's loads included
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t osrc =
        __builtin_amdgcn_make_buffer_rsrc(p.xn_out, 0, (XN != 0 && p.xn_out) ? (unsigned)((long long)p.M * p.xn_out_ld * 2) : 0u, 0x00020000);
    auto consume = [&](const XR (&buf)[CH * 2], int it) {
        const int t = t0 + it / chunks, c = it - (it / chunks) * chunks;
        f16x4 lo[CH], hi[CH];
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) { lo[ks] = HvSt<XH>::h4(buf[2 * ks]); hi[ks] = HvSt<XH>::h4(buf[2 * ks + 1]); }
        if constexpr (XN != 0) {
            const int n = min(t * 16 + col, p.M - 1);
            const float4* pq = pn + (n / p.xn_ppg) * p.K + c * (32 * CH) + g * 4;
            const unsigned ooff = t * 16 + col < p.M ? (unsigned)(((long long)n * p.xn_out_ld + p.xn_out_coff + c * (32 * CH) + g * 4) * 2) : 0x80000000u;
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                float4 a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = pq[ks * 32 + e]; b[e] = pq[ks * 32 + 16 + e]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // norm_apply_kernel's arithmetic in its order; LeakyReLU as hv_act writes it
                    float v = ((float)lo[ks][e] - a[e].x) * a[e].y, u = ((float)hi[ks][e] - b[e].x) * b[e].y;
                    if constexpr (XN == 2) { v = v * a[e].z + a[e].w; u = u * b[e].z + b[e].w; }
                    lo[ks][e] = (_Float16)(v > 0.f ? v : 0.2f * v);
                    hi[ks][e] = (_Float16)(u > 0.f ? u : 0.2f * u);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(hv_u32x2, lo[ks]), osrc, ooff + ks * 64, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(hv_u32x2, hi[ks]), osrc, ooff + ks * 64 + 32, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            const f16x8 xb = {lo[ks][0], lo[ks][1], lo[ks][2], lo[ks][3], hi[ks][0], hi[ks][1], hi[ks][2], hi[ks][3]};
            const _Float16* wp = wrow + (c * CH + ks) * 32;
            const f16x4 w0 = *reinterpret_cast<const f16x4*>(wp), w1 = *reinterpret_cast<const f16x4*>(wp + 16);
            const f16x8 wa = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb, acc, 0, 0, 0);
        }
        if (c == chunks - 1) {   // rows of acc = taps 4g..4g+3, column = pixel
            const int n = t * 16 + col;
            if (n < p.M) *reinterpret_cast<float4*>(p.P + (long long)n * 16 + g * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    issue(bufA, 0);
    for (int it = 0; it < nitems; it += 2) {
        if (it + 1 < nitems) issue(bufB, it + 1);
        consume(bufA, it);
        if (it + 1 < nitems) {
            if (it + 2 < nitems) issue(bufA, it + 2);
            consume(bufB, it + 1);
        }
    }
}

struct TapSumK {
    const float* P; const float* bias; void* y; const void* mul_src; int y_half, mul_half;
    int B, H, W, KH, KW, stride, pad, transposed, Ho, Wo, y_ld, y_coff;
    float alpha; int act, accumulate, mul_ld, mul_coff, mul_act;
};

__device__ __forceinline__ void head_store(const TapSumK& p, long long o, float acc) {
    const long long yi = o * p.y_ld + p.y_coff;
    float t = acc * p.alpha;
    if (p.bias) t += p.bias[0];
    if (p.accumulate == 2) t += hv_ld1(p.y, yi, p.y_half);
    t = hv_act(t, p.act);
    if (p.mul_src) t *= hv_act_grad_from_out(hv_ld1(p.mul_src, o * p.mul_ld + p.mul_coff, p.mul_half), p.mul_act);
    hv_st1(p.y, yi, p.accumulate == 1 ? hv_ld1(p.y, yi, p.y_half) + t : t, p.y_half);
}

// any filter / stride
__global__ __launch_bounds__(256) void head_tapsum_kernel(const TapSumK p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Ho * p.Wo;
    if (o >= total) return;
    const int wo = (int)(o % p.Wo), ho = (int)((o / p.Wo) % p.Ho), n = (int)(o / ((long long)p.Wo * p.Ho));
    const float* Pn = p.P + (long long)n * p.H * p.W * 16;
    float acc = 0.f;
    for (int r = 0; r < p.KH; ++r) {
        int hi;
        if (!p.transposed) hi = ho * p.stride - p.pad + r;
        else {
            const int vh = ho + p.pad - r;
            if (vh < 0 || (vh % p.stride)) continue;
            hi = vh / p.stride;
        }
        if ((unsigned)hi >= (unsigned)p.H) continue;
        for (int s = 0; s < p.KW; ++s) {
            int wi;
            if (!p.transposed) wi = wo * p.stride - p.pad + s;
            else {
                const int vw = wo + p.pad - s;
                if (vw < 0 || (vw % p.stride)) continue;
                wi = vw / p.stride;
            }
            if ((unsigned)wi >= (unsigned)p.W) continue;
            acc += Pn[((long long)hi * p.W + wi) * 16 + r * p.KW + s];
        }
    }
    head_store(p, o, acc);
}

// KS x KS filter, stride S known at compile time: every table read of an output is issued before the first one is used (clamped
// address + zero weight instead of a branch); the transposed form only visits the KS/S taps per axis whose parity matches.
template <int KS, int S, bool TR>
__global__ __launch_bounds__(256) void head_tapsum_fixed_kernel(const TapSumK p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Ho * p.Wo;
    if (o >= total) return;
    const int wo = (int)(o % p.Wo), ho = (int)((o / p.Wo) % p.Ho), n = (int)(o / ((long long)p.Wo * p.Ho));
    const float* Pn = p.P + (long long)n * p.H * p.W * 16;
    constexpr int NT = TR ? (KS + S - 1) / S : KS;      // taps visited per axis
    int hi[NT], wi[NT], rr[NT], ss[NT];
    bool hok[NT], wok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (!TR) {
            rr[j] = ss[j] = j;
            hi[j] = ho * S - p.pad + j; wi[j] = wo * S - p.pad + j;
            hok[j] = (unsigned)hi[j] < (unsigned)p.H; wok[j] = (unsigned)wi[j] < (unsigned)p.W;
        } else {   // r = r0 + S*j with r0 = (ho + pad) mod S;  hi = (ho + pad - r) / S exactly
            const int vh = ho + p.pad, vw = wo + p.pad;
            rr[j] = vh % S + S * j; ss[j] = vw % S + S * j;
            hi[j] = vh / S - j; wi[j] = vw / S - j;
            hok[j] = rr[j] < KS && (unsigned)hi[j] < (unsigned)p.H; wok[j] = ss[j] < KS && (unsigned)wi[j] < (unsigned)p.W;
        }
        hi[j] = min(max(hi[j], 0), p.H - 1); wi[j] = min(max(wi[j], 0), p.W - 1);
        rr[j] = min(rr[j], KS - 1); ss[j] = min(ss[j], KS - 1);
    }
    float v[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) v[a][b] = Pn[(hi[a] * p.W + wi[b]) * 16 + rr[a] * KS + ss[b]];
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc += (hok[a] && wok[b]) ? v[a][b] : 0.f;
    head_store(p, o, acc);
}

static bool head_ok(const hv_conv_desc* d) {
    static const int enabled = getenv("HV_HEAD") ? atoi(getenv("HV_HEAD")) : 1;   // A/B knob
    return enabled && d->Cout == 1 && d->precision == HV_F16 && d->w_f16 && d->dil == 1 && !d->w_bstride && !d->ch_scale && !d->in_shift &&
           d->KH * d->KW <= 16 && (d->Cin & 31) == 0 && d->Cin >= 64 && d->Cin <= 1024 && d->stride <= 2 && !(d->x_ld & 3) && !(d->x_coff & 3) &&
           !((uintptr_t)d->x & 15) && !((uintptr_t)d->w_f16 & 7);
}

// bytes of hv_conv_desc.workspace the single-output-channel path needs for this convolution (0: it does not apply)
size_t hv_conv2d_g4_stats_floats(const hv_conv_desc* d, int* nparts);   // conv_g4.hip
extern "C" size_t hv_conv2d_stats_parts(const hv_conv_desc* d) {
    int parts = 0;
    if (!d || !hv_conv2d_g4_stats_floats(d, &parts)) return 0;
    return (size_t)parts;
}

extern "C" size_t hv_conv2d_workspace_bytes(const hv_conv_desc* d) {
    if (!d || !head_ok(d)) return 0;
    return (size_t)d->B * d->H * d->W * 16 * sizeof(float);
}

int hv_conv2d_head(const hv_conv_desc* d, hipStream_t s) {
    if (!head_ok(d)) return HV_ERR_UNSUPPORTED;
    const size_t need = hv_conv2d_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 15)) return HV_ERR_UNSUPPORTED;
    if (!d->transposed && ((d->H + 2 * d->pad - d->KH) / d->stride + 1 != d->Ho || (d->W + 2 * d->pad - d->KW) / d->stride + 1 != d->Wo))
        return HV_ERR_ARG;
    HeadK k;
    k.x = d->x; k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.P = reinterpret_cast<float*>(d->workspace);
    k.M = d->B * d->H * d->W; k.tiles = hv_cdiv(k.M, 16); k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.K = d->Cin; k.ntaps = d->KH * d->KW;
    k.wstride = d->Cin + ((16 - d->Cin % 128) + 128) % 128;   // row stride = 32 B mod 256 B: the 16 taps x 4 groups of a wave's 8-byte reads
    size_t lds = (size_t)16 * k.wstride * sizeof(_Float16);                                             // spread over all banks
    k.xn_stats = nullptr; k.xn_gamma = k.xn_beta = nullptr; k.xn_out = nullptr; k.xn_groups = 1; k.xn_ppg = k.M; k.xn_act = 0; k.xn_out_ld = k.xn_out_coff = 0;
    if (d->xn_stats) {      // normalisation + activation of the input at staging (hv_conv_desc.xn_*): fp16 x, whole 128-channel items, the constants' table in LDS
        const int G = d->xn_groups > 0 ? d->xn_groups : 1;
        if (!d->x_f16 || d->transposed || (d->Cin & 127) || d->B % G || (d->xn_gamma && !d->xn_beta) || d->xn_act != HV_ACT_LRELU)      // (the PatchGAN's LeakyReLU(0.2) only)
            return HV_ERR_UNSUPPORTED;
        if (d->xn_out && ((d->xn_out_ld & 3) || (d->xn_out_coff & 3) || ((uintptr_t)d->xn_out & 7) || d->xn_out_ld < d->xn_out_coff + d->Cin ||
                          (long long)d->B * d->H * d->W * d->xn_out_ld * 2 >= (1ll << 31)))
            return HV_ERR_UNSUPPORTED;
        lds += (size_t)G * d->Cin * sizeof(float4);
        if (lds > 64 * 1024) return HV_ERR_UNSUPPORTED;      // (InstanceNorm at a large batch: the caller keeps the separate pass)
        k.xn_stats = d->xn_stats; k.xn_gamma = d->xn_gamma; k.xn_beta = d->xn_beta; k.xn_out = reinterpret_cast<_Float16*>(d->xn_out);
        k.xn_groups = G; k.xn_ppg = (d->B / G) * d->H * d->W; k.xn_act = d->xn_act; k.xn_out_ld = d->xn_out_ld; k.xn_out_coff = d->xn_out_coff;
    }
    if (hv_probe_only) return HV_OK;          // hv_conv2d_supported: this path would take the descriptor
    HV_WUSE(2);
    const int ch = (d->Cin % 128 == 0) ? 4 : (d->Cin % 64 == 0) ? 2 : 1;
    const int chunks = d->Cin / (32 * ch);
    // one wave streams >= 8 pipeline items, at least one block per CU
    int tpw = (8 + chunks - 1) / chunks;
    int blocks = hv_cdiv(k.tiles, 4 * tpw);
    if (blocks < 256) blocks = hv_cdiv(k.tiles, 4);
    static const int xn_tpw = getenv("HV_HEAD_XN_TPW") ? atoi(getenv("HV_HEAD_XN_TPW")) : 0;      // tuning knob: tiles per wave of the XN form (0: as the plain form)
    if (k.xn_stats && xn_tpw > 0) blocks = hv_cdiv(k.tiles, 4 * xn_tpw);
    hv_path_note = 5;
    HV_KNAME("head_gemm_kernel<%d, %s>", ch, d->x_f16 ? "true" : "false");
    if (k.xn_stats) {
        HV_KNAME("head_gemm_kernel<4, true, %d>", k.xn_gamma ? 2 : 1);
        if (k.xn_gamma) hipLaunchKernelGGL((head_gemm_kernel<4, true, 2>), dim3(blocks), dim3(256), lds, s, k);
        else hipLaunchKernelGGL((head_gemm_kernel<4, true, 1>), dim3(blocks), dim3(256), lds, s, k);
    } else if (d->x_f16) {
        if (ch == 4) hipLaunchKernelGGL((head_gemm_kernel<4, true>), dim3(blocks), dim3(256), lds, s, k);
        else if (ch == 2) hipLaunchKernelGGL((head_gemm_kernel<2, true>), dim3(blocks), dim3(256), lds, s, k);
        else hipLaunchKernelGGL((head_gemm_kernel<1, true>), dim3(blocks), dim3(256), lds, s, k);
    } else {
        if (ch == 4) hipLaunchKernelGGL((head_gemm_kernel<4, false>), dim3(blocks), dim3(256), lds, s, k);
        else if (ch == 2) hipLaunchKernelGGL((head_gemm_kernel<2, false>), dim3(blocks), dim3(256), lds, s, k);
        else hipLaunchKernelGGL((head_gemm_kernel<1, false>), dim3(blocks), dim3(256), lds, s, k);
    }
    HV_LAUNCH_CHECK();
    TapSumK t;
    t.P = k.P; t.bias = d->bias; t.y = d->y; t.mul_src = d->mul_src; t.y_half = d->y_f16 ? 1 : 0; t.mul_half = d->mul_f16 ? 1 : 0;
    t.B = d->B; t.H = d->H; t.W = d->W; t.KH = d->KH; t.KW = d->KW; t.stride = d->stride; t.pad = d->pad; t.transposed = d->transposed;
    t.Ho = d->Ho; t.Wo = d->Wo; t.y_ld = d->y_ld; t.y_coff = d->y_coff;
    t.alpha = d->alpha; t.act = d->act; t.accumulate = d->accumulate; t.mul_ld = d->mul_ld; t.mul_coff = d->mul_coff; t.mul_act = d->mul_act;
    const long long total = (long long)d->B * d->Ho * d->Wo;
    const dim3 tg(hv_cdiv(total, 256));
    const bool sq = d->KH == d->KW;
    if (sq && d->KH == 4 && d->stride == 1 && !d->transposed) hipLaunchKernelGGL((head_tapsum_fixed_kernel<4, 1, false>), tg, dim3(256), 0, s, t);
    else if (sq && d->KH == 3 && d->stride == 1 && !d->transposed) hipLaunchKernelGGL((head_tapsum_fixed_kernel<3, 1, false>), tg, dim3(256), 0, s, t);
    else if (sq && d->KH == 4 && d->stride == 2 && d->transposed) hipLaunchKernelGGL((head_tapsum_fixed_kernel<4, 2, true>), tg, dim3(256), 0, s, t);
    else if (sq && d->KH == 3 && d->stride == 1 && d->transposed) hipLaunchKernelGGL((head_tapsum_fixed_kernel<3, 1, true>), tg, dim3(256), 0, s, t);
    else hipLaunchKernelGGL(head_tapsum_kernel, tg, dim3(256), 0, s, t);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// (A write-bound fp32 VALU kernel for the matching data gradient -- 4 padded gradient channels into 512 -- was measured at 55 us against
// 26 us for the gather MFMA kernel: 64 multiply-adds per output at one wave64 VALU instruction per 4 cycles is ~25 us before any memory
// traffic.  Not kept.)
