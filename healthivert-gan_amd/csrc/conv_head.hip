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
};

// CH = 32-channel MFMA steps per pipeline item (one item's loads are in flight while the previous one is multiplied)
template <int CH, bool XH>
__global__ __launch_bounds__(256) void head_gemm_kernel(const HeadK p) {
    typedef typename HvSt<XH>::R XR;               // 4 consecutive channels as loaded (16 B of fp32 or 8 B of fp16)
    extern __shared__ __attribute__((aligned(16))) _Float16 wl[];   // [16 taps][wstride], rows >= ntaps zero
    const int tid = threadIdx.x;
    {
        const int k4 = p.K >> 2;
        for (int e = tid; e < 16 * k4; e += 256) {
            const int t = e / k4, c = (e - t * k4) * 4;
            f16x4 v = {0, 0, 0, 0};
            if (t < p.ntaps) v = *reinterpret_cast<const f16x4*>(p.w + (long long)t * p.K + c);
            *reinterpret_cast<f16x4*>(wl + t * p.wstride + c) = v;
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
    auto consume = [&](const XR (&buf)[CH * 2], int it) {
        const int t = t0 + it / chunks, c = it - (it / chunks) * chunks;
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            const f16x4 lo = HvSt<XH>::h4(buf[2 * ks]), hi = HvSt<XH>::h4(buf[2 * ks + 1]);
            const f16x8 xb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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
    HV_WUSE(2);
    k.x = d->x; k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.P = reinterpret_cast<float*>(d->workspace);
    k.M = d->B * d->H * d->W; k.tiles = hv_cdiv(k.M, 16); k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.K = d->Cin; k.ntaps = d->KH * d->KW;
    k.wstride = d->Cin + ((16 - d->Cin % 128) + 128) % 128;   // row stride = 32 B mod 256 B: the 16 taps x 4 groups of a wave's 8-byte reads
    const size_t lds = (size_t)16 * k.wstride * sizeof(_Float16);                                       // spread over all banks
    const int ch = (d->Cin % 128 == 0) ? 4 : (d->Cin % 64 == 0) ? 2 : 1;
    const int chunks = d->Cin / (32 * ch);
    // one wave streams >= 8 pipeline items, at least one block per CU
    int tpw = (8 + chunks - 1) / chunks;
    int blocks = hv_cdiv(k.tiles, 4 * tpw);
    if (blocks < 256) blocks = hv_cdiv(k.tiles, 4);
    hv_path_note = 5;
    HV_KNAME("head_gemm_kernel<%d, %s>", ch, d->x_f16 ? "true" : "false");
    if (d->x_f16) {
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
