// Forward convolution with a single output channel (the generator's four 1-channel heads, the PatchGAN logit layer):
// a 16-wide MFMA tile wastes 15/16 of its work on these and they are HBM-bound, so the forward runs as a plain fp32
// VALU kernel (exact in both precision modes):
//   y[p] = act(alpha * sum_{tap,c} x[p+tap][c] * w[tap][c] + b)
// (VALU versions of the matching data / weight gradients were measured 2-6x SLOWER than the channel-padded MFMA tiles
// and are not kept: round-1 notes in DESIGN.md.)
#include "hv_common.h"

struct NarrowK {
    const float* x; const float* w; const float* bias; float* y;
    int B, H, W, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int KH, KW, stride, pad, transposed;
    int Ho, Wo, y_ld, y_coff, Cout;
    float alpha; int act, accumulate;
};

__device__ __forceinline__ bool narrow_tap(const NarrowK& p, int oy, int ox, int r, int s, int& hi, int& wi) {
    if (!p.transposed) {
        hi = oy * p.stride - p.pad + r;
        wi = ox * p.stride - p.pad + s;
    } else {
        const int vh = oy + p.pad - r, vw = ox + p.pad - s;
        if (vh < 0 || vw < 0 || (vh % p.stride) || (vw % p.stride)) return false;
        hi = vh / p.stride;
        wi = vw / p.stride;
    }
    return (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
}

__device__ __forceinline__ void narrow_store(const NarrowK& p, long long pix, float acc) {
    float* yp = p.y + pix * p.y_ld + p.y_coff;
    float t = acc * p.alpha;
    if (p.bias) t += p.bias[0];
    if (p.accumulate == 2) t += *yp;
    t = hv_act(t, p.act);
    *yp = p.accumulate == 1 ? *yp + t : t;
}

// LPP lanes cooperate on one output pixel: lanes split the channels (16 B each), taps are looped.
template <int LPP>
__global__ __launch_bounds__(256) void narrow_fwd_kernel(const NarrowK p) {
    const long long n = (long long)p.B * p.Ho * p.Wo;
    const int sub = threadIdx.x % LPP;
    constexpr int PPB = 256 / LPP;   // pixels per block pass
    for (long long i0 = (long long)blockIdx.x * PPB; i0 < n; i0 += (long long)gridDim.x * PPB) {
        const long long i = i0 + threadIdx.x / LPP;
        float acc = 0.f;
        if (i < n) {
            const int ox = (int)(i % p.Wo);
            const long long r2 = i / p.Wo;
            const int oy = (int)(r2 % p.Ho), b = (int)(r2 / p.Ho);
            const float* ximg = p.x + (long long)b * p.img_stride + p.x_coff;
            for (int r = 0; r < p.KH; ++r)
                for (int s = 0; s < p.KW; ++s) {
                    int hi, wi;
                    if (!narrow_tap(p, oy, ox, r, s, hi, wi)) continue;
                    const float* xp = ximg + (long long)((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld;
                    const float* wp = p.w + (r * p.KW + s) * p.Cin;
                    for (int c = sub * 4; c < p.Cin; c += LPP * 4) {
                        const float4 xv = *reinterpret_cast<const float4*>(xp + c), wv = *reinterpret_cast<const float4*>(wp + c);
                        acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                    }
                }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0 && i < n) narrow_store(p, i, acc);
    }
}

int hv_conv2d_narrow(const hv_conv_desc* d, hipStream_t s) {
    if (d->w_bstride || d->ch_scale || d->dil != 1 || d->stride > 2) return HV_ERR_UNSUPPORTED;
    NarrowK k;
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    k.x = d->x; k.w = d->w; k.bias = d->bias; k.y = d->y;
    k.B = d->B; k.H = d->H; k.W = d->W; k.in_shift = d->in_shift; k.Wp = Wp; k.img_stride = Hp * Wp * d->x_ld;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.transposed = d->transposed;
    k.Ho = d->Ho; k.Wo = d->Wo; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Cout = d->Cout;
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    const long long npix = (long long)d->B * d->Ho * d->Wo;
    if (d->Cout == 1 && (d->Cin & 3) == 0 && !(d->x_ld & 3) && !(d->x_coff & 3) && !((uintptr_t)d->x & 15) && !((uintptr_t)d->w & 15)) {
        const int c4 = d->Cin / 4;
        const int lpp = c4 >= 64 ? 64 : c4 >= 16 ? 16 : c4 >= 4 ? 4 : 1;
        long long nb = (npix + 256 / lpp - 1) / (256 / lpp);
        if (nb > 65536) nb = 65536;
        hv_path_note = 1;
        if (lpp == 64) hipLaunchKernelGGL((narrow_fwd_kernel<64>), dim3((int)nb), dim3(256), 0, s, k);
        else if (lpp == 16) hipLaunchKernelGGL((narrow_fwd_kernel<16>), dim3((int)nb), dim3(256), 0, s, k);
        else if (lpp == 4) hipLaunchKernelGGL((narrow_fwd_kernel<4>), dim3((int)nb), dim3(256), 0, s, k);
        else hipLaunchKernelGGL((narrow_fwd_kernel<1>), dim3((int)nb), dim3(256), 0, s, k);
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    return HV_ERR_UNSUPPORTED;
}
