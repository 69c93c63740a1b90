// Forward convolution with a single output channel (the generator's four 1-channel heads, the PatchGAN logit layer):
// a 16-wide MFMA tile wastes 15/16 of its work on these and they are HBM-bound, so the forward runs as a plain fp32
// VALU kernel (exact in both precision modes):
//   y[p] = act(alpha * sum_{tap,c} x[p+tap][c] * w[tap][c] + b)
// (VALU versions of the matching data / weight gradients were measured 2-6x SLOWER than the channel-padded MFMA tiles
// and are not kept: round-1 notes in DESIGN.md.)
#include "hv_common.h"

struct NarrowK {
    const void* x; const float* w; const float* bias; void* y;      // x / y: fp32 or fp16 elements (x_half / y_half, wave-uniform)
    int x_half, y_half;
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
    const long long yi = pix * p.y_ld + p.y_coff;
    float t = acc * p.alpha;
    if (p.bias) t += p.bias[0];
    if (p.accumulate == 2) t += hv_ld1(p.y, yi, p.y_half);
    t = hv_act(t, p.act);
    hv_st1(p.y, yi, p.accumulate == 1 ? hv_ld1(p.y, yi, p.y_half) + t : t, p.y_half);
}

// LPP lanes cooperate on one output pixel: lanes split the channels (16 B each), taps are looped.
// grid (pixel blocks of one output row, B*Ho rows): no per-pixel index divisions.
template <int LPP>
__global__ __launch_bounds__(256) void narrow_fwd_kernel(const NarrowK p) {
    const int sub = threadIdx.x % LPP;
    constexpr int PPB = 256 / LPP;   // pixels per block
    const int row = blockIdx.y, b = row / p.Ho, oy = row - b * p.Ho;
    const int ox = blockIdx.x * PPB + threadIdx.x / LPP;
    float acc = 0.f;
    if (ox < p.Wo) {
        const long long ximg = (long long)b * p.img_stride + p.x_coff;
        for (int r = 0; r < p.KH; ++r)
            for (int s = 0; s < p.KW; ++s) {
                int hi, wi;
                if (!narrow_tap(p, oy, ox, r, s, hi, wi)) continue;
                const long long xp = ximg + (long long)((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld;
                const float* wp = p.w + (r * p.KW + s) * p.Cin;
                for (int c = sub * 4; c < p.Cin; c += LPP * 4) {
                    const float4 xv = hv_ld4(p.x, xp + c, p.x_half), wv = *reinterpret_cast<const float4*>(wp + c);
                    acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                }
            }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && ox < p.Wo) narrow_store(p, (long long)row * p.Wo + ox, acc);
}

// Few input channels (C4 float4 groups, the generator's 1-channel heads: 8 -> 1, 12 -> 1 at full resolution), 3x3 forward: one lane = one
// output pixel, its 9 * C4 sixteen-byte loads are all issued before the first is used (clamped address + mask, no branches) -- the
// loop form above serialises a load round trip per tap here.
template <int C4>
__global__ __launch_bounds__(256) void narrow3_fwd_kernel(const NarrowK p) {
    const int row = blockIdx.y, b = row / p.Ho, oy = row - b * p.Ho;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= p.Wo) return;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
    float4 xv[9][C4];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int hi = oy * p.stride - p.pad + r, wi = ox * p.stride - p.pad + s;
            ok[r * 3 + s] = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int hc = min(max(hi, 0), p.H - 1), wc = min(max(wi, 0), p.W - 1);
            const long long xp = ximg + (long long)((hc >> p.in_shift) * p.Wp + (wc >> p.in_shift)) * p.x_ld;
#pragma unroll
            for (int c = 0; c < C4; ++c) xv[r * 3 + s][c] = hv_ld4(p.x, xp + c * 4, p.x_half);
        }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            const float4 wv = *reinterpret_cast<const float4*>(p.w + t * p.Cin + c * 4), v = xv[t][c];
            a += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
        acc += ok[t] ? a : 0.f;
    }
    narrow_store(p, (long long)row * p.Wo + ox, acc);
}

// The same heads with the three input rows of a 256-pixel output segment staged ONCE in LDS (fp32): a lane of the kernel above fetches its
// 9 * C4 pieces itself, 9x what the segment holds (30.9 us for the 25 MB of a 12 -> 1 head at 256^2, bs 16).  Unit stride, pad 1, no fused upsample.
// The summation order per pixel is the one above (taps outer, channels inner): the same bits.
template <int C4, bool XH>      // XH: fp16 input, staged as it is (half the LDS: eight workgroups per CU instead of four)
__global__ __launch_bounds__(256) void narrow3_lds_kernel(const NarrowK p) {
    constexpr int C = C4 * 4, SEG = 256, PWID = SEG + 2;
    typedef typename std::conditional<XH, _Float16, float>::type ST;
    typedef typename std::conditional<XH, f16x4, float4>::type SV;
    __shared__ __attribute__((aligned(16))) ST xs[3 * PWID * C];
    const int row = blockIdx.y, b = row / p.Ho, oy = row - b * p.Ho;
    const int x0 = blockIdx.x * SEG;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
    for (int e = threadIdx.x; e < 3 * PWID * C4; e += 256) {
        const int c4 = e % C4, q = e / C4, px = q % PWID, r = q / PWID;
        const int hi = oy - 1 + r, wi = x0 - 1 + px;
        const bool in = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const long long xi = ximg + (long long)(hi * p.Wp + wi) * p.x_ld + c4 * 4;
        SV v;
        if constexpr (XH) v = in ? *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(p.x) + xi) : (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        else v = in ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.x) + xi) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<SV*>(xs + (r * PWID + px) * C + c4 * 4) = v;
    }
    __syncthreads();
    const int ox = x0 + threadIdx.x;
    if (ox >= p.Wo) return;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < C4; ++c) {
                const float4 wv = *reinterpret_cast<const float4*>(p.w + (r * 3 + s2) * p.Cin + c * 4);
                const SV q = *reinterpret_cast<const SV*>(xs + (r * PWID + threadIdx.x + s2) * C + c * 4);
                float4 v;
                if constexpr (XH) v = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
                else v = q;
                a += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
            }
            acc += a;       // out-of-image taps hold zeros
        }
    narrow_store(p, (long long)row * p.Wo + ox, acc);
}

// Forward convolution of a 1-channel image into 4..64 channels with a KS x KS filter (the PatchGAN stem: 1 -> 64, 4x4, stride 2).
// K = taps is 16, the layer is bound by writing its output (67 MB at B=16): fp32 VALU, exact in both precision modes.
// One lane = one pixel x 4 output channels (16 lanes write a pixel's 256 B); the lane's filter taps live in registers (staged
// once per block through LDS), a block walks PXB pixel groups of one output row.  grid (row segments, B*Ho rows).
template <int KS, int PXB>
__global__ __launch_bounds__(256) void thin1_fwd_kernel(const NarrowK p) {
    constexpr int TAPS = KS * KS;
    __shared__ __attribute__((aligned(16))) float wl[TAPS * 64];        // [tap][Cout]
    const int CG = p.Cout >> 2;                                         // Cout a power of two <= 64
    for (int e = threadIdx.x; e < TAPS * p.Cout; e += 256) {
        const int co = e % p.Cout, t = e / p.Cout;
        wl[e] = p.w[(long long)co * TAPS + t];
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG, PPB = 256 / CG;
    float4 wr[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wr[t] = *reinterpret_cast<const float4*>(wl + t * p.Cout + cg * 4);
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = p.bias[cg * 4 + e];
    }
    const int row = blockIdx.y, b = row / p.Ho, oy = row - b * p.Ho;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
#pragma unroll 2
    for (int it = 0; it < PXB; ++it) {
        const int ox = (blockIdx.x * PXB + it) * PPB + pl;
        if (ox >= p.Wo) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const int hi = oy * p.stride - p.pad + r;
            const bool rok = (unsigned)hi < (unsigned)p.H;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const int wi = ox * p.stride - p.pad + q;
                float xv = 0.f;
                if (rok && (unsigned)wi < (unsigned)p.W) xv = hv_ld1(p.x, ximg + (long long)((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld, p.x_half);
                const float4 wv = wr[r * KS + q];
                acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
            }
        }
        const long long yi = ((long long)row * p.Wo + ox) * p.y_ld + p.y_coff + cg * 4;
        float o[4] = {acc.x, acc.y, acc.z, acc.w};
        const float4 old4 = p.accumulate ? hv_ld4(p.y, yi, p.y_half) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float old[4] = {old4.x, old4.y, old4.z, old4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = o[e] * p.alpha + bias[e];
            if (p.accumulate == 2) t += old[e];
            t = hv_act(t, p.act);
            o[e] = p.accumulate == 1 ? old[e] + t : t;
        }
        hv_st4(p.y, yi, make_float4(o[0], o[1], o[2], o[3]), p.y_half);
    }
}

// The same layer in fp16 mode on MFMA: the VALU form above spends ~45 vector instructions per output (16 taps x 4 channels of separate
// multiply / add) and runs at 41 us where writing its 67 MB takes ~12.  With the 16 taps as the contraction of one
// v_mfma_f32_16x16x16_f16 -- A = filters [16 channels x 16 taps] from registers, B = the image window: lane (pixel, row r) loads the four
// consecutive input pixels of filter row r -- a wave turns 16 pixels x 64 channels into 4 MFMAs, and the lane's four accumulators are four
// consecutive channels of one pixel (one 16-byte store).  One workgroup = one output row; a wave takes 16-pixel groups.
// ST = true (fp16 output, 64 channels, no accumulate form, 16-byte aligned channel rows): a wave's 16 pixels x 64 channels leave through a 2-KB LDS
// image as whole 16-byte pieces -- one contiguous 2 KB per group instead of sixteen 8-byte-per-lane stores that each touch a quarter of 16 lines.
template <bool ST>
__global__ __launch_bounds__(256) void stem1_mfma_kernel(const NarrowK p, const _Float16* __restrict__ wh) {
    constexpr int LDT = 64 + 8;                       // halfs per staged pixel row (144 B)
    __shared__ __attribute__((aligned(16))) _Float16 stg[ST ? 4 * 16 * LDT : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int MT = p.Cout >> 4;                       // 16-channel tiles (<= 4)
    f16x4 wa[4];
    float bias[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        wa[t] = f16x4{0, 0, 0, 0};
        if (t < MT) wa[t] = *reinterpret_cast<const f16x4*>(wh + (t * 16 + n) * 16 + g * 4);      // channel 16t + n, taps 4g .. 4g+3 (filter row g)
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[t][j] = (p.bias && t < MT) ? p.bias[t * 16 + g * 4 + j] : 0.f;
    }
    // the input's storage and the activation are chosen ONCE (wave-uniform): with hv_ld1 / hv_act deciding per element the kernel issued ~250 instructions per
    // 16-pixel group, most of them scalar branches and their bookkeeping -- 25 us for 67 MB that a plain fill writes in 12.  A workgroup walks several output rows
    // (the 20 filter / bias loads of a wave's prologue were paid per row: two 16-pixel groups)
    auto walk = [&](auto ldx, auto actf) __attribute__((always_inline)) {
      for (int row = blockIdx.x; row < p.B * p.Ho; row += gridDim.x) {
        const int b = row / p.Ho, oy = row - b * p.Ho;
        const long long ximg = (long long)b * p.img_stride + p.x_coff;
        const int iy = oy * p.stride - p.pad + g;
        const bool rok = (unsigned)iy < (unsigned)p.H;
        const long long xrow = ximg + (long long)((rok ? iy : 0) >> p.in_shift) * p.Wp * p.x_ld;
        for (int ox0 = wave * 16; ox0 < p.Wo; ox0 += 64) {
            const int ox = ox0 + n, ix0 = ox * p.stride - p.pad;
            _Float16 xv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = ix0 + e;
                const bool ok = rok && ox < p.Wo && (unsigned)ix < (unsigned)p.W;
                const float v = ldx(xrow + (long long)((ok ? ix : 0) >> p.in_shift) * p.x_ld);
                xv[e] = (_Float16)(ok ? v : 0.f);
            }
            const f16x4 xb = {xv[0], xv[1], xv[2], xv[3]};
            const long long yi = ((long long)row * p.Wo + ox) * p.y_ld + p.y_coff + g * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= MT) break;
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[t], xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // all lanes: MFMA ignores EXEC
                if (ox >= p.Wo) continue;                  // ragged last group: columns past the row end are computed on zeros and not stored
                float o[4];
                if constexpr (ST) {
                    f16x4 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = (_Float16)actf(acc[j] * p.alpha + bias[t][j]);
                    *reinterpret_cast<f16x4*>(stg + (wave * 16 + n) * LDT + t * 16 + g * 4) = h;
                    continue;
                }
                const float4 old4 = p.accumulate ? hv_ld4(p.y, yi + t * 16, p.y_half) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float old[4] = {old4.x, old4.y, old4.z, old4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[j] * p.alpha + bias[t][j];
                    if (p.accumulate == 2) v += old[j];
                    v = actf(v);
                    o[j] = p.accumulate == 1 ? old[j] + v : v;
                }
                hv_st4(p.y, yi + t * 16, make_float4(o[0], o[1], o[2], o[3]), p.y_half);
            }
            if constexpr (ST) {
                // the wave's own 16 x 64 image (written and read by this wave only: no workgroup barrier)
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the LDS stores above have landed
                __builtin_amdgcn_wave_barrier();
                _Float16* yb = reinterpret_cast<_Float16*>(p.y);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int q = lane + 64 * k, px = q >> 3, c8 = q & 7;
                    const hv_u32x4 v = *reinterpret_cast<const hv_u32x4*>(stg + (wave * 16 + px) * LDT + c8 * 8);
                    if (ox0 + px < p.Wo) *reinterpret_cast<hv_u32x4*>(yb + ((long long)row * p.Wo + ox0 + px) * p.y_ld + p.y_coff + c8 * 8) = v;
                }
                __builtin_amdgcn_wave_barrier();             // the next group's stores come after these reads
            }
        }
      }
    };
    const float* xf = reinterpret_cast<const float*>(p.x);
    const _Float16* xh = reinterpret_cast<const _Float16*>(p.x);
    auto ld32 = [&](long long i) { return xf[i]; };
    auto ld16 = [&](long long i) { return (float)xh[i]; };
    auto lrelu = [](float v) { return v > 0.f ? v : 0.2f * v; };      // (hv_act's expression)
    auto anyact = [&](float v) { return hv_act(v, p.act); };
    if (p.act == HV_ACT_LRELU) { if (p.x_half) walk(ld16, lrelu); else walk(ld32, lrelu); }
    else { if (p.x_half) walk(ld16, anyact); else walk(ld32, anyact); }
}

// The generators' 5x5 stems (4 -> 16 channels at full resolution), fp16 mode: K = 25 taps x 4 channels = 100, flattened (tap, channel) and
// padded to 128 = four v_mfma_f32_16x16x32_f16 per 16 pixels x 16 channels -- the halo-tiled kernel spends one quarter-filled 16-deep MFMA
// per tap.  No LDS: lane (pixel, k-group) loads its two taps' 4-channel pixels (16 B each) straight from global memory, L1 serves the
// 25-fold reuse; the lane's accumulators are four consecutive output channels of its pixel.  One workgroup = one output row.
struct Stem5K {
    const void* x; const _Float16* w; void* y; int x_half;
    int H, W, x_ld, x_coff, img_stride, y_ld, y_coff;
    HvEpi epi;
};
__global__ __launch_bounds__(256) void stem5_mfma_kernel(const Stem5K p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    f16x8 wa[4];                                       // filters of channel n: k = 32j + 8g .. +7 of the (tap, ci) axis, zero past 100
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k0 = 32 * j + 8 * g;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k0 + 8 <= 100) v = *reinterpret_cast<const f16x8*>(p.w + n * 100 + k0);
        else if (k0 < 100) {
            const f16x4 h = *reinterpret_cast<const f16x4*>(p.w + n * 100 + k0);
            v = f16x8{h[0], h[1], h[2], h[3], 0, 0, 0, 0};
        }
        wa[j] = v;
    }
    const int row = blockIdx.x, b = row / p.H, oy = row - b * p.H;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
    for (int ox0 = wave * 16; ox0 < p.W; ox0 += 64) {
        const int ox = ox0 + n;
        float4 xv[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 8 * j + 2 * g + h;                     // tap index (25..31: padding)
                const int r = t / 5, q = t - r * 5;
                const int iy = oy - 2 + r, ix = ox - 2 + q;
                const bool ok = t < 25 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float4 v = hv_ld4(p.x, ximg + ((long long)(ok ? iy : 0) * p.W + (ok ? ix : 0)) * p.x_ld, p.x_half);
                xv[j][h] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 xb = {(_Float16)xv[j][0].x, (_Float16)xv[j][0].y, (_Float16)xv[j][0].z, (_Float16)xv[j][0].w,
                              (_Float16)xv[j][1].x, (_Float16)xv[j][1].y, (_Float16)xv[j][1].z, (_Float16)xv[j][1].w};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[j], xb, acc, 0, 0, 0);
        }
        if (ox < p.W) hv_conv_epilogue4<true>(p.epi, acc, g * 4, hv_eptr(p.y, ((long long)row * p.W + ox) * p.y_ld + p.y_coff, p.epi.y_half), nullptr);
    }
}

// The same with the five input rows of a 256-pixel output segment staged once in LDS as they are stored (fp16 input): a lane's eight operand
// pixels per 16-pixel group come from LDS (no conversion), one global round trip per workgroup instead of one per group (56 us for 75 MB).
__global__ __launch_bounds__(256) void stem5_lds_kernel(const Stem5K p) {
    constexpr int SEG = 256, PWID = SEG + 4;
    __shared__ __attribute__((aligned(8))) f16x4 xs[5 * PWID];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    f16x8 wa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k0 = 32 * j + 8 * g;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k0 + 8 <= 100) v = *reinterpret_cast<const f16x8*>(p.w + n * 100 + k0);
        else if (k0 < 100) {
            const f16x4 h = *reinterpret_cast<const f16x4*>(p.w + n * 100 + k0);
            v = f16x8{h[0], h[1], h[2], h[3], 0, 0, 0, 0};
        }
        wa[j] = v;
    }
    const int row = blockIdx.y, b = row / p.H, oy = row - b * p.H, x0 = blockIdx.x * SEG;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
    const _Float16* xh = reinterpret_cast<const _Float16*>(p.x);
    for (int e = threadIdx.x; e < 5 * PWID; e += 256) {
        const int r = e / PWID, c = e - r * PWID;
        const int iy = oy - 2 + r, ix = x0 - 2 + c;
        f16x4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = *reinterpret_cast<const f16x4*>(xh + ximg + ((long long)iy * p.W + ix) * p.x_ld);
        xs[e] = v;
    }
    __syncthreads();
    for (int ox0 = wave * 16; ox0 < SEG && x0 + ox0 < p.W; ox0 += 64) {
        const int oxl = ox0 + n, ox = x0 + oxl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x4 lo = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, hi = lo;
            const int t0 = 8 * j + 2 * g, t1 = t0 + 1;                 // tap indices (25..31: padding)
            if (t0 < 25) lo = xs[(t0 / 5) * PWID + oxl + t0 % 5];
            if (t1 < 25) hi = xs[(t1 / 5) * PWID + oxl + t1 % 5];
            const f16x8 xb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[j], xb, acc, 0, 0, 0);
        }
        if (ox < p.W) hv_conv_epilogue4<true>(p.epi, acc, g * 4, hv_eptr(p.y, ((long long)row * p.W + ox) * p.y_ld + p.y_coff, p.epi.y_half), nullptr);
    }
}

// Data gradient of the same stems (16 -> 4 channels, the gather form of conv_transpose): K = 25 taps x 16 channels = 400 -> thirteen 16x16x32
// MFMAs per 16 pixels with the four output channels in rows 0..3 of the A operand (filters [4][25][16], rows 4..15 zero), the (8 + 4) x (64 + 4) gradient
// pixels of an 8 x 64-pixel output tile staged once in LDS as stored.  The halo-tiled kernel spends a quarter-filled 16-deep MFMA per tap and 4-channel block
// (60 us for 54 MB); this one reads every operand fragment with one 16-byte LDS load.
template <int TH, int SEG>       // a workgroup owns TH output rows x SEG pixels: (TH + 4) x (SEG + 4) gradient pixels staged once
__global__ __launch_bounds__(256) void stem5_dgrad_kernel(const Stem5K p) {
    constexpr int PWID = SEG + 4, PH = TH + 4, CG = 16, KSTEPS = 13, GPR = SEG / 16;
    __shared__ __attribute__((aligned(16))) _Float16 xs[PH * PWID * CG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    f16x8 wa[KSTEPS];                                  // row n of the filter matrix [4][400] (zero rows 4..15, zero past k = 400)
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) {
        const int k0 = 32 * j + 8 * g;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < 4 && k0 + 8 <= 400) v = *reinterpret_cast<const f16x8*>(p.w + n * 400 + k0);
        wa[j] = v;
    }
    const int tiles_y = (p.H + TH - 1) / TH;
    const int b = blockIdx.y / tiles_y, y0 = (blockIdx.y - b * tiles_y) * TH, x0 = blockIdx.x * SEG;
    const long long ximg = (long long)b * p.img_stride + p.x_coff;
    const _Float16* xh = reinterpret_cast<const _Float16*>(p.x);
    for (int e = threadIdx.x; e < PH * PWID * 2; e += 256) {
        const int half8 = e & 1, q = e >> 1, r = q / PWID, c = q - r * PWID;
        const int iy = y0 - 2 + r, ix = x0 - 2 + c;
        hv_u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            v = *reinterpret_cast<const hv_u32x4*>(xh + ximg + ((long long)iy * p.W + ix) * p.x_ld + half8 * 8);
        *reinterpret_cast<hv_u32x4*>(xs + q * CG + half8 * 8) = v;
    }
    __syncthreads();
    for (int grp = wave; grp < TH * GPR; grp += 4) {
        const int ty = grp / GPR, oxl = (grp - ty * GPR) * 16 + n, oy = y0 + ty, ox = x0 + oxl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            const int k0 = 32 * j + 8 * g, t = k0 >> 4, c0 = k0 & 15;          // tap (r, q) reads the gradient at (oy + 2 - r, ox + 2 - q)
            f16x8 xb = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t < 25) xb = *reinterpret_cast<const f16x8*>(xs + ((ty + 4 - t / 5) * PWID + oxl + 4 - t % 5) * CG + c0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[j], xb, acc, 0, 0, 0);
        }
        if (oy < p.H && ox < p.W)
            hv_conv_epilogue4<true>(p.epi, acc, g * 4, hv_eptr(p.y, (((long long)b * p.H + oy) * p.W + ox) * p.y_ld + p.y_coff, p.epi.y_half), nullptr);
    }
}

// 4 -> 16 channels, 5x5, stride 1, 'same' padding, fp16 filter copy [16][25][4]
int hv_conv2d_stem5(const hv_conv_desc* d, hipStream_t s) {
    static const int enabled = getenv("HV_STEM5") ? atoi(getenv("HV_STEM5")) : 1;   // A/B knob
    if (!enabled || d->precision != HV_F16 || !d->w_f16 || d->transposed || d->in_shift || d->w_bstride || d->ch_scale || d->mul_src || d->dil != 1 ||
        d->Cin != 4 || d->Cout != 16 || d->KH != 5 || d->KW != 5 || d->stride != 1 || d->pad != 2 || d->Ho != d->H || d->Wo != d->W)
        return HV_ERR_UNSUPPORTED;
    if ((d->x_ld & 3) || (d->x_coff & 3) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->w_f16 & 7) || (long long)d->B * d->H >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    Stem5K k;
    HV_WUSE(2);
    k.x = d->x; k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.y = d->y; k.x_half = d->x_f16 ? 1 : 0;
    k.H = d->H; k.W = d->W; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.img_stride = d->H * d->W * d->x_ld; k.y_ld = d->y_ld; k.y_coff = d->y_coff;
    k.epi.alpha = d->alpha; k.epi.act = d->act; k.epi.accumulate = d->accumulate; k.epi.Cout = 16; k.epi.bias = d->bias; k.epi.scale = nullptr;
    k.epi.mul_act = 0; k.epi.mul_vec = 0; k.epi.y_half = d->y_f16 ? 1 : 0; k.epi.mul_half = 0;
    k.epi.vec_store = ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && ((uintptr_t)d->y & 15) == 0) ? 1 : 0;
    hv_path_note = 4;
    static const int lds5 = getenv("HV_STEM5_LDS") ? atoi(getenv("HV_STEM5_LDS")) : 1;       // A/B knob
    if (lds5 && k.x_half && (long long)d->B * d->H <= 65535) {
        HV_KNAME("stem5_lds_kernel");
        hipLaunchKernelGGL(stem5_lds_kernel, dim3(hv_cdiv(d->W, 256), d->B * d->H), dim3(256), 0, s, k);
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    HV_KNAME("stem5_mfma_kernel");
    hipLaunchKernelGGL(stem5_mfma_kernel, dim3(d->B * d->H), dim3(256), 0, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// data gradient of those stems: 16 -> 4 channels, transposed, fp16 filter copy [4][25][16], fp16 gradient input
int hv_conv2d_stem5_dgrad(const hv_conv_desc* d, hipStream_t s) {
    static const int enabled = getenv("HV_STEM5_DGRAD") ? atoi(getenv("HV_STEM5_DGRAD")) : 1;   // A/B knob
    if (!enabled || d->precision != HV_F16 || !d->w_f16 || !d->transposed || d->in_shift || d->w_bstride || d->ch_scale || d->mul_src || d->dil != 1 ||
        d->Cin != 16 || d->Cout != 4 || d->KH != 5 || d->KW != 5 || d->stride != 1 || d->pad != 2 || d->Ho != d->H || d->Wo != d->W || !d->x_f16 || d->bias)
        return HV_ERR_UNSUPPORTED;
    if ((d->x_ld & 7) || (d->x_coff & 7) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->w_f16 & 15) || (long long)d->B * hv_cdiv(d->H, 8) > 65535) return HV_ERR_UNSUPPORTED;
    Stem5K k;
    HV_WUSE(2);
    k.x = d->x; k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.y = d->y; k.x_half = 1;
    k.H = d->H; k.W = d->W; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.img_stride = d->H * d->W * d->x_ld; k.y_ld = d->y_ld; k.y_coff = d->y_coff;
    k.epi.alpha = d->alpha; k.epi.act = d->act; k.epi.accumulate = d->accumulate; k.epi.Cout = 4; k.epi.bias = nullptr; k.epi.scale = nullptr;
    k.epi.mul_act = 0; k.epi.mul_vec = 0; k.epi.y_half = d->y_f16 ? 1 : 0; k.epi.mul_half = 0;
    k.epi.vec_store = ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && ((uintptr_t)d->y & 15) == 0) ? 1 : 0;
    hv_path_note = 4;
    HV_KNAME("stem5_dgrad_kernel");
    hipLaunchKernelGGL((stem5_dgrad_kernel<8, 64>), dim3(hv_cdiv(d->W, 64), d->B * hv_cdiv(d->H, 8)), dim3(256), 0, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

int hv_conv2d_narrow(const hv_conv_desc* d, hipStream_t s) {
    if (d->w_bstride || d->ch_scale || d->dil != 1 || d->stride > 2) return HV_ERR_UNSUPPORTED;
    NarrowK k;
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    HV_WUSE(1 | 2);      // (fp32 VALU kernels; the MFMA stem reads the fp16 rows)
    k.x = d->x; k.w = d->w; k.bias = d->bias; k.y = d->y; k.x_half = d->x_f16 ? 1 : 0; k.y_half = d->y_f16 ? 1 : 0;
    k.B = d->B; k.H = d->H; k.W = d->W; k.in_shift = d->in_shift; k.Wp = Wp; k.img_stride = Hp * Wp * d->x_ld;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.transposed = d->transposed;
    k.Ho = d->Ho; k.Wo = d->Wo; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Cout = d->Cout;
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    if (d->Cout == 1 && (d->Cin & 3) == 0 && !(d->x_ld & 3) && !(d->x_coff & 3) && !((uintptr_t)d->x & 15) && !((uintptr_t)d->w & 15)) {
        const int c4 = d->Cin / 4;
        const int lpp = c4 >= 64 ? 64 : c4 >= 16 ? 16 : c4 >= 4 ? 4 : 1;
        if ((long long)d->B * d->Ho > 65535) return HV_ERR_UNSUPPORTED;
        const dim3 grid(hv_cdiv(d->Wo, 256 / lpp), d->B * d->Ho);
        hv_path_note = 1;
        if (lpp == 1 && !d->transposed && d->KH == 3 && d->KW == 3 && (c4 == 2 || c4 == 3)) {
            const dim3 g3(hv_cdiv(d->Wo, 256), d->B * d->Ho);
            static const int lds3 = getenv("HV_NARROW3_LDS") ? atoi(getenv("HV_NARROW3_LDS")) : 1;      // A/B knob
            if (lds3 && d->stride == 1 && d->pad == 1 && d->in_shift == 0 && d->Ho == d->H && d->Wo == d->W) {
                HV_KNAME("narrow3_lds_kernel<%d>", c4);
                if (k.x_half) {
                    if (c4 == 2) hipLaunchKernelGGL((narrow3_lds_kernel<2, true>), g3, dim3(256), 0, s, k);
                    else hipLaunchKernelGGL((narrow3_lds_kernel<3, true>), g3, dim3(256), 0, s, k);
                } else {
                    if (c4 == 2) hipLaunchKernelGGL((narrow3_lds_kernel<2, false>), g3, dim3(256), 0, s, k);
                    else hipLaunchKernelGGL((narrow3_lds_kernel<3, false>), g3, dim3(256), 0, s, k);
                }
                HV_LAUNCH_CHECK();
                return HV_OK;
            }
            HV_KNAME("narrow3_fwd_kernel<%d>", c4);
            if (c4 == 2) hipLaunchKernelGGL((narrow3_fwd_kernel<2>), g3, dim3(256), 0, s, k);
            else hipLaunchKernelGGL((narrow3_fwd_kernel<3>), g3, dim3(256), 0, s, k);
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
        HV_KNAME("narrow_fwd_kernel<%d>", lpp);
        if (lpp == 64) hipLaunchKernelGGL((narrow_fwd_kernel<64>), grid, dim3(256), 0, s, k);
        else if (lpp == 16) hipLaunchKernelGGL((narrow_fwd_kernel<16>), grid, dim3(256), 0, s, k);
        else if (lpp == 4) hipLaunchKernelGGL((narrow_fwd_kernel<4>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((narrow_fwd_kernel<1>), grid, dim3(256), 0, s, k);
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    return HV_ERR_UNSUPPORTED;
}

// 1-channel image into 4..64 channels, 4x4 filter, forward only; w is the fp32 filter [Cout][taps][1]
int hv_conv2d_thin_in(const hv_conv_desc* d, hipStream_t s) {
    if (d->transposed || d->w_bstride || d->ch_scale || d->dil != 1 || d->Cin != 1 || d->KH != 4 || d->KW != 4 || d->Cout < 4 || d->Cout > 64 ||
        (d->Cout & (d->Cout - 1)))
        return HV_ERR_UNSUPPORTED;   // Cout a power of two: 256 threads = whole pixels
    if ((d->y_ld & 3) || (d->y_coff & 3) || ((uintptr_t)d->y & 15) || (long long)d->B * d->Ho > 65535) return HV_ERR_UNSUPPORTED;
    NarrowK k;
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    HV_WUSE(1 | 2);      // (fp32 VALU kernels; the MFMA stem reads the fp16 rows)
    k.x = d->x; k.w = d->w; k.bias = d->bias; k.y = d->y; k.x_half = d->x_f16 ? 1 : 0; k.y_half = d->y_f16 ? 1 : 0;
    k.B = d->B; k.H = d->H; k.W = d->W; k.in_shift = d->in_shift; k.Wp = Wp; k.img_stride = Hp * Wp * d->x_ld;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.transposed = 0;
    k.Ho = d->Ho; k.Wo = d->Wo; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Cout = d->Cout;
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    static const int stem_mfma = getenv("HV_STEM_MFMA") ? atoi(getenv("HV_STEM_MFMA")) : 1;   // A/B knob
    if (stem_mfma && d->precision == HV_F16 && d->w_f16 && (d->Cout & 15) == 0 && !((uintptr_t)d->w_f16 & 7) && (long long)d->B * d->Ho < (1ll << 31)) {
        hv_path_note = 4;
        static const int stem_st = getenv("HV_STEM_ST") ? atoi(getenv("HV_STEM_ST")) : 1;   // A/B knob
        if (stem_st && d->y_f16 && d->Cout == 64 && !d->accumulate && !(d->y_ld & 7) && !(d->y_coff & 7) && !((uintptr_t)d->y & 15)) {
            HV_KNAME("stem1_mfma_kernel<true>");
            static const int rpw = getenv("HV_STEM1_ROWS") ? atoi(getenv("HV_STEM1_ROWS")) : 1;      // tuning knob: output rows per workgroup (B32 / B16 at 256^2, us: 1 row 23.9 / 12.5, 2 rows 23.4 / 12.3, 4 rows 22.3 / 15.1, 8 rows 28.7 / 22.2)
            hipLaunchKernelGGL(stem1_mfma_kernel<true>, dim3(hv_cdiv(d->B * d->Ho, rpw > 0 ? rpw : 1)), dim3(256), 0, s, k, reinterpret_cast<const _Float16*>(d->w_f16));
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
        HV_KNAME("stem1_mfma_kernel");
        hipLaunchKernelGGL(stem1_mfma_kernel<false>, dim3(d->B * d->Ho), dim3(256), 0, s, k, reinterpret_cast<const _Float16*>(d->w_f16));
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    constexpr int PXB = 8;
    const int ppb = 256 / (d->Cout / 4);
    const dim3 grid(hv_cdiv(d->Wo, ppb * PXB), d->B * d->Ho);
    hv_path_note = 4;
    HV_KNAME("thin1_fwd_kernel<4, %d>", PXB);
    hipLaunchKernelGGL((thin1_fwd_kernel<4, PXB>), grid, dim3(256), 0, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
