// Data gradient of the stride-2 convolutions (3x3 and 4x4, pad 1) -- the gather form of conv_transpose2d(stride 2) -- with the four
// output-parity classes FUSED in one workgroup: one wave per parity.
//
//   y[n, 2i+ph, 2j+pw, co] = sum over the taps (r, q) with (ph+1-r), (pw+1-q) even, and ci:
//                            x[n, i + (ph+1-r)/2, j + (pw+1-q)/2, ci] * w[co][(r,q)][ci]
//
// The halo-tiled kernels (conv_halo.hip / conv_halo2.hip) run every parity class as its own set of workgroups: each stages its own
// input patch for one (4x4: 4 of 16, 3x3: 1, 2, 2 or 4 of 9) quarter of the taps, so a chunk's barrier and staging are paid per
// 64 .. 128 MFMAs.  All four classes read the SAME (TH+2) x (TW+2) low-resolution patch: here it is staged once per chunk, wave w
// computes parity (w >> 1, w & 1) for the whole TH x 16 tile of (i, j) and all BN output channels (its own accumulators, its own tap
// list), and the four waves write the four interleaved pixel sets of the 2TH x 32 output tile.  Filter rows go straight from L2 into
// MFMA A-operand registers as in conv_halo2.hip; fp16 storage only (the patch is copied as it is, 16 bytes per item).
#include <stdlib.h>

#include "conv_halo.h"

struct S2TK {
    const _Float16* x; const _Float16* w; const float* bias; void* y;
    int B, Hi, Wi, x_ld, x_coff, Cin, img_stride;
    int Cout, w_row, y_ld, y_coff, Ho, Wo;
    int tiles_x, tiles;
    float alpha; int act, accumulate, vec_store;
    const void* mul_src; int mul_ld, mul_coff, mul_act, mul_vec, y_half, mul_half;
    unsigned x_bytes, w_bytes;
    const _Float16* wt; unsigned wt_bytes;      // filters in MFMA-fragment order (hv_conv_desc.w_f16_tiled) or NULL
    int ep16;                                   // fp16 output tile handed through LDS, stored as 16-byte pieces
};

template <int KS, int TH, int BN, int CK>
__global__ __launch_bounds__(256, 2) void conv_s2t_kernel(const S2TK p) {
    constexpr int TW = 16, PH = TH + 2, PW = TW + 2;
    constexpr int FK = CK == 16 ? 16 : 32;
    constexpr int LDP = CK + (CK >= 32 ? 16 : 8);                 // 96-B (48-B) patch rows: conflict-free unit-step b128 (b64) reads
    constexpr int NT = BN / 16, MT = TH;                          // a wave: all BN channels x TH rows of 16 pixels, for its parity
    constexpr int PV = CK / 8, PMAX = (PH * PW * PV + 255) / 256; // 16-byte staging items
    typedef typename HFrag<FK>::V V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* patch = reinterpret_cast<_Float16*>(smem);          // [2][PH*PW][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, pw = wave & 1;
    int t = blockIdx.x;
    const int n_img = t / p.tiles;
    t -= n_img * p.tiles;
    const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;     // low-resolution tile origin
    const int n_base = blockIdx.y * BN;
    // this wave's taps, in closed form (scalar arithmetic; a per-class table in the kernel arguments would be indexed by the wave number
    // and the compiler then keeps the whole argument struct in scratch memory): filter row r_a = 1 - ph + 2a (a = 0, 1; valid while
    // r_a < KS) reads the patch row shifted by dh = ph - a; columns alike.  Slot k = a * nb + b.
    const int na = (1 - ph + 2 < KS) ? 2 : 1, nb = (1 - pw + 2 < KS) ? 2 : 1;
    const int ntaps = na * nb;                                            // wave-uniform (scalar): 1, 2 or 4
    auto tap_a = [&](int k) { return nb == 2 ? (k >> 1) : k; };
    auto tap_b = [&](int k) { return nb == 2 ? (k & 1) : 0; };
    auto tap_widx = [&](int k) { return (1 - ph + 2 * tap_a(k)) * KS + (1 - pw + 2 * tap_b(k)); };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool tiledw = p.wt != nullptr;        // scalar; layouts as in conv_halo2_kernel (the host has put the tiled table in p.w / p.w_bytes)
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const int wsc = tiledw ? 32 : 2;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned wvo[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int row = n_base + n * 16 + (lane & 15);
        wvo[n] = row < p.Cout ? (unsigned)((row * p.w_row + (lane >> 4) * (FK / 4)) * 2) : HV_OOB;
        if (tiledw) wvo[n] = row < ((p.Cout + 15) & ~15) ? (unsigned)((row >> 4) * (16 * p.w_row * 2) + lane * (FK / 2)) : HV_OOB;
    }
    u32x4 preg[PMAX];
    unsigned pvo[PMAX];
    int plo[PMAX];
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int e = tid + i * 256;
        const int c8 = e % PV, pix = e / PV, py = pix / PW, px = pix - py * PW;
        const int hi = ty0 - 1 + py, wi = tx0 - 1 + px;
        const bool in = e < PH * PW * PV;
        plo[i] = in ? pix * LDP + c8 * 8 : -1;
        pvo[i] = (in && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi)
                     ? (unsigned)(n_img * p.img_stride + (hi * p.Wi + wi) * p.x_ld + p.x_coff + c8 * 8) * 2u : HV_OOB;
    }
    const bool ragged = (p.Cin % CK) != 0;      // Cin % 8 == 0 is guaranteed by the host: an 8-channel item is wholly inside or outside
    auto ppref = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int c8 = (tid + i * 256) % PV;
            preg[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, (!ragged || c0 + c8 * 8 < p.Cin) ? pvo[i] : HV_OOB, c0 * 2, 0);
        }
    };
    auto pflush = [&](_Float16* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i)
            if (plo[i] >= 0) *reinterpret_cast<u32x4*>(dst + plo[i]) = preg[i];
    };
    const int kgc = (lane >> 4) * (FK / 4);
    auto wld = [&](int n, int widx_, int c0) __attribute__((always_inline)) {
        const unsigned vo = (ragged && c0 + kgc >= p.Cin) ? HV_OOB : wvo[n];
        if constexpr (FK == 32) return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrc, vo, (widx_ * p.Cin + c0) * wsc, 0));
        else {
            typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
            return __builtin_bit_cast(f16x4v, (u32x2_)__builtin_amdgcn_raw_buffer_load_b64(wsrc, vo, (widx_ * p.Cin + c0) * wsc, 0));
        }
    };
    // LDS offsets (halfs) of this lane's pixel (tile row m, column lane & 15) at shift (0, 0) of the padded patch
    const int pbase = (1 * PW + (lane & 15) + 1) * LDP;

    const int nchunks = (p.Cin + CK - 1) / CK;
    ppref(0);
    V wf[2][NT];        // this tap's and the next tap's filter fragments
#pragma unroll
    for (int n = 0; n < NT; ++n) wf[0][n] = wld(n, tap_widx(0), 0);
    pflush(patch);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const _Float16* pb = patch + (c & 1) * (PH * PW * LDP);
        if (c + 1 < nchunks) ppref((c + 1) * CK);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= ntaps) break;                                        // scalar branch (MFMA ignores EXEC)
            // next tap's (or the next chunk's first tap's) filter rows fly behind this tap's MFMAs
            const bool last = q + 1 >= ntaps;
            if (!last || c + 1 < nchunks) {
#pragma unroll
                for (int n = 0; n < NT; ++n) wf[(q + 1) & 1][n] = wld(n, tap_widx(last ? 0 : q + 1), (last ? c + 1 : c) * CK);
            }
            const int toff = ((ph - tap_a(q)) * PW + (pw - tap_b(q))) * LDP;
            V xf[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xf[m] = HFrag<FK>::ld(pb + pbase + m * PW * LDP + toff, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n][m] = HFrag<FK>::mma(wf[q & 1][n], xf[m], acc[n][m]);
        }
        // taps of a class are 1, 2 or 4: the ring parity after the last tap must be 0 again for the next chunk (ntaps odd: copy)
        if ((ntaps & 1) && c + 1 < nchunks) {
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[0][n] = wf[1][n];
        }
        if (c + 1 < nchunks) pflush(patch + ((c + 1) & 1) * (PH * PW * LDP));
        __syncthreads();
    }

    const HvEpi epi = {p.alpha, p.act, p.accumulate, p.vec_store, p.Cout, p.bias, nullptr, p.mul_act, p.mul_vec, p.y_half, p.mul_half};
    if (p.ep16) {
        // The four parity waves' pixels interleave in the output: a wave's direct stores touch every other pixel of every other row, 8 bytes per
        // lane.  Through LDS (the patch buffers are free behind the loop's last barrier) the 2TH x 32-pixel tile leaves as whole channel rows.
        constexpr int LDO = BN + 8, OW = 2 * TW;
        _Float16* ot = patch;
        // the data gradients that take this path carry no bias / activation of their own (their act' factor comes in the second stage): that case is one
        // multiply per element; with the shared per-element epilogue -- channel bound, scale, bias and activation each a wave-uniform branch per ELEMENT -- the
        // 16 -> 16 layer issued ~820 scalar and ~680 vector instructions per wave around its 18 MFMAs (PMC), 28.5 us for 75 MB
        const bool plain = p.ep16 == 2 && !p.bias && p.act == HV_ACT_NONE && n_base + BN <= p.Cout;      // (scalar)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int i = ty0 + m, j = tx0 + (lane & 15);
            const int ho = 2 * i + ph, wo = 2 * j + pw;
            const bool inside = i < p.Hi && j < p.Wi && ho < p.Ho && wo < p.Wo;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            const void* mp = (p.mul_src && inside && p.ep16 == 1) ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;     // (2: second stage)
            const int q = (2 * m + ph) * OW + 2 * (lane & 15) + pw;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int cl = n * 16 + (lane >> 4) * 4;
                f32x4 v;
                if (plain) v = f32x4{acc[n][m][0] * p.alpha, acc[n][m][1] * p.alpha, acc[n][m][2] * p.alpha, acc[n][m][3] * p.alpha};
                else v = hv_conv_value4<true>(epi, acc[n][m], n_base + cl, mp);
                *reinterpret_cast<f16x4v*>(ot + q * LDO + cl) = (f16x4v){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        __syncthreads();
        constexpr int PIECES = BN / 8;
        _Float16* yb = reinterpret_cast<_Float16*>(p.y);
        auto second = [&](auto gradf) __attribute__((always_inline)) {      // (the multiplier's activation chosen once, not per element)
            for (int it = tid; it < 2 * TH * OW * PIECES; it += 256) {
                const int q = it / PIECES, pc = it - q * PIECES;
                const int ho = 2 * ty0 + q / OW, wo = 2 * tx0 + q % OW, ch = n_base + pc * 8;
                if (ho >= p.Ho || wo >= p.Wo || ch >= p.Cout) continue;
                const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
                u32x4 o = *reinterpret_cast<const u32x4*>(ot + q * LDO + pc * 8);
                if (p.ep16 == 2) {      // act' multiplier from 16-byte loads (as in conv_halo2_kernel)
                    const f16x8 m8 = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(p.mul_src) + opix * p.mul_ld + p.mul_coff + ch);
                    f16x8 v8 = __builtin_bit_cast(f16x8, o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] * gradf((float)m8[e]));
                    o = __builtin_bit_cast(u32x4, v8);
                }
                *reinterpret_cast<u32x4*>(yb + opix * p.y_ld + p.y_coff + ch) = o;
            }
        };
        if (p.ep16 != 2 || p.mul_act == HV_ACT_ELU) second([](float y) { return y > 0.f ? 1.f : y + 1.f; });      // (hv_act_grad_from_out's expressions)
        else if (p.mul_act == HV_ACT_LRELU) second([](float y) { return y > 0.f ? 1.f : 0.2f; });
        else second([&](float y) { return hv_act_grad_from_out(y, p.mul_act); });
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int i = ty0 + m, j = tx0 + (lane & 15);
        const int ho = 2 * i + ph, wo = 2 * j + pw;
        if (i >= p.Hi || j >= p.Wi || ho >= p.Ho || wo >= p.Wo) continue;
        const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
        void* yp = hv_eptr(p.y, opix * p.y_ld + p.y_coff, p.y_half);
        const void* mp = p.mul_src ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
        for (int n = 0; n < NT; ++n) hv_conv_epilogue4<true>(epi, acc[n][m], n_base + n * 16 + (lane >> 4) * 4, yp, mp);
    }
}

template <int KS, int TH, int BN, int CK>
static int launch_s2t(S2TK& k, hipStream_t s) {
    constexpr int LDP = CK + (CK >= 32 ? 16 : 8);
    k.tiles_x = hv_cdiv(k.Wi, 16);
    k.tiles = k.tiles_x * hv_cdiv(k.Hi, TH);
    size_t lds = (size_t)2 * (TH + 2) * 18 * LDP * sizeof(_Float16);
    {   // coalesced fp16 epilogue through LDS (HV_HALO2_EP16=0: direct stores); the output tile may need more LDS than the two patch buffers
        static const int ep16 = getenv("HV_HALO2_EP16") ? atoi(getenv("HV_HALO2_EP16")) : 1;
        k.ep16 = (ep16 && k.y_half && k.accumulate == 0 && !(k.Cout & 7) && !(k.y_ld & 7) && !(k.y_coff & 7) && !((uintptr_t)k.y & 15) && k.Ho == 2 * k.Hi &&
                  k.Wo == 2 * k.Wi) ? 1 : 0;
        if (k.ep16 && k.mul_src && k.mul_half && !(k.mul_ld & 7) && !(k.mul_coff & 7) && !((uintptr_t)k.mul_src & 15)) k.ep16 = 2;
        const size_t need = (size_t)2 * TH * 32 * (BN + 8) * sizeof(_Float16);
        if (k.ep16 && need > lds) lds = need;
    }
    auto kern = conv_s2t_kernel<KS, TH, BN, CK>;
    static int lds_limit = 48 * 1024;       // per instantiation
    if ((int)lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        lds_limit = 150 * 1024;
    }
    dim3 grid(k.tiles * k.B, hv_cdiv(k.Cout, BN));
    hv_path_note = 6;
    HV_KNAME("conv_s2t_kernel<%d, %d, %d, %d>", KS, TH, BN, CK);
    HV_WUSE(k.wt ? 4 : 2);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

static int s2t_mode = -1;
extern "C" int hv_set_s2t_mode(int mode) {
    const int prev = s2t_mode < 0 ? (getenv("HV_S2T") ? atoi(getenv("HV_S2T")) : 1) : s2t_mode;
    s2t_mode = mode;
    return prev;
}

// Called by hv_conv2d for transposed (data-gradient) stride-2 3x3 / 4x4 convolutions in the fp16 mode; HV_ERR_UNSUPPORTED: other kernels.
int hv_conv2d_s2t(const hv_conv_desc* d, hipStream_t s) {
    // A/B knob: 0 off, 1 the 3x3 filters (2 - 3.6x faster than the per-class kernels at the generators' shapes), 2 also the 4x4 filters
    // (measured SLOWER than conv_halo2's classes at the PatchGAN shapes: 64.3 vs 47.1 us for 64<-128 @64^2, 50.7 vs 40.4 us for 128<-256 @32^2)
    if (s2t_mode < 0) s2t_mode = getenv("HV_S2T") ? atoi(getenv("HV_S2T")) : 1;
    const int enabled = s2t_mode;
    if (!enabled || (d->KH == 4 && enabled < 2) || !d->transposed || d->stride != 2 || d->pad != 1 || d->dil != 1 || d->KH != d->KW || (d->KH != 3 && d->KH != 4)) return HV_ERR_UNSUPPORTED;
    if (d->precision != HV_F16 || !d->w_f16 || !d->x_f16 || d->w_bstride || d->ch_scale || d->in_shift) return HV_ERR_UNSUPPORTED;
    if (d->Ho != 2 * d->H || d->Wo != 2 * d->W) return HV_ERR_UNSUPPORTED;
    if ((d->Cin & 7) || (d->x_ld & 7) || (d->x_coff & 7) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->w_f16 & 15) || d->Cout < 8) return HV_ERR_UNSUPPORTED;
    if ((long long)d->B * d->H * d->W * d->x_ld >= (1ll << 30) || (long long)d->Cout * d->KH * d->KW * d->Cin >= (1ll << 30)) return HV_ERR_UNSUPPORTED;
    S2TK k;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.bias = d->bias; k.y = d->y;
    k.B = d->B; k.Hi = d->H; k.Wi = d->W; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin; k.img_stride = d->H * d->W * d->x_ld;
    k.Cout = d->Cout; k.w_row = d->KH * d->KW * d->Cin; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Ho = d->Ho; k.Wo = d->Wo;
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    k.mul_src = d->mul_src; k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act;
    k.y_half = d->y_f16 ? 1 : 0; k.mul_half = d->mul_f16 ? 1 : 0;
    k.mul_vec = (d->mul_src && !(d->mul_ld & 3) && !(d->mul_coff & 3) && !((uintptr_t)d->mul_src & 15)) ? 1 : 0;
    k.vec_store = ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && ((uintptr_t)d->y & 15) == 0) ? 1 : 0;
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * sizeof(_Float16));
    k.w_bytes = (unsigned)((size_t)d->Cout * k.w_row * sizeof(_Float16));
    const bool ck32 = (d->Cin & 31) == 0;
    {   // the tiled table needs whole chunks (Cin % 16 == 0; fragments of 32 channels exactly when Cin % 32 == 0, as the CK choice below)
        static const int tiled = getenv("HV_W_TILED") ? atoi(getenv("HV_W_TILED")) : 1;
        k.wt = (tiled && d->w_f16_tiled && !((uintptr_t)d->w_f16_tiled & 15) && (d->Cin & 15) == 0) ? reinterpret_cast<const _Float16*>(d->w_f16_tiled) : nullptr;
        k.wt_bytes = (unsigned)((size_t)hv_cdiv(d->Cout, 16) * 16 * k.w_row * sizeof(_Float16));
        if (k.wt) { k.w = k.wt; k.w_bytes = k.wt_bytes; }
    }
    const long long wgs8 = (long long)d->B * hv_cdiv(d->H, 8) * hv_cdiv(d->W, 16) * hv_cdiv(d->Cout, 64);
    const bool th4 = wgs8 < 512;            // small maps: 4-row tiles double the workgroups
    const int bn = d->Cout > 32 ? 64 : d->Cout > 16 ? 32 : 16;
#define S2T(TH_, BN_)                                                                                              \
    do {                                                                                                           \
        if (d->KH == 4) return ck32 ? launch_s2t<4, TH_, BN_, 32>(k, s) : launch_s2t<4, TH_, BN_, 16>(k, s);       \
        return ck32 ? launch_s2t<3, TH_, BN_, 32>(k, s) : launch_s2t<3, TH_, BN_, 16>(k, s);                       \
    } while (0)
    if (bn == 64) S2T(4, 64);               // 8 rows x 64 channels: 128 accumulator registers + two filter rings spill
    if (bn == 32) { if (th4) S2T(4, 32); S2T(8, 32); }
    if (th4) S2T(4, 16);
    S2T(8, 16);
#undef S2T
}
