// Implicit-GEMM convolution family on MFMA for gfx950 (MI355X).
//
//   hv_conv2d        : conv2d and the gather form of conv_transpose2d / data-gradient
//   hv_conv2d_wgrad  : weight gradient (contraction over pixels, split over pixel chunks)
//
// GEMM view (forward):  D[co][pix] = sum_k W[co][k] * X[pix][k],  k = (tap, ci) flattened.
// One 256-thread workgroup (4 waves) owns a BM(pixels) x BN(channels) tile; per 32-deep K step the
// input pixels of the current tap(s) are gathered from NHWC global memory (16 B per lane, coalesced
// along channels), rounded to the compute type and staged in LDS next to the weight tile; each
// wave then reads 8 contiguous k per lane for its 16x16 MFMA blocks.  Weights are the MFMA A
// operand (rows = channels) so each lane ends up with 4 consecutive output channels of one pixel
// and the epilogue (scale, bias, activation, accumulate) stores 16 B per lane.
//
// Precision: float -> v_mfma_f32_16x16x4_f32 (exact fp32, the parity mode);
//            _Float16 -> v_mfma_f32_16x16x32_f16 with fp32 accumulation.
#include <string.h>

#include "hv_common.h"
#include <stdlib.h>

#define HV_MAX_TAPS 25
#define HV_BK 32

thread_local int hv_path_note = 0;
thread_local int hv_wtable_used = 0;
extern "C" int hv_last_weight_tables(void) { return hv_wtable_used; }
thread_local char hv_kname[192] = "";
extern "C" const char* hv_last_kernel_name(void) { return hv_kname; }
extern "C" int hv_last_kernel_path(void) { return hv_path_note; }
thread_local hipEvent_t hv_ev_start = nullptr, hv_ev_stop = nullptr;
extern "C" int hv_set_kernel_timing(void* ev_start, void* ev_stop) {
    hv_ev_start = (hipEvent_t)ev_start; hv_ev_stop = (hipEvent_t)ev_stop;
    return HV_OK;
}

int hv_conv2d_halo(const hv_conv_desc* d, const void* w_f16, hipStream_t s);   // conv_halo.hip
int hv_conv2d_g4(const hv_conv_desc* d, hipStream_t s);                        // conv_g4.hip
int hv_conv2d_thin_dgrad(const hv_conv_desc* d, hipStream_t s);                // conv_thin.hip
int hv_conv2d_logits_dgrad(const hv_conv_desc* d, hipStream_t s);
size_t hv_conv2d_g4_stats_floats(const hv_conv_desc* d, int* nparts);
size_t hv_conv2d_g4_bstats_parts(const hv_conv_desc* d);
size_t hv_conv2d_logits_bstats_parts(const hv_conv_desc* d);                   // conv_thin.hip
size_t hv_wgrad_halo_workspace_bytes(const hv_wgrad_desc* d);                  // wgrad_halo.hip
int hv_wgrad_halo(const hv_wgrad_desc* d, int* nslabs, hipStream_t s);
size_t hv_wgrad_tr_workspace_bytes(const hv_wgrad_desc* d);                    // wgrad_tr.hip
int hv_wgrad_tr(const hv_wgrad_desc* d, int* nslabs, hipStream_t s);
int hv_wgrad_thin(const hv_wgrad_desc* d, int* nslabs, hipStream_t s);      // wgrad_tr.hip: thin operands (at most 4 input channels / 1-channel gradient carriers)
size_t hv_wgrad_thin_workspace_bytes(const hv_wgrad_desc* d);
int hv_conv2d_narrow(const hv_conv_desc* d, hipStream_t s);                     // conv_narrow.hip
int hv_conv2d_thin_in(const hv_conv_desc* d, hipStream_t s);
int hv_conv2d_stem5(const hv_conv_desc* d, hipStream_t s);
int hv_conv2d_stem5_dgrad(const hv_conv_desc* d, hipStream_t s);
int hv_conv2d_head(const hv_conv_desc* d, hipStream_t s);                       // conv_head.hip
int hv_conv2d_px(const hv_conv_desc* d, hipStream_t s);                         // conv_px.hip
int hv_conv2d_s2t(const hv_conv_desc* d, hipStream_t s);                        // conv_s2t.hip

struct ConvCls {
    int ph, pw, Hc, Wc, ntaps, Ktot, m0, mcount;
    uint32_t taps[HV_MAX_TAPS];  // dh(int8) | dw(int8)<<8 | widx<<16
};
struct ConvK {
    const void* x; const float* w; const float* bias; const float* scale; void* y;      // x / y: fp32 or fp16 elements (x_half / y_half)
    int x_half, y_half, mul_half;
    long long w_bs, scale_bs;
    int B, Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Cout, w_row, y_ld, y_coff, Ho, Wo;
    int bstep, boff, ostep;
    float alpha; int act, accumulate, vec_store, ncls;
    int xcd_swizzle;   // remap blockIdx.x so that each of the 8 XCDs (round-robin over the linear workgroup id) owns a CONTIGUOUS range of pixel tiles
    const void* mul_src; int mul_ld, mul_coff, mul_act, mul_vec;   // epilogue factor act'(mul_src[..]) or NULL; mul_vec: vector loads are aligned
    ConvCls cls[4];
};

template <typename T> struct Stage;
template <> struct Stage<float> {
    static constexpr int LD = HV_BK + 4;
    static __device__ __forceinline__ void st4(float* dst, float4 v) { *reinterpret_cast<float4*>(dst) = v; }
};
template <> struct Stage<_Float16> {
    static constexpr int LD = HV_BK + 8;
    static __device__ __forceinline__ void st4(_Float16* dst, float4 v) {
        f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<f16x4*>(dst) = h;
    }
};

template <typename T, int MT, int NT> struct Frags;
template <int MT, int NT> struct Frags<float, MT, NT> {
    float4 w[NT][2], x[MT][2];
    __device__ __forceinline__ void load(const float* Bs, const float* As, int wrow0, int xrow0, int lane) {
        const int off = (lane & 15) * Stage<float>::LD + (lane >> 4) * 8;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float* p = Bs + (wrow0 + n * 16) * Stage<float>::LD + off;
            w[n][0] = *reinterpret_cast<const float4*>(p);
            w[n][1] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* p = As + (xrow0 + m * 16) * Stage<float>::LD + off;
            x[m][0] = *reinterpret_cast<const float4*>(p);
            x[m][1] = *reinterpret_cast<const float4*>(p + 4);
        }
    }
    __device__ __forceinline__ void mma(f32x4 (&acc)[NT][MT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 a = acc[n][m];
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][0].x, x[m][0].x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][0].y, x[m][0].y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][0].z, x[m][0].z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][0].w, x[m][0].w, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][1].x, x[m][1].x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][1].y, x[m][1].y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][1].z, x[m][1].z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][1].w, x[m][1].w, a, 0, 0, 0);
                acc[n][m] = a;
            }
    }
};
template <int MT, int NT> struct Frags<_Float16, MT, NT> {
    f16x8 w[NT], x[MT];
    __device__ __forceinline__ void load(const _Float16* Bs, const _Float16* As, int wrow0, int xrow0, int lane) {
        const int off = (lane & 15) * Stage<_Float16>::LD + (lane >> 4) * 8;
#pragma unroll
        for (int n = 0; n < NT; ++n) w[n] = *reinterpret_cast<const f16x8*>(Bs + (wrow0 + n * 16) * Stage<_Float16>::LD + off);
#pragma unroll
        for (int m = 0; m < MT; ++m) x[m] = *reinterpret_cast<const f16x8*>(As + (xrow0 + m * 16) * Stage<_Float16>::LD + off);
    }
    __device__ __forceinline__ void mma(f32x4 (&acc)[NT][MT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[n], x[m], acc[n][m], 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------------ forward / transposed
template <typename T, int BM, int BN, int WM, int WN, bool ASC, bool XH>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK p) {
    constexpr int LD = Stage<T>::LD;
    constexpr int APASS = BM / 32;                  // 32 rows x 8 chunks of 4 floats per pass
    constexpr int BPASS = (BN + 31) / 32;
    constexpr int TMW = BM / WM, TNW = BN / WN;     // per-wave tile
    constexpr int MT = TMW / 16, NT = TNW / 16;
    static_assert(WM * WN == 4 && MT >= 1 && NT >= 1, "bad tile");
    __shared__ __attribute__((aligned(16))) T As[BM * LD];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LD];
    __shared__ uint32_t taps_s[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx = blockIdx.x;
    if (p.xcd_swizzle) {   // every tap re-reads a tile's input rows through L2: neighbouring tiles (the same images) belong on the same XCD's L2
        const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = bx & 7, loc = bx >> 3;
        bx = xcd < r ? xcd * (q + 1) + loc : r * (q + 1) + (xcd - r) * q + loc;
    }
    int ci = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < p.ncls && bx >= p.cls[i].m0) ci = i;
    const ConvCls& C = p.cls[ci];
    const int ntaps = C.ntaps, Ktot = C.Ktot, Hc = C.Hc, Wc = C.Wc, Mc = C.mcount, ph = C.ph, pw = C.pw;
    if (tid < ntaps) taps_s[tid] = C.taps[tid];
    const int m_base = (bx - C.m0) * BM;
    const int n_base = blockIdx.y * BN;
    const int HWc = Hc * Wc;

    // ---- per-thread gather rows
    const int chunk = tid & 7, prow = tid >> 3;
    int a_nb[APASS], a_hw[APASS];   // image base offset, packed (bh<<16)|(bw&0xffff)
    const int sample0 = m_base / HWc;  // only meaningful for per-sample weights (tile within one image)
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m_base + prow + i * 32;
        int bh = -20000, bw = -20000, nb = 0;
        if (m < Mc) {
            const int n = m / HWc, rem = m - n * HWc;
            const int ii = rem / Wc, jj = rem - ii * Wc;
            bh = ii * p.bstep + p.boff;
            bw = jj * p.bstep + p.boff;
            nb = n * p.img_stride;
        }
        a_nb[i] = nb;
        a_hw[i] = (int)(((uint32_t)bh << 16) | ((uint32_t)bw & 0xffffu));
    }
    const float* wbase = p.w + (p.w_bs ? (long long)sample0 * p.w_bs : 0ll);
    const int nk = (Ktot + HV_BK - 1) / HV_BK;

    float4 ra[APASS], rb[BPASS];
    auto gload = [&](int kt) {
        const int kg = kt * HV_BK + chunk * 4;
        if (!ASC) {
            const bool kok = kg < Ktot;
            int tapi = 0, c = 0, dh = 0, dw = 0, widx = 0;
            if (kok) {
                tapi = kg / p.Cin;
                c = kg - tapi * p.Cin;
                const uint32_t e = taps_s[tapi];
                dh = (int)(int8_t)(e & 0xff);
                dw = (int)(int8_t)((e >> 8) & 0xff);
                widx = (int)(e >> 16);
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int hi = (a_hw[i] >> 16) + dh, wi = (int)(int16_t)(a_hw[i] & 0xffff) + dw;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kok && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl) {
                    const int off = a_nb[i] + ((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + p.x_coff + c;
                    v = hv_ld4(p.x, off, XH);
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int r = prow + i * 32, n = n_base + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kok && r < BN && n < p.Cout) v = *reinterpret_cast<const float4*>(wbase + (long long)n * p.w_row + widx * p.Cin + c);
                rb[i] = v;
            }
        } else {
            int dh[4], dw[4], wo[4], cc[4];
            bool kok[4];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int k = kg + e4;
                kok[e4] = k < Ktot;
                const int tapi = kok[e4] ? k / p.Cin : 0;
                cc[e4] = k - tapi * p.Cin;
                const uint32_t e = taps_s[tapi];
                dh[e4] = (int)(int8_t)(e & 0xff);
                dw[e4] = (int)(int8_t)((e >> 8) & 0xff);
                wo[e4] = (int)(e >> 16) * p.Cin + cc[e4];
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float v[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int hi = (a_hw[i] >> 16) + dh[e4], wi = (int)(int16_t)(a_hw[i] & 0xffff) + dw[e4];
                    v[e4] = 0.f;
                    if (kok[e4] && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
                        v[e4] = hv_ld1(p.x, a_nb[i] + ((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + p.x_coff + cc[e4], XH);
                }
                ra[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int r = prow + i * 32, n = n_base + r;
                float v[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) v[e4] = (kok[e4] && r < BN && n < p.Cout) ? wbase[(long long)n * p.w_row + wo[e4]] : 0.f;
                rb[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < APASS; ++i) Stage<T>::st4(As + (prow + i * 32) * LD + chunk * 4, ra[i]);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int r = prow + i * 32;
            if (r < BN) Stage<T>::st4(Bs + r * LD + chunk * 4, rb[i]);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    __syncthreads();  // taps_s visible
    if (nk > 0) {
        gload(0);
        lstore();
    }
    __syncthreads();
    Frags<T, MT, NT> fr;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        fr.load(Bs, As, wn * TNW, wm * TMW, lane);
        fr.mma(acc);
        __syncthreads();
        if (more) lstore();
        __syncthreads();
    }

    // ---- epilogue: lane holds channels ch0..ch0+3 of pixel (lane & 15)
    const float* scale = p.scale ? p.scale + (p.scale_bs ? (long long)sample0 * p.scale_bs : 0ll) : nullptr;
    const HvEpi epi = {p.alpha, p.act, p.accumulate, p.vec_store, p.Cout, p.bias, scale, p.mul_act, p.mul_vec, p.y_half, p.mul_half};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mm = m_base + wm * TMW + m * 16 + (lane & 15);
        if (mm >= Mc) continue;
        const int n = mm / HWc, rem = mm - n * HWc;
        const int ii = rem / Wc, jj = rem - ii * Wc;
        const int ho = ph + ii * p.ostep, wo = pw + jj * p.ostep;
        const long long opix = (long long)(n * p.Ho + ho) * p.Wo + wo;
        void* yp = hv_eptr(p.y, opix * p.y_ld + p.y_coff, p.y_half);
        const void* mp = p.mul_src ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const int ch0 = n_base + wn * TNW + nn * 16 + (lane >> 4) * 4;
            hv_conv_epilogue4<!std::is_same<T, float>::value>(epi, acc[nn][m], ch0, yp, mp);       // fp16 operands: hardware exp2 / rcp activations
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool ASC>
static int launch_conv(const ConvK& k, int mtiles, hipStream_t s) {
    dim3 grid(mtiles, hv_cdiv(k.Cout, BN));
    const bool xh = sizeof(T) == 2 && k.x_half;
    HV_KNAME("conv_igemm_kernel<%s, %d, %d, %d, %d, %s, %s>", sizeof(T) == 2 ? "_Float16" : "float", BM, BN, WM, WN, ASC ? "true" : "false", xh ? "true" : "false");
    if constexpr (sizeof(T) == 2) {
        if (xh) {
            hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, ASC, true>), grid, dim3(256), 0, s, k);
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, ASC, false>), grid, dim3(256), 0, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

template <typename T, bool ASC>
static int dispatch_conv(ConvK& k, hipStream_t s) {
    // tile choice: BN = smallest tile covering Cout (<=128); BM 256 for narrow outputs, 128 otherwise,
    // 64 when the launch would not fill the chip.
    long long M = 0;
    for (int c = 0; c < k.ncls; ++c) M += k.cls[c].mcount;
    int BM, BN;
    if (k.Cout <= 16) { BN = 16; BM = 256; }
    else if (k.Cout <= 32) { BN = 32; BM = 256; }
    else if (k.Cout <= 64) { BN = 64; BM = 128; }
    else { BN = 128; BM = 128; }
    if (BN >= 64) {
        long long tiles = ((M + BM - 1) / BM) * ((k.Cout + BN - 1) / BN);
        static const int small_thr = getenv("HV_IGEMM_SMALL") ? atoi(getenv("HV_IGEMM_SMALL")) : 600;   // tuning knob (step-level A/B: 15.05 ms at 600 vs 15.16 at 160)
        if (tiles < small_thr) { BM = 64; BN = 64; }
    } else if (BN == 16 && (M + 255) / 256 < 256) {
        BM = 64;   // narrow outputs on small feature maps (PatchGAN logits): more, smaller workgroups
    }
    // (Attention paste -- per-sample filters, K = 4 taps x 1024 patches: every 64-pixel tile of a sample re-reads 1 MB of filters, 1.06 GB of
    // HBM fetches per launch = 8x the operands.  256-pixel tiles quarter the re-reads but leave one workgroup per CU: measured 212 us
    // against 168 us, step +0.1 ms; 128-pixel tiles: 174 us.  The fetches are served fast enough (MALL): the launch is bound by the gather's
    // issue / latency structure, not by HBM.  Not kept; a blocked attention kernel is the real fix.)
    if (k.w_bs || k.scale_bs) {  // per-sample operands: a tile must stay inside one image
        for (int c = 0; c < k.ncls; ++c)
            if ((k.cls[c].Hc * k.cls[c].Wc) % BM) {
                if ((k.cls[c].Hc * k.cls[c].Wc) % 64 == 0) { BM = 64; BN = 64; }
                else return HV_ERR_UNSUPPORTED;
            }
    }
    int mt = 0;
    for (int c = 0; c < k.ncls; ++c) {
        k.cls[c].m0 = mt;
        mt += hv_cdiv(k.cls[c].mcount, BM);
    }
    if (mt <= 0) return HV_OK;
    {   // XCD-contiguous tile ranges when the linear workgroup id modulo 8 is blockIdx.x modulo 8 (HV_XCD=0: A/B knob)
        static const int xcd = getenv("HV_XCD") ? atoi(getenv("HV_XCD")) : 1;
        const int gy = (k.Cout + BN - 1) / BN;
        k.xcd_swizzle = (xcd && mt >= 64 && (gy == 1 || (mt & 7) == 0)) ? 1 : 0;
    }
    if (BM == 256 && BN == 16) return launch_conv<T, 256, 16, 4, 1, ASC>(k, mt, s);
    if (BM == 64 && BN == 16) return launch_conv<T, 64, 16, 4, 1, ASC>(k, mt, s);
    if (BM == 256 && BN == 32) return launch_conv<T, 256, 32, 4, 1, ASC>(k, mt, s);
    if (BM == 128 && BN == 64) return launch_conv<T, 128, 64, 2, 2, ASC>(k, mt, s);
    if (BM == 128 && BN == 128) return launch_conv<T, 128, 128, 2, 2, ASC>(k, mt, s);
    return launch_conv<T, 64, 64, 2, 2, ASC>(k, mt, s);
}

// parts of hv_conv_desc.bstats: the kernels with that epilogue are the PatchGAN data gradients (logits layer, 4x4 stride 1 and stride 2)
extern "C" size_t hv_conv2d_bstats_parts(const hv_conv_desc* d) {
    if (!d || !d->transposed || d->precision != HV_F16) return 0;
    if (d->Cin == 4 && d->KH == 4 && d->stride == 1) return hv_conv2d_logits_bstats_parts(d);
    return hv_conv2d_g4_bstats_parts(d);
}

static int conv2d_dispatch(const hv_conv_desc* d, void* stream) {
    hv_wtable_used = 0;
    if (!d || !d->x || !d->w || !d->y) return HV_ERR_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 ||
        d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->Ho <= 0 || d->Wo <= 0)
        return HV_ERR_ARG;
    if (d->KH * d->KW > HV_MAX_TAPS || d->in_shift < 0 || d->in_shift > 1) return HV_ERR_UNSUPPORTED;
    if (d->in_shift && ((d->H | d->W) & 1)) return HV_ERR_UNSUPPORTED;
    if (d->x_ld < d->x_coff + d->Cin || d->y_ld < d->y_coff + d->Cout) return HV_ERR_ARG;
    if (d->act < HV_ACT_NONE || d->act > HV_ACT_CLAMP) return HV_ERR_ARG;
    if (d->precision != HV_F32 && d->precision != HV_F16) return HV_ERR_ARG;
    if (d->precision == HV_F32 && (d->x_f16 || d->y_f16 || d->mul_f16)) return HV_ERR_UNSUPPORTED;   // fp16 storage only under fp16 operands
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    if ((long long)d->B * Hp * Wp * d->x_ld >= (1ll << 31) || (long long)d->B * d->Ho * d->Wo * d->y_ld >= (1ll << 31))
        return HV_ERR_UNSUPPORTED;
    if (d->H > 16000 || d->W > 16000 || (d->KH - 1) * d->dil > 120 || (d->KW - 1) * d->dil > 120) return HV_ERR_UNSUPPORTED;

    // statistics epilogue asked for: only a kernel that has one may run (a silent fallback would leave the caller's partial sums unwritten and its
    // normalisation reading zeros) -- callers size `stats` with hv_conv2d_stats_parts, which is 0 exactly when this refuses
    if (d->stats && !hv_conv2d_g4_stats_floats(d, nullptr)) return HV_ERR_UNSUPPORTED;
    if (d->bstats && !hv_conv2d_bstats_parts(d)) return HV_ERR_UNSUPPORTED;      // (the same for the batch-norm backward sums of a data gradient)
    if (d->xn_stats) {   // input normalised at staging: the [pixel][tap] path of the 1-channel layers or nothing (the caller keeps the separate normalisation pass)
        if (d->Cout != 1 || !d->workspace) return HV_ERR_UNSUPPORTED;
        return hv_conv2d_head(d, (hipStream_t)stream);
    }
    if (d->x1) {         // extra input channel: the filters-in-LDS kernel or nothing (the caller keeps the materialised concat)
        if (d->precision != HV_F16 || !d->w_f16 || !d->w_f16_tiled || !d->y_f16 || d->transposed || d->dil != 1 || d->stride != 1 || d->KH != 3 || d->KW != 3 || !d->w1 ||
            d->pool2 || d->stats || d->x1_ld < 1)
            return HV_ERR_UNSUPPORTED;
        return hv_conv2d_halo(d, d->w_f16, (hipStream_t)stream);
    }
    if (d->pool2) {      // pooled data gradient: the filters-in-LDS kernel or nothing (the caller keeps the conv + copy form)
        if (d->precision != HV_F16 || !d->w_f16 || !d->y_f16 || (d->Ho & 1) || (d->Wo & 1) || d->dil != 1 || d->stride != 1 || d->KH != 3 || d->KW != 3 || d->bias ||
            d->act != HV_ACT_NONE || d->in_shift || d->stats)
            return HV_ERR_UNSUPPORTED;
        return hv_conv2d_halo(d, d->w_f16, (hipStream_t)stream);
    }
    if (d->precision == HV_F16 && d->Cin <= 16 && d->Cout <= 16 && d->stride == 1) {   // thin full-resolution layers: one lane per pixel on the 4x4x4 MFMA (conv_px.hip)
        const int rc = hv_conv2d_px(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    // single-channel heads / logits: VALU kernels (conv_narrow.hip), except where the halo-tiled MFMA kernel stages the input
    // once instead of once per tap (many input channels, fp16 mode)
    static const int narrow_max_cin = getenv("HV_NARROW_MAX_CIN") ? atoi(getenv("HV_NARROW_MAX_CIN")) : 15;   // A/B knob
    const bool halo_ok = d->precision == HV_F16 && d->w_f16 && d->dil == 1 && (d->Cin & 15) == 0 && !d->w_bstride && !d->ch_scale;
    if (d->Cout == 1 && d->workspace) {   // many input channels: taps as the GEMM's second dimension (conv_head.hip)
        const int rc = hv_conv2d_head(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->Cout == 1 && !d->transposed && !d->mul_src && !(halo_ok && d->Cin > narrow_max_cin)) {
        const int rc = hv_conv2d_narrow(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->Cin <= 4 && !d->transposed && !d->mul_src) {   // image-like inputs: direct fp32 VALU kernel, output-write bound (conv_narrow.hip)
        const int rc = hv_conv2d_thin_in(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->Cin == 4 && d->KH == 5 && d->precision == HV_F16 && d->w_f16) {   // 5x5 stems of the generators: (tap, channel) as one MFMA contraction
        const int rc = hv_conv2d_stem5(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->Cout == 4 && d->Cin == 16 && d->KH == 5 && d->transposed && d->precision == HV_F16 && d->w_f16) {   // ... and their data gradient
        const int rc = hv_conv2d_stem5_dgrad(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->precision == HV_F16 && d->transposed && d->Cin == 4 && d->KH == 4 && d->stride == 1) {   // the PatchGAN logits layer's data gradient (taps as the MFMA contraction)
        const int rc = hv_conv2d_logits_dgrad(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->precision == HV_F16 && d->transposed && d->Cin <= 4 && d->KH == 3) {   // the 1-channel heads' data gradient: one lane per pixel (conv_thin.hip)
        const int rc = hv_conv2d_thin_dgrad(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->precision == HV_F16 && d->w_f16 && d->KH == 4 && d->w_f16_tiled) {   // 4x4 layers (stride 2 and stride 1) and their data gradients as a pipelined implicit GEMM
        const int rc = hv_conv2d_g4(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->precision == HV_F16 && d->w_f16 && d->transposed && d->stride == 2) {   // stride-2 data gradients: the four parity classes in one workgroup
        const int rc = hv_conv2d_s2t(d, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if (d->precision == HV_F16 && d->w_f16) {   // halo-tiled fast path (conv_halo.hip) when the shape qualifies
        const int rc = hv_conv2d_halo(d, d->w_f16, (hipStream_t)stream);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    ConvK k;
    k.x_half = d->x_f16 ? 1 : 0; k.y_half = d->y_f16 ? 1 : 0; k.mul_half = d->mul_f16 ? 1 : 0;
    k.x = d->x; k.w = d->w; k.bias = d->bias; k.scale = d->ch_scale; k.y = d->y;
    k.w_bs = d->w_bstride; k.scale_bs = d->ch_scale ? d->ch_scale_bstride : 0;
    k.mul_src = d->mul_src; k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act;
    k.mul_vec = (d->mul_src && !(d->mul_ld & 3) && !(d->mul_coff & 3) && !((uintptr_t)d->mul_src & 15)) ? 1 : 0;
    k.B = d->B; k.Hl = d->H; k.Wl = d->W; k.in_shift = d->in_shift; k.Wp = Wp; k.img_stride = Hp * Wp * d->x_ld;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Cout = d->Cout; k.w_row = d->KH * d->KW * d->Cin; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Ho = d->Ho; k.Wo = d->Wo;
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    k.vec_store = ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && ((uintptr_t)d->y & 15) == 0) ? 1 : 0;
    if (!d->transposed) {
        const int eh = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
        const int ew = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
        if (eh != d->Ho || ew != d->Wo) return HV_ERR_ARG;
        k.ncls = 1; k.bstep = d->stride; k.boff = -d->pad; k.ostep = 1;
        ConvCls& c = k.cls[0];
        c.ph = c.pw = 0; c.Hc = d->Ho; c.Wc = d->Wo; c.ntaps = d->KH * d->KW; c.Ktot = c.ntaps * d->Cin;
        c.mcount = d->B * d->Ho * d->Wo;
        for (int r = 0; r < d->KH; ++r)
            for (int s = 0; s < d->KW; ++s) {
                const int t = r * d->KW + s;
                c.taps[t] = (uint32_t)(uint8_t)(int8_t)(r * d->dil) | ((uint32_t)(uint8_t)(int8_t)(s * d->dil) << 8) | ((uint32_t)t << 16);
            }
    } else {
        if (d->stride > 2) return HV_ERR_UNSUPPORTED;
        k.bstep = 1; k.boff = 0; k.ostep = d->stride; k.ncls = 0;
        for (int ph = 0; ph < d->stride; ++ph)
            for (int pw = 0; pw < d->stride; ++pw) {
                ConvCls& c = k.cls[k.ncls];
                c.ph = ph; c.pw = pw;
                c.Hc = (d->Ho - ph + d->stride - 1) / d->stride;
                c.Wc = (d->Wo - pw + d->stride - 1) / d->stride;
                if (c.Hc <= 0 || c.Wc <= 0) continue;
                c.ntaps = 0;
                for (int r = 0; r < d->KH; ++r) {
                    const int vh = ph + d->pad - r * d->dil;
                    if (((vh % d->stride) + d->stride) % d->stride) continue;
                    for (int s = 0; s < d->KW; ++s) {
                        const int vw = pw + d->pad - s * d->dil;
                        if (((vw % d->stride) + d->stride) % d->stride) continue;
                        const int dh = vh / d->stride, dw = vw / d->stride;
                        if (dh < -127 || dh > 127 || dw < -127 || dw > 127) return HV_ERR_UNSUPPORTED;
                        c.taps[c.ntaps++] = (uint32_t)(uint8_t)(int8_t)dh | ((uint32_t)(uint8_t)(int8_t)dw << 8) |
                                            ((uint32_t)(r * d->KW + s) << 16);
                    }
                }
                c.Ktot = c.ntaps * d->Cin;
                c.mcount = d->B * c.Hc * c.Wc;
                ++k.ncls;
            }
        if (k.ncls == 0) return HV_ERR_ARG;
    }
    const bool vec_in = (d->Cin & 3) == 0 && (d->x_ld & 3) == 0 && (d->x_coff & 3) == 0 && ((uintptr_t)d->x & 15) == 0 &&
                        ((uintptr_t)d->w & 15) == 0 && (d->w_bstride & 3) == 0;
    hipStream_t s = (hipStream_t)stream;
    hv_path_note = 0;
    HV_WUSE(1);      // the gather kernel reads the fp32 table (and converts)
    if (d->precision == HV_F32) return vec_in ? dispatch_conv<float, false>(k, s) : dispatch_conv<float, true>(k, s);
    return vec_in ? dispatch_conv<_Float16, false>(k, s) : dispatch_conv<_Float16, true>(k, s);
}

// Would hv_conv2d serve this descriptor?  Only the forms WITHOUT a generic fallback can be refused for their shape -- the extra input channel (x1), the
// pooled data gradient (pool2), a statistics epilogue (stats) -- so callers ask before they drop the materialised alternative (engine.ConvNode.split_forward,
// ops.pool2_ok).  The answer comes from the dispatch itself run in probe mode (every launch site returns before its launch): no mirror of the checks.
thread_local int hv_probe_only = 0;
extern "C" int hv_conv2d_supported(const hv_conv_desc* d) {
    if (!d) return 0;
    if (d->bstats && !hv_conv2d_bstats_parts(d)) return 0;
    if (!d->x1 && !d->pool2 && !d->xn_stats) return d->stats ? (hv_conv2d_g4_stats_floats(d, nullptr) ? 1 : 0) : 1;
    hv_probe_only = 1;
    const int rc = conv2d_dispatch(d, nullptr);
    hv_probe_only = 0;
    return rc == HV_OK ? 1 : 0;
}

// profiling: with hv_set_kernel_timing armed, the two events are recorded here, microseconds apart on the host, right around the launch(es) of
// this convolution -- an event pair recorded from Python around the ctypes call also times the host's own latency whenever the GPU runs dry
extern "C" int hv_conv2d(const hv_conv_desc* d, void* stream) {
    HV_TIMING_BEGIN((hipStream_t)stream);
    const int rc = conv2d_dispatch(d, stream);
    HV_TIMING_END((hipStream_t)stream);
    return rc;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// D[co][j] = sum_pix G[pix][co] * X[pix(tap_j)][c_j],  j = (tap, ci) flattened.  A workgroup owns a BN x BC tile
// of dW and one chunk of pixels.  Staging is the issue-bound part of this kernel (the 16x16x32 MFMA leaves 8 of its
// 16 cycles to other vector instructions), so it is kept branch-free: one item = 8 consecutive pixels x 4 channels
// fetched with 8 buffer_load_dwordx4 whose out-of-image / out-of-chunk lanes get an out-of-range offset (the
// descriptor's range check returns zeros), addresses advance by a constant per pixel when the output width is a
// power of two (FAST), and the next tile's loads are issued before this tile's MFMAs.
struct WgradK {
    const void* x; const void* g; float* out;      // x / g: fp32 elements, or fp16 with the SH instantiations (half)
    int half;
    int B, Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Ho, Wo, g_ld, g_coff, Cout;
    int KW, stride, pad, dil, J /* taps*Cin */, M /* B*Ho*Wo */, chunk;
    int lw, lhw;                 // FAST: log2(Wo), log2(Ho*Wo)
    int kt_q, kt_r;              // !FAST: KT / Wo, KT % Wo
    unsigned x_bytes, g_bytes;   // buffer descriptor ranges
    long long slab;              // Cout*J floats per split
    float* bias_out;             // per-split column sums of g (bias gradient), [splits][Cout], or NULL
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define HV_OOB 0x80000000u       // beyond every descriptor range (tensors are < 2 GiB): the load returns zeros

template <typename T, int BN, int BC, int WN, int WC, int KT, bool FAST, bool SH>
__global__ __launch_bounds__(WN * WC * 64) void wgrad_kernel(const WgradK p) {
    typedef HvSt<SH> SS;                         // storage of x and g: fp32 (converted when staged) or fp16
    typedef typename SS::R SR;
    constexpr unsigned SB = SS::B;
    constexpr int NTHR = WN * WC * 64;
    constexpr bool F16 = sizeof(T) == 2;
    // fp32: [pixel][ch] (ch contiguous, padded); fp16: [ch][pixel] (pixel contiguous, +8 pad) so that a lane
    // reads 8 contiguous k (= pixels) for the 16x16x32 MFMA.
    // fp32 rows are padded so that row stride = 16 (mod 32) banks: the two 16-lane halves of a ds_read_b32
    // (pixels k and k+1) then hit disjoint banks.
    // fp16 rows are KT + 16 halfs (row stride = 8 mod 16 dwords, odd multiple of 8) and the 16-B column of a row is
    // XOR-swizzled with (row >> 2) & 7: the four 16-lane groups of a ds_read_b128 (16 rows x two k-groups) then cover
    // all 64 banks once, and so do the 8 lanes of a staging ds_write_b128 group (8 channel groups, same pixel run).
    constexpr int LDG = F16 ? (KT + 16) : (BN + ((BN % 32 == 16) ? 0 : 16));
    constexpr int LDX = F16 ? (KT + 16) : (BC + ((BC % 32 == 16) ? 0 : 16));
    __shared__ __attribute__((aligned(16))) T Gs[F16 ? BN * LDG : KT * LDG];
    __shared__ __attribute__((aligned(16))) T Xs[F16 ? BC * LDX : KT * LDX];
    constexpr int TNW = BN / WN, TCW = BC / WC, NT = TNW / 16, CT = TCW / 16;
    static_assert(WN * WC == 4 && NT >= 1 && CT >= 1 && KT % 32 == 0 && (!F16 || KT % 64 == 0), "bad tile");
    constexpr int GI = (BN / 4) * (KT / 8);     // staging items: (channel groups of 4) x (pixel runs of 8)
    constexpr int XI = (BC / 4) * (KT / 8);
    constexpr int XPT = (XI + NTHR - 1) / NTHR; // X items per thread (wide J tiles re-read G fewer times)
    static_assert(GI <= NTHR && (XI <= NTHR || XI % NTHR == 0), "one G item per thread; X items fill whole thread passes");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_base = blockIdx.y * BN, j_base = blockIdx.z * BC;
    const int pix_begin = blockIdx.x * p.chunk;
    const int pix_end = min(p.M, pix_begin + p.chunk);
    const int HWo = p.Ho * p.Wo;

    // G items sit on the first threads, X items on the last: when a tile has fewer items than threads the two
    // roles land on different waves (different SIMDs) and stage in parallel.
    const bool has_g = tid < GI;
    const int xt = XI < NTHR ? tid - (NTHR - XI) : tid;
    const bool has_x = xt >= 0;
    const int g_cg = tid % (BN / 4), g_run = tid / (BN / 4);
    int x_cg[XPT], x_run[XPT], dh[XPT], dw[XPT], cx[XPT];
    bool jok[XPT];
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
        const int xi = (has_x ? xt : 0) + it * NTHR;
        x_cg[it] = xi % (BC / 4);
        x_run[it] = xi / (BC / 4);
        const int jx = j_base + x_cg[it] * 4;
        jok[it] = has_x && jx < p.J;
        dh[it] = dw[it] = cx[it] = 0;
        if (jok[it]) {
            const int tap = jx / p.Cin;
            cx[it] = jx - tap * p.Cin;
            const int r = tap / p.KW, q = tap - r * p.KW;
            dh[it] = r * p.dil - p.pad;
            dw[it] = q * p.dil - p.pad;
        }
    }
    const int gch = n_base + g_cg * 4;
    const bool gok = has_g && gch < p.Cout;  // Cout % 4 == 0 is guaranteed by the host

    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0, p.g_bytes, 0x00020000);

    // decode state of this thread's X items (first pixel of the run) for the non-power-of-two path; load_x is called
    // with pix_begin, pix_begin + KT, ... in order and advances it
    int xs_n[XPT], xs_ho[XPT], xs_wo[XPT];
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
        xs_n[it] = xs_ho[it] = xs_wo[it] = 0;
        if (!FAST) {
            const int m0 = pix_begin + x_run[it] * 8;
            xs_n[it] = m0 / HWo;
            const int rem = m0 - xs_n[it] * HWo;
            xs_ho[it] = rem / p.Wo;
            xs_wo[it] = rem - xs_ho[it] * p.Wo;
        }
    }
    SR rg[8], rx[XPT][8];
    auto load_g = [&](int pix0) __attribute__((always_inline)) {
        const int m0 = pix0 + g_run * 8;
        if (FAST) {   // M % 8 == 0 and chunk % KT == 0: a run is wholly inside or outside the chunk
            unsigned off = (gok && m0 < pix_end) ? (unsigned)(m0 * p.g_ld + p.g_coff + gch) * SB : HV_OOB;
            const unsigned step = (unsigned)p.g_ld * SB;
#pragma unroll
            for (int e = 0; e < 8; ++e, off += step) rg[e] = SS::ld(gsrc, off, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned off = (gok && m0 + e < pix_end) ? (unsigned)((m0 + e) * p.g_ld + p.g_coff + gch) * SB : HV_OOB;
                rg[e] = SS::ld(gsrc, off, 0);
            }
        }
    };
    auto load_x1 = [&](int pix0, int it, SR (&r)[8]) __attribute__((always_inline)) {
        const int m0 = pix0 + x_run[it] * 8;
        if (FAST) {   // Wo is a power of two >= 8: the 8 pixels of a run share one output row
            const int n = m0 >> p.lhw, ho = (m0 >> p.lw) & (p.Ho - 1), wo0 = m0 & (p.Wo - 1);
            const int hi = ho * p.stride + dh[it];
            const bool rok = jok[it] && m0 < pix_end && (unsigned)hi < (unsigned)p.Hl;
            const int base = n * p.img_stride + (hi >> p.in_shift) * p.Wp * p.x_ld + p.x_coff + cx[it];
            int wi = wo0 * p.stride + dw[it];
#pragma unroll
            for (int e = 0; e < 8; ++e, wi += p.stride) {
                const unsigned off = (rok && (unsigned)wi < (unsigned)p.Wl) ? (unsigned)(base + (wi >> p.in_shift) * p.x_ld) * SB : HV_OOB;
                r[e] = SS::ld(xsrc, off, 0);
            }
        } else if (p.Wo >= 8) {   // any output size: the run crosses an output row at most once -- two row bases, one select per pixel
            const int cross = p.Wo - xs_wo[it];                   // pixels e >= cross sit on the next output row
            int ho1 = xs_ho[it] + 1, n1 = xs_n[it];
            if (ho1 == p.Ho) { ho1 = 0; ++n1; }
            const int hiA = xs_ho[it] * p.stride + dh[it], hiB = ho1 * p.stride + dh[it];
            const bool rokA = jok[it] && (unsigned)hiA < (unsigned)p.Hl, rokB = jok[it] && (unsigned)hiB < (unsigned)p.Hl;
            const int baseA = xs_n[it] * p.img_stride + (hiA >> p.in_shift) * p.Wp * p.x_ld + p.x_coff + cx[it];
            const int baseB = n1 * p.img_stride + (hiB >> p.in_shift) * p.Wp * p.x_ld + p.x_coff + cx[it];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool nb = e >= cross;
                const int wi = (xs_wo[it] + e - (nb ? p.Wo : 0)) * p.stride + dw[it];
                const bool ok = (nb ? rokB : rokA) && m0 + e < pix_end && (unsigned)wi < (unsigned)p.Wl;
                const unsigned off = ok ? (unsigned)((nb ? baseB : baseA) + (wi >> p.in_shift) * p.x_ld) * SB : HV_OOB;
                r[e] = SS::ld(xsrc, off, 0);
            }
        } else {                  // tiny maps: step pixel by pixel
            int n = xs_n[it], ho = xs_ho[it], wo = xs_wo[it];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int hi = ho * p.stride + dh[it], wi = wo * p.stride + dw[it];
                const bool ok = jok[it] && m0 + e < pix_end && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl;
                const unsigned off = ok ? (unsigned)(n * p.img_stride + ((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + p.x_coff + cx[it]) * SB : HV_OOB;
                r[e] = SS::ld(xsrc, off, 0);
                if (++wo == p.Wo) { wo = 0; if (++ho == p.Ho) { ho = 0; ++n; } }
            }
        }
        if (!FAST) {
            // advance the decode state by one step (KT pixels) without dividing
            xs_wo[it] += p.kt_r; xs_ho[it] += p.kt_q;
            if (xs_wo[it] >= p.Wo) { xs_wo[it] -= p.Wo; ++xs_ho[it]; }
            while (xs_ho[it] >= p.Ho) { xs_ho[it] -= p.Ho; ++xs_n[it]; }
        }
    };
    auto load_x = [&](int pix0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XPT; ++it) load_x1(pix0, it, rx[it]);
    };
    auto store = [&](T* dst, int ld, const SR (&r)[8], int cg, int run) __attribute__((always_inline)) {
        if constexpr (F16) {
            // transpose 8 pixels x 4 channels -> 4 rows of 8 halfs
            f16x8 h0, h1, h2, h3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f16x4 q = SS::h4(r[e]);
                h0[e] = q[0]; h1[e] = q[1]; h2[e] = q[2]; h3[e] = q[3];
            }
            _Float16* base = reinterpret_cast<_Float16*>(dst) + (cg * 4) * ld + (run ^ (cg & 7)) * 8;
            *reinterpret_cast<f16x8*>(base) = h0;
            *reinterpret_cast<f16x8*>(base + ld) = h1;
            *reinterpret_cast<f16x8*>(base + 2 * ld) = h2;
            *reinterpret_cast<f16x8*>(base + 3 * ld) = h3;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + (run * 8 + e) * ld + cg * 4) = SS::f4(r[e]);
        }
    };

    f32x4 acc[NT][CT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave / WC, wc = wave % WC;
    // bias gradient: the first column tile also sums the g values it stages (fp32, before the fp16 conversion)
    const bool do_bias = p.bias_out != nullptr && blockIdx.z == 0;
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;

    if (pix_begin < pix_end) {
        if (has_g) load_g(pix_begin);
        if (has_x) load_x(pix_begin);
    }
    for (int pix0 = pix_begin; pix0 < pix_end; pix0 += KT) {
        __syncthreads();  // previous step's MFMA reads done
        if (has_g) store(Gs, LDG, rg, g_cg, g_run);
        if (do_bias && has_g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 q = SS::f4(rg[e]);
                bs0 += q.x; bs1 += q.y; bs2 += q.z; bs3 += q.w;
            }
        }
        if (has_x) {
#pragma unroll
            for (int it = 0; it < XPT; ++it) store(Xs, LDX, rx[it], x_cg[it], x_run[it]);
        }
        __syncthreads();
        if (pix0 + KT < pix_end) {   // next tile's loads fly behind this tile's MFMAs
            if (has_g) load_g(pix0 + KT);
            if (has_x) load_x(pix0 + KT);
        }
        if constexpr (F16) {
            const _Float16* G16 = reinterpret_cast<const _Float16*>(Gs);
            const _Float16* X16 = reinterpret_cast<const _Float16*>(Xs);
#pragma unroll
            for (int ks = 0; ks < KT / 32; ++ks) {
                f16x8 a[NT], b[CT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int row = wn * TNW + n * 16 + (lane & 15);
                    a[n] = *reinterpret_cast<const f16x8*>(G16 + row * LDG + ((ks * 4 + (lane >> 4)) ^ ((row >> 2) & 7)) * 8);
                }
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const int row = wc * TCW + c * 16 + (lane & 15);
                    b[c] = *reinterpret_cast<const f16x8*>(X16 + row * LDX + ((ks * 4 + (lane >> 4)) ^ ((row >> 2) & 7)) * 8);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b[c], acc[n][c], 0, 0, 0);
            }
        } else {
            const float* G32 = reinterpret_cast<const float*>(Gs);
            const float* X32 = reinterpret_cast<const float*>(Xs);
#pragma unroll
            for (int k4 = 0; k4 < KT / 4; ++k4) {
                const int pk = k4 * 4 + (lane >> 4);
                float a[NT], b[CT];
#pragma unroll
                for (int n = 0; n < NT; ++n) a[n] = G32[pk * LDG + wn * TNW + n * 16 + (lane & 15)];
#pragma unroll
                for (int c = 0; c < CT; ++c) b[c] = X32[pk * LDX + wc * TCW + c * 16 + (lane & 15)];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[n][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n], b[c], acc[n][c], 0, 0, 0);
            }
        }
    }
    if (do_bias) {   // fold the pixel runs of a channel group in LDS (fixed order), one row per split
        __syncthreads();
        float* bsh = reinterpret_cast<float*>(Xs);          // GI x 4 floats <= the X tile
        if (has_g) { bsh[tid * 4 + 0] = bs0; bsh[tid * 4 + 1] = bs1; bsh[tid * 4 + 2] = bs2; bsh[tid * 4 + 3] = bs3; }
        __syncthreads();
        if (tid < BN && n_base + tid < p.Cout) {
            const int cg = tid >> 2, c = tid & 3;
            float t = 0.f;
            for (int run = 0; run < KT / 8; ++run) t += bsh[(run * (BN / 4) + cg) * 4 + c];
            p.bias_out[(long long)blockIdx.x * p.Cout + n_base + tid] = t;
        }
    }
    // D layout: row (= co) = (lane>>4)*4 + r, col (= j) = lane & 15
    float* out = p.out + (long long)blockIdx.x * p.slab;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int j = j_base + wc * TCW + c * 16 + (lane & 15);
            if (j >= p.J) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = n_base + wn * TNW + n * 16 + (lane >> 4) * 4 + r;
                if (co < p.Cout) out[(long long)co * p.J + j] = acc[n][c][r];
            }
        }
}

// Sum of the split-K slabs, in a fixed order (deterministic).  256 threads = Q element quads (16 bytes per lane and slab) x G split groups, Q * G =
// 256: the layers with a small dW are the ones cut into many slabs (up to 512), and with 4 groups a lane walked 128 dependent-latency iterations
// (measured 30 us for 19 MB, twice the weight-gradient kernel itself); G follows the slab count so that a lane issues <= 8 loads, all in flight.
// Blocks beyond the dW range fold the per-split bias rows (bslabs [splits][nb]) into dbias.  n and nb are multiples of 4 (channel counts are).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, long long n, int splits, int accumulate,
                                                           const float* __restrict__ bslabs, float* __restrict__ dbias, int nb, int bias_accumulate) {
    constexpr int Q = 256 / G;
    __shared__ float4 sh[G][Q];
    const int e = threadIdx.x % Q, grp = threadIdx.x / Q;
    const long long wblocks = (n / 4 + Q - 1) / Q;
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    const bool bias_block = (long long)blockIdx.x >= wblocks;       // block-uniform
    // bias rows: one element per lane (dbias is a view into the flat gradient buffer: any 4-byte offset)
    const long long i = bias_block ? ((long long)blockIdx.x - wblocks) * Q + e : ((long long)blockIdx.x * Q + e) * 4;
    const bool live = i < (bias_block ? (long long)nb : n);
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int k = grp;
        if (bias_block) {
            for (; k + 3 * G < splits; k += 4 * G)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u].x += bslabs[(long long)(k + u * G) * nb + i];
            for (; k < splits; k += G) acc[0].x += bslabs[(long long)k * nb + i];
        } else {
            for (; k + 3 * G < splits; k += 4 * G)
#pragma unroll
                for (int u = 0; u < 4; ++u) add(acc[u], *reinterpret_cast<const float4*>(slabs + (long long)(k + u * G) * n + i));
            for (; k < splits; k += G) add(acc[0], *reinterpret_cast<const float4*>(slabs + (long long)k * n + i));
        }
    }
    add(acc[0], acc[1]); add(acc[2], acc[3]); add(acc[0], acc[2]);
    sh[grp][e] = acc[0];
    __syncthreads();
#pragma unroll
    for (int h = G / 2; h >= 1; h >>= 1) {
        if (grp < h) add(sh[grp][e], sh[grp + h][e]);
        __syncthreads();
    }
    if (grp == 0 && live) {
        float4 t = sh[0][e];
        if (bias_block) dbias[i] = bias_accumulate ? dbias[i] + t.x : t.x;
        else {
            if (accumulate) add(t, *reinterpret_cast<const float4*>(dw + i));
            *reinterpret_cast<float4*>(dw + i) = t;
        }
    }
}

thread_local HvFold hv_carry = {};
thread_local int hv_carry_taken = 0;
// the fold of this call: now, or recorded for the caller to hand to the NEXT weight gradient (hv_wgrad_desc.pending -> hv_wgrad_desc.carry) / hv_wgrad_fold_now
static int launch_wgrad_reduce(const float* slabs, float* dw, long long n, int splits, int accumulate, const float* bslabs, float* dbias, int nb,
                                int bias_accumulate, hipStream_t s);
static int wgrad_fold_or_defer(const hv_wgrad_desc* d, const float* slabs, long long n, int splits, const float* bslabs, hipStream_t s) {
    if (d->pending && !(n & 3) && !(((uintptr_t)slabs | (uintptr_t)d->dw) & 15)) {
        hv_wgrad_fold* f = d->pending;
        f->slabs = slabs; f->dw = d->dw; f->numel = n; f->nslabs = splits; f->accumulate = d->accumulate;
        f->bias_slabs = d->dbias ? bslabs : nullptr; f->dbias = d->dbias; f->Cout = d->Cout; f->dbias_accumulate = d->dbias_accumulate;
        return HV_OK;
    }
    if (d->pending) d->pending->nslabs = 0;      // (folded here after all)
    return launch_wgrad_reduce(slabs, d->dw, n, splits, d->accumulate, bslabs, d->dbias, d->Cout, d->dbias_accumulate, s);
}
extern "C" int hv_wgrad_fold_now(const hv_wgrad_fold* f, void* stream) {
    if (!f) return HV_ERR_ARG;
    if (f->nslabs <= 0) return HV_OK;
    if (!f->slabs || !f->dw || f->numel <= 0 || (f->numel & 3)) return HV_ERR_ARG;
    return launch_wgrad_reduce(f->slabs, f->dw, f->numel, f->nslabs, f->accumulate, f->bias_slabs, f->dbias, f->Cout, f->dbias_accumulate, (hipStream_t)stream);
}
static int launch_wgrad_reduce(const float* slabs, float* dw, long long n, int splits, int accumulate, const float* bslabs, float* dbias, int nb,
                                int bias_accumulate, hipStream_t s) {
    static const int skip_red = getenv("HV_DIAG_SKIP") && strstr(getenv("HV_DIAG_SKIP"), "wgrad_reduce") ? 1 : 0;     // timing-only diagnostic (wrong results)
    if (skip_red) return HV_OK;
#define HV_RED(G_)                                                                                                                          \
    hipLaunchKernelGGL(wgrad_reduce_kernel<G_>, dim3(hv_cdiv(n / 4, 256 / G_) + (dbias ? hv_cdiv(nb, 256 / G_) : 0)), dim3(256), 0, s, slabs, dw, n, splits, \
                       accumulate, bslabs, dbias, nb, bias_accumulate)
    if (splits <= 16) HV_RED(4);
    else if (splits <= 64) HV_RED(16);
    else HV_RED(32);
#undef HV_RED
    HV_LAUNCH_CHECK();
    return HV_OK;
}

struct WgradPlan { int BN, BC, KT, splits, chunk; };
static int wgrad_plan(const hv_wgrad_desc* d, WgradPlan* pl) {
    const int J = d->KH * d->KW * d->Cin;
    const long long M = (long long)d->B * d->Ho * d->Wo;
    static const bool wide64 = !(getenv("HV_WGRAD_WIDE") && atoi(getenv("HV_WGRAD_WIDE")) == 0);   // A/B knob
    static const int bc128_minj = getenv("HV_WGRAD_BC128_MINJ") ? atoi(getenv("HV_WGRAD_BC128_MINJ")) : 4096;   // tuning knob
    int BN, BC;
    if (d->Cout <= 16) { BN = 16; BC = 128; }
    else if (d->Cout <= 32) { BN = 32; BC = 128; }
    else if (d->Cout <= 64) { BN = 64; BC = (J >= 512 && wide64) ? 256 : 64; }   // wide J tile: G is re-read J/256 instead of J/64 times
    else { BN = 128; BC = J >= bc128_minj ? 128 : 64; }
    const long long tiles = (long long)hv_cdiv(d->Cout, BN) * hv_cdiv(J, BC);
    // ~2 workgroups per CU; layers whose dW already has many tiles get few splits (slab traffic grows with splits)
    static const int want_wg = getenv("HV_WGRAD_WANT") ? atoi(getenv("HV_WGRAD_WANT")) : 512;   // tuning knob (step time 15.13 ms at 512 vs 15.33 at 384, 15.46 at 640, 15.56 at 768+)
    static const int want_big = getenv("HV_WGRAD_WANT_BIG") ? atoi(getenv("HV_WGRAD_WANT_BIG")) : 512;   // tuning knob (layers with >= 128 tiles)
    long long want = tiles >= 128 ? (want_big + tiles - 1) / tiles : (want_wg + tiles - 1) / tiles;
    if (want > 256) want = 256;
    long long maxs = (M + 255) / 256;                     // at least 256 pixels per split
    long long splits = want < 1 ? 1 : want;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    // pixels per staging step: fp16 tiles take 64 (128 for the 64x64 tile) so that every thread stages an X item
    const int KT = d->precision == HV_F32 ? 32 : ((BN == 64 && BC == 64) ? 128 : 64);
    long long chunk = (M + splits - 1) / splits;
    chunk = (chunk + KT - 1) / KT * KT;
    splits = (M + chunk - 1) / chunk;
    pl->BN = BN; pl->BC = BC; pl->KT = KT; pl->splits = (int)splits; pl->chunk = (int)chunk;
    return HV_OK;
}

static int wgrad_validate(const hv_wgrad_desc* d) {
    if (!d || !d->x || !d->g || !d->dw) return HV_ERR_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->dil <= 0 || d->pad < 0 || d->Ho <= 0 || d->Wo <= 0)
        return HV_ERR_ARG;
    if ((d->Cin & 3) || (d->Cout & 3) || (d->x_ld & 3) || (d->x_coff & 3) || (d->g_ld & 3) || (d->g_coff & 3)) return HV_ERR_UNSUPPORTED;
    if (((uintptr_t)d->x & 15) || ((uintptr_t)d->g & 15)) return HV_ERR_UNSUPPORTED;
    if (d->in_shift < 0 || d->in_shift > 1) return HV_ERR_UNSUPPORTED;
    if (d->precision != HV_F32 && d->precision != HV_F16) return HV_ERR_ARG;
    // storage follows the compute type: fp16 MFMA operands <=> x and g stored as fp16, exact-fp32 mode <=> fp32 storage
    if ((d->x_f16 != 0) != (d->g_f16 != 0) || (d->x_f16 != 0) != (d->precision == HV_F16)) return HV_ERR_UNSUPPORTED;
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    // byte offsets are 32-bit buffer offsets with 0x80000000 as the out-of-range marker: tensors stay below 2 GiB
    if ((long long)d->B * Hp * Wp * d->x_ld >= (1ll << 29) || (long long)d->B * d->Ho * d->Wo * d->g_ld >= (1ll << 29)) return HV_ERR_UNSUPPORTED;
    return HV_OK;
}

extern "C" size_t hv_conv2d_wgrad_workspace_bytes(const hv_wgrad_desc* d) {
    if (wgrad_validate(d) != HV_OK) return 0;
    const size_t thin = hv_wgrad_thin_workspace_bytes(d);
    if (thin) return thin;
    const size_t halo = hv_wgrad_halo_workspace_bytes(d);
    if (halo) return halo;
    const size_t trb = hv_wgrad_tr_workspace_bytes(d);
    if (trb) return trb;
    WgradPlan pl;
    wgrad_plan(d, &pl);
    if (pl.splits <= 1 && !d->accumulate && !d->dbias) return 0;
    return (size_t)pl.splits * ((size_t)d->Cout * d->KH * d->KW * d->Cin + (d->dbias ? d->Cout : 0)) * sizeof(float);
}

template <typename T, bool FAST>
static int launch_wgrad(const WgradK& k, const WgradPlan& pl, hipStream_t s) {
    constexpr bool F16 = sizeof(T) == 2;
    constexpr int KT = F16 ? 64 : 32, KT64 = F16 ? 128 : 32;
    dim3 grid(pl.splits, hv_cdiv(k.Cout, pl.BN), hv_cdiv(k.J, pl.BC));
    if (pl.KT != ((pl.BN == 64 && pl.BC == 64) ? KT64 : KT)) return HV_ERR_ARG;
    {
        const int bn = pl.BN, bc = pl.BN == 16 || pl.BN == 32 ? 128 : pl.BC, wn = (pl.BN <= 32 || (pl.BN == 64 && pl.BC == 256)) ? 1 : 2;
        HV_KNAME("wgrad_kernel<%s, %d, %d, %d, %d, %d, %s>", F16 ? "_Float16" : "float", bn, bc, wn, 4 / wn, pl.KT, FAST ? "true" : "false");
    }
    constexpr bool SH = F16;        // fp16 compute <=> fp16 storage of x and g (checked by the caller); fp32 compute <=> fp32 storage
    if (pl.BN == 16) hipLaunchKernelGGL((wgrad_kernel<T, 16, 128, 1, 4, KT, FAST, SH>), grid, dim3(256), 0, s, k);
    else if (pl.BN == 32) hipLaunchKernelGGL((wgrad_kernel<T, 32, 128, 1, 4, KT, FAST, SH>), grid, dim3(256), 0, s, k);
    else if (pl.BN == 64 && pl.BC == 256) hipLaunchKernelGGL((wgrad_kernel<T, 64, 256, 1, 4, KT, FAST, SH>), grid, dim3(256), 0, s, k);
    else if (pl.BN == 64) hipLaunchKernelGGL((wgrad_kernel<T, 64, 64, 2, 2, KT64, FAST, SH>), grid, dim3(256), 0, s, k);
    else if (pl.BC == 128) hipLaunchKernelGGL((wgrad_kernel<T, 128, 128, 2, 2, KT, FAST, SH>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((wgrad_kernel<T, 128, 64, 2, 2, KT, FAST, SH>), grid, dim3(256), 0, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

static int wgrad_dispatch(const hv_wgrad_desc* d, void* stream);
extern "C" int hv_conv2d_wgrad(const hv_wgrad_desc* d, void* stream) {
    int rc = wgrad_validate(d);
    if (rc != HV_OK) return rc;
    // the carried fold (hv_wgrad_desc.carry): offered to the launcher of this call's main kernel; one that cannot take it along leaves it to a launch of its own, first
    hv_carry.splits = 0;
    hv_carry_taken = 0;
    const hv_wgrad_fold* c = d->carry;
    if (c && c->nslabs > 0) {
        if (!c->slabs || !c->dw || c->numel <= 0 || (c->numel & 3)) return HV_ERR_ARG;
        static const int carry_on = getenv("HV_FOLD_CARRY") ? atoi(getenv("HV_FOLD_CARRY")) : 1;      // A/B knob (0: every carried fold as its own launch)
        const bool overlap = (c->slabs == d->workspace);      // (a caller that did not alternate its slab buffers: the fold must be done before this call writes)
        if (carry_on && !overlap && c->dw != d->dw) {
            hv_carry = HvFold{c->slabs, c->dw, c->numel, c->nslabs, c->accumulate, c->dbias ? c->bias_slabs : nullptr, c->dbias, c->Cout, c->dbias_accumulate};
        } else {
            rc = hv_wgrad_fold_now(c, stream);
            if (rc != HV_OK) return rc;
            c = nullptr;
        }
    } else c = nullptr;
    rc = wgrad_dispatch(d, stream);
    const bool taken = hv_carry_taken != 0;
    hv_carry.splits = 0;
    hv_carry_taken = 0;
    if (rc != HV_OK) return rc;
    if (c && !taken) return hv_wgrad_fold_now(c, stream);      // (behind this call's kernel: the two folds touch different dW)
    return HV_OK;
}
static int wgrad_dispatch(const hv_wgrad_desc* d, void* stream) {
    int rc;
    const long long nW = (long long)d->Cout * d->KH * d->KW * d->Cin;
    {   // single-output-channel VALU path (conv_narrow.hip), then the halo-tiled fast path (wgrad_halo.hip: 3x3 / 5x5, stride 1, fp16)
        int nslabs = 0;
        rc = hv_wgrad_thin(d, &nslabs, (hipStream_t)stream);   // thin operands on the transposed-LDS-read kernel (round 5)
        if (rc == HV_ERR_UNSUPPORTED) rc = hv_wgrad_halo(d, &nslabs, (hipStream_t)stream);   // (measured: the VALU hv_wgrad_narrow is slower than the padded MFMA tiles)
        if (rc == HV_ERR_UNSUPPORTED) rc = hv_wgrad_tr(d, &nslabs, (hipStream_t)stream);   // transposed-LDS-read form (fp16 storage)
        if (rc == HV_OK) {
            const float* bsl = d->dbias ? d->workspace + (long long)nslabs * nW : nullptr;
            return wgrad_fold_or_defer(d, d->workspace, nW, nslabs, bsl, (hipStream_t)stream);
        }
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    WgradPlan pl;
    wgrad_plan(d, &pl);
    const bool direct = pl.splits <= 1 && !d->accumulate && !d->dbias;
    if (!direct) {
        if (!d->workspace || d->workspace_bytes < (size_t)pl.splits * (nW + (d->dbias ? d->Cout : 0)) * sizeof(float)) return HV_ERR_WORKSPACE;
    }
    hv_path_note = 10;
    WgradK k;
    k.x = d->x; k.g = d->g; k.out = direct ? d->dw : d->workspace;
    k.B = d->B; k.Hl = d->H; k.Wl = d->W; k.in_shift = d->in_shift; k.Wp = d->W >> d->in_shift;
    k.img_stride = (d->H >> d->in_shift) * k.Wp * d->x_ld; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho; k.Wo = d->Wo; k.g_ld = d->g_ld; k.g_coff = d->g_coff; k.Cout = d->Cout;
    k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.dil = d->dil; k.J = d->KH * d->KW * d->Cin;
    k.M = d->B * d->Ho * d->Wo; k.chunk = pl.chunk; k.slab = nW;
    k.bias_out = d->dbias ? d->workspace + (long long)pl.splits * nW : nullptr;
    k.half = d->x_f16 ? 1 : 0;
    const size_t es = k.half ? 2 : 4;
    k.x_bytes = (unsigned)((size_t)d->B * (d->H >> d->in_shift) * k.Wp * d->x_ld * es);
    k.g_bytes = (unsigned)((size_t)k.M * d->g_ld * es);
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const bool fast = pow2(d->Wo) && pow2(d->Ho) && d->Wo >= 8;
    k.lw = k.lhw = 0;
    k.kt_q = pl.KT / d->Wo; k.kt_r = pl.KT % d->Wo;
    if (fast) { k.lw = __builtin_ctz(d->Wo); k.lhw = k.lw + __builtin_ctz(d->Ho); }
    hipStream_t s = (hipStream_t)stream;
    HV_TIMING_BEGIN(s);
    if (d->precision == HV_F32) rc = fast ? launch_wgrad<float, true>(k, pl, s) : launch_wgrad<float, false>(k, pl, s);
    else rc = fast ? launch_wgrad<_Float16, true>(k, pl, s) : launch_wgrad<_Float16, false>(k, pl, s);
    HV_TIMING_END(s);
    if (rc != HV_OK) return rc;
    if (!direct) return wgrad_fold_or_defer(d, d->workspace, nW, pl.splits, k.bias_out, s);
    if (d->pending) d->pending->nslabs = 0;      // written directly
    return HV_OK;
}
