// Weight gradient on fp16-stored tensors with TRANSPOSED LDS READS (gfx950 ds_read_b64_tr_b16): no register transposes, no conversions,
// the input patch staged once per pixel tile for all taps.
//
//   dW[co][(r,q)][ci] = sum_{n,oy,ox} g[n,oy,ox,co] * x[n, oy*S - pad + r, ox*S - pad + q, ci]
//
// As a GEMM the contraction index k is the PIXEL, while both operands are stored channel-contiguous (NHWC): the 16x16x32 MFMA wants, per
// lane, 8 consecutive k of one channel -- a transposed access.  The gather kernel (conv_igemm.hip: wgrad_kernel) transposes in registers
// (8 loads + 32 moves per staged item, 8.7 VALU instructions per MFMA on the PatchGAN layers) and re-reads x once per tap through L2.
// Here the tiles sit in LDS exactly as they sit in memory,
//     G[8x16 output pixels][BN channels],   X[(7*S+KS) x (15*S+KS) input pixels][BC channels]     (fp16, row = pixel)
// staged by plain 16-byte copies, and every MFMA operand is two ds_read_b64_tr_b16: a 16-lane group reads a block of 4 pixel rows x 16
// channels and each lane receives one channel's 4 pixels.  The instruction takes one ADDRESS PER LANE (block row = pixel), so the tap
// shift (r, q) of the input operand is nothing but another constant added to the lanes' patch addresses: x is staged once for all KS*KS
// taps.  k-step = 2 tile rows x 16 pixels: group g's first read takes pixels (ty, 4g .. 4g+3), its second (ty+1, 4g .. 4g+3); a
// 32-lane half therefore reads 8 consecutive pixel rows, which are conflict-free with a row stride = 32 B x odd (stride 1) or 16 B x odd
// (stride 2: every other patch pixel).
// The taps are dealt round-robin to the four waves (wave w owns taps w, w+4, ...): each wave keeps (BN/16)(BC/16)ceil(KS^2/4)
// accumulator tiles and reads every A fragment once per k-step.  One slab per workgroup, summed by wgrad_reduce_kernel (fixed order).
#include <stdlib.h>

#include "hv_common.h"

typedef __fp16 hv_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WTrK {
    const _Float16* x; const _Float16* g; float* slabs;
    int Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Ho, Wo, g_ld, g_coff, Cout, pad;
    int tiles_x, tiles_per_img, ntiles;
    long long slab;              // floats per slab = Cout * KS*KS * Cin
    unsigned x_bytes, g_bytes;   // buffer descriptor ranges
    float* bias_out;             // per-workgroup column sums of g (bias gradient), [gridDim.x][Cout], or NULL
};

__device__ __forceinline__ f16x4 tr_read(const _Float16* lds_addr) {
    typedef hv_fp16x4 __attribute__((address_space(3)))* lp;
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr)));
}

// row strides (bytes) of the LDS images: see the bank argument above
__host__ __device__ constexpr int wtr_stride(int ch, int st) {
    const int row = ch * 2;
    if (st == 1) return (row / 32) % 2 ? row : row + 32;                 // 32 B x odd
    return row + 16;                                                     // 16 B x odd (row is a multiple of 32)
}

template <int KS, int ST, int BN, int BC>
__global__ __launch_bounds__(256, 2) void wgrad_tr_kernel(const WTrK p) {
    constexpr int TH = 8, TW = 16, TAPS = KS * KS, SLOTS = (TAPS + 3) / 4;
    constexpr int PH = (TH - 1) * ST + KS, PW = (TW - 1) * ST + KS;
    constexpr int NT = BN / 16, CT = BC / 16;
    constexpr int SG = wtr_stride(BN, 1), SX = wtr_stride(BC, ST);       // bytes per pixel row
    constexpr int GI = TH * TW * (BN / 8), XI = PH * PW * (BC / 8);       // 16-byte staging items
    constexpr int GPT = (GI + 255) / 256, XPT = (XI + 255) / 256;
    static_assert(NT * CT * SLOTS <= 32, "accumulator budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Gs = smem;                               // [TH*TW][SG]
    char* Xs = smem + TH * TW * SG;                // [PH*PW][SX]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BN, ci0 = blockIdx.z * BC;
    const int grp = lane >> 4, sub = lane & 15, qr = sub >> 2, pc = sub & 3;

    f32x4 acc[SLOTS][NT][CT];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[s][n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr && blockIdx.z == 0 && wave == 0;      // wave-uniform
    f32x4 bacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS addresses of the transposed reads: block row = pixel (4*grp + qr) of a tile row, 4 channels at 8*pc bytes
    const _Float16* ga = reinterpret_cast<const _Float16*>(Gs + (4 * grp + qr) * SG + pc * 8);
    const _Float16* xa[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s, r = t / KS, q = t - r * KS;       // this wave's tap of slot s (t >= TAPS: unused)
        xa[s] = reinterpret_cast<const _Float16*>(Xs + ((4 * grp + qr) * ST + (t < TAPS ? r * PW + q : 0)) * SX + pc * 8);
    }

    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    u32x4 rg[GPT], rx[XPT];
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        const int n_img = tile / p.tiles_per_img, tr = tile - n_img * p.tiles_per_img;
        const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            const int c8 = e % (BN / 8), pix = e / (BN / 8), ty = pix / TW, tx = pix - ty * TW;
            const int oy = oy0 + ty, ox = ox0 + tx, co = co0 + c8 * 8;
            const bool ok = e < GI && oy < p.Ho && ox < p.Wo && co < p.Cout;
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc, ok ? (unsigned)(((n_img * p.Ho + oy) * p.Wo + ox) * p.g_ld + p.g_coff + co) * 2u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            const int c8 = e % (BC / 8), pix = e / (BC / 8), py = pix / PW, px = pix - py * PW;
            const int hi = oy0 * ST - p.pad + py, wi = ox0 * ST - p.pad + px, ci = ci0 + c8 * 8;
            const bool ok = e < XI && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl && ci < p.Cin;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(
                xsrc, ok ? (unsigned)(n_img * p.img_stride + ((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + p.x_coff + ci) * 2u : 0x80000000u, 0, 0);
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            if (e < GI) *reinterpret_cast<u32x4*>(Gs + (e / (BN / 8)) * SG + (e % (BN / 8)) * 16) = rg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XI) *reinterpret_cast<u32x4*>(Xs + (e / (BC / 8)) * SX + (e % (BC / 8)) * 16) = rx[i];
        }
    };
    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};

    if ((int)blockIdx.x < p.ntiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();   // previous tile's reads are done
        flush();
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles) prefetch(tile + gridDim.x);   // next tile's loads fly behind this tile's MFMAs
        // software pipeline over the (k-step, tap slot) pairs of the tile: the transposed LDS reads of the next pair are issued before the MFMAs of
        // the current one (with the reads right in front of their MFMAs a wave alternated ~130 cycles of LDS latency with 128 cycles of MFMAs; at
        // one workgroup per CU nothing else filled the gaps)
        auto lda = [&](int ks, f16x8 (&a)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f16x4 lo = tr_read(ga + ((2 * ks) * TW * SG + n * 32) / 2), hi = tr_read(ga + ((2 * ks + 1) * TW * SG + n * 32) / 2);
                a[n] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        auto ldb = [&](int ks, int sl, f16x8 (&bb)[CT]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const f16x4 lo = tr_read(xa[sl] + ((2 * ks) * ST * PW * SX + c * 32) / 2), hi = tr_read(xa[sl] + ((2 * ks + 1) * ST * PW * SX + c * 32) / 2);
                bb[c] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        f16x8 aq[2][NT], bq[2][CT];
        lda(0, aq[0]);
        ldb(0, 0, bq[0]);
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            if (do_bias) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[ks & 1][n], ones, bacc[n], 0, 0, 0);
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int idx = ks * SLOTS + sl;
                __builtin_amdgcn_sched_barrier(0);
                if (sl + 1 < SLOTS) ldb(ks, sl + 1, bq[(idx + 1) & 1]);
                else if (ks + 1 < TH / 2) { lda(ks + 1, aq[(ks + 1) & 1]); ldb(ks + 1, 0, bq[(idx + 1) & 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (wave + 4 * sl < TAPS) {     // wave-uniform (scalar) branch: MFMA ignores EXEC
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[sl][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[ks & 1][n], bq[idx & 1][c], acc[sl][n][c], 0, 0, 0);
                }
            }
        }
    }
    // ---- one slab per workgroup; D layout: row (= co) = (lane>>4)*4 + r, col (= ci) = lane & 15
    float* out = p.slabs + (long long)blockIdx.x * p.slab;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s;
        if (t >= TAPS) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int ci = ci0 + c * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Cout && ci < p.Cin) out[((long long)co * TAPS + t) * p.Cin + ci] = acc[s][n][c][r];
                }
            }
    }
    if (do_bias && (lane & 15) == 0) {     // every column of the ones-product holds the row sums: take column 0
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                if (co < p.Cout) p.bias_out[(long long)blockIdx.x * p.Cout + co] = bacc[n][r];
            }
    }
}

struct WTrPlan { int BN, BC, gx; size_t lds; };

static bool wgrad_tr_plan(const hv_wgrad_desc* d, WTrPlan* pl) {
    static const int enabled = getenv("HV_WGRAD_TR") ? atoi(getenv("HV_WGRAD_TR")) : 1;   // A/B knob
    if (!enabled || d->precision != HV_F16 || !d->x_f16 || !d->g_f16 || d->dil != 1 || d->KH != d->KW) return false;
    if (!((d->KH == 3 || d->KH == 4) && (d->stride == 1 || d->stride == 2))) return false;
    if (d->Ho != (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return false;
    // 16-byte staging items: 8-channel groups must be whole and aligned in both tensors
    if ((d->Cin & 7) || (d->Cout & 7) || (d->x_ld & 7) || (d->x_coff & 7) || (d->g_ld & 7) || (d->g_coff & 7)) return false;
    if (d->Cout < 32 || d->Cin < 16) return false;        // narrower layers: wgrad_halo_kernel / the gather kernel
    pl->BN = d->Cout >= 64 ? 64 : 32;
    pl->BC = d->Cin >= 32 ? 32 : 16;
    if (d->stride == 2 && pl->BN == 64) pl->BC = 16;      // the stride-2 patch is 3x larger: its prefetch registers leave room for 16 accumulator tiles
    const int PH = 7 * d->stride + d->KH, PW = 15 * d->stride + d->KW;
    pl->lds = (size_t)128 * wtr_stride(pl->BN, 1) + (size_t)PH * PW * wtr_stride(pl->BC, d->stride);
    const long long ntiles = (long long)d->B * hv_cdiv(d->Ho, 8) * hv_cdiv(d->Wo, 16);
    const long long pairs = (long long)hv_cdiv(d->Cout, pl->BN) * hv_cdiv(d->Cin, pl->BC);
    // workgroups wanted per launch (split over pixel chunks): every chunk writes a whole slab of dW, so a layer with a small dW tile count (the
    // generators' 64-channel layers: 2 tiles, 256 slabs of 147 KB = 38 MB for 17 MB of operands) is bound by its slab traffic, not by its MFMAs
    // (step-level A/B, round 3, same box: PatchGAN layers 512 -> 256 workgroups 9.43 -> 9.25 ms (their slabs are 8 MB each); 192: 9.23; the
    // generators' layers 512 / 256: no difference, 128: +0.2 ms)
    static const int want_big = getenv("HV_WGRAD_TR_WGS") ? atoi(getenv("HV_WGRAD_TR_WGS")) : 256;
    static const int want_small = getenv("HV_WGRAD_TR_WGS_SMALL") ? atoi(getenv("HV_WGRAD_TR_WGS_SMALL")) : 512;
    const int want = pairs <= 4 ? want_small : want_big;
    long long gx = (want + pairs - 1) / pairs;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    pl->gx = (int)gx;
    return true;
}

size_t hv_wgrad_tr_workspace_bytes(const hv_wgrad_desc* d) {
    WTrPlan pl;
    if (!wgrad_tr_plan(d, &pl)) return 0;
    return (size_t)pl.gx * ((size_t)d->Cout * d->KH * d->KW * d->Cin + (d->dbias ? d->Cout : 0)) * sizeof(float);
}

template <int KS, int ST, int BN, int BC>
static int launch_wtr(const WTrK& k, const WTrPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_tr_kernel<KS, ST, BN, BC>;
    static int lds_limit = 48 * 1024;
    if ((int)pl.lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        lds_limit = 150 * 1024;
    }
    dim3 grid(pl.gx, hv_cdiv(d->Cout, BN), hv_cdiv(d->Cin, BC));
    hv_path_note = 12;
    HV_KNAME("wgrad_tr_kernel<%d, %d, %d, %d>", KS, ST, BN, BC);
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, s, k);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// returns HV_ERR_UNSUPPORTED when the shape does not qualify; on success the slabs (*nslabs of them) are in d->workspace
int hv_wgrad_tr(const hv_wgrad_desc* d, int* nslabs, hipStream_t s) {
    WTrPlan pl;
    if (!wgrad_tr_plan(d, &pl)) return HV_ERR_UNSUPPORTED;
    const size_t need = hv_wgrad_tr_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need) return HV_ERR_WORKSPACE;
    WTrK k;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.g = reinterpret_cast<const _Float16*>(d->g); k.slabs = d->workspace;
    k.Hl = d->H; k.Wl = d->W; k.in_shift = d->in_shift; k.Wp = d->W >> d->in_shift;
    k.img_stride = (d->H >> d->in_shift) * k.Wp * d->x_ld; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho; k.Wo = d->Wo; k.g_ld = d->g_ld; k.g_coff = d->g_coff; k.Cout = d->Cout; k.pad = d->pad;
    k.tiles_x = hv_cdiv(d->Wo, 16); k.tiles_per_img = k.tiles_x * hv_cdiv(d->Ho, 8); k.ntiles = k.tiles_per_img * d->B;
    k.slab = (long long)d->Cout * d->KH * d->KW * d->Cin;
    k.bias_out = d->dbias ? d->workspace + (long long)pl.gx * k.slab : nullptr;
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * sizeof(_Float16));
    k.g_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->g_ld * sizeof(_Float16));
    *nslabs = pl.gx;
#define WTR(KS_, ST_)                                                                                     \
    do {                                                                                                  \
        if (pl.BN == 64 && pl.BC == 32) return launch_wtr<KS_, ST_, 64, 32>(k, pl, d, s);                 \
        if (pl.BN == 64 && pl.BC == 16) return launch_wtr<KS_, ST_, 64, 16>(k, pl, d, s);                 \
        if (pl.BN == 32 && pl.BC == 32) return launch_wtr<KS_, ST_, 32, 32>(k, pl, d, s);                 \
        return launch_wtr<KS_, ST_, 32, 16>(k, pl, d, s);                                                 \
    } while (0)
    if (d->KH == 3 && d->stride == 1) WTR(3, 1);
    if (d->KH == 3) WTR(3, 2);
    if (d->stride == 1) WTR(4, 1);
    WTR(4, 2);
#undef WTR
}
