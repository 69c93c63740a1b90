// Halo-tiled implicit-GEMM convolution, third form: the whole filter bank of the workgroup lives in LDS.
//
// conv_halo2_kernel fetches a wave's filter rows from L2 into MFMA operand registers, tap by tap, for every 64-pixel tile: a 64 -> 64 channel
// 3x3 layer re-reads its 73 KB of filters once per 64 output pixels (75 MB of L2 -> CU traffic for 16.8 MB of activations at 64x64, bs 16)
// and every workgroup is one chain of dependent round trips (chunk 0 patch -> barrier -> tap ring refills -> chunk 1 ...); its prologue and
// epilogue cost more than its MFMA loop.  The generator's stride-1 3x3 layers have at most 64 x 64 (x 9) filter elements
// per 64-channel output block, so here
//   * ONE workgroup of 8 waves owns a 16 x 16-pixel output tile (4x the pixels per filter fetch) and all CO output channels of its block;
//   * its filters (MFMA-fragment order, hv_weight_tile_f16: one contiguous piece per workgroup) and its WHOLE input patch (18 x 18 pixels, every
//     input channel) are requested up front -- every load of the workgroup is in flight before the first wait: one memory round trip, one barrier;
//   * both MFMA operands come from LDS: a wave computes 2 pixel rows x CO channels (A fragments are 1-KB linear pieces, B fragments are read from
//     96-byte (48-byte at 16-channel planes) pixel rows: conflict-free for ds_read_b128 / b64 lane groups);
//   * the fp16 output tile leaves through LDS as 16-byte pieces (whole channel rows per pixel), like conv_halo2_kernel's epilogue, with the same
//     act' multiplier / accumulate forms, so the kernel serves the layers' data gradients (filters = the transposed table) as well.
// Same argument block (HaloK) and tap table as the other halo kernels; dilation d runs as d*d residue sub-grids, a fused nearest x2 upsample of the
// input is address arithmetic (in_shift), exactly as in conv_halo2_kernel.
#include <stdlib.h>

#include "conv_halo.h"

// SGE < 16: the tile is (16 / SGE)^2 whole residue sub-grids of a dilated layer, each SGE x SGE pixels (see conv_lfd_kernel)
// ST = 2: stride-2 layers (conv_lf2_kernel): the patch is (2 TH + 1) x 33 input pixels, tile pixel (r, c) reads cell (2 r + dy, 2 c + dx)
template <int CIN, int CO, int TH, int SGE = 16, int ST = 1>
struct LfCfg {
    static constexpr int T = (CIN % 32 == 0) ? 32 : 16;      // channels per plane = per MFMA step (the tiled filter table's fragment width)
    static constexpr int NPL = CIN / T;                      // channel planes of the patch
    // halfs per patch pixel row: 96 B / 48 B; stride 2 with 32-channel planes: 80 B -- the lanes of a fragment read are TWO cells apart, and 160 B steps walk
    // eight distinct 16-byte bank groups where 192 B steps walk four
    static constexpr int LDP = (ST == 2 && T == 32) ? 40 : T + (T == 32 ? 16 : 8);
    static constexpr int TW = 16, NW = 8, NTHR = NW * 64;
    static constexpr int MT = TH / NW;                       // pixel rows (16-pixel MFMA groups) per wave
    static constexpr int NT = CO / 16;
    static constexpr int PH = ST == 2 ? 2 * TH + 1 : TH + TH / SGE + 1;       // one ring cell between / around the sub-grids (SGE = 16: the plain 18 x 18 patch)
    static constexpr int PW = ST == 2 ? 2 * TW + 1 : TW + TW / SGE + 1;
    static_assert(ST == 1 || SGE == 16, "stride 2: plain tiles");
    static constexpr int PLANE = PH * PW * LDP;              // halfs
    static constexpr int WHALFS = CO * 9 * CIN;              // filter halfs of the workgroup
    static constexpr int LDO = CO + 8;                       // output staging: halfs per pixel row
    static constexpr int PATCH_HALFS = NPL * PLANE;
    static constexpr int OST_HALFS = TH * TW * LDO;
    static constexpr int TAIL_HALFS = PATCH_HALFS > OST_HALFS ? PATCH_HALFS : OST_HALFS;
    static constexpr int X1_FLOATS = PH * PW + CO * 9;       // the extra input channel's patch + its filters (hv_conv_desc.x1)
    static constexpr size_t LDS_BYTES = (size_t)(WHALFS + TAIL_HALFS) * 2 + (size_t)X1_FLOATS * 4;
    static constexpr int WITEMS = (WHALFS * 2 / 16 + NTHR - 1) / NTHR;           // 16-byte filter items per thread
    static constexpr int PITEMS = (PH * PW * (CIN / 8) + NTHR - 1) / NTHR;       // 16-byte patch items per thread
};

template <int CIN, int CO, int TH, bool X1, int SGE, int ST = 1>
__device__ __forceinline__ void conv_lf_body(const HaloK& p) {
    typedef LfCfg<CIN, CO, TH, SGE, ST> G;
    static_assert(ST == 1 || !X1, "stride 2: no extra-channel form");
    constexpr bool PK = SGE < 16;                 // packed residue sub-grids (conv_lfd_kernel)
    constexpr int NSG = 16 / SGE;
    static_assert(!(PK && X1) && (!PK || TH == 16), "packed sub-grids: plain 16 x 16 tiles");
    constexpr int T = G::T, NPL = G::NPL, LDP = G::LDP, MT = G::MT, NT = G::NT, PW = G::PW, PH = G::PH, TW = G::TW;
    typedef typename HFrag<T>::V V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* wl = reinterpret_cast<_Float16*>(smem);                 // [CO/16][9][NPL][16 * T]  (the tiled table's own order)
    _Float16* patch = wl + G::WHALFS;                                 // [NPL][PH * PW][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef LF_STAMPS     // diagnostic build only (tools/lf_stamps.sh): phase times of one workgroup, written over the first bytes of y
    unsigned long long st[8];
#define LF_STAMP(i) st[i] = __builtin_amdgcn_s_memrealtime()
#else
#define LF_STAMP(i)
#endif
    LF_STAMP(0);
    const HaloCls& C = p.cls[0];
    int t = (int)blockIdx.x;
    int ry = 0, rx = 0;
    if (!PK && p.dil > 1) {
        const int per = C.tiles * p.B, rid = t / per;
        t -= rid * per;
        ry = rid / p.dil; rx = rid - ry * p.dil;
    }
    const int n_img = t / C.tiles;
    t -= n_img * C.tiles;
    const int tile_y = t / C.tiles_x, tile_x = t - tile_y * C.tiles_x;
    if (PK) { ry = tile_y * NSG; rx = tile_x * NSG; }      // the tile = the NSG x NSG residue classes (ry + gy, rx + gx), whole
    const int i0 = PK ? 0 : tile_y * TH, j0 = PK ? 0 : tile_x * TW;
    const int n_base = blockIdx.y * CO;
    const int h0 = i0 * ST + p.boff + C.dh_min, w0 = j0 * ST + p.boff + C.dw_min;

    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);

    // ---- every load of the workgroup, issued before the first wait; plane by plane (a plane = T input channels of the filters and of the patch),
    // so that the MFMAs of plane 0 run while plane 1 is still landing
    constexpr int FB = 16 * T * 2, IPF = FB / 16;                                 // bytes / 16-byte items of one filter fragment
    constexpr int WPI = ((CO / 16) * 9 * IPF + G::NTHR - 1) / G::NTHR;            // filter items per plane and thread
    constexpr int PPI = (PH * PW * (T / 8) + G::NTHR - 1) / G::NTHR;              // patch items per plane and thread
    u32x4 wreg[NPL][WPI], preg[NPL][PPI];
    int plo[PPI];
    unsigned pvo[PPI];
    const unsigned wbase = (unsigned)(n_base / 16) * (unsigned)(16 * 9 * CIN * 2);
    const unsigned xbase = (unsigned)(n_img * p.img_stride + p.x_coff) * 2u;
#pragma unroll
    for (int i = 0; i < PPI; ++i) {
        const int e = tid + i * G::NTHR;
        const int c8 = e % (T / 8), pix = e / (T / 8);
        const int py = pix / PW, px = pix - py * PW;
        const int hi = h0 + py, wi = w0 + px;
        const bool in = pix < PH * PW;
        plo[i] = in ? pix * LDP + c8 * 8 : -1;
        if (PK) {
            // cell (py, px): ring / separator cells (every (SGE + 1)-th) are zeros -- a sub-grid IS its whole residue class, so its neighbours at local
            // -1 / SGE lie outside the image; the others are pixel (wy - 1, wx - 1) of sub-grid (by, bx)
            const int by = py / (SGE + 1), wy = py - by * (SGE + 1), bx = px / (SGE + 1), wx = px - bx * (SGE + 1);
            pvo[i] = (in && wy != 0 && wx != 0)
                         ? xbase + (unsigned)(((ry + by + (wy - 1) * p.dil) * p.Wp + rx + bx + (wx - 1) * p.dil) * p.x_ld + c8 * 8) * 2u : HV_OOB;
        } else
        pvo[i] = (in && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
                     ? xbase + (unsigned)((((ry + hi * p.dil) >> p.in_shift) * p.Wp + ((rx + wi * p.dil) >> p.in_shift)) * p.x_ld + c8 * 8) * 2u : HV_OOB;
    }
#pragma unroll
    for (int kc = 0; kc < NPL; ++kc) {
#pragma unroll
        for (int i = 0; i < WPI; ++i) {
            const int e = tid + i * G::NTHR;
            wreg[kc][i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, e < (CO / 16) * 9 * IPF ? wbase + (unsigned)(((e / IPF) * NPL + kc) * FB + (e % IPF) * 16) : HV_OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PPI; ++i) preg[kc][i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, pvo[i], kc * T * 2, 0);
    }
    // the extra input channel (scalar branch): its (TH + 2) x 18 patch and the workgroup's CO x 9 filter values, as floats behind the tail region
    float* x1p = reinterpret_cast<float*>(smem + (size_t)(G::WHALFS + G::TAIL_HALFS) * 2);
    float* w1s = x1p + PH * PW;
    // (requested now -- one patch value and up to two filter values per thread -- and parked in LDS behind the MFMA loop: as LDS stores up here the address
    // arithmetic and the stores sat on top of the filter / patch prefetch registers and spilled)
    float x1v[2] = {0.f, 0.f}, w1v[2] = {0.f, 0.f};
    static_assert(!X1 || (PH * PW <= 2 * G::NTHR && CO * 9 <= 2 * G::NTHR), "two patch values, two filter values per thread");
    if constexpr (X1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * G::NTHR;
            if (e < PH * PW) {
                const int py = e / PW, px = e - py * PW, hi = h0 + py, wi = w0 + px;
                if ((unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl) x1v[u] = hv_ld1(p.x1, ((long long)(n_img * p.Hl + hi) * p.Wl + wi) * p.x1_ld + p.x1_coff, p.x1_half);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * G::NTHR, co = n_base + e / 9, tp = e % 9;
            if (e < CO * 9 && co < p.Cout) w1v[u] = p.w1[(long long)co * p.w1_row + tp * p.w1_tap];
        }
    }
    // scalar tap table
    int toff[9], widx[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const uint32_t e = C.taps[q];
        toff[q] = ((int)(e & 0xff) * PW + (int)((e >> 8) & 0xff)) * LDP;
        widx[q] = (int)(e >> 16);
    }
    // bias (branch-free: lanes beyond Cout read zeros through the range check)
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
    f32x4 bias_r[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
        bias_r[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, (unsigned)(n_base + n * 16 + (lane >> 4) * 4) * 4u, 0, 0));
    LF_STAMP(1);
    // ---- MFMA loop: wave = MT pixel rows x CO channels
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int poffm[MT];      // tile row r, column c -> patch cell (r + r / SGE, c + c / SGE) (+ the tap): SGE = 16 is the plain r * PW + c
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = wave * MT + m, c = lane & 15;
        poffm[m] = ST == 2 ? (2 * r * PW + 2 * c) * LDP + (lane >> 4) * (T / 4) : ((r + r / SGE) * PW + c + c / SGE) * LDP + (lane >> 4) * (T / 4);
    }
    const int aoff = lane * (T / 4);
    // per plane: its filters and patch rows -> LDS, one barrier, then a software pipeline over its 9 taps (the fragments of tap q + 1 are requested
    // before the MFMAs of tap q)
    V a[2][NT], b[2][MT];
#pragma unroll
    for (int kc = 0; kc < NPL; ++kc) {
#pragma unroll
        for (int i = 0; i < WPI; ++i) {
            const int e = tid + i * G::NTHR;
            if (e < (CO / 16) * 9 * IPF) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(wl) + ((e / IPF) * NPL + kc) * FB + (e % IPF) * 16) = wreg[kc][i];
        }
#pragma unroll
        for (int i = 0; i < PPI; ++i)
            if (plo[i] >= 0) *reinterpret_cast<u32x4*>(patch + kc * G::PLANE + plo[i]) = preg[kc][i];
        __syncthreads();
        if (kc == 0) { LF_STAMP(2); }
        auto frags = [&](int q, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < NT; ++n) a[buf][n] = *reinterpret_cast<const V*>(wl + ((n * 9 + widx[q]) * NPL + kc) * (16 * T) + aoff);
#pragma unroll
            for (int m = 0; m < MT; ++m) b[buf][m] = *reinterpret_cast<const V*>(patch + kc * G::PLANE + poffm[m] + toff[q]);
        };
        frags(0, 0);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (q + 1 < 9) frags(q + 1, (q + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler would sink the reads to their use)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n][m] = HFrag<T>::mma(a[q & 1][n], b[q & 1][m], acc[n][m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    LF_STAMP(3);
    constexpr int PIECES = CO / 8;
    constexpr int OITEMS = TH * TW * PIECES / G::NTHR;
    static_assert(TH * TW * PIECES % G::NTHR == 0, "output pieces per thread");
    const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.mul_src), 0, p.mul_src ? 0x7ffffff0u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.accumulate ? 0x7ffffff0u : 0u, 0x00020000);
    // this thread's pieces of the act' multiplier and of the gradient buffer it adds to (data-gradient forms): requested now, they land behind the
    // activation arithmetic of the staging pass
    u32x4 mreg[OITEMS], yreg[OITEMS];
    long long ooff[OITEMS];         // element offset of the piece in y (-1: outside the image / beyond Cout)
    // pooled form (hv_conv_desc.pool2): the tile leaves as (TH / 2) x (TW / 2) pixels, each the sum of its four fp16-rounded values; a thread's item
    // k < PITEMS is then (pooled pixel, piece) and y / mul_src are the pooled tensors
    constexpr int PPIX = (TH / 2) * (TW / 2), PITEMS = (PPIX * PIECES + G::NTHR - 1) / G::NTHR;
    const bool pooled = p.pool2 != 0;      // scalar
    if (!pooled) {
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            const int it = tid + k * G::NTHR;
            const int q = it / PIECES, pc = it - q * PIECES;
            const int i = i0 + (q >> 4), j = j0 + (q & 15), ch = n_base + pc * 8;
            const bool ok = (PK || (i < C.Hc && j < C.Wc)) && ch < p.Cout;
            // packed: tile pixel (r, c) = pixel (r % SGE, c % SGE) of sub-grid (r / SGE, c / SGE)
            const int ho = PK ? C.ph + ry + i / SGE + (i % SGE) * p.ostep : C.ph + ry + i * p.ostep;
            const int wo = PK ? C.pw + rx + j / SGE + (j % SGE) * p.ostep : C.pw + rx + j * p.ostep;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            ooff[k] = ok ? opix * p.y_ld + p.y_coff + ch : -1;
            mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(msrc, ok ? (unsigned)((opix * p.mul_ld + p.mul_coff + ch) * 2) : HV_OOB, 0, 0);
            yreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, ok ? (unsigned)(ooff[k] * 2) : HV_OOB, 0, 0);
        }
    } else {
#pragma unroll
        for (int k = PITEMS; k < OITEMS; ++k) { ooff[k] = -1; mreg[k] = yreg[k] = (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int it = tid + k * G::NTHR;
            const int q = it / PIECES, pc = it - q * PIECES;
            const int il = (i0 >> 1) + q / (TW / 2), jl = (j0 >> 1) + q % (TW / 2), ch = n_base + pc * 8;
            const bool ok = q < PPIX && il < p.Ho && jl < p.Wo && ch < p.Cout;
            const long long opix = (long long)(n_img * p.Ho + il) * p.Wo + jl;
            ooff[k] = ok ? opix * p.y_ld + p.y_coff + ch : -1;
            mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(msrc, ok ? (unsigned)((opix * p.mul_ld + p.mul_coff + ch) * 2) : HV_OOB, 0, 0);
            yreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, ok ? (unsigned)(ooff[k] * 2) : HV_OOB, 0, 0);
        }
    }
    if constexpr (X1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (tid + u * G::NTHR < PH * PW) x1p[tid + u * G::NTHR] = x1v[u];
            if (tid + u * G::NTHR < CO * 9) w1s[tid + u * G::NTHR] = w1v[u];
        }
    }
    __syncthreads();          // every wave is done with the patch: its room becomes the output staging tile

    // ---- epilogue: (alpha, +bias, activation) -> fp16 tile in LDS -> 16-byte pieces (act' multiplier / accumulate forms applied there).
    // The activation is chosen ONCE (scalar switch around the whole tile) and its body is branch-free: the shared per-element epilogue code of the
    // other kernels (a scalar switch, range checks and a pointer path per value) is ~25 instructions x 32 values per lane here.
    _Float16* ot = patch;
    constexpr int LDO = G::LDO;
    if constexpr (X1) {      // acc += the extra channel's 9 taps (before alpha / bias / activation): 9 patch values per pixel row, 9 filter values per channel
        float cv[MT][9];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int q = (wave * MT + m) * 16 + (lane & 15);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const uint32_t e = C.taps[t9];
                cv[m][t9] = x1p[((q >> 4) + (int)(e & 0xff)) * PW + (q & 15) + (int)((e >> 8) & 0xff)];
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            // the 4 x 9 filter values of this lane's four channels of block n in one batch of LDS reads (one (channel, tap) at a time was a chain of
            // LDS latencies: +20 us on the 256 x 256 layer), then the sums -- each "used" by an empty asm so that the next block's reads are not
            // fetched ahead of them (all CO x 9 values up front: +108 registers, spills)
            float w9[4][9];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) w9[r][t9] = w1s[(n * 16 + (lane >> 4) * 4 + r) * 9 + widx[t9]];      // (an LDS address, not a register index)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float ex = 0.f;
#pragma unroll
                    for (int t9 = 0; t9 < 9; ++t9) ex += cv[m][t9] * w9[r][t9];
                    float v = acc[n][m][r] + ex;
                    asm volatile("" : "+v"(v) : : "memory");
                    acc[n][m][r] = v;
                }
        }
    }
    auto stage = [&](auto actf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int q = (wave * MT + m) * 16 + (lane & 15);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f16x4v h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (_Float16)actf(acc[n][m][r] * p.alpha + bias_r[n][r]);
                *reinterpret_cast<f16x4v*>(ot + q * LDO + n * 16 + (lane >> 4) * 4) = h;
            }
        }
    };
    switch (p.act) {
        case HV_ACT_ELU:        // hv_act_fast's ELU as selects
            stage([](float v) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504f) - 1.f, sm = v + 0.5f * v * v; return v > 0.f ? v : (v > -0.00390625f ? sm : e); });
            break;
        case HV_ACT_RELU: stage([](float v) { return v > 0.f ? v : 0.f; }); break;
        case HV_ACT_LRELU: stage([](float v) { return v > 0.f ? v : 0.2f * v; }); break;
        case HV_ACT_SIGMOID: stage([](float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504f)); }); break;
        case HV_ACT_CLAMP: stage([](float v) { return fminf(fmaxf(v, -1.f), 1.f); }); break;
        default: stage([](float v) { return v; }); break;
    }
    __syncthreads();
    LF_STAMP(4);
    _Float16* yb = reinterpret_cast<_Float16*>(p.y);
    u32x4 o[OITEMS];
    if (!pooled) {
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            const int it = tid + k * G::NTHR;
            const int q = it / PIECES, pc = it - q * PIECES;
            o[k] = *reinterpret_cast<const u32x4*>(ot + q * LDO + pc * 8);
        }
    } else {
#pragma unroll
        for (int k = PITEMS; k < OITEMS; ++k) o[k] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < PITEMS; ++k) {
            const int it = tid + k * G::NTHR;
            const int q = it / PIECES, pc = it - q * PIECES;
            const int r2 = 2 * (q / (TW / 2)), c2 = 2 * (q % (TW / 2));
            float sum[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[e] = 0.f;
            if (q < PPIX) {
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int rr = r2 + (dd >> 1), cc = c2 + (dd & 1);
                    if (i0 + rr < C.Hc && j0 + cc < C.Wc) {      // tile pixels beyond the convolution's grid are not part of the sum
                        const f16x8 v8 = *reinterpret_cast<const f16x8*>(ot + (rr * TW + cc) * LDO + pc * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum[e] += (float)v8[e];
                    }
                }
            }
            f16x8 h8;
#pragma unroll
            for (int e = 0; e < 8; ++e) h8[e] = (_Float16)sum[e];
            o[k] = __builtin_bit_cast(u32x4, h8);
        }
    }
    if (p.mul_src) {        // v * act'(m), m = the producer's fp16 output at the same pixel / channels (scalar switch around the tile, branch-free bodies)
        auto mulf = [&](auto gradf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < OITEMS; ++k) {
                const f16x8 m8 = __builtin_bit_cast(f16x8, mreg[k]);
                f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] * gradf((float)m8[e]));
                o[k] = __builtin_bit_cast(u32x4, v8);
            }
        };
        switch (p.mul_act) {
            case HV_ACT_ELU: mulf([](float y) { return y > 0.f ? 1.f : y + 1.f; }); break;
            case HV_ACT_RELU: mulf([](float y) { return y > 0.f ? 1.f : 0.f; }); break;
            case HV_ACT_LRELU: mulf([](float y) { return y > 0.f ? 1.f : 0.2f; }); break;
            case HV_ACT_SIGMOID: mulf([](float y) { return y * (1.f - y); }); break;
            case HV_ACT_CLAMP: mulf([](float y) { return (y > -1.f && y < 1.f) ? 1.f : 0.f; }); break;
            default: break;
        }
    }
    if (p.accumulate) {
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            const f16x8 y8 = __builtin_bit_cast(f16x8, yreg[k]);
            f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[e]);
            o[k] = __builtin_bit_cast(u32x4, v8);
        }
    }
#pragma unroll
    for (int k = 0; k < OITEMS; ++k)
        if (ooff[k] >= 0) *reinterpret_cast<u32x4*>(yb + ooff[k]) = o[k];
#ifdef LF_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    LF_STAMP(5);
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.y);
        for (int i = 0; i < 6; ++i) d[i] = st[i];
    }
#endif
}

template <int CIN, int CO, int TH, int WPS, bool X1 = false>       // X1: the hv_conv_desc.x1 form (its own instantiations: the extra code must not cost the others registers)
__global__ __launch_bounds__(512, WPS) void conv_lf_kernel(const HaloK p) {
    conv_lf_body<CIN, CO, TH, X1, 16>(p);
}

// Dilated layers whose residue sub-grids are SMALLER than a tile (64 x 64 maps: d = 8 -> 8 x 8, d = 16 -> 4 x 4 pixels): the residue scheme above would leave a
// 16 x 16 tile 25 % / 6 % full, so conv_halo2_kernel (d = 8, 22 us) and the gather kernel (d = 16, 30 us) served them against 10.5 us for the undilated layer.
// Here a tile is (16 / SGE)^2 WHOLE sub-grids -- the residue classes (ry .. ry + 16 / SGE, rx .. rx + 16 / SGE) -- side by side in the LDS patch with one ring of
// zeros around each (a sub-grid that is its whole residue class has no neighbours inside the image), so every pixel of the tile is an output, every input pixel is
// read once (no halo), and the MFMA loop is the plain layer's with another row / column map.  Same arithmetic per output as the residue scheme.
template <int CIN, int CO, int WPS, int SGE>
__global__ __launch_bounds__(512, WPS) void conv_lfd_kernel(const HaloK p) {
    conv_lf_body<CIN, CO, 16, false, SGE>(p);
}

// 3x3 stride-2 layers (the generators' down-sampling convolutions, forward): the same workgroup over a 33 x 33-pixel patch.  They ran in conv_halo_kernel
// (filters re-fetched per 8 x 16 tile, 1.2 - 2 TB/s of their bytes).
template <int CIN, int CO, int WPS>
__global__ __launch_bounds__(512, WPS) void conv_lf2_kernel(const HaloK p) {
    conv_lf_body<CIN, CO, 16, false, 16, 2>(p);
}

typedef __attribute__((address_space(3))) void* lf_lds_ptr;
// LDS-DMA: 16 bytes per lane, global -> LDS at dst + 16 * lane (device-only body: see conv_g4.hip)
__device__ __forceinline__ void lf_dma16(__amdgpu_buffer_rsrc_t r, lf_lds_ptr dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// The same convolution for grids of more than one round of workgroups (the 128 x 128 and 256 x 256 layers: 4 .. 16 tiles per CU).  conv_lf_kernel pays its
// prologue, its filter fetch (73 KB of the 136 KB a 64 -> 64 workgroup reads) and a launch-to-launch bubble once per TILE; here a workgroup stays, keeps
// its filters in LDS and walks tiles blockIdx.x, + gridDim.x, ...: the next tile's patch is requested (into the registers the current patch has just left
// for LDS) before the MFMAs of the current one, so it lands behind the MFMA loop and the epilogue; the filters go global -> LDS by LDS-DMA (their LDS image is the
// tiled table's own byte order: no staging registers).  32-channel planes only (CIN 32 / 64).  Same arithmetic, same order: bit-identical output.
template <int CIN, int CO, int TH, int WPS, bool X1 = false>
__global__ __launch_bounds__(512, WPS) void conv_lfp_kernel(const HaloK p, const int total) {
    typedef LfCfg<CIN, CO, TH> G;
    constexpr int T = G::T, NPL = G::NPL, LDP = G::LDP, MT = G::MT, NT = G::NT, PW = G::PW, PH = G::PH, TW = G::TW;
    typedef typename HFrag<T>::V V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* wl = reinterpret_cast<_Float16*>(smem);
    _Float16* patch = wl + G::WHALFS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HaloCls& C = p.cls[0];
    const int n_base = blockIdx.y * CO;
    struct Tile { int ry, rx, n_img, i0, j0; };
    auto decode = [&](int t) __attribute__((always_inline)) {
        Tile tl; tl.ry = 0; tl.rx = 0;
        if (p.dil > 1) {
            const int per = C.tiles * p.B, rid = t / per;
            t -= rid * per;
            tl.ry = rid / p.dil; tl.rx = rid - tl.ry * p.dil;
        }
        tl.n_img = t / C.tiles;
        t -= tl.n_img * C.tiles;
        const int tile_y = t / C.tiles_x, tile_x = t - tile_y * C.tiles_x;
        tl.i0 = tile_y * TH; tl.j0 = tile_x * TW;
        return tl;
    };
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    constexpr int FB = 16 * T * 2, IPF = FB / 16;
    constexpr int PPI = (PH * PW * (T / 8) + G::NTHR - 1) / G::NTHR;
    static_assert(T == 32, "1-KB filter fragments");
    constexpr int NFRAG = (CO / 16) * 9;
    u32x4 preg[NPL][PPI];
    int plo[PPI], ppy[PPI], ppx[PPI];
#pragma unroll
    for (int i = 0; i < PPI; ++i) {
        const int e = tid + i * G::NTHR;
        const int c8 = e % (T / 8), pix = e / (T / 8);
        ppy[i] = pix / PW; ppx[i] = pix - ppy[i] * PW;
        plo[i] = pix < PH * PW ? pix * LDP + c8 * 8 : -1;
    }
    // byte offsets of this thread's patch items for a tile (HV_OOB: outside the image -> zeros)
    auto patch_offsets = [&](const Tile& tl, unsigned (&pvo)[PPI]) __attribute__((always_inline)) {
        const int h0 = tl.i0 + p.boff + C.dh_min, w0 = tl.j0 + p.boff + C.dw_min;
        const unsigned xbase = (unsigned)(tl.n_img * p.img_stride + p.x_coff) * 2u;
#pragma unroll
        for (int i = 0; i < PPI; ++i) {
            const int hi = h0 + ppy[i], wi = w0 + ppx[i];
            const int c8 = (tid + i * G::NTHR) % (T / 8);
            pvo[i] = (plo[i] >= 0 && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
                         ? xbase + (unsigned)((((tl.ry + hi * p.dil) >> p.in_shift) * p.Wp + ((tl.rx + wi * p.dil) >> p.in_shift)) * p.x_ld + c8 * 8) * 2u : HV_OOB;
        }
    };
    float* x1p = reinterpret_cast<float*>(smem + (size_t)(G::WHALFS + G::TAIL_HALFS) * 2);
    float* w1s = x1p + PH * PW;
    float x1v[2] = {0.f, 0.f}, w1v[2] = {0.f, 0.f};
    auto x1_loads = [&](const Tile& tl) __attribute__((always_inline)) {
        const int h0 = tl.i0 + p.boff + C.dh_min, w0 = tl.j0 + p.boff + C.dw_min;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * G::NTHR;
            x1v[u] = 0.f;
            if (e < PH * PW) {
                const int py = e / PW, px = e - py * PW, hi = h0 + py, wi = w0 + px;
                if ((unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl) x1v[u] = hv_ld1(p.x1, ((long long)(tl.n_img * p.Hl + hi) * p.Wl + wi) * p.x1_ld + p.x1_coff, p.x1_half);
            }
        }
    };
    int tcur = (int)blockIdx.x;
    Tile cur = decode(tcur);
    {
        unsigned pvo[PPI];
        patch_offsets(cur, pvo);
        const unsigned wbase = (unsigned)(n_base / 16) * (unsigned)(16 * 9 * CIN * 2);
#pragma unroll
        for (int kc = 0; kc < NPL; ++kc) {
            // plane kc of the filters (fragment f of the plane = 1 KB = one wave's DMA), then of the first patch: loads return in order, so the wait for the
            // patch plane in front of its LDS stores covers the filter plane as well
#pragma unroll
            for (int i = 0; i < (NFRAG + G::NW - 1) / G::NW; ++i) {
                const int f = wave + G::NW * i;
                if (f < NFRAG) lf_dma16(wsrc, (lf_lds_ptr)(reinterpret_cast<char*>(wl) + (f * NPL + kc) * FB), (unsigned)lane * 16u, (int)wbase + (f * NPL + kc) * FB);
            }
#pragma unroll
            for (int i = 0; i < PPI; ++i) preg[kc][i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, pvo[i], kc * T * 2, 0);
        }
    }
    static_assert(!X1 || (PH * PW <= 2 * G::NTHR && CO * 9 <= 2 * G::NTHR), "two patch values, two filter values per thread");
    if constexpr (X1) {
        x1_loads(cur);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * G::NTHR, co = n_base + e / 9, tp = e % 9;
            if (e < CO * 9 && co < p.Cout) w1v[u] = p.w1[(long long)co * p.w1_row + tp * p.w1_tap];
            if (e < CO * 9) w1s[e] = w1v[u];      // (tile-invariant: parked once; x1p changes with the tile)
        }
    }
    int toff[9], widx[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const uint32_t e = C.taps[q];
        toff[q] = ((int)(e & 0xff) * PW + (int)((e >> 8) & 0xff)) * LDP;
        widx[q] = (int)(e >> 16);
    }
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
    f32x4 bias_r[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
        bias_r[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, (unsigned)(n_base + n * 16 + (lane >> 4) * 4) * 4u, 0, 0));
    const int poff = (wave * MT * PW + (lane & 15)) * LDP + (lane >> 4) * (T / 4);
    const int aoff = lane * (T / 4);
    constexpr int PIECES = CO / 8;
    constexpr int OITEMS = TH * TW * PIECES / G::NTHR;
    static_assert(TH * TW * PIECES % G::NTHR == 0, "output pieces per thread");
    constexpr int PPIX = (TH / 2) * (TW / 2), PITEMS = (PPIX * PIECES + G::NTHR - 1) / G::NTHR;
    constexpr int LDO = G::LDO;
    const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.mul_src), 0, p.mul_src ? 0x7ffffff0u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.accumulate ? 0x7ffffff0u : 0u, 0x00020000);
    constexpr bool EPI = !X1;              // act' multiplier / accumulate forms (data gradients): never together with the extra input channel
    const bool pooled = p.pool2 != 0;      // scalar
    _Float16* yb = reinterpret_cast<_Float16*>(p.y);
    _Float16* ot = patch;
    bool more;

    auto pass = [&]() __attribute__((always_inline)) {      // one tile
        const int tn = tcur + (int)gridDim.x;
        more = tn < total;
        const Tile nxt = decode(more ? tn : tcur);
        unsigned pvn[PPI];
        patch_offsets(nxt, pvn);
        if (!more) {
#pragma unroll
            for (int i = 0; i < PPI; ++i) pvn[i] = HV_OOB;
        }
        f32x4 acc[NT][MT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        V a[2][NT], b[2][MT];
#pragma unroll
        for (int kc = 0; kc < NPL; ++kc) {
#pragma unroll
            for (int i = 0; i < PPI; ++i)
                if (plo[i] >= 0) *reinterpret_cast<u32x4*>(patch + kc * G::PLANE + plo[i]) = preg[kc][i];
            // the next tile's plane, into the registers this one has just left
#pragma unroll
            for (int i = 0; i < PPI; ++i) preg[kc][i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, pvn[i], kc * T * 2, 0);
            __syncthreads();
            auto frags = [&](int q, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int n = 0; n < NT; ++n) a[buf][n] = *reinterpret_cast<const V*>(wl + ((n * 9 + widx[q]) * NPL + kc) * (16 * T) + aoff);
#pragma unroll
                for (int m = 0; m < MT; ++m) b[buf][m] = *reinterpret_cast<const V*>(patch + kc * G::PLANE + poff + m * PW * LDP + toff[q]);
            };
            frags(0, 0);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                if (q + 1 < 9) frags(q + 1, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n][m] = HFrag<T>::mma(a[q & 1][n], b[q & 1][m], acc[n][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int i0 = cur.i0, j0 = cur.j0, n_img = cur.n_img, ry = cur.ry, rx = cur.rx;
        u32x4 mreg[OITEMS], yreg[OITEMS];
        long long ooff[OITEMS];
        if (!pooled) {
#pragma unroll
            for (int k = 0; k < OITEMS; ++k) {
                const int it = tid + k * G::NTHR;
                const int q = it / PIECES, pc = it - q * PIECES;
                const int i = i0 + (q >> 4), j = j0 + (q & 15), ch = n_base + pc * 8;
                const bool ok = i < C.Hc && j < C.Wc && ch < p.Cout;
                const int ho = C.ph + ry + i * p.ostep, wo = C.pw + rx + j * p.ostep;
                const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
                ooff[k] = ok ? opix * p.y_ld + p.y_coff + ch : -1;
                if constexpr (EPI) {
                    mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(msrc, ok ? (unsigned)((opix * p.mul_ld + p.mul_coff + ch) * 2) : HV_OOB, 0, 0);
                    yreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, ok ? (unsigned)(ooff[k] * 2) : HV_OOB, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int k = PITEMS; k < OITEMS; ++k) { ooff[k] = -1; if constexpr (EPI) mreg[k] = yreg[k] = (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
            for (int k = 0; k < PITEMS; ++k) {
                const int it = tid + k * G::NTHR;
                const int q = it / PIECES, pc = it - q * PIECES;
                const int il = (i0 >> 1) + q / (TW / 2), jl = (j0 >> 1) + q % (TW / 2), ch = n_base + pc * 8;
                const bool ok = q < PPIX && il < p.Ho && jl < p.Wo && ch < p.Cout;
                const long long opix = (long long)(n_img * p.Ho + il) * p.Wo + jl;
                ooff[k] = ok ? opix * p.y_ld + p.y_coff + ch : -1;
                if constexpr (EPI) {
                    mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(msrc, ok ? (unsigned)((opix * p.mul_ld + p.mul_coff + ch) * 2) : HV_OOB, 0, 0);
                    yreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, ok ? (unsigned)(ooff[k] * 2) : HV_OOB, 0, 0);
                }
            }
        }
        if constexpr (X1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (tid + u * G::NTHR < PH * PW) x1p[tid + u * G::NTHR] = x1v[u];
            }
            if (more) x1_loads(nxt);
        }
        __syncthreads();          // every wave is done with the patch: its room becomes the output staging tile
        if constexpr (X1) {
            float cv[MT][9];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int q = (wave * MT + m) * 16 + (lane & 15);
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) {
                    const uint32_t e = C.taps[t9];
                    cv[m][t9] = x1p[((q >> 4) + (int)(e & 0xff)) * PW + (q & 15) + (int)((e >> 8) & 0xff)];
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                // two channels' filter values per batch of LDS reads (conv_lf_kernel takes four: the next patch's registers are live here)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    float w9[2][9];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int t9 = 0; t9 < 9; ++t9) w9[r][t9] = w1s[(n * 16 + (lane >> 4) * 4 + rh * 2 + r) * 9 + widx[t9]];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            float ex = 0.f;
#pragma unroll
                            for (int t9 = 0; t9 < 9; ++t9) ex += cv[m][t9] * w9[r][t9];
                            float v = acc[n][m][rh * 2 + r] + ex;
                            asm volatile("" : "+v"(v) : : "memory");
                            acc[n][m][rh * 2 + r] = v;
                        }
                }
            }
        }
        auto stage = [&](auto actf) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int q = (wave * MT + m) * 16 + (lane & 15);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f16x4v h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (_Float16)actf(acc[n][m][r] * p.alpha + bias_r[n][r]);
                    *reinterpret_cast<f16x4v*>(ot + q * LDO + n * 16 + (lane >> 4) * 4) = h;
                }
            }
        };
        switch (p.act) {
            case HV_ACT_ELU:
                stage([](float v) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504f) - 1.f, sm = v + 0.5f * v * v; return v > 0.f ? v : (v > -0.00390625f ? sm : e); });
                break;
            case HV_ACT_RELU: stage([](float v) { return v > 0.f ? v : 0.f; }); break;
            case HV_ACT_LRELU: stage([](float v) { return v > 0.f ? v : 0.2f * v; }); break;
            case HV_ACT_SIGMOID: stage([](float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504f)); }); break;
            case HV_ACT_CLAMP: stage([](float v) { return fminf(fmaxf(v, -1.f), 1.f); }); break;
            default: stage([](float v) { return v; }); break;
        }
        __syncthreads();
        u32x4 o[OITEMS];
        if (!pooled) {
#pragma unroll
            for (int k = 0; k < OITEMS; ++k) {
                const int it = tid + k * G::NTHR;
                const int q = it / PIECES, pc = it - q * PIECES;
                o[k] = *reinterpret_cast<const u32x4*>(ot + q * LDO + pc * 8);
            }
        } else {
#pragma unroll
            for (int k = PITEMS; k < OITEMS; ++k) o[k] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < PITEMS; ++k) {
                const int it = tid + k * G::NTHR;
                const int q = it / PIECES, pc = it - q * PIECES;
                const int r2 = 2 * (q / (TW / 2)), c2 = 2 * (q % (TW / 2));
                float sum[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[e] = 0.f;
                if (q < PPIX) {
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int rr = r2 + (dd >> 1), cc = c2 + (dd & 1);
                        if (i0 + rr < C.Hc && j0 + cc < C.Wc) {
                            const f16x8 v8 = *reinterpret_cast<const f16x8*>(ot + (rr * TW + cc) * LDO + pc * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) sum[e] += (float)v8[e];
                        }
                    }
                }
                f16x8 h8;
#pragma unroll
                for (int e = 0; e < 8; ++e) h8[e] = (_Float16)sum[e];
                o[k] = __builtin_bit_cast(u32x4, h8);
            }
        }
        if (more) __syncthreads();      // the staging tile has been read: the next pass may put its patch there
        if (EPI && p.mul_src) {
            auto mulf = [&](auto gradf) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < OITEMS; ++k) {
                    const f16x8 m8 = __builtin_bit_cast(f16x8, mreg[k]);
                    f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] * gradf((float)m8[e]));
                    o[k] = __builtin_bit_cast(u32x4, v8);
                }
            };
            switch (p.mul_act) {
                case HV_ACT_ELU: mulf([](float y) { return y > 0.f ? 1.f : y + 1.f; }); break;
                case HV_ACT_RELU: mulf([](float y) { return y > 0.f ? 1.f : 0.f; }); break;
                case HV_ACT_LRELU: mulf([](float y) { return y > 0.f ? 1.f : 0.2f; }); break;
                case HV_ACT_SIGMOID: mulf([](float y) { return y * (1.f - y); }); break;
                case HV_ACT_CLAMP: mulf([](float y) { return (y > -1.f && y < 1.f) ? 1.f : 0.f; }); break;
                default: break;
            }
        }
        if (EPI && p.accumulate) {
#pragma unroll
            for (int k = 0; k < OITEMS; ++k) {
                const f16x8 y8 = __builtin_bit_cast(f16x8, yreg[k]);
                f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[e]);
                o[k] = __builtin_bit_cast(u32x4, v8);
            }
        }
#pragma unroll
        for (int k = 0; k < OITEMS; ++k)
            if (ooff[k] >= 0) *reinterpret_cast<u32x4*>(yb + ooff[k]) = o[k];
        cur = nxt;
        tcur = tn;
    };
    // the filters arrive by DMA, which the compiler's own wait counting does not tie to the LDS reads below: everything requested so far (filters, first patch,
    // bias) has landed before the first pass (one full wait per workgroup, not per tile)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    do pass(); while (more);
}


template <int CIN, int CO, int TH, int WPS, bool X1 = false>
static int launch_lf(HaloK& k, hipStream_t s) {
    typedef LfCfg<CIN, CO, TH> G;
    static_assert(G::LDS_BYTES <= 160 * 1024, "filters + patch exceed the LDS");
    HaloK kk = k;
    HaloCls& C = kk.cls[0];
    C.tiles_x = hv_cdiv(C.Wc, G::TW);
    C.tiles = C.tiles_x * hv_cdiv(C.Hc, TH);
    C.t0 = 0;
    C.PH = G::PH; C.PW = G::PW;
    kk.w = kk.wt; kk.w_bytes = kk.wt_bytes;
    if (hv_probe_only) return HV_OK;          // hv_conv2d_supported: this instantiation would take the descriptor
    auto kern = conv_lf_kernel<CIN, CO, TH, WPS, X1>;
    static bool raised = false;
    if (G::LDS_BYTES > 48 * 1024 && !raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    const int total = C.tiles * kk.B * (kk.dil > 1 ? kk.dil * kk.dil : 1);
    dim3 grid(total, hv_cdiv(kk.Cout, CO));
    hv_path_note = 7;
    HV_WUSE(X1 ? 4 | 1 : 4);      // (the extra channel's filters come from the fp32 forward table)
    // more than one round of workgroups where only ONE workgroup fits a CU (64 input channels: 80 .. 136 KB of LDS; the extra-channel forms: registers): the
    // resident form (conv_lfp_kernel), one workgroup per CU walking its share of the tiles.  Back-to-back launches, bs 16: 64 -> 64 @128x128 40.7 -> 34.3 us,
    // 64 -> 32 27.1 -> 21.6 us; with two workgroups per CU (32 input channels) the second workgroup already hides the first one's loads and the resident form
    // gains nothing (32 -> 32 @128x128 14.1 vs 14.3 us, 32 -> 64 with the registers of one workgroup per CU 22.7 -> 25.2 us): those stay as they are.
    static const int persist = getenv("HV_LF_PERSIST") ? atoi(getenv("HV_LF_PERSIST")) : 1;
    constexpr int per_cu = (int)((160 * 1024) / G::LDS_BYTES) < WPS / 2 ? (int)((160 * 1024) / G::LDS_BYTES) : WPS / 2;
    // (in the step: 64 -> 64 + 1 @128x128 52.7 -> 44.0 us, 64 -> 32 24.2 -> 19.9, 64 -> 64 data gradient 34.3 -> 26.9; the 32 + 1 -> 32 layer @256x256, one workgroup per CU
    // by registers only, 64.6 -> 75.3 us: 64 input channels only)
    if constexpr (CIN == 64 && per_cu <= 1) {
        constexpr int slots = 256;      // MI355X: 256 CUs
        if (persist && grid.y == 1 && total > slots && !(X1 && (kk.mul_src || kk.accumulate))) {      // (single-round grids, measured: 11.4 vs 11.7 us -- stay)      // (the extra channel is a forward form: no act' / accumulate there)
            constexpr int WPSP = 2;     // (the next patch stays in registers across the whole pass)
            auto kp = conv_lfp_kernel<CIN, CO, TH, WPSP, X1>;
            static bool raised_p = false;
            if (G::LDS_BYTES > 48 * 1024 && !raised_p) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return -1000 - (int)e;
                raised_p = true;
            }
            grid.x = slots;
            HV_KNAME("conv_lfp_kernel<%d, %d, %d, %d, %s>", CIN, CO, TH, WPSP, X1 ? "true" : "false");
            hipLaunchKernelGGL(kp, grid, dim3(G::NTHR), G::LDS_BYTES, s, kk, total);
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
    }
    HV_KNAME("conv_lf_kernel<%d, %d, %d, %d, %s>", CIN, CO, TH, WPS, X1 ? "true" : "false");      // (as rocprofv3 prints the instantiation)
    hipLaunchKernelGGL(kern, grid, dim3(G::NTHR), G::LDS_BYTES, s, kk);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

template <int CIN, int CO, int WPS, int SGE>
static int launch_lfd(HaloK& k, hipStream_t s) {
    typedef LfCfg<CIN, CO, 16, SGE> G;
    static_assert(G::LDS_BYTES <= 160 * 1024, "filters + patch exceed the LDS");
    constexpr int NSG = 16 / SGE;
    HaloK kk = k;
    HaloCls& C = kk.cls[0];
    C.tiles_x = kk.dil / NSG;
    C.tiles = C.tiles_x * C.tiles_x;
    C.t0 = 0;
    C.PH = G::PH; C.PW = G::PW;
    kk.w = kk.wt; kk.w_bytes = kk.wt_bytes;
    if (hv_probe_only) return HV_OK;
    auto kern = conv_lfd_kernel<CIN, CO, WPS, SGE>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    dim3 grid(C.tiles * kk.B, hv_cdiv(kk.Cout, CO));
    hv_path_note = 7;
    HV_WUSE(4);
    HV_KNAME("conv_lfd_kernel<%d, %d, %d, %d>", CIN, CO, WPS, SGE);
    hipLaunchKernelGGL(kern, grid, dim3(G::NTHR), G::LDS_BYTES, s, kk);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

template <int CIN, int CO, int WPS>
static int launch_lf2(HaloK& k, hipStream_t s) {
    typedef LfCfg<CIN, CO, 16, 16, 2> G;
    static_assert(G::LDS_BYTES <= 160 * 1024, "filters + patch exceed the LDS");
    HaloK kk = k;
    HaloCls& C = kk.cls[0];
    C.tiles_x = hv_cdiv(C.Wc, G::TW);
    C.tiles = C.tiles_x * hv_cdiv(C.Hc, 16);
    C.t0 = 0;
    C.PH = G::PH; C.PW = G::PW;
    kk.w = kk.wt; kk.w_bytes = kk.wt_bytes;
    if (hv_probe_only) return HV_OK;
    auto kern = conv_lf2_kernel<CIN, CO, WPS>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    dim3 grid(C.tiles * kk.B, hv_cdiv(kk.Cout, CO));
    hv_path_note = 7;
    HV_WUSE(4);
    HV_KNAME("conv_lf2_kernel<%d, %d, %d>", CIN, CO, WPS);
    hipLaunchKernelGGL(kern, grid, dim3(G::NTHR), G::LDS_BYTES, s, kk);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// 3x3 stride-1 layers (forward, or the data gradient = the same convolution with the transposed filter table) whose input and output are fp16
// NHWC views with 16-byte aligned channel rows.  Returns HV_ERR_UNSUPPORTED for everything else (the caller goes on to conv_halo2_kernel).
int hv_convlf_launch(HaloK& k, int KH, int KW, hipStream_t s) {
    static const int on = getenv("HV_CONV_LF") ? atoi(getenv("HV_CONV_LF")) : 1;
    static const int s2on = getenv("HV_LF_S2") ? atoi(getenv("HV_LF_S2")) : 1;      // A/B knob: stride-2 forward layers (conv_lf2_kernel)
    if (!on || KH != 3 || KW != 3 || k.ncls != 1 || (k.bstep != 1 && !(k.bstep == 2 && s2on)) || k.cls[0].ntaps != 9) return HV_ERR_UNSUPPORTED;
    if (!k.wt || ((uintptr_t)k.wt & 15) || !k.x_half || !k.y_half || k.accumulate > 1) return HV_ERR_UNSUPPORTED;
    if ((k.x_ld & 7) || (k.x_coff & 7) || ((uintptr_t)k.x & 15) || (k.Cout & 7) || (k.y_ld & 7) || (k.y_coff & 7) || ((uintptr_t)k.y & 15)) return HV_ERR_UNSUPPORTED;
    if (k.mul_src && (!k.mul_half || (k.mul_ld & 7) || (k.mul_coff & 7) || ((uintptr_t)k.mul_src & 15))) return HV_ERR_UNSUPPORTED;
    // the 3x3 taps must span dh, dw in 0..2 around (dh_min, dw_min) -- true for pad-1 forward and its data gradient
    const int Cin = k.Cin, Cout = k.Cout;
    // HV_CONV_LF_MASK: bit per (Cin class 16 / 32 / 64) x (Cout class <= 16 / <= 32 / > 32), an A/B knob
    static const int mask = getenv("HV_CONV_LF_MASK") ? atoi(getenv("HV_CONV_LF_MASK")) : 0x1ff;
    // (Measured and not kept, round 3: the two concat layers (32 + 1 / 64 + 1 input channels) with their buffers widened to 48 / 80 channels and
    // <48, 32, 16> / <80, 64, 8> instantiations on 16-channel planes: step 8.76 vs 8.78 ms -- no gain over conv_halo2's ragged 16-channel chunks.)
    const int ci = Cin == 16 ? 0 : Cin == 32 ? 1 : Cin == 64 ? 2 : -1, co = Cout <= 16 ? 0 : Cout <= 32 ? 1 : 2;
    if (ci < 0 || !((mask >> (ci * 3 + co)) & 1)) return HV_ERR_UNSUPPORTED;
    if (k.bstep == 2) {      // stride-2 forward (a stride-2 data gradient has four tap classes: never here)
        if (k.dil != 1 || k.x1 || k.pool2 || k.in_shift || k.mul_src) return HV_ERR_UNSUPPORTED;
        switch (ci * 3 + co) {
            case 0: return launch_lf2<16, 16, 4>(k, s);
            case 1: return launch_lf2<16, 32, 4>(k, s);
            case 4: return launch_lf2<32, 32, 2>(k, s);
            case 5: return launch_lf2<32, 64, 2>(k, s);
            default: return HV_ERR_UNSUPPORTED;
        }
    }
    if (k.dil > 1 && (k.Hl < 16 || k.Wl < 16)) {
        // residue sub-grids smaller than a tile: the packed form, where the sub-grid is the whole residue class (sub-grid edge x dilation = the map) -- the
        // generators' d = 8 / d = 16 layers on 64 x 64 maps.  Other small sub-grids: d <= 4 as partly filled tiles below (as before), d >= 8 on to
        // conv_halo2_kernel / the gather kernel
        static const int packed = getenv("HV_LF_PACKED") ? atoi(getenv("HV_LF_PACKED")) : 1;      // A/B knob
        const HaloCls& C = k.cls[0];
        const bool fits = packed && !k.x1 && !k.pool2 && !k.in_shift && k.Hl == k.Wl && k.boff + C.dh_min == -1 && k.boff + C.dw_min == -1 && C.Hc == k.Hl && C.Wc == k.Wl &&
                          Cin == 64 && Cout == 64;
        if (fits && k.Hl == 8 && k.dil % 2 == 0) return launch_lfd<64, 64, 2, 8>(k, s);
        if (fits && k.Hl == 4 && k.dil % 4 == 0) return launch_lfd<64, 64, 2, 4>(k, s);
        if (k.dil > 4) return HV_ERR_UNSUPPORTED;
    }
    // (Measured and not kept, round 3: 32 x 16-pixel tiles for the 256 x 256 layers -- half the filter loads and workgroups per pixel: 30.1 / 21.4 / 37.5 /
    // 22.7 us against 28.8 / 23.1 / 34.9 / 23.8 us with 16 x 16 tiles, step 8.20 -> 8.23 ms.)
    if (k.x1) {      // extra input channel: the two shapes that have it (32 + 1 -> 32, 64 + 1 -> 64)
        if (Cin == 32 && Cout == 32) return launch_lf<32, 32, 16, 3, true>(k, s);
        if (Cin == 64 && Cout == 64) return launch_lf<64, 64, 16, 2, true>(k, s);
        return HV_ERR_UNSUPPORTED;
    }
    switch (ci * 3 + co) {
        case 0: return launch_lf<16, 16, 16, 4>(k, s);
        case 1: return launch_lf<16, 32, 16, 4>(k, s);
        case 2: return HV_ERR_UNSUPPORTED;      // (no such layer)
        case 3: return launch_lf<32, 16, 16, 4>(k, s);
        case 4: return launch_lf<32, 32, 16, 4>(k, s);
        case 5: return launch_lf<32, 64, 16, 4>(k, s);
        case 6: return launch_lf<64, 16, 16, 2>(k, s);
        case 7: return launch_lf<64, 32, 16, 2>(k, s);
        default: return launch_lf<64, 64, 16, 2>(k, s);
    }
    return HV_ERR_UNSUPPORTED;
}
