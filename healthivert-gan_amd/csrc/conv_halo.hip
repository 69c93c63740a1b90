// Halo-tiled implicit-GEMM convolution (fp16 MFMA operands, fp32 accumulate) for gfx950.
//
// The gather kernel in conv_igemm.hip re-reads the input from L2 once per filter tap, which pins it at
// the per-CU L2->LDS load rate (~70 GB/s/CU) far below the MFMA rate.  This kernel stages the input
// ONCE per (output tile, channel chunk): a TH x TW tile of output pixels needs a (TH-1)*s+span+1 square
// patch of the input, kept in LDS as fp16 [patch pixel][CK channels]; every tap then reads its MFMA
// B-fragments (8 contiguous channels of 16 consecutive pixels) straight from the patch at a shifted
// address -- no im2col copy, 9-16x less L2 traffic.  Weights (pre-converted to fp16 by hv_weight_prep)
// stream through a double-buffered LDS tile, TG taps per barrier.
//
// Used for: dilation 1, Cin % 16 == 0, shared (not per-sample) filters, HV_F16 precision; conv and
// the gather form of conv_transpose / data gradient (per output-parity class).  Everything else stays on
// conv_igemm_kernel.
#include <stdlib.h>

#include "conv_halo.h"

template <int TH, int TW, int BN, int WM, int WN, int TG, int CK, int PMAX, bool XH>
__global__ __launch_bounds__(256) void conv_halo_kernel(const HaloK p) {
    typedef HvSt<XH> XS;                         // storage of the input tensor: fp32 (converted when staged) or fp16 (staged as it is)
    constexpr int BM = TH * TW;
    // halfs per patch pixel / weight row.  With 16-B fragments, a 96-B row stride (CK 32 + 16 pad) maps the 16 lanes of every
    // ds_read_b128 lane group onto all 64 banks exactly once for unit-step pixel reads; stride-2 reads prefer 80 B.
    constexpr int LDP = CK + ((CK == 32 && PMAX < 16) ? 16 : 8);
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
    constexpr int GX = TW / 16;                  // 16-pixel groups per tile row
    static_assert(WM * WN == 4 && MT >= 1 && NT >= 1 && TW % 16 == 0, "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* wbuf = reinterpret_cast<_Float16*>(smem);                  // [2][TG][BN][LDP]
    _Float16* patch = wbuf + 2 * TG * BN * LDP;                         // [PH*PW][LDP]
    __shared__ uint32_t taps_s[16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ci = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < p.ncls && (int)blockIdx.x >= p.cls[i].t0) ci = i;
    const HaloCls& C = p.cls[ci];
    const int ntaps = C.ntaps, PH = C.PH, PW = C.PW;
    if (tid < ntaps) taps_s[tid] = C.taps[tid];
    int t = (int)blockIdx.x - C.t0;
    const int n_img = t / C.tiles;
    t -= n_img * C.tiles;
    const int tile_y = t / C.tiles_x, tile_x = t - tile_y * C.tiles_x;
    const int i0 = tile_y * TH, j0 = tile_x * TW;           // class-pixel origin of the tile
    const int n_base = blockIdx.y * BN;
    // input coordinate of patch pixel (0,0)
    const int h0 = i0 * p.bstep + p.boff + C.dh_min, w0 = j0 * p.bstep + p.boff + C.dw_min;
    const void* ximg = hv_eptr(p.x, (long long)n_img * p.img_stride + p.x_coff, XH);
    const int npatch = PH * PW;
    const int wm = wave / WN, wn = wave % WN;

    // per-lane patch offsets of the MT pixel groups this wave owns (without the tap shift)
    int poff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
        poff[m] = (ty * p.bstep * PW + tx * p.bstep) * LDP;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight tile loader: rows (tap-in-group, n) x CK halfs, 16 B (8 halfs) per thread-load.  Two register sets
    // keep two tap groups in flight (prefetch distance 2) behind the MFMAs of the current group.
    constexpr int WVEC = CK / 8;                             // 16-B vectors per row
    constexpr int WLOADS = (TG * BN * WVEC + 255) / 256;
    constexpr bool WTAP_UNIFORM = (BN * WVEC) % 256 == 0;    // every load instruction of the workgroup stays inside one tap
    // Staging is kept off the vector ALU (an MFMA leaves only half its cycles to other vector instructions): row
    // offsets are computed once, the tap/chunk part of the address is a scalar buffer offset, rows beyond Cout get an
    // out-of-range offset (the descriptor's range check returns zeros) instead of a branch.
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned wvo[WLOADS];                                    // byte offset of (row n, vector) inside w, without tap / chunk
    int wlo[WLOADS];                                         // LDS offset (halfs) inside one weight buffer, -1 = no element
    int wtg[WLOADS];                                         // tap-in-group of the element
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
        const int e = tid + i * 256;
        const int vec = e % WVEC, r = e / WVEC, n = r % BN;
        const bool in = e < TG * BN * WVEC;
        wtg[i] = r / BN;
        wlo[i] = in ? r * LDP + vec * 8 : -1;
        wvo[i] = (in && n_base + n < p.Cout) ? (unsigned)(((n_base + n) * p.w_row + vec * 8) * 2) : HV_OOB;
    }
    uint4 wra[WLOADS], wrb[WLOADS];
    auto wload = [&](uint4 (&wr)[WLOADS], int grp, int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            if (WTAP_UNIFORM) {
                const int tap = grp * TG + (i * 256) / (BN * WVEC);
                u32x4 v = {0u, 0u, 0u, 0u};
                if (tap < ntaps) {   // scalar condition
                    const int widx = (int)(__builtin_amdgcn_readfirstlane(taps_s[tap]) >> 16);
                    v = __builtin_amdgcn_raw_buffer_load_b128(wsrc, wvo[i], (widx * p.Cin + c0) * 2, 0);
                }
                wr[i] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                const int tap = grp * TG + wtg[i];
                const int widx = (int)(taps_s[tap < ntaps ? tap : 0] >> 16);
                const unsigned off = tap < ntaps ? wvo[i] + (unsigned)((widx * p.Cin + c0) * 2) : HV_OOB;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wsrc, off, 0, 0);
                wr[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto wstore = [&](const uint4 (&wr)[WLOADS], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i)
            if (wlo[i] >= 0) *reinterpret_cast<uint4*>(wbuf + buf * TG * BN * LDP + wlo[i]) = wr[i];
    };
    // input patch: small patches are prefetched into registers one chunk ahead (offsets computed once, buffer loads
    // with the chunk as scalar offset), large ones staged synchronously
    constexpr int PV = CK / 4;
    const bool patch_pf = npatch * PV <= PMAX * 256;
    typename XS::R preg[PMAX];
    unsigned pvo[PMAX];          // byte offset of (patch pixel, channel quad) in x, HV_OOB outside the image / patch
    int plo[PMAX];               // LDS offset (halfs), -1 = no element
    const unsigned xbase = (unsigned)(n_img * p.img_stride + p.x_coff) * XS::B;
    if (patch_pf) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int e = tid + i * 256;
            const int c4 = e % PV, pix = e / PV;
            const int py = pix / PW, px = pix - py * PW;
            const int hi = h0 + py, wi = w0 + px;
            const bool in = e < npatch * PV;
            plo[i] = in ? pix * LDP + c4 * 4 : -1;
            pvo[i] = (in && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
                         ? xbase + (unsigned)(((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + c4 * 4) * XS::B : HV_OOB;
        }
    }
    auto ppref = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) preg[i] = XS::ld(xsrc, pvo[i], c0 * (int)XS::B);
    };
    auto pflush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            if (plo[i] < 0) continue;
            *reinterpret_cast<f16x4v*>(patch + plo[i]) = XS::h4(preg[i]);
        }
    };
    auto pload1 = [&](int e, int c0) -> float4 {
        const int c4 = e % PV, pix = e / PV;
        const int py = pix / PW, px = pix - py * PW;
        const int hi = h0 + py, wi = w0 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
            v = hv_ld4(ximg, (long long)((hi >> p.in_shift) * p.Wp + (wi >> p.in_shift)) * p.x_ld + c0 + c4 * 4, XH);
        return v;
    };
    auto pstore1 = [&](int e, float4 v) {
        const int c4 = e % PV, pix = e / PV;
        f16x4v h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<f16x4v*>(patch + pix * LDP + c4 * 4) = h;
    };
    auto compute = [&](int g) {
        const _Float16* wb = wbuf + (g & 1) * TG * BN * LDP;
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            const int tap = g * TG + tg;
            if (tap >= ntaps) break;
            const uint32_t e = __builtin_amdgcn_readfirstlane(taps_s[tap]);
            const int toff = ((int)(e & 0xff) * PW + (int)((e >> 8) & 0xff)) * LDP;
            typename HFrag<CK>::V wf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = HFrag<CK>::ld(wb + (tg * BN + wn * (BN / WN) + n * 16 + (lane & 15)) * LDP, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const typename HFrag<CK>::V xf = HFrag<CK>::ld(patch + poff[m] + toff, lane);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n][m] = HFrag<CK>::mma(wf[n], xf, acc[n][m]);
            }
        }
    };

    __syncthreads();   // taps_s
    const int ngroups = (ntaps + TG - 1) / TG;
    if (patch_pf) ppref(0);
    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        // ---- patch of this chunk -> LDS (all MFMA reads of the previous chunk finished at its last barrier)
        if (patch_pf) {
            pflush();
        } else {
            for (int e = tid; e < npatch * PV; e += 256) pstore1(e, pload1(e, c0));
        }
        wload(wra, 0, c0);
        wstore(wra, 0);
        if (ngroups > 1) wload(wrb, 1, c0);
        if (ngroups > 2) wload(wra, 2, c0);
        if (patch_pf && c0 + CK < p.Cin) ppref(c0 + CK);   // next chunk's patch rides behind this chunk's MFMAs
        __syncthreads();
        for (int g = 0; g < ngroups; g += 2) {
            compute(g);
            if (g + 1 < ngroups) { wstore(wrb, 1); if (g + 3 < ngroups) wload(wrb, g + 3, c0); }
            __syncthreads();
            if (g + 1 >= ngroups) break;
            compute(g + 1);
            if (g + 2 < ngroups) { wstore(wra, 0); if (g + 4 < ngroups) wload(wra, g + 4, c0); }
            __syncthreads();
        }
    }

    // ---- epilogue (same contract as conv_igemm_kernel)
    const HvEpi epi = {p.alpha, p.act, p.accumulate, p.vec_store, p.Cout, p.bias, nullptr, p.mul_act, p.mul_vec, p.y_half, p.mul_half};
    if (p.ep16) {
        // fp16 output tile through LDS, stored as 16-byte pieces of whole channel rows (as in conv_halo2_kernel: the direct 8-byte stores below were
        // 13.5 of the 46 us of the 64 -> 128 stride-2 PatchGAN layer).  All LDS buffers are free: the last tap group ends behind a barrier.
        constexpr int LDO = BN + 8;
        _Float16* ot = reinterpret_cast<_Float16*>(smem);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
            const int i = i0 + ty, j = j0 + tx;
            const bool inside = i < C.Hc && j < C.Wc;
            const int ho = C.ph + i * p.ostep, wo = C.pw + j * p.ostep;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            const void* mp = (p.mul_src && inside && p.ep16 == 1) ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                const int cl = wn * (BN / WN) + nn * 16 + (lane >> 4) * 4;
                const f32x4 v = hv_conv_value4<true>(epi, acc[nn][m], n_base + cl, mp);
                *reinterpret_cast<f16x4v*>(ot + (g * 16 + (lane & 15)) * LDO + cl) = (f16x4v){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        __syncthreads();
        constexpr int PIECES = BN / 8;
        _Float16* yb = reinterpret_cast<_Float16*>(p.y);
        for (int it = tid; it < TH * TW * PIECES; it += 256) {
            const int q = it / PIECES, pc = it - q * PIECES;
            const int g = q >> 4, ty = g / GX, tx = (g % GX) * 16 + (q & 15);
            const int i = i0 + ty, j = j0 + tx, ch = n_base + pc * 8;
            if (i >= C.Hc || j >= C.Wc || ch >= p.Cout) continue;
            const int ho = C.ph + i * p.ostep, wo = C.pw + j * p.ostep;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            u32x4 o = *reinterpret_cast<const u32x4*>(ot + q * LDO + pc * 8);
            if (p.ep16 == 2) {
                const f16x8 m8 = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(p.mul_src) + opix * p.mul_ld + p.mul_coff + ch);
                f16x8 v8 = __builtin_bit_cast(f16x8, o);
                {
                    float f0[4] = {(float)m8[0], (float)m8[1], (float)m8[2], (float)m8[3]}, f1[4] = {(float)m8[4], (float)m8[5], (float)m8[6], (float)m8[7]};
                    hv_act_grad4(f0, p.mul_act); hv_act_grad4(f1, p.mul_act);      // (one switch per quad, not per element)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v8[e] = (_Float16)((float)v8[e] * f0[e]); v8[4 + e] = (_Float16)((float)v8[4 + e] * f1[e]); }
                }
                o = __builtin_bit_cast(u32x4, v8);
            }
            if (p.accumulate) {
                const f16x8 y8 = *reinterpret_cast<const f16x8*>(yb + opix * p.y_ld + p.y_coff + ch);
                f16x8 v8 = __builtin_bit_cast(f16x8, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[e]);
                o = __builtin_bit_cast(u32x4, v8);
            }
            *reinterpret_cast<u32x4*>(yb + opix * p.y_ld + p.y_coff + ch) = o;
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
        const int i = i0 + ty, j = j0 + tx;
        if (i >= C.Hc || j >= C.Wc) continue;
        const int ho = C.ph + i * p.ostep, wo = C.pw + j * p.ostep;
        const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
        void* yp = hv_eptr(p.y, opix * p.y_ld + p.y_coff, p.y_half);
        const void* mp = p.mul_src ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const int ch0 = n_base + wn * (BN / WN) + nn * 16 + (lane >> 4) * 4;
            hv_conv_epilogue4<true>(epi, acc[nn][m], ch0, yp, mp);
        }
    }
}

template <int TH, int TW, int BN, int WM, int WN, int TG, int CK, int PMAX, bool XH>
static int launch_halo_t(const HaloK& k, int tiles, int maxpatch, hipStream_t s) {
    constexpr int LDP = CK + ((CK == 32 && PMAX < 16) ? 16 : 8);
    const size_t lds = (size_t)(2 * TG * BN * LDP + maxpatch * LDP) * sizeof(_Float16);
    if (lds > 150 * 1024) return HV_ERR_UNSUPPORTED;
    auto kern = conv_halo_kernel<TH, TW, BN, WM, WN, TG, CK, PMAX, XH>;
    static int lds_limit = 48 * 1024;   // per instantiation: raise the dynamic-LDS cap once (not a stream operation)
    if ((int)lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        lds_limit = 150 * 1024;
    }
    dim3 grid(tiles, hv_cdiv(k.Cout, BN));
    hv_path_note = 2;
    HV_KNAME("conv_halo_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %s>", TH, TW, BN, WM, WN, TG, CK, PMAX, XH ? "true" : "false");
    HaloK kk = k;
    {   // coalesced fp16 epilogue through LDS (HV_HALO2_EP16=0: direct 8-byte stores)
        static const int ep16 = getenv("HV_HALO2_EP16") ? atoi(getenv("HV_HALO2_EP16")) : 1;
        kk.ep16 = (ep16 && kk.y_half && kk.accumulate <= 1 && !(kk.Cout & 7) && !(kk.y_ld & 7) && !(kk.y_coff & 7) && !((uintptr_t)kk.y & 15) &&
                   lds >= (size_t)TH * TW * (BN + 8) * sizeof(_Float16)) ? 1 : 0;
        if (kk.ep16 && kk.mul_src && kk.mul_half && !(kk.mul_ld & 7) && !(kk.mul_coff & 7) && !((uintptr_t)kk.mul_src & 15)) kk.ep16 = 2;
    }
    HV_WUSE(2);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, kk);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
template <int TH, int TW, int BN, int WM, int WN, int TG, int CK, int PMAX>
static int launch_halo(const HaloK& k, int tiles, int maxpatch, hipStream_t s) {
    return launch_halo_t<TH, TW, BN, WM, WN, TG, CK, PMAX, true>(k, tiles, maxpatch, s);
}

template <int TH, int TW, int CK, bool S2>
static int dispatch_halo_bn(HaloK& k, int maxpatch, hipStream_t s) {
    int tiles = 0;
    for (int c = 0; c < k.ncls; ++c) {
        HaloCls& C = k.cls[c];
        C.tiles_x = hv_cdiv(C.Wc, TW);
        C.tiles = C.tiles_x * hv_cdiv(C.Hc, TH);
        C.t0 = tiles;
        tiles += C.tiles * k.B;
    }
    int bn = k.Cout <= 16 ? 16 : k.Cout <= 32 ? 32 : k.Cout <= 64 ? 64 : 128;
    // one wave per SIMD cannot hide the staging latency: prefer >= 2 workgroups per CU over the widest tile
    if (bn == 128 && (long long)tiles * hv_cdiv(k.Cout, 128) < 512) bn = 64;
    static const char* force = getenv("HV_HALO_BN");   // tuning knob (tools/bench_conv.py)
    if (force && k.Cout > 64) bn = atoi(force) == 64 ? 64 : 128;
    // stride-2 patches are 4x larger: prefetch them with more registers and halve the weight tile so that two
    // workgroups still fit in a CU's LDS
    constexpr int PM = S2 ? 20 : 11, TD = S2 ? 2 : 1;
    if (bn == 16) return launch_halo<TH, TW, 16, 4, 1, 16 / TD, CK, PM>(k, tiles, maxpatch, s);
    if (bn == 32) return launch_halo<TH, TW, 32, 4, 1, 8 / TD, CK, PM>(k, tiles, maxpatch, s);
    if (bn == 64) return launch_halo<TH, TW, 64, 2, 2, 4 / TD, CK, PM>(k, tiles, maxpatch, s);
    return launch_halo<TH, TW, 128, 2, 2, 2 / TD, CK, PM>(k, tiles, maxpatch, s);
}

// Called by hv_conv2d when the fp16 weight copy is present and the shape qualifies; returns HV_ERR_UNSUPPORTED to
// fall back to the gather kernel.
int hv_conv2d_halo(const hv_conv_desc* d, const void* w_f16, hipStream_t s) {
    // Dilation d in {2, 4, 8} of a same-size 3x3 layer: the output pixels of one residue class (y mod d, x mod d) only read input pixels of the same
    // class, so the layer is d*d independent UNDILATED 3x3 convolutions on sub-grids of H/d x W/d pixels whose neighbours lie d pixels apart.
    // conv_halo2_kernel tiles a sub-grid like an image (pixel step d in its address arithmetic only); the gather kernel these layers used to take
    // re-reads the input per tap through L2 (64 -> 64 channels @64^2: 29 us against 13-20 us; d = 16 leaves 4x4-pixel sub-grids: gather kernel)
    const int dil = d->dil;
    hv_conv_desc dd;
    const int Hf = d->H, Wf = d->W, Hof = d->Ho, Wof = d->Wo;
    if (dil != 1) {
        static const int dilated = getenv("HV_HALO_DILATED") ? atoi(getenv("HV_HALO_DILATED")) : 1;      // A/B knob
        if (!dilated || (dil != 2 && dil != 4 && dil != 8 && dil != 16) || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != dil || d->in_shift) return HV_ERR_UNSUPPORTED;
        if (d->H % dil || d->W % dil || d->Ho != d->H || d->Wo != d->W || (d->Cin & 31) || d->Cout > 64 || d->Cout <= 32) return HV_ERR_UNSUPPORTED;
        dd = *d;
        dd.H /= dil; dd.W /= dil; dd.Ho /= dil; dd.Wo /= dil; dd.pad = 1; dd.dil = 1;
        d = &dd;          // from here on: the undilated convolution of ONE sub-grid; the image-level strides are put back below
    }
    if ((d->Cin & 3) || d->w_bstride || d->ch_scale || d->KH * d->KW > 25 || d->stride > 2) return HV_ERR_UNSUPPORTED;
    if ((d->x_ld & 3) || (d->x_coff & 3) || ((uintptr_t)d->x & 15) || ((uintptr_t)w_f16 & 15)) return HV_ERR_UNSUPPORTED;
    if (!d->x_f16) return HV_ERR_UNSUPPORTED;      // halo-tiled kernels are built for fp16 storage (an fp32 input with fp16 operands: gather kernel)
    HaloK k;
    k.x_half = d->x_f16 ? 1 : 0; k.y_half = d->y_f16 ? 1 : 0; k.mul_half = d->mul_f16 ? 1 : 0;
    {   // fragment-ordered filters (A/B knob HV_W_TILED=0: plain rows)
        static const int tiled = getenv("HV_W_TILED") ? atoi(getenv("HV_W_TILED")) : 1;
        k.wt = (tiled && d->w_f16_tiled && !((uintptr_t)d->w_f16_tiled & 15)) ? reinterpret_cast<const _Float16*>(d->w_f16_tiled) : nullptr;
        k.wt_bytes = (unsigned)((size_t)hv_cdiv(d->Cout, 16) * 16 * d->KH * d->KW * d->Cin * sizeof(_Float16));
    }
    const size_t xs = k.x_half ? 2 : 4;
    const int Hp = d->H >> d->in_shift, Wp = d->W >> d->in_shift;
    k.x = d->x; k.w = (const _Float16*)w_f16; k.bias = d->bias; k.y = d->y;
    k.B = d->B; k.Hl = d->H; k.Wl = d->W; k.in_shift = d->in_shift; k.Wp = Wp; k.img_stride = Hp * Wp * d->x_ld;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    if ((long long)d->B * k.img_stride >= (1ll << 29) || (long long)d->Cout * d->KH * d->KW * d->Cin >= (1ll << 30)) return HV_ERR_UNSUPPORTED;
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * xs);
    k.w_bytes = (unsigned)((size_t)d->Cout * d->KH * d->KW * d->Cin * sizeof(_Float16));
    k.Cout = d->Cout; k.w_row = d->KH * d->KW * d->Cin; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Ho = d->Ho; k.Wo = d->Wo;
    k.x1 = d->x1; k.x1_half = d->x1_f16 ? 1 : 0; k.x1_ld = d->x1_ld; k.x1_coff = d->x1_coff; k.w1 = d->w1; k.w1_row = d->w1_row; k.w1_tap = d->w1_tap;
    k.pool2 = d->pool2 ? 1 : 0;
    if (k.pool2) { k.Ho = d->Ho / 2; k.Wo = d->Wo / 2; }      // the stored tensor; the classes below keep the convolution's own grid
    k.alpha = d->alpha; k.act = d->act; k.accumulate = d->accumulate;
    k.mul_src = d->mul_src; k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act;
    k.mul_vec = (d->mul_src && !(d->mul_ld & 3) && !(d->mul_coff & 3) && !((uintptr_t)d->mul_src & 15)) ? 1 : 0;
    k.vec_store = ((d->y_ld & 3) == 0 && (d->y_coff & 3) == 0 && ((uintptr_t)d->y & 15) == 0) ? 1 : 0;
    int dhs[4][25], dws[4][25], wis[4][25];
    if (!d->transposed) {
        k.ncls = 1; k.bstep = d->stride; k.boff = -d->pad; k.ostep = 1;
        HaloCls& c = k.cls[0];
        c.ph = c.pw = 0; c.Hc = d->Ho; c.Wc = d->Wo; c.ntaps = d->KH * d->KW;
        for (int r = 0; r < d->KH; ++r)
            for (int q = 0; q < d->KW; ++q) { dhs[0][r * d->KW + q] = r; dws[0][r * d->KW + q] = q; wis[0][r * d->KW + q] = r * d->KW + q; }
    } else {
        k.bstep = 1; k.boff = 0; k.ostep = d->stride; k.ncls = 0;
        for (int ph = 0; ph < d->stride; ++ph)
            for (int pw = 0; pw < d->stride; ++pw) {
                HaloCls& c = k.cls[k.ncls];
                c.ph = ph; c.pw = pw;
                c.Hc = (d->Ho - ph + d->stride - 1) / d->stride;
                c.Wc = (d->Wo - pw + d->stride - 1) / d->stride;
                if (c.Hc <= 0 || c.Wc <= 0) continue;
                c.ntaps = 0;
                for (int r = 0; r < d->KH; ++r) {
                    const int vh = ph + d->pad - r;
                    if (((vh % d->stride) + d->stride) % d->stride) continue;
                    for (int q = 0; q < d->KW; ++q) {
                        const int vw = pw + d->pad - q;
                        if (((vw % d->stride) + d->stride) % d->stride) continue;
                        dhs[k.ncls][c.ntaps] = vh / d->stride; dws[k.ncls][c.ntaps] = vw / d->stride; wis[k.ncls][c.ntaps] = r * d->KW + q;
                        ++c.ntaps;
                    }
                }
                if (c.ntaps == 0) return HV_ERR_UNSUPPORTED;   // rare (k < stride): let the gather kernel write the zeros
                ++k.ncls;
            }
        if (k.ncls == 0) return HV_ERR_UNSUPPORTED;
    }
    // tile shape: 8x32 output pixels for unit input step, 8x16 when the patch grows with stride 2
    bool small_tile = k.bstep == 2;
    if (!small_tile) {   // 8x32 tiles would leave CUs idle on small feature maps: fall back to 8x16
        long long wgs = 0;
        const int bn = d->Cout <= 16 ? 16 : d->Cout <= 32 ? 32 : d->Cout <= 64 ? 64 : 128;
        for (int c = 0; c < k.ncls; ++c) wgs += (long long)hv_cdiv(k.cls[c].Hc, 8) * hv_cdiv(k.cls[c].Wc, 32) * d->B * hv_cdiv(d->Cout, bn);
        if (wgs < 400) small_tile = true;
    }
    // 3x3 layers with whole 32-channel chunks: the 8x16-pixel conv_halo2 instantiations (twice the workgroups, half the accumulators)
    // measured faster than 8x32 on the 128x128 and 256x256 maps as well (53 vs 65, 22 vs 29, 36 vs 51 us); HV_HALO_TW16=0 restores 8x32
    static const int tw16 = getenv("HV_HALO_TW16") ? atoi(getenv("HV_HALO_TW16")) : 1;
    static const int tw16r = getenv("HV_HALO_TW16R") ? atoi(getenv("HV_HALO_TW16R")) : 1;   // also for 16-channel chunks (measured: 49 vs 54, 31 vs 42, 65 vs 81, 115 vs 105 us)
    if (tw16 && k.bstep == 1 && d->KH == 3 && d->KW == 3 && ((d->Cin & 31) == 0 || (tw16r && d->Cout <= 64))) small_tile = true;
    static const int tw16x = getenv("HV_HALO_TW16X") ? atoi(getenv("HV_HALO_TW16X")) : 1;   // bit 1 = stride-2 data gradient classes (54 vs 61 us), bit 2 = 5x5 (no gain)
    if ((tw16x & 1) && d->transposed && d->stride == 2 && d->KH == 4) small_tile = true;
    if ((tw16x & 2) && d->KH == 5 && k.bstep == 1) small_tile = true;
    const int TH = 8, TW = small_tile ? 16 : 32;
    int maxpatch = 0;
    for (int c = 0; c < k.ncls; ++c) {
        HaloCls& C = k.cls[c];
        int dh0 = 1 << 20, dh1 = -(1 << 20), dw0 = 1 << 20, dw1 = -(1 << 20);
        for (int t = 0; t < C.ntaps; ++t) {
            dh0 = dhs[c][t] < dh0 ? dhs[c][t] : dh0; dh1 = dhs[c][t] > dh1 ? dhs[c][t] : dh1;
            dw0 = dws[c][t] < dw0 ? dws[c][t] : dw0; dw1 = dws[c][t] > dw1 ? dws[c][t] : dw1;
        }
        C.dh_min = dh0; C.dw_min = dw0;
        C.PH = (TH - 1) * k.bstep + (dh1 - dh0) + 1;
        C.PW = (TW - 1) * k.bstep + (dw1 - dw0) + 1;
        for (int t = 0; t < C.ntaps; ++t)
            C.taps[t] = (uint32_t)(dhs[c][t] - dh0) | ((uint32_t)(dws[c][t] - dw0) << 8) | ((uint32_t)wis[c][t] << 16);
        if (C.PH * C.PW > maxpatch) maxpatch = C.PH * C.PW;
    }
    const bool ck32 = (d->Cin & 31) == 0;
    k.dil = dil;
    if (dil != 1) {     // image-level addressing of the sub-grid convolution (bounds stay those of the sub-grid: k.Hl, k.Wl, cls[].Hc / Wc)
        k.Wp = Wf; k.img_stride = Hf * Wf * d->x_ld; k.Ho = Hof; k.Wo = Wof; k.ostep = dil;
        if ((long long)d->B * k.img_stride >= (1ll << 29)) return HV_ERR_UNSUPPORTED;
        k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * xs);
        {   // filters-in-LDS form: residue sub-grids of at least a tile, or whole residue classes packed into a tile (conv_lfd_kernel)
            static const int lf8 = getenv("HV_LF_DIL8") ? atoi(getenv("HV_LF_DIL8")) : 1;      // A/B knob: 0 = round 3's rule (d <= 4 only)
            if (dil <= 4 || lf8) {
                const int rc = hv_convlf_launch(k, d->KH, d->KW, s);
                if (rc != HV_ERR_UNSUPPORTED) return rc;
            }
        }
        if (dil > 8) return HV_ERR_UNSUPPORTED;      // (d = 16 elsewhere: the gather kernel, as before)
        return hv_halo2_launch(k, TW, d->KH, d->KW, maxpatch, s);      // (conv_halo_kernel has no pixel step)
    }
    static const bool halo2 = !(getenv("HV_HALO2") && atoi(getenv("HV_HALO2")) == 0);   // A/B knob
    {   // filters-in-LDS form (3x3 stride-1 layers with whole-chunk channel counts)
        const int rc = hv_convlf_launch(k, d->KH, d->KW, s);
        if (rc != HV_ERR_UNSUPPORTED || d->pool2 || d->x1) return rc;
    }
    if (halo2) {   // weights-in-registers form where an instantiation exists
        const int rc = hv_halo2_launch(k, TW, d->KH, d->KW, maxpatch, s);
        if (rc != HV_ERR_UNSUPPORTED) return rc;
    }
    if ((d->Cin & 15) || d->KH * d->KW > 16) return HV_ERR_UNSUPPORTED;   // conv_halo_kernel: whole 16-channel chunks, 16-entry tap table
    if (k.bstep == 2) return ck32 ? dispatch_halo_bn<8, 16, 32, true>(k, maxpatch, s) : dispatch_halo_bn<8, 16, 16, true>(k, maxpatch, s);
    if (small_tile) return ck32 ? dispatch_halo_bn<8, 16, 32, false>(k, maxpatch, s) : dispatch_halo_bn<8, 16, 16, false>(k, maxpatch, s);
    return ck32 ? dispatch_halo_bn<8, 32, 32, false>(k, maxpatch, s) : dispatch_halo_bn<8, 32, 16, false>(k, maxpatch, s);
}
