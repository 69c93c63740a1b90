// Halo-tiled weight gradient (fp16 MFMA operands, fp32 accumulate) for stride-1, dilation-1, 3x3 / 4x4 / 5x5 convolutions
// with at most 16 output channels.
//
//   dW[co][(r,q)][ci] = sum_{n,oy,ox} g[n,oy,ox,co] * x[n, oy-pad+r, ox-pad+q, ci]
//
// The gather kernel (conv_igemm.hip: wgrad_kernel) re-reads x from L2 once per tap and g once per column tile.  Here a
// workgroup walks output tiles of 8 rows x 32 pixels; per tile it stages g (8x32 pixels x BN channels) and the
// (8+k-1) x (32+k-1) input patch (x BC channels) ONCE, transposed to [channel][row][pixel] fp16 in LDS so that the
// contraction index (pixels along W) is contiguous for v_mfma_f32_16x16x32_f16.  A tap (r,q) is a shifted window of the
// patch: row shift r is an address offset; column shift q is taken out of a 16-pixel aligned window in registers
// (dword select for even q, v_alignbit for odd q) -- no misaligned LDS reads, no per-tap reloads.  The taps are dealt
// round-robin to the four waves; one slab per workgroup goes to the deterministic slab reduction.
#include <stdlib.h>

#include "hv_common.h"

struct WHaloK {
    const _Float16* x; const _Float16* g; float* slabs;       // fp16 storage (the fp16 mode's activations and gradients)
    int B, Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Ho, Wo, g_ld, g_coff, Cout, pad;
    int tiles_x, tiles_per_img, ntiles;
    long long slab;   // floats per slab = Cout * KS*KS * Cin
    unsigned x_bytes, g_bytes;   // buffer descriptor ranges
    float* bias_out;             // per-workgroup column sums of g (bias gradient), [gridDim.x][Cout], or NULL
    int gx;                      // x extent of the main grid (pixel chunks = slabs); blocks beyond it run the carried fold
    HvFold fold;                 // the previous weight gradient's slab fold, carried along (hv_wgrad_desc.carry; splits == 0: none)
};

__device__ __forceinline__ uint32_t hv_pack2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)a, (_Float16)b};
    return *reinterpret_cast<uint32_t*>(&v);
}

template <int Q> __device__ __forceinline__ f16x8 window_frag(const uint32_t (&w)[8]) {
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (Q & 1) o[i] = __builtin_amdgcn_alignbit(w[(Q + 1) / 2 + i], w[(Q - 1) / 2 + i], 16);
        else o[i] = w[Q / 2 + i];
    }
    uint4 u = make_uint4(o[0], o[1], o[2], o[3]);
    return *reinterpret_cast<f16x8*>(&u);
}

// TS (tap split): the taps are dealt round-robin to the four waves, each wave walks all eight rows -- few accumulator registers
// (3-4 workgroups per CU) for 4x more LDS reads: wins where the tile count is small (PatchGAN logits layer); otherwise every
// wave owns two of the eight rows, keeps all taps' accumulators and the four waves are summed in LDS at the end.
template <int KS, int BN, int BC, bool TS>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const WHaloK p) {
    constexpr int TH = 8, TW = 32, PH = TH + KS - 1, PWP = 40;   // 40 halfs per patch row: 32 + k - 1 <= 40, 16 B aligned
    constexpr int NT = BN / 16, CT = BC / 16, TAPS = KS * KS;
    constexpr int GROW = TW + 8;                                  // padded g row (halfs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* Gt = reinterpret_cast<_Float16*>(smem);             // [BN][TH][GROW]
    _Float16* Xt = Gt + BN * TH * GROW;                          // [BC][PH][PWP]
    float* red = reinterpret_cast<float*>(smem);                  // [TAPS][BN][BC] after the tile loop (aliases Gt/Xt), !TS

    // the wave index steers MFMA-only branches: it must be a scalar (MFMA ignores EXEC, so an exec-masked "skipped" block
    // whose skip branch the compiler elides would still accumulate)
    if ((int)blockIdx.x >= p.gx) {      // (block-uniform) the carried fold's workgroups
        const int fx = (int)gridDim.x - p.gx;
        hv_fold_blocks(p.fold, ((int)blockIdx.x - p.gx) + fx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z), fx * (int)gridDim.y * (int)gridDim.z, smem);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BN, ci0 = blockIdx.z * BC;
    // taps are dealt round-robin to the four waves (tap t belongs to wave t & 3, accumulator slot t >> 2): a wave keeps
    // only TAPS/4 accumulator tiles (two or three workgroups fit a CU instead of one) and owns its taps' results outright
    constexpr int SLOTS = TS ? (TAPS + 3) / 4 : TAPS;
    f32x4 acc[SLOTS][NT][CT];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging is software-pipelined through registers: the loads of tile i+1 are issued (branch-free buffer loads, lanes
    // outside the image / channel range carry an out-of-range offset and read zeros) before the MFMAs of tile i, so the
    // one workgroup a CU can hold (100-130 accumulator registers per lane) does not sit idle for an HBM round trip per tile.
    constexpr int GU = (BN / 4) * TH * 4, XU = (BC / 4) * PH * 5;     // staging units of 8 pixels x 4 channels
    constexpr int GPT = (GU + 255) / 256, XPT = (XU + 255) / 256;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    typedef HvSt<true> SS;
    typedef SS::R SR;                            // 4 consecutive channels of one pixel: 8 bytes
    SR rg[GPT][8], rx[XPT][8];
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        const int n_img = tile / p.tiles_per_img, tr = tile - n_img * p.tiles_per_img;
        const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int u = tid + i * 256;
            const int cg = u % (BN / 4), rr = u / (BN / 4), run = rr & 3, ty = rr >> 2;
            const int oy = oy0 + ty, co = co0 + cg * 4, ox = ox0 + run * 8;
            const bool rok = u < GU && oy < p.Ho && co < p.Cout;
            const unsigned base = (unsigned)(((n_img * p.Ho + oy) * p.Wo + ox) * p.g_ld + p.g_coff + co) * SS::B;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                rg[i][e] = SS::ld(gsrc, (rok && ox + e < p.Wo) ? base + (unsigned)(e * p.g_ld) * SS::B : 0x80000000u, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * 256;
            const int cg = u % (BC / 4), rr = u / (BC / 4), run = rr % 5, py = rr / 5;
            const int hi = oy0 - p.pad + py, ci = ci0 + cg * 4, wi0 = ox0 - p.pad + run * 8;
            const bool rok = u < XU && (unsigned)hi < (unsigned)p.Hl && ci < p.Cin;
            const unsigned base = (unsigned)(n_img * p.img_stride + p.x_coff + (hi >> p.in_shift) * p.Wp * p.x_ld + ci) * SS::B;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int wi = wi0 + e;
                rx[i][e] = SS::ld(xsrc, (rok && (unsigned)wi < (unsigned)p.Wl) ? base + (unsigned)((wi >> p.in_shift) * p.x_ld) * SS::B : 0x80000000u, 0);
            }
        }
    };
    // channel c of 8 consecutive pixels -> 8 halfs (the transpose to [channel][pixel]): 16-bit selects, no conversion
    auto pack4 = [&](const SR (&v)[8], int c) __attribute__((always_inline)) {
        auto h = [&](int e) -> uint32_t { const uint32_t w = (c & 2) ? v[e].y : v[e].x; return (c & 1) ? (w >> 16) : (w & 0xffffu); };
        return make_uint4(h(0) | (h(1) << 16), h(2) | (h(3) << 16), h(4) | (h(5) << 16), h(6) | (h(7) << 16));
    };
    const bool do_bias = p.bias_out != nullptr && blockIdx.z == 0;
    float bsum[GPT][4];
#pragma unroll
    for (int i = 0; i < GPT; ++i) bsum[i][0] = bsum[i][1] = bsum[i][2] = bsum[i][3] = 0.f;
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int u = tid + i * 256;
            if (u >= GU) continue;
            if (do_bias) {   // bias gradient: fp32 sums of the staged g values (zeros outside the image)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float4 q = SS::f4(rg[i][e]);
                    bsum[i][0] += q.x; bsum[i][1] += q.y; bsum[i][2] += q.z; bsum[i][3] += q.w;
                }
            }
            const int cg = u % (BN / 4), rr = u / (BN / 4), run = rr & 3, ty = rr >> 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(Gt + ((cg * 4 + c) * TH + ty) * GROW + run * 8) = pack4(rg[i], c);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * 256;
            if (u >= XU) continue;
            const int cg = u % (BC / 4), rr = u / (BC / 4), run = rr % 5, py = rr / 5;
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(Xt + ((cg * 4 + c) * PH + py) * PWP + run * 8) = pack4(rx[i], c);
        }
    };

    if ((int)blockIdx.x < p.ntiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += p.gx) {
        __syncthreads();   // previous tile's MFMA reads are done
        flush();
        __syncthreads();
        if (tile + p.gx < p.ntiles) prefetch(tile + p.gx);   // next tile's loads fly behind this tile's MFMAs
        if (TS) {
            // ---- MFMA: every wave walks all rows for its own taps
    #pragma unroll 1
            for (int ty = 0; ty < TH; ++ty) {
                f16x8 gf[NT];
    #pragma unroll
                for (int n = 0; n < NT; ++n) gf[n] = *reinterpret_cast<const f16x8*>(Gt + ((n * 16 + (lane & 15)) * TH + ty) * GROW + (lane >> 4) * 8);
    #pragma unroll
                for (int r = 0; r < KS; ++r) {
    #pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const _Float16* row = Xt + ((c * 16 + (lane & 15)) * PH + ty + r) * PWP + (lane >> 4) * 8;
                        uint32_t w[8];
                        *reinterpret_cast<uint4*>(&w[0]) = *reinterpret_cast<const uint4*>(row);
                        *reinterpret_cast<uint4*>(&w[4]) = *reinterpret_cast<const uint4*>(row + 8);
    #pragma unroll
                        for (int q = 0; q < KS; ++q) {
                            const int t = r * KS + q;
                            if ((t & 3) != wave) continue;     // wave-uniform
                            f16x8 xf;
                            if (q == 0) xf = window_frag<0>(w);
                            else if (q == 1) xf = window_frag<1>(w);
                            else if (q == 2) xf = window_frag<2>(w);
                            else if (q == 3) xf = window_frag<3>(w);
                            else xf = window_frag<4>(w);
    #pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[t >> 2][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gf[n], xf, acc[t >> 2][n][c], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
            // ---- MFMA: this wave's rows ty = wave, wave + 4
    #pragma unroll
            for (int rrow = 0; rrow < 2; ++rrow) {
                const int ty = wave + rrow * 4;
                f16x8 gf[NT];
    #pragma unroll
                for (int n = 0; n < NT; ++n) gf[n] = *reinterpret_cast<const f16x8*>(Gt + ((n * 16 + (lane & 15)) * TH + ty) * GROW + (lane >> 4) * 8);
    #pragma unroll
                for (int r = 0; r < KS; ++r) {
    #pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const _Float16* row = Xt + ((c * 16 + (lane & 15)) * PH + ty + r) * PWP + (lane >> 4) * 8;
                        uint32_t w[8];
                        *reinterpret_cast<uint4*>(&w[0]) = *reinterpret_cast<const uint4*>(row);
                        *reinterpret_cast<uint4*>(&w[4]) = *reinterpret_cast<const uint4*>(row + 8);
    #pragma unroll
                        for (int q = 0; q < KS; ++q) {
                            f16x8 xf;
                            if (q == 0) xf = window_frag<0>(w);
                            else if (q == 1) xf = window_frag<1>(w);
                            else if (q == 2) xf = window_frag<2>(w);
                            else if (q == 3) xf = window_frag<3>(w);
                            else xf = window_frag<4>(w);
    #pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[r * KS + q][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gf[n], xf, acc[r * KS + q][n][c], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if (do_bias) {   // fold the units of a channel group (fixed order) -> one row per workgroup
        __syncthreads();
        float* bsh = reinterpret_cast<float*>(smem);          // GU x 4 floats (the staging tiles are dead)
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int u = tid + i * 256;
            if (u < GU) { bsh[u * 4 + 0] = bsum[i][0]; bsh[u * 4 + 1] = bsum[i][1]; bsh[u * 4 + 2] = bsum[i][2]; bsh[u * 4 + 3] = bsum[i][3]; }
        }
        __syncthreads();
        if (tid < BN && co0 + tid < p.Cout) {
            const int cg = tid >> 2, c = tid & 3;
            float t = 0.f;
            for (int k = 0; k < TH * 4; ++k) t += bsh[(cg + k * (BN / 4)) * 4 + c];
            p.bias_out[(long long)blockIdx.x * p.Cout + co0 + tid] = t;
        }
        __syncthreads();
    }
    if (TS) {
        // ---- one slab per workgroup: every wave writes its own taps (D layout: row (= co) = (lane>>4)*4 + r, col (= ci) = lane & 15)
        float* out = p.slabs + (long long)blockIdx.x * p.slab;
    #pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int t = sl * 4 + wave;
            if (t >= TAPS) continue;
    #pragma unroll
            for (int n = 0; n < NT; ++n)
    #pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const int ci = ci0 + c * 16 + (lane & 15);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                        if (co < p.Cout && ci < p.Cin) out[((long long)co * TAPS + t) * p.Cin + ci] = acc[sl][n][c][r];
                    }
                }
        }
    } else {
        // ---- sum the four waves in LDS (fixed order), write one slab per workgroup
        __syncthreads();
        for (int wv = 0; wv < 4; ++wv) {
            if (wave == wv) {
    #pragma unroll
                for (int t = 0; t < TAPS; ++t)
    #pragma unroll
                    for (int n = 0; n < NT; ++n)
    #pragma unroll
                        for (int c = 0; c < CT; ++c)
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                // D layout: row (= co) = (lane>>4)*4 + r, col (= ci) = lane & 15
                                float* d = red + (t * BN + n * 16 + (lane >> 4) * 4 + r) * BC + c * 16 + (lane & 15);
                                *d = wv == 0 ? acc[t][n][c][r] : *d + acc[t][n][c][r];
                            }
            }
            __syncthreads();
        }
        float* out = p.slabs + (long long)blockIdx.x * p.slab;
        for (int e = tid; e < TAPS * BN * BC; e += 256) {
            const int ci = e % BC, r2 = e / BC, co = r2 % BN, t = r2 / BN;
            if (co0 + co < p.Cout && ci0 + ci < p.Cin) out[((long long)(co0 + co) * TAPS + t) * p.Cin + ci0 + ci] = red[e];
        }
    }
}

struct WHaloPlan { int BN, BC, gx; size_t lds; };
static bool wgrad_halo_plan(const hv_wgrad_desc* d, WHaloPlan* pl) {
    if (d->precision != HV_F16 || !d->x_f16 || !d->g_f16 || d->stride != 1 || d->dil != 1 || d->KH != d->KW || d->KH < 3 || d->KH > 5) return false;
    if (d->Wo != d->W + 2 * d->pad - d->KW + 1 || d->Ho != d->H + 2 * d->pad - d->KH + 1) return false;
    if (d->KH == 5 && (d->Cout > 16 || d->Cin > 16)) return false;      // 25 taps: 16x16 tiles only (register budget)
    if (d->Cout > 16) return false;   // measured: with more than one 16-wide co tile the gather kernel's larger MFMA tiles win
    pl->BN = (d->KH == 5 || d->Cout <= 16) ? 16 : 32;
    pl->BC = (d->KH == 5 || d->Cin <= 16) ? 16 : 32;   // k4: 16 taps x (16 x 32) = 128 accumulator registers
    const int PH = 8 + d->KH - 1;
    size_t stage = (size_t)(pl->BN * 8 * 40 + pl->BC * PH * 40) * 2, red = (size_t)d->KH * d->KW * pl->BN * pl->BC * 4;
    pl->lds = stage > red ? stage : red;
    const long long ntiles = (long long)d->B * hv_cdiv(d->Ho, 8) * hv_cdiv(d->Wo, 32);
    const long long pairs = (long long)hv_cdiv(d->Cout, pl->BN) * hv_cdiv(d->Cin, pl->BC);
    static const int gx_env = getenv("HV_WHALO_GX") ? atoi(getenv("HV_WHALO_GX")) : 0;
    // Workgroups per launch.  Round 1 (fp32 storage): one persistent workgroup per CU won on the 256x256 layers (53 / 55 / 35 us at 256 against 87 / 72 / 48 us
    // at 1024 -- fewer slabs, longer tile pipelines).  Re-measured at the end of round 3 (fp16 storage: half the bytes per tile, the loads of one workgroup no
    // longer keep a CU's memory pipe busy; tools/ab_env_stats.sh, us at 256 / 512 / 1024): 3x3 16<-16 26.1 / 19.1 / 18.5, 3x3 16<-32 46.3 / 34.2 / 42.9,
    // 4x4 stem 20.5 / 20.5 / 16.6, 5x5 45.8 / 52.6 / 65.8 -> per filter size
    const int gx_target = gx_env ? gx_env : (d->KH == 3 ? 512 : d->KH == 4 ? 1024 : 256);
    long long gx = gx_target / pairs;
    if (gx < 32) gx = 32;
    if (gx > ntiles) gx = ntiles;
    pl->gx = (int)gx;
    return true;
}

size_t hv_wgrad_halo_workspace_bytes(const hv_wgrad_desc* d) {
    WHaloPlan pl;
    if (!wgrad_halo_plan(d, &pl)) return 0;
    return (size_t)pl.gx * ((size_t)d->Cout * d->KH * d->KW * d->Cin + (d->dbias ? d->Cout : 0)) * sizeof(float);
}

template <int KS, int BN, int BC, bool TS = false>
static int launch_wh(const WHaloK& k, const WHaloPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_halo_kernel<KS, BN, BC, TS>;
    static int lds_limit = 48 * 1024;
    if ((int)pl.lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        lds_limit = 150 * 1024;
    }
    dim3 grid(pl.gx, hv_cdiv(d->Cout, BN), hv_cdiv(d->Cin, BC));
    WHaloK kk = k;      // the previous weight gradient's fold rides along as extra x-blocks (needs 4 KB of the launch's LDS)
    kk.gx = pl.gx;
    kk.fold.splits = 0;
    const int fx = pl.lds >= 4096 ? hv_carry_blocks((int)(grid.y * grid.z)) : 0;
    if (fx > 0) { kk.fold = hv_carry; hv_carry_taken = 1; grid.x += fx; }
    hv_path_note = 11;
    HV_KNAME("wgrad_halo_kernel<%d, %d, %d, %s>", KS, BN, BC, TS ? "true" : "false");
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, s, kk);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// returns HV_ERR_UNSUPPORTED when the shape does not qualify (caller falls back to the gather kernel);
// on success the slabs (pl.gx of them) are in d->workspace and *nslabs is set.
int hv_wgrad_halo(const hv_wgrad_desc* d, int* nslabs, hipStream_t s) {
    WHaloPlan pl;
    if (!wgrad_halo_plan(d, &pl)) return HV_ERR_UNSUPPORTED;
    const size_t need = hv_wgrad_halo_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need) return HV_ERR_WORKSPACE;
    WHaloK k;
    if (!d->x_f16 || !d->g_f16) return HV_ERR_UNSUPPORTED;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.g = reinterpret_cast<const _Float16*>(d->g); k.slabs = d->workspace;
    k.B = d->B; k.Hl = d->H; k.Wl = d->W; k.in_shift = d->in_shift; k.Wp = d->W >> d->in_shift;
    k.img_stride = (d->H >> d->in_shift) * k.Wp * d->x_ld; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho; k.Wo = d->Wo; k.g_ld = d->g_ld; k.g_coff = d->g_coff; k.Cout = d->Cout; k.pad = d->pad;
    k.tiles_x = hv_cdiv(d->Wo, 32); k.tiles_per_img = k.tiles_x * hv_cdiv(d->Ho, 8); k.ntiles = k.tiles_per_img * d->B;
    k.slab = (long long)d->Cout * d->KH * d->KW * d->Cin;
    k.bias_out = d->dbias ? d->workspace + (long long)pl.gx * k.slab : nullptr;
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * sizeof(_Float16));
    k.g_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->g_ld * sizeof(_Float16));
    *nslabs = pl.gx;
    if (d->KH == 5) return launch_wh<5, 16, 16>(k, pl, d, s);
    if (d->KH == 4) return pl.BC == 16 ? launch_wh<4, 16, 16, true>(k, pl, d, s) : launch_wh<4, 16, 32, true>(k, pl, d, s);
    if (pl.BN == 16 && pl.BC == 16) return launch_wh<3, 16, 16>(k, pl, d, s);
    if (pl.BN == 16) return launch_wh<3, 16, 32>(k, pl, d, s);
    if (pl.BC == 16) return launch_wh<3, 32, 16>(k, pl, d, s);
    return launch_wh<3, 32, 32>(k, pl, d, s);
}
