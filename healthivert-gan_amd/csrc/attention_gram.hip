// Contextual attention, matching scores and their gradient on the PIXEL Gram matrix (reference models/inpaint_networks.py:327-344 + autograd).
//
// The reference convolves the (zero-padded) downsampled map fd with its own normalised 3x3 patches: S0[p][l] = rnorm[l] * <patch_l, patch_p>, a
// K = 9 C contraction per score (19.3 GFLOP per step at bs 16; patch tables wp / wp_h / wpT of 57-75 MB written and re-read per forward / backward).
// The patches are both filters and inputs, so with the pixel Gram matrix G[a][b] = <fd[a], fd[b]> (K = C)
//     T[p][l]  = sum over the 3x3 offsets t of G[p + t][l + t]      (terms with p + t or l + t outside the map dropped)
//     S0[p][l] = rnorm[l] * T[p][l],      norm[l]^2 = sum_t |fd[l + t]|^2                      (tests/test_host_cpu.py: identity against the oracle)
// and, for the gradient, with Gs[i][j] = dS0[j][i] rnorm[i] + dS0[i][j] rnorm[j] (hv_ca_score_backward_prep) and E = the same box filter applied to Gs,
//     d fd[a] = sum_b E[a][b] fd[b] + (sum_t coef[a - t]) fd[a]
// -- the L x 9C gradient GEMM, the transpose of wp and the col2im pass collapse into one K = L product with N = C.
// Blocks are grid rows: p in row py, l in row ly is a w x w block whose box filter needs the three Gram blocks (py + ty, ly + ty) on its diagonal;
// the x shifts stay inside the block (a shifted pixel outside [0, w) is outside the map).  w = 32 or 64, C = 64, fp16 operands, fp32 accumulation.
#include "hv_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- downsample (nearest, even positions) -> fd_h [B][L][C] fp16, fdT_h [B][C][L] fp16, q[B][L] = |fd|^2 (fp32) ----------------------------------------
template <int C>
__global__ __launch_bounds__(256) void ca_gram_down_kernel(const _Float16* __restrict__ f, int H, int W, int f_ld, _Float16* __restrict__ fd_h,
                                                           _Float16* __restrict__ fdT_h, float* __restrict__ q) {
    constexpr int PC = C / 8;                       // 16-byte pieces per pixel
    __shared__ _Float16 t[C][64 + 8];               // [channel][pixel of the row segment]
    const int h = H / 2, w = W / 2, L = h * w;
    const int segs = (w + 63) / 64;
    const int seg = blockIdx.x % segs, y = (blockIdx.x / segs) % h, b = blockIdx.x / (segs * h);
    const int x0 = seg * 64, nx = min(64, w - x0);
    for (int it = threadIdx.x; it < nx * PC; it += 256) {
        const int xl = it / PC, pc = it - xl * PC, x = x0 + xl;
        const f16x8 v = *reinterpret_cast<const f16x8*>(f + (((long long)b * H + 2 * y) * W + 2 * x) * f_ld + pc * 8);
        const long long l = (long long)y * w + x;
        *reinterpret_cast<f16x8*>(fd_h + ((long long)b * L + l) * C + pc * 8) = v;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float a = (float)v[e]; s += a * a; t[pc * 8 + e][xl] = v[e]; }
        // the PC lanes of a pixel are consecutive lanes of one wave (256 % PC == 0, PC a power of two)
#pragma unroll
        for (int o = 1; o < PC; o <<= 1) s += __shfl_xor(s, o, 64);
        if (pc == 0) q[(long long)b * L + l] = s;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < C * (nx / 8); it += 256) {       // rows of 8 pixels = 16 bytes (w is a multiple of 8)
        const int c = it / (nx / 8), x8 = (it - c * (nx / 8)) * 8;
        *reinterpret_cast<f16x8*>(fdT_h + ((long long)b * C + c) * L + (long long)y * w + x0 + x8) = *reinterpret_cast<const f16x8*>(&t[c][x8]);
    }
}

extern "C" int hv_ca_gram_down(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, void* fd_h, void* fdT_h, float* q, void* stream) {
    if (!f || !fd_h || !fdT_h || !q || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return HV_ERR_ARG;
    if (!f_f16 || C != 64 || (f_ld & 7) || ((uintptr_t)f & 15) || ((W / 2) & 7)) return HV_ERR_UNSUPPORTED;
    const int h = H / 2, w = W / 2;
    hipLaunchKernelGGL(ca_gram_down_kernel<64>, dim3(B * h * ((w + 63) / 64)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(f), H, W, f_ld,
                       reinterpret_cast<_Float16*>(fd_h), reinterpret_cast<_Float16*>(fdT_h), q);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// The 3x3 box filter on the diagonal is separable: T = box_x(sum over ty of the three diagonal blocks).  The ty sum is taken where the blocks are produced
// (in the MFMA accumulators of the scores kernel, in the loading registers of the gradient kernel), so one w x w block sits in LDS and an output is three
// reads along its diagonal: out[r][c] = G[r - 1][c - 1] + G[r][c] + G[r + 1][c + 1] (terms outside [0, w) dropped: they are outside the map)
template <int BW>
__device__ __forceinline__ float gram_box(const float (*G)[BW + 1], int r, int c) {
    float acc = G[r][c];
    if (r >= 1 && c >= 1) acc += G[r - 1][c - 1];
    if (r + 1 < BW && c + 1 < BW) acc += G[r + 1][c + 1];
    return acc;
}

// norm[l] = max(sqrt(3x3 box of the pixels' squared norms), 1e-4), rnorm = 1 / norm  (= hv_ca_patches' outputs)
__global__ __launch_bounds__(256) void ca_gram_norm_kernel(const float* __restrict__ q, int h, int w, long long n, float* __restrict__ norm, float* __restrict__ rnorm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int L = h * w, l = (int)(i % L), y = l / w, x = l - y * w;
    const float* qb = q + (i - l);
    float n2 = 0.f;
#pragma unroll
    for (int ty = -1; ty <= 1; ++ty)
#pragma unroll
        for (int tx = -1; tx <= 1; ++tx)
            if ((unsigned)(y + ty) < (unsigned)h && (unsigned)(x + tx) < (unsigned)w) n2 += qb[(y + ty) * w + x + tx];
    const float nv = fmaxf(sqrtf(n2), 1e-4f);
    norm[i] = nv;
    rnorm[i] = 1.f / nv;
}

// ---- scores: one workgroup per (sample, grid row py of p, group of LG grid rows of l); the block's three Gram products are summed in the accumulators.
// Every operand fragment of a round is requested before its first MFMA (branch-free: a row outside the map is an out-of-range buffer offset = zeros), the p
// side's fragments once for the whole group -- the first version fetched per ty inside scalar branches, three dependent L2 round trips per block: 162 us.
#define GRAM_OOB 0x80000000u
// NLB grid rows of l per round: a row of S0 then leaves as NLB * w * 4 bytes contiguous per store instruction (128-byte segments 4 KB apart held the kernel at
// 1.2 TB/s: 52-62 us whatever the round structure)
template <int BW, int NLB>
__global__ __launch_bounds__(256) void ca_gram_scores_kernel(const _Float16* __restrict__ fd_h, const float* __restrict__ rnorm, int h, int lg, float* __restrict__ S0, int diag) {
    constexpr int C = 64, NT = BW / 16, KS = C / 32, SW = NLB * BW;      // SW: strip width (columns per round)
    constexpr int TILES = NT * NT * NLB, TPW = TILES / 4;                  // MFMA tiles per wave and round: tile t -> (block j, mi, ni)
    __shared__ float Gt[2][NLB][BW][BW + 1];
    const int w = BW, L = h * w, ngr = (h + lg - 1) / lg;
    int id = blockIdx.x;
    const int lgi = id % ngr; id /= ngr;
    const int py = id % h;
    const long long b = id / h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t fsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(fd_h + b * (long long)L * C), 0, (unsigned)(L * C * 2), 0x00020000);
    const float* rb = rnorm + b * L;
    float* Sb = S0 + b * (long long)L * L;
    const unsigned lane_off = (unsigned)(((lane & 15) * C + 8 * (lane >> 4)) * 2);
    auto frag = [&](int row, int t16, int ks) __attribute__((always_inline)) {      // 16 pixels t16 of grid row `row`, channels 32 ks + 8 (lane >> 4) ..
        const unsigned off = (unsigned)row < (unsigned)h ? (unsigned)((row * w + t16 * 16) * C + ks * 32) * 2u + lane_off : GRAM_OOB;
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(fsrc, off, 0, 0));
    };
    // wave -> its TPW tiles.  NLB == 4: wave j owns block j of the round (all NT x NT tiles); NLB == 1: a wave owns one tile row mi (NT tiles).  Either way the
    // wave's tiles span `NMI` tile rows and `NNI` tile columns of ONE block: its p-side fragments (3 x NMI x KS) are fetched once for the whole row group,
    // the l-side fragments (3 x NNI x KS) once per round -- with a fetch per tile the kernel moved 786 MB of fragments through L2 (14 TB/s: 54 us)
    constexpr int NMI = NLB == 4 ? NT : 1, NNI = TPW / NMI;
    static_assert(TPW == NMI * NNI && (NLB == 4 ? TPW == NT * NT : NNI == NT), "tile ownership");
    const int jw = NLB == 4 ? wave : 0, mi0 = NLB == 4 ? 0 : wave % NT;
    f16x8 afr[3][NMI][KS];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int m = 0; m < NMI; ++m)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) afr[ty][m][ks] = frag(py + ty - 1, mi0 + m, ks);
    // rounds: wave jw walks CONSECUTIVE grid rows ly = lw0, lw0 + 1, .. (NLB == 4: the four waves own four runs of lg / 4 rows; NLB == 1: all waves the same
    // run), so two of a block's three l-side rows are the previous round's: one row of fragments (NNI x KS loads) is fetched per round and the other two
    // rotate through registers -- a third of the l-side traffic again (fragments per tile: 54 us; per wave and round: 37 us)
    const int ly0 = lgi * lg, ly1 = min(h, (lgi + 1) * lg);
    const int rounds = NLB == 4 ? lg / 4 : lg, lw0 = ly0 + jw * rounds;
    f16x8 brow[4][NNI][KS];      // rows ly - 1, ly, ly + 1 of the round + the next round's new row, requested a round ahead
    auto load_row = [&](int slot, int row) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NNI; ++n)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) brow[slot][n][ks] = frag(row, n, ks);
    };
    load_row(0, lw0 - 1);
    load_row(1, lw0);
    load_row(2, lw0 + 1);
    int buf = 0;
    for (int rd = 0; rd < rounds; ++rd, buf ^= 1) {
        const int ly = lw0 + rd;                 // this wave's grid row of l in this round
        load_row(3, rd + 1 < rounds ? ly + 2 : -1);
        if (!(diag & 2)) {
#pragma unroll
            for (int m = 0; m < NMI; ++m)
#pragma unroll
                for (int n = 0; n < NNI; ++n) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)      // (a block outside the map contributes zeros: one of its operands is)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[ty][m][ks], brow[ty][n][ks], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Gt[buf][jw][(mi0 + m) * 16 + 4 * (lane >> 4) + r][n * 16 + (lane & 15)] = acc[r];
                }
        }
#pragma unroll
        for (int n = 0; n < NNI; ++n)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { brow[0][n][ks] = brow[1][n][ks]; brow[1][n][ks] = brow[2][n][ks]; brow[2][n][ks] = brow[3][n][ks]; }
        // items of the box / store pass: (row r, block j, 4 columns): block j's columns start at grid row lyj = ly0 + j * rounds + rd
        constexpr int NIT = BW * SW / 4 / 256;
        float4 rn[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = threadIdx.x + k * 256, c = (it % (SW / 4)) * 4, j = c / BW, lyj = ly0 + (NLB == 4 ? j * rounds : 0) + rd;
            rn[k] = lyj < ly1 ? *reinterpret_cast<const float4*>(rb + lyj * w + (c - j * BW)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();      // (two buffers: the next round's stores go to the other one, and the round after that is behind the next barrier)
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = threadIdx.x + k * 256, r = it / (SW / 4), c = (it - r * (SW / 4)) * 4, j = c / BW, c0 = c - j * BW;
            const int lyj = ly0 + (NLB == 4 ? j * rounds : 0) + rd;
            if (lyj < ly1 && !(diag & 1))
                *reinterpret_cast<float4*>(Sb + (long long)(py * w + r) * L + lyj * w + c0) =
                    make_float4(gram_box<BW>(Gt[buf][j], r, c0) * rn[k].x, gram_box<BW>(Gt[buf][j], r, c0 + 1) * rn[k].y, gram_box<BW>(Gt[buf][j], r, c0 + 2) * rn[k].z,
                                gram_box<BW>(Gt[buf][j], r, c0 + 3) * rn[k].w);
        }
    }
}

extern "C" int hv_ca_gram_scores(const void* fd_h, const float* q, int B, int h, int w, int C, float* S0, float* norm, float* rnorm, void* stream) {
    if (!fd_h || !q || !S0 || !norm || !rnorm || B <= 0 || h <= 0 || w <= 0) return HV_ERR_ARG;
    if (C != 64 || (w != 32 && w != 64) || ((uintptr_t)fd_h & 15) || ((uintptr_t)S0 & 15) || ((uintptr_t)rnorm & 15) || (long long)B * h * h >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    const long long n = (long long)B * h * w;
    hipLaunchKernelGGL(ca_gram_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, h, w, n, norm, rnorm);
    HV_LAUNCH_CHECK();
    static const int lg_env = getenv("HV_CA_GRAM_LG") ? atoi(getenv("HV_CA_GRAM_LG")) : 0;      // tuning knob: grid rows of l per workgroup
    static const int diag = getenv("HV_GRAM_DIAG") ? atoi(getenv("HV_GRAM_DIAG")) : 0;      // timing-only: 1 no stores, 2 no operand loads / MFMAs
    int lg = lg_env > 0 ? lg_env : 16;
    while (lg > 4 && (long long)B * h * ((h + lg - 1) / lg) < 1024) lg >>= 1;      // keep >= 4 workgroups per CU in flight
    lg = (lg + 3) / 4 * 4;
    const dim3 grid((unsigned)(B * h * ((h + lg - 1) / lg)));
    if (w == 32) hipLaunchKernelGGL((ca_gram_scores_kernel<32, 4>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(fd_h), rnorm, h, lg, S0, diag);
    else hipLaunchKernelGGL((ca_gram_scores_kernel<64, 1>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(fd_h), rnorm, h, lg, S0, diag);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- gradient: one workgroup per (sample, grid row ay); d fd[row ay] = sum over by of E(ay, by) fd[row by]  (+ the norm term), added to the even positions of df.
// NBY grid rows by per round (one K = NBY * w product per round: half the barriers per block at NBY = 2; NBY = 4, and Gs stored as 4-KB blocks instead of
// 128-byte rows 4 KB apart, measured the same step time in round 5 -- the kernel runs beside the weight-gradient stream and is not what the step waits for).
template <int BW, int NBY>
__global__ __launch_bounds__(256) void ca_gram_backward_kernel(const float* __restrict__ Gs, const _Float16* __restrict__ fd_h, const _Float16* __restrict__ fdT_h,
                                                               const float* __restrict__ coef, int h, float* __restrict__ df, int df_ld, int xcd_rows) {
    constexpr int C = 64, MT = BW / 16, NTL = C / 16, TPW = MT * NTL / 4, KW = NBY * BW, LDE = KW + 8;
    constexpr int NV = BW * BW / 4 / 256;                // float4 items of ONE block per thread; the three diagonal blocks are summed item by item as they arrive
    __shared__ float Gt[NBY][BW][BW + 1];
    __shared__ __attribute__((aligned(16))) _Float16 Eh[BW][LDE];
    const int w = BW, L = h * w, H = 2 * h, W = 2 * w;
    // xcd_rows (h a multiple of 8): the workgroups of grid rows ay - 1, ay, ay + 1 read the same blocks of Gs; an XCD (linear id & 7) gets h / 8 consecutive
    // grid rows of every sample, so a block crosses the fabric 1 + 2 / (h / 8) times instead of three
    int ay;
    long long b;
    if (xcd_rows) {
        const int per = h >> 3, q = (int)blockIdx.x >> 3;
        ay = ((int)blockIdx.x & 7) * per + q % per;
        b = q / per;
    } else {
        ay = blockIdx.x % h;
        b = blockIdx.x / h;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Gb = Gs + b * (long long)L * L;
    const _Float16* fTb = fdT_h + b * (long long)C * L;
    f32x4 acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 pre[NBY][3][NV];
    auto fetch = [&](int by0) __attribute__((always_inline)) {       // the three diagonal blocks (ay + ty, by + ty) of Gs per by, zeros where a block leaves the map
#pragma unroll
        for (int j = 0; j < NBY; ++j)
#pragma unroll
            for (int tyi = 0; tyi < 3; ++tyi) {
                const int ra = ay + tyi - 1, rb = by0 + j + tyi - 1;
                const bool ok = (unsigned)ra < (unsigned)h && (unsigned)rb < (unsigned)h;       // scalar
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int it = threadIdx.x + k * 256, r = it / (BW / 4), c4 = (it - r * (BW / 4)) * 4;
                    pre[j][tyi][k] = ok ? *reinterpret_cast<const float4*>(Gb + (long long)(ra * w + r) * L + rb * w + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
    };
    fetch(0);
    for (int by0 = 0; by0 < h; by0 += NBY) {
#pragma unroll
        for (int j = 0; j < NBY; ++j)
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int it = threadIdx.x + k * 256, r = it / (BW / 4), c4 = (it - r * (BW / 4)) * 4;
                Gt[j][r][c4] = (pre[j][0][k].x + pre[j][1][k].x) + pre[j][2][k].x; Gt[j][r][c4 + 1] = (pre[j][0][k].y + pre[j][1][k].y) + pre[j][2][k].y;
                Gt[j][r][c4 + 2] = (pre[j][0][k].z + pre[j][1][k].z) + pre[j][2][k].z; Gt[j][r][c4 + 3] = (pre[j][0][k].w + pre[j][1][k].w) + pre[j][2][k].w;
            }
        // this round's B operands (fd of grid rows by0 .., pixel-contiguous: the transposed table), requested before the barrier
        f16x8 bf[TPW][KW / 32];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int ni = (wave * TPW + u) % NTL;
#pragma unroll
            for (int ks = 0; ks < KW / 32; ++ks)
                bf[u][ks] = *reinterpret_cast<const f16x8*>(fTb + (long long)(ni * 16 + (lane & 15)) * L + by0 * w + ks * 32 + 8 * (lane >> 4));
        }
        __syncthreads();
        if (by0 + NBY < h) fetch(by0 + NBY);        // the next blocks fly behind the box filter and the MFMAs
#pragma unroll
        for (int j = 0; j < NBY; ++j)
            for (int it = threadIdx.x; it < BW * BW / 4; it += 256) {
                const int r = it / (BW / 4), c0 = (it - r * (BW / 4)) * 4;
                f16x4 e4;
#pragma unroll
                for (int u = 0; u < 4; ++u) e4[u] = (_Float16)gram_box<BW>(Gt[j], r, c0 + u);
                *reinterpret_cast<f16x4*>(&Eh[r][j * BW + c0]) = e4;
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int mi = (wave * TPW + u) / NTL;
#pragma unroll
            for (int ks = 0; ks < KW / 32; ++ks) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(&Eh[mi * 16 + (lane & 15)][ks * 32 + 8 * (lane >> 4)]);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[u][ks], acc[u], 0, 0, 0);
            }
        }
        __syncthreads();          // Eh and Gt are rewritten by the next round
    }
    // epilogue: + (sum of coef over the 3x3 neighbourhood) * fd[a]; the even positions of the full-resolution gradient map receive the sum
    const float* cb = coef + b * L;
    const _Float16* fb = fd_h + b * (long long)L * C;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tile = wave * TPW + u, mi = tile / NTL, ni = tile - mi * NTL;
        const int ch = ni * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ax = mi * 16 + 4 * (lane >> 4) + r;
            float cs = 0.f;
#pragma unroll
            for (int ty = -1; ty <= 1; ++ty)
#pragma unroll
                for (int tx = -1; tx <= 1; ++tx)
                    if ((unsigned)(ay - ty) < (unsigned)h && (unsigned)(ax - tx) < (unsigned)w) cs += cb[(ay - ty) * w + ax - tx];
            const float v = acc[u][r] + cs * (float)fb[(long long)(ay * w + ax) * C + ch];
            float* d = df + (((long long)b * H + 2 * ay) * W + 2 * ax) * df_ld + ch;
            *d += v;
        }
    }
}

extern "C" int hv_ca_gram_backward(const float* Gs, const void* fd_h, const void* fdT_h, const float* coef, int B, int h, int w, int C, float* df, int df_ld,
                                   void* stream) {
    if (!Gs || !fd_h || !fdT_h || !coef || !df || B <= 0 || h <= 0 || w <= 0 || df_ld < C) return HV_ERR_ARG;
    if (C != 64 || (w != 32 && w != 64) || ((uintptr_t)Gs & 15) || ((uintptr_t)fdT_h & 15) || (long long)B * h >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(B * h));
    static const int nby = getenv("HV_CA_GRAM_NBY") ? atoi(getenv("HV_CA_GRAM_NBY")) : 2;      // A/B knob
    static const int xcd_env = getenv("HV_CA_GRAM_BWD_XCD") ? atoi(getenv("HV_CA_GRAM_BWD_XCD")) : 1;      // A/B knob (same bits either way)
    const int xcd_rows = xcd_env && !(h & 7);
    if (w == 32 && nby == 2 && !(h & 1))
        hipLaunchKernelGGL((ca_gram_backward_kernel<32, 2>), grid, dim3(256), 0, (hipStream_t)stream, Gs, reinterpret_cast<const _Float16*>(fd_h),
                           reinterpret_cast<const _Float16*>(fdT_h), coef, h, df, df_ld, xcd_rows);
    else if (w == 32)
        hipLaunchKernelGGL((ca_gram_backward_kernel<32, 1>), grid, dim3(256), 0, (hipStream_t)stream, Gs, reinterpret_cast<const _Float16*>(fd_h),
                           reinterpret_cast<const _Float16*>(fdT_h), coef, h, df, df_ld, xcd_rows);
    else
        hipLaunchKernelGGL((ca_gram_backward_kernel<64, 1>), grid, dim3(256), 0, (hipStream_t)stream, Gs, reinterpret_cast<const _Float16*>(fd_h),
                           reinterpret_cast<const _Float16*>(fdT_h), coef, h, df, df_ld, xcd_rows);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
