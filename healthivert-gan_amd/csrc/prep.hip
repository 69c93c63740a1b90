// Weight preparation (spectral norm + layout), its backward, activation/bias gradient, Adam, fills.
// All HBM-bound or latency-bound; the spectral-norm kernels are batched over layers (one workgroup per
// layer) so a whole generator needs one launch instead of 47.
#include <stdlib.h>

#include "hv_common.h"

#define SN_EPS 1e-12f
#define PREP_THREADS 512
#define PREP_MAXK 4608   // v vector staged in LDS
#define PREP_MAXCO 1024

// logical weight W(co, ci, tap) in the torch source layout
__device__ __forceinline__ long long wsrc_index(const hv_wprep_layer& L, int co, int ci, int tap) {
    return L.transposed_src ? ((long long)ci * L.Cout + co) * L.taps + tap : ((long long)co * L.Cin + ci) * L.taps + tap;
}

__global__ __launch_bounds__(PREP_THREADS) void weight_prep_kernel(const hv_wprep_layer* __restrict__ layers) {
    const hv_wprep_layer L = layers[blockIdx.x];
    __shared__ float v_s[PREP_MAXK];
    __shared__ float u_s[PREP_MAXCO];
    __shared__ float red[20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = PREP_THREADS / 64;
    const int K = L.Cin * L.taps;
    float sigma = 1.f;
    if (L.sn) {
        const float* W = L.w_orig;  // [Cout][K]
        for (int i = tid; i < L.Cout; i += PREP_THREADS) u_s[i] = L.u[i];
        for (int i = tid; i < K; i += PREP_THREADS) v_s[i] = L.v[i];
        __syncthreads();
        if (L.power_iter) {
            // v = normalize(W^T u)
            float ss = 0.f;
            for (int k = tid; k < K; k += PREP_THREADS) {
                float t = 0.f;
#pragma unroll 32
                for (int co = 0; co < L.Cout; ++co) t += W[(long long)co * K + k] * u_s[co];      // (same summation order; the loads run ahead: 32 in flight -- this kernel is a chain of memory round trips on the step's tail)
                v_s[k] = t;
                ss += t * t;
            }
            ss = hv_block_sum(ss, red);
            const float nv = fmaxf(sqrtf(ss), SN_EPS);
            __syncthreads();
            for (int k = tid; k < K; k += PREP_THREADS) v_s[k] = v_s[k] / nv;
            __syncthreads();
        }
        // s = W v  (one wave per row)
        float ss = 0.f;
        for (int co0 = wave; co0 < L.Cout; co0 += 4 * nwave) {      // four rows of a wave at a time: their loads are all in flight before the first reduction (same sums per row, same order of ss)
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + r * nwave;
                if (co < L.Cout) {
#pragma unroll 16
                    for (int k = lane; k < K; k += 64) t[r] += W[(long long)co * K + k] * v_s[k];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + r * nwave;
                if (co >= L.Cout) break;
                const float tr = hv_wave_sum(t[r]);
                if (lane == 0) {
                    if (L.power_iter) {
                        ss += tr * tr;
                        u_s[co] = tr;  // raw s = W v, normalised below
                    } else {
                        ss += u_s[co] * tr;  // sigma = u . (W v) with the stored u
                    }
                }
            }
        }
        ss = hv_block_sum(ss, red);
        if (L.power_iter) {
            const float nu = fmaxf(sqrtf(ss), SN_EPS);
            __syncthreads();
            float sg = 0.f;
            for (int i = tid; i < L.Cout; i += PREP_THREADS) {
                const float s = u_s[i], un = s / nu;
                u_s[i] = un;
                sg += un * s;
            }
            sigma = hv_block_sum(sg, red);
            for (int i = tid; i < L.Cout; i += PREP_THREADS) L.u[i] = u_s[i];
            for (int i = tid; i < K; i += PREP_THREADS) L.v[i] = v_s[i];
        } else {
            sigma = ss;
        }
    }
    if (tid == 0 && L.sigma) L.sigma[0] = sigma;
}

// phase 2, grid-parallel over the layers' work items: W/sigma written in the kernels' layouts (rows/channels beyond the real extent are zero)
// half index of (row, tap, k) in the MFMA-fragment order documented at hv_weight_tile_f16 (include/hvgan.h); T = fragment width (32 or 16)
static __device__ __forceinline__ long long tiled_index(int row, int tap, int k, int taps, int K, int T) {
    const int q = T >> 2, kk = k % T;
    return (long long)(row >> 4) * 16 * taps * K + (long long)((tap * K + k) / T) * 16 * T + ((kk / q) * 16 + (row & 15)) * q + kk % q;
}
static __host__ __device__ __forceinline__ int tile_width(int K) { return (K % 32 == 0) ? 32 : (K % 16 == 0) ? 16 : 0; }

extern "C" size_t hv_weight_tiled_elems(int rows, int taps, int K) {
    if (rows <= 0 || taps <= 0 || K <= 0 || !tile_width(K)) return 0;
    return (size_t)((rows + 15) / 16 * 16) * taps * K;
}

__global__ __launch_bounds__(256) void weight_tile_kernel(const _Float16* __restrict__ w, _Float16* __restrict__ wt, int rows, int taps, int K, int T) {
    const long long n = (long long)((rows + 15) / 16 * 16) * taps * K;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // index into the padded plain table
    if (i >= n) return;
    const int k = (int)(i % K);
    const long long r = i / K;
    const int tap = (int)(r % taps), row = (int)(r / taps);
    wt[tiled_index(row, tap, k, taps, K, T)] = row < rows ? w[i] : (_Float16)0.f;
}

extern "C" int hv_weight_tile_f16(const void* w_f16, void* w_tiled, int rows, int taps, int K, void* stream) {
    if (!w_f16 || !w_tiled || rows <= 0 || taps <= 0 || K <= 0) return HV_ERR_ARG;
    const int T = tile_width(K);
    if (!T) return HV_ERR_UNSUPPORTED;
    const long long n = (long long)((rows + 15) / 16 * 16) * taps * K;
    hipLaunchKernelGGL(weight_tile_kernel, dim3(hv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(w_f16),
                       reinterpret_cast<_Float16*>(w_tiled), rows, taps, K, T);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// Forward tables.  A work item = (filter co, chunk of input channels): its source elements w_orig[co][ci0 .. ci0+cn)[taps] are ONE contiguous piece,
// read coalesced into LDS and written out tap-major.  (Indexing the threads by the destination and gathering the source -- the first version --
// read 36 .. 64-byte-strided floats: every XCD's L2 fetched the same lines again, 178 MB of fabric reads per launch for 16 MB of weights.)
// conv_transpose sources (transposed_src) have no such contiguity and keep the gather.
__global__ __launch_bounds__(256) void weight_layout_fwd_kernel(const hv_wprep_layer* __restrict__ layers, int only_legacy) {
    const hv_wprep_layer L = layers[blockIdx.y];
    if (only_legacy && !L.transposed_src) return;      // (taken by weight_layout_fused_kernel)
    __shared__ float sh[4096 + 256];
    const int taps = L.taps, ldt = taps + 1;                                    // LDS rows [ci][taps + 1]: the tap-major read-out is conflict-free
    int CC = 256;
    while (CC > 1 && CC * ldt > 4096 + 256) CC >>= 1;                           // 256 channels up to 4x4 filters, 128 for 5x5, 64 for 7x7 ...
    const int nch = (L.CinP + CC - 1) / CC;
    const long long work = (long long)L.CoutF * nch;
    const float sigma = L.sigma[0];
    const int T = L.w_fwd_t ? tile_width(L.CinP) : 0;
    for (long long w = blockIdx.x; w < work; w += gridDim.x) {
        const int co = (int)(w / nch), ci0 = (int)(w % nch) * CC;
        const int cn = min(CC, L.CinP - ci0);                                   // destination channels of this item
        const int cs = (co < L.Cout) ? max(0, min(cn, L.Cin - ci0)) : 0;        // ... of which real
        __syncthreads();
        if (!L.transposed_src) {
            const float* src = L.w_orig + ((long long)co * L.Cin + ci0) * taps;
            for (int i = threadIdx.x; i < cs * taps; i += 256) sh[(i / taps) * ldt + i % taps] = src[i];
        } else {
            for (int i = threadIdx.x; i < cs * taps; i += 256) sh[(i / taps) * ldt + i % taps] = L.w_orig[wsrc_index(L, co, ci0 + i / taps, i % taps)];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < cn * taps; j += 256) {
            const int tap = j / cn, cil = j - tap * cn;
            const float val = cil < cs ? sh[cil * ldt + tap] / sigma : 0.f;
            const long long i = ((long long)co * taps + tap) * L.CinP + ci0 + cil;
            L.w_fwd[i] = val;
            if (L.w_fwd_h) reinterpret_cast<_Float16*>(L.w_fwd_h)[i] = (_Float16)val;
            if (T) reinterpret_cast<_Float16*>(L.w_fwd_t)[tiled_index(co, tap, ci0 + cil, taps, L.CinP, T)] = (_Float16)val;
        }
    }
}

// Data-gradient tables: w_bwd[ci][tap][co] = w_fwd[co][tap][ci], a 32 x 32 LDS tile transpose per tap (both sides 128-byte rows); runs after the
// forward tables of the same call (stream order).
__global__ __launch_bounds__(256) void weight_layout_bwd_kernel(const hv_wprep_layer* __restrict__ layers, int only_legacy) {
    const hv_wprep_layer L = layers[blockIdx.y];
    if (!L.w_bwd || (only_legacy && !L.transposed_src)) return;
    __shared__ float sh[32][33];
    const int taps = L.taps, tco = (L.CoutP + 31) / 32, tci = (L.CinB + 31) / 32;
    const long long work = (long long)taps * tco * tci;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int T = L.w_bwd_t ? tile_width(L.CoutP) : 0;
    for (long long w = blockIdx.x; w < work; w += gridDim.x) {
        const int tap = (int)(w % taps);
        const long long r = w / taps;
        const int co0 = (int)(r % tco) * 32, ci0 = (int)(r / tco) * 32;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = co0 + ty + 8 * k, ci = ci0 + tx;
            sh[ty + 8 * k][tx] = (co < L.Cout && co < L.CoutF && ci < L.Cin && ci < L.CinP) ? L.w_fwd[((long long)co * taps + tap) * L.CinP + ci] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ci0 + ty + 8 * k, co = co0 + tx;
            if (ci >= L.CinB || co >= L.CoutP) continue;
            const float val = sh[tx][ty + 8 * k];
            const long long i = ((long long)ci * taps + tap) * L.CoutP + co;
            L.w_bwd[i] = val;
            if (L.w_bwd_h) reinterpret_cast<_Float16*>(L.w_bwd_h)[i] = (_Float16)val;
            if (T) reinterpret_cast<_Float16*>(L.w_bwd_t)[tiled_index(ci, tap, co, taps, L.CoutP, T)] = (_Float16)val;
        }
    }
}

extern "C" int hv_weight_prep(const hv_wprep_layer* d_layers, int n_layers, long long max_numel, void* stream) {
    if (!d_layers || n_layers <= 0 || max_numel <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(n_layers), dim3(PREP_THREADS), 0, (hipStream_t)stream, d_layers);
    HV_LAUNCH_CHECK();
    // grid-stride over the layers' work items (the host knows only the largest table, not every layer's shape)
    const int gx = hv_cdiv(max_numel, 2048) < 1 ? 1 : hv_cdiv(max_numel, 2048);
    hipLaunchKernelGGL(weight_layout_fwd_kernel, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, d_layers, 0);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(weight_layout_bwd_kernel, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, d_layers, 0);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// All six tables of a layer from ONE read of its weights (hv_weight_prep2).  The two kernels above cost 23 + 15 us per discriminator and sit between the Adam
// step and the next forward, i.e. on the step's critical path four times (timing-only run without them: 8.45 -> 8.07 ms): they write element by element,
// the fragment-ordered tables as scattered 2-byte stores, and the data-gradient pass re-reads the forward table.  Here a work item is a 32 x 32
// (filter, input channel) tile with all its taps: the source rows (32 ci x taps contiguous floats per filter) go to LDS once, the plain tables leave as
// 128-byte (fp32) / 64-byte (fp16) runs and a 16 x 32 MFMA fragment of either ordered table as 64 sixteen-byte pieces = 1 KB contiguous.
// conv_transpose sources (no contiguous source rows) are left to the kernels above (hv_weight_prep2's any_legacy).
#define WL_T 32
__global__ __launch_bounds__(256) void weight_layout_fused_kernel(const hv_wprep_layer* __restrict__ layers) {
    const hv_wprep_layer L = layers[blockIdx.y];
    if (L.transposed_src) return;
    // a work item = (32 filters x 32 input channels) x a group of up to 4 taps: the largest PatchGAN layer is 128 tiles, with the taps split it is 512 items
    // (41 us with whole-tap tiles: each thread walked 16 taps x 4 rows three times, on 128 of the 256 CUs)
    constexpr int TG = 4, LDT = TG + 1, RS = WL_T * LDT + 1;
    __shared__ float wl_sh[WL_T * RS];                                         // [32 filters][32 channels x 5 + 1]
    const int taps = L.taps, ntg = (taps + TG - 1) / TG;
    // every table is optional on its own (round 4): a caller that knows which tables the layer's convolutions read (hv_last_weight_tables) passes NULL for
    // the others -- in the fp16 mode the big layers read the fragment-ordered tables only, 4 of the 20 bytes per weight this pass used to write
    const bool has_b = L.w_bwd || L.w_bwd_h || L.w_bwd_t;
    const int rows_f = L.CoutF, rows_b = has_b ? L.CinB : 0;
    const int nco = (max(L.CoutF, has_b ? L.CoutP : 0) + WL_T - 1) / WL_T, nci = (max(L.CinP, rows_b) + WL_T - 1) / WL_T;
    const float sigma = L.sigma[0];
    const int Tf = L.w_fwd_t ? tile_width(L.CinP) : 0, Tb = L.w_bwd_t ? tile_width(L.CoutP) : 0;
    const int tid = threadIdx.x, lane = tid & 31, sub = tid >> 5;
    for (int w = blockIdx.x; w < nco * nci * ntg; w += gridDim.x) {
        const int tg = w % ntg, tile = w / ntg;
        const int co0 = (tile / nci) * WL_T, ci0 = (tile % nci) * WL_T;
        const int t0 = tg * TG, tn = min(TG, taps - t0);                      // this item's taps t0 .. t0 + tn
        const int cs = max(0, min(WL_T, L.Cin - ci0));                        // real input channels of the tile
        __syncthreads();
        // ---- source -> LDS (zeros where the tile leaves the real weight): lane = input channel, its taps are contiguous in the source
        for (int r = sub; r < WL_T; r += 8) {
            const int co = co0 + r;
            const bool real = co < L.Cout && lane < cs;
            const float* src = L.w_orig + ((long long)co * L.Cin + ci0 + lane) * taps + t0;
            float* dst = wl_sh + r * RS + lane * LDT;
            if (real && tn == TG && !(taps & 3)) {
                const float4 v = *reinterpret_cast<const float4*>(src);
                dst[0] = v.x / sigma; dst[1] = v.y / sigma; dst[2] = v.z / sigma; dst[3] = v.w / sigma;       // (the division the element-wise kernels do: same bits)
            } else {
#pragma unroll
                for (int t = 0; t < TG; ++t) dst[t] = (real && t < tn) ? src[t] / sigma : 0.f;
            }
        }
        __syncthreads();
        // ---- forward tables [co][tap][ci]: a warp-wide run of 32 input channels per (co, tap)
        {
            const int ci = ci0 + lane;
            for (int r = sub; r < WL_T; r += 8) {
                const int co = co0 + r;
                if (co >= rows_f || ci >= L.CinP) continue;
                for (int t = 0; t < tn; ++t) {
                    const float val = wl_sh[r * RS + lane * LDT + t];
                    const long long i = ((long long)co * taps + t0 + t) * L.CinP + ci;
                    if (L.w_fwd) L.w_fwd[i] = val;
                    if (L.w_fwd_h) reinterpret_cast<_Float16*>(L.w_fwd_h)[i] = (_Float16)val;
                    if (Tf == 16) reinterpret_cast<_Float16*>(L.w_fwd_t)[tiled_index(co, t0 + t, ci, taps, L.CinP, 16)] = (_Float16)val;
                }
            }
        }
        if (Tf == 32 && ci0 < L.CinP) {       // fragment (16-row block rb, tap): [kq][row][8 channels], pieces of 16 bytes; threads = (tap parity, rb, kq, row)
            const int row = tid & 15, kq = (tid >> 4) & 3, half = tid >> 6, rb = half & 1;
            const int cob = co0 + rb * 16;
            if (cob < (rows_f + 15) / 16 * 16) {
                for (int t = half >> 1; t < tn; t += 2) {
                    f16x8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (_Float16)wl_sh[(rb * 16 + row) * RS + (kq * 8 + e) * LDT + t];
                    _Float16* base = reinterpret_cast<_Float16*>(L.w_fwd_t) + (long long)(cob >> 4) * 16 * taps * L.CinP + (long long)(((t0 + t) * L.CinP + ci0) >> 5) * 512;
                    *reinterpret_cast<f16x8*>(base + (kq * 16 + row) * 8) = h;
                }
            }
        }
        if (L.w_fwd_t2 && ci0 < L.K2) {       // the first K2 channels once more as a fragment-ordered table of their own (hv_conv_desc.x1 layers; K2 % 32 == 0)
            const int row = tid & 15, kq = (tid >> 4) & 3, half = tid >> 6, rb = half & 1;
            const int cob = co0 + rb * 16;
            if (cob < (rows_f + 15) / 16 * 16) {
                for (int t = half >> 1; t < tn; t += 2) {
                    f16x8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (_Float16)wl_sh[(rb * 16 + row) * RS + (kq * 8 + e) * LDT + t];
                    _Float16* base = reinterpret_cast<_Float16*>(L.w_fwd_t2) + (long long)(cob >> 4) * 16 * taps * L.K2 + (long long)(((t0 + t) * L.K2 + ci0) >> 5) * 512;
                    *reinterpret_cast<f16x8*>(base + (kq * 16 + row) * 8) = h;
                }
            }
        }
        // ---- data-gradient tables [ci][tap][co]: a run of 32 filters per (ci, tap)
        if (has_b) {
            const int co = co0 + lane;
            for (int cil = sub; cil < WL_T; cil += 8) {
                const int ci = ci0 + cil;
                if (ci >= rows_b || co >= L.CoutP) continue;
                const bool in = ci < L.CinP && co < L.CoutF;                                  // (the old pass read the forward table: zero beyond it)
                for (int t = 0; t < tn; ++t) {
                    const float val = in ? wl_sh[lane * RS + cil * LDT + t] : 0.f;
                    const long long i = ((long long)ci * taps + t0 + t) * L.CoutP + co;
                    if (L.w_bwd) L.w_bwd[i] = val;
                    if (L.w_bwd_h) reinterpret_cast<_Float16*>(L.w_bwd_h)[i] = (_Float16)val;
                    if (Tb == 16) reinterpret_cast<_Float16*>(L.w_bwd_t)[tiled_index(ci, t0 + t, co, taps, L.CoutP, 16)] = (_Float16)val;
                }
            }
            if (Tb == 32 && co0 < L.CoutP) {
                const int row = tid & 15, kq = (tid >> 4) & 3, half = tid >> 6, rb = half & 1;
                const int cib = ci0 + rb * 16;
                if (cib < (rows_b + 15) / 16 * 16) {
                    const int cil = rb * 16 + row;
                    const bool in = ci0 + cil < L.CinP;
                    for (int t = half >> 1; t < tn; t += 2) {
                        f16x8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (in && co0 + kq * 8 + e < L.CoutF) ? (_Float16)wl_sh[(kq * 8 + e) * RS + cil * LDT + t] : (_Float16)0.f;
                        _Float16* base = reinterpret_cast<_Float16*>(L.w_bwd_t) + (long long)(cib >> 4) * 16 * taps * L.CoutP + (long long)(((t0 + t) * L.CoutP + co0) >> 5) * 512;
                        *reinterpret_cast<f16x8*>(base + (kq * 16 + row) * 8) = h;
                    }
                }
            }
        }
    }
}

extern "C" int hv_weight_prep2(const hv_wprep_layer* d_layers, int n_layers, long long max_numel, int any_sn, int any_legacy, void* stream) {
    if (!d_layers || n_layers <= 0 || max_numel <= 0) return HV_ERR_ARG;
    static const int fused = getenv("HV_WPREP_FUSED") ? atoi(getenv("HV_WPREP_FUSED")) : 1;      // A/B knob
    if (!fused) return hv_weight_prep(d_layers, n_layers, max_numel, stream);
    if (any_sn) {      // sigma (and the power iteration); layers without spectral norm keep the 1.0 their sigma slot was created with
        hipLaunchKernelGGL(weight_prep_kernel, dim3(n_layers), dim3(PREP_THREADS), 0, (hipStream_t)stream, d_layers);
        HV_LAUNCH_CHECK();
    }
    // one item = 32 x 32 x 4 elements of a table: about one workgroup per item of the largest layer, grid-stride for the rest
    int gx = (int)hv_cdiv(max_numel, 2 * 1024 * 4);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(weight_layout_fused_kernel, dim3(gx, n_layers), dim3(256), 0, (hipStream_t)stream, d_layers);
    HV_LAUNCH_CHECK();
    if (any_legacy) {
        const int gl = hv_cdiv(max_numel, 2048) < 1 ? 1 : hv_cdiv(max_numel, 2048);
        hipLaunchKernelGGL(weight_layout_fwd_kernel, dim3(gl, n_layers), dim3(256), 0, (hipStream_t)stream, d_layers, 1);
        HV_LAUNCH_CHECK();
        hipLaunchKernelGGL(weight_layout_bwd_kernel, dim3(gl, n_layers), dim3(256), 0, (hipStream_t)stream, d_layers, 1);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}


// backward phase 1 (spectral-norm layers only): dot = <dWsn, Wsn> -> sigma[1]
__global__ __launch_bounds__(PREP_THREADS) void weight_prep_bwd_dot_kernel(const hv_wprep_bwd_layer* __restrict__ layers) {
    const hv_wprep_bwd_layer L = layers[blockIdx.x];
    if (!L.sn) return;
    __shared__ float red[20];
    const long long n = (long long)L.Cout * L.taps * L.CinP;
    // one workgroup per layer: four independent 16-byte streams per lane keep the loads in flight (a scalar loop was a chain of load latencies:
    // 63 us for the generator's 147 K-element layers); n is a multiple of 4 (padded channel counts) and both tables are 16-byte aligned
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    const long long n4 = n / 4;
    const float4* ga = reinterpret_cast<const float4*>(L.dw_ohwi);
    const float4* wa = reinterpret_cast<const float4*>(L.w_fwd);
    const bool vec = !(n & 3) && !(((uintptr_t)L.dw_ohwi | (uintptr_t)L.w_fwd) & 15);
    if (vec) {
        long long i = threadIdx.x;
        for (; i + 3 * PREP_THREADS < n4; i += 4 * PREP_THREADS) {
            const float4 a0 = ga[i], a1 = ga[i + PREP_THREADS], a2 = ga[i + 2 * PREP_THREADS], a3 = ga[i + 3 * PREP_THREADS];
            const float4 b0 = wa[i], b1 = wa[i + PREP_THREADS], b2 = wa[i + 2 * PREP_THREADS], b3 = wa[i + 3 * PREP_THREADS];
            d0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
            d1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
            d2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
            d3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
        }
        for (; i < n4; i += PREP_THREADS) { const float4 a0 = ga[i], b0 = wa[i]; d0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w; }
    } else {
        for (long long i = threadIdx.x; i < n; i += PREP_THREADS) d0 += L.dw_ohwi[i] * L.w_fwd[i];
    }
    float dot = (d0 + d1) + (d2 + d3);
    dot = hv_block_sum(dot, red);
    if (threadIdx.x == 0) const_cast<float*>(L.sigma)[1] = dot;
}

__global__ __launch_bounds__(256) void weight_prep_bwd_kernel(const hv_wprep_bwd_layer* __restrict__ layers) {
    const hv_wprep_bwd_layer L = layers[blockIdx.y];
    const long long no = (long long)L.Cout * L.Cin * L.taps;
    const long long base = (long long)blockIdx.x * 1024;
    if (base >= no) return;
    float dot = 0.f, sigma = 1.f;
    if (L.sn) { sigma = L.sigma[0]; dot = L.sigma[1]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i >= no) break;
        int co, ci;
        // (a layer has < 2^31 weights: 32-bit divisions -- the 64-bit ones were most of this kernel's instructions)
        const unsigned iu = (unsigned)i, r = iu / (unsigned)L.taps;
        const int tap = (int)(iu - r * (unsigned)L.taps);
        if (L.transposed_src) { ci = (int)(r / (unsigned)L.Cout); co = (int)(r - (unsigned)ci * (unsigned)L.Cout); }
        else { co = (int)(r / (unsigned)L.Cin); ci = (int)(r - (unsigned)co * (unsigned)L.Cin); }
        // conv: dw_ohwi is [co][tap][CinP]; conv_transpose (transposed_src): [ci][tap][CinP] with CinP = padded Cout
        float g = L.transposed_src ? L.dw_ohwi[((long long)ci * L.taps + tap) * L.CinP + co]
                                   : L.dw_ohwi[((long long)co * L.taps + tap) * L.CinP + ci];
        if (L.sn) g = (g - dot * L.u[co] * L.v[ci * L.taps + tap]) / sigma;
        L.dw_orig[i] = L.accumulate ? L.dw_orig[i] + g : g;
    }
}

extern "C" int hv_weight_prep_backward(const hv_wprep_bwd_layer* d_layers, int n_layers, long long max_numel, int any_sn, void* stream) {
    if (!d_layers || n_layers <= 0 || max_numel <= 0) return HV_ERR_ARG;
    if (max_numel >= (1ll << 31)) return HV_ERR_UNSUPPORTED;      // (weight_prep_bwd_kernel indexes a layer's weights in 32 bits)
    if (any_sn) {
        hipLaunchKernelGGL(weight_prep_bwd_dot_kernel, dim3(n_layers), dim3(PREP_THREADS), 0, (hipStream_t)stream, d_layers);
        HV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(weight_prep_bwd_kernel, dim3(hv_cdiv(max_numel, 1024), n_layers), dim3(256), 0, (hipStream_t)stream, d_layers);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ 1-channel heads: loss seed -> gradient carrier
// g = seed * act'(y) written as channel 0 of an fp16 [pixel][4] carrier (channels 1-3 zero: the carrier is the padded gradient operand of the head's
// weight / data gradient kernels) + the bias gradient sum(g).  Replaces three passes at the head of the generator backward (fp32 seed -> carrier copy,
// in-place act' pass whose 1-channel form took 18 us for 2 MB, column sums): four pixels per thread, one 32-byte store.
__global__ __launch_bounds__(256) void head_seed_kernel(const float* __restrict__ seed, const void* __restrict__ y, int yh, int y_ld, int y_coff,
                                                        _Float16* __restrict__ carrier, long long npix, int act, float* __restrict__ part) {
    __shared__ float red[20];
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    float s = 0.f;
    if (p0 + 3 < npix) {
        const float4 sd = *reinterpret_cast<const float4*>(seed + p0);
        float g[4] = {sd.x, sd.y, sd.z, sd.w};
        f16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = hi[e] = (_Float16)0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[u] *= hv_act_grad_from_out(hv_ld1(y, (p0 + u) * y_ld + y_coff, yh), act);
            const _Float16 h = (_Float16)g[u];
            s += (float)h;                                   // the bias gradient sums what the carrier holds (as the in-place pass did)
            if (u < 2) lo[u * 4] = h; else hi[(u - 2) * 4] = h;
        }
        *reinterpret_cast<f16x8*>(carrier + p0 * 4) = lo;
        *reinterpret_cast<f16x8*>(carrier + p0 * 4 + 8) = hi;
    } else {
        for (long long p = p0; p < npix; ++p) {
            const _Float16 h = (_Float16)(seed[p] * hv_act_grad_from_out(hv_ld1(y, p * y_ld + y_coff, yh), act));
            s += (float)h;
            carrier[p * 4] = h; carrier[p * 4 + 1] = carrier[p * 4 + 2] = carrier[p * 4 + 3] = (_Float16)0.f;
        }
    }
    if (part) {
        s = hv_block_sum(s, red);
        if (threadIdx.x == 0) part[blockIdx.x] = s;
    }
    // (Measured and not kept: the last of the 1 024 workgroups folding the block sums itself -- the ticket atomics on one address serialise: 8 -> 22 us for
    // the 6-us launch it saves; hv_common.h.)
}

__global__ __launch_bounds__(64) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out, int accumulate);

extern "C" size_t hv_head_seed_workspace_bytes(long long npix) { return (size_t)((npix + 1023) / 1024) * sizeof(float); }

extern "C" int hv_head_seed_backward(const float* seed, const void* y, int y_f16, int y_ld, int y_coff, void* carrier_f16, long long npix, int act, float* dbias,
                                     int dbias_accumulate, float* workspace, size_t workspace_bytes, void* stream) {
    if (!seed || !y || !carrier_f16 || npix <= 0 || y_ld < 1 || act < HV_ACT_NONE || act > HV_ACT_CLAMP) return HV_ERR_ARG;
    if (((uintptr_t)seed | (uintptr_t)carrier_f16) & 15) return HV_ERR_UNSUPPORTED;
    const int nb = (int)((npix + 1023) / 1024);
    if (dbias && (!workspace || workspace_bytes < (size_t)nb * sizeof(float))) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_seed_kernel, dim3(nb), dim3(256), 0, s, seed, y, y_f16, y_ld, y_coff, reinterpret_cast<_Float16*>(carrier_f16), npix, act,
                       dbias ? workspace : nullptr);
    HV_LAUNCH_CHECK();
    if (dbias) {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(1), dim3(64), 0, s, workspace, nb, 1, dbias, dbias_accumulate);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ activation gradient
// vector path: C % 4 == 0 and C/4 a power of two <= 256: thread owns one 4-channel group, rows strided.
template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(void* __restrict__ dy, const void* __restrict__ y, int H, int YH, long long npix, int C,
                                                      int dy_ld, int dy_coff, int y_ld, int y_coff, int act,
                                                      float* __restrict__ part, int rows_per_block) {
    __shared__ float sh[256 * 4];
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(npix, r0 + rows_per_block);
    if (VEC) {
        const int C4 = C >> 2, cg = tid % C4, rp = tid / C4, rstep = 256 / C4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        long long r = r0 + rp;
        for (; r + 3 * rstep < r1; r += 4 * rstep) {   // four rows in flight per lane (HBM streaming needs the loads, not the math)
            float4 g[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = hv_ld4(dy, (r + u * rstep) * dy_ld + dy_coff + cg * 4, H);
                o[u] = hv_ld4(y, (r + u * rstep) * y_ld + y_coff + cg * 4, YH);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
                hv_act_grad4(f, act);      // (one switch per quad)
                g[u].x *= f[0]; g[u].y *= f[1]; g[u].z *= f[2]; g[u].w *= f[3];
                hv_st4(dy, (r + u * rstep) * dy_ld + dy_coff + cg * 4, g[u], H);
                s.x += g[u].x; s.y += g[u].y; s.z += g[u].z; s.w += g[u].w;
            }
        }
        for (; r < r1; r += rstep) {
            float4 g = hv_ld4(dy, r * dy_ld + dy_coff + cg * 4, H);
            const float4 o = hv_ld4(y, r * y_ld + y_coff + cg * 4, YH);
            float f[4] = {o.x, o.y, o.z, o.w};
            hv_act_grad4(f, act);
            g.x *= f[0]; g.y *= f[1]; g.z *= f[2]; g.w *= f[3];
            hv_st4(dy, r * dy_ld + dy_coff + cg * 4, g, H);
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        }
        if (part) {
            reinterpret_cast<float4*>(sh)[tid] = s;
            __syncthreads();
            if (tid < C4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = tid; k < 256; k += C4) {
                    const float4 q = reinterpret_cast<float4*>(sh)[k];
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
                *reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + tid * 4) = t;
            }
        }
    } else {  // scalar path: C a power of two <= 256
        const int c = tid % C, rp = tid / C, rstep = 256 / C;
        float s = 0.f;
        long long r = r0 + rp;
        for (; r + 3 * rstep < r1; r += 4 * rstep) {       // four rows in flight (the 1-channel image tensors: a lane's loop was a chain of load latencies)
            float g[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = hv_ld1(dy, (r + u * rstep) * dy_ld + dy_coff + c, H);
                o[u] = hv_ld1(y, (r + u * rstep) * y_ld + y_coff + c, YH);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] *= hv_act_grad_from_out(o[u], act);
                hv_st1(dy, (r + u * rstep) * dy_ld + dy_coff + c, g[u], H);
                s += g[u];
            }
        }
        for (; r < r1; r += rstep) {
            float g = hv_ld1(dy, r * dy_ld + dy_coff + c, H) * hv_act_grad_from_out(hv_ld1(y, r * y_ld + y_coff + c, YH), act);
            hv_st1(dy, r * dy_ld + dy_coff + c, g, H);
            s += g;
        }
        if (part) {
            sh[tid] = s;
            __syncthreads();
            if (tid < C) {
                float t = 0.f;
                for (int k = tid; k < 256; k += C) t += sh[k];
                part[(long long)blockIdx.x * C + tid] = t;
            }
        }
    }
}

__global__ __launch_bounds__(64) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x;   // one wave per channel
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[(long long)b * C + c];
    s = hv_wave_sum(s);
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + s : s;
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool act_vec_ok(int C) { return (C % 4 == 0) && pow2(C / 4) && C / 4 <= 256; }

// rows handled per block (multiple of the row step of the thread mapping) and the block count
static int act_bwd_blocks(long long npix, int C, int* rows_per_block) {
    const int rstep = act_vec_ok(C) ? 256 / (C / 4) : 256 / C;
    long long rpb = (long long)rstep * 4;       // one 4-row batch per lane unless that exceeds the block cap below (256 blocks = one per CU was a latency chain)
    long long nb = (npix + rpb - 1) / rpb;
    static const int ab = getenv("HV_ACT_BLOCKS") ? atoi(getenv("HV_ACT_BLOCKS")) : 2048;   // tuning knob
    if (nb > ab) {   // ~8 workgroups per CU keep enough loads in flight to stream from HBM
        rpb = (npix + ab - 1) / ab;
        rpb = (rpb + rstep - 1) / rstep * rstep;
        nb = (npix + rpb - 1) / rpb;
    }
    *rows_per_block = (int)rpb;
    return (int)nb;
}

extern "C" size_t hv_act_backward_workspace_bytes(long long npix, int C) {
    if (npix <= 0 || C <= 0 || !(act_vec_ok(C) || (pow2(C) && C <= 256))) return 0;
    int rpb;
    return (size_t)act_bwd_blocks(npix, C, &rpb) * C * sizeof(float);
}

extern "C" int hv_act_backward(void* dy, int dy_f16, const void* y, int y_f16, long long npix, int C, int dy_ld, int dy_coff, int y_ld, int y_coff,
                               int act, float* dbias, int dbias_accumulate, float* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !y || npix <= 0 || C <= 0) return HV_ERR_ARG;
    const bool aligned = !(dy_ld & 3) && !(dy_coff & 3) && !(y_ld & 3) && !(y_coff & 3) && !((uintptr_t)dy & 15) && !((uintptr_t)y & 15);
    if (!(act_vec_ok(C) && aligned) && !(pow2(C) && C <= 256)) return HV_ERR_UNSUPPORTED;
    // the block plan depends only on C (the workspace query must agree), the code path also on alignment
    const bool vec = act_vec_ok(C) && aligned;
    int rpb;
    const int nb = act_bwd_blocks(npix, C, &rpb);
    if (!vec && act_vec_ok(C)) return HV_ERR_UNSUPPORTED;  // vector-shaped C on unaligned views is not needed by the path
    if (dbias && (!workspace || workspace_bytes < (size_t)nb * C * sizeof(float))) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = dbias ? workspace : nullptr;
    if (vec) hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(nb), dim3(256), 0, s, dy, y, dy_f16, y_f16, npix, C, dy_ld, dy_coff, y_ld, y_coff, act, part, rpb);
    else hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(nb), dim3(256), 0, s, dy, y, dy_f16, y_f16, npix, C, dy_ld, dy_coff, y_ld, y_coff, act, part, rpb);
    HV_LAUNCH_CHECK();
    if (dbias) {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, s, workspace, nb, C, dbias, dbias_accumulate);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void adam_tick_kernel(float* step) { step[0] += 1.f; }

__global__ __launch_bounds__(256) void adam_kernel(const hv_adam_tensor* __restrict__ ts, const float* __restrict__ lr_p, float beta1,
                                                   float beta2, float eps, const float* __restrict__ step_p) {
    const hv_adam_tensor t = ts[blockIdx.y];
    const long long base = (long long)blockIdx.x * 1024;
    if (base >= t.n) return;
    const float step = step_p[0], lr = lr_p[0];
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < t.n) {
            const float g = t.g[i];
            const float m = t.m[i] + (g - t.m[i]) * (1.f - beta1);
            const float v = t.v[i] * beta2 + (1.f - beta2) * g * g;
            t.m[i] = m;
            t.v[i] = v;
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            t.p[i] = t.p[i] - step_size * (m / denom);
        }
    }
}

// ---- guarded form (fp16 storage mode: the gradients carry a loss scale and may have overflowed to inf / nan in an fp16 gradient buffer)
// state[0] = step count, [1] = "this gradient is not finite" (set by the check, consumed by the tick), [2] = skipped steps so far, [3] = skip THIS step,
// [4] = ticket of the check kernel's workgroups (an unsigned; back at zero when the launch ends)
// One pass over the flat gradient: the loss scale taken out (mul, a power of two: exact; 1 = leave the values alone), the finite test on the way, and the
// LAST workgroup to arrive (publish -> ticket -> collect, hv_common.h) does what used to be a one-thread launch between the check
// and the update: count the step or the skip, publish "skip this step", clear the flag.
__global__ __launch_bounds__(256) void grad_unscale_check_kernel(float* __restrict__ g, long long n, float mul, float* __restrict__ state) {
    bool bad = false;
    const bool scale = mul != 1.f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        if (i + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(g + i);
            bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);      // false for inf AND nan
            if (scale) { v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul; *reinterpret_cast<float4*>(g + i) = v; }
        } else {
            for (long long j = i; j < n; ++j) { bad |= !(fabsf(g[j]) <= 3.0e38f); if (scale) g[j] *= mul; }
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) { hv_publish(state + 1, 1.f); hv_stores_done(); }      // every writer stores the same value
    __syncthreads();
    if (threadIdx.x == 0) {      // (hv_common.h: publish / ticket / collect without agent-scope fences -- a release fence here would also write back the gradients)
        unsigned* ticket = reinterpret_cast<unsigned*>(state + 4);
        if (hv_take_ticket_is_last(ticket, gridDim.x)) {
            const bool any_bad = hv_collect(state + 1) != 0.f;
            if (any_bad) state[2] += 1.f; else state[0] += 1.f;
            state[3] = any_bad ? 1.f : 0.f;
            hv_publish(state + 1, 0.f);
            hv_ticket_reset(ticket);
        }
    }
}
__global__ __launch_bounds__(256) void adam_guarded_kernel(const hv_adam_tensor* __restrict__ ts, const float* __restrict__ lr_p, float beta1,
                                                           float beta2, float eps, const float* __restrict__ state) {
    if (state[3] != 0.f) return;          // a non-finite gradient: weights, moments and step count stay as they are
    const hv_adam_tensor t = ts[blockIdx.y];
    const long long base = (long long)blockIdx.x * 1024;
    if (base >= t.n) return;
    const float step = state[0], lr = lr_p[0];
    const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < t.n) {
            const float g = t.g[i];
            const float m = t.m[i] + (g - t.m[i]) * (1.f - beta1);
            const float v = t.v[i] * beta2 + (1.f - beta2) * g * g;
            t.m[i] = m;
            t.v[i] = v;
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            t.p[i] = t.p[i] - step_size * (m / denom);
        }
    }
}
extern "C" int hv_adam_step_guarded(const hv_adam_tensor* d_tensors, int n_tensors, long long max_numel, const float* d_lr, float beta1,
                                    float beta2, float eps, float* d_state, float* flat_grad, long long n_grad, float grad_mul, void* stream) {
    if (!d_tensors || n_tensors <= 0 || max_numel <= 0 || !d_lr || !d_state || !flat_grad || n_grad <= 0 || ((uintptr_t)flat_grad & 15) || !(grad_mul > 0.f))
        return HV_ERR_ARG;
    const int blocks = (int)(n_grad / 4096 + 1 < 256 ? n_grad / 4096 + 1 : 256);      // (every workgroup takes a ticket on one address: few of them)
    hipLaunchKernelGGL(grad_unscale_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flat_grad, n_grad, grad_mul, d_state);
    HV_LAUNCH_CHECK();
    dim3 grid(hv_cdiv(max_numel, 1024), n_tensors);
    hipLaunchKernelGGL(adam_guarded_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_tensors, d_lr, beta1, beta2, eps, d_state);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

extern "C" int hv_adam_step(const hv_adam_tensor* d_tensors, int n_tensors, long long max_numel, const float* d_lr, float beta1,
                            float beta2, float eps, float* d_step, void* stream) {
    if (!d_tensors || n_tensors <= 0 || max_numel <= 0 || !d_lr || !d_step) return HV_ERR_ARG;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_step);
    HV_LAUNCH_CHECK();
    dim3 grid(hv_cdiv(max_numel, 1024), n_tensors);
    hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_tensors, d_lr, beta1, beta2, eps, d_step);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ fills
__global__ void fill_kernel(float* p, long long n, float v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = v;
}
__global__ void axpy_kernel(float* y, const float* x, long long n, float a, int assign) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) y[i] = assign ? a * x[i] : y[i] + a * x[i];
}
static int ew_grid(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
extern "C" int hv_fill(float* p, long long n, float value, void* stream) {
    if (!p || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, n, value);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_axpy(float* y, const float* x, long long n, float a, void* stream) {
    if (!y || !x || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, a, 0);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
__global__ void affine_kernel(float* y, const float* x, long long n, float a, float b) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) y[i] = a * x[i] + b;
}
extern "C" int hv_affine(float* y, const float* x, long long n, float a, float b, void* stream) {
    if (!y || !x || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(affine_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, a, b);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

__global__ void mul_kernel(float* y, const float* x, const float* z, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) y[i] = z ? (y[i] * x[i]) * z[i] : y[i] * x[i];
}
extern "C" int hv_mul(float* y, const float* x, long long n, void* stream) {
    if (!y || !x || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(mul_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, (const float*)nullptr, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_mul3(float* y, const float* x, const float* z, long long n, void* stream) {
    if (!y || !x || !z || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(mul_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, z, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

extern "C" int hv_version(void) { return 101; }
extern "C" const char* hv_arch(void) { return "gfx950"; }
