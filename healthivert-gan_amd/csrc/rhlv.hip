// Relative height loss value (RHLV) of a generated vs the original vertebra label volume, on the device
// (reference evaluation/RHLV_quantification.py:41-147 and the per-vertebra body of process_datasets_to_excel :160-178).
//
// Integer / HBM-byte work: both volumes are read once (column counts per z-slice), everything after that is a few KB.
//   rhlv_counts_kernel   grid (Z, 2): cnt[v][z][w] = #{h : vol_v[h][w][z] == label_index}, tot[v][z]   (lanes along w; a z-fastest
//                        variant with lanes along z serves the reference's [H][W][Z] arrays)
//   rhlv_range_kernel    <<<1,256>>>: z-extent of the original vertebra -> centre, half-length -> [lo, hi) (numpy slice rules)
//   rhlv_slice_kernel    grid (Z): thirds of the generated vertebra's column extent, centre columns, rescale ratios,
//                        thresholded integer sums per (all | pre | mid | post) x (generated | original)
//   rhlv_final_kernel    <<<1,64>>>: means over the slices, the four RHLVs and the relative height of the original
// All floating-point steps are doubles in the reference's operation order (-ffp-contract=off); the only deviation is that a
// slice's selected heights are summed as integers and scaled once (sum(c)*r instead of sum(c*r)): ~1e-16 relative.
#include "hv_common.h"

struct RhlvRec { double ratio[4]; long long Sf[4], nf[4], Sl[4], nl[4]; };

template <typename T>
__global__ __launch_bounds__(256) void rhlv_counts_kernel(const T* __restrict__ fake, const T* __restrict__ label, long long sh, long long sw,
                                                          long long sz, int H, int W, float label_index, int* __restrict__ cnt,
                                                          int* __restrict__ tot) {
    __shared__ int red[256];
    const int z = blockIdx.x, v = blockIdx.y, Z = gridDim.x;
    const T* vol = v == 0 ? fake : label;
    int mine = 0;
    for (int w = threadIdx.x; w < W; w += 256) {
        int c = 0;
        const T* p = vol + (long long)w * sw + (long long)z * sz;
        for (int h = 0; h < H; ++h) {
            const float val = (float)p[(long long)h * sh];
            c += label_index < 0.f ? (val != 0.f) : (val == label_index);
        }
        cnt[((long long)v * Z + z) * W + w] = c;
        mine += c;
    }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[v * Z + z] = red[0];
}

// the same for volumes whose z index is the fastest-varying one in memory (the reference's [H][W][Z] arrays): a lane owns one (w, z)
// column and walks h, 64 consecutive lanes read 64 consecutive z -- coalesced.  grid (ceil(W*Z/256), 2); tot by rhlv_tot_kernel.
template <typename T>
__global__ __launch_bounds__(256) void rhlv_counts_zfast_kernel(const T* __restrict__ fake, const T* __restrict__ label, long long sh, long long sw,
                                                                long long sz, int H, int W, int Z, float label_index, int* __restrict__ cnt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)W * Z) return;
    const int v = blockIdx.y, w = (int)(i / Z), z = (int)(i - (long long)w * Z);
    const T* p = (v == 0 ? fake : label) + (long long)w * sw + (long long)z * sz;
    int c = 0;
    for (int h = 0; h < H; ++h) {
        const float val = (float)p[(long long)h * sh];
        c += label_index < 0.f ? (val != 0.f) : (val == label_index);
    }
    cnt[((long long)v * Z + z) * W + w] = c;
}
__global__ __launch_bounds__(256) void rhlv_tot_kernel(const int* __restrict__ cnt, int W, int* __restrict__ tot) {
    __shared__ int red[256];
    const int z = blockIdx.x, v = blockIdx.y, Z = gridDim.x;
    int mine = 0;
    for (int w = threadIdx.x; w < W; w += 256) mine += cnt[((long long)v * Z + z) * W + w];
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[v * Z + z] = red[0];
}

// params: [0] lo, [1] hi, [2] center_z, [3] length, [4] valid (label has voxels)
__global__ void rhlv_range_kernel(const int* __restrict__ tot, int Z, int length_divisor, int z_lo, int z_hi, int* __restrict__ params) {
    if (threadIdx.x != 0) return;
    int lo, hi, cz = 0, len = 0, valid = 1;
    if (z_lo == INT_MIN) {
        long long n = 0, sz = 0;
        int mn = Z, mx = -1;
        for (int z = 0; z < Z; ++z) {
            const int t = tot[Z + z];
            if (t > 0) { n += t; sz += (long long)t * z; mn = min(mn, z); mx = max(mx, z); }
        }
        if (n == 0) { valid = 0; lo = hi = 0; }
        else {
            cz = (int)((double)sz / (double)n);          // int(np.mean(loc))
            len = (mx - mn) / length_divisor;           // (max_z - min_z) // length_divisor
            lo = cz - len; hi = cz + len;
        }
    } else { lo = z_lo; hi = z_hi; cz = (z_lo + z_hi) / 2; len = (z_hi - z_lo) / 2; }
    // numpy slice normalisation of [lo:hi] on an axis of length Z
    if (lo < 0) lo = max(lo + Z, 0);
    if (hi < 0) hi = max(hi + Z, 0);
    lo = min(lo, Z); hi = min(hi, Z);
    params[0] = lo; params[1] = hi; params[2] = cz; params[3] = len; params[4] = valid;
}

__device__ __forceinline__ long long rhlv_block_sum(long long v, long long* sh) {
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}
__device__ __forceinline__ int rhlv_block_max(int v, long long* sh) {
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    return (int)sh[0];
}

__global__ __launch_bounds__(256) void rhlv_slice_kernel(const int* __restrict__ cnt, const int* __restrict__ tot, const int* __restrict__ params,
                                                         int W, double thr, RhlvRec* __restrict__ recs) {
    __shared__ long long sh[256];
    const int z = blockIdx.x, Z = gridDim.x, tid = threadIdx.x;
    RhlvRec* rec = recs + z;
    const int* cf = cnt + (long long)z * W;
    const int* cl = cnt + ((long long)Z + z) * W;
    const bool on = z >= params[0] && z < params[1] && tot[z] > 0 && tot[Z + z] > 0 && params[4];
    if (!on) {
        if (tid < 4) { rec->ratio[tid] = 1.0; rec->Sf[tid] = rec->nf[tid] = rec->Sl[tid] = rec->nl[tid] = 0; }
        return;
    }
    // column statistics of the generated and the original vertebra
    int ymin = W, ymax = -1;
    long long swf = 0, swl = 0;
    for (int w = tid; w < W; w += 256) {
        if (cf[w] > 0) { ymin = min(ymin, w); ymax = max(ymax, w); }
        swf += (long long)cf[w] * w;
        swl += (long long)cl[w] * w;
    }
    ymax = rhlv_block_max(ymax, sh);
    ymin = -rhlv_block_max(-ymin, sh);
    swf = rhlv_block_sum(swf, sh);
    swl = rhlv_block_sum(swl, sh);
    const int y_range = ymax - ymin;
    const int t1 = (int)((double)ymin + (double)y_range / 3.0);            // int(y_min + y_range/3)
    const int t2 = (int)((double)ymin + (double)(2 * y_range) / 3.0);      // int(y_min + 2*y_range/3)
    const int ccf = (int)((double)swf / (double)tot[z]);                   // int(np.mean(loc))
    const int ccl = (int)((double)swl / (double)tot[Z + z]);
    const int center_f_i = cf[ccf], center_l = cl[ccl];
    const int r0[4] = {0, 0, t1, t2}, r1[4] = {W, t1, t2, W};
    double ratio[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int mf = -1, ml = -1;
        for (int w = r0[c] + tid; w < r1[c]; w += 256) { mf = max(mf, cf[w]); ml = max(ml, cl[w]); }
        mf = rhlv_block_max(mf, sh);
        ml = rhlv_block_max(ml, sh);
        ratio[c] = (r1[c] > r0[c] && ml > mf) ? (double)ml / ((double)mf + 1e-6) : 1.0;
    }
    const double center_f = (double)center_f_i * ratio[0];
    const double thr_f = center_f * thr, thr_l = (double)center_l * thr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        long long Sf = 0, nf = 0, Sl = 0, nl = 0;
        for (int w = r0[c] + tid; w < r1[c]; w += 256) {
            if ((double)cf[w] * ratio[c] > thr_f) { Sf += cf[w]; ++nf; }
            if ((double)cl[w] > thr_l) { Sl += cl[w]; ++nl; }
        }
        Sf = rhlv_block_sum(Sf, sh); nf = rhlv_block_sum(nf, sh);
        Sl = rhlv_block_sum(Sl, sh); nl = rhlv_block_sum(nl, sh);
        if (tid == 0) { rec->ratio[c] = ratio[c]; rec->Sf[c] = Sf; rec->nf[c] = nf; rec->Sl[c] = Sl; rec->nl[c] = nl; }
    }
}

// out[0..4] = all, pre, mid, post RHLV, relative height of the original; out[5..12] = the eight mean heights
// (all_f, all_l, pre_f, pre_l, mid_f, mid_l, post_f, post_l); out[13] = 1 if the original vertebra exists, else 0
__global__ void rhlv_final_kernel(const RhlvRec* __restrict__ recs, const int* __restrict__ params, int Z, double* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double m[8];
    for (int c = 0; c < 4; ++c) {
        double sf = 0.0, sl = 0.0;
        long long nf = 0, nl = 0;
        for (int z = 0; z < Z; ++z) {
            sf += (double)recs[z].Sf[c] * recs[z].ratio[c];
            sl += (double)recs[z].Sl[c];
            nf += recs[z].nf[c];
            nl += recs[z].nl[c];
        }
        m[2 * c] = nf > 0 ? sf / (double)nf : 0.0;
        m[2 * c + 1] = nl > 0 ? sl / (double)nl : 0.0;
    }
    for (int c = 0; c < 4; ++c) out[c] = (m[2 * c] - m[2 * c + 1]) / (m[2 * c] + 1e-6);
    const double mn = fmin(m[3], fmin(m[5], m[7])), mx = fmax(m[3], fmax(m[5], m[7]));
    out[4] = mn / (mx + 1e-6);
    for (int i = 0; i < 8; ++i) out[5 + i] = m[i];
    out[13] = (double)params[4];
}

extern "C" size_t hv_rhlv_workspace_bytes(int W, int Z) {
    if (W <= 0 || Z <= 0) return 0;
    return (size_t)2 * Z * W * sizeof(int) + (size_t)2 * Z * sizeof(int) + 64 + (size_t)Z * sizeof(RhlvRec) + 64;
}

extern "C" int hv_rhlv(const void* fake, const void* label, int dtype, long long stride_h, long long stride_w, long long stride_z, int H, int W, int Z,
                       float label_index, int length_divisor, int z_lo, int z_hi, double height_threshold, double* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!fake || !label || !out || H <= 0 || W <= 0 || Z <= 0 || length_divisor <= 0 || (dtype != 0 && dtype != 1)) return HV_ERR_ARG;
    if (Z > 65535) return HV_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < hv_rhlv_workspace_bytes(W, Z) || ((uintptr_t)workspace & 7)) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    RhlvRec* recs = (RhlvRec*)ws; ws += (size_t)Z * sizeof(RhlvRec);
    int* cnt = (int*)ws; ws += (size_t)2 * Z * W * sizeof(int);
    int* tot = (int*)ws; ws += (size_t)2 * Z * sizeof(int);
    int* params = (int*)ws;
    if (stride_z == 1 && stride_w != 1) {   // z fastest in memory: lanes along z
        const dim3 grid(hv_cdiv((long long)W * Z, 256), 2);
        if (dtype == 0)
            hipLaunchKernelGGL((rhlv_counts_zfast_kernel<float>), grid, dim3(256), 0, s, (const float*)fake, (const float*)label, stride_h, stride_w,
                               stride_z, H, W, Z, label_index, cnt);
        else
            hipLaunchKernelGGL((rhlv_counts_zfast_kernel<unsigned char>), grid, dim3(256), 0, s, (const unsigned char*)fake, (const unsigned char*)label,
                               stride_h, stride_w, stride_z, H, W, Z, label_index, cnt);
        HV_LAUNCH_CHECK();
        hipLaunchKernelGGL(rhlv_tot_kernel, dim3(Z, 2), dim3(256), 0, s, cnt, W, tot);
    } else if (dtype == 0) {
        hipLaunchKernelGGL((rhlv_counts_kernel<float>), dim3(Z, 2), dim3(256), 0, s, (const float*)fake, (const float*)label, stride_h, stride_w, stride_z,
                           H, W, label_index, cnt, tot);
    } else {
        hipLaunchKernelGGL((rhlv_counts_kernel<unsigned char>), dim3(Z, 2), dim3(256), 0, s, (const unsigned char*)fake, (const unsigned char*)label,
                           stride_h, stride_w, stride_z, H, W, label_index, cnt, tot);
    }
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(rhlv_range_kernel, dim3(1), dim3(64), 0, s, tot, Z, length_divisor, z_lo, z_hi, params);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(rhlv_slice_kernel, dim3(Z), dim3(256), 0, s, cnt, tot, params, W, height_threshold, recs);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(rhlv_final_kernel, dim3(1), dim3(64), 0, s, recs, params, Z, out);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
