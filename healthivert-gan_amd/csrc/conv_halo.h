// Shared declarations of the halo-tiled convolution kernels (conv_halo.hip: weights staged through LDS;
// conv_halo2.hip: weights fetched straight into MFMA operand registers).
#pragma once
#include "hv_common.h"

typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define HV_OOB 0x80000000u       // beyond every descriptor range (tensors are < 2 GiB): the load returns zeros

struct HaloCls {
    int ph, pw, Hc, Wc, ntaps, tiles_x, tiles, t0;   // t0 = first tile index of the class in the grid
    int dh_min, dw_min, PH, PW;
    uint32_t taps[25];   // (dh-dh_min) | (dw-dw_min)<<8 | widx<<16   (conv_halo_kernel: at most 16, conv_halo2_kernel: up to 5x5)
};
struct HaloK {
    const void* x; const _Float16* w; const float* bias; void* y;      // x / y: fp32 or fp16 elements (x_half / y_half)
    int B, Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Cout, w_row, y_ld, y_coff, Ho, Wo;
    int bstep, boff, ostep;
    float alpha; int act, accumulate, vec_store, ncls;
    const void* mul_src; int mul_ld, mul_coff, mul_act, mul_vec;   // epilogue factor act'(mul_src[..]) or NULL; mul_vec: vector loads are aligned
    int x_half, y_half, mul_half;
    int ep16;  // conv_halo2_kernel: fp16 output tile handed through LDS and stored as 16-byte pieces (whole channel rows per pixel)
    const void* x1; int x1_half, x1_ld, x1_coff; const float* w1; int w1_row, w1_tap;   // conv_lf_kernel: one extra input channel (hv_conv_desc.x1) or NULL
    int pool2; // conv_lf_kernel: the output tile leaves 2x2 sum-pooled (hv_conv_desc.pool2); Ho / Wo are then the POOLED tensor's size
    int dil;   // conv_halo2_kernel: dilation d > 1 runs as d*d residue sub-grids (pixel step d), each an undilated conv; 1 otherwise
    const _Float16* wt; unsigned wt_bytes;   // the filters in MFMA-fragment order (hv_conv_desc.w_f16_tiled) or NULL; conv_halo2_kernel only
    unsigned x_bytes, w_bytes;   // buffer descriptor ranges
    HaloCls cls[4];
};

template <int CK> struct HFrag;
template <> struct HFrag<32> {
    typedef f16x8 V;
    static __device__ __forceinline__ V ld(const _Float16* p, int lane) { return *reinterpret_cast<const V*>(p + (lane >> 4) * 8); }
    static __device__ __forceinline__ f32x4 mma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct HFrag<16> {
    typedef f16x4v V;
    static __device__ __forceinline__ V ld(const _Float16* p, int lane) { return *reinterpret_cast<const V*>(p + (lane >> 4) * 4); }
    static __device__ __forceinline__ f32x4 mma(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};


// conv_halo2.hip: returns HV_ERR_UNSUPPORTED when no instantiation covers the shape (the caller falls back to conv_halo_kernel)
int hv_halo2_launch(HaloK& k, int TW, int KH, int KW, int maxpatch, hipStream_t s);
// conv_lf.hip: filters resident in LDS, 16x16-pixel tiles (3x3 stride-1 layers); HV_ERR_UNSUPPORTED -> the caller goes on to hv_halo2_launch
int hv_convlf_launch(HaloK& k, int KH, int KW, hipStream_t s);
