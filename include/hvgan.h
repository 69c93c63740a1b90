/* hvgan.h -- C ABI of libhvgan.so: hand-written gfx950 (MI355X) kernels for the HealthiVert-GAN
 * generator / discriminator train + inference hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_*; every tensor is NHWC, dense in W/H with an
 *     explicit channel stride `*_ld` (elements per pixel) and channel offset `*_coff` so a conv can read or
 *     write a channel slice of a wider (concat) buffer.  Elements are fp32; the activation / gradient tensors INSIDE
 *     a network may be stored as fp16 instead where an `*_f16` flag says so (precision HV_F16 only: their values
 *     are rounded to fp16 as MFMA operands anyway).  A (B,1,H,W) NCHW tensor is bit-identical in NHWC, so all
 *     image-level inputs/outputs of the reference API (always fp32) cross the boundary without a copy.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No entry point allocates, frees or
 *     synchronises: the caller owns every buffer, including workspaces sized by the *_workspace_bytes calls,
 *     so every call is legal inside a hipGraph capture.
 *   - return value: 0 = ok; HV_ERR_* (negative) = rejected before launch; <= -1000 = -(hipError_t) - 1000.
 *     Nothing throws across the ABI.  The Python host (healthivert-gan_amd/lib.py) turns non-zero into
 *     RuntimeError; there is no CPU fallback.
 *   - `precision`: HV_F32 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32; the |d| <= 1e-3 parity mode),
 *     HV_F16 = operands rounded to fp16 when staged into LDS, fp32 accumulate (v_mfma_f32_16x16x32_f16).
 *
 * Each entry point names the reference code it replaces (paths relative to the reference root).
 */
#ifndef HVGAN_H
#define HVGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HV_OK 0
#define HV_ERR_ARG (-1)         /* null pointer / non-positive size / bad enum */
#define HV_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define HV_ERR_WORKSPACE (-3)   /* workspace too small */

enum { HV_F32 = 0, HV_F16 = 1 };
enum { HV_ACT_NONE = 0, HV_ACT_ELU = 1, HV_ACT_RELU = 2, HV_ACT_LRELU = 3, HV_ACT_SIGMOID = 4, HV_ACT_CLAMP = 5 };
enum { HV_NORM_NONE = 0, HV_NORM_BATCH = 1, HV_NORM_INSTANCE = 2 };

int hv_version(void);
const char* hv_arch(void); /* "gfx950" */
/* which kernel family the calling thread's most recent hv_conv2d / hv_conv2d_wgrad launched (profiling labels):
 * 0 conv_igemm_kernel, 1 narrow_fwd_kernel, 2 conv_halo_kernel, 3 conv_halo2_kernel, 4 thin1_fwd_kernel, 5 head_gemm_kernel, 6 conv_s2t_kernel, 10 wgrad_kernel,
 * 11 wgrad_halo_kernel, 12 wgrad_tr_kernel, 13 wgrad_trd_kernel, 14 conv_px_kernel (one lane per pixel, 4x4x4 MFMA: the thin full-resolution layers) */
int hv_last_kernel_path(void);
/* name of that kernel instantiation as rocprofv3 prints it, e.g. "conv_halo2_kernel<8, 16, 128, 1, 4, 32, 1, 4, 4>" (the gather and weight-
 * gradient kernels are templated on _Float16, which rocprofv3 leaves mangled: _Z12wgrad_kernelIDF16_Li128E... = wgrad_kernel<_Float16, 128, ...>) */
const char* hv_last_kernel_name(void);
/* profiling: the calling thread's NEXT hv_conv2d or hv_conv2d_wgrad records these two hipEvent_t right around its kernel launch (for a weight
 * gradient: around the main kernel; the slab reduction that follows is a separate kernel); the pair is consumed by that call.  NULL, NULL cancels. */
int hv_set_kernel_timing(void* ev_start, void* ev_stop);
/* tuning (tests and A/B tools): which stride-2 data gradients take the fused-parity kernel (path 6): 0 none, 1 the 3x3 filters (default;
 * environment HV_S2T), 2 also the 4x4 filters.  Process-wide; returns the previous mode. */
int hv_set_s2t_mode(int mode);

/* ---------------------------------------------------------------- convolution (implicit GEMM on MFMA)
 * Replaces F.conv2d / F.conv_transpose2d as called by Conv2dBlock.forward (models/inpaint_networks.py:494-503),
 * NLayerDiscriminator (models/networks.py:575-602), Down/UpsampleBlock (models/UnetG_CT_mask.py:69-100), the
 * matching and pasting convolutions of ContextualAttention (models/inpaint_networks.py:348,379) and -- with
 * transposed=1 -- their autograd data-gradients.
 *
 *   y[n,ho,wo,co] (op)= act( alpha * ch_scale[co] * sum_{r,s,ci} x[n,hi,wi,ci] * w[co,(r,s),ci] + bias[co] )
 *   transposed=0: hi = ho*stride - pad + r*dil
 *   transposed=1: hi = (ho + pad - r*dil)/stride where divisible (gather form of conv_transpose2d / dgrad)
 *   in_shift=1 reads x through a fused nearest x2 upsample (physical index = logical >> 1).
 *   w is [Cout][KH*KW][Cin] ("OHWI"); w_bstride != 0 selects per-sample filters (contextual attention).
 */
typedef struct {
    const float* x; int B, H, W;      /* logical input size (after the fused upsample) */
    int in_shift; int x_ld, x_coff, Cin;
    const float* w; long long w_bstride;
    int Cout, KH, KW, stride, pad, dil, transposed;
    const float* bias;                /* [Cout] or NULL */
    const float* ch_scale;            /* [Cout] or NULL */
    long long ch_scale_bstride;
    float alpha; int act; int accumulate;   /* accumulate: 0 y = r; 1 y += r (after act); 2 y = act(r + y) */
    float* y; int Ho, Wo, y_ld, y_coff;
    int precision;
    const void* w_f16;                /* optional fp16 copy of w (same layout, from hv_weight_prep): enables the
                                         halo-tiled kernel for HV_F16, dilation 1, Cin % 16 == 0, shared filters */
    const void* w_f16_tiled;          /* optional: the same fp16 filters in MFMA-fragment order (hv_weight_prep's w_fwd_t / w_bwd_t, or
                                         hv_weight_tile_f16; needs w_f16 too).  The kernels that fetch filter rows straight into MFMA operand
                                         registers then read 1-KB contiguous pieces instead of 16 rows x 64 B that lie a whole filter row apart
                                         (and on one L2 channel): 256 -> 512 4x4 PatchGAN layer 93 -> 76 us.  NULL = plain rows */
    const float* mul_src; int mul_ld, mul_coff, mul_act;
                                      /* optional epilogue factor: r *= act'(m) with m = mul_src[pixel*mul_ld + mul_coff + channel] the OUTPUT
                                         of activation mul_act at the same pixel/channel, applied after act and before accumulate == 1.
                                         Used by data gradients to hand the producer layer its pre-activation gradient directly
                                         (the separate in-place multiply pass over the gradient disappears).  NULL = none */
    int x_f16, y_f16, mul_f16;        /* storage of x / y / mul_src: 0 = fp32 (pointers are float*), 1 = fp16 (the pointers address _Float16
                                         elements; *_ld / *_coff stay in elements).  fp16 storage needs precision == HV_F16 (operands are rounded
                                         to fp16 at staging anyway) and serves the tensors of a network that are only conv / norm operands;
                                         1-channel image tensors and the attention score matrices stay fp32 */
    void* workspace; size_t workspace_bytes;
                                      /* optional scratch (16-byte aligned) of hv_conv2d_workspace_bytes(d) bytes: enables the two-kernel path for
                                         Cout == 1 with many input channels (PatchGAN logits, data gradient of the 1-channel stem), which
                                         streams x once into a [pixel][tap] table and sums the taps afterwards.  NULL = other kernels */
    float* stats;                     /* optional: per-channel partial sums of the OUTPUT as stored (fp16-rounded, after act), written by the conv's own
                                         epilogue: stats[(part * Cout + c) * 2 + {0: sum, 1: sum of squares}] for part < hv_conv2d_stats_parts(d).
                                         Feeds hv_norm_desc.partials (BatchNorm statistics without a reduction pass over the tensor).  Only the
                                         kernels that hv_conv2d_stats_parts reports (> 0) write it; NULL = not wanted */
    const void* x1;                   /* optional ONE extra input channel at the convolution's own resolution, added before bias / activation:
                                         y += sum_taps x1[n][i + dh][j + dw] * w1[co * w1_row + tap * w1_tap] (zero padded like x).  With in_shift = 1 on x this
                                         is a convolution over the concatenation [nearest x2 up-sampling of x | x1] that is never materialised (the coarse
                                         generator's conv19 / conv20: models/inpaint_networks.py:97-106); w1 = channel Cin of the full fp32 forward table
                                         [Cout][taps][CinP] (w1_row = taps * CinP, w1_tap = CinP).  Forward 3x3 stride-1 only, filters-in-LDS kernel only:
                                         HV_ERR_UNSUPPORTED otherwise */
    int x1_f16, x1_ld, x1_coff;
    const float* w1; int w1_row, w1_tap;
    int pool2;                        /* 1: the output is stored 2x2 sum-pooled -- y, mul_src (and the accumulate operand) are (Ho/2) x (Wo/2) tensors and
                                         y[n][i][j][c] (+)= act'(mul_src[n][i][j][c]) * sum of the four fp16-rounded conv outputs at (2i + {0,1}, 2j + {0,1}):
                                         the data gradient of a convolution whose input was the nearest x2 up-sampling of a smaller tensor, without the
                                         full-resolution gradient in memory (replaces conv + hv_copy_channels mode 3 + the in-place act' pass).  Ho, Wo keep
                                         the convolution's own output size (even).  Served by the filters-in-LDS 3x3 kernel only: HV_ERR_UNSUPPORTED otherwise */
    const void* bn_x; int bn_x_ld, bn_x_coff;
    const float* bn_stats; int bn_groups;
    float* bstats;                    /* optional, data-gradient forms whose output y is the gradient at the OUTPUT of a batch normalisation (its act' applied through
                                         mul_src): the normalisation's backward sums out of this conv's epilogue.  bn_x = the normalisation's raw input (same pixels /
                                         channels / storage as y, its own ld / coff), bn_stats = its saved [bn_groups][2][Cout] mean / rstd (hv_norm_desc.stats; the
                                         batch's images fall into bn_groups equal groups), bstats[(part * Cout + c) * 2 + {0, 1}] = sum g, sum g * xhat over the part's
                                         pixels (g = the stored value of y, xhat = (bn_x - mean) * rstd), part < hv_conv2d_bstats_parts(d), parts of whole images in
                                         (group, image) order.  Feeds hv_norm_bwd_desc.partials: hv_norm_act_backward then skips its reduction pass over dy and x.
                                         hv_conv2d returns HV_ERR_UNSUPPORTED (and launches nothing) when the kernel of this shape has no such epilogue */
    const float* xn_stats; const float* xn_gamma; const float* xn_beta; int xn_groups, xn_act;
    void* xn_out; int xn_out_ld, xn_out_coff;
                                      /* optional, forward with Cout == 1 and `workspace` (the [pixel][tap] path of the PatchGAN logits layer) only: x is the RAW input of a
                                         normalisation + activation (models/networks.py:583-595) whose statistics hv_norm_act_forward (y == NULL: statistics only) left in
                                         xn_stats -- [xn_groups][2][Cin] mean / rstd, the batch's images falling into xn_groups equal groups (1 BatchNorm, 2 the fake | real
                                         pass, B InstanceNorm).  The operand is xn_act(((x - mean) * rstd) * gamma + beta) rounded to fp16 -- hv_norm_act_forward's
                                         arithmetic, bit for bit -- made where x is staged (xn_gamma / xn_beta NULL = 1 / 0); xn_out != NULL: the normalised map is also
                                         stored there (fp16 [pixel][xn_out_ld], the write a separate normalisation pass would have made; its read of x and this
                                         convolution's read of the result are what the fusion saves).  hv_conv2d returns HV_ERR_UNSUPPORTED (and launches nothing) when the
                                         kernel of this shape has no such staging */
} hv_conv_desc;
int hv_conv2d(const hv_conv_desc* d, void* stream);
size_t hv_conv2d_bstats_parts(const hv_conv_desc* d);      /* parts of hv_conv_desc.bstats this call would write (0: its kernel has no such epilogue) */
size_t hv_conv2d_workspace_bytes(const hv_conv_desc* d);   /* 0 when no kernel for this shape wants scratch */
size_t hv_conv2d_stats_parts(const hv_conv_desc* d);       /* parts of hv_conv_desc.stats this call would write (0: its kernel has no statistics epilogue) */
int hv_last_weight_tables(void);                           /* which prepared tables the hv_conv2d call this thread made last read: 1 = fp32 (w, w1), 2 = fp16 rows
                                                              (w_f16), 4 = fp16 in MFMA-fragment order (w_f16_tiled); may over-report.  A caller that collects
                                                              this per layer can pass NULL for the unread tables of hv_wprep_layer (hv_weight_prep2) */
int hv_conv2d_supported(const hv_conv_desc* d);            /* 1 when hv_conv2d would serve d.  Only the forms without a generic fallback can be refused for their
                                                              shape -- x1, pool2, stats (hv_conv2d then returns HV_ERR_UNSUPPORTED and launches nothing) -- so a caller
                                                              asks before it drops the materialised alternative.  Runs the dispatch itself without launching */

/* One slab fold: dw[i] (+)= sum_k slabs[k * numel + i] in a fixed order (groups by slab count, then a tree: deterministic), dbias likewise from bias_slabs
 * [nslabs][Cout].  nslabs == 0: the call wrote dw directly (nothing to fold). */
typedef struct {
    const float* slabs; float* dw; long long numel; int nslabs; int accumulate;
    const float* bias_slabs; float* dbias; int Cout; int dbias_accumulate;
} hv_wgrad_fold;

/* Weight gradient: dw[co][(r,s)][ci] = sum_{n,ho,wo} g[n,ho,wo,co] * x[n, ho*stride-pad+r*dil, ..., ci].
 * (autograd of the convs above; for a transposed conv swap the roles of x and g on the caller side).
 * Split over pixel chunks; partial slabs go to `workspace` and are summed in a fixed order (deterministic). */
typedef struct {
    const float* x; int B, H, W; int in_shift; int x_ld, x_coff, Cin;
    const float* g; int Ho, Wo, g_ld, g_coff, Cout;
    int KH, KW, stride, pad, dil;
    float* dw; int accumulate;
    float* workspace; size_t workspace_bytes;
    int precision;
    int x_f16, g_f16;                     /* storage of x / g as in hv_conv_desc (both 0 or both 1) */
    float* dbias; int dbias_accumulate;   /* optional: dbias[co] (+)= sum over pixels of g[.., co] (bias gradient of the same conv), folded
                                             into the kernel that already streams g; NULL = not computed */
    hv_wgrad_fold* pending;               /* optional (HOST pointer): the split-K slabs are LEFT in `workspace` -- which then has to stay untouched until their
                                             fold has run -- and *pending describes that fold; the call launches no slab reduction.  The caller hands the record to
                                             the NEXT weight gradient of the same stream as `carry` (or to hv_wgrad_fold_now).  NULL = fold here */
    const hv_wgrad_fold* carry;           /* optional (HOST pointer): the PREVIOUS weight gradient's recorded fold (another dw, another workspace): it runs as extra
                                             workgroups of this call's main kernel where that kernel leaves room for them (the generators' 3x3 / 5x5 layers),
                                             else as a launch of its own; either way it is done when this call's work is.  Same sums bit for bit as a fold
                                             launched on its own.  One dependent ~5-us launch less per layer of a backward chain */
} hv_wgrad_desc;
int hv_wgrad_fold_now(const hv_wgrad_fold* fold /* host */, void* stream);     /* the fold of a `pending` record as a launch of its own (the last of a chain) */
size_t hv_conv2d_wgrad_workspace_bytes(const hv_wgrad_desc* d);
int hv_conv2d_wgrad(const hv_wgrad_desc* d, void* stream);

/* ---------------------------------------------------------------- weight preparation / spectral norm
 * Batched over layers: one workgroup per layer.  Replaces torch.nn.utils.spectral_norm's forward pre-hook
 * (used at models/inpaint_networks.py:449-450,491-492): optional power iteration (in-place u,v), sigma,
 * W/sigma, written directly in the kernels' OHWI layouts (forward: [Cout][taps][CinP]; data-gradient:
 * [CinP'][taps][Cout'] ...).  With sn=0 it is a pure layout transform (discriminator / U-Net weights). */
typedef struct {
    const float* w_orig;  /* [Cout][Cin][KH][KW] (torch layout); for conv_transpose [Cin][Cout][KH][KW] with transposed_src=1 */
    float* u; float* v;   /* [Cout], [Cin*KH*KW]; NULL when sn=0 */
    float* sigma;         /* [4] scratch: [0] = sigma out (1.0 when sn=0), [1] = <dWsn,Wsn> written by the backward */
    float* w_fwd;         /* [CoutF][taps][CinP]  rows >= Cout and channels >= Cin are zero; NULL = not wanted (hv_weight_prep2 only: every table is optional there) */
    float* w_bwd;         /* [CinB][taps][CoutP]  or NULL */
    void* w_fwd_h; void* w_bwd_h;   /* optional fp16 copies of w_fwd / w_bwd (same layouts) or NULL */
    void* w_fwd_t; void* w_bwd_t;   /* optional fp16 copies in MFMA-fragment order (hv_conv_desc.w_f16_tiled; layout at hv_weight_tile_f16) or NULL;
                                       hv_weight_tiled_elems(CoutF, taps, CinP) / (CinB, taps, CoutP) halfs, zero-filled once by the caller */
    int Cout, Cin, taps, CinP, CoutF, CoutP, CinB;
    int sn, power_iter, transposed_src;
    void* w_fwd_t2; int K2;   /* optional: the first K2 (32 or 64) input channels of the forward table once more in MFMA-fragment order, as a table of its
                                 own ([CoutF][taps][K2]: hv_weight_tiled_elems(CoutF, taps, K2) halfs) -- the filters of hv_conv_desc.x1 layers; written by
                                 hv_weight_prep2 only */
} hv_wprep_layer;
int hv_weight_prep(const hv_wprep_layer* d_layers, int n_layers, long long max_numel, void* stream); /* d_layers: DEVICE array; max_numel = largest w_fwd+w_bwd element count of a layer */
/* The same tables with all six of a layer written from one read of its weights (32 x 32 filter x channel tiles through LDS, whole MFMA fragments per store).
 * any_sn: some layer has sn = 1 (otherwise the sigma kernel is not launched: sigma[0] must already hold 1.0); any_legacy: some layer is a conv_transpose
 * source (those take the element-wise kernels of hv_weight_prep). */
int hv_weight_prep2(const hv_wprep_layer* d_layers, int n_layers, long long max_numel, int any_sn, int any_legacy, void* stream);

/* MFMA-fragment order of an fp16 filter table w[rows][taps][K] (K = padded input channels; T = 32 when K % 32 == 0, 16 when K % 16 == 0, otherwise
 * there is no tiled form): element (row, tap, k) lives at half index
 *     (row / 16) * 16 * taps * K  +  ((tap * K + k) / T) * 16 * T  +  (((k % T) / (T/4)) * 16 + row % 16) * (T/4)  +  k % (T/4)
 * i.e. per 16-row block one contiguous 16 x T fragment per (tap, T-channel chunk), stored so that lane l of a wave reads bytes [l*T/2, (l+1)*T/2)
 * of it as its `v_mfma_f32_16x16x{32,16}_f16` A operand.  Rows are padded to a multiple of 16 with zeros.
 * hv_weight_tiled_elems: halfs the tiled table takes (0: no tiled form).  hv_weight_tile_f16: plain -> tiled (writes the zero rows too). */
size_t hv_weight_tiled_elems(int rows, int taps, int K);
int hv_weight_tile_f16(const void* w_f16, void* w_tiled, int rows, int taps, int K, void* stream);

/* Backward of the above: dw_orig = (dWsn - <dWsn, Wsn> u v^T) / sigma  (sn=1) or a layout transform (sn=0).
 * dw_ohwi is the hv_conv2d_wgrad output [Cout][taps][CinP]. */
typedef struct {
    const float* dw_ohwi; const float* w_fwd; const float* u; const float* v; const float* sigma;
    float* dw_orig; int Cout, Cin, taps, CinP, sn, transposed_src, accumulate;
} hv_wprep_bwd_layer;
int hv_weight_prep_backward(const hv_wprep_bwd_layer* d_layers, int n_layers, long long max_numel, int any_sn, void* stream);

/* 1-channel heads (models/inpaint_networks.py:115,230: clamp / sigmoid heads): loss seed -> the head's gradient carrier in one pass.
 * carrier_f16[p][0] = seed[p] * act'(y[p]) (fp16; channels 1-3 of the [pixel][4] carrier are zeroed), dbias (+)= sum_p of the stored values.
 * workspace: hv_head_seed_workspace_bytes(npix) bytes when dbias != NULL. */
size_t hv_head_seed_workspace_bytes(long long npix);
int hv_head_seed_backward(const float* seed, const void* y, int y_f16, int y_ld, int y_coff, void* carrier_f16, long long npix, int act, float* dbias,
                          int dbias_accumulate, float* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- activation gradient + bias gradient
 * g[p,c] = dy[p,c] * act'(y[p,c]) in place over dy; dbias[c] (+)= sum_p g[p,c] when dbias != NULL.
 * (autograd of nn.ELU/ReLU/Sigmoid/clamp after the conv, models/inpaint_networks.py:459-474,115,230). */
int hv_act_backward(void* dy, int dy_f16, const void* y, int y_f16 /* storage of dy / y: 0 fp32, 1 fp16 */, long long npix, int C, int dy_ld, int dy_coff,
                    int y_ld, int y_coff, int act, float* dbias, int dbias_accumulate, float* workspace, size_t workspace_bytes, void* stream);
size_t hv_act_backward_workspace_bytes(long long npix, int C);

/* ---------------------------------------------------------------- normalisation + activation (discriminator, U-Net)
 * BatchNorm2d (train: batch statistics + running-stat update, eval: running stats) or InstanceNorm2d (no affine)
 * followed by LeakyReLU(0.2)/ReLU/none (+ optional sigmoid), models/networks.py:18-36,583-595 and
 * models/UnetG_CT_mask.py:73-100.  `stats` (2*G*C floats: mean, rstd; G = 1 for batch, B for instance) is kept
 * for the backward. */
typedef struct {
    const void* x; void* y; int B, HW, C; int x_ld, x_coff, y_ld, y_coff;      /* x, y: fp32, or fp16 elements when f16 != 0.  y == NULL: statistics only (stats and
                                                                                  the running statistics are written, no pass over x beyond the reduction): the
                                                                                  consumer normalises at its own staging (hv_conv_desc.xn_stats) */
    int norm; int training; float eps, momentum;
    const float* gamma; const float* beta; float* running_mean; float* running_var; long long* num_batches_tracked;
    float* stats; int act; int post_sigmoid;
    float* workspace; size_t workspace_bytes;
    int groups;   /* batch norm: the batch is split into `groups` equal parts with separate statistics, running stats updated
                     part by part (fake | real halves of one discriminator launch == two consecutive calls); 0/1 = one group */
    int f16;      /* storage of x and y (statistics, affine parameters and all arithmetic stay fp32 / fp64) */
    const float* partials; int n_partials;
                  /* optional (batch norm, training): the producing conv's hv_conv_desc.stats -- [n_partials][C][2] partial sums of x, parts of whole
                     images in image order (groups > 1: group g is the g-th of `groups` equal ranges of them; n_partials must divide accordingly, else x is reduced here).
                     The reduction pass over x is skipped; the partials are folded in double precision in a fixed order.  NULL = reduce x here */
} hv_norm_desc;
size_t hv_norm_workspace_bytes(int B, int HW, int C);
int hv_norm_act_forward(const hv_norm_desc* d, void* stream);
/* dx from dy (gradient wrt the activation output y); dgamma/dbeta (+)= when non-NULL.  act == HV_ACT_NONE (and no post_sigmoid): dy is the
 * gradient at the normalisation's own output -- the consumer's data-gradient epilogue already applied act'(y) through hv_conv_desc.mul_src -- and
 * y is never read (may be NULL): one tensor less in both passes. */
typedef struct {
    const void* dy; const void* y; const void* x; void* dx; int B, HW, C;     /* fp32, or fp16 elements when f16 != 0 (all four alike) */
    int dy_ld, dy_coff, y_ld, y_coff, x_ld, x_coff, dx_ld, dx_coff;
    int norm; int training; const float* gamma; const float* stats;
    int act; int post_sigmoid;
    float* dgamma; float* dbeta; int param_accumulate;
    float* workspace; size_t workspace_bytes;
    int groups;
    int f16;
    const float* partials; int n_partials;
                  /* optional (batch norm, training, act == HV_ACT_NONE): hv_conv_desc.bstats of the data gradient that produced dy -- [n_partials][C][2] sums
                     (sum dy, sum dy * xhat), parts of whole images in (group, image) order, n_partials a multiple of `groups`.  The reduction pass over dy and x
                     is skipped.  NULL = reduce here */
} hv_norm_bwd_desc;
int hv_norm_act_backward(const hv_norm_bwd_desc* d, void* stream);

/* ---------------------------------------------------------------- layout / resampling helpers
 * NCHW <-> NHWC (API edge only), nearest x2 upsample + channel concat (F.interpolate + torch.cat,
 * models/inpaint_networks.py:97-99,105-106), channel copies, zero fill. */
int hv_nchw_to_nhwc(const float* src, void* dst, int dst_f16, int B, int C, int H, int W, int dst_ld, int dst_coff, void* stream);
int hv_nhwc_to_nchw(const void* src, int src_f16, float* dst, int B, int C, int H, int W, int src_ld, int src_coff, int accumulate, void* stream);
/* Copies C channels into a channel slice of dst; H,W are the dst size.  mode 0: same size; 1: src is half size
 * (nearest x2 upsample); 2: src is double size (nearest x1/2 downsample, even indices); 3: dst(half) (+)= sum of the
 * 2x2 block of src(full) [adjoint of 1]; 4: dst(full) (+)= src(half) at even indices, 0 elsewhere [adjoint of 2]. */
int hv_copy_channels(const void* src, int src_f16, void* dst, int dst_f16, int B, int H, int W, int C, int src_ld, int src_coff, int dst_ld,
                     int dst_coff, int mode, int accumulate, void* stream);   /* *_f16: storage of src / dst (may differ: the copy converts) */
/* dst = a + b over C channels of same-size [npix] tensors, each with its own storage / channel stride / offset: a gradient with two contributions
 * (models/pix2pix_model.py:310-330 backpropagates x_stage1 and coarse_seg through a loss AND through the refinement generator) in one pass. */
int hv_add_channels(const void* a, int a_f16, int a_ld, int a_coff, const void* b, int b_f16, int b_ld, int b_coff, void* dst, int dst_f16, int dst_ld,
                    int dst_coff, long long npix, int C, void* stream);

/* ---------------------------------------------------------------- generator heads and inputs
 * cat[x, ratio-plane, mask] / cat[x, coarse_seg, mask, ratio-plane] (models/inpaint_networks.py:71-77,173-179)
 * written as a CP-channel NHWC buffer (pad channels zero).  order: 0 = coarse, 1 = fine. */
int hv_gen_input(const float* x, const float* seg, const float* mask, const double* slice_ratio, void* dst, int dst_f16,
                 int B, int H, int W, int CP, int order, void* stream);
/* AdaptiveAvgPool2d(1) -> Linear(C,1) -> sigmoid (models/inpaint_networks.py:90-93,211-214). */
size_t hv_gap_fc_workspace_bytes(int B, int C);
int hv_gap_fc_sigmoid(const void* x, int x_f16, int B, int HW, int C, int x_ld, const float* fc_w, const float* fc_b,
                      float* pooled /*[B][C]*/, float* pred /*[B]*/, float* workspace, size_t workspace_bytes, void* stream);
/* backward: dx[n,p,c] += dpred[n]*pred(1-pred)*w[c]/HW [* act'(mul_src[n,p,c]) when mul_src: dx then holds the gradient wrt the PRE-activation of the
 * pooled tensor, like hv_conv_desc.mul_src]; dw[c] (+)= sum_n dl_n*pooled[n,c]; db (+)= sum_n dl_n */
int hv_gap_fc_sigmoid_backward(const float* dpred, const float* pred, const float* pooled, const float* fc_w,
                               void* dx, int dx_f16, int B, int HW, int C, int dx_ld, float* dw, float* db, int accumulate,
                               const void* mul_src, int mul_f16, int mul_ld, int mul_act, void* stream);

/* ---------------------------------------------------------------- contextual attention pieces
 * (models/inpaint_networks.py:247-410).  The two big contractions (patch matching :348, patch pasting :379) and
 * their gradients go through hv_conv2d with per-sample filters; these are the memory-bound stages around them.
 * Score matrices are [b][p][l]: p = foreground position, l = background patch (contiguous), L = (H/2)*(W/2). */
/* x1/2 nearest downsample fd[B][h][w][C]; 3x3 zero-padded patches wp[B][L][9][C] (+ transposed wpT[B][9*C][L]);
 * norm[B][L] = max(||patch||, 1e-4) and rnorm = 1/norm  (:282-294, :341-345). */
int hv_ca_patches(const void* f, int f_f16 /* storage of f: 0 fp32, 1 fp16 */, int B, int H, int W, int C, int f_ld, float* fd, float* wp, float* wpT, float* norm,
                  float* rnorm, void* stream);
/* 4x4 stride-2 'same' patches of a full-resolution map (:270-277): raw[B][L][16][C] and/or rawT[B][C][16][L]. */
int hv_ca_raw_patches(const float* f, int B, int H, int W, int C, int f_ld, float* raw, float* rawT, void* stream);
/* mm[l] = 1 iff the 3x3 patch of the x1/8-downsampled mask of SAMPLE 0 is all zero (:304-317). */
int hv_ca_mask(const float* mask, int Himg, int Wimg, int h, int w, float* mm, void* stream);
/* the same for every sample's own mask: mm[B][h*w] -- a batch that stands for B independent single-sample calls (the reference's inference
 * loop runs the generator at batch 1, eval_3d_sagittal_twostage.py:101), not for one training batch (which shares sample 0's mask) */
int hv_ca_mask_batched(const float* mask, int B, long long mask_bstride, int Himg, int Wimg, int h, int w, float* mm, void* stream);
/* score fusion (two diagonal 3-tap sums with the (h,w)<->(w,h) transposes, :352-361); adjoint=1 applies the
 * transposed operator (backward).  h == w required. */
int hv_ca_fuse(const float* S, float* out, int B, int h, int w, int adjoint, void* stream);
/* A[b][p][l] = softmax_l(S*mm*scale)*mm (:364-366); optional argmax over l -> argmax[b*L+p] (:368). */
int hv_ca_softmax(const float* S, const float* mm, float* A, int B, int L, float scale, int* argmax, void* stream);
int hv_ca_softmax_batched(const float* S, const float* mm, long long mm_bstride, float* A, int B, int L, float scale, int* argmax,
                          void* stream);   /* mm_bstride = L: per-sample masks from hv_ca_mask_batched; 0: shared */
/* offset_flow of the reference's 7-tuple (:368,:389-410 + inpaint_tools.flow_to_image/compute_color :73-100,181-211): the arg-max offsets
 * coloured with the Middlebury wheel (double precision, running maximum radius over samples 0..b like the reference's batch loop), as
 * uint8/255 and nearest-upsampled x`up` (rate*4): flow[B][3][h*up][w*up] (NCHW like the reference's tensor). */
int hv_ca_flow(const int* argmax, int B, int h, int w, int up, float* flow, void* stream);
int hv_ca_softmax_backward(const float* dA, const float* A, const float* mm, float* dS, int B, int L, float scale, void* stream);
int hv_transpose_batched(const float* src, float* dst, int B, int R, int C, void* stream); /* dst[b][c][r] = src[b][r][c] */
/* Batched "NT" matrix product on fp16 MFMA (fp32 accumulation and result; A / B are fp32 -- converted when staged -- or, with a_f16 / b_f16, fp16
 * tables such as hv_ca_raw_patches_f16 / hv_transpose_batched_f16 write: half the operand bytes through the vector memory path, which bounds the
 * fp32-operand form):
 *   C[b][m][n] = alpha * colscale[b][n] * sum_k A[b][m][k] * B[b][n][k]        (colscale may be NULL; strides in elements)
 * The fp16 mode's route for ContextualAttention's five big contractions (scores, paste, and their three gradients): they are plain products of
 * per-sample matrices that exist with the contraction index contiguous (csrc/bgemm.hip lists them).  K % 32 == 0, N % 4 == 0, 16-byte aligned rows.
 * b_split > 0 reads B's rows in another order: logical row n = t * b_split + c is stored as row c * (N / b_split) + t (the [channel][tap] patch
 * tables consumed in [tap][channel] order, so that C's rows hold whole channel vectors per tap).
 * hv_ca_fold: col2im of 4x4 stride-2 pad-1 patches src[b][p][tap][c] into dst[b][y][x][c] (+)= alpha * (the 4 taps reaching the pixel) -- the tail of
 * F.conv_transpose2d(..., stride=2, padding=1) (models/inpaint_networks.py:379) after the contraction, and the adjoint of hv_ca_raw_patches. */
int hv_bgemm_nt(const void* A, int a_f16, int lda, long long strideA, const void* B, int b_f16, int ldb, long long strideB, float* C, int ldc,
                long long strideC, int M, int N, int K, int batch, float alpha, const float* colscale, long long strideS, int b_split, void* stream);
int hv_ca_fold(const float* src, void* dst, int dst_f16 /* storage of dst */, int B, int H, int W, int C, int dst_ld, float alpha, int accumulate, void* stream);
/* the same pair with the product stored as fp16 (it only feeds the fold, whose result is stored as fp16 as well): half the bytes written and read */
int hv_bgemm_nt_h(const void* A, int a_f16, int lda, long long strideA, const void* B, int b_f16, int ldb, long long strideB, void* C_h, int ldc,
                  long long strideC, int M, int N, int K, int batch, float alpha, const float* colscale, long long strideS, int b_split, void* stream);
int hv_ca_fold_h(const void* src_h, void* dst, int dst_f16, int B, int H, int W, int C, int dst_ld, float alpha, int accumulate, void* stream);
/* fp16-stored forms of two operand producers (same layouts as hv_ca_raw_patches / hv_transpose_batched) */
int hv_ca_raw_patches_f16(const void* f, int f_f16 /* storage of f */, int B, int H, int W, int C, int f_ld, void* raw_h, void* rawT_h, void* stream);
int hv_transpose_batched_f16(const float* src, void* dst_h, int B, int R, int C, void* stream);
/* The GEMM route's fp16 operand copies written by their producers (round 3): hv_ca_patches_h = hv_ca_patches that also stores the [b][l][tap][c] patch
 * table as fp16 (wp_h: both operands of the score GEMM, same rounding as the GEMM's own staging of an fp32 operand -> same bits);
 * hv_ca_softmax_f16 = hv_ca_softmax(_batched) with the attention matrix stored as fp16 only (its readers are the paste GEMM, the transpose for the
 * d(raw patches) GEMM and the soft-max backward; models/inpaint_networks.py:364-381); hv_ca_softmax_backward_f16 reads that matrix;
 * hv_transpose_batched_h2h: fp16 -> fp16 (exact). */
int hv_ca_patches_h(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, float* fd, float* wp, void* wp_h, float* norm, float* rnorm,
                    void* stream);
int hv_ca_softmax_f16(const float* S, const float* mm, long long mm_bstride /* L: per-sample masks, 0: shared */, void* A_h, int B, int L, float scale,
                      int* argmax, void* stream);
int hv_ca_softmax_backward_f16(const float* dA, const void* A_h, const float* mm, float* dS, int B, int L, float scale, void* stream);
int hv_transpose_batched_h2h(const void* src_h, void* dst_h, int B, int R, int C, void* stream);
/* Gs[b][i][j] = dS[b][j][i]*rnorm[b][i] + dS[b][i][j]*rnorm[b][j];  coef[b][l] = -(sum_p dS[p][l]*S0[p][l])/norm[l]^2.
 * coef must hold 17*B*L floats: the first B*L are the result, the rest is scratch for the row-chunk partial sums. */
int hv_ca_score_backward_prep(const float* dS, const float* S0, const float* norm, const float* rnorm, float* Gs, float* coef,
                              int B, int L, void* stream);
/* hv_ca_fuse(adjoint) + hv_ca_score_backward_prep in one pass for the 32 x 32 attention map (csrc/attention.hip ca_fuse_adj_prep32_kernel): dS1 is the gradient
 * of the FUSED scores; the gradient of the plain scores stays on chip.  Gs has the bits of the two-call route, coef its value in a different summation order
 * (32 row blocks).  coef must hold 33*B*L floats.  HV_ERR_UNSUPPORTED unless h = w = 32 (the caller keeps the two calls for other maps). */
int hv_ca_fuse_backward_prep(const float* dS1, const float* S0, const float* norm, const float* rnorm, float* Gs, float* coef, int B, int h, int w,
                             void* stream);
/* col2im of (dwp + coef*wp) back to the even positions of the full-resolution map (adjoint of hv_ca_patches). */
int hv_ca_patches_backward(const float* dwp, const float* wp, const float* coef, float* df, int B, int H, int W, int C,
                           int df_ld, int accumulate, void* stream);

/* The matching scores and their gradient on the PIXEL Gram matrix (csrc/attention_gram.hip; fp16 mode, C = 64, attention map width 32 or 64): the 3x3
 * patches are both filters and inputs of models/inpaint_networks.py:327-344, so S0[p][l] = rnorm[l] * sum over the 3x3 offsets t of <fd[p + t], fd[l + t]>
 * (K = C instead of 9 C; no patch tables), and d fd = box(Gs) fd + (3x3 sum of coef) fd with Gs / coef from hv_ca_score_backward_prep.
 * hv_ca_gram_down: nearest 1/2 downsampling of the fp16 map f [B][H][W][f_ld] -> fd_h [B][L][C], fdT_h [B][C][L] (fp16), q [B][L] = |fd|^2.
 * hv_ca_gram_scores: S0 [B][L][L] (row p, column l), norm / rnorm [B][L] (= hv_ca_patches' outputs).
 * hv_ca_gram_backward: df[b][2y][2x][c] += the gradient wrt fd (replaces the L x 9C gradient GEMM + hv_ca_patches_backward).
 * HV_ERR_UNSUPPORTED for other shapes (the caller keeps the patch-table route). */
int hv_ca_gram_down(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, void* fd_h, void* fdT_h, float* q, void* stream);
int hv_ca_gram_scores(const void* fd_h, const float* q, int B, int h, int w, int C, float* S0, float* norm, float* rnorm, void* stream);
int hv_ca_gram_backward(const float* Gs, const void* fd_h, const void* fdT_h, const float* coef, int B, int h, int w, int C, float* df, int df_ld,
                        void* stream);

/* ---------------------------------------------------------------- step-level fused operators (Pix2PixModel)
 * Sobel edge magnitude (models/edge_operator.py:29-49). */
int hv_sobel(const float* img, float* out, int B, int H, int W, void* stream);
/* Everything Pix2PixModel.forward does after netG (models/pix2pix_model.py:191-264), one launch, no host sync:
 * pred_h = pred*maxheight; seg thresholds; SHRM compositing of fake_B / fake_B_coarse with per-sample row
 * bounds computed on the device; local crops; writes rows[B][4] = {xu2, xb2, xu1, xb1}. */
typedef struct {
    const float* real_B; const float* mask; const float* x_stage1; const float* x_stage2;
    const float* fine_seg; const float* coarse_seg; const float* pred1; const float* pred2;
    const long long* height; const long long* x1; const long long* x2; const long long* maxheight;
    float* fake_B; float* fake_B_coarse; float* fake_B_local; float* real_B_local;
    float* fine_bin; float* coarse_bin; float* pred1_h; float* pred2_h; int* rows;
    int B, H, W, half_band;
} hv_postg_desc;
int hv_post_generator(const hv_postg_desc* d, void* stream);
/* Generic SHRM compositing of one image set (train.py:81-99, eval_3d_sagittal_twostage.py:103-130). */
int hv_shrm_composite(const float* gen, const float* real, const float* pred_scaled, const long long* height,
                      const long long* x1, const long long* x2, float* out, int* rows, int B, int H, int W, void* stream);

/* GAN loss on PatchGAN logits (models/networks.py:212-278): loss (+)= weight*mean(l(z,t)); dz = weight_grad * dl/dz / n
 * mode 0 vanilla (BCE with logits), 1 lsgan (MSE). loss may be NULL (gradient only) and dz may be NULL. */
int hv_gan_loss(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate,
                float grad_weight, float* dz, void* stream);
/* the same over many workgroups (the one-workgroup form needs ~19 us for the 14 400 PatchGAN logits): per-workgroup partial sums in the caller's
 * scratch (hv_gan_loss_workspace_bytes(n) bytes, 4-byte aligned), added in index order by a second stage -- deterministic */
size_t hv_gan_loss_workspace_bytes(long long n);
int hv_gan_loss_ws(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate,
                   float grad_weight, float* dz, void* workspace, size_t workspace_bytes, void* stream);

/* The PatchGAN loss head in two launches: hv_gan_loss_ws + d loss / d logit written into the logits layer's padded fp16 gradient carrier
 * (carrier_f16[i][0]; channels 1-3 of the [n][4] carrier zeroed) + the logits layer's bias gradient dbias[0] (+)= sum of the stored values.
 * dz (fp32) and loss, dbias are optional.  workspace: hv_gan_loss_head_workspace_bytes(n) bytes. */
size_t hv_gan_loss_head_workspace_bytes(long long n);
int hv_gan_loss_head(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate, float grad_weight,
                     float* dz, void* carrier_f16, float* dbias, int dbias_accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The same head for up to TWO logit ranges in ONE launch (the fake | real halves of a batched discriminator pass, models/pix2pix_model.py:272-283:
 * criterionGAN(pred_fake, False), criterionGAN(pred_real, True); or one range with n1 = 0): per range its own target and loss slot, the carrier written in place,
 * dbias[0] (+)= the sum of both ranges' stored gradients (range 0 first).  Each range's mean is over its own n.  The workgroup that finishes last folds the block
 * sums in block order (deterministic): four tiny dependent launches between a discriminator's forward and backward become one.
 * workspace: hv_gan_loss_head_pair_workspace_bytes(n0, n1) bytes of scratch; ticket: ONE zero-initialised unsigned that stays with the call site (the kernel
 * leaves it at zero; launches that share it must be ordered on one stream). */
size_t hv_gan_loss_head_pair_workspace_bytes(long long n0, long long n1);
int hv_gan_loss_head_pair(const float* z0, long long n0, int real0, float* loss0, void* carrier0_f16, const float* z1, long long n1, int real1, float* loss1,
                          void* carrier1_f16, int mode, float loss_weight, int loss_accumulate, float grad_weight, float* dbias, int dbias_accumulate,
                          void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream);

/* Generator losses and their gradient seeds (models/pix2pix_model.py:331-353): writes
 * losses[0..5] = {G_maskL1, G_Dice, coarse_Dice, edge, h, sum of those five} and the seeds
 * d_fake_B (L1 part), d_fake_B_coarse, d_fine_seg, d_coarse_seg, d_pred1 (raw sigmoid output), d_pred2. */
typedef struct {
    const float* fake_B; const float* fake_B_coarse; const float* real_B; const float* mask;
    const float* fine_seg; const float* coarse_seg; const float* real_B_mask; const float* normal_vert;
    const float* fake_edges; const float* real_edges; const float* pred1_h; const float* pred2_h;
    const long long* height; const long long* maxheight;
    float lambda_L1;
    float* losses; float* d_fake_B; float* d_fake_B_coarse; float* d_fine_seg; float* d_coarse_seg;
    float* d_pred1; float* d_pred2;
    int B, H, W;
    float* workspace; size_t workspace_bytes;
    float grad_scale;   /* the six gradient seeds are multiplied by this (0 = 1): the gradient (loss) scale of the fp16 storage mode, a power of two
                           removed again from the parameter gradients by the caller (the losses themselves are not scaled) */
    /* optional bookkeeping folded into the same launches (each was a 1-element kernel at the head of the generator backward):
       loss_G_GAN = gan_terms[0] + ... + gan_terms[n_gan_terms-1] (in this order), loss_G = losses[5] + loss_G_GAN;
       add_d_fake_B: d_fake_B[i] += add_d_fake_B[i] (the discriminator's gradient wrt the composited image) */
    const float* gan_terms; int n_gan_terms; float* loss_G_GAN; float* loss_G;
    const float* add_d_fake_B;
} hv_gloss_desc;
size_t hv_generator_losses_workspace_bytes(int B);
int hv_generator_losses(const hv_gloss_desc* d, void* stream);
/* Gradient of the compositing + local crop: d_gen[row] = (d_fake[row] + d_local[row]*mask*band) for xu<=row<xb else 0.
 * which: 0 -> rows[b][0..1] (stage 2), 1 -> rows[b][2..3] (stage 1). */
int hv_shrm_backward(const float* d_fake, const float* d_local, const float* mask, const int* rows, int which,
                     float* d_gen, int B, int H, int W, int half_band, int accumulate, void* stream);

/* ---------------------------------------------------------------- optimiser
 * Multi-tensor Adam, torch.optim.Adam semantics (models/pix2pix_model.py:127-130): one launch per optimiser.
 * `state[0]` holds the step count as float (incremented by the kernel), lr is read from d_lr[0] so a captured
 * graph follows the scheduler. */
typedef struct { float* p; const float* g; float* m; float* v; long long n; } hv_adam_tensor;
int hv_adam_step(const hv_adam_tensor* d_tensors, int n_tensors, long long max_numel, const float* d_lr, float beta1,
                 float beta2, float eps, float* d_step, void* stream);
/* The same behind an overflow guard (fp16 storage mode: scaled gradients may have overflowed to inf / nan in an fp16 gradient buffer):
 * flat_grad[n_grad] -- the network's flat gradient buffer, after any all-reduce -- is checked first and multiplied by grad_mul on the way (1 / the loss
 * scale, a power of two; 1 = left alone); if it holds a non-finite value the whole step is skipped (weights, moments and step count unchanged) and
 * d_state[2] counts it.  d_state: 8 floats {step count, scratch, skipped steps, scratch, ticket, -, -, -}, zero-initialised by the caller.  Two launches
 * (check + update); device-side only (no host read): safe inside a captured graph. */
int hv_adam_step_guarded(const hv_adam_tensor* d_tensors, int n_tensors, long long max_numel, const float* d_lr, float beta1,
                         float beta2, float eps, float* d_state, float* flat_grad, long long n_grad, float grad_mul, void* stream);

/* misc */
int hv_threshold(const float* x, float* y, long long n, float thr, float value, void* stream); /* y = x > thr ? value : 0 (torch.where(seg > 0.5, ...)) */
int hv_fill(float* p, long long n, float value, void* stream);
int hv_axpy(float* y, const float* x, long long n, float a, void* stream); /* y += a*x */
int hv_affine(float* y, const float* x, long long n, float a, float b, void* stream); /* y = a*x + b (e.g. 1 - CAM) */
int hv_mul(float* y, const float* x, long long n, void* stream);                  /* y *= x (pred_h = pred * maxheight) */
int hv_mul3(float* y, const float* x, const float* z, long long n, void* stream); /* y = (y*x)*z (mask * image * centre band, pix2pix_model.py:254-263) */

/* ---- in-training evaluation metrics (reference train.py:37-48 dice_score / iou_score, :101-141 per-sample body of evaluate_model;
 * SURVEY.md section 8f row f3).  All images (B,1,H,W) fp32: `inpainted` = the SHRM-composited stage-2 output (hv_shrm_composite with
 * pred_h = pred2*maxheight), gt = real_B, coarse_bin / fine_bin = (seg > 0.5) (hv_threshold), normal_vert / label / mask as loaded;
 * pred_h [B] fp32, height [B] int64.  out[B][5] = { SSIM(gt*mask, inpainted*mask; data_range = max(inpainted) - min(inpainted)),
 * PSNR(gt*mask, inpainted*mask; data_range = max(inpainted) - min(gt)), dice(coarse_bin, normal_vert), iou(fine_bin, label),
 * |pred_h - height| / height * 100 }.  SSIM / PSNR follow scikit-image 0.22's published algorithm (7x7 uniform window, sample
 * covariance, K1 0.01, K2 0.03, 3-pixel border cropped; float64 means) -- see csrc/eval_metrics.hip. */
size_t hv_eval_metrics_workspace_bytes(int B, int H, int W);
int hv_eval_metrics(const float* inpainted, const float* gt, const float* mask, const float* coarse_bin, const float* normal_vert,
                    const float* fine_bin, const float* label, const float* pred_h, const long long* height, int B, int H, int W, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- RHLV quantification (reference evaluation/RHLV_quantification.py:41-147,160-178) ----
 * fake / label: straightened label volumes of the generated and the original vertebra, element (h, w, z) at
 * base[h*stride_h + w*stride_w + z*stride_z] (strides in elements); dtype 0 = float32, 1 = uint8.  A voxel belongs to the vertebra if it
 * equals label_index (label_index < 0: if it is non-zero).  z_lo == INT_MIN: the slice range is derived from the original vertebra's
 * z-extent like the reference (centre = int(mean z), half-length = (max_z - min_z) // length_divisor, numpy slice rules), else [z_lo, z_hi).
 * out (14 doubles, device): all / pre / mid / post RHLV, relative height of the original, the eight mean heights
 * (all_f, all_l, pre_f, pre_l, mid_f, mid_l, post_f, post_l), and 1.0 / 0.0 = the original vertebra exists. */
size_t hv_rhlv_workspace_bytes(int W, int Z);
int hv_rhlv(const void* fake, const void* label, int dtype, long long stride_h, long long stride_w, long long stride_z, int H, int W, int Z,
            float label_index, int length_divisor, int z_lo, int z_hi, double height_threshold, double* out, void* workspace,
            size_t workspace_bytes, void* stream);

/* ---- batch assembly on the device (reference data/aligned_dataset.py:204-280, AlignedDataset.__getitem__) ----
 * One item = one sagittal slice of a vertebra volume whose four uint8 planes [H][W] are resident on the device: ct = ct_data.astype(uint8)
 * (:245), vert = component-filtered vertebra mask * 255 (:247-248), normal = the patient's normal vertebrae as 0/255 (:190-196), cam =
 * (CAM * 255).astype(uint8) (:167,250) -- the reference quantises per item, a resident volume is quantised once.  [min_x, max_x) is the
 * masked band (:214-226), x1 / x2 the vertebra's first / last row (:205).  Outputs are the six float32 planes [B][H][W] the loader
 * collates: A = norm(ct), B = norm(restack(ct)), A_mask = vert / 255, mask = band, normal_vert = restack(normal) / 255,
 * CAM = restack(cam) / 255, with restack(v)[r] = v[r + x1 - min_x] above the band, v[x2 + r - max_x] below it, 0 inside (:232-243),
 * norm(u) = (u / 255 - 0.5) / 0.5 (torchvision ToTensor + Normalize((0.5,), (0.5,)), float32 arithmetic, bit-identical).
 * Every source row an item addresses must lie in [0, H) (the reference raises a broadcasting error otherwise): HV_ERR_ARG is NOT
 * detected for device-side descriptors -- the host mirror validates before upload. */
typedef struct {
    const uint8_t* ct; const uint8_t* vert; const uint8_t* normal; const uint8_t* cam;
    int x1, x2, min_x, max_x;
} hv_assemble_item;
int hv_assemble_batch(const hv_assemble_item* d_items, int B, int H, int W, float* A, float* Bimg, float* A_mask, float* mask,
                      float* normal_vert, float* CAM, void* stream);   /* d_items: DEVICE array of B descriptors */

/* ---- slice preparation of the inference driver on the device (reference eval_3d_sagittal_twostage.py:15-30,46-98; 8f rows f1 / f2) ----
 * hv_slice_components: per slice s of label [S][H][W]: 8-connected components of (label == value), components with fewer than min_size
 * pixels dropped (remove_small_connected_components); stats[s] = {pixel count, first row, last row, sum of row indices} of what remains
 * (count 0: first row -1).  workspace: hv_slice_components_workspace_bytes(S, H, W) bytes.
 * hv_infer_prepare: from those stats the rows run_model works with (x1, x2, height; the 40-row window around the mean row when the vertebra
 * is taller than maxheight) and the generator's four input planes [S][1][H][W]: ct_masked / cam = uint8-quantised slice re-stacked around the
 * band [min_x, max_x), ori_ct = the quantised slice, mask = rows [min_x, max_x] -- ToTensor / Normalize applied.  ct / cam: [S][H][W] float32 in
 * [0, 256).  selected (or NULL = all): slices this stage runs on.  valid[s] = vertebra present and selected; slices that are not valid get
 * rows that keep the re-compositing in bounds (x1 = x2 = 0, height = H) and zero planes.
 * hv_select_slices: dst[s] = flag[s] ? src[s] : (keep_unflagged ? dst[s] : 0)  -- slices without the vertebra pass a stage unchanged.
 * hv_slice_components_u8: the same component filter on a uint8 mask plane [S][H][W] (pixels == value, 1..255) -- the loader's
 * remove_small_connected_components on a drawn slice (data/aligned_dataset.py:16-31,:186-188) for every slice of a resident volume at once;
 * filtered (NULL, or [S][H][W], may be `plane` itself) receives the mask without the dropped components; stats as above
 * (count / first row / last row feed the loader's slice acceptance test and band rows, aligned_dataset.py:126-146,:198-226).
 * hv_slice_count: count[s] = number of elements of slice s ([S][per_slice] floats) equal to value -- the `> 200 pixels of the neighbour on the
 * original labels` gate of the volume driver (eval_3d_sagittal_twostage.py:208,217). */
size_t hv_slice_components_workspace_bytes(int S, int H, int W);
int hv_slice_components(const float* label, int S, int H, int W, float value, int min_size, int* stats, void* workspace,
                        size_t workspace_bytes, void* stream);
int hv_infer_prepare(const float* ct, const float* cam, const int* stats, const int* selected, int S, int H, int W, int maxheight,
                     float* ct_masked, float* ori_ct, float* mask, float* cam_out, long long* x1, long long* x2, long long* height,
                     int* valid, void* stream);
int hv_select_slices(const int* flag, const float* src, float* dst, int S, long long per_slice, int keep_unflagged, void* stream);
int hv_slice_components_u8(const void* plane /* uint8 */, int S, int H, int W, int value, int min_size, int* stats, void* filtered /* uint8 */,
                           void* workspace, size_t workspace_bytes, void* stream);
int hv_slice_count(const float* label, int S, long long per_slice, float value, int* count, void* stream);

/* Volume intake / output of the inference driver (reference eval_3d_sagittal_twostage.py:186-197,:208,:217,:236-239): the float64 [H][W][Z] volumes
 * (z fastest, as nibabel hands them over) are uploaded as they lie and everything else happens on the device.
 * hv_volume_scan: counts[j * Z + z] = number of voxels of slice z equal to id_j, j = 0..2 (an unused id is passed as a negative number; no id may
 *   be 0) -- the vertebra's z-extent (`np.any(label == vert_id)` per slice, :186-190) and the two neighbours' `> 200 pixels` gates (:208,:217) from
 *   one pass over the label volume.  counts: 3 * Z ints, zeroed by the call.  Z <= 2048.
 * hv_volume_slices: out[s][p] = (float)vol[p * Z + z0 + s], s < S, p < HW: the z-range cut out, converted (C cast = numpy astype) and transposed
 *   to [S][H*W] float32 slices.
 * hv_volume_merge: out[p * Z + z] = (z0 <= z < z0 + S and flag[z - z0]) ? (double)src[(z - z0) * HW + p] : 0 for ALL z < Z: the float64 output
 *   volume as the reference builds it (np.zeros, then the processed slices, :236-239). */
int hv_volume_scan(const double* label, long long HW, int Z, double id0, double id1, double id2, int* counts, void* stream);
int hv_volume_slices(const double* vol, long long HW, int Z, int z0, int S, float* out, void* stream);
int hv_volume_merge(const float* src, const int* flag, long long HW, int Z, int z0, int S, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
