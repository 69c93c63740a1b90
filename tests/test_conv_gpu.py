"""hv_conv2d / hv_conv2d_wgrad (through the C ABI) against torch CPU fp32 convolutions.

Tolerances: HV_F32 path (exact fp32 MFMA) |d| <= 1e-4 * scale (far inside the 1e-3 north-star gate);
HV_F16 path (fp16 operands, fp32 accumulate) |d| <= 4e-3 * scale, reported only.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, CinP, Cout, k, stride, pad, dil, act
    (2, 32, 32, 16, 16, 16, 3, 1, 1, 1, 'elu'),
    (2, 32, 32, 3, 4, 16, 5, 1, 2, 1, 'elu'),
    (2, 32, 32, 16, 16, 32, 3, 2, 1, 1, 'elu'),
    (1, 64, 64, 64, 64, 64, 3, 1, 16, 16, 'elu'),
    (2, 32, 32, 64, 64, 64, 3, 1, 4, 4, 'relu'),
    (2, 32, 32, 1, 1, 64, 4, 2, 1, 1, 'lrelu'),
    (2, 32, 32, 8, 8, 1, 3, 1, 1, 1, 'sigmoid'),
    (2, 32, 32, 9, 12, 1, 3, 1, 1, 1, 'clamp'),
    (2, 9, 300, 12, 12, 1, 3, 1, 1, 1, 'sigmoid'),      # 1-channel head over two 256-pixel row segments (LDS-staged rows), ragged end
    (2, 32, 32, 65, 68, 64, 3, 1, 1, 1, 'elu'),
    (3, 31, 31, 128, 128, 256, 4, 1, 1, 1, 'none'),
    (2, 16, 16, 256, 256, 1, 4, 1, 1, 1, 'none'),
    (16, 64, 64, 64, 64, 128, 4, 2, 1, 1, 'none'),
]


def _mk(case, seed=0):
    B, H, W, Cin, CinP, Cout, k, s, p, d, act = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    return x, w, b


def _ref_act(y, act):
    return {'elu': F.elu, 'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.2), 'sigmoid': torch.sigmoid,
            'clamp': lambda t: t.clamp(-1, 1), 'none': lambda t: t}[act](y)


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-4), ('fp16', 4e-3)])
@pytest.mark.parametrize('case', CASES)
def test_conv_forward(case, prec, tol):
    from hvtest import to_act, from_act, ohwi, dev, maxerr, st
    from hvgan import ops
    B, H, W, Cin, CinP, Cout, k, s, p, d, act = case
    x, w, b = _mk(case)
    ref = _ref_act(F.conv2d(x, w, b, stride=s, padding=p, dilation=d), act)
    # fp16 mode: fp16 storage inside the networks; the 1-channel image ends (stem input, head outputs) stay fp32
    xa = to_act(x, CinP, dtype=st(prec) if Cin > 1 else torch.float32)
    xa = ops.Act(xa.t, CinP, 0) if CinP % 4 == 0 else xa
    Ho, Wo = ref.shape[2], ref.shape[3]
    ya = ops.Act.empty(B, Ho, Wo, Cout, dev(), dtype=st(prec) if Cout > 1 else torch.float32)
    ops.conv2d(xa, ohwi(w, CinP), ya, k, s, p, d, bias=b.to(dev()), act=act, precision=prec)
    torch.cuda.synchronize()
    err = maxerr(from_act(ya), ref)
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-4), ('fp16', 4e-3)])
@pytest.mark.parametrize('case', CASES)
def test_conv_dgrad_is_transposed_gather(case, prec, tol):
    """d/dx of conv == hv_conv2d(transposed=1) with the [Cin][taps][Cout] filter layout."""
    from hvtest import to_act, from_act, ohwi_T, dev, maxerr, st
    from hvgan import ops
    B, H, W, Cin, CinP, Cout, k, s, p, d, act = case
    x, w, b = _mk(case)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(g)
    CoutP = (Cout + 3) // 4 * 4
    ga = to_act(g, CoutP, dtype=st(prec))
    ga = ops.Act(ga.t, CoutP, 0)
    dxa = ops.Act.empty(B, H, W, Cin, dev(), dtype=st(prec) if Cin > 1 else torch.float32)
    ops.conv2d(ga, ohwi_T(w, CoutP), dxa, k, s, p, d, transposed=True, precision=prec)
    torch.cuda.synchronize()
    err = maxerr(from_act(dxa), x.grad)
    assert err <= tol * max(1.0, x.grad.abs().max().item()), err


@pytest.mark.parametrize('prec,tol', [('fp32', 2e-4), ('fp16', 6e-3)])
@pytest.mark.parametrize('case', CASES)
def test_conv_wgrad(case, prec, tol):
    from hvtest import to_act, ohwi, dev, maxerr, st
    from hvgan import ops
    B, H, W, Cin, CinP, Cout, k, s, p, d, act = case
    x, w, b = _mk(case)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(g)
    CinP4, CoutP = (CinP + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xa = to_act(x, CinP4, dtype=st(prec))
    xa = ops.Act(xa.t, CinP4, 0)
    ga = to_act(g, CoutP, dtype=st(prec))
    ga = ops.Act(ga.t, CoutP, 0)
    dw = torch.empty(CoutP, k * k, CinP4, device=dev())
    db = torch.full((CoutP,), 7.0, device=dev())
    ops.conv2d_wgrad(xa, ga, dw, k, s, p, d, precision=prec, dbias=db)          # bias gradient rides in the same kernels
    torch.cuda.synchronize()
    got = dw[:Cout, :, :Cin].cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    err = maxerr(got, w.grad)
    assert err <= tol * max(1.0, w.grad.abs().max().item()), err
    gsum = g.sum(dim=(0, 2, 3))
    assert maxerr(db[:Cout].cpu(), gsum) <= 1e-4 * max(1.0, g.abs().sum(dim=(0, 2, 3)).max().item()), (db[:Cout].cpu() - gsum).abs().max()
    ops.conv2d_wgrad(xa, ga, dw, k, s, p, d, precision=prec, accumulate=True, dbias=db, dbias_accumulate=True)
    torch.cuda.synchronize()
    assert maxerr(db[:Cout].cpu(), 2 * gsum) <= 2e-4 * max(1.0, g.abs().sum(dim=(0, 2, 3)).max().item())
    assert maxerr(dw[:Cout, :, :Cin].cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2), 2 * w.grad) <= 2 * tol * max(1.0, w.grad.abs().max().item())


HALO_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, act, in_shift
    (2, 32, 32, 16, 16, 3, 1, 1, 'elu', 0),
    (2, 40, 24, 32, 64, 3, 1, 1, 'elu', 0),
    (2, 32, 32, 64, 32, 3, 2, 1, 'elu', 0),
    (3, 31, 31, 128, 256, 4, 1, 1, 'none', 0),
    (2, 64, 64, 64, 128, 4, 2, 1, 'lrelu', 0),
    (2, 16, 16, 64, 32, 3, 1, 1, 'elu', 1),
    (16, 32, 32, 256, 512, 4, 1, 1, 'none', 0),
    (2, 30, 30, 48, 8, 3, 1, 1, 'sigmoid', 0),
    # weights-in-registers form (conv_halo2.hip): every instantiation, forward and data gradient
    (2, 32, 32, 32, 16, 3, 1, 1, 'elu', 0),
    (2, 32, 32, 32, 32, 3, 1, 1, 'elu', 0),
    (2, 32, 32, 64, 128, 3, 1, 1, 'none', 0),
    (16, 128, 128, 16, 16, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 16, 32, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 32, 16, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 32, 32, 3, 1, 1, 'none', 0),
    (16, 128, 128, 32, 64, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 64, 68, 3, 1, 1, 'none', 0),
    (4, 64, 64, 128, 256, 4, 2, 1, 'lrelu', 0),
    # ragged channel counts (16-channel chunks with range-checked lanes) and the 5x5 stems
    (16, 128, 128, 4, 16, 5, 1, 2, 'elu', 0),
    (16, 128, 128, 36, 32, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 68, 64, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 12, 16, 3, 1, 1, 'none', 0),
    (4, 64, 64, 16, 32, 3, 1, 1, 'elu', 0),
    (4, 64, 64, 20, 64, 3, 1, 1, 'elu', 0),
    (16, 128, 128, 64, 128, 4, 2, 1, 'lrelu', 0),
]


@pytest.mark.parametrize('case', HALO_CASES)
def test_conv_halo_tiled_fp16_forward_and_dgrad(case):
    """The halo-tiled fp16 kernel (taken when the fp16 weight copy is passed) against torch CPU fp32."""
    from hvtest import to_act, from_act, ohwi, ohwi_T, dev, maxerr, st
    from hvgan import ops
    B, H, W, Cin, Cout, k, s, p, act, shift = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if shift else x
    xin = xin.clone().requires_grad_(True)
    y0 = F.conv2d(xin, w, b, stride=s, padding=p)
    ref = _ref_act(y0, act)
    Ho, Wo = ref.shape[2], ref.shape[3]
    ya = ops.Act.empty(B, Ho, Wo, Cout, dev(), dtype=torch.float16)          # fp16 mode: fp16 storage
    wf = ohwi(w)
    ops.conv2d(to_act(x, dtype=torch.float16), wf, ya, k, s, p, 1, bias=b.to(dev()), act=act, in_shift=shift, precision='fp16', w_h=wf.half())
    torch.cuda.synchronize()
    err = maxerr(from_act(ya), ref.detach())
    assert err <= 4e-3 * max(1.0, ref.abs().max().item()), err
    # the same filters in MFMA-fragment order (hv_conv_desc.w_f16_tiled): same arithmetic in the same order, so the same bits
    wt = ops.tile_weights(wf.half(), Cout, k * k, Cin)
    assert (wt is None) == (Cin % 16 != 0)
    from hvgan import lib
    pa = lib.get().size('hv_last_kernel_path')
    if wt is not None:
        yt = ops.Act.empty(B, Ho, Wo, Cout, dev(), dtype=torch.float16)
        ops.conv2d(to_act(x, dtype=torch.float16), wf, yt, k, s, p, 1, bias=b.to(dev()), act=act, in_shift=shift, precision='fp16', w_h=wf.half(), w_t=wt)
        torch.cuda.synchronize()
        if lib.get().size('hv_last_kernel_path') == 8:      # the 4x4 stride-2 layers take conv_g4_kernel with the tiled table: another summation order
            assert pa != 8 and maxerr(from_act(yt), ref.detach()) <= 4e-3 * max(1.0, ref.abs().max().item())
        else:
            assert torch.equal(yt.t, ya.t)
    if shift or Cout % 16:
        return
    gy = torch.randn(y0.shape, generator=g)
    y0.backward(gy)
    wb = ohwi_T(w)
    dxa = ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16)
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxa, k, s, p, 1, transposed=True, precision='fp16', w_h=wb.half())
    torch.cuda.synchronize()
    err = maxerr(from_act(dxa), xin.grad)
    assert err <= 4e-3 * max(1.0, xin.grad.abs().max().item()), err
    wbt = ops.tile_weights(wb.half(), Cin, k * k, Cout)
    dxt = ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16)
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxt, k, s, p, 1, transposed=True, precision='fp16', w_h=wb.half(), w_t=wbt)
    torch.cuda.synchronize()
    if lib.get().size('hv_last_kernel_path') == 8:
        assert maxerr(from_act(dxt), xin.grad) <= 4e-3 * max(1.0, xin.grad.abs().max().item())
    else:
        assert torch.equal(dxt.t, dxa.t)
    # a second writer of the same gradient buffer (accumulate = 1) with the producer's act' factor: y += v * act'(m)
    m = torch.randn(B, Cin, H, W, generator=g).half().float()        # the kernel sees the fp16-stored value (a tiny positive m may round to 0)
    fac = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxt, k, s, p, 1, transposed=True, precision='fp16', w_h=wb.half(), w_t=wbt, accumulate=1,
               mul=(to_act(m, dtype=torch.float16), 'lrelu'))
    torch.cuda.synchronize()
    want = xin.grad * (1.0 + fac)
    err = maxerr(from_act(dxt), want)
    assert err <= 6e-3 * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize('dil,H,W', [(2, 32, 32), (4, 32, 48), (8, 64, 64), (2, 20, 12), (16, 64, 64), (16, 128, 128), (8, 128, 128), (8, 64, 32)])
def test_conv_dilated_3x3_as_residue_subgrids(dil, H, W):
    """Dilated same-size 3x3 layers (the generators' d = 2, 4, 8 blocks): conv_halo2_kernel runs the d*d residue classes as undilated
    convolutions on sub-grids with pixel step d -- forward (bias + ELU) and data gradient (with the act' factor) against torch CPU fp32,
    plain and fragment-ordered filters bit-identical, and the kernel actually taken (path 3).  With the fragment-ordered table the filters-in-LDS
    kernels take the layer (path 7): sub-grids of at least a tile as residue classes, the generators' 8 x 8 (d = 8) and 4 x 4 (d = 16) sub-grids of a
    64 x 64 map packed whole into one tile (conv_lfd_kernel) -- same bits as conv_halo2_kernel; d = 16 without the table stays in the gather kernel."""
    from hvtest import to_act, from_act, ohwi, ohwi_T, dev, maxerr
    from hvgan import ops, lib
    B, Cin, Cout, k = 2, 64, 64, 3
    g = torch.Generator().manual_seed(11 + dil)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    y0 = F.conv2d(x, w, b, stride=1, padding=dil, dilation=dil)
    ref = F.elu(y0)
    wf = ohwi(w)
    ya, yt = (ops.Act.empty(B, H, W, Cout, dev(), dtype=torch.float16) for _ in range(2))
    ops.conv2d(to_act(x.detach(), dtype=torch.float16), wf, ya, k, 1, dil, dil, bias=b.to(dev()), act='elu', precision='fp16', w_h=wf.half())
    halo2 = dil <= 8
    assert (lib.get().size('hv_last_kernel_path') == 3) == halo2
    ops.conv2d(to_act(x.detach(), dtype=torch.float16), wf, yt, k, 1, dil, dil, bias=b.to(dev()), act='elu', precision='fp16', w_h=wf.half(),
               w_t=ops.tile_weights(wf.half(), Cout, k * k, Cin))
    packed = H == W and H // dil in (4, 8)          # whole residue classes in one tile
    lf = packed or min(H, W) // dil >= 16 or dil <= 4
    assert lib.get().size('hv_last_kernel_path') == (7 if lf else 3)
    if packed:
        assert b'conv_lfd_kernel' in lib.get().cdll.hv_last_kernel_name()
    torch.cuda.synchronize()
    assert maxerr(from_act(ya), ref.detach()) <= 4e-3 * max(1.0, ref.abs().max().item())
    assert maxerr(from_act(yt), ref.detach()) <= 4e-3 * max(1.0, ref.abs().max().item())
    if halo2:
        assert torch.equal(ya.t, yt.t)
    gy = torch.randn(y0.shape, generator=g)
    y0.backward(gy)
    m = torch.randn(B, Cin, H, W, generator=g)
    fac = torch.where(m > 0, torch.ones_like(m), m + 1.0)          # ELU' from the producer's output
    wb = ohwi_T(w)
    dxa = ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16)
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxa, k, 1, dil, dil, transposed=True, precision='fp16', w_h=wb.half(),
               mul=(to_act(m, dtype=torch.float16), 'elu'))
    assert (lib.get().size('hv_last_kernel_path') == 3) == halo2
    dxt = ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16)
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxt, k, 1, dil, dil, transposed=True, precision='fp16', w_h=wb.half(),
               w_t=ops.tile_weights(wb.half(), Cin, k * k, Cout), mul=(to_act(m, dtype=torch.float16), 'elu'))
    assert lib.get().size('hv_last_kernel_path') == (7 if lf else 3)
    torch.cuda.synchronize()
    want = x.grad * fac
    assert maxerr(from_act(dxa), want) <= 4e-3 * max(1.0, want.abs().max().item())
    assert maxerr(from_act(dxt), want) <= 4e-3 * max(1.0, want.abs().max().item())
    if halo2:
        assert torch.equal(dxa.t, dxt.t)
    # accumulate form on the packed tiles
    ops.conv2d(to_act(gy, dtype=torch.float16), wb, dxt, k, 1, dil, dil, transposed=True, precision='fp16', w_h=wb.half(),
               w_t=ops.tile_weights(wb.half(), Cin, k * k, Cout), mul=(to_act(m, dtype=torch.float16), 'elu'), accumulate=1)
    torch.cuda.synchronize()
    assert maxerr(from_act(dxt), 2 * want) <= 8e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 64, 64, 16, 16), (2, 64, 48, 16, 32), (3, 70, 50, 32, 64), (2, 33, 47, 32, 32), (16, 128, 128, 32, 64)])
def test_conv_3x3_stride2_filters_in_lds(B, H, W, Cin, Cout):
    """conv_lf2_kernel: the generators' 3x3 stride-2 down-sampling layers (forward) with the filter bank and a 33 x 33-pixel patch in LDS -- against torch CPU
    fp32 (bias + ELU), bit-identical to conv_halo_kernel (the route without the fragment-ordered table), odd map sizes incl., and the kernel actually taken."""
    from hvtest import to_act, from_act, ohwi, dev, maxerr
    from hvgan import ops, lib
    g = torch.Generator().manual_seed(5 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.elu(F.conv2d(x, w, b, stride=2, padding=1))
    Ho, Wo = ref.shape[2], ref.shape[3]
    wf = ohwi(w)
    ya, yt = (ops.Act.empty(B, Ho, Wo, Cout, dev(), dtype=torch.float16) for _ in range(2))
    xa = to_act(x, dtype=torch.float16)
    ops.conv2d(xa, wf, ya, 3, 2, 1, 1, bias=b.to(dev()), act='elu', precision='fp16', w_h=wf.half())
    assert lib.get().size('hv_last_kernel_path') == 2
    ops.conv2d(xa, wf, yt, 3, 2, 1, 1, bias=b.to(dev()), act='elu', precision='fp16', w_h=wf.half(), w_t=ops.tile_weights(wf.half(), Cout, 9, Cin))
    assert lib.get().size('hv_last_kernel_path') == 7 and b'conv_lf2_kernel' in lib.get().cdll.hv_last_kernel_name()
    torch.cuda.synchronize()
    assert maxerr(from_act(yt), ref) <= 4e-3 * max(1.0, ref.abs().max().item())
    assert torch.equal(ya.t, yt.t)


@pytest.mark.parametrize('rows,taps,K', [(512, 16, 256), (20, 9, 48), (4, 25, 16), (64, 9, 36), (1, 16, 512)])
def test_weight_table_in_mfma_fragment_order(rows, taps, K):
    """hv_weight_tile_f16 / hv_weight_tiled_elems against the index formula documented in include/hvgan.h (rows padded to 16 with zeros;
    no tiled form unless K % 16 == 0), and hv_weight_prep's own tiled output against the tiling of its plain fp16 output."""
    import numpy as np
    from hvtest import dev
    from hvgan import ops
    T = 32 if K % 32 == 0 else 16 if K % 16 == 0 else 0
    n = ops.tiled_elems(rows, taps, K)
    assert n == (0 if not T else (rows + 15) // 16 * 16 * taps * K)
    w = torch.randn(rows, taps, K, generator=torch.Generator().manual_seed(3)).half()
    wt = ops.tile_weights(w.to(dev()), rows, taps, K)
    if not T:
        assert wt is None
        return
    torch.cuda.synchronize()
    r, t, k = np.meshgrid(np.arange(rows), np.arange(taps), np.arange(K), indexing='ij')
    q = T // 4
    idx = (r // 16) * 16 * taps * K + ((t * K + k) // T) * 16 * T + (((k % T) // q) * 16 + r % 16) * q + k % q
    ref = np.zeros(n, dtype=np.float16)
    ref[idx.ravel()] = w.numpy().ravel()
    assert np.array_equal(wt.cpu().numpy(), ref)


def test_weight_prep_writes_the_fragment_ordered_tables():
    """hv_weight_prep's w_fwd_t / w_bwd_t are the tilings of its w_fwd_h / w_bwd_h (spectral-norm layer and plain layer)."""
    from hvtest import dev
    from hvgan import ops, engine
    import torch.nn as nn
    torch.manual_seed(5)
    convs = []
    for i, (cin, cout, k) in enumerate([(32, 48, 3), (16, 64, 4), (4, 16, 5)]):
        m = nn.Conv2d(cin, cout, k).to(dev())
        u = torch.randn(cout, device=dev()) if i == 0 else None
        v = torch.randn(cin * k * k, device=dev()) if i == 0 else None
        convs.append(engine.ConvParams('c%d' % i, m.weight, m.bias, cin, cout, k, u=u, v=v))
    ps = engine.ParamSet(convs)
    ps.prep(dev(), power_iter=True)
    torch.cuda.synchronize()
    for c in convs:
        for plain, tiled, rows, K in ((c.w_fwd_h, c.w_fwd_t, c.cout, c.cin_fwd), (c.w_bwd_h, c.w_bwd_t, c.cin_fwd, c.coutP)):
            want = ops.tile_weights(plain.contiguous(), rows, c.taps, K)
            assert (want is None) == (tiled is None), (c.name, rows, K)
            if want is not None:
                torch.cuda.synchronize()
                assert torch.equal(want, tiled), c.name


@pytest.mark.parametrize('sn', [False, True])
def test_weight_prep_fused_layout_kernel_writes_the_same_six_tables(sn):
    """hv_weight_prep2 (one read of the weights, 32 x 32 tiles through LDS, whole fragments per store) against hv_weight_prep's element-wise kernels:
    all six tables of every layer bit for bit -- PatchGAN shapes (1 -> 64 stem, 256 -> 512, 512 -> 1 logits), generator shapes (ragged 33-channel
    concat input, 5 x 5 stem that stays on the element-wise path), with and without spectral norm."""
    from hvtest import dev
    from hvgan import ops, engine, lib
    import ctypes
    import torch.nn as nn
    torch.manual_seed(7)
    shapes = [(1, 64, 4), (64, 128, 4), (256, 512, 4), (512, 1, 4), (33, 32, 3), (64, 64, 3), (8, 1, 3), (4, 16, 5), (24, 40, 3)]
    convs = []
    for i, (cin, cout, k) in enumerate(shapes):
        m = nn.Conv2d(cin, cout, k).to(dev())
        u = torch.randn(cout, device=dev()) if sn else None
        v = torch.randn(cin * k * k, device=dev()) if sn else None
        convs.append(engine.ConvParams('c%d' % i, m.weight, m.bias, cin, cout, k, u=u, v=v))
    ps = engine.ParamSet(convs)
    ps.prep(dev(), power_iter=False)
    torch.cuda.synchronize()
    names = ('w_fwd', 'w_bwd', 'w_fwd_h', 'w_bwd_h', 'w_fwd_t', 'w_bwd_t')
    got = [{n: (getattr(c, n).clone() if getattr(c, n) is not None else None) for n in names} for c in convs]
    for c in convs:      # scribble over the tables (not the padding of the ordered ones: the element-wise kernels never write it), then the old entry point
        for n in names[:4]:
            getattr(c, n).fill_(7.0)
    L = lib.get()
    table = ps.t_prep[False]
    L.call('hv_weight_prep', ctypes.cast(table.ptr(), ctypes.POINTER(L.hv_wprep_layer)), table.n,
           ctypes.c_longlong(max(c.sizes()[0] + c.sizes()[1] for c in convs)), ops.stream())
    torch.cuda.synchronize()
    for c, g in zip(convs, got):
        for n in names:
            t = getattr(c, n)
            assert (t is None) == (g[n] is None)
            if t is not None:
                assert torch.equal(t, g[n]), (c.name, n, (t.float() - g[n].float()).abs().max().item())


HEAD_CASES = [   # B, H, W, Cin, k, stride, pad, transposed, act
    (16, 31, 31, 512, 4, 1, 1, 0, 'none'),     # PatchGAN logits
    (3, 17, 23, 96, 3, 1, 1, 0, 'sigmoid'),    # 9 taps, one 32-channel step per item, ragged pixel count
    (2, 20, 20, 192, 4, 2, 1, 0, 'lrelu'),     # strided forward, two steps per item
    (4, 64, 64, 64, 4, 2, 1, 1, 'none'),       # data gradient of the 1-channel stem (conv_transpose form)
    (2, 15, 15, 128, 3, 1, 1, 1, 'none'),
]


@pytest.mark.parametrize('case', HEAD_CASES)
def test_conv_single_output_channel_tap_gemm(case):
    """Cout == 1 with many input channels (conv_head.hip: [pixel][tap] table + tap sum) against torch CPU fp32, incl. the epilogue
    options (bias, activation, accumulate, act' multiplier) and the kernel actually taken."""
    from hvtest import to_act, from_act, ohwi, ohwi_T, dev, maxerr, st
    from hvgan import ops, lib
    B, H, W, Cin, k, s, p, tr, act = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=g)
    if not tr:
        w = torch.randn(1, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(1, generator=g) * 0.1
        ref = _ref_act(F.conv2d(x, w, b, stride=s, padding=p), act)
        wf = ohwi(w)
        ya = ops.Act.empty(B, ref.shape[2], ref.shape[3], 1, dev())
        xh = to_act(x, dtype=torch.float16)                 # the layer's input is a network-internal (fp16) tensor, its 1-channel output an fp32 image
        ops.conv2d(xh, wf, ya, k, s, p, 1, bias=b.to(dev()), act=act, precision='fp16', w_h=wf.half())
        assert lib.get().size('hv_last_kernel_path') == 5
        torch.cuda.synchronize()
        assert maxerr(from_act(ya), ref) <= 4e-3 * max(1.0, ref.abs().max().item())
        # y += r (accumulate 1) on top of the first result, and y = act(r + y) (accumulate 2)
        ops.conv2d(xh, wf, ya, k, s, p, 1, bias=b.to(dev()), act=act, accumulate=1, precision='fp16', w_h=wf.half())
        torch.cuda.synchronize()
        assert maxerr(from_act(ya), 2 * ref) <= 8e-3 * max(1.0, ref.abs().max().item())
        y0 = torch.randn(ref.shape, generator=g)
        yb = to_act(y0)
        ops.conv2d(xh, wf, yb, k, s, p, 1, act=act, accumulate=2, precision='fp16', w_h=wf.half())
        torch.cuda.synchronize()
        ref2 = _ref_act(F.conv2d(x, w, None, stride=s, padding=p) + y0, act)
        assert maxerr(from_act(yb), ref2) <= 4e-3 * max(1.0, ref2.abs().max().item())
        return
    # x is the gradient of a conv 1 -> Cin (stride s): the data gradient is conv_transpose2d(x, w) with w [Cin][1][k][k]
    w = torch.randn(Cin, 1, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = F.conv_transpose2d(x, w, stride=s, padding=p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    wb = ohwi_T(w)                          # [1][taps][Cin]
    assert tuple(wb.shape) == (1, k * k, Cin)
    m = torch.randn(B, 1, Ho, Wo, generator=g)        # output of a LeakyReLU producer: factor 1 or 0.2
    ya = ops.Act.empty(B, Ho, Wo, 1, dev())
    ops.conv2d(to_act(x, dtype=torch.float16), wb, ya, k, s, p, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(to_act(m), 'lrelu'))
    assert lib.get().size('hv_last_kernel_path') == 5
    torch.cuda.synchronize()
    ref = ref * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    assert maxerr(from_act(ya), ref) <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-4), ('fp16', 4e-3)])
@pytest.mark.parametrize('case', [(16, 31, 31, 512, 4, 1), (2, 9, 14, 96, 3, 1), (3, 20, 20, 320, 4, 2)])
def test_conv_single_output_channel_data_gradient(case, prec, tol):
    """Data gradient of a Cout == 1 conv: gradient stored channel-padded to 4, act' multiplier of the producer layer and
    accumulate == 1 included."""
    from hvtest import to_act, from_act, ohwi_T, dev, maxerr, st
    from hvgan import ops, lib
    B, H, W, C, k, p = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(1, C, k, k, generator=g) / (C * k * k) ** 0.5
    y = F.conv2d(x, w, None, stride=1, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    wb = ohwi_T(w, CoutP=4)                                    # [C][taps][4]
    ga = to_act(gy, 4, dtype=st(prec))
    ga = ops.Act(ga.t, 4, 0)
    m = torch.randn(B, C, H, W, generator=g)
    fac = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    dxa = ops.Act.empty(B, H, W, C, dev(), dtype=st(prec))
    ops.conv2d(ga, wb, dxa, k, 1, p, 1, transposed=True, precision=prec, w_h=wb.half(), mul=(to_act(m, dtype=st(prec)), 'lrelu'))
    torch.cuda.synchronize()
    ref = x.grad * fac
    assert maxerr(from_act(dxa), ref) <= tol * max(1.0, ref.abs().max().item())
    ops.conv2d(ga, wb, dxa, k, 1, p, 1, transposed=True, precision=prec, w_h=wb.half(), accumulate=1)
    torch.cuda.synchronize()
    assert maxerr(from_act(dxa), ref + x.grad) <= 2 * tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('case', [(2, 32, 32, 32, 32, 'elu'), (2, 16, 48, 64, 64, 'elu'), (3, 24, 16, 16, 32, 'none'), (2, 32, 32, 64, 32, 'relu'), (2, 16, 16, 32, 64, 'elu')])
def test_conv_data_gradient_of_upsampled_input_leaves_pooled(case):
    """hv_conv_desc.pool2: the data gradient of a 3x3 conv whose input was the nearest x2 up-sampling of a smaller tensor, written 2x2 sum-pooled
    by the conv's own epilogue, times act' of the small tensor, assigned and accumulated -- against torch autograd through
    F.interpolate(nearest) + conv2d on the CPU."""
    from hvtest import to_act, from_act, ohwi_T, dev, maxerr
    from hvgan import ops, lib
    B, h, w_, Cin, Cout, act = case          # the small tensor is B x Cin x h x w_; the conv runs at 2h x 2w_ and has Cout output channels
    g = torch.Generator().manual_seed(3 + Cin + Cout)
    pre = torch.randn(B, Cin, h, w_, generator=g)
    pre.requires_grad_(True)
    fn = {'elu': F.elu, 'relu': F.relu, 'none': lambda t: t}[act]
    low = fn(pre)
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().float()
    y = F.conv2d(F.interpolate(low, scale_factor=2, mode='nearest'), wt, None, stride=1, padding=1)
    gy = torch.randn(y.shape, generator=g).half().float()
    y.backward(gy)
    want = pre.grad                                               # d/d(pre-activation of the small tensor)
    wb = ohwi_T(wt)                                               # [Cin][9][Cout]: the data-gradient table
    wh = wb.half()
    wtl = ops.tile_weights(wh, Cin, 9, Cout)
    ga = to_act(gy, dtype=torch.float16)
    la = to_act(low.detach(), dtype=torch.float16)
    dxa = ops.Act.empty(B, h, w_, Cin, dev(), dtype=torch.float16)
    assert ops.pool2_ok(ga, dxa, 3, 1, 1, 1, 'fp16', wh, wtl)
    ops.conv2d(ga, wb, dxa, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wh, w_t=wtl, pool2=True, mul=(la, act) if act != 'none' else None)
    assert lib.get().size('hv_last_kernel_path') == 7
    torch.cuda.synchronize()
    scale = max(1.0, want.abs().max().item())
    assert maxerr(from_act(dxa), want) <= 4e-3 * scale
    ops.conv2d(ga, wb, dxa, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wh, w_t=wtl, pool2=True, accumulate=1, mul=(la, act) if act != 'none' else None)
    torch.cuda.synchronize()
    assert maxerr(from_act(dxa), 2 * want) <= 8e-3 * scale
    # a shape the pooled kernel does not serve is refused, not silently computed at full size
    with pytest.raises(RuntimeError):
        ops.conv2d(ga, wb, dxa, 3, 1, 1, 1, transposed=True, precision='fp32', pool2=True)


@pytest.mark.parametrize('case', [(2, 16, 16, 32, 'elu'), (2, 24, 16, 64, 'elu'), (3, 8, 24, 32, 'none')])
def test_conv_over_upsampled_map_plus_one_extra_channel_without_the_concat(case):
    """hv_conv_desc.x1: a 3x3 conv over the concatenation [nearest x2 up-sampling of a K-channel map | one extra channel] that is never built -- the main
    input is read with the fused up-sampling (in_shift), the extra channel's nine taps are added in the epilogue -- against torch on the materialised
    concat; the filters of the first K channels come from hv_weight_prep2's split table (w_fwd_t2)."""
    from hvtest import to_act, from_act, dev, maxerr
    from hvgan import ops, engine, lib
    import torch.nn as nn
    B, h, w_, K, act = case
    g = torch.Generator().manual_seed(21 + K)
    low = torch.randn(B, K, h, w_, generator=g).half().float()
    x1 = torch.randn(B, 1, 2 * h, 2 * w_, generator=g).half().float()
    m = nn.Conv2d(K + 1, K, 3, padding=1)
    with torch.no_grad():
        m.weight.copy_((torch.randn(m.weight.shape, generator=g) / ((K + 1) * 9) ** 0.5))
        m.bias.copy_(torch.randn(K, generator=g) * 0.1)
    fn = {'elu': F.elu, 'none': lambda t: t}[act]
    ref = fn(F.conv2d(torch.cat([F.interpolate(low, scale_factor=2, mode='nearest'), x1], 1), m.weight.half().float(), m.bias, padding=1))
    m = m.to(dev())
    cp = engine.ConvParams('c', m.weight, m.bias, K + 1, K, 3, cin_fwd=ops.cpad(K + 1))
    cp.split_k = K
    ps = engine.ParamSet([cp])
    ps.prep(dev(), power_iter=False)
    cat = ops.Act.empty(B, 2 * h, 2 * w_, K + 1, dev(), ld=ops.cpad(K + 1), dtype=torch.float16, zero=True)
    ops.copy_channels(to_act(x1, dtype=torch.float16), cat.slice(K, 1), mode=0)
    node = engine.ConvNode(cp, cat, ops.Act.empty(B, 2 * h, 2 * w_, K, dev(), dtype=torch.float16), 1, 1, 1, act)
    node.split = (to_act(low, dtype=torch.float16), cat.slice(K, 1))
    assert node.split_forward('fp16')
    node.forward('fp16')
    assert lib.get().size('hv_last_kernel_path') == 7
    torch.cuda.synchronize()
    assert maxerr(from_act(node.y), ref) <= 4e-3 * max(1.0, ref.abs().max().item())
    # the same node through the materialised concat (what the fp32 mode and unsupported shapes do)
    ops.copy_channels(to_act(low, dtype=torch.float16), cat.slice(0, K), mode=1)
    node.split = None
    node.forward('fp16')
    torch.cuda.synchronize()
    assert maxerr(from_act(node.y), ref) <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('case', [(2, 7, 7, 32, 1), (2, 18, 21, 64, 3), (3, 31, 31, 512, 1), (2, 33, 17, 256, 4)])
def test_conv_logits_data_gradient_taps_as_mfma_contraction(case):
    """logits_dgrad_kernel: data gradient of a 4x4 / stride 1 / pad 1 conv with <= 4 output channels (gradient stored with a channel stride
    of 4) back to 32..512 channels, the 16 taps as the contraction of one 16x16x16 MFMA; ragged 16-pixel groups, 1..4 live gradient channels,
    act' multiplier and both accumulate modes, against torch CPU fp32."""
    from hvtest import to_act, from_act, dev, maxerr
    from hvgan import ops, lib
    B, H, W, C, live = case
    g = torch.Generator().manual_seed(11 + C)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(live, C, 4, 4, generator=g) / (C * 16) ** 0.5
    y = F.conv2d(x, w.half().float(), None, stride=1, padding=1)
    gy = torch.randn(y.shape, generator=g).half().float()
    y.backward(gy)
    wb = torch.zeros(C, 16, 4)
    wb[:, :, :live] = w.reshape(live, C, 16).permute(1, 2, 0)          # [C][taps][4]
    wb = wb.to(dev())
    gp = torch.zeros(B, 4, H - 1, W - 1)
    gp[:, :live] = gy
    ga = to_act(gp, dtype=torch.float16)
    m = torch.randn(B, C, H, W, generator=g)
    fac = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    dxa = ops.Act.empty(B, H, W, C, dev(), dtype=torch.float16)
    ops.conv2d(ga, wb, dxa, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(to_act(m, dtype=torch.float16), 'lrelu'))
    assert lib.get().size('hv_last_kernel_path') == 9
    torch.cuda.synchronize()
    ref = x.grad * fac
    scale = max(1.0, ref.abs().max().item())
    assert maxerr(from_act(dxa), ref) <= 3e-3 * scale
    ops.conv2d(ga, wb, dxa, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), accumulate=1)
    torch.cuda.synchronize()
    assert maxerr(from_act(dxa), ref + x.grad) <= 6e-3 * scale


@pytest.mark.parametrize('case', [(16, 256, 256, 64, 2, 'lrelu'), (2, 37, 41, 32, 2, 'none'), (3, 20, 22, 16, 1, 'elu'), (2, 70, 50, 64, 2, 'lrelu')])
def test_conv_one_channel_stem_mfma(case):
    """1-channel image into 16..64 channels, 4x4 filter (PatchGAN stem), fp16 mode with the fp16 filter copy: the 16 taps are the contraction
    of one 16x16x16 MFMA (stem1_mfma_kernel); against torch CPU fp32, incl. ragged row ends, bias, activation and accumulate."""
    from hvtest import to_act, from_act, ohwi, dev, maxerr, st
    from hvgan import ops, lib
    B, H, W, Cout, s, act = case
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    w = torch.randn(Cout, 1, 4, 4, generator=g) / 4
    b = torch.randn(Cout, generator=g) * 0.1
    ref = _ref_act(F.conv2d(x, w, b, stride=s, padding=1), act)
    wf = ohwi(w)
    ya = ops.Act.empty(B, ref.shape[2], ref.shape[3], Cout, dev(), dtype=torch.float16)
    ops.conv2d(to_act(x), wf, ya, 4, s, 1, 1, bias=b.to(dev()), act=act, precision='fp16', w_h=wf.half())
    # 64 channels, fp16 output: the form whose tiles leave through LDS as 16-byte pieces (ragged row ends in the 70 x 50 case)
    assert (lib.get().cdll.hv_last_kernel_name() or b'').decode() == ('stem1_mfma_kernel<true>' if Cout == 64 else 'stem1_mfma_kernel')
    torch.cuda.synchronize()
    assert maxerr(from_act(ya), ref) <= 4e-3 * max(1.0, ref.abs().max().item())
    ops.conv2d(to_act(x), wf, ya, 4, s, 1, 1, bias=b.to(dev()), act=act, accumulate=1, precision='fp16', w_h=wf.half())
    assert (lib.get().cdll.hv_last_kernel_name() or b'').decode() == 'stem1_mfma_kernel'
    torch.cuda.synchronize()
    assert maxerr(from_act(ya), 2 * ref) <= 8e-3 * max(1.0, ref.abs().max().item())


def test_conv_upsample_fused_and_transposed_conv_layer():
    """in_shift=1 == conv(F.interpolate(x, 2)); transposed=1 == F.conv_transpose2d (k4 s2 p1)."""
    from hvtest import to_act, from_act, ohwi, dev, maxerr, st
    from hvgan import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 16, 16, generator=g)
    w = torch.randn(16, 32, 3, 3, generator=g) / 17.
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), w, None, padding=1)
    ya = ops.Act.empty(2, 32, 32, 16, dev())
    ops.conv2d(to_act(x), ohwi(w), ya, 3, 1, 1, 1, in_shift=1, precision='fp32')
    assert maxerr(from_act(ya), ref) <= 1e-4
    wt = torch.randn(32, 8, 4, 4, generator=g) / 10.      # ConvTranspose2d weight [Cin][Cout][kh][kw]
    ref = F.conv_transpose2d(x, wt, None, stride=2, padding=1)
    wl = torch.zeros(8, 16, 32)
    wl[:] = wt.permute(1, 2, 3, 0).reshape(8, 16, 32)
    ya = ops.Act.empty(2, 32, 32, 8, dev())
    ops.conv2d(to_act(x), wl.to(dev()), ya, 4, 2, 1, 1, transposed=True, precision='fp32')
    assert maxerr(from_act(ya), ref) <= 1e-4


def test_conv_per_sample_filters_and_accumulate_modes():
    from hvtest import to_act, from_act, ohwi, dev, maxerr, st
    from hvgan import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 16, 16, generator=g)
    ws = [torch.randn(64, 16, 3, 3, generator=g) / 12. for _ in range(2)]
    sc = torch.rand(2, 64, generator=g) + 0.5
    ref = torch.cat([F.conv2d(x[i:i + 1], ws[i], None, padding=1) * sc[i].view(1, -1, 1, 1) for i in range(2)])
    wb = torch.stack([ohwi(w) for w in ws]).contiguous()
    ya = ops.Act.empty(2, 16, 16, 64, dev())
    ops.conv2d(to_act(x), wb, ya, 3, 1, 1, 1, w_bstride=wb[0].numel(), ch_scale=sc.to(dev()), ch_scale_bstride=64, precision='fp32')
    assert maxerr(from_act(ya), ref) <= 1e-4
    # accumulate=1 adds after the activation; accumulate=2 before it
    base = torch.randn(2, 64, 16, 16, generator=g)
    ya = to_act(base)
    ops.conv2d(to_act(x), ohwi(ws[0]), ya, 3, 1, 1, 1, accumulate=1, precision='fp32')
    assert maxerr(from_act(ya), base + F.conv2d(x, ws[0], None, padding=1)) <= 1e-4
    ya = to_act(base)
    ops.conv2d(to_act(x), ohwi(ws[0]), ya, 3, 1, 1, 1, accumulate=2, act='elu', precision='fp32')
    assert maxerr(from_act(ya), F.elu(base + F.conv2d(x, ws[0], None, padding=1))) <= 1e-4


@pytest.mark.parametrize('prec,tol', [('fp32', 2e-4), ('fp16', 6e-3)])
def test_conv_per_sample_paste_long_contraction(prec, tol):
    """The attention paste shape class: conv_transpose2d(k4, s2, p1) with per-sample filters and a 4 x 1024-long contraction per parity class
    against torch."""
    from hvtest import to_act, from_act, dev, maxerr, st
    from hvgan import ops
    g = torch.Generator().manual_seed(8)
    B, L, C = 2, 1024, 64
    a = torch.rand(B, L, 32, 32, generator=g) / 8                       # attention scores [B, L, h, w]
    raw = torch.randn(B, L, C, 4, 4, generator=g) / 4                   # raw 4x4 patches per sample (conv_transpose weights [L, C, 4, 4])
    ref = torch.cat([F.conv_transpose2d(a[i:i + 1], raw[i], stride=2, padding=1) / 4. for i in range(B)])
    # gather form: y[co] = sum_{tap, l} x[l] * w[co][tap][l]  ->  per-sample filters [C][16][L]
    wt = raw.permute(0, 2, 3, 4, 1).reshape(B, C, 16, L).contiguous().to(dev())
    ya = ops.Act.empty(B, 64, 64, C, dev())
    ops.conv2d(to_act(a), wt, ya, 4, 2, 1, 1, transposed=True, alpha=0.25, w_bstride=C * 16 * L, precision=prec)
    torch.cuda.synchronize()
    assert maxerr(from_act(ya), ref) <= tol * max(1.0, ref.abs().max().item())


def test_conv_rejects_bad_arguments():
    from hvtest import to_act, ohwi, dev
    from hvgan import ops
    x = torch.randn(1, 8, 8, 8)
    w = torch.randn(8, 8, 3, 3)
    ya = ops.Act.empty(1, 7, 8, 8, dev())            # wrong output height
    with pytest.raises(RuntimeError):
        ops.conv2d(to_act(x), ohwi(w), ya, 3, 1, 1, 1)


WTR_CASES = [   # B, H, W, Cin, Cout, k, stride, pad
    (3, 31, 31, 256, 512, 4, 1, 1),       # PatchGAN 256 -> 512 (ragged 31 x 31 map: partial tiles)
    (2, 64, 64, 64, 128, 4, 2, 1),        # PatchGAN 64 -> 128, stride 2
    (2, 32, 32, 128, 256, 4, 2, 1),       # PatchGAN 128 -> 256, stride 2
    (2, 40, 24, 32, 32, 3, 1, 1),         # generator 3x3 layers
    (2, 33, 47, 64, 64, 3, 1, 1),
    (2, 32, 32, 16, 32, 3, 1, 1),
    (2, 32, 32, 64, 32, 3, 1, 1),
    (4, 16, 16, 24, 40, 3, 1, 1),         # channel counts that are multiples of 8 but not of the tile widths
    (2, 32, 32, 72, 64, 3, 1, 1),         # the 65 (+7 pad) channel concat input of conv20
    (2, 32, 32, 32, 64, 3, 2, 1),         # 3x3 stride-2 downsampling layers
    (2, 64, 48, 16, 32, 3, 2, 1),
    (2, 64, 48, 16, 16, 3, 2, 1),         # 16 filters at stride 2: a half-empty 32-filter block
    (12, 128, 96, 64, 128, 4, 2, 1),      # enough pixel tiles per workgroup for the LDS-DMA form (wgrad_trd_kernel), stride 2 (32-channel blocks on the parity de-interleaved image) ...
    (12, 40, 56, 128, 256, 4, 2, 1),
    (4, 33, 47, 256, 296, 4, 1, 1),       # ... and stride 1 with ragged tiles and a partial 64-channel output block
]
WTR_DMA = {0, 12, 13, 14}                 # indices of the cases the plan gives to the LDS-DMA form


@pytest.mark.parametrize('case', WTR_CASES)
def test_wgrad_transposed_lds_read_kernel(case):
    """wgrad_tr_kernel (fp16 storage, ds_read_b64_tr_b16 operands, input patch staged once for all taps) against torch CPU fp32:
    weight gradient, the bias gradient folded into the same launch, accumulate, and the kernel actually taken."""
    from hvtest import to_act, dev, maxerr
    from hvgan import ops, lib
    B, H, W, Cin, Cout, k, s, p = case
    g_ = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g_)
    w = (torch.randn(Cout, Cin, k, k, generator=g_) / (Cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    g = torch.randn(y.shape, generator=g_)
    y.backward(g)
    xa, ga = to_act(x, dtype=torch.float16), to_act(g, dtype=torch.float16)
    dw = torch.empty(Cout, k * k, Cin, device=dev())
    db = torch.full((Cout,), 3.0, device=dev())
    ops.conv2d_wgrad(xa, ga, dw, k, s, p, 1, precision='fp16', dbias=db)
    assert lib.get().size('hv_last_kernel_path') == (13 if WTR_CASES.index(case) in WTR_DMA else 12)
    torch.cuda.synchronize()
    got = dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    scale = max(1.0, w.grad.abs().max().item())
    assert maxerr(got, w.grad) <= 6e-3 * scale, maxerr(got, w.grad)
    gsum = g.half().float().sum(dim=(0, 2, 3))          # the kernel sums the fp16-stored values exactly (fp32 accumulation)
    assert maxerr(db.cpu(), gsum) <= 1e-3 * max(1.0, g.abs().sum(dim=(0, 2, 3)).max().item())
    ops.conv2d_wgrad(xa, ga, dw, k, s, p, 1, precision='fp16', accumulate=True, dbias=db, dbias_accumulate=True)
    torch.cuda.synchronize()
    assert maxerr(dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2), 2 * w.grad) <= 1.2e-2 * scale
    assert maxerr(db.cpu(), 2 * gsum) <= 2e-3 * max(1.0, g.abs().sum(dim=(0, 2, 3)).max().item())


@pytest.mark.parametrize('dil,B,H,W,Cin,Cout', [(2, 2, 32, 32, 64, 64), (4, 2, 32, 48, 64, 64), (8, 3, 64, 64, 64, 64), (2, 2, 20, 12, 32, 64), (16, 2, 64, 64, 64, 64)])
def test_wgrad_dilated_3x3_through_the_transposed_read_kernel(dil, B, H, W, Cin, Cout):
    """Weight gradients of the generators' dilated same-size 3x3 layers (d = 2, 4, 8): wgrad_tr_kernel walks the d*d residue sub-grids as undilated layers
    with pixel step d and sums them in the same accumulators (round 4; they ran in the gather kernel).  Against torch CPU fp32, with the bias gradient and
    the accumulate form; d = 16 (4 x 4-pixel sub-grids) stays in the gather kernel."""
    from hvtest import to_act, dev, maxerr
    from hvgan import ops, lib
    g_ = torch.Generator().manual_seed(40 + dil)
    x = torch.randn(B, Cin, H, W, generator=g_)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g_) / (Cin * 9) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=dil, dilation=dil)
    g = torch.randn(y.shape, generator=g_)
    y.backward(g)
    xa, ga = to_act(x, dtype=torch.float16), to_act(g, dtype=torch.float16)
    dw = torch.empty(Cout, 9, Cin, device=dev())
    db = torch.full((Cout,), 3.0, device=dev())
    ops.conv2d_wgrad(xa, ga, dw, 3, 1, dil, dil, precision='fp16', dbias=db)
    assert lib.get().size('hv_last_kernel_path') == (12 if dil <= 8 else 10)
    torch.cuda.synchronize()
    scale = max(1.0, w.grad.abs().max().item())
    assert maxerr(dw.cpu().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), w.grad) <= 6e-3 * scale
    gsum = g.half().float().sum(dim=(0, 2, 3))
    assert maxerr(db.cpu(), gsum) <= 1e-3 * max(1.0, g.abs().sum(dim=(0, 2, 3)).max().item())
    ops.conv2d_wgrad(xa, ga, dw, 3, 1, dil, dil, precision='fp16', accumulate=True, dbias=db, dbias_accumulate=True)
    torch.cuda.synchronize()
    assert maxerr(dw.cpu().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), 2 * w.grad) <= 1.2e-2 * scale


S2T_CASES = [   # B, H (conv input = gradient output size), W, Cin, Cout (of the forward conv), k
    (2, 64, 64, 64, 128, 4),       # PatchGAN 64 -> 128 (data gradient 128 -> 64 at 64 x 64)
    (2, 32, 32, 128, 256, 4),
    (16, 32, 32, 16, 32, 3),       # generator 3x3 stride-2 layers: parity classes of 1, 2, 2 and 4 taps
    (2, 48, 32, 32, 64, 3),
    (2, 16, 48, 16, 16, 3),
    (3, 24, 40, 40, 72, 4),        # channel counts off the tile widths, ragged tiles
]


@pytest.mark.parametrize('case', S2T_CASES)
def test_conv_stride2_data_gradient_fused_parity_classes(case):
    """conv_s2t_kernel (the four output parities of a stride-2 data gradient in one workgroup, one wave each) against torch CPU fp32,
    with the producer's act' multiplier and accumulate, and the kernel actually taken."""
    from hvtest import to_act, from_act, ohwi_T, dev, maxerr
    from hvgan import ops, lib
    B, H, W, Cin, Cout, k = case
    g_ = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=g_, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g_) / (Cin * k * k) ** 0.5
    y = F.conv2d(x, w, None, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g_)
    y.backward(gy)
    m = torch.randn(B, Cin, H, W, generator=g_)
    fac = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    wb = ohwi_T(w)
    dxa = ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16)
    ga = to_act(gy, dtype=torch.float16)
    prev = lib.get().size('hv_set_s2t_mode', 2)   # the 4x4 filters take it only in mode 2
    try:
        ops.conv2d(ga, wb, dxa, k, 2, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(to_act(m, dtype=torch.float16), 'lrelu'))
        assert lib.get().size('hv_last_kernel_path') == 6
        torch.cuda.synchronize()
        ref = x.grad * fac
        scale = max(1.0, ref.abs().max().item())
        assert maxerr(from_act(dxa), ref) <= 4e-3 * scale, maxerr(from_act(dxa), ref)
        ops.conv2d(ga, wb, dxa, k, 2, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), accumulate=1)
        torch.cuda.synchronize()
        assert maxerr(from_act(dxa), ref + x.grad) <= 8e-3 * scale
        wbt = ops.tile_weights(wb.half(), Cin, k * k, Cout)       # fragment-ordered filters: the same bits
        if wbt is not None:
            d0, d1 = (ops.Act.empty(B, H, W, Cin, dev(), dtype=torch.float16) for _ in range(2))
            ops.conv2d(ga, wb, d0, k, 2, 1, 1, transposed=True, precision='fp16', w_h=wb.half())
            ops.conv2d(ga, wb, d1, k, 2, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), w_t=wbt)
            path = lib.get().size('hv_last_kernel_path')
            torch.cuda.synchronize()
            if k == 4 and Cout % 32 == 0 and Cin % 64 == 0:      # with the tiled table the 4x4 layers take conv_g4_kernel (another summation order)
                assert path == 8 and maxerr(from_act(d1), x.grad) <= 4e-3 * max(1.0, x.grad.abs().max().item())
            else:
                assert path == 6 and torch.equal(d0.t, d1.t)
    finally:
        lib.get().size('hv_set_s2t_mode', prev)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(16, 31, 31, 512, 'batch', 2), (4, 31, 31, 512, 'batch', 1), (3, 7, 9, 128, 'batch', 1), (4, 10, 10, 256, 'instance', 4)])
def test_conv_logits_layer_normalises_its_input_at_staging(case):
    """hv_conv_desc.xn_*: the PatchGAN logits layer reads the RAW output of the layer below, applies BatchNorm / InstanceNorm + LeakyReLU where it stages
    the operand and stores the normalised map on the way -- logits and map bit-identical to the separate normalisation pass followed by the plain
    convolution (reference models/networks.py:583-598: norm_layer, LeakyReLU(0.2), Conv2d(.., 1, 4, 1, 1))."""
    from hvtest import to_act, from_act, ohwi, dev
    from hvgan import ops, lib
    B, H, W, C, norm, groups = case
    g = torch.Generator().manual_seed(5)
    z = to_act(torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3, dtype=torch.float16)
    w = torch.randn(1, C, 4, 4, generator=g) / (C * 16) ** 0.5
    b = (torch.randn(1, generator=g) * 0.1).to(dev())
    wf = ohwi(w)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev()) if norm == 'batch' else None
    bet = (torch.randn(C, generator=g) * 0.2).to(dev()) if norm == 'batch' else None
    rm, rv, nbt = torch.zeros(C, device=dev()), torch.ones(C, device=dev()), torch.zeros(1, dtype=torch.long, device=dev())
    Ho, Wo = H - 1, W - 1

    def run(fused):
        stats = torch.zeros(2 * B * C, device=dev())
        y = ops.Act(torch.zeros(B, H, W, C, dtype=torch.float16, device=dev()))
        out = ops.Act.empty(B, Ho, Wo, 1, dev())
        kw = dict(gamma=gam, beta=bet, running_mean=rm.clone(), running_var=rv.clone(), nbt=nbt.clone(), groups=groups) if norm == 'batch' else {}
        ops.norm_act_forward(z, None if fused else y, norm, True, stats, act='lrelu', **kw)
        xn = (stats, gam, bet, groups, 'lrelu', y) if fused else None
        if fused:
            assert ops.conv2d_supported(z, wf, out, 4, 1, 1, 1, bias=b, precision='fp16', w_h=wf.half(), xn=xn)
        ops.conv2d(z if fused else y, wf, out, 4, 1, 1, 1, bias=b, precision='fp16', w_h=wf.half(), xn=xn)
        assert lib.get().size('hv_last_kernel_path') == 5
        torch.cuda.synchronize()
        return from_act(out), y.t.clone()
    o0, y0 = run(False)
    o1, y1 = run(True)
    assert torch.equal(y0, y1), 'normalised map differs from the separate pass'
    assert torch.equal(o0, o1), 'logits differ from the two-pass form'
    assert o0.abs().max().item() > 0.05
    # a shape the fused form does not serve is refused by the probe, not silently run elsewhere
    z3 = to_act(torch.randn(2, 8, 6, 6, generator=g), dtype=torch.float16)
    w3 = ohwi(torch.randn(16, 8, 3, 3, generator=g))
    y3 = ops.Act.empty(2, 6, 6, 16, dev(), dtype=torch.float16)
    st3 = torch.zeros(2 * 2 * 8, device=dev())
    assert not ops.conv2d_supported(z3, w3, y3, 3, 1, 1, 1, precision='fp16', w_h=w3.half(), xn=(st3, None, None, 1, 'lrelu', None))


@pytest.mark.parametrize('case', [(2, 40, 37, 8, 1), (2, 33, 50, 12, 1), (3, 19, 16, 16, 4), (2, 24, 70, 12, 3)])
def test_conv_heads_data_gradient_taps_as_one_mfma(case):
    """Data gradient of the generators' 1-channel 3x3 heads (and of any <= 4-channel gradient carrier) back to 8 / 12 / 16 channels
    (conv_px_kernel / thin3_mfma_dgrad_kernel; autograd of Conv2dBlock at reference models/inpaint_networks.py:115,230) against torch: act' multiplier, accumulate form,
    ragged widths, padded gradient channels."""
    from hvtest import dev
    from hvgan import ops, lib
    B, H, W, Cout, live = case
    g_ = torch.Generator().manual_seed(Cout + live)
    gy = torch.zeros(B, 4, H, W)
    gy[:, :live] = torch.randn(B, live, H, W, generator=g_)
    w = torch.randn(live, Cout, 3, 3, generator=g_) / 3.0                   # forward conv Cout -> live
    m = torch.randn(B, Cout, H, W, generator=g_)
    ga = ops.Act(gy.permute(0, 2, 3, 1).contiguous().to(dev()).half(), 4, 0)
    ma = ops.Act(m.permute(0, 2, 3, 1).contiguous().to(dev()).half())
    wb = torch.zeros(Cout, 9, 4)
    wb[:, :, :live] = w.reshape(live, Cout, 9).permute(1, 2, 0)             # [layer input channel][tap][gradient channel, padded to 4]
    wb = wb.to(dev())
    y = ops.Act(torch.full((B, H, W, Cout), 0.25, device=dev(), dtype=torch.float16))
    ops.conv2d(ga, wb, y, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'elu'), accumulate=1, cin=4)
    assert lib.get().size('hv_last_kernel_path') in (9, 14)      # (14: conv_px_kernel takes the shape first unless HV_CONV_PX masks it)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(gy[:, :live].half().float(), w.half().float(), None, stride=1, padding=1)
    mh = m.half().float()
    want = 0.25 + ref * torch.where(mh > 0, torch.ones_like(mh), mh + 1)
    got = y.t.float().cpu().permute(0, 3, 1, 2)
    assert (got - want).abs().max().item() <= 6e-3 * max(1.0, want.abs().max().item())
    y2 = ops.Act(torch.zeros(B, H, W, Cout, device=dev(), dtype=torch.float16))
    ops.conv2d(ga, wb, y2, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), cin=4)
    torch.cuda.synchronize()
    got2 = y2.t.float().cpu().permute(0, 3, 1, 2)
    assert (got2 - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())


PX_CASES = [   # B, H, W, Cin, Cout, k, transposed, act, bias, mul, accumulate
    (2, 40, 37, 12, 1, 3, 0, 'clamp', 1, None, 0),      # heads (fp32 image out)
    (2, 33, 70, 8, 1, 3, 0, 'sigmoid', 1, None, 0),
    (2, 24, 50, 8, 16, 3, 1, 'none', 0, 'elu', 1),      # 8 -> 16 and the data gradient of 16 -> 8
    (2, 31, 66, 8, 16, 3, 0, 'elu', 1, None, 0),
    (3, 300, 20, 8, 16, 3, 0, 'none', 0, None, 0),      # more rows than one tile, narrow
    (1, 5, 3, 8, 1, 3, 0, 'none', 0, None, 0),          # smaller than a tile in both directions
    (2, 9, 260, 4, 8, 3, 1, 'none', 0, 'elu', 0),       # two 256-pixel segments per row, four live gradient channels
    (16, 256, 256, 12, 1, 3, 0, 'clamp', 1, None, 0),   # the step's own shape
]


@pytest.mark.parametrize('case', PX_CASES)
def test_conv_thin_full_resolution_layers_one_lane_per_pixel(case):
    """conv_px_kernel (v_mfma_f32_4x4x4f16, lane = pixel): the generators' thin 3x3 / 5x5 layers and their data gradients (Conv2dBlock at reference
    models/inpaint_networks.py:494-503 with cnum/2 .. cnum channels, the heads at :115,230) against torch fp32 on the fp16-rounded operands: bias, activation,
    act' multiplier, accumulate form, ragged widths."""
    from hvtest import to_act, from_act, ohwi, ohwi_T, dev
    from hvgan import ops, lib
    B, H, W, Cin, Cout, k, tr, act, use_b, mul, acc = case
    g = torch.Generator().manual_seed(Cin * 31 + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    xh = to_act(x, dtype=torch.float16)
    xr = x.half().float()
    if not tr:
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(Cout, generator=g) * 0.1 if use_b else None
        ref = _ref_act(F.conv2d(xr, w.half().float(), b, padding=k // 2), act)
        wf = ohwi(w)
        if Cout == 1:
            ya = ops.Act.empty(B, H, W, 1, dev())
        else:
            ya = ops.Act.empty(B, H, W, Cout, dev(), dtype=torch.float16)
        ops.conv2d(xh, wf, ya, k, 1, k // 2, 1, bias=None if b is None else b.to(dev()), act=act, precision='fp16', w_h=wf.half())
    else:
        w = torch.randn(Cin, Cout, k, k, generator=g) / (Cin * k * k) ** 0.5      # conv_transpose2d weight: the data gradient of a Cout -> Cin conv
        ref = F.conv_transpose2d(xr, w.half().float(), stride=1, padding=k // 2)
        wb = ohwi_T(w)
        m = torch.randn(B, Cout, H, W, generator=g)
        y0 = torch.randn(B, Cout, H, W, generator=g) * 0.5
        ya = to_act(y0, dtype=torch.float16)
        if mul:
            mh = m.half().float()
            ref = ref * torch.where(mh > 0, torch.ones_like(mh), mh + 1)
        if acc:
            ref = ref + y0.half().float()
        ops.conv2d(xh, wb, ya, k, 1, k // 2, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(to_act(m, dtype=torch.float16), mul) if mul else None, accumulate=acc)
    assert lib.get().size('hv_last_kernel_path') == 14, (lib.get().cdll.hv_last_kernel_name() or b'').decode()
    torch.cuda.synchronize()
    got = from_act(ya).float()
    assert (got - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())
