"""Batch assembly: host mirror on the CPU, the HIP kernel on the GPU; both against the reference's own
AlignedDataset.__getitem__ outputs (golden G9) and the oracle restatement."""
import numpy as np
import pytest
import torch

from test_oracle_golden import g9_cases

KEYS = ('A', 'B', 'A_mask', 'mask', 'normal_vert', 'CAM')


def test_host_volume_draws_the_reference_slices_and_planes():
    """VertebraVolume (quantise once, z-major) + draw(): same slice, ratio, rows as the reference for the same numpy seed, and its uint8
    planes re-stacked on the host equal the reference's images (no GPU involved)."""
    from hvgan.batch_assembly import VertebraVolume, band_rows
    for name, ct, label, cam, vert_id, normals, nseed, exp in g9_cases():
        v = VertebraVolume(ct, label, cam, vert_id, normals)
        np.random.seed(nseed)
        z, ratio, x1, x2 = v.draw()
        assert (z, x1, x2, x2 - x1) == (exp['slice'], exp['x1'], exp['x2'], exp['height']), name
        assert ratio == exp['slice_ratio']
        min_x, max_x = band_rows(x1, x2, v.H, v.maxheight)

        def restack(p):
            out = np.zeros_like(p)
            out[:min_x] = p[x1 - min_x:x1]
            out[max_x:] = p[x2:x2 + v.H - max_x]
            return out
        assert np.array_equal(v.ct[z], exp['A']) and np.array_equal(restack(v.ct[z]), exp['B']), name
        assert np.array_equal(v.vert[z], exp['A_mask']) and np.array_equal(restack(v.normal[z]), exp['normal_vert']), name
        assert np.array_equal(restack(v.cam[z]), exp['CAM']), name
        band = np.zeros_like(exp['mask'])
        band[min_x:max_x] = 255
        assert np.array_equal(band, exp['mask']), name


def test_assembler_refuses_to_run_without_a_gpu():
    from hvgan.batch_assembly import DeviceBatchAssembler
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        DeviceBatchAssembler([], 'cpu')


@pytest.mark.gpu
def test_device_batch_matches_reference_getitem_bit_exactly():
    """hv_assemble_batch through DeviceBatchAssembler: every float of the six planes equals ToTensor/Normalize of the reference's uint8
    images (G9), metadata included; single items and one mixed batch."""
    from oracle import restate as R
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
    cases = list(g9_cases())
    by_shape = {}
    for c in cases:
        by_shape.setdefault(c[1].shape, []).append(c)
    for shape, group in by_shape.items():
        vols = [VertebraVolume(ct, label, cam, vid, normals, path=name) for name, ct, label, cam, vid, normals, _, _ in group]
        asm = DeviceBatchAssembler(vols, 'cuda:0')
        for i, (name, *_rest, nseed, exp) in enumerate(group):
            np.random.seed(nseed)
            b = asm.batch([i])
            torch.cuda.synchronize()
            for k in KEYS:
                ref = R.to_tensor_u8(exp[k], k in ('A', 'B')).unsqueeze(0)
                assert b[k].shape == ref.shape and b[k].dtype == torch.float32
                assert torch.equal(b[k].cpu(), ref), (name, k)
            assert (int(b['height'][0]), int(b['x1'][0]), int(b['x2'][0]), int(b['h2'][0])) == (exp['height'], exp['x1'], exp['x2'], exp['h2'])
            assert float(b['slice_ratio'][0]) == exp['slice_ratio'] and b['slice_ratio'].dtype == torch.float64
            assert b['A_paths'] == [name]
        # one mixed batch (slices re-drawn from the now cached filtered volumes): against the oracle restatement on the same draws
        idx = list(range(len(group))) * 2
        np.random.seed(77)
        b = asm.batch(idx)
        torch.cuda.synchronize()
        np.random.seed(77)
        for j, i in enumerate(idx):
            _, ct, label, cam, vid, normals, _, _ = group[i]
            item = R.dataset_item(ct.astype(np.float64), label.astype(np.float64), cam.astype(np.float64) * 255, vid, normals)
            for k in KEYS:
                assert torch.equal(b[k][j].cpu(), item[k]), (j, k)
            assert int(b['x1'][j]) == item['x1'] and int(b['x2'][j]) == item['x2'] and b['slice'][j] == item['slice']


@pytest.mark.gpu
def test_device_batch_scalar_path_and_band_validation():
    """Slice width not a multiple of 4 (scalar kernel) against the oracle; a vertebra taller than the band is refused like the reference."""
    from oracle import restate as R
    from hvgan import synth
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
    ct, label, cam = synth.make_spine_volume(9, H=90, W=50, Z=8)
    asm = DeviceBatchAssembler([VertebraVolume(ct, label, cam, 12, ['11'])], 'cuda:0')
    np.random.seed(5)
    b = asm.batch([0, 0, 0])
    torch.cuda.synchronize()
    np.random.seed(5)
    for j in range(3):
        item = R.dataset_item(ct.astype(np.float64), label.astype(np.float64), cam.astype(np.float64) * 255, 12, ['11'])
        for k in KEYS:
            assert torch.equal(b[k][j].cpu(), item[k]), (j, k)
    v = VertebraVolume(ct, label, cam, 12, ['11'])
    v.draw = lambda: (3, 0.0, 10, 60)          # 50 rows: does not fit the 40-row band
    with pytest.raises(ValueError):
        DeviceBatchAssembler([v], 'cuda:0').batch([0])


@pytest.mark.gpu
def test_assembled_batch_drives_a_train_step():
    """The assembler's dict goes straight into Pix2PixModel.set_input / optimize_parameters (256 x 256 slices, batch 2)."""
    import bench
    from hvgan import synth
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
    from hvgan.models.pix2pix_model import Pix2PixModel
    ct, label, cam = synth.make_spine_volume(3, H=256, W=256, Z=12, pitch=48)
    asm = DeviceBatchAssembler([VertebraVolume(ct, label, cam, 12, ['11', '13']), VertebraVolume(ct, label, cam, 11, ['12'])], 'cuda:0')
    np.random.seed(1)
    batch = asm.batch([0, 1])
    torch.manual_seed(0)
    opt = bench.make_opt('fp32')
    model = Pix2PixModel(opt)
    model.setup(opt)
    model.set_input(batch)
    model.optimize_parameters()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert torch.equal(model.real_B.cpu(), batch['A'].cpu()) and torch.equal(model.mask.cpu(), batch['mask'].cpu())


def _cubic_spine(seed, n=128):
    """Synthetic volume whose BOTH in-plane axes can serve as the slicing axis (config #5: sagittal + coronal slices of one volume)."""
    from hvgan import synth
    return synth.make_spine_volume(seed, H=n, W=n, Z=n, pitch=24, n_vert=4)


def test_coronal_view_is_the_axis_swapped_volume():
    """view='coronal' slices along the reference's axis 1 (`[:, z, :]`, evaluation/RHLV_quantification_coronal.py:51-54): the planes are those of the
    sagittal construction on the axis-swapped arrays, and a draw follows the reference's item arithmetic on them (oracle.restate.dataset_item_u8)."""
    from hvgan.batch_assembly import VertebraVolume
    from oracle import restate as R
    ct, label, cam = _cubic_spine(5, 64)
    vc = VertebraVolume(ct, label, cam, 11, ['10', '12'], view='coronal')
    sw = lambda v: np.swapaxes(v, 1, 2)
    vs = VertebraVolume(sw(ct), sw(label), sw(cam), 11, ['10', '12'])
    for a in ('ct', 'cam', 'normal', 'vert'):
        assert np.array_equal(getattr(vc, a), getattr(vs, a)), a
    assert (vc.H, vc.W, vc.Z) == (64, 64, 64) and vc.view == 'coronal'
    np.random.seed(9)
    z, ratio, x1, x2 = vc.draw()
    np.random.seed(9)
    ref = R.dataset_item_u8(sw(ct).astype(np.float64), sw(label).astype(np.float64), sw(cam).astype(np.float64) * 255, 11, ['10', '12'])
    assert (x1, x2) == (ref['x1'], ref['x2']) and np.array_equal(vc.ct[z], ref['A'])
    with pytest.raises(ValueError):
        VertebraVolume(ct, label, cam, 11, [], view='axial')


@pytest.mark.gpu
def test_sagittal_and_coronal_feed_at_512_drives_train_steps(monkeypatch):
    """BASELINE config #5's feed: slices of BOTH views of the same volumes at 512 x 512 (h2 = 80), assembled on the device, alternately through the
    same train step (fp16 mode, graph replay from the third step on).  The coronal batch equals the oracle's item arithmetic on the axis-swapped
    volume float for float; the losses stay finite and no step is skipped by the overflow guard."""
    monkeypatch.setenv('HV_PRECISION', 'fp16')
    import bench
    from oracle import restate as R
    from hvgan import synth
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
    from hvgan.models.pix2pix_model import Pix2PixModel
    # (a cubic 512^3 float volume would be 0.5 GB per array: a 512 x 512 x 24 sagittal stack and a 512 x 24 x 512 coronal one stand for the two views)
    cs, ls, ms = synth.make_spine_volume(7, H=512, W=512, Z=24, pitch=96, n_vert=4)
    cc, lc, mc = (np.swapaxes(v, 1, 2) for v in synth.make_spine_volume(8, H=512, W=512, Z=24, pitch=96, n_vert=4))       # [H, Z', W]: coronal slicing axis = 1
    sag = [VertebraVolume(cs, ls, ms, 11, ['10', '12'], maxheight=80), VertebraVolume(cs, ls, ms, 12, ['11'], maxheight=80)]
    cor = [VertebraVolume(cc, lc, mc, 11, ['10', '12'], maxheight=80, view='coronal'), VertebraVolume(cc, lc, mc, 12, ['11'], maxheight=80, view='coronal')]
    a_s, a_c = DeviceBatchAssembler(sag, 'cuda:0'), DeviceBatchAssembler(cor, 'cuda:0')
    np.random.seed(4)
    b = a_c.batch([0])
    np.random.seed(4)
    sw = lambda v: np.swapaxes(v, 1, 2)
    ref = R.dataset_item(sw(cc).astype(np.float64), sw(lc).astype(np.float64), sw(mc).astype(np.float64) * 255, 11, ['10', '12'], maxheight=80)
    for k in KEYS:
        assert torch.equal(b[k][0].cpu(), ref[k]), k
    torch.manual_seed(0)
    opt = bench.make_opt('fp16')
    model = Pix2PixModel(opt)
    model.setup(opt)
    np.random.seed(2)
    for it in range(4):
        model.set_input((a_s if it % 2 == 0 else a_c).batch([0, 1]))
        model.optimize_parameters()
    torch.cuda.synchronize()
    losses = model.get_current_losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    assert model.real_B.shape[-2:] == (512, 512) and sum(model.overflow_steps().values()) == 0


@pytest.mark.gpu
def test_component_filter_of_a_resident_volume_runs_on_the_device():
    """Row f1's residue: the loader's remove_small_connected_components (data/aligned_dataset.py:16-31) for every slice of a volume at once
    (hv_slice_components_u8, in place on the uint8 mask plane).  Against scipy.ndimage.label on slices with specks, 8-connected diagonal
    chains and a spiral: the filtered planes byte for byte, and the per-slice table (count, first row, last row) the draw reads."""
    import ctypes
    from test_infer_gpu import _blob_slices
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler, _remove_small_components
    L = lib.get()
    dev = torch.device('cuda:0')
    S, H, W = 7, 128, 96
    blobs = _blob_slices(5, S, H, W, 20.0)                     # [S][H][W] labels, specks below 50 pixels among them
    plane = ((blobs == 20.0) * 255).astype(np.uint8)
    exp = np.stack([_remove_small_components((plane[s] > 0).astype(np.float64), 50) for s in range(S)])
    assert (exp.sum(axis=(1, 2)) < (plane > 0).sum(axis=(1, 2))).any(), 'the case must contain components the filter drops'
    d = torch.from_numpy(plane).to(dev)
    need = L.size('hv_slice_components_workspace_bytes', S, H, W)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    stats = torch.full((S, 4), -7, dtype=torch.int32, device=dev)
    out = torch.full_like(d, 9)
    L.call('hv_slice_components_u8', ptr(d), S, H, W, 255, 50, ptr(stats), ptr(out), ptr(ws), ctypes.c_size_t(need), stream())
    assert np.array_equal(out.cpu().numpy(), (exp * 255).astype(np.uint8))
    L.call('hv_slice_components_u8', ptr(d), S, H, W, 255, 50, ptr(stats), ptr(d), ptr(ws), ctypes.c_size_t(need), stream())      # in place
    assert torch.equal(d, out)
    got = stats.cpu().numpy()
    for s in range(S):
        rows = np.flatnonzero(exp[s].any(axis=1))
        assert list(got[s][:3]) == [int(exp[s].sum()), int(rows.min()) if rows.size else -1, int(rows.max()) if rows.size else -1], s
    # through the assembler: a volume [H][W][Z] whose vertebra 20 carries those shapes -> device table == host mirror for every slice,
    # resident mask plane == host-filtered plane, and no scipy call after construction
    label = np.moveaxis(blobs, 0, 2).astype(np.float64)
    ct = np.random.RandomState(0).randint(0, 255, label.shape).astype(np.float64)
    cam = np.random.RandomState(1).rand(*label.shape)
    v_dev, v_host = (VertebraVolume(ct, label, cam, 20, ['21']) for _ in range(2))
    asm = DeviceBatchAssembler([v_dev], 'cuda:0')
    for z in range(S):
        assert v_dev._info[z] == v_host.slice_info(z), z
    assert np.array_equal(asm._planes[0][1].cpu().numpy(), v_host.vert)
    import scipy.ndimage
    real = scipy.ndimage.label

    def refuse(*a, **k):
        raise AssertionError('the training loop called scipy.ndimage.label')
    scipy.ndimage.label = refuse
    try:
        v_dev.maxheight = H            # accept any vertebra height: this case is about the filter, not the band
        np.random.seed(3)
        b = asm.batch([0, 0, 0])
        torch.cuda.synchronize()
    finally:
        scipy.ndimage.label = real
    for j, z in enumerate(b['slice']):
        assert np.array_equal((b['A_mask'][j, 0].cpu().numpy() * 255).astype(np.uint8), v_host.vert[z])
