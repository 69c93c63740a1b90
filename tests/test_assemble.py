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
