"""Batched inference tensor path (eval_3d_sagittal_twostage.run_model :96-130) against the CPU oracle."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage_batched_synthesis_matches_oracle():
    import hvgan
    from hvgan import synth, infer
    from hvgan.models.inpaint_networks import Generator
    from oracle import restate as R
    torch.manual_seed(3)
    net = Generator({'input_dim': 1, 'ngf': 16}, True)
    net.fine_generator.fc_height.bias.data.fill_(0.37)     # keep pred_h*40 away from an integer (the ceil() discontinuity)
    net.fine_generator.fc_height.weight.data.mul_(1e-2)
    net.cuda().train()
    b = synth.to_model_inputs(synth.make_batch(3, 256, seed=9))
    dev = torch.device('cuda:0')
    for _ in range(3):   # a never-trained spectral norm has random u/v (sigma far too small): let the power iteration settle
        net.run_forward(b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev), training=True)
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    label = b['real_B_mask'] * 20.0
    lab, ct, pred = infer.synthesize(net, b['real_A'].to(dev), b['mask'].to(dev), b['CAM'].to(dev), b['slice_ratio'].to(dev),
                                     b['real_B'].to(dev), label.to(dev), b['x1'].to(dev), b['x2'].to(dev), b['height'].to(dev), 20)
    # the reference synthesises slice by slice at batch 1 (every slice's own mask selects its attention patches): oracle per sample
    assert not torch.equal(b['mask'][0], b['mask'][1]) or not torch.equal(b['mask'][0], b['mask'][2])
    with torch.no_grad():
        per = [R.generator_forward(sd, b['real_A'][i:i + 1], b['mask'][i:i + 1], 1 - b['CAM'][i:i + 1], b['slice_ratio'][i:i + 1],
                                   training=False)[0] for i in range(3)]
    cs, fs, x1s, x2s, p1, p2 = (torch.cat([o[k] for o in per]) for k in range(6))
    assert (pred.cpu() - p2.view(-1)).abs().max().item() <= 1e-3
    ph = p2.view(-1) * 40
    # keep the comparison away from the ceil() discontinuity
    assert all(abs(float(v) - round(float(v))) > 1e-3 for v in ph), ph
    ref_ct = (R.shrm_composite(x2s, b['real_B'], ph, b['height'], b['x1'], b['x2']) + 1) * 127.5
    assert (ct.cpu() - ref_ct[:, 0]).abs().max().item() <= 0.15          # 1e-3 in [-1,1] units
    seg = (fs > 0.5).float() * 20
    ref_lab = R.shrm_composite(seg, label, ph, b['height'], b['x1'], b['x2'])
    assert ((lab.cpu() - ref_lab[:, 0]).abs() > 0).float().mean().item() <= 1e-4


@pytest.mark.parametrize('nz', [6, 64])
def test_stage_batched_volume_matches_sequential_oracle(nz):
    """process_volume (3 batched launches per volume) == the reference's per-slice chain restated with the CPU oracle.
    nz = 64: BASELINE config #4's full straightened volume (256 x 256 x 64; stage batches of ~45-50 slices -- other tile / split-K choices than
    the 6-slice case), checked on the first, the middle and the last processed slice."""
    import numpy as np
    import hvgan
    from hvgan import synth, infer
    from hvgan.models.inpaint_networks import Generator
    from oracle import restate as R
    torch.manual_seed(5)
    net = Generator({'input_dim': 1, 'ngf': 16}, True)
    net.fine_generator.fc_height.bias.data.fill_(0.41)
    net.fine_generator.fc_height.weight.data.mul_(1e-2)
    net.cuda().train()
    dev = torch.device('cuda:0')
    b = synth.to_model_inputs(synth.make_batch(2, 256, seed=3))
    for _ in range(3):
        net.run_forward(b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev), training=True)
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ct, label, cam = synth.make_volume(nz=nz, size=256, seed=2)
    cam255 = cam * 255
    out_ct, out_seg = infer.process_volume(net, ct, label, cam255, 20, dev)

    def oracle_run(cam2d, lab2d, ct2d, vid, ratio):
        p = infer.prepare_slice(cam2d, lab2d, ct2d, vid)
        if p is None:
            return None
        t = lambda a: torch.from_numpy(a)[None, None]
        with torch.no_grad():
            (cs, fs, s1, s2, p1, p2), _ = R.generator_forward(sd, t(p['ct_masked']), t(p['mask']), 1 - t(p['cam']),
                                                              torch.tensor([ratio], dtype=torch.float64), training=False)
        ph = p2.view(-1) * 40
        h, x1, x2 = (torch.tensor([p[k]]) for k in ('height', 'x1', 'x2'))
        fb = (R.shrm_composite(s2, t(p['ori_ct']), ph, h, x1, x2) + 1) * 127.5
        seg = R.shrm_composite((fs > 0.5).float() * vid, t(lab2d.astype(np.float32)), ph, h, x1, x2)
        return seg[0, 0].numpy().astype(np.float64), fb[0, 0].numpy().astype(np.float64)

    zs = [z for z in range(ct.shape[2]) if out_seg[:, :, z].any()]
    assert len(zs) >= (3 if nz == 6 else 40)
    # z-range arithmetic of process_nii_files (eval_3d_sagittal_twostage.py:186-199)
    zhas = np.flatnonzero((label == 20).any(axis=(0, 1)))
    rng_len = int(zhas.max()) - int(zhas.min()) + 1
    new_len = int(rng_len * 4 / 5)
    nz0 = int(zhas.min()) + (rng_len - new_len) // 2
    centre = (nz0 + nz0 + new_len - 1) // 2
    assert zs[0] >= nz0 and zs[-1] <= nz0 + new_len - 1
    bad = 0
    for z in (zs[:3] if nz == 6 else [zs[0], zs[len(zs) // 2], zs[-1]]):
        ratio = abs(z - centre) / rng_len * 2
        l, c = label[:, :, z], ct[:, :, z]
        for vid in (19, 21, 20):
            r = oracle_run(cam255[:, :, z], l, c, vid, ratio)
            if r is not None:
                l, c = r
        # a stage's float output is truncated to uint8 by the next stage: where the device and the CPU generator differ by 1e-3 across an
        # integer the re-stacked pixel differs by exactly 1 -- a handful of pixels per slice, everything else within float rounding
        dct = np.abs(out_ct[:, :, z] - c)
        assert dct.max() <= 1.0 + 1e-3 and (dct > 0.6).mean() <= 1e-3, (dct.max(), (dct > 0.6).mean())
        assert np.median(dct) <= 1e-3
        bad += (out_seg[:, :, z] != l).mean()
    assert bad / 3 <= 1e-3


def test_eval_forward_graph_replay_matches_eager():
    """Generator.forward in eval mode under no_grad replays a captured hipGraph per input shape: same 7-tuple as eager launches,
    for changing inputs and after a weight update in place."""
    import hvgan
    from hvgan import synth
    from hvgan.models.inpaint_networks import Generator
    torch.manual_seed(21)
    net = Generator({'input_dim': 1, 'ngf': 16}, True).cuda().eval()
    dev = torch.device('cuda:0')

    def args(seed, B):
        b = synth.to_model_inputs(synth.make_batch(B, 256, seed=seed))
        return [b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev)]

    def both(a):
        with torch.no_grad():
            net.use_graph = False
            e = net(*a)
            net.use_graph = True
            g = net(*a)
        return e, g

    for seed, B in ((1, 1), (2, 1), (3, 2), (4, 1)):
        e, g = both(args(seed, B))
        for i in (0, 1, 2, 3, 5, 6):
            assert torch.equal(e[i], g[i]), (seed, i)
    with torch.no_grad():
        net.coarse_generator.conv5.conv.weight_orig.mul_(1.05)          # in-place update: the graph re-prepares the weights
    e, g = both(args(5, 1))
    assert torch.equal(e[3], g[3]) and torch.equal(e[0], g[0])
    assert len(net._eval_graphs) == 2


def test_recomposite_matches_reference_run_model_outputs():
    """infer.recomposite (device re-compositing of run_model :103-130) on the stand-in network outputs of fixture G10: label slice exact,
    CT slice within float32 rounding of the reference's (x + 1) * 127.5."""
    import numpy as np
    import hvgan  # noqa: F401
    from hvgan import infer
    from oracle import restate as R
    from oracle.make_golden_infer import fake_outputs
    from test_oracle_golden import g10_cases
    dev = torch.device('cuda:0')
    n = 0
    for i, (name, ct, label, cam, vert_id, ratio, model, exp) in enumerate(g10_cases()):
        if exp is None:
            continue
        p = R.infer_prepare(cam, label, ct, vert_id)
        seg, raw = fake_outputs(500 + i)
        t = lambda a: a.to(dev)
        iv = lambda v: torch.tensor([v], dtype=torch.int64, device=dev)
        lab, ctf = infer.recomposite(t(raw), t(seg), torch.tensor([[model.frac]], device=dev), t(p['ori_ct'][None]),
                                     t(torch.from_numpy(label.astype(np.float32))[None, None]), iv(p['x1']), iv(p['x2']), iv(p['height']), vert_id)
        assert np.array_equal(lab[0].cpu().numpy().astype(np.float64), exp['label_fake']), name
        assert np.abs(ctf[0].cpu().numpy().astype(np.float64) - exp['ct_fake']).max() <= 5e-5, name
        n += 1
    assert n >= 6


def _blob_slices(seed, S, H, W, value):
    """Label slices with assorted shapes: rectangles, diagonal chains (8- but not 4-connected), a spiral (long label chains), specks."""
    import numpy as np
    rng = np.random.RandomState(seed)
    lab = np.zeros((S, H, W), dtype=np.float32)
    for s in range(S):
        for _ in range(rng.randint(1, 5)):
            r, c, h, w = rng.randint(0, H - 30), rng.randint(0, W - 30), rng.randint(2, 30), rng.randint(2, 30)
            lab[s, r:r + h, c:c + w] = value if rng.rand() < 0.7 else value + 1
        for _ in range(6):
            r, c = rng.randint(0, H - 2), rng.randint(0, W - 3)
            lab[s, r:r + rng.randint(1, 3), c:c + rng.randint(1, 4)] = value          # specks
        r, c = rng.randint(0, H - 70), rng.randint(0, W - 70)
        for i in range(60):                                                           # diagonal chain, 60 pixels
            lab[s, r + i, c + i] = value
    # one spiral: a single component whose label has to travel ~1500 pixels
    sp = np.zeros((H, W), dtype=np.float32)
    r0, r1, c0, c1 = 10, H - 10, 10, W - 10
    while r1 - r0 > 8 and c1 - c0 > 8:
        sp[r0, c0:c1] = value; sp[r0:r1, c1 - 1] = value; sp[r1 - 1, c0 + 4:c1] = value; sp[r0 + 4:r1, c0 + 4] = value
        sp[r0 + 4, c0 + 4:c0 + 9] = value
        r0 += 8; c0 += 8; r1 -= 8; c1 -= 8
    lab[0] = sp
    lab[1] = 0                                                                        # empty slice
    return lab


def test_slice_components_match_scipy():
    """hv_slice_components (device connected components + small-component filter + row statistics) against scipy.ndimage.label."""
    import ctypes
    import numpy as np
    from scipy.ndimage import label as cc_label
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    L = lib.get()
    dev = torch.device('cuda:0')
    for (S, H, W, seed) in ((6, 128, 96, 1), (5, 256, 256, 2)):
        lab = _blob_slices(seed, S, H, W, 20.0)
        d = torch.from_numpy(lab).to(dev)
        need = L.size('hv_slice_components_workspace_bytes', S, H, W)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        stats = torch.full((S, 4), -7, dtype=torch.int32, device=dev)
        L.call('hv_slice_components', ptr(d), S, H, W, ctypes.c_float(20.0), 50, ptr(stats), ptr(ws), ctypes.c_size_t(need), stream())
        got = stats.cpu().numpy()
        for s in range(S):
            m = (lab[s] == 20.0).astype(np.int32)
            cl, n = cc_label(m, np.ones((3, 3), dtype=np.int32))
            for i in range(1, n + 1):
                if np.sum(cl == i) < 50:
                    m[cl == i] = 0
            rows = np.argwhere(m)[:, 0]
            exp = [len(rows), rows.min() if len(rows) else -1, rows.max() if len(rows) else -1, rows.sum() if len(rows) else 0]
            assert list(got[s]) == [int(v) for v in exp], (S, s, got[s], exp)


def test_slice_count_matches_numpy():
    """hv_slice_count: the `np.sum(label[:, :, z] == neighbour)` of the volume driver (eval_3d_sagittal_twostage.py:208,217), exact integers."""
    import ctypes
    import numpy as np
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    L = lib.get()
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(4)
    for S, per in ((51, 256 * 256), (3, 1000), (1, 1)):
        lab = rng.randint(8, 12, (S, per)).astype(np.float32)
        lab[0] = 3.0                                                     # a slice without the value
        d = torch.from_numpy(lab).to(dev)
        cnt = torch.full((S,), -1, dtype=torch.int32, device=dev)
        L.call('hv_slice_count', ptr(d), S, ctypes.c_longlong(per), ctypes.c_float(9.0), ptr(cnt), stream())
        assert np.array_equal(cnt.cpu().numpy(), (lab == 9.0).sum(axis=1).astype(np.int32))


def test_infer_prepare_matches_reference_network_inputs():
    """hv_slice_components + hv_infer_prepare on the G10 slices as one batch: the generator's input planes equal what the reference's
    run_model fed its network, bit for bit; rows / height / presence flags too."""
    import ctypes
    import numpy as np
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    from oracle import restate as R
    from test_oracle_golden import g10_cases
    L = lib.get()
    dev = torch.device('cuda:0')
    cases = list(g10_cases())
    S = len(cases)
    H, W = cases[0][1].shape
    vids = sorted({c[4] for c in cases})
    for vid in vids:          # one stage = one vertebra id for all slices of the batch
        f32 = lambda k: torch.from_numpy(np.stack([c[k] for c in cases]).astype(np.float32)).to(dev)
        ct, lab, cam = f32(1), f32(2), f32(3)
        need = L.size('hv_slice_components_workspace_bytes', S, H, W)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        stats = torch.empty(S, 4, dtype=torch.int32, device=dev)
        L.call('hv_slice_components', ptr(lab), S, H, W, ctypes.c_float(float(vid)), 50, ptr(stats), ptr(ws), ctypes.c_size_t(need), stream())
        planes = torch.empty(4, S, 1, H, W, device=dev)
        x1, x2, height = (torch.empty(S, dtype=torch.int64, device=dev) for _ in range(3))
        valid = torch.empty(S, dtype=torch.int32, device=dev)
        L.call('hv_infer_prepare', ptr(ct), ptr(cam), ptr(stats), None, S, H, W, 40, ptr(planes[0]), ptr(planes[1]), ptr(planes[2]), ptr(planes[3]),
               ptr(x1), ptr(x2), ptr(height), ptr(valid), stream())
        pl = planes.cpu()
        for s, (name, c_ct, c_lab, c_cam, c_vid, ratio, model, exp) in enumerate(cases):
            q = R.infer_prepare(c_cam, c_lab, c_ct, vid)           # oracle for THIS stage's vertebra id (the fixture pins the case's own id)
            assert int(valid[s]) == (q is not None), (name, vid)
            if q is None:
                assert (int(x1[s]), int(x2[s]), int(height[s])) == (0, 0, H)
                continue
            assert (int(x1[s]), int(x2[s]), int(height[s])) == (q['x1'], q['x2'], q['height']), (name, vid)
            assert torch.equal(pl[0, s], q['ct_batch']) and torch.equal(pl[1, s], q['ori_ct']), (name, vid)
            assert torch.equal(pl[2, s], q['mask_batch']) and torch.equal(pl[3, s], q['cam']), (name, vid)
            if c_vid == vid and exp is not None:
                assert torch.equal(pl[0, s], R.to_tensor_u8(exp['in_ct'], True)) and torch.equal(pl[2, s], R.to_tensor_u8(exp['in_mask'], False))
                assert torch.equal(pl[3, s], R.to_tensor_u8(exp['in_cam'], False)) and int(height[s]) == int(exp['height'][0])


def test_slice_kernels_reject_bad_arguments_and_select_semantics():
    """Error convention of the slice-preparation entry points (no exception across the ABI, negative status) and hv_select_slices' two modes."""
    import ctypes
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    L = lib.get()
    dev = torch.device('cuda:0')
    lab = torch.zeros(2, 16, 16, device=dev)
    stats = torch.zeros(2, 4, dtype=torch.int32, device=dev)
    ws = torch.empty(64, dtype=torch.uint8, device=dev)                     # far too small
    rc = L.cdll.hv_slice_components(ptr(lab), 2, 16, 16, ctypes.c_float(1.0), 50, ptr(stats), ptr(ws), ctypes.c_size_t(64), stream())
    assert rc == -3                                                         # HV_ERR_WORKSPACE
    assert L.cdll.hv_slice_components(None, 2, 16, 16, ctypes.c_float(1.0), 50, ptr(stats), ptr(ws), ctypes.c_size_t(64), stream()) == -1
    with pytest.raises(RuntimeError, match='HV_ERR_WORKSPACE'):
        L.call('hv_slice_components', ptr(lab), 2, 16, 16, ctypes.c_float(1.0), 50, ptr(stats), ptr(ws), ctypes.c_size_t(64), stream())
    src = torch.arange(3 * 8, dtype=torch.float32, device=dev).view(3, 8)
    flag = torch.tensor([1, 0, 1], dtype=torch.int32, device=dev)
    keep = torch.full((3, 8), -1.0, device=dev)
    L.call('hv_select_slices', ptr(flag), ptr(src), ptr(keep), 3, ctypes.c_longlong(8), 1, stream())
    zero = torch.full((3, 8), -1.0, device=dev)
    L.call('hv_select_slices', ptr(flag), ptr(src), ptr(zero), 3, ctypes.c_longlong(8), 0, stream())
    torch.cuda.synchronize()
    assert torch.equal(keep[0], src[0]) and torch.equal(keep[2], src[2]) and bool((keep[1] == -1).all())
    assert torch.equal(zero[0], src[0]) and bool((zero[1] == 0).all())
    assert L.cdll.hv_select_slices(ptr(flag), ptr(src), ptr(zero), 0, ctypes.c_longlong(8), 0, stream()) == -1


def test_volume_intake_kernels_match_numpy():
    """hv_volume_scan / hv_volume_slices / hv_volume_merge against the reference's own numpy expressions (eval_3d_sagittal_twostage.py:186-190 `np.any(label ==
    vert_id)` per slice, :208,:217 `np.sum(label[:, :, z] == neighbour)`, the z-range cut of the float64 [H, W, Z] arrays, :236-239 zeros + per-slice
    assignment): exact integers, float32 casts bit for bit; a Z that is not a multiple of the 32-slice tile and an H*W that is not a multiple of 64."""
    import ctypes
    import numpy as np
    import hvgan
    from hvgan import lib as _lib
    L = _lib.get()
    rng = np.random.RandomState(3)
    H, W, Z = 37, 29, 45
    label = rng.choice([0, 0, 0, 7, 8, 9, 20], size=(H, W, Z)).astype(np.float64)
    ct = rng.uniform(0, 255.99, size=(H, W, Z))
    dev = torch.device('cuda:0')
    dl, dc = torch.from_numpy(label).to(dev), torch.from_numpy(ct).to(dev)
    counts = torch.full((3 * Z,), -1, dtype=torch.int32, device=dev)
    L.call('hv_volume_scan', _lib.ptr(dl), ctypes.c_longlong(H * W), Z, ctypes.c_double(8.0), ctypes.c_double(7.0), ctypes.c_double(-1.0), _lib.ptr(counts), _lib.stream())
    got = counts.cpu().numpy().reshape(3, Z)
    assert (got[0] == (label == 8).sum(axis=(0, 1))).all() and (got[1] == (label == 7).sum(axis=(0, 1))).all() and (got[2] == 0).all()
    z0, S = 5, 33
    out = torch.empty(S, H * W, dtype=torch.float32, device=dev)
    L.call('hv_volume_slices', _lib.ptr(dc), ctypes.c_longlong(H * W), Z, z0, S, _lib.ptr(out), _lib.stream())
    want = np.ascontiguousarray(ct[:, :, z0:z0 + S].astype(np.float32).reshape(H * W, S).T)
    assert np.array_equal(out.cpu().numpy(), want)
    flag = torch.from_numpy((rng.rand(S) > 0.3).astype(np.int32)).to(dev)
    vol = torch.full((H * W * Z,), 7.0, dtype=torch.float64, device=dev)
    L.call('hv_volume_merge', _lib.ptr(out), _lib.ptr(flag), ctypes.c_longlong(H * W), Z, z0, S, _lib.ptr(vol), _lib.stream())
    ref = np.zeros((H, W, Z))
    f = flag.cpu().numpy()
    for s in range(S):
        if f[s]:
            ref[:, :, z0 + s] = want[s].reshape(H, W)
    assert np.array_equal(vol.cpu().numpy().reshape(H, W, Z), ref)


def test_pipelined_volumes_equal_single_volume_calls():
    """infer.process_volumes (double-buffered: the next volume's pinned copy, upload and scan and the previous volume's download overlap the current
    volume's stages) returns, volume for volume, exactly what process_volume returns for each alone -- fresh arrays and pinned views -- including a volume
    without the vertebra (all zeros, no stage runs) in the middle of the stream and a different vertebra id (other neighbours)."""
    import numpy as np
    import hvgan
    from hvgan import synth, infer
    from hvgan.models.inpaint_networks import Generator
    torch.manual_seed(5)
    net = Generator({'input_dim': 1, 'ngf': 16}, True)
    net.fine_generator.fc_height.bias.data.fill_(0.41)
    net.fine_generator.fc_height.weight.data.mul_(1e-2)
    net.cuda().eval()
    dev = torch.device('cuda:0')
    vols = []
    for seed, nz, vid in ((2, 16, 20), (3, 12, 20), (4, 16, 21), (5, 16, 20)):
        ct, label, cam = synth.make_volume(nz=nz, size=256, seed=seed)
        vols.append((ct, label, cam * 255, vid))
    ct, label, cam = synth.make_volume(nz=16, size=256, seed=6)
    vols.insert(2, (ct, np.where(label == 20, 0.0, label), cam * 255, 20))          # the vertebra is absent
    single = [infer.process_volume(net, *v[:3], v[3], dev) for v in vols]
    assert not single[2][0].any() and not single[2][1].any()
    assert all(s[1].any() for i, s in enumerate(single) if i != 2)
    for copy in (True, False):
        n = 0
        for i, (oc, os_) in enumerate(infer.process_volumes(net, vols, dev, copy=copy)):
            assert oc.dtype == np.float64 and oc.shape == vols[i][0].shape
            assert np.array_equal(oc, single[i][0]) and np.array_equal(os_, single[i][1]), (copy, i)
            n += 1
        assert n == len(vols)
