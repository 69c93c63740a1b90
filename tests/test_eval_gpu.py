"""In-training evaluation metrics on the device (hv_eval_metrics) against fixture G11 -- outputs of the
reference's own evaluate_model (train.py:50-160; SSIM / PSNR through the restated skimage algorithm, see oracle/make_golden_eval.py)."""
import numpy as np
import pytest
import torch

from test_oracle_golden import g11_cases

pytestmark = pytest.mark.gpu


def _dev(t):
    return t.to('cuda:0') if torch.is_tensor(t) else t


def test_batch_metrics_match_reference_evaluate_model():
    import hvgan  # noqa: F401
    from hvgan import eval_metrics as EM
    for name, b, outs, exp in g11_cases():
        coarse, fine, _, stage2, _, _, pred2 = outs
        m, inp, cb, fb = EM.batch_metrics(_dev(stage2), _dev(fine), _dev(coarse), _dev(pred2), _dev(b['real_B']), _dev(b['real_B_mask']),
                                          _dev(b['normal_vert']), _dev(b['mask']), _dev(b['height']), _dev(b['x1']), _dev(b['x2']), _dev(b['maxheight']))
        m = m.cpu().double().numpy()
        ref = exp['per_sample']
        assert np.abs(m[:, 0] - ref[:, 0]).max() <= 1e-5, (name, 'ssim', m[:, 0], ref[:, 0])
        assert np.abs(m[:, 1] - ref[:, 1]).max() <= 1e-4, (name, 'psnr', m[:, 1], ref[:, 1])
        assert np.abs(m[:, 2] - ref[:, 2]).max() <= 1e-6 and np.abs(m[:, 3] - ref[:, 3]).max() <= 1e-6, (name, 'dice/iou')
        assert np.abs(m[:, 4] - ref[:, 4]).max() <= 1e-4, (name, 'diff_h', m[:, 4], ref[:, 4])
        masked = (inp * _dev(b['mask'])).cpu().numpy()
        for i in range(masked.shape[0]):
            assert np.array_equal(masked[i, 0, ::4, ::4], exp['masked_result_sparse/%d' % i]), (name, i)
        assert torch.equal(cb.cpu(), (coarse > 0.5).float()) and torch.equal(fb.cpu(), (fine > 0.5).float())


def test_evaluate_model_mirror_returns_the_reference_averages():
    """evaluate_model(model, test_loader, ...) -- the reference's signature -- over a loader of two batches with a stand-in model whose netG
    returns the fixture's outputs on the device: the five averages equal the mean of the reference's per-sample values."""
    import hvgan  # noqa: F401
    from hvgan import eval_metrics as EM
    from oracle.make_golden_eval import fake_generator_outputs
    cases = {n: (b, o, e) for n, b, o, e in g11_cases()}

    class Model:
        def __init__(self):
            self.mode = None

        def eval(self):
            self.mode = 'eval'

        def train(self):
            self.mode = 'train'

        def set_input(self, item):
            self.b, self.outs = item
            for k, v in self.b.items():
                setattr(self, k, _dev(v))

        def netG(self, x, mask, cam_inv, ratio):
            assert self.mode == 'eval' and torch.equal(cam_inv.cpu(), 1 - self.b['CAM'])
            return tuple(_dev(t) for t in self.outs)
    model = Model()
    loader = [(cases['b3'][0], cases['b3'][1]), (cases['b5'][0], cases['b5'][1])]
    got = EM.evaluate_model(model, loader, 'cuda:0', '/tmp/x', 3)
    want = np.concatenate([cases['b3'][2]['per_sample'], cases['b5'][2]['per_sample']]).mean(0)
    assert model.mode == 'train' and len(got) == 5
    for g, w, tol in zip(got, want, (1e-5, 1e-4, 1e-6, 1e-6, 1e-4)):
        assert abs(g - w) <= tol, (got, want)
    assert all(np.isnan(v) for v in EM.evaluate_model(model, [], 'cuda:0', '/tmp/x', 0))


def test_eval_metrics_reject_bad_arguments():
    import ctypes
    import hvgan  # noqa: F401
    from hvgan import lib
    L = lib.get()
    z = torch.zeros(1, 1, 8, 8, device='cuda:0')
    h = torch.ones(1, dtype=torch.int64, device='cuda:0')
    out = torch.zeros(5, device='cuda:0')
    ws = torch.zeros(4096, dtype=torch.uint8, device='cuda:0')
    p = lib.ptr
    args = lambda H, W, wsb: (p(z), p(z), p(z), p(z), p(z), p(z), p(z), p(out), p(h), 1, H, W, p(out), p(ws), ctypes.c_size_t(wsb), None)
    assert L.cdll.hv_eval_metrics(*args(6, 8, 4096)) == -2          # 7x7 window does not fit: unsupported, like skimage's ValueError
    assert L.cdll.hv_eval_metrics(*args(8, 8, 8)) == -3             # workspace too small
    assert L.cdll.hv_eval_metrics(None, *args(8, 8, 4096)[1:]) == -1
    assert L.size('hv_eval_metrics_workspace_bytes', 1, 8, 8) > 0 and L.size('hv_eval_metrics_workspace_bytes', 1, 6, 6) == 0
