"""Device RHLV quantification (hv_rhlv through hvgan.evaluation) against the reference's outputs (fixture G8) and the CPU oracle.
Integer steps (column counts, thirds, centre columns, selections) are exact; the final means differ from numpy's only in the
summation grouping of doubles: tolerance 1e-12 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= TOL * np.maximum(1.0, np.abs(b)))


def test_rhlv_matches_reference_golden():
    import hvgan
    from hvgan import evaluation
    g = load_golden('g8_rhlv')
    names = sorted({k.split('/')[0] for k in g.keys()})
    for n in names:
        idx, div, thr, cz, length = (float(v) for v in g[n + '/params'])
        fake = torch.from_numpy(np.asarray(g[n + '/fake'])).cuda()       # uint8 volumes, [H, W, Z]
        label = torch.from_numpy(np.asarray(g[n + '/label'])).cuda()
        res, means = evaluation.rhlv_volume(fake, label, idx, int(div), thr, return_means=True)
        assert _close(res, g[n + '/out']), (n, res, g[n + '/out'])
        assert _close(means, g[n + '/means']), (n, means)
        # explicit slice range, binary float volumes, a permuted memory layout ([Z, H, W] storage viewed as [H, W, Z])
        sf = (fake == idx).float().permute(2, 0, 1).contiguous().permute(1, 2, 0)
        sl = (label == idx).float().permute(2, 0, 1).contiguous().permute(1, 2, 0)
        res2 = evaluation.calculate_rhlv(sf, sl, int(cz), int(length), 'v', thr)
        assert _close(res2, g[n + '/out']), (n, res2)


def test_rhlv_random_volumes_match_oracle_and_edge_cases():
    import hvgan
    from hvgan import evaluation, synth
    from oracle import restate as R
    for seed in range(8):
        kw = dict(seed=100 + seed, H=40 + 8 * (seed % 3), W=48 + 16 * (seed % 2), Z=10 + seed, collapse=0.1 * (seed % 5), fake_shorter=seed % 4 == 3)
        fake, label = synth.make_rhlv_pair(**kw)
        ref, means = R.rhlv_volume(fake, label, 20, 3 + seed % 3, 0.5 + 0.05 * seed)
        got, gm = evaluation.rhlv_volume(torch.from_numpy(fake).float().cuda(), torch.from_numpy(label).float().cuda(), 20, 3 + seed % 3,
                                         0.5 + 0.05 * seed, return_means=True)
        assert _close(got, ref) and _close(gm, means), (seed, got, ref)
    # the original volume does not contain the vertebra: the reference skips it
    fake, label = synth.make_rhlv_pair(seed=1)
    assert evaluation.rhlv_volume(torch.from_numpy(fake).float().cuda(), torch.from_numpy(label).float().cuda(), 33) is None
    # generated volume empty: every slice is skipped, all means are zero
    res = evaluation.rhlv_volume(torch.zeros(64, 64, 16).cuda(), torch.from_numpy(label).float().cuda(), 20)
    ref, _ = R.rhlv_volume(np.zeros((64, 64, 16)), label, 20)
    assert _close(res, ref)
