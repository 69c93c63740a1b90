"""Bisect which part of the train step survives hipGraph capture (each case in its own process)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = ['phase_a', 'phase_b', 'phase_c']


def run_case(case):
    import faulthandler
    faulthandler.enable()
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from test_step_gpu import make_opt
    import hvgan
    from hvgan import synth
    from hvgan.models.pix2pix_model import Pix2PixModel
    torch.manual_seed(0)
    if case == 'phase_a_serial':
        os.environ['HV_CONCURRENT_D'] = '0'
    if case == 'phase_a_dstreams':
        pass
    if case == 'phase_a_wgrad':
        os.environ['HV_CONCURRENT_D'] = '0'
    model = Pix2PixModel(make_opt())
    model.use_graph = False
    model.set_input(synth.make_batch(2, 256, seed=1))
    for _ in range(2):
        model.optimize_parameters()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()

    def body():
        if case == 'affine':
            from hvgan import ops
            ops.axpy(model._loss_buf[0:1].view(()), model._loss_buf[1:2].view(()), 1.0)
        elif case == 'gen_forward':
            model.netG.run_forward(model.real_A, model.mask, model.CAM, model.slice_ratio, training=True)
        elif case == 'forward':
            model.forward()
        elif case.startswith('phase_a'):
            model._phase_a()
        elif case == 'phase_b':
            model._phase_b()
        elif case == 'phase_c':
            model._phase_c()
    if case in ('phase_b', 'phase_c'):
        model._phase_a()
    if case == 'phase_c':
        model._phase_b()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        body()
    print(case, 'captured', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(case, 'replayed', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
            tail = (r.stdout.strip().splitlines() or [''])[-1]
            err = [l for l in r.stderr.splitlines() if 'Error' in l or 'error' in l or 'Fatal' in l][-3:]
            print('%-16s rc=%d  %s  %s' % (c, r.returncode, tail, ' | '.join(err)), flush=True)
