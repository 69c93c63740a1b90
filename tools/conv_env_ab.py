"""Two settings of ONE environment switch of the 4x4 convolution kernels against each other: bit identity and time per shape.
    python tools/conv_env_ab.py HV_G4S1_SPREAD [s1|s2]       (runs itself once more with the switch at 0 for the reference bits and times)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

S1 = [  # B, H, W, Cin, Cout, transposed, stride
    (2, 11, 13, 128, 128, 0, 1), (3, 20, 17, 32, 256, 0, 1), (2, 11, 13, 128, 128, 1, 1), (3, 19, 16, 256, 128, 1, 1),
    (16, 32, 32, 256, 512, 0, 1), (32, 32, 32, 256, 512, 0, 1), (16, 31, 31, 512, 256, 1, 1), (32, 31, 31, 512, 256, 1, 1), (16, 64, 64, 256, 512, 0, 1),
]
S2 = [
    (2, 20, 24, 32, 128, 0, 2), (3, 36, 28, 64, 256, 0, 2), (2, 10, 12, 128, 64, 1, 2),
    (16, 128, 128, 64, 128, 0, 2), (32, 128, 128, 64, 128, 0, 2), (16, 64, 64, 128, 256, 0, 2), (32, 64, 64, 128, 256, 0, 2),
    (16, 64, 64, 128, 64, 1, 2), (32, 64, 64, 128, 64, 1, 2), (16, 32, 32, 256, 128, 1, 2), (32, 32, 32, 256, 128, 1, 2),
]


def run(tag, shapes):
    import torch
    import hvgan
    from hvgan import ops, lib
    dev = torch.device('cuda:0')
    out = {}
    for (B, H, W, Cin, Cout, tr, st) in shapes:
        g = torch.Generator().manual_seed(B + H + Cin + tr)
        x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
        w = (torch.randn(Cout, 16, Cin, generator=g) / (Cin * 16) ** 0.5).to(dev)
        wh = w.half(); wt = ops.tile_weights(wh, Cout, 16, Cin)
        bias = None if tr else (torch.randn(Cout, generator=g) * 0.1).to(dev)
        if st == 1:
            Ho, Wo = (H + 1, W + 1) if tr else (H - 1, W - 1)
        else:
            Ho, Wo = (2 * H, 2 * W) if tr else (H // 2, W // 2)
        rot = 4
        ys = [ops.Act(torch.zeros(B, Ho, Wo, Cout, device=dev, dtype=torch.float16)) for _ in range(rot)]
        xs = [x] + [ops.Act(x.t.clone()) for _ in range(rot - 1)]
        f = lambda i: ops.conv2d(xs[i % rot], w, ys[i % rot], 4, st, 1, 1, transposed=bool(tr), precision='fp16', w_h=wh, w_t=wt, act='none' if tr else 'lrelu', bias=bias)
        f(0)
        path = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        key = '%d_%d_%d_%d_%d_%d_%d' % (B, H, W, Cin, Cout, tr, st)
        out[key] = ys[0].t.cpu().clone()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(20):
                f(i)
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        fl = 2.0 * B * (H * W if tr and st == 2 else Ho * Wo) * Cout * 16 * Cin * (4 if tr and st == 2 else 1) / (4 if tr and st == 2 else 1)
        out[key + '_us'] = best
        print('%s %s B%-2d %3dx%-3d %3d->%3d s%d path %d  %.1f us  %.0f TF' % (tag, 'T' if tr else 'F', B, H, W, Cin, Cout, st, path, best, fl / best / 1e6), flush=True)
    return out


if __name__ == '__main__':
    import torch
    var = sys.argv[1]
    shapes = S2 if len(sys.argv) > 2 and sys.argv[2] == 's2' else S1
    if os.environ.get('CONV_AB_CHILD'):
        torch.save(run('%s=0' % var, shapes), os.environ['CONV_AB_CHILD'])
        sys.exit(0)
    ref_path = '/tmp/conv_ab_ref.pt'
    env = dict(os.environ, CONV_AB_CHILD=ref_path)
    env[var] = '0'
    subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=True)      # (before this process touches the GPU)
    os.environ[var] = '1'
    new = run('%s=1' % var, shapes)
    ref = torch.load(ref_path)
    bad = [k for k in new if not k.endswith('_us') and not torch.equal(new[k], ref[k])]
    print('bit-identical' if not bad else 'DIFFERENT: %s' % bad)
    for k in new:
        if k.endswith('_us'):
            print('%-30s 0: %.1f us -> 1: %.1f us  (%.2fx)' % (k[:-3], ref[k], new[k], ref[k] / new[k]))
    sys.exit(1 if bad else 0)
