"""Build libhvgan.so (gfx950) in-tree with hipcc; cross-compiles without a GPU.

    python healthivert-gan_amd/csrc/build.py [--force]

Objects are rebuilt only when their source (or a header) is newer.  The .so lands next to the
package (healthivert-gan_amd/libhvgan.so) so it travels with the tree to the GPU box.
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SOURCES = ['conv_igemm.hip', 'conv_halo.hip', 'conv_halo2.hip', 'conv_lf.hip', 'conv_g4.hip', 'conv_thin.hip', 'conv_px.hip', 'conv_s2t.hip', 'wgrad_halo.hip', 'wgrad_tr.hip', 'conv_narrow.hip', 'conv_head.hip', 'prep.hip', 'norm.hip', 'pointwise.hip', 'attention.hip', 'attention_gram.hip', 'bgemm.hip', 'rhlv.hip', 'assemble.hip', 'infer_prep.hip', 'eval_metrics.hip']
FLAGS = os.environ.get('HV_EXTRA_FLAGS', '').split() + ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-Wno-inline-asm', '-Rpass-analysis=kernel-resource-usage']
OUT = os.path.join(PKG, 'libhvgan.so')


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def _resources(remarks):
    """kernel (mangled name) -> {vgprs, agprs, scratch, occupancy, lds} from the compiler's kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r'remark: (?:\S+ )?Function Name: (\S+)', line) or re.search(r'Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (('vgprs', r' VGPRs: (\d+)'), ('agprs', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(HERE, 'hv_common.h'), os.path.join(HERE, 'conv_halo.h'), os.path.join(ROOT, 'include', 'hvgan.h')]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(objdir, s.replace('.hip', '.o'))
        if force or _newer(src, obj, deps):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        # registers / scratch / occupancy of every kernel, next to the object (tests/test_host_cpu.py: the hot kernels must not spill)
        with open(obj.replace('.o', '.resources.json'), 'w') as f:
            json.dump(_resources(r.stderr), f, indent=0, sort_keys=True)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    if jobs or force or not os.path.exists(OUT):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
