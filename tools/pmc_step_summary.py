"""Per kernel instantiation: mean HBM bytes per launch from the two rocprofv3 --pmc passes of tools/pmc_step.sh.

traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Kernel names are normalised to the form
hv_last_kernel_name() / bench.py's roofline use ("conv_halo2_kernel<8, 16, 128, 1, 4, 32, 1, 4, 4>").
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

def demangle(n):
    """_Z<len><name>I<template args>E...: the only argument kinds the kernels use are types (DF16_, f) and integral / bool literals"""
    m = re.match(r'_Z(\d+)', n)
    if not m:
        return n
    ln = int(m.group(1))
    name, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
    if not rest.startswith('I'):
        return name
    rest, args = rest[1:], []
    while rest and not rest.startswith('E'):
        for pat, fn in ((r'DF16_', lambda g: '_Float16'), (r'f', lambda g: 'float'), (r'Li(\d+)E', lambda g: g.group(1)),
                        (r'Lb([01])E', lambda g: 'true' if g.group(1) == '1' else 'false')):
            g = re.match(pat, rest)
            if g:
                args.append(fn(g))
                rest = rest[g.end():]
                break
        else:
            return n
    return '%s<%s>' % (name, ', '.join(args))


def norm(names):
    out = {}
    for n in names:
        d = demangle(n) if n.startswith('_Z') else n
        d = re.sub(r'^void ', '', d)
        depth = 0
        for i, ch in enumerate(d):        # cut the argument list: first '(' outside the template brackets
            if ch == '<':
                depth += 1
            elif ch == '>':
                depth -= 1
            elif ch == '(' and depth == 0:
                d = d[:i]
                break
        d = d.strip()
        # (round 4: hv_last_kernel_name() reports all five template arguments of conv_lf_kernel, as rocprofv3 prints them: no folding)
        out[n] = d
    return out


def main():
    root, dst = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    names = norm(list(acc))
    kernels = {}
    for raw, cs in acc.items():
        if 'FETCH_SIZE' not in cs or 'WRITE_SIZE' not in cs:
            continue
        name = names[raw]
        if name.startswith('at::') or 'elementwise' in name:
            continue
        f = sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE'])
        w = sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])
        kernels[name] = {'launches': len(cs['FETCH_SIZE']), 'FETCH_SIZE_KiB_raw': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1),
                         'traffic_bytes': int((2 * f + w) * 1024)}
    # steps of the profiled run = launches of a kernel that runs exactly once per train step
    once = [v['launches'] for k, v in kernels.items() if k.startswith('post_generator_kernel')]
    nsteps = once[0] if once else None
    doc = {'_how': 'tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a second run, --pmc WRITE_SIZE over '
                   '"bench.py --serial --no-graph --steps 2 --warmup 1"; mean per launch of every kernel instantiation over all the '
                   'layer shapes it serves in the step. traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction '
                   'from MI355X_MICROARCH.md, HBM section).',
           'steps_in_run': nsteps,
           'step_total_bytes': (int(sum(v['traffic_bytes'] * v['launches'] for v in kernels.values()) / nsteps) if nsteps else None),
           'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1]['traffic_bytes'] * kv[1]['launches']))}
    json.dump(doc, open(dst, 'w'), indent=1)
    for k, v in list(doc['kernels'].items())[:25]:
        print('%-70s n=%5d  %8.1f MB/launch' % (k[:70], v['launches'], v['traffic_bytes'] / 1e6))


if __name__ == '__main__':
    main()
