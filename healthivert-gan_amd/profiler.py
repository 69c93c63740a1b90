"""Live per-kernel timing of the MFMA kernels with HIP events on the launch stream (bench.py's `roofline`).

Two phases so that the timed region is not perturbed: `survey()` times every convolution launch of one
untimed step and picks the dominant kernel instantiation (largest total time over all the layer shapes it serves -- the
granularity of a rocprofv3 stats row); `enable(only=...)` then brackets only its launches with events.  FLOPs are algorithmic: 2*pixels*Cout*taps*Cin (valid taps only
for the transposed/data-gradient form).
"""
import ctypes

import torch

from . import lib as _lib
from . import ops


class KernelTimer:
    def __init__(self):
        self.records = []
        self.paths = {}
        self.bytes = {}       # shape key -> algorithmic bytes per launch
        self.names = {}       # shape key -> kernel instantiation name as rocprofv3 prints it (hv_last_kernel_name)
        self.only = None      # set of shape keys to time, or None = all
        self.active = False
        self.pair_us = 0.0    # calibrate(): an empty event pair's own reading, subtracted from every launch

    def calibrate(self, n=200):
        """Reading of an event pair recorded around NOTHING on the current stream (microseconds; the 5 % quantile of n pairs -- a lower bound of the
        events' own share, so the net durations stay on the conservative side): the part of every kernel's event duration that is the two barrier
        packets' own, which rocprofv3's kernel durations do not contain (measured: 3.7 us on a 13.9 us kernel, profiles/r03l)."""
        torch.cuda.synchronize()
        pairs = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        v = sorted(s.elapsed_time(e) * 1e3 for s, e in pairs)
        self.pair_us = v[len(v) // 20]
        return self.pair_us

    # ---- hook called by ops.conv2d / ops.conv2d_wgrad around each launch
    def wrap(self, key, flops, launch, nbytes=0):
        self.bytes[key] = nbytes                            # algorithmic HBM bytes of one launch (each operand once)
        if not self.active or (self.only is not None and key not in self.only):
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L = _lib.get().cdll
        # the C side records the two events right around the kernel launch (for a weight gradient: around the main kernel only, not its slab
        # reduction), so the duration is that of the kernel rocprofv3 lists under the same name and no host latency sits between the records
        s.record(); e.record()          # create the underlying hipEvents
        L.hv_set_kernel_timing(ctypes.c_void_p(s.cuda_event), ctypes.c_void_p(e.cuda_event))
        r = launch()
        self.paths[key] = L.hv_last_kernel_path()          # which kernel family the C side dispatched to
        self.names[key] = (L.hv_last_kernel_name() or b'').decode()
        self.records.append((key, flops, s, e))
        return r

    def enable(self, only=None):
        self.records, self.only, self.active = [], (set(only) if only else None), True
        ops.set_timer(self)

    def disable(self):
        self.active = False
        ops.set_timer(None)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, s, e in self.records:
            a = agg.setdefault(key, [0.0, 0, flops])
            a[0] += max(s.elapsed_time(e) - self.pair_us * 1e-3, 1e-4)      # net of the event pair's own reading
            a[1] += 1
        return agg   # shape key -> [total ms (net), launches, flops per launch]

    def by_kernel(self):
        """kernel instantiation name -> [total ms, launches, total flops, [shape keys]] (the granularity of a rocprofv3 stats row)."""
        out = {}
        for key, (ms, n, flops) in self.summary().items():
            a = out.setdefault(self.names.get(key) or describe(key, self.paths.get(key)), [0.0, 0, 0.0, []])
            a[0] += ms; a[1] += n; a[2] += flops * n; a[3].append(key)
        return out

    def dominant(self):
        """(kernel name, [shape keys]) of the instantiation with the largest total time."""
        bk = self.by_kernel()
        if not bk:
            return None
        name, a = max(bk.items(), key=lambda kv: kv[1][0])
        return name, a[3]

    def dominant_mfma(self, peak_tflops):
        """(kernel name, [shape keys]) of the MFMA-bound instantiation (arithmetic intensity above the ridge) with the largest total time."""
        ridge = peak_tflops * 1e12 / (HBM_PEAK_GBS * 1e9)
        best = None
        for name, (ms, n, flops, keys) in self.by_kernel().items():
            nbytes = sum(self.bytes.get(k, 0) * self.summary()[k][1] for k in keys)
            if nbytes and flops / nbytes >= ridge and (best is None or ms > best[0]):
                best = (ms, name, keys)
        return (best[1], best[2]) if best else None

    def roofline(self, precision, peak_tflops, total_ms=None, name=None):
        bk = self.by_kernel()
        if not bk or (name is not None and name not in bk):
            return None
        if name is None:
            name = max(bk.items(), key=lambda kv: kv[1][0])[0]
        ms, n, flops, keys = bk[name]
        agg = self.summary()
        achieved = flops / (ms * 1e-3) / 1e12
        ridge = peak_tflops * 1e12 / (HBM_PEAK_GBS * 1e9)

        def shape_row(k):
            avg_s = agg[k][0] / agg[k][1] * 1e-3
            tf, gbs = agg[k][2] / avg_s / 1e12, self.bytes.get(k, 0) / avg_s / 1e9
            hbm = bool(self.bytes.get(k)) and agg[k][2] / self.bytes[k] < ridge        # this shape's own bound
            return {'shape': describe(k, self.paths.get(k), self.names.get(k)), 'launches': agg[k][1], 'avg_us': round(avg_s * 1e6, 2), 'tflops': round(tf, 1),
                    'algorithmic_gbs': round(gbs, 1), 'bound': 'hbm' if hbm else 'mfma',
                    'frac': round(gbs / HBM_PEAK_GBS if hbm else tf / peak_tflops, 4)}
        shapes = [shape_row(k) for k in sorted(keys, key=lambda k: -agg[k][0])]
        nbytes = sum(self.bytes.get(k, 0) * agg[k][1] for k in keys)          # algorithmic bytes over all launches of the instantiation
        out = {'kernel': name, 'launches': n, 'avg_us': round(ms / n * 1e3, 2), 'avg_us_events': round(ms / n * 1e3 + self.pair_us, 2),
               'event_pair_overhead_us': round(self.pair_us, 2), 'gflop_per_launch': round(flops / n / 1e9, 3),
               'algorithmic_mb_per_launch': round(nbytes / n / 1e6, 2), 'shapes': shapes, 'traffic': None}
        # which roofline bounds it: arithmetic intensity against the ridge of the two peaks
        if nbytes and flops / nbytes < ridge:
            gbs = nbytes / (ms * 1e-3) / 1e9
            out.update(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                       achieved_tflops=round(achieved, 2), flop_per_byte=round(flops / nbytes, 1), ridge_flop_per_byte=round(ridge, 1))
        else:
            out.update(bound='mfma', achieved=round(achieved, 2), peak=peak_tflops, unit='TFLOP/s', frac=round(achieved / peak_tflops, 4),
                       flop_per_byte=round(flops / nbytes, 1) if nbytes else None, ridge_flop_per_byte=round(ridge, 1))
        return out


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s

# hv_last_kernel_path codes (the `hv_path_note = N` of each launcher in csrc/): the kernel FAMILY; the exact instantiation comes from hv_last_kernel_name
KERNEL_NAMES = {0: 'conv_igemm_kernel', 1: 'narrow_fwd_kernel', 2: 'conv_halo_kernel', 3: 'conv_halo2_kernel', 4: 'thin1_fwd_kernel', 5: 'head_gemm_kernel', 6: 'conv_s2t_kernel',
                7: 'conv_lf_kernel', 8: 'conv_g4_kernel', 9: 'thin_dgrad_kernel', 10: 'wgrad_kernel', 11: 'wgrad_halo_kernel', 12: 'wgrad_tr_kernel', 13: 'wgrad_trd_kernel', 14: 'conv_px_kernel'}


def describe(key, path=None, kname=None):
    """One shape row's label: the kernel that RAN for it -- the instantiation name the launcher reported (as rocprofv3 prints it) when there is one, else
    the family of its path code -- and the layer's shape."""
    kind, B, H, W, cin, cout, k, s, d, tr = key
    name = kname or KERNEL_NAMES.get(path, {'conv': 'conv_igemm_kernel', 'wgrad': 'wgrad_kernel'}[kind])
    if name.startswith('void '):
        name = name[5:]
    name = name.split('(')[0]
    return '%s %s B%d %dx%d Cin%d->Cout%d k%d s%d d%d' % (name, 'transposed' if tr else 'forward', B, H, W, cin, cout, k, s, d)
