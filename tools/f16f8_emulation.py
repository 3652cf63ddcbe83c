"""CPU emulation of the MLP's GEMM arithmetic modes on rays of the benchmark frame -- what the raw head looks like in f16x3, f16x2
and the experimental f16f8 (fp16 main product + e4m3 cross terms, csrc/mlp_f16f8_kernel.hip) against the exact chain:
python tools/f16f8_emulation.py [model] [n_rays].  Products are formed exactly (float64), so only the OPERAND roundings of each
mode are emulated, not the fp32 accumulation order.  No GPU needed; tools/head_error.py is the measured counterpart."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from hyperreel_amd import config as C, scenes
from hyperreel_oracle import HyperReelOracle

F32, F64 = np.float32, np.float64


def _e4m3_table():
    v = []
    for b in range(127):                               # 0x7f is NaN
        e, m = b >> 3, b & 7
        v.append(m / 8 * 2.0 ** -6 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))
    return np.array(v, F64)


E4M3 = _e4m3_table()
MID = (E4M3[1:] + E4M3[:-1]) / 2


def e4m3(x):
    """round to nearest even onto OCP e4m3fn, saturating at 448 (what v_cvt_pk_fp8_f32 of a clamped operand and the host packer do)"""
    a = np.minimum(np.abs(x.astype(F64)), 448.0)
    i = np.searchsorted(MID, a, side='left')           # first midpoint >= a -> candidate index i (value below the midpoint)
    tie = (i < MID.size) & (a == MID[np.minimum(i, MID.size - 1)])
    i = np.where(tie & (i % 2 == 1), i + 1, i)         # on a tie take the even mantissa
    return np.sign(x) * E4M3[np.minimum(i, 126)]


def f16(x):
    return x.astype(np.float16).astype(F64)


def layer_scale(w):
    mx = np.abs(w).max()
    _, e = np.frexp(F32(mx))
    return 2.0 ** min(max(14 - int(e), -14), 40)


def run(o, x0, mode):
    x, inp = x0.astype(F64), x0.astype(F64)
    for i, (w, b) in enumerate(o.layers):
        skip = i in o.skips
        if mode == 'exact':
            xin = np.concatenate([inp, x], -1) if skip else x
            y = xin @ w.astype(F64).T + b
        else:
            s = layer_scale(w)
            v = w.astype(F64) * s
            wh = f16(v); wl = v - wh
            def three(xs, vh, vl):                      # x_hi w_hi + x_hi w_lo + x_lo w_hi with half halves
                xh = f16(xs); xl = f16(xs - xh)
                return xh @ vh.T + xh @ f16(vl).T + xl @ vh.T
            def hidden(xs, vh, vl):
                xh = f16(xs); xlo = xs - xh
                if mode == 'f16x3': return xh @ vh.T + xh @ f16(vl).T + f16(xlo) @ vh.T
                if mode == 'f16x2': return xh @ vh.T + f16(xlo) @ vh.T
                if mode.startswith('f16f8'):
                    # f16f8:<a>:<b>:<c>:<d> = log2 scales of x (for x_hi w_lo), w_lo, x_lo, w_hi; the product scales are undone exactly
                    a, b_, c, d = ([float(t) for t in mode.split(':')[1:]] + [0, 0, 12, -12])[:4] if ':' in mode else (0, 0, 12, -12)
                    return (xh @ vh.T + (e4m3(xs * 2.0 ** a) @ e4m3(vl * 2.0 ** b_).T) * 2.0 ** -(a + b_)
                            + (e4m3(xlo * 2.0 ** c) @ e4m3(vh * 2.0 ** d).T) * 2.0 ** -(c + d))
                raise ValueError(mode)
            k_in = inp.shape[1]
            if i == 0:
                acc = (three if mode != 'f16x2' else (lambda xs, vh, vl: f16(xs) @ vh.T + f16(xs - f16(xs)) @ vh.T))(x, wh, wl)
            elif skip:
                first = three if mode != 'f16x2' else (lambda xs, vh, vl: f16(xs) @ vh.T + f16(xs - f16(xs)) @ vh.T)
                acc = first(inp, wh[:, :k_in], wl[:, :k_in]) + hidden(x, wh[:, k_in:], wl[:, k_in:])
            else:
                acc = hidden(x, wh, wl)
            y = acc / s + b
        y = y.astype(F32).astype(F64)                   # the epilogue's fp32 value
        if i < o.D + 1:
            y = np.where(y >= 0, y, y * F64(F32(0.01)))
            y = y.astype(F32).astype(F64)
        x = y
    return x


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'donerf_sphere'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    sd = scenes.make_state_dict(cfg, ds, [16, 16, 16], seed=7, density='dense', app_scale=1.0)
    rays = scenes.benchmark_rays(name, 800, 800, frame=7)
    r = np.ascontiguousarray(rays[np.random.default_rng(0).choice(rays.shape[0], n, replace=False)])
    o = HyperReelOracle(cfg, ds, sd)
    x0 = o._param_pe(r)
    ref = run(o, x0, 'exact')
    scale = np.abs(ref).max()
    print(f'{name}: {n} rays, MLP input {x0.shape[1]} columns, head {ref.shape[1]} columns, max|head| = {scale:.3f}, max|hidden input| = {np.abs(x0).max():.2f}')
    for mode in ['f16x3', 'f16f8', 'f16x2'] + sys.argv[3:]:
        d = np.abs(run(o, x0, mode) - ref)
        print(f'  {mode}: max |d head| / max|head| = {d.max() / scale:.3e}   rms = {np.sqrt((d ** 2).mean()) / scale:.3e}')


if __name__ == '__main__':
    main()
