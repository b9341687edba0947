"""CPU simulation (float64 reference) of the reduced-cost splits of the dense correlation volume.

    x  ->  h = f16(256 x),  l = 256 (256 x - h)          (exact in f32)
    2^16 <k, q> = sum h h  +  2^-8 (sum h~ l~ + sum l~ h~)      (l l dropped)

where ~ is the narrow format of the cross terms:
    f16f8 : e4m3, uniform scale                                  (fgvc_corr_volume_f16f8, 1024 pipe cycles per 32x32 tile)
    f16f6 : e2m3 with an E8M0 scale per 32 consecutive channels   (fgvc_corr_volume_f16f6,  768 pipe cycles per tile)
    f16f4 : e2m1 with an E8M0 scale per 32 channels               (not built: shown to fail)
Prints the max / rms logit error (temperature 0.07) over sampled (k, q) pairs for several row families.
Run on the CPU: python tools/sim_split_formats.py
"""
import numpy as np

TAU = 0.07


def q_e4m3(x):
    """round to nearest e4m3fn (OCP): 3 mantissa bits, exponents 2^-6 .. 2^8, max 448, subnormal step 2^-9"""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    r = np.round(a / step) * step          # numpy rounds half to even
    r = np.minimum(r, 448.0)
    return np.sign(x) * r


def q_block(x, mant_bits, emax_val, e_lo):
    """block-scaled minifloat: per 32 consecutive elements of the last axis an E8M0 scale 2^s with max|x| / 2^s <= emax_val;
    element format: exponent range [e_lo, log2(emax)] with `mant_bits` mantissa bits and subnormals below 2^e_lo."""
    x = np.asarray(x, np.float64)
    shp = x.shape
    b = x.reshape(shp[:-1] + (shp[-1] // 32, 32))
    m = np.abs(b).max(-1, keepdims=True)
    s = np.ceil(np.log2(np.maximum(m, 1e-300) / emax_val))
    y = b / 2.0 ** s
    a = np.abs(y)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, e_lo)
    step = 2.0 ** (e - mant_bits)
    r = np.minimum(np.round(a / step) * step, emax_val)
    return (np.sign(y) * r * 2.0 ** s).reshape(shp)


def q_e2m3(x):
    return q_block(x, 3, 7.5, 0)


def q_e2m1(x):
    return q_block(x, 1, 6.0, 0)


def split(x):
    xs = (x.astype(np.float32) * np.float32(256.0))
    h = xs.astype(np.float16).astype(np.float32)
    l = (xs - h) * np.float32(256.0)
    return h.astype(np.float64), l.astype(np.float64)


def families(rng, n, C):
    def nrm(a):
        return a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
    out = {}
    out["gauss"] = nrm(rng.standard_normal((n, C)))
    a = rng.standard_normal((n, C)) * (rng.random((n, C)) < 0.05)
    a[:, 0] += 1e-3
    out["sparse5%"] = nrm(a)
    a = rng.standard_normal((n, C)) * 0.02
    a[np.arange(n), rng.integers(0, C, n)] = 1.0
    out["one-hot+noise"] = nrm(a)
    out["heavy-tail"] = nrm(rng.standard_t(1.5, (n, C)))
    a = np.abs(rng.standard_normal((n, C)))            # ReLU-like, all positive: errors cannot cancel by sign of x
    out["positive"] = nrm(a)
    a = np.exp(rng.standard_normal((n, C)) * 3.0) * np.sign(rng.standard_normal((n, C)))
    out["lognormal"] = nrm(a)
    return out


def main():
    rng = np.random.default_rng(0)
    n, C = 3000, 256
    print(f"{'family':16s} {'variant':8s} {'max err':>10s} {'rms err':>10s}   (logit units, tau = {TAU})")
    for name, f in families(rng, 2 * n, C).items():
        f = f.astype(np.float32)
        k, q = f[:n], f[n:]
        ref = (k.astype(np.float64) @ q.astype(np.float64).T) / TAU
        hk, lk = split(k)
        hq, lq = split(q)
        main_sum = hk @ hq.T
        for var, qf in (("f16 only", None), ("f16f8", q_e4m3), ("f16f6", q_e2m3), ("f16f4", q_e2m1)):
            if qf is None:
                tot = main_sum
            else:
                tot = main_sum + (qf(hk) @ qf(lq).T + qf(lk) @ qf(hq).T) / 256.0
            got = tot / 65536.0 / TAU
            err = np.abs(got - ref)
            print(f"{name:16s} {var:8s} {err.max():10.2e} {np.sqrt((err ** 2).mean()):10.2e}")


if __name__ == "__main__":
    main()
