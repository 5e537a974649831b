"""
Host restatements (plain Python floats: IEEE f64, no FMA) of the arithmetic the device render kernels use for stochastic paths, so
that a path evaluated on the host (source_amd/optical/hybrid.py, Ray.trace of a single ray) takes the same turns as the same path
on the device: Philox4x32-10 with librsx's counter convention (include/rsx.h), the portable sin / cos / asin of
source_amd/csrc/dev_render.hpp (Cody-Waite reduction + minimax polynomials; oracle/rsx_oracle.c holds the C form).
"""
import math
import struct

import numpy as np

M32 = 0xFFFFFFFF
TWO_PI = 2.0 * math.pi


def philox2(seed, pixel, sample):
    """Two uniforms in [0, 1) from counter (pixel, sample) under key seed — dev_render.hpp philox2."""
    c0, c1, c2, c3 = pixel & M32, (pixel >> 32) & M32, sample & M32, (sample >> 32) & M32
    k0, k1 = seed & M32, (seed >> 32) & M32
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    a, b = (c1 << 32) | c0, (c3 << 32) | c2
    return (a >> 11) * (1.0 / 9007199254740992.0), (b >> 11) * (1.0 / 9007199254740992.0)


def philox2_array(seed, pixel, sample):
    """philox2 over numpy uint64 arrays `pixel`, `sample` (primary-ray jitter of a whole block of pixels)."""
    pixel, sample = np.asarray(pixel, dtype=np.uint64), np.asarray(sample, dtype=np.uint64)
    m = np.uint64(M32)
    s32 = np.uint64(32)
    c0, c1, c2, c3 = pixel & m, (pixel >> s32) & m, sample & m, (sample >> s32) & m
    k0, k1 = int(seed) & M32, (int(seed) >> 32) & M32
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ np.uint64(k0)) & m, p1 & m, ((p0 >> s32) ^ c3 ^ np.uint64(k1)) & m, p0 & m
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    a, b = (c1 << s32) | c0, (c3 << s32) | c2
    scale = 1.0 / 9007199254740992.0
    return (a >> np.uint64(11)).astype(np.float64) * scale, (b >> np.uint64(11)).astype(np.float64) * scale


def sincos(phi):
    """(sin, cos) of phi in [0, 2 pi] — portable_sincos."""
    PIO2_HI, PIO2_LO, TWO_OVER_PI = 1.57079632673412561417e+00, 6.07710050650619224932e-11, 6.36619772367581382433e-01
    S1, S2, S3, S4, S5, S6 = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                              2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10)
    C1, C2, C3, C4, C5, C6 = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                              -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11)
    kf = math.floor(phi * TWO_OVER_PI + 0.5)
    r = (phi - kf * PIO2_HI) - kf * PIO2_LO
    z = r * r
    ps = r + (r * z) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))))
    pc = (1.0 - 0.5 * z) + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))
    q = int(kf) & 3
    if q == 0:
        return ps, pc
    if q == 1:
        return pc, -ps
    if q == 2:
        return -ps, -pc
    return -pc, ps


def asin(x):
    """asin on [0, 1] — portable_asin."""
    PIO2_HI, PIO2_LO, PIO4_HI = 1.57079632679489655800e+00, 6.12323399573676603587e-17, 7.85398163397448278999e-01
    P0, P1, P2, P3, P4, P5 = (1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01,
                              -4.00555345006794114027e-02, 7.91534994289814532176e-04, 3.47933107596021167570e-05)
    Q1, Q2, Q3, Q4 = -2.40339491173441421878e+00, 2.02094576023350569471e+00, -6.88283971605453293030e-01, 7.70381505559019352791e-02
    if x >= 1.0:
        return x * PIO2_HI + x * PIO2_LO
    if x < 0.5:
        if x < 7.450580596923828e-09:
            return x
        t = x * x
        p = t * (P0 + t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5)))))
        q = 1.0 + t * (Q1 + t * (Q2 + t * (Q3 + t * Q4)))
        return x + x * (p / q)
    w = 1.0 - x
    t = w * 0.5
    p = t * (P0 + t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5)))))
    q = 1.0 + t * (Q1 + t * (Q2 + t * (Q3 + t * Q4)))
    s = math.sqrt(t)
    if x >= 0.975:
        w = p / q
        return PIO2_HI - (2.0 * (s + s * w) - PIO2_LO)
    w = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", s))[0] & 0xFFFFFFFF00000000))[0]
    c = (t - w * w) / (s + w)
    r = p / q
    p = 2.0 * s * r - (PIO2_LO - 2.0 * c)
    q = PIO4_HI - 2.0 * w
    return PIO4_HI - (p - q)


def _two_sum(a, b):
    t = a + b
    bb = t - a
    return t, (a - (t - bb)) + (b - bb)


def _split(a):
    c = 134217729.0 * a
    h = c - (c - a)
    return h, a - h


def _two_prod(a, b):
    ah, al = _split(a)
    bh, bl = _split(b)
    p = a * b
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _from_bits(b):
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def pow(x, y):                                              # noqa: A001 (mirrors the C name)
    """x ** y for x > 0 — portable_pow of oracle/rsx_oracle.c and dev_render.hpp, operation for operation."""
    if not (x > 0.0) or not (x < math.inf) or y != y or not (abs(y) < math.inf):
        return math.pow(x, y)
    if y == 0.0 or x == 1.0:
        return 1.0
    bits = _bits(x)
    e = (bits >> 52) & 0x7FF
    if e == 0:
        bits = _bits(x * 18014398509481984.0)
        e = ((bits >> 52) & 0x7FF) - 54
    e -= 1023
    m = _from_bits((bits & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000)
    if m > 1.4142135623730951:
        m *= 0.5
        e += 1
    num = m - 1.0
    den_h, den_l = _two_sum(m, 1.0)
    s_h = num / den_h
    ph, pl = _two_prod(s_h, den_h)
    s_l = (((num - ph) - pl) - s_h * den_l) / den_h
    z_h, z_l = _two_prod(s_h, s_h)
    z_l += 2.0 * s_h * s_l
    c_h, c_l = _two_prod(s_h, z_h)
    c_l += s_h * z_l + s_l * z_h
    THIRD_H, THIRD_L = 3.33333333333333314830e-01, 1.85037170770859413132e-17
    q_h, q_l = _two_prod(c_h, THIRD_H)
    q_l += c_h * THIRD_L + c_l * THIRD_H
    z = z_h
    poly = 1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 + z * (1.0 / 13.0 + z * (1.0 / 15.0 + z * (1.0 / 17.0
           + z * (1.0 / 19.0 + z * (1.0 / 21.0 + z * (1.0 / 23.0)))))))))
    tail = 2.0 * ((c_h * z) * poly) + 2.0 * (q_l + s_l)
    a_h, a_l = _two_sum(2.0 * s_h, 2.0 * q_h)
    l_h, l_l = _two_sum(a_h, a_l + tail)
    LN2_H, LN2_L = 6.93147180559945286227e-01, 2.31904681384629955842e-17
    eh, el = _two_prod(float(e), LN2_H)
    el += float(e) * LN2_L
    t_h, t_l = _two_sum(eh, l_h)
    t_l += el + l_l
    L_h, L_l = _two_sum(t_h, t_l)
    p_h, p_l = _two_prod(y, L_h)
    p_l += y * L_l
    if p_h > 709.8:
        return math.inf
    if p_h < -745.2:
        return 0.0
    LN2_CW_H, LN2_CW_L = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    kf = math.floor(p_h * 1.44269504088896338700e+00 + 0.5)
    r = ((p_h - kf * LN2_CW_H) - kf * LN2_CW_L) + p_l
    ex = 1.0 + r * (1.0 + r * (1.0 / 2.0 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0 + r * (1.0 / 720.0 + r * (1.0 / 5040.0
         + r * (1.0 / 40320.0 + r * (1.0 / 362880.0 + r * (1.0 / 3628800.0 + r * (1.0 / 39916800.0 + r * (1.0 / 479001600.0
         + r * (1.0 / 6227020800.0 + r * (1.0 / 87178291200.0))))))))))))))
    k = int(kf)
    k1 = int(k / 2)                                         # C integer division truncates towards zero
    k2 = k - k1
    return (ex * _from_bits((k1 + 1023) << 52)) * _from_bits((k2 + 1023) << 52)


# ---- the same functions over numpy arrays (hybrid.py evaluates whole waves of path nodes at once): element by element the operations
# of the scalar forms above in the same order, so the same bits (numpy's f64 ufuncs are IEEE + - * / sqrt, one rounding each) ----------

def sincos_array(phi):
    """sincos over an array: (sin, cos)."""
    phi = np.asarray(phi, dtype=np.float64)
    PIO2_HI, PIO2_LO, TWO_OVER_PI = 1.57079632673412561417e+00, 6.07710050650619224932e-11, 6.36619772367581382433e-01
    S1, S2, S3, S4, S5, S6 = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                              2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10)
    C1, C2, C3, C4, C5, C6 = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                              -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11)
    kf = np.floor(phi * TWO_OVER_PI + 0.5)
    r = (phi - kf * PIO2_HI) - kf * PIO2_LO
    z = r * r
    ps = r + (r * z) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))))
    pc = (1.0 - 0.5 * z) + (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))
    q = kf.astype(np.int64) & 3
    sn = np.where(q == 0, ps, np.where(q == 1, pc, np.where(q == 2, -ps, -pc)))
    cs = np.where(q == 0, pc, np.where(q == 1, -ps, np.where(q == 2, -pc, ps)))
    return sn, cs


def asin_array(x):
    """asin over an array of values in [0, 1]."""
    x = np.asarray(x, dtype=np.float64)
    PIO2_HI, PIO2_LO, PIO4_HI = 1.57079632679489655800e+00, 6.12323399573676603587e-17, 7.85398163397448278999e-01
    P0, P1, P2, P3, P4, P5 = (1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01,
                              -4.00555345006794114027e-02, 7.91534994289814532176e-04, 3.47933107596021167570e-05)
    Q1, Q2, Q3, Q4 = -2.40339491173441421878e+00, 2.02094576023350569471e+00, -6.88283971605453293030e-01, 7.70381505559019352791e-02
    with np.errstate(all="ignore"):
        small = x < 0.5
        w = 1.0 - x
        t = np.where(small, x * x, w * 0.5)
        p = t * (P0 + t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5)))))
        q = 1.0 + t * (Q1 + t * (Q2 + t * (Q3 + t * Q4)))
        lo = np.where(x < 7.450580596923828e-09, x, x + x * (p / q))
        s = np.sqrt(t)
        wq = p / q
        near_one = PIO2_HI - (2.0 * (s + s * wq) - PIO2_LO)
        wh = (s.view(np.uint64) & np.uint64(0xFFFFFFFF00000000)).view(np.float64)
        c = (t - wh * wh) / (s + wh)
        pp = 2.0 * s * wq - (PIO2_LO - 2.0 * c)
        qq = PIO4_HI - 2.0 * wh
        mid = PIO4_HI - (pp - qq)
        out = np.where(small, lo, np.where(x >= 0.975, near_one, mid))
        return np.where(x >= 1.0, x * PIO2_HI + x * PIO2_LO, out)


def _two_sum_a(a, b):
    t = a + b
    bb = t - a
    return t, (a - (t - bb)) + (b - bb)


def _split_a(a):
    c = 134217729.0 * a
    h = c - (c - a)
    return h, a - h


def _two_prod_a(a, b):
    ah, al = _split_a(a)
    bh, bl = _split_a(b)
    p = a * b
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def pow_array(x, y):
    """pow over broadcastable arrays x (> 0, finite) and y (finite); elements outside that domain go through the scalar form."""
    x, y = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    with np.errstate(all="ignore"):
        odd = ~(x > 0.0) | ~(x < np.inf) | (y != y) | ~(np.abs(y) < np.inf)
        bits = x.view(np.uint64)
        e = ((bits >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64)
        den = e == 0
        if den.any():
            bits2 = (x * 18014398509481984.0).view(np.uint64)
            bits = np.where(den, bits2, bits)
            e = np.where(den, ((bits2 >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64) - 54, e)
        e = e - 1023
        m = ((bits & np.uint64(0x000FFFFFFFFFFFFF)) | np.uint64(0x3FF0000000000000)).view(np.float64)
        big = m > 1.4142135623730951
        m = np.where(big, m * 0.5, m)
        e = np.where(big, e + 1, e)
        num = m - 1.0
        den_h, den_l = _two_sum_a(m, np.float64(1.0))
        s_h = num / den_h
        ph, pl = _two_prod_a(s_h, den_h)
        s_l = (((num - ph) - pl) - s_h * den_l) / den_h
        z_h, z_l = _two_prod_a(s_h, s_h)
        z_l = z_l + 2.0 * s_h * s_l
        c_h, c_l = _two_prod_a(s_h, z_h)
        c_l = c_l + (s_h * z_l + s_l * z_h)
        THIRD_H, THIRD_L = 3.33333333333333314830e-01, 1.85037170770859413132e-17
        q_h, q_l = _two_prod_a(c_h, np.float64(THIRD_H))
        q_l = q_l + (c_h * THIRD_L + c_l * THIRD_H)
        z = z_h
        poly = 1.0 / 5.0 + z * (1.0 / 7.0 + z * (1.0 / 9.0 + z * (1.0 / 11.0 + z * (1.0 / 13.0 + z * (1.0 / 15.0 + z * (1.0 / 17.0
               + z * (1.0 / 19.0 + z * (1.0 / 21.0 + z * (1.0 / 23.0)))))))))
        tail = 2.0 * ((c_h * z) * poly) + 2.0 * (q_l + s_l)
        a_h, a_l = _two_sum_a(2.0 * s_h, 2.0 * q_h)
        l_h, l_l = _two_sum_a(a_h, a_l + tail)
        LN2_H, LN2_L = 6.93147180559945286227e-01, 2.31904681384629955842e-17
        ef = e.astype(np.float64)
        eh, el = _two_prod_a(ef, np.float64(LN2_H))
        el = el + ef * LN2_L
        t_h, t_l = _two_sum_a(eh, l_h)
        t_l = t_l + (el + l_l)
        L_h, L_l = _two_sum_a(t_h, t_l)
        p_h, p_l = _two_prod_a(y, L_h)
        p_l = p_l + y * L_l
        LN2_CW_H, LN2_CW_L = 6.93147180369123816490e-01, 1.90821492927058770002e-10
        kf = np.floor(p_h * 1.44269504088896338700e+00 + 0.5)
        r = ((p_h - kf * LN2_CW_H) - kf * LN2_CW_L) + p_l
        ex = 1.0 + r * (1.0 + r * (1.0 / 2.0 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0 + r * (1.0 / 720.0 + r * (1.0 / 5040.0
             + r * (1.0 / 40320.0 + r * (1.0 / 362880.0 + r * (1.0 / 3628800.0 + r * (1.0 / 39916800.0 + r * (1.0 / 479001600.0
             + r * (1.0 / 6227020800.0 + r * (1.0 / 87178291200.0))))))))))))))
        over, under = p_h > 709.8, p_h < -745.2
        kc = np.where(over | under | odd, 0.0, kf)
        k = kc.astype(np.int64)
        k1 = np.trunc(k / 2).astype(np.int64)
        k2 = k - k1
        f1 = ((k1 + 1023).astype(np.uint64) << np.uint64(52)).view(np.float64)
        f2 = ((k2 + 1023).astype(np.uint64) << np.uint64(52)).view(np.float64)
        out = (ex * f1) * f2
        out = np.where(over, np.inf, np.where(under, 0.0, out))
        out = np.where((y == 0.0) | (x == 1.0), 1.0, out)
    if odd.any():
        flat, fx, fy = out.reshape(-1), x.reshape(-1), y.reshape(-1)
        for i in np.nonzero(odd.reshape(-1))[0]:
            flat[i] = pow(float(fx[i]), float(fy[i]))
        out = flat.reshape(out.shape)
    return out
