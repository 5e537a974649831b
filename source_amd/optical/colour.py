"""
CIE XYZ projection of spectra (raysect/optical/colour.pyx:123-266). The colour matching functions are the CIE 1931 2-degree standard
observer tables in data/cie1931.npz (tools/make_cie_table.py); everything else is arithmetic restated in the reference's order.
"""
import os

import numpy as np

from .spectral import InterpolatedSF

_curves = None


def _ciexyz():
    global _curves
    if _curves is None:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cie1931.npz"))
        _curves = tuple(InterpolatedSF(d["wavelengths"], d[k]) for k in ("x", "y", "z"))
    return _curves


def resample_ciexyz(min_wavelength, max_wavelength, bins):
    """colour.pyx:123-153 — [bins, 3] bin averages of the X, Y, Z sensitivity curves."""
    if bins < 1:
        raise ValueError("Number of samples can not be less than 1.")
    if min_wavelength <= 0.0 or max_wavelength <= 0.0:
        raise ValueError("Wavelength can not be less than or equal to zero.")
    if min_wavelength >= max_wavelength:
        raise ValueError("Minimum wavelength can not be greater or equal to the maximum wavelength.")
    xyz = np.zeros((bins, 3))
    for c, curve in enumerate(_ciexyz()):
        xyz[:, c] = curve.sample(min_wavelength, max_wavelength, bins)
    return xyz


def spectrum_to_ciexyz(spectrum, resampled_xyz=None):
    """colour.pyx:158-187"""
    if resampled_xyz is None:
        resampled_xyz = resample_ciexyz(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins)
    x = y = z = 0.0
    for i in range(spectrum.bins):
        x += spectrum.delta_wavelength * spectrum.samples[i] * resampled_xyz[i, 0]
        y += spectrum.delta_wavelength * spectrum.samples[i] * resampled_xyz[i, 1]
        z += spectrum.delta_wavelength * spectrum.samples[i] * resampled_xyz[i, 2]
    return x, y, z


def _srgb_transfer(v):                                       # colour.pyx:223-232
    return 12.92 * v if v <= 0.0031308 else 1.055 * v ** 0.4166666666666667 - 0.055


def ciexyz_to_srgb(x, y, z):
    """colour.pyx:235-266 — XYZ (D65) to sRGB, clamped to [0, 1] (IEC 61966-2-1 matrix and transfer function)."""
    r = 3.2404542 * x - 1.5371385 * y - 0.4985314 * z
    g = -0.9692660 * x + 1.8760108 * y + 0.0415560 * z
    b = 0.0556434 * x - 0.2040259 * y + 1.0572252 * z
    return tuple(min(max(_srgb_transfer(c), 0.0), 1.0) for c in (r, g, b))
