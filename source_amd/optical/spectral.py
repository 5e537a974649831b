"""
Spectrum and spectral functions (host side). The device never evaluates a SpectralFunction: each material's
function is resampled once per observe() pass into an f64[bins] table with the reference's bin-average rule
and uploaded with the render call.

Mirrors raysect/optical/spectrum.pyx and raysect/optical/spectralfunction.pyx (sample :171-217,
InterpolatedSF :383-470, ConstantSF :473-560) and the trapezium integrator
raysect/core/math/cython/utility.pyx:40-240 (find_index / lerp / integrate).
"""
import numpy as np


def _find_index(x, v):                                      # utility.pyx:40-92
    if v < x[0]:
        return -1
    top = len(x) - 1
    if v >= x[top]:
        return top
    bottom = 0
    mid = top // 2
    while top - bottom != 1:
        if v >= x[mid]:
            bottom = mid
        else:
            top = mid
        mid = (top + bottom) // 2
    return bottom


def _lerp(x0, x1, y0, y1, x):                               # utility.pxd:95-96
    return ((y1 - y0) / (x1 - x0)) * (x - x0) + y0


def _integrate(x, y, x0, x1):                               # utility.pyx:135-240
    if x1 <= x0:
        return 0.0
    lower = _find_index(x, x0) + 1
    upper = _find_index(x, x1)
    if upper == -1:
        return y[0] * (x1 - x0)
    top = len(x) - 1
    if lower > top:
        return y[top] * (x1 - x0)
    if lower > upper:
        m = (y[lower] - y[upper]) / (x[lower] - x[upper])
        y0 = m * (x0 - x[upper]) + y[upper]
        y1 = m * (x1 - x[upper]) + y[upper]
        return 0.5 * (y0 + y1) * (x1 - x0)
    total = 0.0
    if lower == 0:
        total += y[0] * (x[0] - x0)
    else:
        y0 = _lerp(x[lower - 1], x[lower], y[lower - 1], y[lower], x0)
        total += 0.5 * (y0 + y[lower]) * (x[lower] - x0)
    for i in range(lower, upper):
        total += 0.5 * (y[i] + y[i + 1]) * (x[i + 1] - x[i])
    if upper == top:
        total += y[top] * (x1 - x[top])
    else:
        y1 = _lerp(x[upper], x[upper + 1], y[upper], y[upper + 1], x1)
        total += 0.5 * (y[upper] + y1) * (x1 - x[upper])
    return total


class SpectralFunction:
    """raysect/optical/spectralfunction.pyx:45-237"""

    def __init__(self):
        self._sample_key = None
        self._sample_cache = None

    def evaluate(self, wavelength):
        raise NotImplementedError("Virtual method evaluate() not implemented.")

    __call__ = lambda self, wavelength: self.evaluate(wavelength)  # noqa: E731

    def integrate(self, min_wavelength, max_wavelength):
        raise NotImplementedError("Virtual method integrate() not implemented.")

    def average(self, min_wavelength, max_wavelength):     # :135-169, with the reference's one-entry cache (a dielectric asks per hit)
        key = (float(min_wavelength), float(max_wavelength))
        if getattr(self, "_average_key", None) != key:
            self._average_key, self._average_cache = key, self.integrate(min_wavelength, max_wavelength) / (max_wavelength - min_wavelength)
        return self._average_cache

    def sample_mv(self, min_wavelength, max_wavelength, bins):
        return self.sample(min_wavelength, max_wavelength, bins)

    def sample(self, min_wavelength, max_wavelength, bins):  # :171-217 (bin average = integral / bin width)
        key = (float(min_wavelength), float(max_wavelength), int(bins))
        if self._sample_key == key:
            return self._sample_cache
        samples = np.zeros(bins, dtype=np.float64)
        delta = (max_wavelength - min_wavelength) / bins
        lower = min_wavelength
        reciprocal = 1.0 / delta
        for i in range(bins):
            upper = min_wavelength + (i + 1) * delta
            samples[i] = reciprocal * self.integrate(lower, upper)
            lower = upper
        self._sample_key, self._sample_cache = key, samples
        return samples


class ConstantSF(SpectralFunction):
    """spectralfunction.pyx:473-560"""

    def __init__(self, value):
        super().__init__()
        self.value = float(value)

    def evaluate(self, wavelength):
        return self.value

    def integrate(self, min_wavelength, max_wavelength):
        return self.value * (max_wavelength - min_wavelength)

    def average(self, min_wavelength, max_wavelength):
        return self.value

    def sample(self, min_wavelength, max_wavelength, bins):
        return np.full(bins, self.value, dtype=np.float64)


class NumericallyIntegratedSF(SpectralFunction):
    """spectralfunction.pyx:330-413 — midpoint-rule integration of function() at sample_resolution nm."""

    def __init__(self, sample_resolution=1.0):
        super().__init__()
        if sample_resolution <= 0:
            raise ValueError("Sampling resolution must be greater than zero.")
        self.sample_resolution = float(sample_resolution)

    def evaluate(self, wavelength):
        return self.function(wavelength)

    def integrate(self, min_wavelength, max_wavelength):
        import math
        samples = max(int(math.ceil((max_wavelength - min_wavelength) / self.sample_resolution)), 1)
        total = 0.0
        delta = (max_wavelength - min_wavelength) / samples
        for i in range(samples):
            centre = min_wavelength + (0.5 + i) * delta
            total += self.function(centre) * delta
        return total

    def function(self, wavelength):
        raise NotImplementedError("Virtual method function() not implemented.")


class InterpolatedSF(SpectralFunction):
    """spectralfunction.pyx:383-470 — linear interpolation, nearest-neighbour extrapolation."""

    def __init__(self, wavelengths, samples, normalise=False):
        super().__init__()
        self.wavelengths = np.array(wavelengths, dtype=np.float64)
        self.samples = np.array(samples, dtype=np.float64)
        if self.wavelengths.ndim != 1:
            raise ValueError("Wavelength array must be 1D.")
        if self.samples.shape[0] != self.wavelengths.shape[0]:
            raise ValueError("Wavelength and sample arrays must be the same length.")
        order = np.argsort(self.wavelengths)
        self.wavelengths = self.wavelengths[order]
        self.samples = self.samples[order]
        if normalise:
            self.samples /= self.integrate(self.wavelengths.min(), self.wavelengths.max())

    def evaluate(self, wavelength):
        x, y = self.wavelengths, self.samples
        i = _find_index(x, wavelength)
        if i == -1:
            return float(y[0])
        if i == len(x) - 1:
            return float(y[-1])
        return float(_lerp(x[i], x[i + 1], y[i], y[i + 1], wavelength))

    def integrate(self, min_wavelength, max_wavelength):
        return float(_integrate([float(v) for v in self.wavelengths], [float(v) for v in self.samples], float(min_wavelength), float(max_wavelength)))


class Spectrum(SpectralFunction):
    """raysect/optical/spectrum.pyx — (min, max, bins) + f64 samples."""

    def __init__(self, min_wavelength, max_wavelength, bins):
        super().__init__()
        if bins < 1:
            raise ValueError("Number of bins cannot be less than 1.")
        if min_wavelength <= 0.0 or max_wavelength <= 0.0:
            raise ValueError("Wavelength cannot be less than or equal to zero.")
        if min_wavelength >= max_wavelength:
            raise ValueError("Minimum wavelength cannot be greater or equal to the maximum wavelength.")
        self.min_wavelength, self.max_wavelength, self.bins = float(min_wavelength), float(max_wavelength), int(bins)
        self.delta_wavelength = (self.max_wavelength - self.min_wavelength) / self.bins
        self.samples = np.zeros(self.bins, dtype=np.float64)

    @property
    def wavelengths(self):
        return np.array([self.min_wavelength + (0.5 + i) * self.delta_wavelength for i in range(self.bins)])

    def mul_scalar(self, value):                            # spectrum.pyx:449-453
        self.samples *= value

    def div_scalar(self, value):                            # spectrum.pyx:459-467: multiplies by the reciprocal
        self.samples *= 1.0 / value if value != 0.0 else float("inf")

    def add_array(self, array):
        self.samples += array

    def sub_array(self, array):
        self.samples -= array

    def mul_array(self, array):                             # spectrum.pyx:429-437
        self.samples *= array

    def mad_scalar(self, scalar, array):                    # spectrum.pyx:479-487: samples += scalar * array
        self.samples += scalar * np.asarray(array)

    def clear(self):
        self.samples[:] = 0.0

    def is_zero(self):
        return not self.samples.any()

    def is_compatible(self, min_wavelength, max_wavelength, bins):
        return self.min_wavelength == min_wavelength and self.max_wavelength == max_wavelength and self.bins == bins

    def new_spectrum(self):
        return Spectrum(self.min_wavelength, self.max_wavelength, self.bins)

    def copy(self):
        s = Spectrum(self.min_wavelength, self.max_wavelength, self.bins)
        s.samples[:] = self.samples
        return s

    def _wavelength_check(self, min_wavelength, max_wavelength):     # spectrum.pyx:105-111
        if min_wavelength <= 0.0 or max_wavelength <= 0.0:
            raise ValueError("Wavelength cannot be less than or equal to zero.")
        if min_wavelength >= max_wavelength:
            raise ValueError("Minimum wavelength cannot be greater or equal to the maximum wavelength.")

    def _attribute_check(self):                             # spectrum.pyx:113-120
        if self.samples is None:
            raise ValueError("Cannot generate sample as the sample array is None.")
        if self.samples.shape[0] != self.bins:
            raise ValueError("Sample array length is inconsistent with the number of bins.")

    # spectrum.pyx:202-300: the samples are point values at the bin CENTRES, integrated as the piecewise-linear curve through them,
    # extended by its end values (utility.pyx integrate) — and nothing is cached: `samples` is mutated in place by the *_scalar /
    # *_array operations, so a cached resampling would go stale. (The reference's Spectrum defines no evaluate().)
    def integrate(self, min_wavelength, max_wavelength):
        self._wavelength_check(min_wavelength, max_wavelength)
        self._attribute_check()
        return float(_integrate([float(v) for v in self.wavelengths], [float(v) for v in self.samples], float(min_wavelength), float(max_wavelength)))

    def average(self, min_wavelength, max_wavelength):
        return self.integrate(min_wavelength, max_wavelength) / (max_wavelength - min_wavelength)

    def sample(self, min_wavelength, max_wavelength, bins):
        self._wavelength_check(min_wavelength, max_wavelength)
        self._attribute_check()
        x, y = [float(v) for v in self.wavelengths], [float(v) for v in self.samples]
        samples = np.zeros(bins, dtype=np.float64)
        delta = (max_wavelength - min_wavelength) / bins
        lower = min_wavelength
        reciprocal = 1.0 / delta
        for i in range(bins):
            upper = min_wavelength + (i + 1) * delta
            samples[i] = reciprocal * _integrate(x, y, float(lower), float(upper))
            lower = upper
        return samples

    def total(self):
        return float(self.samples.sum() * self.delta_wavelength)
