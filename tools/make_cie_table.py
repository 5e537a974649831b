#!/usr/bin/env python3
"""Writes source_amd/optical/data/cie1931.npz: the CIE 1931 2-degree standard observer colour matching functions (x-bar, y-bar,
z-bar) at the wavelengths the reference tabulates them (raysect/optical/colour.pyx:39-88: 360-830 nm). Standard colorimetric data
(CIE 15:2004, table T.4), taken here from the compiled reference's module attributes so that resample_ciexyz() reproduces the
reference's bins bit for bit. Run in the development container (needs the out-of-tree reference build, see
tests/golden/build_reference.sh):  python tools/make_cie_table.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("RAYSECT_REF_BUILD", "/tmp/rs_oracle"))
from raysect.optical import colour  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "source_amd", "optical", "data", "cie1931.npz")
np.savez_compressed(out, wavelengths=np.array(colour.ciexyz_wavelength_samples, dtype=np.float64),
                    x=np.array(colour.ciexyz_x_samples, dtype=np.float64), y=np.array(colour.ciexyz_y_samples, dtype=np.float64),
                    z=np.array(colour.ciexyz_z_samples, dtype=np.float64))
print(out, os.path.getsize(out), "bytes")
