#!/bin/bash
# Builds the Cython reference (raysect/source @ /root/reference) OUT OF TREE into $1 (default
# /tmp/rs_oracle) so that tests/golden/make_golden.py can import it and emit golden vectors.
# Development-container only: nothing produced here is committed or shipped; only the .npz
# vectors written by make_golden.py are. meson/meson-python are not installed, so a throw-away
# setuptools+cythonize driver is used (same compiler directive meson passes: meson.build:17).
set -euo pipefail
DEST=${1:-/tmp/rs_oracle}
mkdir -p "$DEST" && cd "$DEST"
if [ -d raysect ] && MPLBACKEND=Agg python3 -c "import sys; sys.path.insert(0,'.'); import raysect.primitive.mesh.mesh" 2>/dev/null; then
    echo "reference already built in $DEST"; exit 0
fi
cp -r /root/reference/raysect . && chmod -R u+w raysect
printf '__version__ = version = "0.0.0+oracle"\n' > raysect/_version.py
cat > setup.py <<'EOF'
import os
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy
exts = []
for root, dirs, files in os.walk("raysect"):
    for f in files:
        if f.endswith(".pyx"):
            path = os.path.join(root, f)
            exts.append(Extension(path[:-4].replace(os.sep, "."), [path], include_dirs=[numpy.get_include()],
                                  define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")],
                                  extra_compile_args=["-O2", "-w"]))
setup(name="raysect", ext_modules=cythonize(exts, nthreads=8, language_level=3, quiet=True,
      compiler_directives={"legacy_implicit_noexcept": True}))
EOF
python3 setup.py build_ext --inplace -j 8 > build.log 2>&1
MPLBACKEND=Agg python3 -c "import sys; sys.path.insert(0,'.'); from raysect.core.math.random import seed, uniform; seed(1234567890); assert uniform() == 0.8114659955555504; print('reference built OK')"
