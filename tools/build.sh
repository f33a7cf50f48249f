#!/usr/bin/env bash
# build everything in-tree; non-zero exit (and the compiler's errors) when anything fails
set -e
cd "$(dirname "$0")/.."
make -C frosting_amd/csrc -j8 2>&1 | grep -E "error|Error" -A4 && { echo "BUILD FAILED"; exit 1; } || true
python setup.py -q build_ext --inplace > /tmp/build_ext.log 2>&1 || { tail -20 /tmp/build_ext.log; echo "BUILD FAILED"; exit 1; }
# (tool only: the stand-in copy kernel of tools/combine_bench.py --side-copy)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/side_copy.hip -o tools/micro/libside_copy.so > /tmp/build_side_copy.log 2>&1 || echo "(tools/micro/side_copy.hip did not build: see /tmp/build_side_copy.log)"
python -c "
import sys; sys.path.insert(0, '.')
from frosting_amd import _lib
L = _lib.lib(); print('library ok, version', L.frg_version())"
