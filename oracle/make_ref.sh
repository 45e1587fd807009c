#!/bin/sh
# TEST INFRASTRUCTURE ONLY -- makes the reference's Python package available to the GPU box.
#
# /root/reference does not exist on the GPU box, and a Python reference cannot be "built" into a binary.  This recipe
# copies the reference package (elfi/ only: no docs, tests or notebooks) from where it lies to oracle/_ref/, which is
# git-ignored (never in history, never part of the product) but travels with the gpurun snapshot like the built .so
# files do.  tests/test_reference_loop_gpu.py then drives the REAL elfi.Rejection / elfi.BOLFI loops over the HIP
# objects there (oracle/ref_shim.py stubs the uninstallable third-party imports); without oracle/_ref those tests skip.
# Nothing under elfi_amd/ ever imports from here.
set -e
SRC="${ELFI_REFERENCE_SRC:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ ! -d "$SRC/elfi" ]; then
  echo "make_ref: no reference package under $SRC (GPU box?) -- keeping whatever oracle/_ref holds"
  exit 0
fi
rm -rf "$HERE/_ref"
mkdir -p "$HERE/_ref"
cp -r "$SRC/elfi" "$HERE/_ref/elfi"
find "$HERE/_ref" -name '__pycache__' -type d -prune -exec rm -rf {} +
echo "make_ref: copied $SRC/elfi -> $HERE/_ref/elfi ($(find "$HERE/_ref" -name '*.py' | wc -l) files)"
