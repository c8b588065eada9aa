#!/bin/bash
# Run gen_fst_golden.py inside the scratch reference environment (see make_scratch.py).
set -e
S=${LVSR_ORACLE_SCRATCH:-/tmp/lvsr_oracle_scratch}
HERE=$(cd "$(dirname "$0")" && pwd)
export THEANO_FLAGS=device=cpu,floatX=float32,cxx=,optimizer_excluding=fusion,base_compiledir=/tmp/theano_cc
export PYTHONDONTWRITEBYTECODE=1
export PYTHONPATH=$S/pkg:$S/shims
cd /tmp
exec python3 "$HERE/gen_fst_golden.py" "$@" 2> >(grep -v -e Warning -e "is 'default'" -e "is not" -e "^  if " -e "^  elif " -e "No PyFST" >&2)
