#!/bin/bash
# in turn, twice: scripts/ab_pyr.py CONFIG 0 with each library build/ab/libaoenv_<name>.so given (A/B of kernel variants on one box)
C=$1; shift
for r in 1 2; do for n in "$@"; do AOENV_LIB=$PWD/build/ab/libaoenv_$n.so python scripts/ab_pyr.py $C 0 | sed "s/^/$n  /"; done; done
