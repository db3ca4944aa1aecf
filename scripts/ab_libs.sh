#!/bin/bash
# in turn, twice: a timing script with each library build/ab/libaoenv_<name>.so (A/B of kernel variants on one box)
#   bash scripts/ab_libs.sh "scripts/ab_pyr.py C3 0" main variant ...      bash scripts/ab_libs.sh scripts/time_papyrus.py main variant ...
CMD=$1; shift
for r in 1 2; do for n in "$@"; do AOENV_LIB=$PWD/build/ab/libaoenv_$n.so python $CMD | grep "measure()" | sed "s/^/$n  /"; done; done
