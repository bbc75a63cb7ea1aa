#!/bin/bash
# tools/exp/abshape.sh "<D C P [fft]>" rounds lib... : time_shape on the same box for several libraries
SH=$1; N=$2; shift 2
for r in $(seq $N); do for lib in "$@"; do echo -n "$(basename $lib) "; MDT_LIB=$PWD/$lib python tools/time_shape.py $SH 2>/dev/null | tail -1; done; done
