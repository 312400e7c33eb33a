#!/bin/bash
export TMPDIR=/tmp
for abl in 0 1 2 3 4 7 8 16 19 23 31; do P2L_CONV_FORCE=0 P2L_ABL=$abl python tools/conv_abl.py 2>&1 | grep ABL; done
