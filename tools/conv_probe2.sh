#!/bin/bash
for d in 0 7 15 31 14 6; do echo "== dbg $d"; ISB_CONV_DBG=$d SWEEP_VARIANTS=54,55 timeout -k 10 200 python tools/conv_probe.py 2>&1 | grep -E "^hw8 (384|3072)->2304 k1|^hw16 224" ; done
