#!/bin/bash
# builds the diagnostic (stamped) binary of the multi-tap weight-gradient kernel into /tmp/wgrad3_stamp
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -DLO_STAMPS"
for f in lo_wgrad3 lo_wgrad2 lo_norm lo_conv3 lo_conv; do hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/$f.hip -o /tmp/sw_$f.o; done
hipcc $F -c $R/lunaris_orion_amd/csrc/lo_util.cpp -o /tmp/sw_util.o
hipcc $F -x hip -c $R/tools/wgrad3_stamp.cpp -o /tmp/sw_main.o
hipcc --offload-arch=gfx950 /tmp/sw_lo_wgrad3.o /tmp/sw_lo_wgrad2.o /tmp/sw_lo_norm.o /tmp/sw_lo_conv3.o /tmp/sw_lo_conv.o /tmp/sw_util.o /tmp/sw_main.o -o /tmp/wgrad3_stamp
