#!/bin/bash
# builds the diagnostic (stamped) binary of the fused-tap conv kernel into tools/_bin/conv3_stamp (git-ignored; travels with gpurun)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/_bin
mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -DLO_STAMPS"
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_conv3.hip -o /tmp/s_conv3.o
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_conv.hip -o /tmp/s_conv.o
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_wgrad3.hip -o /tmp/s_wgrad3.o
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_wgrad2.hip -o /tmp/s_wgrad2.o
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_norm.hip -o /tmp/s_norm.o
hipcc $F -x hip -c $R/lunaris_orion_amd/csrc/lo_util.cpp -o /tmp/s_util.o
hipcc $F -x hip -c $R/tools/conv3_stamp.cpp -o /tmp/s_main.o
hipcc --offload-arch=gfx950 /tmp/s_conv3.o /tmp/s_conv.o /tmp/s_wgrad3.o /tmp/s_wgrad2.o /tmp/s_norm.o /tmp/s_util.o /tmp/s_main.o -o $O/conv3_stamp
