#!/usr/bin/env bash
# usage: scripts/build_variant.sh name "-DFLAG ..."   -> scripts/ubench/lib_<name>.so
# (the flags go to rr_fused.hip and rr_kernels.hip, where the RR_V_* switches live)
set -e
cd "$(dirname "$0")/.."
python radiorust_amd/build.py >/dev/null
hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 $2 -c radiorust_amd/csrc/rr_fused.hip -o /tmp/rr_fused_$1.o
hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 $2 -c radiorust_amd/csrc/rr_kernels.hip -o /tmp/rr_kernels_$1.o
hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 $2 -c radiorust_amd/csrc/rr_filter_ols.hip -o /tmp/rr_filter_ols_$1.o
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/ubench/lib_$1.so radiorust_amd/lib/rr_design.cpp.o /tmp/rr_kernels_$1.o /tmp/rr_fused_$1.o /tmp/rr_filter_ols_$1.o radiorust_amd/lib/rr_decim.hip.o radiorust_amd/lib/rr_metering.hip.o radiorust_amd/lib/rr_f64.hip.o radiorust_amd/lib/rr_api.hip.o
echo scripts/ubench/lib_$1.so
