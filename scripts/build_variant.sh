#!/usr/bin/env bash
# usage: scripts/build_variant.sh name "-DFLAG ..."   -> scripts/ubench/lib_<name>.so
# (the flags go to the kernel files where the RR_V_* switches live: rr_ols.hip, rr_ols_frame.hip, rr_fft_regs.hip, rr_bluestein.hip,
#  rr_channelizer.hip, rr_kernels.hip, rr_filter_ols.hip)
set -e
cd "$(dirname "$0")/.."
python radiorust_amd/build.py >/dev/null
OBJS=""
for f in rr_ols rr_ols_frame rr_ols_wave2k rr_ols_wg rr_fft_regs rr_bluestein rr_channelizer rr_kernels rr_filter_ols; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 $2 -c radiorust_amd/csrc/$f.hip -o /tmp/${f}_$1.o &
  OBJS="$OBJS /tmp/${f}_$1.o"
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/ubench/lib_$1.so radiorust_amd/lib/rr_design.cpp.o $OBJS radiorust_amd/lib/rr_decim.hip.o radiorust_amd/lib/rr_metering.hip.o radiorust_amd/lib/rr_f64.hip.o radiorust_amd/lib/rr_api.hip.o radiorust_amd/lib/rr_api_blocks.hip.o radiorust_amd/lib/rr_api_fourier.hip.o radiorust_amd/lib/rr_api_chain.hip.o
echo scripts/ubench/lib_$1.so
