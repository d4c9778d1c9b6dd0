#!/bin/bash
# Tuning build of the library (-DOCF_TUNING_BUILD: environment-driven kernel variants and phase-skipping switches, never shipped) into
# tools/tuning_lib/ next to the product build; tools/kbench picks it up through LD_LIBRARY_PATH (its rpath is a RUNPATH).
set -e
cd "$(dirname "$0")/.."
OUT=tools/tuning_lib; mkdir -p $OUT/obj
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=off -DOCF_TUNING_BUILD -I include -I ocflow_amd/csrc"
pids=()
for f in ocflow_amd/csrc/*.hip; do
  o=$OUT/obj/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find ocflow_amd/csrc include -name '*.h' -newer $o)" ]; then
    hipcc $FLAGS -c $f -o $o 2> $o.log &
    pids+=($!)
    if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libocflow_hip.so $OUT/obj/*.o
readelf -d tools/kbench | grep -i "runpath\|rpath"
echo built $OUT/libocflow_hip.so
