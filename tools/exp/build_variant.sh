#!/bin/bash
# A/B builds of the library with extra -D flags on selected translation units:
#   tools/exp/build_variant.sh NAME "k_mnw_f32 k_niw_f64" -DVBMP_FOO=1 ...   ->  tools/exp/ab/libvbmp_NAME.so
set -e
cd "$(dirname "$0")/../../pyvbmp_amd/csrc"
name=$1; units=$2; shift 2
out=../../tools/exp/ab; mkdir -p $out/obj_$name
objs=""
for f in *.hip; do
  u=${f%.hip}
  if [[ " $units " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-variable -ffp-contract=off "$@" -c $f -o $out/obj_$name/$u.o &
    objs="$objs $out/obj_$name/$u.o"
  else
    objs="$objs $u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvbmp_$name.so $objs
echo built $out/libvbmp_$name.so
