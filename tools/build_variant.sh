#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: a tuning build of the library into nbody-simulation_amd/lib/var/NAME.so
set -e
cd "$(dirname "$0")/../nbody-simulation_amd"
name=$1; shift
tmp=$(mktemp -d)
mkdir -p lib/var
for f in $(ls csrc/*.hip | xargs -n1 basename | sed "s/\.hip$//"); do
  c=-ffp-contract=off; [ $f = pm_fft ] && c=-ffp-contract=fast
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $c -I../include -Icsrc -Wno-unused-result -Wno-unused-value "$@" -c csrc/$f.hip -o $tmp/$f.o 2> $tmp/$f.log &
done
wait
for f in $(ls csrc/*.hip | xargs -n1 basename | sed "s/\.hip$//"); do
  [ -f $tmp/$f.o ] || { echo "build_variant: csrc/$f.hip did not compile:"; grep -m3 "error" $tmp/$f.log | cut -c1-300; rm -rf $tmp; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o lib/var/$name.so $tmp/*.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm -rf $tmp
