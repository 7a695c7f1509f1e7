#!/bin/bash
# usage: tools/rebuild_unit.sh fno|pw|tiles|loss|ns2d   -- recompile ONE translation unit of csrc/ and relink (developer shortcut;
# torch_cfd_amd._lib.build_library is the build the driver runs)
set -e
cd "$(dirname "$0")/../torch-cfd_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
case "$1" in
  fno)  hipcc $F -c tcfd_fno.hip -o tcfd_fno.o ;;
  pw)   hipcc $F -c tcfd_fno_pw.hip -o tcfd_fno_pw.o ;;
  tiles) hipcc $F -c tcfd_fno_tiles.hip -o tcfd_fno_tiles.o ;;
  loss) hipcc $F -c tcfd_loss.hip -o tcfd_loss.o ;;
  ns2d) hipcc $F -DTCFD_UNIT=0 -c tcfd_ns2d.hip -o tcfd_ns2d.o & hipcc $F -DTCFD_UNIT=1 -c tcfd_ns2d.hip -o tcfd_ns2d_f32.o & wait ;;
  *) echo "unit?"; exit 2 ;;
esac
hipcc --offload-arch=gfx950 -shared -fPIC tcfd_ns2d.o tcfd_ns2d_f32.o tcfd_fno.o tcfd_fno_pw.o tcfd_fno_tiles.o tcfd_loss.o -o libtcfd_hip.so.tmp -Wl,-rpath,/opt/rocm/lib
mv libtcfd_hip.so.tmp libtcfd_hip.so
