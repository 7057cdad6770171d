#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  -> build_exp/lib_<name>.so (experimental builds of the library;
# run them with OPUSGPU_LIB=$PWD/build_exp/lib_<name>.so)
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p build_exp
C=esp32-opus-player_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $C/og_api.hip $C/og_recon.hip $C/og_leaves.hip $C/og_parse64.hip $C/og_rfc.hip $C/og_compat.cpp $C/og_pages.cpp -I include -pthread \
  $(cat $C/BUILD_FLAGS) "$@" -o build_exp/lib_$name.so
