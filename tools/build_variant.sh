#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  -> build_ab/lib_<name>.so (git-ignored, travels to the GPU box; build_exp/ does not) (experimental builds of the library;
# run them with OPUSGPU_LIB=$PWD/build_ab/lib_<name>.so)
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p build_ab
C=esp32-opus-player_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $C/og_api.hip $C/og_recon.hip $C/og_parse64.hip $C/og_rfc.hip $C/og_silk_nb.hip $C/og_silk_synth.hip $C/og_compat.cpp $C/og_pages.cpp -I include -pthread \
  $(cat $C/BUILD_FLAGS) "$@" -o build_ab/lib_$name.so
