#!/bin/bash
# Developer helper: build libsdrk.so variants with extra -D flags into lib_<name>/ for A/B runs on the GPU box.
#   tools/variant.sh name "-DFOO=1 -DBAR"      (sources: every .hip; objects under build_<name>/)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../sdr-iq-visualizer_amd/csrc"
make -s -j8 OBJDIR=../build_$NAME LIB=../lib_$NAME/libsdrk.so CXXFLAGS="-O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=on -Wall -Wno-unused-function $FLAGS"
