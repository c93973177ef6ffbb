#!/bin/sh
# The PRODUCT binary's host side under AddressSanitizer (the HIP library as shipped: its kernels and host code are not
# instrumented; GPU sanitizers are not available on the pool). Builds popdel_amd/lib/popdel_asan; run it like `popdel`:
#   tools/asan_product.sh && ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 popdel_amd/lib/popdel_asan call profiles.txt -o out.vcf
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/popdel_amd/csrc/host
SRC="$H/main.cpp $H/histogram.cpp $H/profile_io.cpp $H/chrom_profile.cpp $H/call_params.cpp $H/merge_calls.cpp $H/vcf_writer.cpp $H/call_driver.cpp"
g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" -I"$H" -I"$ROOT/popdel_amd/csrc/hip" -o "$ROOT/popdel_amd/lib/popdel_asan" $SRC \
    -L"$ROOT/popdel_amd/lib" -lpopdel_hip -lz -lpthread -Wl,-rpath,'$ORIGIN'
