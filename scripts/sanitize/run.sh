#!/bin/bash
# AddressSanitizer + UBSan over the library's host code (no GPU needed): tree builder, loaders, solve.
set -e
cd "$(dirname "$0")/../.."
g++ -std=c++17 -O1 -g -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -Iinclude -Iicp_amd/csrc scripts/sanitize/host_driver.cc icp_amd/csrc/vptree_build.cc icp_amd/csrc/io.cc \
    icp_amd/csrc/align3.cc -o /tmp/icp_host_sanitize
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/icp_host_sanitize
