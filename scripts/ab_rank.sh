#!/bin/bash
# A/B of prebuilt library variants (variants/*.so) on one replayed rank of a 64M registration, ONE gpurun call:
# usage: scripts/ab_rank.sh <share> <iters>
cp icp_amd/lib/libicp_hip.so /tmp/libicp_hip.keep
for so in variants/*.so; do
  cp $so icp_amd/lib/libicp_hip.so
  timeout -k 10 500 python3 scripts/rank_replay.py 64000000 $1 spatial $2 2>&1 | grep "RECENTRE=1" | cut -c40-200 | sed "s|^|$so: |" || { cp /tmp/libicp_hip.keep icp_amd/lib/libicp_hip.so; exit 1; }
done
cp /tmp/libicp_hip.keep icp_amd/lib/libicp_hip.so
