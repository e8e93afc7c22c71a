#!/bin/bash
# tools/ab_build.sh NAME "EXTRA flags": rebuild kq_full16k.hip with the flags and keep the library as ab/NAME.so
cd /root/repo/ka9q_sdr_amd/csrc && touch kq_full16k.hip && make EXTRA="$2" 2>&1 | grep -E "error|Error" -A3
cp /root/repo/ka9q_sdr_amd/lib/libka9q_hip.so /root/repo/ab/$1.so
