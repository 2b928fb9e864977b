#!/bin/bash
# run tools/dbg_case.py with the tree's library and with ab_libs/lib_old.so:  bash tools/ab_case.sh "<cfg dict>"
cp titanet_amd/libtitanet_amd.so /tmp/lib_new.so
echo "== new"; python tools/dbg_case.py "$1" 2>&1 | grep -v Warn | tail -8
cp ab_libs/lib_old.so titanet_amd/libtitanet_amd.so
echo "== old"; python tools/dbg_case.py "$1" 2>&1 | grep -v Warn | tail -8
cp /tmp/lib_new.so titanet_amd/libtitanet_amd.so
