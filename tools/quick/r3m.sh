#!/bin/bash
# round 3: workgroup-count targets of the two-pass transforms (SEALHIP_NTT_WG_TARGET)
set -u
export TMPDIR=/tmp
ROUNDS=2 tools/quick/ab_multi.sh cur wg16k wgall
