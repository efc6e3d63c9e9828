#!/bin/bash
set -u
export TMPDIR=/tmp
ROUNDS=2 tools/quick/ab_multi.sh cur wg16k wgall
