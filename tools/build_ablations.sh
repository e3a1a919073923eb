#!/bin/bash
# builds beat_amd/libbeat_amd_abl.so: the library with the timing-only variants of the k_gfstack_runs programs
# (GR_ABLATIONS=1; BEATAMD_GR_VAR=<n> picks one, results are wrong by construction) next to the shipped build, and
# restores the committed include.  Use: BEATAMD_LIB=$PWD/beat_amd/libbeat_amd_abl.so python tools/time_ml.py --envs ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
GR_ABLATIONS=1 python $R/tools/gen_gfruns_asm.py
make -C $R/beat_amd/csrc OBJDIR=build_abl${GR_NCONS} TARGET=../libbeat_amd_abl${GR_NCONS}.so 2>&1 | grep -E "error|warning: unused|Error" || true
GR_NCONS= python $R/tools/gen_gfruns_asm.py
touch $R/beat_amd/csrc/gfruns_asm.inc
