#!/bin/bash
# usage (on this container): bash scripts/ab.sh <name> "<extra hipcc flags>"  -> cellranger_amd/variants/libcrgpu_<name>.so
name=$1; shift
mkdir -p cellranger_amd/variants
CRGPU_EXTRA_FLAGS="$1" python -m cellranger_amd.build --force > /dev/null 2>&1 || CRGPU_EXTRA_FLAGS="$1" python -m cellranger_amd.build
cp cellranger_amd/libcrgpu.so cellranger_amd/variants/libcrgpu_$name.so
