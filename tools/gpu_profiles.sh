#!/bin/bash
# Everything under profiles/<tag>_* in one call on the GPU box (reduced afterwards by tools/pmc_traffic.py <tag>): kernel traces (headline, uniform, disk-1024, 1M / 128^3), the
# FETCH_SIZE / WRITE_SIZE passes (headline and uniform set), the SQ counter passes of the particle kernels and the transforms.
set -e
bash tools/make_profiles.sh
bash tools/make_profiles_other.sh
bash tools/pmc_deposit.sh > /dev/null
bash tools/pmc_fft.sh > /dev/null
ls gpurun_out/profiles_raw gpurun_out/profiles_other
