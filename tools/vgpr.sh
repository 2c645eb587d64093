#!/bin/bash
# tools/vgpr.sh <file.hip> [grep-pattern] — VGPRs / occupancy / spills of every kernel in one csrc file (cross-compiles, no GPU)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
F=$1; PAT=${2:-.}
cd $ROOT/opencl-development-real-time-image-processing_amd/csrc
B=$(basename $F .hip)
EXTRA=""
case " gauss_slide gauss_wide gauss_exact sobel_slide pipe_slide pipe_slide8 " in *" $B "*) EXTRA="$EXTRA -fno-slp-vectorize";; esac
case " gauss_mfma gauss_mfma_reg gauss_mfma_dma gauss_mfma_i8 " in *" $B "*) EXTRA="$EXTRA -mllvm -amdgpu-mfma-vgpr-form";; esac
case " pipe_slide pipe_slide8 gauss_exact " in *" $B "*) EXTRA="$EXTRA -mllvm -pragma-unroll-threshold=131072";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden $EXTRA \
  -Rpass-analysis=kernel-resource-usage -c $B.hip -o /tmp/vgpr_$B.o 2>&1 |
  grep -E "Function Name|    VGPRs:|Occupancy|VGPRs Spill|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  paste - - - - - | c++filt | sed -E 's/Function Name: //; s/mi355::\(anonymous namespace\):://; s/\(unsigned char const\*.*\)//' | grep -E "$PAT"
