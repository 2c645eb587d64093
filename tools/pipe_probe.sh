#!/bin/bash
# one-box probe of what limits the fused pipeline kernel: band height sweep (2 alternating rounds) + PMC groups
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$1" python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling --filter pipeline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['achieved']))"; }
for r in 1 2; do for v in 24 32 48 64 108 200; do run MI355_TUNE_BAND_ROWS=$v; done; done
for f in pipeline sobel; do
  $ROOT/tools/pmc_quick.sh q1_$f SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -- --filter $f
  $ROOT/tools/pmc_quick.sh q2_$f SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -- --filter $f
  $ROOT/tools/pmc_quick.sh q3_$f SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- --filter $f
done
