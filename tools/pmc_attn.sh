#!/bin/bash
# two PMC passes over tools/pmc_attn.py (old vs new attention.hip when tools/isa_out/attention_old.hip is present)
cd "$(dirname "$0")/.."
R=$PWD; out=$R/gpurun_out/pmc_attn; mkdir -p $out
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"
prof() { # tag what
  for i in 1 2 3; do
    eval "C=\$P$i"
    (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/$1_p$i -- python3 $R/tools/pmc_attn.py $2 > $out/$1_p$i.log 2>&1)
  done
  python3 tools/pmc_attn.py fold $out/$1_p1 $out/$1_p2 $out/$1_p3 > $out/$1.txt 2>&1
  rm -rf $out/$1_p1 $out/$1_p2 $out/$1_p3
}
W=${1:-fwd}
cp sea_amd/csrc/attention.hip /tmp/attention_new.hip
if [ -f tools/isa_out/attention_old.hip ] && [ "$W" = fwd ]; then
  cp tools/isa_out/attention_old.hip sea_amd/csrc/attention.hip; touch sea_amd/csrc/attention.hip; python -m sea_amd.build > /dev/null 2>&1
  prof old $W
  cp /tmp/attention_new.hip sea_amd/csrc/attention.hip; touch sea_amd/csrc/attention.hip; python -m sea_amd.build > /dev/null 2>&1
fi
prof new $W
cat $out/new.txt
