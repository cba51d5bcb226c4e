#!/bin/bash
# PMC passes over tools/time_density.py: LDS / VALU activity of the density kernels -> gpurun_out/pmc_density_<W>.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT -d $R/gpurun_out/pda_$W --output-format csv -- python3 $R/tools/time_density.py $W > $R/gpurun_out/pda_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pdb_$W --output-format csv -- python3 $R/tools/time_density.py $W > $R/gpurun_out/pdb_$W.log 2>&1
  (cd $R && python3 tools/pmc_summary.py gpurun_out/pda_$W gpurun_out/pdb_$W > gpurun_out/pmc_density_$W.json)
  rm -rf $R/gpurun_out/pda_$W $R/gpurun_out/pdb_$W
  echo "$W done"
done
