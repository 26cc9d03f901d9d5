#!/bin/bash
# instruction mix and memory pipe of PartRecog's k_label_nn (rocprofv3 --pmc, passes of their own): issue-bound, address-bound or latency-bound?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pl1 /tmp/pl2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pl1 -- python3 scripts/label_sweep.py > /dev/null 2> /tmp/pl.err || { tail -5 /tmp/pl.err; exit 1; }
python3 scripts/pmc_summary.py /tmp/pl1 | grep -A9 "k_label_nn"
rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pl2 -- python3 scripts/label_sweep.py > /dev/null 2> /tmp/pl.err || { tail -5 /tmp/pl.err; exit 1; }
python3 scripts/pmc_summary.py /tmp/pl2 | grep -A9 "k_label_nn"
