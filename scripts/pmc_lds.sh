#!/bin/bash
# LDS behaviour of the patch sweep: bank-conflict cycles against all LDS cycles (rocprofv3 --pmc, its own pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pl && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/pl.err || { tail -5 /tmp/pl.err; exit 1; }
python3 scripts/pmc_summary.py /tmp/pl | grep -A4 "k_ras_sweep\|k_arap_local\|k_assoc_local"
