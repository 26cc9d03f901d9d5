#!/bin/bash
# instruction mix of the kernels of one outer iteration (rocprofv3 --pmc, a pass of its own): issue-bound or latency-bound?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ps && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/ps -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt-solver --no-single-solve --no-cold-process --no-tolerance-headroom > /dev/null 2> /tmp/ps.err || { tail -5 /tmp/ps.err; exit 1; }
python3 scripts/pmc_summary.py /tmp/ps | grep -E "^k_ras_sweep|^k_arap_rhs|^k_smooth|^k_arap_finalize|^k_ras_prepare|^k_assoc_all|^k_assoc_prep"
