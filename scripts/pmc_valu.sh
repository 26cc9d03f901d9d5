#!/bin/bash
# instruction mix of the association kernels: are they bound by issue slots or by latency? (rocprofv3 --pmc, its own pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pv && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/pv.err || { tail -5 /tmp/pv.err; exit 1; }
python3 scripts/pmc_summary.py /tmp/pv | grep -A9 "k_assoc_local\|k_assoc_heavy_knn\|k_arap_local"
