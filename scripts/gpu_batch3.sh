set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/gap_probe.py > gpurun_out/r04a_gap_probe.txt 2>&1
