set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 300 python scripts/first_launches.py 60 > $o/r04a_first_launches.txt 2>&1
timeout -k 10 300 python scripts/first_launches.py 40 xcc_balance=0 > $o/r04a_first_launches_nobal.txt 2>&1
echo first done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/r04a_gputests.txt 2>&1
echo tests done
