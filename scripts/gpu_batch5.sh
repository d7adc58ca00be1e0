set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $o/r04c_gputests.txt 2>&1 || (tail -40 $o/r04c_gputests.txt; exit 1)
tail -2 $o/r04c_gputests.txt
bash scripts/profile_record.sh r04c "1"
