set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $o/r04f_gputests.txt 2>&1 || (tail -40 $o/r04f_gputests.txt; exit 1)
tail -2 $o/r04f_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_record.sh r04f "1"
