set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
V="default:0 g_nt:0:10 g_sc1:0:11 g_sc0sc1:0:12 both_nt:0:13 q_nt:0:14 g_nt_sc1:0:15 g_sc0:0:16 pairs:0:20 pairs_nt:0:21 nofilter:4 nofilter_noA:36 nofilter_noB:68 noB_Bonce:16452 directB_model:49220"
(cd image-search-engine-for-historical-research_amd && timeout -k 10 300 ./build/kbench --rounds 4 --reps 5 $V > ../$o/r04a_kbench.txt 2>&1)
echo kbench done
timeout -k 10 400 bash scripts/kbench_pmc.sh r04a $V > $o/r04a_kbench_pmc.log 2>&1
echo pmc done
timeout -k 10 300 python scripts/first_launches.py 60 > $o/r04a_first_launches.txt 2>&1
timeout -k 10 300 python scripts/first_launches.py 40 xcc_balance=0 > $o/r04a_first_launches_nobal.txt 2>&1
echo first done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/r04a_gputests.txt 2>&1
echo tests done
