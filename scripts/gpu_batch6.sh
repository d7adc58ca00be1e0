set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
(cd image-search-engine-for-historical-research_amd && timeout -k 10 300 ./build/kbench --rotate --rounds 12 --reps 4 default:0 zc:0:5 g_sc0:0:16 g_sc0sc1:0:12 g_sc1:0:11 zc_sc0:0:17 zc_sc0sc1:0:18 > ../$o/r04c_kbench_rotated.txt 2>&1)
grep -v "^#" $o/r04c_kbench_rotated.txt
timeout -k 10 900 python scripts/shard_model_10m.py > $o/r04c_shard_model_10m.txt 2>&1 || true
tail -4 $o/r04c_shard_model_10m.txt
bash scripts/timeline.sh r04c_full --async-tail 0 > $o/r04c_timeline_full.txt 2>&1 || true
bash scripts/timeline.sh r04c_s8 --rows 125750 --async-tail 0 > $o/r04c_timeline_s8.txt 2>&1 || true
bash scripts/timeline.sh r04c_q1 --queries 1 > $o/r04c_timeline_q1.txt 2>&1 || true
bash scripts/timeline.sh r04c_q70 --queries 70 > $o/r04c_timeline_q70.txt 2>&1 || true
