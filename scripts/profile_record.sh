# Everything DESIGN.md quotes, on ONE box, kept under gpurun_out/<tag>_* (then: python scripts/summarize_profile.py <tag>;
# python scripts/collect_record.py <tag>).  Usage on the GPU box: bash scripts/profile_record.sh <tag>
# Optional second argument: the stages to run, e.g. "1 2" (default: all of 1..7) -- one gpurun call is limited to 20 minutes.
set -e
tag=$1
stages=" ${2:-1 2 3 4 5 6 7 8} "
st() { case "$stages" in *" $1 "*) return 0;; *) return 1;; esac; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
# 1. headline bench + its rocprofv3 kernel trace + PMC passes (separate runs, as the microarchitecture guide prescribes)
if st 1; then
python bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err            # the driver's default invocation: 200 steps
echo "bench done"
# the SAME command under the profiler (minus the host-side CPU baseline): headline + the 10 M-row secondary block, whose bf16 launches
# are other template instantiations of the kernels and get rows of their own in the statistics
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -- python3 bench.py --no-cpu-baseline > $o/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_fetch -- python3 bench.py --scale-10m off --extra-blocks off --steps 5 --warmup 2 --no-cpu-baseline > $o/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_pmc_write -- python3 bench.py --scale-10m off --extra-blocks off --steps 5 --warmup 2 --no-cpu-baseline > $o/${tag}_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $o/${tag}_pmc_l2 -- python3 bench.py --scale-10m off --extra-blocks off --steps 5 --warmup 2 --no-cpu-baseline > $o/${tag}_pmc_l2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/${tag}_pmc_sq -- python3 bench.py --scale-10m off --extra-blocks off --steps 5 --warmup 2 --no-cpu-baseline > $o/${tag}_pmc_sq.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_pmc_grbm -- python3 bench.py --scale-10m off --extra-blocks off --steps 5 --warmup 2 --no-cpu-baseline > $o/${tag}_pmc_grbm.log 2>&1 || true
echo "pmc done"
fi
# 2. the other configurations, each as bench line + rocprofv3 kernel stats of the same command
if st 2; then
for v in "sync:--async-tail 0" "q1:--queries 1" "q70:--queries 70" "bf16:--image-dtype bf16" "aqe:--with-aqe" "aqe_rparis:--workload rparis6k+1m --with-aqe" "10m:--workload 10m --steps 10 --warmup 2"; do
  name=${v%%:*}; args=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_${name}_trace -- python3 bench.py --scale-10m off --extra-blocks off --no-cpu-baseline $args > $o/${tag}_${name}_bench.json 2> $o/${tag}_${name}.err || echo "$name failed"
  echo "$name done"
done
fi
# 3. timelines (per-dispatch) of the full gallery and of a 1/8 shard; scripts
if st 3; then
bash scripts/timeline.sh ${tag}_full --async-tail 0 > $o/${tag}_timeline_full.txt 2>&1 || true
bash scripts/timeline.sh ${tag}_s8 --rows 125750 --async-tail 0 > $o/${tag}_timeline_s8.txt 2>&1 || true
bash scripts/timeline.sh ${tag}_q1 --queries 1 > $o/${tag}_timeline_q1.txt 2>&1 || true
bash scripts/timeline.sh ${tag}_q70 --queries 70 > $o/${tag}_timeline_q70.txt 2>&1 || true
python scripts/rank_all_timing.py > $o/${tag}_rank_all.txt 2>&1 || true
python scripts/host_api_rate.py > $o/${tag}_host_api.txt 2>&1 || true
python scripts/gallery_io_rate.py > $o/${tag}_gallery_io.json 2> /dev/null || true
python scripts/xcc_report.py > $o/${tag}_xcc_report.txt 2>&1 || true
python -m pytest tests/test_gpu_diffusion.py -q -s -k reference_size > $o/${tag}_diffusion_refsize.txt 2>&1 || true
python scripts/ladder_probe.py > $o/${tag}_ladder_probe.txt 2> /dev/null || true
(for g in 2 4 8; do python scripts/shard_step_model.py $g 1005994 rescore_grid_x=$((96 / g)) 2> /dev/null | tail -2; done) > $o/${tag}_shard_model.txt || true
(for f in "" "--force-protocol"; do python bench.py --scale-10m off --no-cpu-baseline --rows 125750 $f 2> /dev/null | tail -1 | cut -c1-700; done) > $o/${tag}_protocol_rccl1.txt || true
bash scripts/layout_model.sh > $o/${tag}_layout_model.txt 2> /dev/null || true
echo "scripts done"
fi
# 4. kernel A/B driver and MFMA probe (C++, no torch)
if st 4; then
bash scripts/kbench_build.sh > /dev/null 2>&1 || true    # the driver shares struct layouts with the library: never run a stale one
(cd image-search-engine-for-historical-research_amd && ./build/mfma_probe > ../$o/${tag}_mfma_probe.txt 2>&1 || true)
KB="default:0 structure1:0:1 zc:0:5 zc_inter:0:6 inter:0:7 g_nt:0:10 g_sc1:0:11 g_sc0sc1:0:12 both_nt:0:13 q_nt:0:14 g_sc0:0:16 pairs:0:20 pairs_nt:0:21 nofilter:4 nofilter_s1:4:1 nofilter_noA:36 nofilter_noB:68 noB_Bonce:16452 directB_model:49220 nodma:5 nodma_nofrag:133 filter_stamps:2048 stamps_inter:2048:6"
(cd image-search-engine-for-historical-research_amd && timeout -k 10 400 ./build/kbench --rounds 4 --reps 5 $KB > ../$o/${tag}_kbench.txt 2>&1 || true)
# fabric traffic and L2 hit rate per variant (round 4: which operand's DMA costs what, cache policies, paired-XCD walk)
(timeout -k 10 400 bash scripts/kbench_pmc.sh ${tag} default:0 g_nt:0:10 g_sc1:0:11 q_nt:0:14 pairs_nt:0:21 nofilter:4 nofilter_noA:36 nofilter_noB:68 directB_model:49220 > $o/${tag}_kbench_pmc.log 2>&1 || true)
(timeout -k 10 200 python scripts/gap_probe.py 2> /dev/null | grep -v amdgpu > $o/${tag}_gap_probe.txt || true)
(timeout -k 10 300 python scripts/first_launches.py 40 2> /dev/null | grep -v amdgpu > $o/${tag}_first_launches.txt || true)
(timeout -k 10 900 python scripts/shard_model_10m.py > $o/${tag}_shard_model_10m.txt 2>&1 || true)
(cd image-search-engine-for-historical-research_amd && timeout -k 10 200 ./build/tile4_probe --rounds 3 --reps 5 > ../$o/${tag}_tile4_probe.txt 2>&1 || true)
(cd image-search-engine-for-historical-research_amd && timeout -k 10 200 ./build/tailbench --reps 50 > ../$o/${tag}_tailbench.txt 2>&1 || true)
(cd image-search-engine-for-historical-research_amd && for t in 0.0663 0.0700 0.0760; do timeout -k 10 100 ./build/kbench --rounds 2 --reps 5 --thr $t default:0 | tail -1; done > ../$o/${tag}_kbench_thr.txt 2>&1 || true)
fi
# 5. multi-rank rehearsal of bench.py (ranks share the GPU, gloo)
if st 5; then
bash scripts/rehearse_sharded.sh 2 > $o/${tag}_rehearse2.txt 2>&1 || true
bash scripts/rehearse_sharded.sh 4 > $o/${tag}_rehearse4.txt 2>&1 || true
bash scripts/rehearse_sharded.sh 4 --layout 1x4 --pipeline > $o/${tag}_rehearse4_rows_pipelined.txt 2>&1 || true
# bench.py --gpus 2 started BARE (it starts its two ranks itself); ranks share the GPU over gloo; secondary block at 600 k rows
(ISEHR_DIST_BACKEND=gloo ISEHR_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --rows 200000 --steps 6 --warmup 2 --no-cpu-baseline --multi-gpu-blocks on --scale-10m on --scale-10m-rows 600000 --scale-10m-steps 4 2> /dev/null | tail -1 | cut -c1-6000) > $o/${tag}_bare_gpus2.txt || true
(ISEHR_DIST_BACKEND=gloo ISEHR_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 4 --rows 400000 --steps 6 --warmup 2 --no-cpu-baseline --multi-gpu-blocks on --scale-10m on --scale-10m-rows 1200000 --scale-10m-steps 4 2> /dev/null | tail -1 | cut -c1-6000) > $o/${tag}_bare_gpus4.txt || true
(timeout -k 10 300 python bench.py --graph --no-cpu-baseline --async-tail 0 2> /dev/null | tail -1 | cut -c1-900) > $o/${tag}_graph_replay.txt || true
fi
# 6. same-box A/B against the library of the previous round, when a build of it was left under ab/ (git-ignored)
if st 6; then
if [ -f ab/lib_r02.so ]; then
  cp image-search-engine-for-historical-research_amd/libmi355_retrieval.so ab/lib_now.so
  (for a in "" "--queries 70" "--queries 1"; do echo "# bench.py $a"; bash scripts/ab.sh ab/lib_r02.so ab/lib_now.so $a 2> /dev/null; done) > $o/${tag}_ab_r02.txt || true
fi
fi
# 7. fuzzing: the shape sweep with four other seeds
if st 7; then
(for sd in 101 102 103 104; do ISEHR_SWEEP_SEED=$sd timeout -k 10 600 python -m pytest tests/test_gpu_shape_sweep.py -q 2>&1 | tail -1; done) > $o/${tag}_sweep_seeds.txt || true
fi
# 8. the gallery ingest kernels alone (round 5): times, probes and -- in separate passes -- the fabric read / write traffic of one launch
# of each layout (rocprofv3 --pmc with --kernel-trace only; the program itself after --)
if st 8; then
P=image-search-engine-for-historical-research_amd/build
for pr in 128 134 130 132 640; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=$pr scripts/ingestbench.hip -o $P/ingestbench$pr 2> /dev/null; done
(for l in rows cols cols:1006016; do for pr in 128 134 130 132; do timeout -k 10 60 $P/ingestbench$pr 1005994 2048 $l 2>&1 | grep probe | tail -1; done; done; timeout -k 10 60 $P/ingestbench640 1005994 2048 rows 2>&1 | grep probe | tail -1) > $o/${tag}_ingestbench.txt 2>&1 || true
for l in rows cols; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_ingest_${l}_fetch -- $P/ingestbench128 1005994 2048 $l > $o/${tag}_ingest_${l}_fetch.log 2>&1 || true
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_ingest_${l}_write -- $P/ingestbench128 1005994 2048 $l > $o/${tag}_ingest_${l}_write.log 2>&1 || true
done
python scripts/ingest_pmc_report.py $tag > $o/${tag}_ingest_pmc.txt 2>&1 || true
echo "ingest done"
fi
echo "all done"
# 9. round 6: the whitening kernel (f64 MFMA) -- kernel statistics and, in separate passes, matrix-pipe busy cycles and fabric traffic
if st 9; then
W="--steps 10 --warmup 2 --scale-10m off --no-cpu-baseline --blocks whiten"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_whiten_trace -- python3 bench.py $W > $o/${tag}_whiten_bench.json 2> $o/${tag}_whiten.err || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_whiten_pmc_sq -- python3 bench.py $W > $o/${tag}_whiten_pmc_sq.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_whiten_pmc_fetch -- python3 bench.py $W > $o/${tag}_whiten_pmc_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_whiten_pmc_write -- python3 bench.py $W > $o/${tag}_whiten_pmc_write.log 2>&1 || true
python3 - <<PY > $o/${tag}_whiten_pmc.txt 2>&1 || true
import csv, glob
def per_kernel(d, counter):
    acc = {}
    for f in glob.glob("gpurun_out/${tag}_whiten_pmc_%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "whiten_mfma" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    return acc
for d, c, note in (("sq", "SQ_VALU_MFMA_BUSY_CYCLES", ""), ("sq", "SQ_BUSY_CYCLES", ""), ("sq", "GRBM_GUI_ACTIVE", ""),
                   ("fetch", "FETCH_SIZE", " (KiB; x 2 for a wide read stream on gfx950)"), ("write", "WRITE_SIZE", " (KiB)")):
    for k, v in per_kernel(d, c).items():
        big = [x for x in v if x >= 0.5 * max(v)]          # the launches at dims = 2048 (the dims = 128 ones are 1/16 of the work)
        print("%s %s: max %.4e, mean of the large launches %.4e (%d of %d launches)%s" % (k[:60], c, max(v), sum(big) / len(big), len(big), len(v), note))
PY
echo "whiten done"
fi
